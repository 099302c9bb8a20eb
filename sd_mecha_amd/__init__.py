"""sd_mecha_amd -- MI355X-native backend for sd-mecha's per-key merge hot path.

Public surface mirrors the reference for this path only: the builtin merge-method names
(callable on tensors = the method body, callable on recipe nodes = a recipe, ``.__wrapped__`` = the body),
``StateDictKeyError`` / ``NonFiniteStateDictKeyError``, ``model()``, ``literal()``, ``merge()``, ``serialize()``
and the ``@merge_method`` decorator for user-defined per-key methods.
"""
import collections as _collections

from . import merge_methods as _mm
from .merge_methods import NonFiniteStateDictKeyError, StateDictKeyError, dare_draw_masks  # noqa: F401
from .recipe_nodes import (  # noqa: F401
    Parameter,
    RealizedKeyRelation,
    Return,
    StateDict,
    deserialize,
    LiteralRecipeNode,
    MergeMethod,
    MergeRecipeNode,
    ModelRecipeNode,
    RecipeNode,
    literal,
    merge_method,
    model,
    resolve,
    serialize,
)


class _fallback:
    """fallback (logistics.py:12-35): the first of (`a`, `default`) -- in the order the key analysis stored in
    ``key_relation.meta`` -- whose value for the key can be produced."""

    def __call__(self, a, default, **kwargs):
        (key,), _ = relation = kwargs["key_relation"]
        candidates = {"a": a, "default": default}
        for param in relation.meta:
            try:
                return candidates[param][key]
            except StateDictKeyError:
                continue
        raise StateDictKeyError(key)


# `lazy`: parameters the reference declares as Parameter(StateDict[...]) (linear.py:8-9, :27, :65-66, :80; svd.py:11-12;
# logistics.py:23-24): the executor hands them over as lazy mappings, so `alpha == 0` never loads `b` (merging.py:579-589)
weighted_sum = merge_method(_mm.weighted_sum, lazy=("a", "b"))
n_average = merge_method(_mm.n_average, lazy=("models",))
add_difference = merge_method(_mm.add_difference, lazy=("a", "b"))
subtract = merge_method(_mm.subtract)
scale = merge_method(_mm.scale, lazy=("a",))
slerp = merge_method(_mm.slerp)
perpendicular_component = merge_method(_mm.perpendicular_component)
add_perpendicular = merge_method(_mm.add_perpendicular)
ties_sum = merge_method(_mm.ties_sum)
ties_sum_extended = merge_method(_mm.ties_sum_extended)
ties_sum_with_dropout = merge_method(_mm.ties_sum_with_dropout)
add_difference_ties = merge_method(_mm.add_difference_ties)
model_stock = merge_method(_mm.model_stock)
dropout = merge_method(_mm.dropout)
cast = merge_method(_mm.cast, reuse_outputs=False)
omit_non_finite = merge_method(_mm.omit_non_finite, reuse_outputs=False)
fallback = merge_method(_fallback, identifier="fallback", lazy=("a", "default"), reuse_outputs=False)
rotate = merge_method(_mm.rotate, lazy=("a", "b"), cache_factory=lambda: _collections.defaultdict(dict))   # svd.py:10
truncate_rank = merge_method(_mm.truncate_rank, cache_factory=dict)   # svd.py:89
lora_merge = merge_method(_mm.lora_merge)
lora_delta = merge_method(_mm.lora_delta)

from .lora import convert  # noqa: E402,F401   convert(model(lora_file), base): the reference's sd_mecha.convert for kohya LoRAs
from .merging import merge  # noqa: E402,F401

__version__ = "0.1.0"
