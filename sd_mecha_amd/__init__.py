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


def _fallback(a, default, **kwargs):
    """fallback (logistics.py:12-35): resolved by the executor (first input that has the key)."""
    return a if a is not None else default


weighted_sum = merge_method(_mm.weighted_sum)
n_average = merge_method(_mm.n_average)
add_difference = merge_method(_mm.add_difference)
subtract = merge_method(_mm.subtract)
scale = merge_method(_mm.scale)
slerp = merge_method(_mm.slerp)
perpendicular_component = merge_method(_mm.perpendicular_component)
add_perpendicular = merge_method(_mm.add_perpendicular)
ties_sum = merge_method(_mm.ties_sum)
ties_sum_extended = merge_method(_mm.ties_sum_extended)
ties_sum_with_dropout = merge_method(_mm.ties_sum_with_dropout)
add_difference_ties = merge_method(_mm.add_difference_ties)
cast = merge_method(_mm.cast)
omit_non_finite = merge_method(_mm.omit_non_finite)
fallback = merge_method(_fallback, identifier="fallback")
rotate = merge_method(_mm.rotate, cache_factory=lambda: _collections.defaultdict(dict))   # svd.py:10
lora_merge = merge_method(_mm.lora_merge)
lora_delta = merge_method(_mm.lora_delta)

from .lora import convert  # noqa: E402,F401   convert(model(lora_file), base): the reference's sd_mecha.convert for kohya LoRAs
from .merging import merge  # noqa: E402,F401

__version__ = "0.1.0"
