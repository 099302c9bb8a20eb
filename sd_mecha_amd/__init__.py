"""sd_mecha_amd -- MI355X-native backend for sd-mecha's per-key merge hot path.

Public surface mirrors the reference for this path only: the builtin merge-method names,
``StateDictKeyError`` / ``NonFiniteStateDictKeyError``, and (see ``merging.py``) ``model()`` / ``merge()``.
"""
from .merge_methods import (  # noqa: F401
    NonFiniteStateDictKeyError,
    StateDictKeyError,
    add_difference,
    add_difference_ties,
    add_perpendicular,
    cast,
    dare_draw_masks,
    n_average,
    omit_non_finite,
    perpendicular_component,
    scale,
    slerp,
    subtract,
    ties_sum,
    ties_sum_with_dropout,
    weighted_sum,
)

__version__ = "0.1.0"
