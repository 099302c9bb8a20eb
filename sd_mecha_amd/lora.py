"""Recipe-level LoRA support: ``convert(model(lora_file), base)`` and the key naming of kohya-flavoured LoRA files.

Mirrors, for the hot path only, the reference's ``LycorisToBase`` conversion (extensions/builtin/lycoris.py:52-85),
``StateDictKeyHelper`` (:88-95) and ``_to_lycoris_keys`` (:131-158) with the ``lora`` prefix of the kohya flavour (:21-24) and
the suffixes of ``LoraAlgorithm`` (:202-209).  The base checkpoint must use the kohya key naming (``unet.…``, ``te1.…``,
``te2.…``: the reference's ``sdxl-kohya`` / ``sd1-kohya`` configs); converting other layouts to it is the reference's
graph layer (out of scope).  ``lora_mid`` (Tucker) networks and the LoKr family raise NotImplementedError.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

from .merge_methods import StateDictKeyError
from .recipe_nodes import LiteralRecipeNode, MergeMethod, MergeRecipeNode, ModelRecipeNode, RecipeNode, merge_method

PREFIX = "lora"
UP, MID, DOWN, ALPHA = "lora_up.weight", "lora_mid.weight", "lora_down.weight", "alpha"


def lora_key_prefix(base_key: str, shape: Sequence[int], is_float: bool = True, prefix: str = PREFIX) -> Optional[str]:
    """``lora_<stem>`` of a base key, or None when the key gets no LoRA entry (lycoris.py:138-151: bias keys, non-float keys
    and keys with fewer than 2 dimensions are skipped; a trailing ``.weight`` is dropped; ``.`` becomes ``_``).  ``prefix`` is
    the flavour's (lycoris.py:21-24: ``lora`` for kohya files, ``lycoris`` for lycoris ones)."""
    if base_key.endswith("bias") or not is_float or len(shape) < 2:
        return None
    parts = base_key.split(".")
    if parts[-1] == "weight":
        parts = parts[:-1]
    return f"{prefix}_" + "_".join(parts)


def _convert_body(lora, **kwargs):
    raise RuntimeError("convert(lora, base) is evaluated by the scheduler (sd_mecha_amd.merging), not called per tensor")


# The reference registers one conversion per (base config, flavour) under the identifier  convert_'<base>_<flavour>_lora'_to_base
# (lycoris.py:19-52), with ONE argument, the LoRA file; `.mecha` files name it that way.  The same identifiers are registered here
# so that a LoRA recipe's text reads the same in both directions.  The base checkpoint the delta is made for is the `a` of the
# add_difference that consumes it: a node built by convert() below carries it as the attribute `lora_base`, a node that comes
# out of deserialize() gets it from bind_bases() when merge() starts.  `convert_lora_to_base` is this package's own name for
# a base whose config is not one of the three (synthetic checkpoints with kohya-style names).
FLAVOURS = {"kohya": "lora", "lycoris": "lycoris"}                  # lycoris.py:21-24
BASE_CONFIGS = ("sdxl-kohya", "sdxl-kohya_but_diffusers", "sd1-kohya")   # lycoris.py:29-33
GENERIC_IDENTIFIER = "convert_lora_to_base"


def conversion_identifier(base_config: Optional[str], flavour: str = "kohya") -> str:
    if base_config in BASE_CONFIGS:
        return f"convert_'{base_config}_{flavour}_lora'_to_base"
    return GENERIC_IDENTIFIER


_conversions: Dict[str, MergeMethod] = {}
for _flavour, _prefix in FLAVOURS.items():
    for _base in (*BASE_CONFIGS, None):
        _ident = conversion_identifier(_base, _flavour)
        if _ident not in _conversions:
            _m = merge_method(_convert_body, identifier=_ident)
            _m.lora_prefix, _m.lora_base_config = _prefix, _base
            _m.lora_config = f"{_base}_{_flavour}_lora" if _base is not None else None     # the LoRA file's own config (lycoris.py:115-117)
            _conversions[_ident] = _m
convert_lora_to_base: MergeMethod = _conversions[GENERIC_IDENTIFIER]


def is_conversion(node) -> bool:
    return isinstance(node, MergeRecipeNode) and getattr(node.merge_method, "lora_prefix", None) is not None


def base_of(node: MergeRecipeNode) -> ModelRecipeNode:
    base = getattr(node, "lora_base", None)
    if base is None:
        raise ValueError(f"{node.merge_method.identifier}: the base checkpoint of this LoRA conversion is unknown -- it must be the "
                         "delta of an add_difference(base, convert(...)) (bind_bases)")
    return base


def _config_of(base: ModelRecipeNode) -> Optional[str]:
    """The base node's config: explicit, or inferred from the file's key names if the file can be read now."""
    if base.config is not None:
        return base.config
    try:
        from . import model_configs
        from .streaming import InSafetensorsDict
        reader = InSafetensorsDict(base.path)
        try:
            cfg = model_configs.infer([reader.keys()], [None])
        finally:
            reader.close()
        return cfg.identifier if cfg is not None else None
    except (OSError, ValueError):
        return None


def convert(lora: RecipeNode, base: RecipeNode, flavour: str = "kohya") -> MergeRecipeNode:
    """``sd_mecha.convert(model(lora_file), base)`` (reference examples/lora_merge.py:6): the delta the LoRA file encodes for
    each key of ``base``, as a node of the reference's conversion method for the base's config (one argument: the file)."""
    if not isinstance(lora, ModelRecipeNode) or not isinstance(base, ModelRecipeNode):
        raise TypeError("convert(lora, base) takes two model nodes")
    if flavour not in FLAVOURS:
        raise ValueError(f"unknown LoRA flavour {flavour!r}: {sorted(FLAVOURS)}")
    node = MergeRecipeNode(_conversions[conversion_identifier(_config_of(base), flavour)], (lora,), {})
    node.lora_base = base
    return node


def bind_bases(recipe: RecipeNode) -> None:
    """Give every conversion node without a `lora_base` (deserialized recipes) the checkpoint it is a delta of: the first
    model reached from the `a` argument of the add_difference that consumes it."""
    seen = set()

    def first_model(n):
        while isinstance(n, MergeRecipeNode) and n.args:
            n = n.args[0]
        return n if isinstance(n, ModelRecipeNode) else None

    def walk(n):
        if not isinstance(n, MergeRecipeNode) or id(n) in seen:
            return
        seen.add(id(n))
        children = (*n.args, *n.kwargs.values())
        if n.merge_method.identifier == "add_difference" and len(n.args) >= 2 and is_conversion(n.args[1]) \
                and getattr(n.args[1], "lora_base", None) is None:
            n.args[1].lora_base = first_model(n.args[0])
        for c in children:
            walk(c)
    walk(recipe)


def network_entries(reader, prefix: str):
    """(up_key, down_key, alpha_key or None) of one base key inside a LoRA file; StateDictKeyError when the file holds no
    up/down pair for it (LoraAlgorithm.relation_inputs :216-218, LycorisToBase.__call__ :80-85)."""
    up, down, alpha, mid = f"{prefix}.{UP}", f"{prefix}.{DOWN}", f"{prefix}.{ALPHA}", f"{prefix}.{MID}"
    if up not in reader or down not in reader:
        raise StateDictKeyError(prefix)
    if mid in reader:
        raise NotImplementedError(f"{mid}: Tucker (lora_mid) networks are outside the MI355X path")
    return up, down, (alpha if alpha in reader else None)


# ---------------------------------------------------------------------------------------------- chains of LoRA applications
@dataclass
class LoraExpr:
    """The part of a recipe the LoRA fuser understands:  base | add_difference(expr, convert(L_i, base), alpha_i) |
    fallback(expr, expr)."""
    kind: str                                  # "base" | "apply" | "fallback"
    a: Optional["LoraExpr"] = None
    b: Optional["LoraExpr"] = None             # fallback default
    lora: int = -1                             # index into the plan's LoRA list
    alpha: float = 1.0


def parse_chain(recipe: RecipeNode):
    """(expr, base model node, [lora model nodes], [their key prefixes]) or None.  Every leaf must be the same base model."""
    loras: List[ModelRecipeNode] = []
    prefixes: List[str] = []
    base: List[ModelRecipeNode] = []
    memo: Dict[int, LoraExpr] = {}

    def same(a: ModelRecipeNode, b: ModelRecipeNode) -> bool:
        return a is b or a.path == b.path

    def scalar(node, default):
        if node is None:
            return default
        if isinstance(node, LiteralRecipeNode) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return float(node.value)
        return None

    def parse(n: RecipeNode) -> Optional[LoraExpr]:
        if id(n) in memo:
            return memo[id(n)]
        out = None
        if isinstance(n, ModelRecipeNode):
            if not base:
                base.append(n)
            out = LoraExpr("base") if same(n, base[0]) else None
        elif isinstance(n, MergeRecipeNode):
            ident = n.merge_method.identifier
            if ident == "fallback" and len(n.args) == 2 and not n.kwargs:
                a, b = parse(n.args[0]), parse(n.args[1])
                out = LoraExpr("fallback", a, b) if a is not None and b is not None else None
            elif ident == "add_difference":
                args = list(n.args)
                a_node = args[0] if args else n.kwargs.get("a")
                b_node = args[1] if len(args) > 1 else n.kwargs.get("b")
                alpha = scalar(args[2] if len(args) > 2 else n.kwargs.get("alpha"), 1.0)
                if is_conversion(b_node) and getattr(b_node, "lora_base", None) is not None and alpha is not None and a_node is not None:
                    a = parse(a_node)
                    lora_node, conv_base = b_node.args[0], b_node.lora_base
                    if a is not None and base and same(conv_base, base[0]):
                        pfx = b_node.merge_method.lora_prefix
                        idx = next((i for i, m in enumerate(loras) if same(m, lora_node) and prefixes[i] == pfx), None)
                        if idx is None:
                            loras.append(lora_node)
                            prefixes.append(pfx)
                            idx = len(loras) - 1
                        out = LoraExpr("apply", a, None, idx, alpha)
        memo[id(n)] = out
        return out

    expr = parse(recipe)
    if expr is None or not loras or not base:
        return None
    return expr, base[0], loras, prefixes


def networks_for_key(expr: LoraExpr, has_network) -> List[Tuple[int, float]]:
    """The (lora index, alpha) applications that produce one key, in order, following the reference's per-key evaluation:
    add_difference loads its delta first and raises StateDictKeyError when the LoRA file has no entry for the key
    (linear.py:64-76 via lycoris.py:80-85); ``fallback`` then takes its default (logistics.py:12-35).  ``has_network(i)``
    says whether LoRA ``i`` holds the key.  Raises StateDictKeyError when the whole key is skipped."""
    if expr.kind == "base":
        return []
    if expr.kind == "fallback":
        try:
            return networks_for_key(expr.a, has_network)
        except StateDictKeyError:
            return networks_for_key(expr.b, has_network)
    if not has_network(expr.lora):
        raise StateDictKeyError("no LoRA entry")
    nets = networks_for_key(expr.a, has_network)
    if expr.alpha == 0.0:                      # math.isclose(alpha, 0.0) (rel_tol only): add_difference returns `a` (linear.py:72-73)
        return nets
    return nets + [(expr.lora, expr.alpha)]
