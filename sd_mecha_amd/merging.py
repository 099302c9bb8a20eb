"""The key scheduler: ``merge(recipe, ...)`` for the MI355X backend.

Replaces the executor of ``/root/reference/sd_mecha/merging.py`` (merge :34-239, KeyMergeVisitor :374-482,
CastInputDicts :639-697, _track_output :292-317) for the hot path:

* the reference submits ONE future per key to <= 4 threads (:171-179, :223-229); here the keys of this rank's
  stripe are packed into *waves* -- per-model arenas with identical offsets (arena.py) -- and every wave is
  merged by one batched launch sequence through the C ABI;
* inputs move mmap -> pinned staging -> ``hipMemcpyAsync`` on a copy stream, outputs come back on a third
  stream, double-buffered so H2D(w+1), compute(w) and D2H(w-1) overlap;
* the per-input ``cast`` + ``omit_non_finite`` nodes the reference wraps around every model (:673-681) and the
  ``check_finite_output`` test (:297-308) are fused into the kernels and come back as one flag word per key;
* recipes the fuser does not recognise (user ``@merge_method`` bodies, out-of-scope builtins) are evaluated key by
  key with the same Python bodies on ROCm tensors;
* with ``torch.distributed`` initialised, keys are dealt over ranks (LPT by size) and every rank writes its
  own region of the one output file -- no data-path collective.

Results-changing defaults are the reference's: merge_dtype fp32, output_dtype fp16, omit_non_finite_inputs,
check_finite_output (:112-146).
"""
from __future__ import annotations

import concurrent.futures
import logging
import math
import os
import pathlib
import threading
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Mapping, MutableMapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from . import lora as lora_mod
from . import merge_methods as mm
from . import model_configs
from .arena import ArenaLayout, plan_arena, shard_keys_lpt
from .merge_methods import StateDictKeyError
from .recipe_nodes import LiteralRecipeNode, MergeRecipeNode, ModelRecipeNode, RealizedKeyRelation, RecipeNode, serialize, value_to_node
from .streaming import (DTYPE_REVERSE_MAPPING, _DTYPE_BY_CODE, _HeaderTable, InSafetensorsDict, OutSafetensorsDict, TensorMetadata, compact_holes,
                        open_reader, tensor_bytes)

_DT_CODE = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
log = logging.getLogger("sd_mecha_amd")


# ================================================================== graph analysis
@dataclass
class FusedPlan:
    """A recipe the batched kernels cover end to end."""
    kind: str                                   # "wsum" | "navg" | "ties" | "lora" | "rotate" | "slerp" | "addperp" | "addsub" | "adddiff" | "dare"
    models: List[ModelRecipeNode]               # inputs in kernel order (ties: base first)
    params: Dict[str, Any] = field(default_factory=dict)
    fallback: Optional[ModelRecipeNode] = None  # `| model`: value of a key whose merge was skipped
    node: Optional[MergeRecipeNode] = None      # the merge node the plan stands for (merge(cache=...) is keyed by it)


def _scalar(node: RecipeNode, default=None):
    if node is None:
        return default
    if isinstance(node, LiteralRecipeNode) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
        return float(node.value)
    return None


def _bound(node: MergeRecipeNode) -> Dict[str, RecipeNode]:
    """Bind positional/keyword nodes to parameter names of the method body."""
    sig = node.merge_method.signature
    params = [p for p in sig.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    out = {}
    for p, a in zip(params, node.args):
        out[p.name] = a
    out.update(node.kwargs)
    return out


def fuse(recipe: RecipeNode) -> Optional[FusedPlan]:
    """Recognise the node chains of SURVEY.md §3.2-3.3 (wrappers.py shapes); None = evaluate generically."""
    lora_mod.bind_bases(recipe)
    chain = lora_mod.parse_chain(recipe)
    if chain is not None:                                                     # base + sum of LoRA deltas
        expr, base, loras, prefixes = chain
        return FusedPlan("lora", [base], {"expr": expr, "loras": loras, "prefixes": prefixes})
    fallback = None
    if isinstance(recipe, MergeRecipeNode) and recipe.merge_method.identifier == "fallback":
        a, d = recipe.args[0], recipe.args[1] if len(recipe.args) > 1 else recipe.kwargs.get("default")
        if isinstance(d, ModelRecipeNode):
            recipe, fallback = a, d
    if not isinstance(recipe, MergeRecipeNode):
        return None
    ident = recipe.merge_method.identifier
    if ident == "weighted_sum":
        b = _bound(recipe)
        alpha = _scalar(b.get("alpha"), 0.5) if "alpha" in b else 0.5
        if isinstance(b.get("a"), ModelRecipeNode) and isinstance(b.get("b"), ModelRecipeNode) and alpha is not None:
            return FusedPlan("wsum", [b["a"], b["b"]], {"alpha": alpha}, fallback)
    if ident == "n_average":
        if recipe.args and all(isinstance(a, ModelRecipeNode) for a in recipe.args) and not recipe.kwargs and len(recipe.args) <= _lib.MAX_MODELS:
            return FusedPlan("navg", list(recipe.args), {}, fallback)
    if ident == "slerp":                                                       # linear.py:43-61 on two models, scalar alpha
        b = _bound(recipe)
        alpha = _scalar(b.get("alpha"), 0.5) if "alpha" in b else 0.5
        if isinstance(b.get("a"), ModelRecipeNode) and isinstance(b.get("b"), ModelRecipeNode) and alpha is not None:
            return FusedPlan("slerp", [b["a"], b["b"]], {"alpha": alpha}, fallback)
    if ident == "add_perpendicular":                                           # this package's fused shim of wrappers.py:42-64
        b = _bound(recipe)
        alpha = _scalar(b.get("alpha"), 1.0) if "alpha" in b else 1.0
        ms = [b.get("a"), b.get("b"), b.get("c")]
        if all(isinstance(m, ModelRecipeNode) for m in ms) and alpha is not None and not math.isclose(alpha, 0.0):
            return FusedPlan("addperp", ms, {"alpha": alpha}, fallback)
    if ident == "add_difference":
        b = _bound(recipe)
        alpha = _scalar(b.get("alpha"), 1.0) if "alpha" in b else 1.0
        base, t = b.get("a"), b.get("b")
        plain = isinstance(base, ModelRecipeNode) and alpha is not None and not math.isclose(alpha, 0.0)
        if plain and isinstance(t, ModelRecipeNode):                           # linear.py:64-76 with a delta checkpoint
            return FusedPlan("adddiff", [base, t], {"alpha": alpha}, fallback)
        if plain and isinstance(t, MergeRecipeNode):
            tid, targs = t.merge_method.identifier, t.args
            if tid == "subtract" and len(targs) == 2 and all(isinstance(m, ModelRecipeNode) for m in targs):
                return FusedPlan("addsub", [base, targs[0], targs[1]], {"alpha": alpha}, fallback)          # wrappers.py:9-39 with `c`
            if tid == "perpendicular_component" and len(targs) == 2 and not t.kwargs:                      # wrappers.py:42-64
                da, db = targs
                if (all(isinstance(d, MergeRecipeNode) and d.merge_method.identifier == "subtract" and len(d.args) == 2
                        and all(isinstance(m, ModelRecipeNode) for m in d.args) for d in (da, db))
                        and _same_model(da.args[0], base) and _same_model(da.args[1], db.args[1])):
                    return FusedPlan("addperp", [base, db.args[0], da.args[1]], {"alpha": alpha}, fallback)
            if tid == "ties_sum_with_dropout":                                                            # wrappers.py:231-273
                dare = _dare_params(t, base)
                if dare is not None:
                    models, params = dare
                    return FusedPlan("dare", [base] + models, dict(params, alpha=alpha), fallback)
        if isinstance(base, ModelRecipeNode) and alpha is not None and isinstance(t, MergeRecipeNode) and t.merge_method.identifier == "ties_sum":
            k = _scalar(t.kwargs.get("k"), 1.0) if "k" in t.kwargs else 1.0
            vote = t.kwargs.get("vote_sgn")
            vote = bool(vote.value) if isinstance(vote, LiteralRecipeNode) else False
            models = []
            for d in t.args:
                if not (isinstance(d, MergeRecipeNode) and d.merge_method.identifier == "subtract" and len(d.args) == 2
                        and isinstance(d.args[0], ModelRecipeNode) and _same_model(d.args[1], base)):
                    return None
                models.append(d.args[0])
            if k is not None and models and len(models) < _lib.MAX_MODELS and not math.isclose(alpha, 0.0):
                return FusedPlan("ties", [base] + models, {"k": k, "alpha": alpha, "vote_sgn": vote}, fallback)
    if ident == "rotate":
        # svd.py:10-86 on two models with whole-number alignment and scalar alpha: keys of equal 2-D shape are rotated together
        # (one launch sequence per shape group); the degenerate settings that return an input untouched (svd.py:37-42) and
        # fractional alignment stay on the per-key path
        b = _bound(recipe)
        alignment = _scalar(b.get("alignment"), 1.0) if "alignment" in b else 1.0
        alpha = _scalar(b.get("alpha"), 0.0) if "alpha" in b else 0.0
        cz = _scalar(b.get("centralization"), 1.0) if "centralization" in b else 1.0
        if (isinstance(b.get("a"), ModelRecipeNode) and isinstance(b.get("b"), ModelRecipeNode) and None not in (alignment, alpha, cz)
                and any(math.isclose(alignment, w) for w in (-1.0, 0.0, 1.0))
                and not (math.isclose(alignment, 0.0) and math.isclose(alpha, 0.0))
                and not (math.isclose(alignment, 1.0) and math.isclose(alpha, 1.0))):
            return FusedPlan("rotate", [b["a"], b["b"]], {"alignment": int(round(alignment)), "alpha": alpha, "centralization": cz}, fallback, recipe)
    if ident == "add_difference_ties":
        k = _scalar(recipe.kwargs.get("k"), 1.0) if "k" in recipe.kwargs else 1.0          # wrappers.py:70-75
        alpha = _scalar(recipe.kwargs.get("alpha"), 1.0) if "alpha" in recipe.kwargs else 1.0
        vote = recipe.kwargs.get("vote_sgn")
        vote = bool(vote.value) if isinstance(vote, LiteralRecipeNode) else False
        if (len(recipe.args) >= 2 and all(isinstance(a, ModelRecipeNode) for a in recipe.args) and k is not None and alpha is not None
                and not math.isclose(alpha, 0.0) and len(recipe.args) <= _lib.MAX_MODELS):
            return FusedPlan("ties", list(recipe.args), {"k": k, "alpha": alpha, "vote_sgn": vote}, fallback)
    return None


def _literal(node, default):
    """The python value of a literal hyper-parameter node (None node = the default); _MISSING for anything else."""
    if node is None:
        return default
    if isinstance(node, LiteralRecipeNode) and not isinstance(node.value, Mapping):
        return node.value
    return _MISSING


_MISSING = object()


def _dare_params(t: MergeRecipeNode, base: ModelRecipeNode):
    """([models], params) of a ties_sum_with_dropout node the batched DARE path covers (ties.py:10-50): every delta is
    subtract(model, base), scalar probability not close to 1, della_eps == 0, apply_median False.  apply_stock False (the
    default) APPLIES the Model-Stock factor (ties.py:90), which takes 2..8 deltas."""
    kw = t.kwargs
    vals = {name: _literal(kw.get(name), default) for name, default in
            (("probability", 0.9), ("della_eps", 0.0), ("rescale", True), ("k", 0.2), ("vote_sgn", False), ("apply_stock", False),
             ("apply_median", False), ("seed", None), ("cos_eps", 1e-6))}
    if any(v is _MISSING for v in vals.values()):
        return None
    p, seed = vals["probability"], vals["seed"]
    if isinstance(p, bool) or not isinstance(p, (int, float)) or math.isclose(float(p), 1.0) or not math.isclose(float(vals["della_eps"]), 0.0):
        return None
    if vals["apply_median"] or not (seed is None or isinstance(seed, int)) or isinstance(vals["cos_eps"], bool) or not isinstance(vals["cos_eps"], (int, float)):
        return None
    model_stock = not vals["apply_stock"]
    if model_stock and not 2 <= len(t.args) <= 8:
        return None
    if isinstance(vals["k"], bool) or not isinstance(vals["k"], (int, float)):
        return None
    models = []
    for d in t.args:
        if not (isinstance(d, MergeRecipeNode) and d.merge_method.identifier == "subtract" and len(d.args) == 2
                and isinstance(d.args[0], ModelRecipeNode) and _same_model(d.args[1], base)):
            return None
        models.append(d.args[0])
    if not models or len(models) >= _lib.MAX_MODELS:
        return None
    return models, {"k": float(vals["k"]), "vote_sgn": bool(vals["vote_sgn"]), "probability": float(p), "rescale": bool(vals["rescale"]),
                    "seed": seed, "model_stock": model_stock, "cos_eps": float(vals["cos_eps"])}


def _same_model(a: RecipeNode, b: RecipeNode) -> bool:
    return a is b or (isinstance(a, ModelRecipeNode) and isinstance(b, ModelRecipeNode) and a.path == b.path)


def collect_models(recipe: RecipeNode) -> List[ModelRecipeNode]:
    seen, out = set(), []

    def walk(n):
        if isinstance(n, ModelRecipeNode):
            if id(n) not in seen:
                seen.add(id(n))
                out.append(n)
        elif isinstance(n, MergeRecipeNode):
            if id(n) in seen:                                                   # recipes are DAGs: visit shared nodes once
                return
            seen.add(id(n))
            for a in n.args:
                walk(a)
            for a in n.kwargs.values():
                walk(a)
            if getattr(n, "lora_base", None) is not None:                       # convert(lora, base): the base is no argument
                walk(n.lora_base)
    walk(recipe)
    return out


def _lora_file_nodes(recipe: RecipeNode) -> set:
    """ids of the model nodes that only appear as the LoRA-file argument of convert(lora, base)."""
    as_lora, other, visited = set(), set(), set()

    def walk(n, is_lora_arg=False):
        if isinstance(n, ModelRecipeNode):
            (as_lora if is_lora_arg else other).add(id(n))
        elif isinstance(n, MergeRecipeNode):
            if id(n) in visited:
                return
            visited.add(id(n))
            conv = lora_mod.is_conversion(n)
            for i, a in enumerate(n.args):
                walk(a, conv and i == 0)
            for a in n.kwargs.values():
                walk(a)
    walk(recipe)
    return as_lora - other


class _LoraStore:
    """The factors of every LoRA file of a fused LoRA plan, resident on the compute device: one pread scatter list per file
    into a (pinned) staging buffer at 16-byte aligned offsets, one H2D copy, tensors as views.  Alpha scalars stay on the
    host (they turn into the fp32 scale alpha / rank, lycoris.py:263)."""

    def __init__(self, plan: FusedPlan, readers, device: torch.device, host_threads: int):
        self.expr = plan.params["expr"]
        self.prefixes = list(plan.params.get("prefixes") or [lora_mod.PREFIX] * len(plan.params["loras"]))
        self.files = []
        self._hosts = []                                         # pinned staging buffers, until their copies are done
        self.tables_by_file: List[dict] = []
        copy_stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        pin = dict(pin_memory=True) if device.type == "cuda" else {}
        np_dtypes = {torch.float16: np.float16, torch.float32: np.float32, torch.float64: np.float64}
        up_s, down_s, alpha_s = lora_mod.UP, lora_mod.DOWN, "." + lora_mod.ALPHA
        for node in plan.params["loras"]:
            reader = readers[id(node)]
            # one pass over the header: staging offsets (16-byte aligned) and the scatter list.  From the scanner's tables where the
            # reader has them (no Python object per header entry: 16 files x 2 400 entries), else from its metadata objects.
            data_start = reader.data_start
            tab = reader._meta if isinstance(reader._meta, _HeaderTable) else None
            if tab is not None:
                names = tab.names
                pick = [i for i, k in enumerate(names) if k.endswith(up_s) or k.endswith(down_s) or k.endswith(alpha_s)]
                rows = np.asarray(pick, dtype=np.int64)
                nb = tab._offsets[rows, 1] - tab._offsets[rows, 0]
                aligned = (nb + 15) // 16 * 16
                total = int(aligned.sum())
                picked = [names[i] for i in pick]
                offs = dict(zip(picked, (np.cumsum(aligned) - aligned).tolist()))
                size, foff = nb.astype(np.uint64), (data_start + tab._offsets[rows, 0]).astype(np.int64)
                at = dict(zip(picked, range(len(picked))))
                shapes_l, nd_l, codes_l, nb_l = tab._shapes[rows].tolist(), tab._ndim[rows].tolist(), tab._codes[rows].tolist(), nb.tolist()
                shape_of = lambda k: tuple(shapes_l[at[k]][:nd_l[at[k]]])                 # noqa: E731
                dtype_of = lambda k: _DTYPE_BY_CODE[codes_l[at[k]]]                        # noqa: E731
                nbytes_of = lambda k: nb_l[at[k]]                                          # noqa: E731
            else:
                offs, sizes, foffs, metas, total = {}, [], [], {}, 0
                for k, m in reader.metadata():
                    if k.endswith(up_s) or k.endswith(down_s) or k.endswith(alpha_s):
                        nbk = m.end - m.start
                        offs[k] = total
                        metas[k] = m
                        sizes.append(nbk)
                        foffs.append(data_start + m.start)
                        total += (nbk + 15) // 16 * 16
                size, foff = np.asarray(sizes, dtype=np.uint64), np.asarray(foffs, dtype=np.int64)
                shape_of = lambda k: metas[k].shape                                        # noqa: E731
                dtype_of = lambda k: metas[k].dtype                                        # noqa: E731
                nbytes_of = lambda k: metas[k].end - metas[k].start                        # noqa: E731
            host = torch.empty(max(total, 16), dtype=torch.uint8, **pin)
            if offs:
                fds = np.full(len(offs), reader.fileno(), dtype=np.int32)
                dst = np.uint64(host.data_ptr()) + np.fromiter(offs.values(), dtype=np.uint64, count=len(offs))
                _lib.check(_lib.lib().mecha_host_pread_many(fds.ctypes.data, dst.ctypes.data, size.ctypes.data, foff.ctypes.data,
                                                            len(offs), host_threads), "mecha_host_pread_many")
            if copy_stream is not None:                          # the next file's header pass and preads run beside this copy
                dev_buf = torch.empty(host.numel(), dtype=torch.uint8, device=device)
                with torch.cuda.stream(copy_stream):
                    dev_buf.copy_(host, non_blocking=True)
            else:
                dev_buf = host
            self.files.append((reader, dev_buf, offs))
            self._hosts.append(host)
            # per base-key prefix: (device address of up, of down, rank, scale) -- what a network descriptor holds, computed once
            # per file instead of once per (key, network, wave): alpha / rank as fp32 arithmetic (lycoris.py:263)
            base_ptr = dev_buf.data_ptr()
            host_np = host.numpy()
            table = {}
            cut = len(up_s) + 1
            for k in offs:
                if not k.endswith(up_s):
                    continue
                prefix = k[:-cut]
                down_k = f"{prefix}.{down_s}"
                if down_k not in offs or f"{prefix}.{lora_mod.MID}" in reader:
                    continue                                     # incomplete pair / Tucker network: the slow path reports it
                up_shape, down_shape = shape_of(k), shape_of(down_k)
                rank = down_shape[0]
                scale = 1.0
                alpha_k = prefix + alpha_s
                if alpha_k in offs:                              # the scalar sits in the staging buffer already: no tensor per network
                    o = offs[alpha_k]
                    raw = host_np[o:o + nbytes_of(alpha_k)]
                    a_dt = dtype_of(alpha_k)
                    if a_dt == torch.bfloat16:
                        alpha = (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)[0]
                    else:
                        alpha = np.float32(raw.view(np_dtypes[a_dt])[0])
                    scale = float(np.float32(alpha) / np.float32(rank))
                table[prefix] = (base_ptr + offs[k], base_ptr + offs[down_k], rank, scale, up_shape, down_shape, dtype_of(k), dtype_of(down_k))
            self.tables_by_file.append(table)
        if copy_stream is not None:
            copy_stream.synchronize()
        self._hosts.clear()

    def _tensor(self, f: int, name: str) -> torch.Tensor:
        reader, buf, offs = self.files[f]
        m = reader.meta(name)
        return buf[offs[name]:offs[name] + m.nbytes()].view(m.dtype).view(tuple(m.shape))

    KEY_DTYPE = np.dtype([("offset", "<i8"), ("rows", "<i8"), ("cols", "<i8"), ("tile_start", "<i8"), ("net_begin", "<i4"), ("net_count", "<i4")])
    NET_DTYPE = np.dtype([("up", "<u8"), ("down", "<u8"), ("rank", "<i4"), ("scale", "<f4"), ("weight", "<f4"), ("reserved", "<i4")])

    def tables(self, layout: ArenaLayout, header):
        """The wave's descriptor tables as numpy records laid out like mecha_lora_key / mecha_lora_net (include/mecha_hip.h):
        (indices of the merged keys inside the wave, key records, network records, {skipped keys}, factor dtype).  Same
        per-key rule as entries(); no tensor objects, no ctypes structs -- the tables of an SDXL wave (1 071 keys x 16
        networks) took 0.5 s to build that way."""
        idx, keys, nets, skip = [], [], [], set()
        tile = 0
        lora_dtype = None
        T = _lib.LORA_TILE
        n_files = len(self.files)
        for index, (key, off, n) in enumerate(zip(layout.keys, layout.offsets, layout.numels)):
            shape = tuple(header[key].shape)
            stem = lora_mod.lora_key_prefix(key, shape, True, "")       # "_<stem>": each file's flavour prefix goes in front
            found = [self.tables_by_file[i].get(self.prefixes[i] + stem) if stem is not None else None for i in range(n_files)]
            if stem is not None:
                for i in range(n_files):                            # anything the fast table left out goes the careful way
                    if found[i] is None and f"{self.prefixes[i]}{stem}.{lora_mod.UP}" in self.files[i][0]:
                        lora_mod.network_entries(self.files[i][0], self.prefixes[i] + stem)       # raises for Tucker networks
            try:
                applied = lora_mod.networks_for_key(self.expr, lambda i: found[i] is not None)
            except StateDictKeyError:
                skip.add(key)
                continue
            if not n:
                continue
            rows = shape[0] if len(shape) >= 2 else 1
            cols = n // rows if rows else 0
            begin = len(nets)
            for i, w in applied:
                if math.isclose(float(w), 0.0):
                    continue
                up_ptr, down_ptr, rank, scale, ushape, dshape, udt, ddt = found[i]
                if ushape[0] != rows or math.prod(ushape[1:]) != rank or math.prod(dshape[1:]) != cols:
                    raise ValueError(f"{key}: LoRA {i} holds up {ushape} / down {dshape}, the key is {rows} x {cols}")
                if udt not in (torch.float16, torch.bfloat16) or ddt != udt or (lora_dtype or udt) != udt:
                    raise NotImplementedError("LoRA factors of one batch must share one of float16 / bfloat16")
                lora_dtype = udt
                nets.append((up_ptr, down_ptr, rank, scale, float(w), 0))
            if len(nets) - begin > _lib.MAX_MODELS:
                raise ValueError(f"at most {_lib.MAX_MODELS} networks per key")
            idx.append(index)
            keys.append((off, rows, cols, tile, begin, len(nets) - begin))
            tile += ((cols + T - 1) // T) * ((rows + T - 1) // T)
        return (np.asarray(idx, dtype=np.int64), np.array(keys, dtype=self.KEY_DTYPE), np.array(nets, dtype=self.NET_DTYPE), skip, lora_dtype)

    def entries(self, layout: ArenaLayout, header) -> Tuple[list, set]:
        """([(index in the wave, offset, rows, cols, [(up, down, alpha, weight), ...])] for the keys of a wave that are merged,
        {skipped keys})."""
        out, skip = [], set()
        for index, (key, off, n) in enumerate(zip(layout.keys, layout.offsets, layout.numels)):
            shape = tuple(header[key].shape)
            stem = lora_mod.lora_key_prefix(key, shape, True, "")
            found = {}

            def has(i, stem=stem, found=found):
                if stem is None:
                    return False
                if i not in found:
                    try:
                        found[i] = lora_mod.network_entries(self.files[i][0], self.prefixes[i] + stem)
                    except StateDictKeyError:
                        found[i] = None
                return found[i] is not None
            try:
                applied = lora_mod.networks_for_key(self.expr, has)
            except StateDictKeyError:
                skip.add(key)
                continue
            rows = shape[0] if len(shape) >= 2 else 1
            cols = n // rows if rows else 0
            nets = []
            for i, alpha_w in applied:
                up_k, down_k, alpha_k = found[i]
                alpha = self.files[i][0][alpha_k] if alpha_k is not None else None
                nets.append((self._tensor(i, up_k), self._tensor(i, down_k), alpha, alpha_w))
            if n:
                out.append((index, off, rows, cols, nets))
        return out, skip


class GraphInfo:
    """What the reference's graph finalization hands the executor, in the minimal structural form this backend uses
    (graph_finalization.py:113-141 ``finalize_with_keys``: per node, the keys it can produce): a model produces the keys
    its file holds, ``fallback`` the union of its two inputs (logistics.py:15-19), ``convert(lora, base)`` the keys of
    ``base`` the LoRA file holds a network for, any other method the keys every tensor input can produce with one shape.
    Also: how many consumers each node has (merge_context.py:45-61 -- a node with several is computed once per key) and
    one instance per node of a class-based method (merge_context.py:94, :134)."""

    def __init__(self, recipe: RecipeNode, readers):
        self.readers = readers
        self._keys: Dict[int, Optional[Dict[str, tuple]]] = {}
        self.consumers: Dict[int, int] = {}
        self.depth: Dict[int, int] = {}
        self.instances: Dict[int, Any] = {}
        self._count(recipe)
        self.consumers[id(recipe)] = self.consumers.get(id(recipe), 0) + 1

    def _count(self, node: RecipeNode) -> int:
        """consumer counts + ModelDepthRecipeVisitor (recipe_nodes.py:334-349)."""
        if id(node) in self.depth:
            return self.depth[id(node)]
        d = 0
        if isinstance(node, ModelRecipeNode):
            d = 1
        elif isinstance(node, MergeRecipeNode):
            self.instances[id(node)] = node.merge_method.instantiate()
            for child in (*node.args, *node.kwargs.values()):
                self.consumers[id(child)] = self.consumers.get(id(child), 0) + 1
                d = max(d, self._count(child))
            d += 1
        self.depth[id(node)] = d
        return d

    def keys(self, node: RecipeNode) -> Optional[Dict[str, tuple]]:
        """{key: shape} in production order, or None for a node that has a value for every key (scalar literals)."""
        nid = id(node)
        if nid in self._keys:
            return self._keys[nid]
        out: Optional[Dict[str, tuple]]
        if isinstance(node, ModelRecipeNode):
            out = {k: m.shape for k, m in self.readers[nid].metadata()}
        elif isinstance(node, LiteralRecipeNode):
            out = None
            if isinstance(node.value, Mapping):
                out = {k: (tuple(v.shape) if isinstance(v, torch.Tensor) else ()) for k, v in node.value.items()}
        elif isinstance(node, MergeRecipeNode):
            ident = node.merge_method.identifier
            inputs = [] if lora_mod.is_conversion(node) else [self.keys(a) for a in (*node.args, *node.kwargs.values())]
            if ident == "fallback":
                out = {}
                for ks in inputs:
                    if ks is None:
                        out = None
                        break
                    for k, shape in ks.items():
                        out.setdefault(k, shape)
            elif lora_mod.is_conversion(node):
                lora_reader, base_reader = self.readers[id(node.args[0])], self.readers[id(lora_mod.base_of(node))]
                out = {}
                for k, m in base_reader.metadata():
                    prefix = lora_mod.lora_key_prefix(k, m.shape, m.dtype.is_floating_point, node.merge_method.lora_prefix)
                    if prefix is not None and f"{prefix}.{lora_mod.UP}" in lora_reader and f"{prefix}.{lora_mod.DOWN}" in lora_reader:
                        out[k] = m.shape
            else:
                known = [ks for ks in inputs if ks is not None]
                out = None
                if known:
                    out = {k: shape for k, shape in known[0].items() if all(ks.get(k) == shape for ks in known[1:])}
        else:
            raise TypeError(type(node))
        self._keys[nid] = out
        return out


# ================================================================== compute backends
class HipBackend:
    """libmecha_hip.so on the current ROCm device.  The only product backend: no CPU path."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"merge_device={device!r}: the MI355X backend runs on a ROCm device only (there is no CPU path)")
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device is visible; the MI355X backend has no CPU fallback")
        if self.device.index is not None:
            torch.cuda.set_device(self.device)               # one process per GPU: this rank's device becomes the current one
        else:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.lib()
        self.methods = mm

    def stream(self):
        return torch.cuda.Stream(device=self.device)

    def run_fused(self, plan: FusedPlan, arenas: Sequence[torch.Tensor], out: torch.Tensor, layout: ArenaLayout,
                  segs_dev: torch.Tensor, segs_host, flags: torch.Tensor, workspace: Optional[torch.Tensor], shapes=None) -> None:
        import ctypes
        if plan.kind == "rotate":
            return self._run_rotate(plan, arenas, out, layout, flags, shapes)
        L = self.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        in_dt, out_dt, n = _DT_CODE[arenas[0].dtype], _DT_CODE[out.dtype], len(layout.keys)
        if plan.kind == "wsum":
            st = L.mecha_batched_weighted_sum(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(), segs_dev.data_ptr(), n,
                                              layout.total_elems, in_dt, out_dt, plan.params["alpha"], flags.data_ptr(), stream)
        elif plan.kind == "navg":
            st = L.mecha_batched_n_average(_lib.ptr_array([a.data_ptr() for a in arenas]), len(arenas), out.data_ptr(), segs_dev.data_ptr(),
                                           n, layout.total_elems, in_dt, out_dt, flags.data_ptr(), stream)
        elif plan.kind == "ties":
            st = L.mecha_batched_add_difference_ties(
                arenas[0].data_ptr(), _lib.ptr_array([a.data_ptr() for a in arenas[1:]]), len(arenas) - 1, out.data_ptr(),
                segs_dev.data_ptr(), ctypes.addressof(segs_host), n, layout.total_elems, in_dt, out_dt, plan.params["k"],
                plan.params["alpha"], int(plan.params["vote_sgn"]), None, workspace.data_ptr(), workspace.numel(), flags.data_ptr(), stream)
        elif plan.kind == "adddiff":
            st = L.mecha_batched_add_difference(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(), segs_dev.data_ptr(), n,
                                                layout.total_elems, in_dt, out_dt, plan.params["alpha"], flags.data_ptr(), stream)
        elif plan.kind == "addsub":
            st = L.mecha_batched_add_difference_sub(arenas[0].data_ptr(), arenas[1].data_ptr(), arenas[2].data_ptr(), out.data_ptr(),
                                                    segs_dev.data_ptr(), n, layout.total_elems, in_dt, out_dt, plan.params["alpha"],
                                                    flags.data_ptr(), stream)
        elif plan.kind == "slerp":
            st = L.mecha_batched_slerp(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(), segs_dev.data_ptr(), ctypes.addressof(segs_host),
                                       n, layout.total_elems, in_dt, out_dt, plan.params["alpha"], workspace.data_ptr(), workspace.numel(),
                                       flags.data_ptr(), stream)
        elif plan.kind == "addperp":
            st = L.mecha_batched_add_perpendicular(arenas[0].data_ptr(), arenas[1].data_ptr(), arenas[2].data_ptr(), out.data_ptr(),
                                                   segs_dev.data_ptr(), ctypes.addressof(segs_host), n, layout.total_elems, in_dt, out_dt,
                                                   plan.params["alpha"], workspace.data_ptr(), workspace.numel(), flags.data_ptr(), stream)
        elif plan.kind == "dare":
            bits, n_bits = self._dare_stream(plan, max(layout.numels))
            rows_dev = rows_host = None
            if plan.params["model_stock"]:                          # the factor is per slice of the key's last dimension (ties.py:154)
                rows_host = np.asarray([shape[-1] for shape in shapes], dtype=np.int64)
                rows_dev = torch.from_numpy(rows_host).to(self.device)
            st = L.mecha_batched_add_difference_dare(
                arenas[0].data_ptr(), _lib.ptr_array([a.data_ptr() for a in arenas[1:]]), len(arenas) - 1, out.data_ptr(),
                segs_dev.data_ptr(), ctypes.addressof(segs_host), n, layout.total_elems, in_dt, out_dt, plan.params["k"],
                plan.params["alpha"], int(plan.params["vote_sgn"]), plan.params["probability"], int(plan.params["rescale"]),
                bits.data_ptr(), n_bits, int(plan.params["model_stock"]), plan.params["cos_eps"],
                rows_dev.data_ptr() if rows_dev is not None else None, rows_host.ctypes.data if rows_host is not None else None,
                None, workspace.data_ptr(), workspace.numel(), flags.data_ptr(), stream)
        else:
            raise ValueError(plan.kind)
        _lib.check(st, f"batched {plan.kind}")

    def prepare(self, plan: FusedPlan, max_numel: int) -> None:
        """Before the first wave of a fused run.  DARE: the Bernoulli stream every key's masks are cut from (the reference seeds a
        fresh generator per key, ties.py:29-31) is drawn once, for the largest key of the run, on a thread of its own beside
        the staging of the first wave."""
        if plan.kind != "dare":
            return
        n_bits = (len(plan.models) - 1) * int(max_numel)
        have = getattr(self, "_dare", None)
        ident = (mm._dare_seed(plan.params["seed"]), float(plan.params["probability"]))
        if have is not None and have[0] == ident and have[1] >= n_bits:
            return

        def draw():
            host = torch.empty((n_bits + 31) // 32 + 1, dtype=torch.int32, pin_memory=self.device.type == "cuda")
            _lib.check(self.lib.mecha_dare_draw_stream_host(ident[0] & 0xFFFFFFFFFFFFFFFF, ident[1], n_bits, host.data_ptr()),
                       "mecha_dare_draw_stream_host")
            return host
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        self._dare = (ident, n_bits, pool.submit(draw), None)
        pool.shutdown(wait=False)

    def _dare_stream(self, plan: FusedPlan, max_numel: int):
        ident, n_bits, fut, dev = self._dare
        if (len(plan.models) - 1) * int(max_numel) > n_bits:
            raise RuntimeError("the DARE mask stream was prepared for smaller keys than this wave holds")
        if dev is None:
            dev = fut.result().to(self.device)                 # once per run: the waves share it
            self._dare = (ident, n_bits, fut, dev)
        return dev, n_bits

    # most workspace one batched rotate call may take; a shape group that needs more is cut into several calls
    ROTATE_WORKSPACE_BYTES = 24 << 30

    def _run_rotate(self, plan: FusedPlan, arenas, out, layout: ArenaLayout, flags: torch.Tensor, shapes) -> None:
        """rotate over a wave (svd.py:10-86 per key): keys of <= 1 dimension are lerp(a, b, alpha) (svd.py:46-47) in one batched
        launch over their own segment table; the others are grouped by their 2-D shape (svd.py:49-53) and every group runs
        as one batched launch sequence (mecha_batched_rotate)."""
        import ctypes
        L = self.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        in_dt, out_dt = _DT_CODE[arenas[0].dtype], _DT_CODE[out.dtype]
        alpha, alignment, cz = plan.params["alpha"], plan.params["alignment"], plan.params["centralization"]
        flat, groups = [], {}
        for i, shape in enumerate(shapes):
            if len(shape) <= 1:
                flat.append(i)
            else:
                groups.setdefault(mm.rotate_shape_2d(shape), []).append(i)
        # one index tensor for the whole wave: [flat keys..., group 0 keys..., group 1 keys...]; per-call flag words are
        # gathered back into the wave's order at the end
        order = flat + [i for idx in groups.values() for i in idx]

        def upload(values):                                     # pinned + non-blocking: the host must not wait for the stream here
            return torch.tensor(values, dtype=torch.int64, pin_memory=True).to(self.device, non_blocking=True)
        offs_all = upload([layout.offsets[i] for i in order] or [0])
        fl_all = torch.zeros(max(len(order), 1), dtype=torch.int32, device=self.device)
        pos = 0
        if flat:
            # ORDER MATTERS: the batched lerp computes EVERY element of the arena (its segment table only attributes the flag
            # words), so it also writes lerp values into the output regions of the 2-D keys; the rotate launches below run later
            # on the same stream and overwrite those regions completely (GEMM epilogues, rot_same_kernel).  Keep this launch
            # first.  (test_gpu_rotate.py::test_batched_rotate_through_merge mixes 1-D and 2-D keys in one wave and would see
            # lerp values in a 2-D key if the order changed.)
            segs = upload([[layout.offsets[i], layout.numels[i]] for i in flat])
            _lib.check(L.mecha_batched_weighted_sum(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(), segs.data_ptr(), len(flat),
                                                    layout.total_elems, in_dt, out_dt, alpha, fl_all.data_ptr(), stream), "batched lerp (rotate, ndim <= 1)")
            pos = len(flat)
        # the calls of the wave, planned first: ONE workspace of the largest size serves them all (growing it between calls
        # frees the old one, and a free waits for the device: the host then sits behind every queued kernel)
        calls = []
        for (n, p), idx in groups.items():
            per_key = int(L.mecha_batched_rotate_workspace_bytes(n, p, 1))
            chunk = max(1, min(len(idx), self.ROTATE_WORKSPACE_BYTES // max(per_key, 1)))
            chunk = -(-len(idx) // -(-len(idx) // chunk))          # equal shares when the group needs several calls
            calls.append((n, p, idx, chunk, max(int(L.mecha_batched_rotate_workspace_bytes(n, p, min(chunk, len(idx) - c0)))
                                                   for c0 in range(0, len(idx), chunk))))
        ws = self._rotate_ws(max((c[4] for c in calls), default=16))
        for n, p, idx, chunk, _ in calls:
            for c0 in range(0, len(idx), chunk):
                cnt = min(chunk, len(idx) - c0)
                _lib.check(L.mecha_batched_rotate(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(),
                                                  offs_all[pos + c0:].data_ptr(), cnt, n, p, in_dt, out_dt, float(cz), int(alignment), float(alpha),
                                                  ws.data_ptr(), ws.numel(), fl_all[pos + c0:].data_ptr(), stream), "mecha_batched_rotate")
            pos += len(idx)
        if order:
            flags.index_copy_(0, upload(order), fl_all[:len(order)])

    def _rotate_ws(self, nbytes: int) -> torch.Tensor:
        """One growing workspace per backend: the calls of a wave run one after the other on the compute stream."""
        ws = getattr(self, "_rot_ws", None)
        if ws is None or ws.numel() < nbytes:
            self._rot_ws = None
            ws = self._rot_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def run_lora(self, entries, base: torch.Tensor, out: torch.Tensor, flags: torch.Tensor) -> None:
        """Every key of a wave in one launch; `entries` as _LoraStore.entries returns them, `flags` one word per wave key."""
        if not entries:
            return
        batch = mm.LoraBatch([(off, rows, cols, nets) for _, off, rows, cols, nets in entries], self.device)
        entry_flags = torch.zeros(len(entries), dtype=torch.int32, device=self.device)
        batch.run(base, out, entry_flags)
        flags.index_copy_(0, torch.tensor([i for i, *_ in entries], dtype=torch.int64, device=self.device), entry_flags)

    def run_lora_tables(self, tables, base: torch.Tensor, out: torch.Tensor, flags: torch.Tensor) -> None:
        """run_lora from the numpy descriptor tables of _LoraStore.tables: two uploads and one launch per wave."""
        import ctypes
        idx, keys, nets, _, lora_dtype = tables
        if not len(idx):
            return
        if not len(nets):
            nets = np.zeros(1, dtype=nets.dtype)
        keys_dev = torch.from_numpy(keys.view(np.uint8).reshape(-1)).to(self.device)
        nets_dev = torch.from_numpy(nets.view(np.uint8).reshape(-1)).to(self.device)
        entry_flags = torch.zeros(len(idx), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.mecha_batched_lora_accumulate(
            base.data_ptr(), out.data_ptr(), keys_dev.data_ptr(), keys.ctypes.data, len(idx), nets_dev.data_ptr(), nets.ctypes.data,
            _DT_CODE[base.dtype], _DT_CODE[out.dtype], _DT_CODE[lora_dtype or torch.float16], entry_flags.data_ptr(),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mecha_batched_lora_accumulate")
        flags.index_copy_(0, torch.from_numpy(idx).to(self.device), entry_flags)

    def workspace_bytes(self, plan: FusedPlan, layout: ArenaLayout, shapes=None) -> int:
        if plan.kind == "ties":
            return int(self.lib.mecha_batched_ties_workspace_bytes(len(plan.models) - 1, len(layout.keys), layout.total_elems))
        if plan.kind == "dare":
            rows = sum(n // shape[-1] for n, shape in zip(layout.numels, shapes)) if plan.params["model_stock"] else 0
            return int(self.lib.mecha_batched_dare_workspace_bytes(len(plan.models) - 1, len(layout.keys), layout.total_elems, rows))
        if plan.kind in ("slerp", "addperp"):
            return int(self.lib.mecha_batched_reduce_workspace_bytes(len(layout.keys), layout.total_elems))
        return 0


# ================================================================== the executor
@dataclass
class _Wave:
    keys: List[str]
    layout: ArenaLayout


def _plan_waves(items: Sequence[Tuple[str, int]], wave_elems: int, ramp: bool = False) -> Tuple[List[_Wave], List[_Wave]]:
    """(waves, split keys).  Keys (file order) are packed into waves of at most `wave_elems` elements per model.  A key
    LARGER than that window (an embedding of a FLUX / LLM checkpoint under a small staging budget) becomes a wave of its
    own that is moved through the window in pieces of whole rows of elements (SURVEY.md §8e "row-split"; the reference
    instead grows its buffer to the tensor, streaming.py:154-158)."""
    waves, cur, cur_elems, split = [], [], 0, []
    # `ramp`: the first and the last waves are smaller (a quarter, then half of the window): the pipeline's fill -- nothing runs
    # on the device before the first wave is staged and copied -- and its drain -- the last wave's D2H copy and file write
    # overlap nothing -- shrink with them.  Pays for jobs bound by their kernels (SDXL x2 rotate end to end 1.96 -> 1.88 s);
    # jobs bound by the host side lose to the extra waves (SDXL x4 TIES 0.76 -> 0.83 s, LoRA 0.85 -> 0.87, FLUX stripe equal),
    # so the caller asks for it per plan.  MECHA_WAVE_RAMP=0 / 1 overrides.
    ramp = {"0": False, "1": True}.get(os.environ.get("MECHA_WAVE_RAMP", ""), ramp)
    total = sum((n + 127) // 128 * 128 for _, n in items if n <= wave_elems)
    done = 0

    def window() -> int:
        if not ramp:
            return wave_elems
        left = total - done
        head = wave_elems // 4 if done < wave_elems // 4 else wave_elems // 2 if done < 3 * wave_elems // 4 else wave_elems
        tail = wave_elems // 4 if left <= wave_elems // 4 else wave_elems // 2 if left <= 3 * wave_elems // 4 else wave_elems
        return max(min(head, tail), 1)

    cap = window()
    for key, n in items:
        if n > wave_elems:
            split.append(_Wave([key], plan_arena([(key, n)])))
            continue
        if cur and cur_elems + n > cap:                      # a key above a reduced window still gets a wave (of its own)
            waves.append(cur)
            done += cur_elems
            cur, cur_elems = [], 0
            cap = window()
        cur.append((key, n))
        cur_elems += (n + 127) // 128 * 128
    if cur:
        waves.append(cur)
    return [_Wave([k for k, _ in w], plan_arena(w)) for w in waves], split


# host seconds per phase of the last fused merge() of this process (tools/bench_e2e.py prints it)
last_stats: Dict[str, float] = {}


class _phase:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *a):
        last_stats[self.name] = last_stats.get(self.name, 0.0) + time.perf_counter() - self.t0


def default_host_threads() -> int:
    """Threads of this rank's staging / writer team: MECHA_HOST_THREADS, else the cores this process may run on (after the NUMA
    binding) shared with the other local ranks, at most 16 -- 128 copy threads on a node of 8 ranks helped nobody."""
    env = os.environ.get("MECHA_HOST_THREADS")
    if env:
        return max(1, int(env))
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    _, world, _ = _dist_info()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world)) if world > 1 else 1
    from . import numa
    share = cores if numa._bound is not None else cores // max(local_world, 1)      # a bound rank already owns its share
    return max(2, min(16, share)) if local_world > 1 else max(1, min(16, cores))


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist
    return 0, 1, None


def merge(
    recipe,
    *,
    fallback_model=None,
    merge_device: Optional[str] = "cuda",
    merge_dtype: torch.dtype = torch.float32,
    output_device: Optional[str] = "cpu",
    output_dtype: torch.dtype = torch.float16,
    total_buffer_size: int = 2 ** 31,
    check_finite_output: bool = True,
    omit_non_finite_inputs: bool = True,
    strict_mandatory_keys: bool = False,
    output=None,
    threads: Optional[int] = None,
    cache: Optional[Dict[RecipeNode, Any]] = None,
    backend=None,
) -> Optional[MutableMapping[str, torch.Tensor]]:
    """Materialise a state dict from a recipe graph (signature subset of reference merging.py:34-54).

    ``output``: a ``.safetensors`` path, a dict-like, or None (a new dict is returned).  ``total_buffer_size`` is the
    byte budget of the pinned staging buffers (the reference's 2**28 read window scaled to a 288 GB device).
    Under ``torch.distributed`` every rank merges its own stripe of keys; with a file output all ranks fill the same
    file, with a dict output each rank returns its stripe.  ``cache``: {merge node: its method's ``create_cache()`` object}
    (merging.py:199), handed to the node's body on every key; today ``rotate`` keeps its transforms there (svd.py:55-79).
    ``threads`` (merging.py:59-62, :171-179): worker threads of the per-key path, each on its own HIP stream like the
    reference's pool threads (merging.py:330-341); None = min(max(input files, 2), cpu count, 4), 0 = the caller's thread.
    The fused batched path merges whole waves of keys per launch and does not use it."""
    if merge_dtype != torch.float32:
        raise NotImplementedError("the MI355X kernels compute in float32 (the reference default merge_dtype)")
    if not omit_non_finite_inputs:
        raise NotImplementedError("omit_non_finite_inputs=False is not supported: the finite check is fused into the kernels")
    recipe = value_to_node(recipe)
    original_recipe = recipe                                                 # what the output's metadata records (merging.py:152, :200)
    if fallback_model is not None:
        recipe = recipe | value_to_node(fallback_model)                      # merging.py:154-155
    backend = backend or HipBackend(merge_device)
    rank, world, dist = _dist_info()
    if world > 1:                                                            # one process per GPU: keep this rank's host side on one NUMA node
        from . import numa
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        bus = None
        if backend.device.type == "cuda":
            pr = torch.cuda.get_device_properties(backend.device)
            if hasattr(pr, "pci_bus_id") and hasattr(pr, "pci_domain_id"):
                bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        numa.bind_rank(int(os.environ.get("LOCAL_RANK", rank)) % max(local_world, 1), local_world, bus)

    lora_mod.bind_bases(recipe)
    models = collect_models(recipe)
    if not models:
        raise ValueError("the recipe has no model inputs")
    last_stats.clear()
    with _phase("open_inputs"):
        readers = {id(m): open_reader(m.path) for m in models}
    t_plan = time.perf_counter()
    try:
        plan = fuse(recipe)
        if plan is not None and plan.kind == "rotate" and cache and any(n is plan.node for n in cache):
            plan = None                                                      # svd.py:55-79: the factor cache lives on the per-key path
        # LoRA files name their tensors after the base keys (lora_<stem>.lora_up.weight ...): they take no part in config inference
        aux = _lora_file_nodes(recipe)
        merge_inputs = [m for m in models if id(m) not in aux]
        if not merge_inputs:
            raise ValueError("the recipe has no model inputs besides LoRA files")
        # model config: explicit (model(path, config=...)) or inferred from the key names (graph_finalization.py:1024-1058,
        # :591-624, :698); no candidate = the structural config (:695-696)
        config = model_configs.infer([readers[id(m)].keys() for m in merge_inputs], [getattr(m, "config", None) for m in merge_inputs])
        if config is not None:
            for m in merge_inputs:                                           # inputs may name a key by an alias (merging.py:394-403)
                readers[id(m)].apply_aliases(config.resolve_alias)
        # the keys the recipe can produce (graph_finalization.py:113-141, structural form: GraphInfo), in the config's order
        # (merging.py:202) or the order the inputs hold them.  A key some input lacks is still produced when a `|` further
        # up has a default for it (logistics.py:15-19).
        try:                                                                # merging.py:259-263
            recipe_text = serialize(original_recipe, {id(m): (config.identifier if config is not None else "structural") for m in merge_inputs})
        except TypeError:
            log.warning("The recipe graph could not be serialized. The output state dict will not contain the recipe.")
            recipe_text = None
        info = GraphInfo(recipe, readers)
        root_keys = info.keys(recipe)
        if root_keys is None:
            raise ValueError("the recipe has no model inputs")
        if config is not None:
            order = [k for k in config.keys() if k in root_keys]
        else:                                                               # structural config = the keys every input holds (:626-637, :695-696)
            common = set.intersection(*[set(readers[id(m)].keys()) for m in merge_inputs])
            order = [k for k in root_keys if k in common]
        items, header = [], {}
        for key in order:
            shape = root_keys[key]
            items.append((key, int(math.prod(shape))))
            header[key] = TensorMetadata(shape, output_dtype)
        log.debug("model config: %s, %d keys", config.identifier if config is not None else "structural", len(items))
        my_items = shard_keys_lpt(items, world)[rank] if world > 1 else items
        order = {k: i for i, (k, _) in enumerate(items)}
        my_items.sort(key=lambda kv: order[kv[0]])

        last_stats["plan"] = time.perf_counter() - t_plan       # fuse, config inference, key analysis, recipe text
        to_file = isinstance(output, (str, pathlib.Path))
        sink: Any
        t_sink = time.perf_counter()
        if to_file:
            out_path = pathlib.Path(output)
            esz = DTYPE_REVERSE_MAPPING[output_dtype][1]
            sizes = [sum(n for _, n in shard) * esz for shard in (shard_keys_lpt(items, world) if world > 1 else [items])]
            region_start = sum(sizes[:rank])
            if world > 1:
                # rank 0 creates the file at its full length; then every rank reserves the pages of ITS OWN region in the background
                # (round 2: rank 0 fallocated the whole file while the others waited on the broadcast)
                if rank == 0:
                    sink = OutSafetensorsDict(out_path, header, recipe_text, create=True, region_start=region_start, region_bytes=sizes[0])
                dist.barrier()                                  # the file exists and has its length
                if rank != 0:
                    sink = OutSafetensorsDict(out_path, header, recipe_text, create=False, region_start=region_start, region_bytes=sizes[rank])
            else:
                sink = OutSafetensorsDict(out_path, header, recipe_text)
        else:
            sink = output if output is not None else {}
            if recipe_text is not None:
                sink["__metadata__"] = {"mecha_recipe": recipe_text}          # merging.py:287-288

        # Keys the batched kernels cover go through the fused pipeline, one run per storage dtype the inputs hold them in
        # (a checkpoint may keep, say, its VAE in fp32 beside a bf16 UNet); everything else -- integer keys (position_ids: the
        # reference casts them to fp32 like any other key, merging.py:673-681), keys whose inputs disagree on the dtype, empty
        # tensors, keys one of the fused inputs lacks -- is evaluated per key into the same sink.
        last_stats["open_output"] = time.perf_counter() - t_sink
        skipped: List[str] = []
        with _phase("split_fusable"):
            groups, generic_items = _split_fusable(plan, readers, my_items, header)
        for in_dtype, group_items in groups.items():
            _run_fused(plan, backend, readers, group_items, in_dtype, header, sink, to_file, output_device, output_dtype,
                       total_buffer_size, check_finite_output, skipped)
        if generic_items:
            if threads is None:
                threads = min(max(len(models), 2), os.cpu_count() or 1, 4)       # merging.py:171-172
            with _phase("generic"):
                _run_generic(recipe, info, backend, readers, generic_items, header, sink, to_file, output_device, output_dtype,
                             check_finite_output, strict_mandatory_keys, skipped, cache or {}, threads)
        last_stats["fused_keys"] = sum(len(g) for g in groups.values())
        last_stats["generic_keys"] = len(generic_items)
        for k in skipped:
            log.debug("Skipping key: %s", k)

        if to_file:
            if world > 1:
                gathered = [None] * world if rank == 0 else None
                dist.gather_object((sink.entries, skipped), gathered, dst=0)
                if rank == 0:
                    extra = {}
                    any_skipped = False
                    for r, (entries, sk) in enumerate(gathered):
                        any_skipped |= bool(sk)
                        if r != 0:
                            extra.update(entries)
                    sink.close(extra_entries=extra)
                    if any_skipped:
                        compact_holes(out_path)
                else:
                    sink.close(write_header=False)
                dist.barrier()
            else:
                with _phase("close_output"):
                    sink.close()
                if skipped:
                    compact_holes(out_path)
            return None
        return sink
    finally:
        for r in readers.values():
            r.close()


def _split_fusable(plan: Optional[FusedPlan], readers, items, header):
    """({storage dtype: [(key, numel)]}, [(key, numel) for the per-key path]).  A key is fused when the recipe is one the
    batched kernels cover and every input of the plan holds it, non-empty, with the output's shape and ONE of the storage
    dtypes the kernels read (fp32 / fp16 / bf16)."""
    groups: Dict[torch.dtype, list] = {}
    generic = []
    for key, n in items:
        dt = None
        if plan is not None and n > 0:
            for m in plan.models:
                r = readers[id(m)]
                if key not in r:
                    dt = None
                    break
                meta = r.meta(key)
                if meta.shape != header[key].shape or meta.dtype not in _DT_CODE or (dt is not None and meta.dtype != dt):
                    dt = None
                    break
                dt = meta.dtype
            if dt is not None and plan.kind == "dare" and plan.params["model_stock"] and len(header[key].shape) == 0:
                dt = None                   # no last dimension to take the cosine along: the per-key body raises like the reference's
        if dt is None:
            generic.append((key, n))
        else:
            groups.setdefault(dt, []).append((key, n))
    return groups, generic


class StagingRehearsalBackend:
    """``bench.py --rehearse-staging``: the host side of a fused run without a GPU (device copies and kernels left out)."""
    rehearsal = True

    def __init__(self):
        self.device = torch.device("cpu")
        self.methods = mm

    def stream(self):
        return None

    def workspace_bytes(self, plan, layout, shapes=None):
        return 0


class NullSink(dict):
    """``merge(output=NullSink())``: every key is merged and copied back to pinned host memory, then dropped.  For timing
    the files -> staging -> H2D -> kernels -> D2H pipeline without the cost of an output file (tools/bench_e2e.py --sink none)."""
    discard = True


# ------------------------------------------------------------------ fused (batched) path
def _run_fused(plan, backend, readers, items, in_dtype, header, sink, to_file, output_device, output_dtype, total_buffer_size,
               check_finite_output, skipped):
    dev = backend.device
    is_cuda = dev.type == "cuda"
    n_models = len(plan.models)
    in_esz, out_esz = DTYPE_REVERSE_MAPPING[in_dtype][1], DTYPE_REVERSE_MAPPING[output_dtype][1]
    # pinned staging: n_sets input sets (2 = double buffering: wave w + 1 is read from the files while wave w crosses the link)
    # and a ring of n_outs output buffers of their own (the file writer may lag two waves behind the link without holding up
    # the next wave's upload: with the output buffer inside the set, H2D of wave w + 2 waited for the write of wave w); the
    # budget covers all of them
    n_sets = max(1, int(os.environ.get("MECHA_BUFFER_SETS", "2")))
    n_outs = max(n_sets, int(os.environ.get("MECHA_OUT_BUFFERS", str(n_sets + 2))))
    wave_elems = max(max(total_buffer_size, 1 << 16) // (n_sets * n_models * in_esz + n_outs * out_esz), 1024)
    waves, split_waves = _plan_waves(items, wave_elems, ramp=plan.kind == "rotate")
    if not waves and not split_waves:
        return
    # pieces of a split key: whole TIES tiles (4096 elements), as large as the window allows
    piece_elems = max(wave_elems // 4096 * 4096, 4096)
    cap_dev = max((w.layout.total_elems for w in waves), default=128)
    cap = max(cap_dev, piece_elems if split_waves else 0)              # pinned staging: never more than the window

    preloaded = is_cuda and not split_waves and all(hasattr(readers[id(m)], "host_address") for m in plan.models)
    # a backend that only rehearses the host side (bench.py --rehearse-staging): real wave plan, real staging reads, real writer;
    # the device copies and the kernels are left out, the staging buffers stand in for the device arenas
    rehearsal = bool(getattr(backend, "rehearsal", False))

    def alloc_set():
        pin = dict(pin_memory=True) if is_cuda else {}
        return {
            "h_in": [torch.empty(cap if not preloaded else 8, dtype=in_dtype, **pin) for _ in range(n_models)],
            "h_out": None, "out_slot": None,
            "h_flags_pinned": torch.empty(max((len(w.layout.keys) for w in waves), default=1), dtype=torch.int32, **pin),
            "d_in": [torch.empty(cap_dev if not rehearsal else 8, dtype=in_dtype, device=dev) for _ in range(n_models)],
            "d_out": torch.empty(cap_dev if not rehearsal else 8, dtype=output_dtype, device=dev),
            "flags": None, "h_flags": None, "ev_h2d": None, "ev_comp": None, "ev_d2h": None, "wave": None,
        }
    with _phase("alloc"):
        sets = [alloc_set() for _ in range(min(n_sets, max(len(waves), 2 if split_waves else 1)))]
        pin_out = dict(pin_memory=True) if is_cuda else {}
        outs = [{"h_out": torch.empty(cap, dtype=output_dtype, **pin_out)} for _ in range(min(n_outs, max(len(waves), 2 if split_waves else 1)))]
    out_cursor = [0]
    last_stats["pinned_bytes"] = max(last_stats.get("pinned_bytes", 0), cap * (len(sets) * n_models * in_esz + len(outs) * out_esz))
    last_stats["split_keys"] = last_stats.get("split_keys", 0) + len(split_waves)
    s_copy_in = backend.stream() if is_cuda else None
    s_comp = backend.stream() if is_cuda else None
    s_copy_out = backend.stream() if is_cuda else None

    if hasattr(backend, "prepare"):
        backend.prepare(plan, max(n for _, n in items))
    host_threads = default_host_threads()
    writer = concurrent.futures.ThreadPoolExecutor(max_workers=1)
    # the LoRA files' factors (pread + H2D + per-file tables) load on a thread of their own beside the staging of the first wave
    lora_store = None
    def _load_loras():
        # a new thread's current device is 0, not this rank's: pinned allocations and the host allocator's events would go to GPU 0
        with (torch.cuda.device(dev) if is_cuda else _null()):
            return _LoraStore(plan, readers, dev, host_threads)
    lora_future = writer.submit(_load_loras) if plan.kind == "lora" else None
    hostlib = _lib.lib()

    # every input already in pinned memory (streaming.preload_pinned): no staging copies, the H2D copies read the images
    from_pinned = preloaded and plan.kind != "lora"
    last_stats["source"] = "pinned images" if from_pinned else "files"

    def stage(bs, wave: _Wave):
        """files -> pinned staging: one scatter list per wave (every (model, key) pair to its arena offset, and the
        gaps after each key zero-filled so they stay finite), moved by the library's host thread team."""
        if from_pinned:
            return
        L = wave.layout
        nk = len(L.keys)
        offs = np.asarray(L.offsets, dtype=np.int64)
        nums = np.asarray(L.numels, dtype=np.int64)
        ends = offs + nums
        gaps = np.minimum((ends + 127) // 128 * 128, cap) - ends
        has_gap = gaps > 0
        fds, dsts, sizes, foffs = [], [], [], []
        for mi in range(n_models):
            reader = readers[id(plan.models[mi])]
            base = bs["h_in"][mi].data_ptr()
            fds += [np.full(nk, reader.fileno(), dtype=np.int32), np.full(int(has_gap.sum()), -1, dtype=np.int32)]
            dsts += [base + offs * in_esz, base + ends[has_gap] * in_esz]
            sizes += [nums * in_esz, gaps[has_gap] * in_esz]
            foffs += [np.fromiter((reader.file_offset(k) for k in L.keys), dtype=np.int64, count=nk), np.zeros(int(has_gap.sum()), dtype=np.int64)]
        fds_a = np.ascontiguousarray(np.concatenate(fds), dtype=np.int32)
        dst_a = np.ascontiguousarray(np.concatenate(dsts), dtype=np.uint64)
        size_a = np.ascontiguousarray(np.concatenate(sizes), dtype=np.uint64)
        foff_a = np.ascontiguousarray(np.concatenate(foffs), dtype=np.int64)
        _lib.check(hostlib.mecha_host_pread_many(fds_a.ctypes.data, dst_a.ctypes.data, size_a.ctypes.data, foff_a.ctypes.data,
                                                 len(fds_a), host_threads), "mecha_host_pread_many")

    def wait_write(slot):
        fut = slot.pop("write_fut", None)
        if fut is not None:
            with _phase("wait_write"):
                fut.result()

    def next_out():
        """The next output buffer of the ring, once the file writer is done with what it held."""
        slot = outs[out_cursor[0] % len(outs)]
        out_cursor[0] += 1
        wait_write(slot)
        return slot

    def launch(bs, wave: _Wave):
        L = wave.layout
        bs["wave"] = wave
        if rehearsal:
            bs["skip"] = ()
            bs["out_slot"] = next_out()
            bs["h_out"] = bs["out_slot"]["h_out"]
            bs["h_flags"] = torch.zeros(len(L.keys), dtype=torch.int32)
            return
        segs_host = L.host_segments()
        bs["segs_host"] = segs_host
        ctx_in = torch.cuda.stream(s_copy_in) if is_cuda else _null()
        with ctx_in:
            # the (small, pageable, hence synchronous) segment-table copy goes FIRST: behind the arena copies it would
            # block the host until ~GBs of H2D have drained
            segs_dev = L.device_segments(dev)
            if from_pinned:
                import ctypes
                nk = len(L.keys)
                offs_b = np.asarray(L.offsets, dtype=np.uint64) * np.uint64(in_esz)
                sizes = np.ascontiguousarray(np.tile(np.asarray(L.numels, dtype=np.uint64) * np.uint64(in_esz), n_models))
                dsts = np.ascontiguousarray(np.concatenate([np.uint64(bs["d_in"][mi].data_ptr()) + offs_b for mi in range(n_models)]))
                srcs = np.ascontiguousarray(np.concatenate([
                    np.fromiter((readers[id(plan.models[mi])].host_address(k) for k in L.keys), dtype=np.uint64, count=nk) for mi in range(n_models)]))
                _lib.check(hostlib.mecha_host_h2d_many(dsts.ctypes.data, srcs.ctypes.data, sizes.ctypes.data, len(sizes),
                                                       ctypes.c_void_p(s_copy_in.cuda_stream)), "mecha_host_h2d_many")
            else:
                for mi in range(n_models):
                    bs["d_in"][mi][:L.total_elems].copy_(bs["h_in"][mi][:L.total_elems], non_blocking=True)
            if is_cuda:
                bs["ev_h2d"] = torch.cuda.Event()
                bs["ev_h2d"].record(s_copy_in)
        ctx = torch.cuda.stream(s_comp) if is_cuda else _null()
        with ctx:
            if is_cuda:
                s_comp.wait_event(bs["ev_h2d"])
            flags = torch.zeros(len(L.keys), dtype=torch.int32, device=dev)
            nbytes = backend.workspace_bytes(plan, L, [tuple(header[k].shape) for k in L.keys])
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev) if nbytes else None
            if plan.kind == "lora" and hasattr(backend, "run_lora_tables"):
                tables = lora_store.tables(L, header)
                bs["skip"] = tables[3]
                backend.run_lora_tables(tables, bs["d_in"][0][:L.total_elems], bs["d_out"][:L.total_elems], flags)
            elif plan.kind == "lora":
                entries, bs["skip"] = lora_store.entries(L, header)
                backend.run_lora(entries, bs["d_in"][0][:L.total_elems], bs["d_out"][:L.total_elems], flags)
            else:
                bs["skip"] = ()
                backend.run_fused(plan, [a[:L.total_elems] for a in bs["d_in"]], bs["d_out"][:L.total_elems], L, segs_dev, segs_host, flags, ws,
                                  shapes=[tuple(header[k].shape) for k in L.keys])
            bs["flags"], bs["ws"], bs["segs_dev"] = flags, ws, segs_dev
            if is_cuda:
                bs["ev_comp"] = torch.cuda.Event()
                bs["ev_comp"].record(s_comp)
        bs["out_slot"] = next_out()                         # only the copy back needs it: uploads and kernels are already enqueued
        bs["h_out"] = bs["out_slot"]["h_out"]
        ctx_out = torch.cuda.stream(s_copy_out) if is_cuda else _null()
        with ctx_out:
            if is_cuda:
                s_copy_out.wait_event(bs["ev_comp"])
            bs["h_out"][:L.total_elems].copy_(bs["d_out"][:L.total_elems], non_blocking=True)
            if is_cuda:
                bs["h_flags"] = bs["h_flags_pinned"][:len(L.keys)]
                bs["h_flags"].copy_(flags, non_blocking=True)
            else:
                bs["h_flags"] = flags.to("cpu", non_blocking=False)
            if is_cuda:
                bs["ev_d2h"] = torch.cuda.Event()
                bs["ev_d2h"].record(s_copy_out)

    def drain(bs):
        """Wait for the wave's D2H, then hand every key to the sink (file region or dict)."""
        wave = bs["wave"]
        if wave is None:
            return
        if is_cuda:
            with _phase("drain_wait_gpu"):
                bs["ev_d2h"].synchronize()
        L = wave.layout
        fl = bs["h_flags"].numpy()
        out_base = bs["h_out"].data_ptr()
        put_keys, put_metas, put_ptrs = [], [], []
        for i, (key, off, n) in enumerate(zip(L.keys, L.offsets, L.numels)):
            meta = header[key]
            if key in bs["skip"]:                                            # no LoRA entry and no fallback: the key is omitted
                skipped.append(key)
                continue
            if fl[i] & _lib.FLAG_INPUT_NONFINITE:
                if not _write_fallback(plan, readers, key, meta, sink, to_file, output_device, output_dtype):
                    skipped.append(key)
                continue
            if (fl[i] & _lib.FLAG_OUTPUT_NONFINITE) and check_finite_output:
                log.warning("There are non finite values in key '%s': %s", key, meta)     # merging.py:303-308
            if to_file:
                put_keys.append(key)
                put_metas.append(meta)
                put_ptrs.append(out_base + off * out_esz)
            elif getattr(sink, "discard", False):
                continue                                                     # NullSink: results stay in the staging buffer (link timing)
            else:
                t = bs["h_out"][off:off + n].clone().view(meta.shape)
                sink[key] = t.to(output_device) if output_device not in (None, "cpu") else t
        if to_file:
            # one scatter list per wave, written by the host team behind the scheduler's back: the next wave's staging
            # and launches overlap the page-cache work of this write; launch() waits before it reuses this set's h_out
            def write(keys=put_keys, metas=put_metas, ptrs=put_ptrs):
                with _phase("write"):
                    sink.put_many(keys, metas, ptrs, host_threads)
            bs["out_slot"]["write_fut"] = writer.submit(write)
        bs["wave"] = None

    def run_split(wave: _Wave):
        """A key larger than the staging window.  288 GB of HBM hold the whole key for every model, so only the TRANSFER is
        split: its pieces go up through the pinned window (alternating the buffer sets: piece i + 1 is read from the files
        while piece i crosses the link), the kernels run once over the whole key in HBM -- TIES thresholds are global per key
        (ties.py:128-132), nothing about them changes --, and the result comes back through the window piece by piece into
        the key's region of the file.  Every buffer set is idle when this runs."""
        L = wave.layout
        key, n = L.keys[0], L.numels[0]
        meta = header[key]
        big_in = [torch.empty(L.total_elems, dtype=in_dtype, device=dev) for _ in range(n_models)]
        big_out = torch.empty(L.total_elems, dtype=output_dtype, device=dev)
        pieces = [(s0, min(piece_elems, n - s0)) for s0 in range(0, n, piece_elems)]
        events = []
        for i, (s0, cnt) in enumerate(pieces):
            bs = sets[i % len(sets)]
            ev = bs.pop("ev_piece", None)
            if ev is not None:
                ev.synchronize()                                       # the copy that last read this set's staging buffers
            with _phase("stage"):
                fds = np.asarray([readers[id(m)].fileno() for m in plan.models], dtype=np.int32)
                dst = np.asarray([bs["h_in"][mi].data_ptr() for mi in range(n_models)], dtype=np.uint64)
                size = np.full(n_models, cnt * in_esz, dtype=np.uint64)
                foff = np.asarray([readers[id(m)].file_offset(key) + s0 * in_esz for m in plan.models], dtype=np.int64)
                _lib.check(hostlib.mecha_host_pread_many(fds.ctypes.data, dst.ctypes.data, size.ctypes.data, foff.ctypes.data,
                                                         n_models, host_threads), "mecha_host_pread_many")
            with (torch.cuda.stream(s_copy_in) if is_cuda else _null()):
                for mi in range(n_models):
                    big_in[mi][s0:s0 + cnt].copy_(bs["h_in"][mi][:cnt], non_blocking=True)
                if is_cuda:
                    bs["ev_piece"] = torch.cuda.Event()
                    bs["ev_piece"].record(s_copy_in)
                    events.append(bs["ev_piece"])
        with (torch.cuda.stream(s_comp) if is_cuda else _null()):
            if is_cuda:
                for ev in events:
                    s_comp.wait_event(ev)
            for a in big_in:
                a[n:].zero_()                                          # the arena's tail gap stays finite
            segs_host, segs_dev = L.host_segments(), L.device_segments(dev)
            flags = torch.zeros(1, dtype=torch.int32, device=dev)
            nbytes = backend.workspace_bytes(plan, L, [tuple(meta.shape)])
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev) if nbytes else None
            skip = ()
            if plan.kind == "lora" and hasattr(backend, "run_lora_tables"):
                tables = lora_store.tables(L, header)
                skip = tables[3]
                backend.run_lora_tables(tables, big_in[0], big_out, flags)
            elif plan.kind == "lora":
                entries, skip = lora_store.entries(L, header)
                backend.run_lora(entries, big_in[0], big_out, flags)
            else:
                backend.run_fused(plan, big_in, big_out, L, segs_dev, segs_host, flags, ws, shapes=[tuple(meta.shape)])
            fl = int(flags.cpu()[0])                                   # waits for the kernels
        for bs in sets:
            bs.pop("ev_piece", None)
        if key in skip:
            skipped.append(key)
            return
        if fl & _lib.FLAG_INPUT_NONFINITE:
            if not _write_fallback(plan, readers, key, meta, sink, to_file, output_device, output_dtype):
                skipped.append(key)
            return
        if (fl & _lib.FLAG_OUTPUT_NONFINITE) and check_finite_output:
            log.warning("There are non finite values in key '%s': %s", key, meta)
        discard = getattr(sink, "discard", False)
        region = sink.reserve(key, meta) if to_file else None
        whole = None if (to_file or discard) else torch.empty(n, dtype=output_dtype)
        for i, (s0, cnt) in enumerate(pieces):
            slot = next_out()                                          # its last piece may still be on its way to the file
            with (torch.cuda.stream(s_copy_out) if is_cuda else _null()):
                slot["h_out"][:cnt].copy_(big_out[s0:s0 + cnt], non_blocking=True)
                if is_cuda:
                    ev = torch.cuda.Event()
                    ev.record(s_copy_out)
                    ev.synchronize()
            if to_file:
                def write(ptr=slot["h_out"].data_ptr(), off=region + s0 * out_esz, nb=cnt * out_esz):
                    with _phase("write"):
                        sink.write_at(off, ptr, nb, host_threads)
                slot["write_fut"] = writer.submit(write)
            elif not discard:
                whole[s0:s0 + cnt].copy_(slot["h_out"][:cnt])
        if whole is not None:
            t = whole.view(meta.shape)
            sink[key] = t.to(output_device) if output_device not in (None, "cpu") else t

    # software pipeline over waves: stage(w+1) on the host overlaps H2D/compute/D2H of w
    trace = last_stats.setdefault("trace", []) if os.environ.get("MECHA_TRACE") else None
    if trace is not None:                                     # MECHA_TRACE=1: host seconds of every stage / launch / drain call, in order
        def timed(name, fn):
            def run(*a):
                t0 = time.perf_counter()
                r = fn(*a)
                trace.append((name, round(time.perf_counter() - t0, 4)))
                return r
            return run
        stage, launch, drain = timed("stage", stage), timed("launch", launch), timed("drain", drain)
    if waves:
        with _phase("stage"):
            stage(sets[0], waves[0])
    if lora_future is not None:
        with _phase("lora_load"):
            lora_store = lora_future.result()
    for w, wave in enumerate(waves):
        bs = sets[w % len(sets)]
        with _phase("launch"):
            launch(bs, wave)
        if w + 1 < len(waves):
            nxt = sets[(w + 1) % len(sets)]
            with _phase("drain"):
                drain(nxt)                  # the set we are about to refill must have been written out
            with _phase("stage"):
                stage(nxt, waves[w + 1])
    with _phase("drain"):
        for bs in sets:
            drain(bs)
    for slot in outs:
        wait_write(slot)
    for wave in split_waves:
        with _phase("split_transfer"):
            run_split(wave)
    for slot in outs:
        wait_write(slot)
    writer.shutdown()
    if is_cuda:
        torch.cuda.synchronize(dev)
    last_stats["waves"] = last_stats.get("waves", 0) + len(waves) + len(split_waves)


def _write_fallback(plan, readers, key, meta, sink, to_file, output_device, output_dtype) -> bool:
    """`recipe | model` (logistics.py:12-35): a key whose merge was skipped comes from the fallback model, cast to the
    output dtype.  Returns False when there is no usable fallback (the key is then omitted, merging.py:312-316)."""
    if plan.fallback is None:
        return False
    reader = readers[id(plan.fallback)]
    if key not in reader:
        return False
    t = reader[key]
    if t.is_floating_point() and not bool(torch.isfinite(t.float()).all()):
        return False                                                          # the fallback's own omit_non_finite
    t = t.to(output_dtype)
    if to_file:
        sink.put(key, meta, tensor_bytes(t))
    else:
        sink[key] = t.to(output_device) if output_device not in (None, "cpu") else t
    return True


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ------------------------------------------------------------------ generic (per key) path
class _LazyStateDict(Mapping):
    """MergeNodeWrapperStateDict (merging.py:566-608): what a ``StateDict``-typed parameter receives.  Indexing it evaluates
    the input node for that key -- a body that never indexes it never loads it."""

    def __init__(self, node: RecipeNode, run: "_KeyRun", interface):
        self.node, self._run, self._interface = node, run, interface

    def __getitem__(self, key):
        return self._run.value(self.node, key, self._interface)

    def keys(self):
        ks = self._run.ex.info.keys(self.node)
        return ks.keys() if ks is not None else ()

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, key):
        ks = self._run.ex.info.keys(self.node)
        return ks is None or key in ks


class _KeyRun:
    """One key's evaluation: KeyMergeVisitor (merging.py:374-482).  `memo` holds the outputs of nodes with several consumers
    for the keys requested so far (merge_context.py:164-205 keeps them until the last consumer has read them; here they
    live until the key is done -- the same values, released together)."""

    def __init__(self, ex: "_GenericExecutor"):
        self.ex = ex
        self.memo: Dict[Tuple[int, str], Any] = {}

    def value(self, node: RecipeNode, key: str, interface=None):
        return _cast_node_value(self.node_value(node, key), interface)

    def node_value(self, node: RecipeNode, key: str):
        ex = self.ex
        shared = ex.memoize and ex.info.consumers.get(id(node), 0) > 1
        if shared and (id(node), key) in self.memo:
            hit = self.memo[(id(node), key)]
            if isinstance(hit, _Raised):
                raise hit.error
            return hit
        try:
            res = self._compute(node, key)
        except StateDictKeyError as e:
            if shared:
                self.memo[(id(node), key)] = _Raised(e)
            raise
        if shared:
            self.memo[(id(node), key)] = res
        return res

    def _compute(self, node: RecipeNode, key: str):
        ex = self.ex
        if isinstance(node, ModelRecipeNode):                                  # visit_model (merging.py:391-403) + the cast and
            reader = ex.readers[id(node)]                                      # omit_non_finite nodes above it (:673-681)
            if key not in reader:
                raise StateDictKeyError(key)
            t = reader[key].to(ex.dev)
            if t.dtype in _DT_CODE and t.numel():
                return ex.methods.cast(t, torch.float32, key=key)
            return t.to(torch.float32)
        if isinstance(node, LiteralRecipeNode):
            if isinstance(node.value, Mapping):
                if key not in node.value:
                    raise StateDictKeyError(key)
                v = node.value[key]
                if isinstance(v, torch.Tensor):
                    v = v.to(ex.dev)
                    if v.dtype in _DT_CODE and v.numel():
                        return ex.methods.cast(v, torch.float32, key=key)
                    return v.to(torch.float32)
                return v
            return node.value
        if not isinstance(node, MergeRecipeNode):
            raise TypeError(type(node))
        method = node.merge_method
        node_keys = ex.info.keys(node)
        if node_keys is not None and key not in node_keys:                     # merging.py:412-414
            raise StateDictKeyError(key)
        if lora_mod.is_conversion(node):                                       # LycorisToBase.__call__ (lycoris.py:69-85)
            lora_node, base_node = node.args[0], lora_mod.base_of(node)
            reader, meta = ex.readers[id(lora_node)], ex.readers[id(base_node)].meta(key)
            prefix = lora_mod.lora_key_prefix(key, tuple(meta.shape), meta.dtype.is_floating_point, method.lora_prefix)
            if prefix is None:
                raise StateDictKeyError(key)
            up_k, down_k, alpha_k = lora_mod.network_entries(reader, prefix)
            alpha = reader[alpha_k] if alpha_k is not None else None
            return ex.methods.lora_delta(reader[up_k].to(ex.dev), reader[down_k].to(ex.dev), alpha, tuple(meta.shape))
        # __visit_deeper_first (merging.py:446-482): StateDict-typed parameters become lazy mappings; the others are
        # evaluated now, deepest input first, every one of them even after an error (ErrorHolder :544-563)
        names = method.param_names(len(node.args))
        slots = [(i, names[i], a) for i, a in enumerate(node.args)] + [(k, k, a) for k, a in node.kwargs.items()]
        merged: Dict[Any, Any] = {}
        error = None
        for index, pname, child in sorted(slots, key=lambda s: ex.info.depth.get(id(s[2]), 0), reverse=True):
            iface = method.param_interfaces.get(pname)
            if pname in method.lazy_params:
                merged[index] = _LazyStateDict(child, self, getattr(iface, "item", None))
            else:
                try:
                    merged[index] = self.value(child, key, iface)
                except Exception as e:                                          # noqa: BLE001 -- re-raised below
                    error, merged[index] = e, None
        if error is not None:
            raise error
        args = [merged[i] for i in range(len(node.args))]
        kwargs = {k: v for k, v in merged.items() if not isinstance(k, int)}
        meta = None
        if method.identifier == "fallback":                                    # logistics.py:15-19: the inputs that can produce the key
            def can_produce(child):
                ks = ex.info.keys(child)
                return ks is None or key in ks
            meta = tuple(pname for _, pname, child in slots if can_produce(child))
        relation = RealizedKeyRelation((key,), {pname: (key,) for _, pname, _ in slots}, meta)
        return method.merge_key(args, kwargs, key, relation, ex.cache.get(node), ex.info.instances.get(id(node)), ex.memoize,
                                body=ex.bodies.get(method.identifier))


class _Raised:
    def __init__(self, error):
        self.error = error


def _cast_node_value(value, interface):
    """cast_node_value (merging.py:611-636): python values bound to a ``Tensor`` parameter become 0-d fp32 CPU tensors."""
    if value is None or interface is None or isinstance(interface, str):
        return value
    if interface is torch.Tensor:
        return value if isinstance(value, torch.Tensor) else torch.tensor(value, dtype=torch.float32)
    if isinstance(interface, type) and not isinstance(value, interface):
        if interface in (int, float, bool):
            return interface(value)
    return value


class _GenericExecutor:
    def __init__(self, info: GraphInfo, backend, readers, cache, memoize=True):
        self.info, self.readers, self.cache, self.memoize = info, readers, cache, memoize
        self.dev, self.methods = backend.device, backend.methods
        self.bodies = getattr(backend, "bodies", None) or {}           # {identifier: body}: a backend's own method bodies


def _run_generic(recipe, info, backend, readers, items, header, sink, to_file, output_device, output_dtype, check_finite_output,
                 strict_mandatory_keys, skipped, cache, threads=0):
    """KeyMergeVisitor (merging.py:374-482) for keys and graphs the fuser does not cover: every model value is uploaded,
    cast to fp32 with the fused finite check, and the method bodies run per key on device tensors."""
    dev = backend.device
    ex = _GenericExecutor(info, backend, readers, cache)

    def evaluate(node: RecipeNode, key: str):
        return _KeyRun(ex).node_value(node, key)

    lock = threading.Lock()

    def merge_one(key):
        meta = header[key]
        try:
            res = evaluate(recipe, key)
        except StateDictKeyError as e:
            if strict_mandatory_keys:
                raise RuntimeError(f"Could not merge mandatory key: {e}") from e
            with lock:
                skipped.append(key)
            return
        if not isinstance(res, torch.Tensor):
            res = torch.full(meta.shape, float(res), dtype=torch.float32, device=dev)
        if check_finite_output and res.numel() and not bool(torch.isfinite(res).all()):
            log.warning("There are non finite values in key '%s': %s", key, meta)
        res = res.to(output_dtype).reshape(meta.shape)
        if not to_file:
            res = res.to(output_device) if output_device is not None else res
        if res.is_cuda:
            torch.cuda.current_stream(res.device).synchronize()                # the sink may read it from another stream / the host
        with lock:
            sink[key] = res

    if threads and threads > 0 and dev.type == "cuda" and len(items) > 1:
        # keys are independent (merging.py:223-229): pool threads, each inside its own HIP stream (merging.py:330-341), overlap
        # the latency-bound work of different keys (rotate's chains of small GEMMs: 5.9 s -> 3.4 s for SDXL on 4 streams)
        local = threading.local()

        def task(key):
            if not hasattr(local, "stream"):
                local.stream = torch.cuda.Stream(device=dev)
            with torch.cuda.device(dev), torch.cuda.stream(local.stream):
                merge_one(key)

        with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
            futures = [pool.submit(task, key) for key, _ in items]
            done, pending = concurrent.futures.wait(futures, return_when=concurrent.futures.FIRST_EXCEPTION)   # merging.py:354-371
            for f in pending:
                f.cancel()
            for f in done:
                f.result()
            for f in pending:
                if not f.cancelled():
                    f.result()
    else:
        for key, _ in items:
            merge_one(key)
