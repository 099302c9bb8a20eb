"""Host-side mirror of sd-mecha's builtin merge methods for the MI355X backend.

Same names, argument meaning and error behaviour as the reference's method bodies
(``/root/reference/sd_mecha/extensions/builtin/merge_methods/{linear,ties,logistics}.py``), so a call
``sd_mecha.ties_sum.__wrapped__(*deltas, k=k)`` reads ``sd_mecha_amd.ties_sum(*deltas, k=k)`` here.
Tensors are torch tensors on a ROCm device; the arithmetic runs in libmecha_hip.so through the C ABI of
``include/mecha_hip.h`` (PyTorch only provides device memory and the stream).  There is no CPU path.

Differences that are extensions, not changes:
  * inputs may be fp16/bf16 (the ``cast`` node of merging.py:673-681 is fused into the kernel); the result
    dtype is ``out_dtype`` (default float32 = the reference's merge_dtype);
  * ``StateDict``-typed parameters accept a tensor, or a mapping plus ``key=`` like the reference;
  * non-finite inputs raise ``NonFiniteStateDictKeyError`` (logistics.py:132-142) when ``check_finite``
    is true (one device sync); the scheduler passes ``flags=`` instead and checks once per batch.
"""
from __future__ import annotations

import ctypes
import math
from typing import Mapping, Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, F16, F32, FLAG_INPUT_NONFINITE

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


class StateDictKeyError(KeyError):
    """A requested key is missing (reference streaming.py:181-186): the executor skips the key."""

    def __init__(self, key: str):
        super().__init__(key)


class NonFiniteStateDictKeyError(StateDictKeyError):
    """A requested key contains non-finite values (reference streaming.py:189-194)."""

    def __init__(self, key):
        super().__init__(f"key '{key}' contains non-finite values")


# ------------------------------------------------------------------ plumbing
def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _resolve(x, key):
    """StateDict-typed parameter: a tensor, or a mapping indexed by ``key`` (merging.py:579-589)."""
    if isinstance(x, Mapping):
        try:
            return x[key]
        except KeyError as e:
            raise StateDictKeyError(key) from e
    return x


def _check_inputs(tensors: Sequence[torch.Tensor], what: str) -> None:
    """Host-side validation of everything the kernels and their grids assume."""
    t0 = tensors[0]
    for i, t in enumerate(tensors):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{what}: argument {i} is {type(t).__name__}, expected a torch.Tensor")
        if not t.is_cuda:
            raise RuntimeError(f"{what}: argument {i} is on {t.device}; the MI355X backend has no CPU path")
        if t.dtype not in _DT:
            raise TypeError(f"{what}: unsupported dtype {t.dtype} (float32, float16, bfloat16)")
        if t.shape != t0.shape or t.dtype != t0.dtype or t.device != t0.device:
            raise ValueError(f"{what}: argument {i} has shape {tuple(t.shape)} / {t.dtype} / {t.device}, "
                             f"expected {tuple(t0.shape)} / {t0.dtype} / {t0.device}")


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _alpha(alpha) -> float:
    """Scalar hyper-parameters arrive as 0-d fp32 tensors in the reference (merging.py:632-633)."""
    if isinstance(alpha, torch.Tensor):
        if alpha.numel() != 1:
            raise NotImplementedError("this method takes a scalar alpha on the MI355X path (weighted_sum and add_difference take full tensors)")
        alpha = alpha.item()
    return float(torch.tensor(float(alpha), dtype=torch.float32).item())


def _is_full_alpha(alpha) -> bool:
    """linear.py:16, :71: the 0 / 1 shortcuts apply to `alpha.numel() == 1` only; anything larger is a weight per element."""
    return isinstance(alpha, torch.Tensor) and alpha.numel() > 1


def _alpha_like(alpha: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """A full alpha tensor as the kernels take it: fp32, on the device, broadcast to the shape of `a` (torch.lerp / addcmul
    broadcast it the same way; a result larger than `a` is refused -- the reference would return a bigger tensor)."""
    w = alpha.to(device=like.device, dtype=torch.float32)
    if torch.broadcast_shapes(tuple(w.shape), tuple(like.shape)) != tuple(like.shape):
        raise ValueError(f"alpha of shape {tuple(alpha.shape)} does not broadcast to the key's shape {tuple(like.shape)}")
    return w.expand(like.shape).contiguous()


class _Flags:
    """Device flag word for one call; raises like omit_non_finite / warns like _track_output."""

    def __init__(self, device, external: Optional[torch.Tensor], check: bool):
        self.external = external
        self.check = check and external is None
        self.t = external if external is not None else (torch.zeros(1, dtype=torch.int32, device=device) if check else None)

    @property
    def ptr(self):
        return ctypes.c_void_p(self.t.data_ptr()) if self.t is not None else None

    def finish(self, key):
        if self.check:
            bits = int(self.t.item())
            if bits & FLAG_INPUT_NONFINITE:
                raise NonFiniteStateDictKeyError(key)


def _out(like: torch.Tensor, out_dtype) -> torch.Tensor:
    return torch.empty(like.shape, dtype=out_dtype or torch.float32, device=like.device)


# ------------------------------------------------------------------ logistics.py
def cast(a, dtype=torch.float32, *, key=None, flags=None, check_finite=True) -> torch.Tensor:
    """cast (logistics.py:38-53) fused with omit_non_finite (logistics.py:132-142)."""
    a = _c(_resolve(a, key))
    _check_inputs([a], "cast")
    out = _out(a, dtype)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(_lib.lib().mecha_cast(a.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype], fl.ptr, _stream()), "mecha_cast")
    fl.finish(key)
    return out


def omit_non_finite(a, *, key=None) -> torch.Tensor:
    """omit_non_finite (logistics.py:132-142): raises NonFiniteStateDictKeyError, else returns ``a``."""
    a = _resolve(a, key)
    ac = _c(a)
    _check_inputs([ac], "omit_non_finite")
    fl = _Flags(a.device, None, True)
    _lib.check(_lib.lib().mecha_cast(ac.data_ptr(), None, ac.numel(), _DT[ac.dtype], _DT[ac.dtype], fl.ptr, _stream()), "mecha_cast")
    fl.finish(key)
    return a


# ------------------------------------------------------------------ linear.py
def weighted_sum(a, b, alpha=0.5, *, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """weighted_sum (linear.py:7-23).  alpha == 0 returns a, alpha == 1 returns b, without touching the other; a full alpha
    tensor (block weights: one weight per row / element) is lerp with a weight per element."""
    if _is_full_alpha(alpha):
        a, b = _c(_resolve(a, key)), _c(_resolve(b, key))
        _check_inputs([a, b], "weighted_sum")
        w = _alpha_like(alpha, a)
        out = _out(a, out_dtype)
        fl = _Flags(a.device, flags, check_finite)
        _lib.check(_lib.lib().mecha_weighted_sum_tensor(a.data_ptr(), b.data_ptr(), w.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype],
                                                        _DT[out.dtype], fl.ptr, _stream()), "mecha_weighted_sum_tensor")
        fl.finish(key)
        return out
    al = _alpha(alpha)
    if math.isclose(al, 0.0):
        return cast(a, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
    if math.isclose(al, 1.0):
        return cast(b, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
    a, b = _c(_resolve(a, key)), _c(_resolve(b, key))
    _check_inputs([a, b], "weighted_sum")
    out = _out(a, out_dtype)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(_lib.lib().mecha_weighted_sum(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype],
                                             al, fl.ptr, _stream()), "mecha_weighted_sum")
    fl.finish(key)
    return out


def n_average(*models, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """n_average (linear.py:26-40)."""
    assert models
    ms = [_c(_resolve(m, key)) for m in models]
    _check_inputs(ms, "n_average")
    if len(ms) > _lib.MAX_MODELS:
        raise ValueError(f"n_average: at most {_lib.MAX_MODELS} models per call")
    out = _out(ms[0], out_dtype)
    fl = _Flags(ms[0].device, flags, check_finite)
    _lib.check(_lib.lib().mecha_n_average(_lib.ptr_array([m.data_ptr() for m in ms]), len(ms), out.data_ptr(), out.numel(),
                                          _DT[ms[0].dtype], _DT[out.dtype], fl.ptr, _stream()), "mecha_n_average")
    fl.finish(key)
    return out


def add_difference(a, b, alpha=1.0, *, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """add_difference (linear.py:64-76): a.addcmul(b, alpha); alpha == 0 returns a; a full alpha tensor weighs every element."""
    if _is_full_alpha(alpha):
        b = _c(_resolve(b, key))
        a = _c(_resolve(a, key))
        _check_inputs([a, b], "add_difference")
        w = _alpha_like(alpha, a)
        out = _out(a, out_dtype)
        fl = _Flags(a.device, flags, check_finite)
        _lib.check(_lib.lib().mecha_add_difference_tensor(a.data_ptr(), b.data_ptr(), w.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype],
                                                          _DT[out.dtype], fl.ptr, _stream()), "mecha_add_difference_tensor")
        fl.finish(key)
        return out
    al = _alpha(alpha)
    if math.isclose(al, 0.0):
        return cast(a, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
    b = _c(_resolve(b, key))      # b first, like the reference (linear.py:75)
    a = _c(_resolve(a, key))
    _check_inputs([a, b], "add_difference")
    out = _out(a, out_dtype)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(_lib.lib().mecha_add_difference(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype],
                                               al, fl.ptr, _stream()), "mecha_add_difference")
    fl.finish(key)
    return out


def subtract(a, b, *, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """subtract (linear.py:96-101)."""
    a, b = _c(_resolve(a, key)), _c(_resolve(b, key))
    _check_inputs([a, b], "subtract")
    out = _out(a, out_dtype)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(_lib.lib().mecha_subtract(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype],
                                         fl.ptr, _stream()), "mecha_subtract")
    fl.finish(key)
    return out


def scale(a, alpha=1.0, *, key=None, out_dtype=None, flags=None, check_finite=True):
    """scale (linear.py:79-93); alpha == 0 returns python 0.0 like the reference."""
    al = _alpha(alpha)
    if math.isclose(al, 0.0):
        return 0.0
    a = _c(_resolve(a, key))
    _check_inputs([a], "scale")
    out = _out(a, out_dtype)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(_lib.lib().mecha_scale(a.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype], al, fl.ptr, _stream()),
               "mecha_scale")
    fl.finish(key)
    return out


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def slerp(a, b, alpha=0.5, *, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """slerp (linear.py:43-61)."""
    a, b = _c(_resolve(a, key)), _c(_resolve(b, key))
    _check_inputs([a, b], "slerp")
    out = _out(a, out_dtype)
    L = _lib.lib()
    ws = _workspace(L.mecha_reduce_workspace_bytes(a.numel()), a.device)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(L.mecha_slerp(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype], _alpha(alpha),
                             ws.data_ptr(), ws.numel(), fl.ptr, _stream()), "mecha_slerp")
    fl.finish(key)
    return out


def perpendicular_component(a, b, *, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """perpendicular_component (linear.py:104-113)."""
    a, b = _c(_resolve(a, key)), _c(_resolve(b, key))
    _check_inputs([a, b], "perpendicular_component")
    out = _out(a, out_dtype)
    L = _lib.lib()
    ws = _workspace(L.mecha_reduce_workspace_bytes(a.numel()), a.device)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(L.mecha_perpendicular_component(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype], _DT[out.dtype],
                                               ws.data_ptr(), ws.numel(), fl.ptr, _stream()), "mecha_perpendicular_component")
    fl.finish(key)
    return out


def add_perpendicular(a, b, c, alpha=1.0, *, key=None, out_dtype=None, flags=None, check_finite=True) -> torch.Tensor:
    """The recipe of wrappers.py:42-64, fused: a + alpha * perpendicular_component(a - c, b - c)."""
    al = _alpha(alpha)
    if math.isclose(al, 0.0):
        return cast(a, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
    a, b, c = _c(_resolve(a, key)), _c(_resolve(b, key)), _c(_resolve(c, key))
    _check_inputs([a, b, c], "add_perpendicular")
    out = _out(a, out_dtype)
    L = _lib.lib()
    ws = _workspace(L.mecha_reduce_workspace_bytes(a.numel()), a.device)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(L.mecha_add_perpendicular(a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(), a.numel(), _DT[a.dtype],
                                         _DT[out.dtype], al, ws.data_ptr(), ws.numel(), fl.ptr, _stream()), "mecha_add_perpendicular")
    fl.finish(key)
    return out


# ------------------------------------------------------------------ ties.py
def _thresholds(n_models: int, device, want: bool):
    return torch.empty(n_models, dtype=torch.float32, device=device) if want else None


def ties_sum(*models, k: float = 1.0, vote_sgn: bool = False, out_dtype=None, flags=None, check_finite=True,
             return_thresholds: bool = False):
    """ties_sum (ties.py:98-125).  No models -> python 0; empty tensor -> returned as is."""
    if not models:
        return 0
    if models[0].numel() == 0:
        return models[0]
    ms = [_c(m) for m in models]
    _check_inputs(ms, "ties_sum")
    if len(ms) > _lib.MAX_MODELS:
        raise ValueError(f"ties_sum: at most {_lib.MAX_MODELS} deltas per call")
    out = _out(ms[0], out_dtype)
    L = _lib.lib()
    ws = _workspace(L.mecha_ties_workspace_bytes(len(ms), out.numel()), out.device)
    thr = _thresholds(len(ms), out.device, return_thresholds)
    fl = _Flags(out.device, flags, check_finite)
    _lib.check(L.mecha_ties_sum(_lib.ptr_array([m.data_ptr() for m in ms]), len(ms), out.data_ptr(), out.numel(), _DT[ms[0].dtype],
                                _DT[out.dtype], float(k), int(bool(vote_sgn)), thr.data_ptr() if thr is not None else None,
                                ws.data_ptr(), ws.numel(), fl.ptr, _stream()), "mecha_ties_sum")
    fl.finish(None)
    return (out, thr) if return_thresholds else out


def add_difference_ties(base, *models, k: float = 1.0, alpha=1.0, vote_sgn: bool = False, key=None, out_dtype=None,
                        flags=None, check_finite=True, return_thresholds: bool = False):
    """The recipe of wrappers.py:70-94 fused into one pipeline:
    add_difference(base, ties_sum(m_0 - base, ..., k), alpha).  Reads bf16/fp16 inputs directly.  Defaults are the
    wrapper's (wrappers.py:70-75): k = 1.0 keeps every delta element, alpha = 1.0."""
    al = _alpha(alpha)
    if math.isclose(al, 0.0) or not models:
        r = cast(base, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
        return (r, None) if return_thresholds else r
    ts = [_c(_resolve(t, key)) for t in (base, *models)]
    _check_inputs(ts, "add_difference_ties")
    if ts[0].numel() == 0:
        r = _out(ts[0], out_dtype)
        return (r, None) if return_thresholds else r
    out = _out(ts[0], out_dtype)
    L = _lib.lib()
    n = len(ts) - 1
    ws = _workspace(L.mecha_ties_workspace_bytes(n, out.numel()), out.device)
    thr = _thresholds(n, out.device, return_thresholds)
    fl = _Flags(out.device, flags, check_finite)
    _lib.check(L.mecha_add_difference_ties(ts[0].data_ptr(), _lib.ptr_array([m.data_ptr() for m in ts[1:]]), n, out.data_ptr(),
                                           out.numel(), _DT[ts[0].dtype], _DT[out.dtype], float(k), al, int(bool(vote_sgn)),
                                           thr.data_ptr() if thr is not None else None, ws.data_ptr(), ws.numel(), fl.ptr, _stream()),
               "mecha_add_difference_ties")
    fl.finish(key)
    return (out, thr) if return_thresholds else out


def dare_draw_masks(seed: Optional[int], probability: float, n_models: int, numel: int, device) -> torch.Tensor:
    """The keep masks the reference draws with torch's CPU generator (ties.py:29-33, :63-64), bit-packed,
    drawn on the host by libmecha_hip and uploaded: int32 [n_models, ceil(numel / 32)].  seed None or < 0: the stream of a
    generator that was never seeded."""
    words = (numel + 31) // 32
    host = torch.empty((n_models, words), dtype=torch.int32).pin_memory() if torch.cuda.is_available() else torch.empty((n_models, words), dtype=torch.int32)
    _lib.check(_lib.lib().mecha_dare_draw_masks_host(_dare_seed(seed) & 0xFFFFFFFFFFFFFFFF, float(probability), n_models, numel, host.data_ptr()),
               "mecha_dare_draw_masks_host")
    return host.to(device, non_blocking=False)


# torch.Generator()'s state before any manual_seed: the reference creates a fresh generator per key and seeds it only for
# seed >= 0 (ties.py:29-31), so seed=None draws the SAME masks every time, those of torch's default seed.  The caller's global
# RNG is never touched.
_TORCH_DEFAULT_SEED = 67280421310721


def _dare_seed(seed) -> int:
    return _TORCH_DEFAULT_SEED if seed is None or seed < 0 else int(seed)


def _ties_extended(ds, keep_bits, k, vote_sgn, model_stock, cos_eps, probability, rescale, out_dtype, flags, check_finite,
                   return_thresholds=False):
    n, numel = len(ds), ds[0].numel()
    if model_stock:
        if ds[0].dim() < 1:
            raise ValueError("the model-stock factor takes the cosine along the last dimension: 0-d keys have none "
                             "(torch.nn.CosineSimilarity(dim=-1) raises in the reference)")
        if not 2 <= n <= 8:
            raise NotImplementedError(f"the model-stock factor on MI355X takes 2..8 deltas (got {n}; the reference itself raises for 1)")
    row_len = ds[0].shape[-1] if ds[0].dim() >= 1 else 1
    if keep_bits is not None:
        if keep_bits.dtype != torch.int32 or tuple(keep_bits.shape) != (n, (numel + 31) // 32) or keep_bits.device != ds[0].device:
            raise ValueError("keep_bits must be int32 [n_models, ceil(numel/32)] on the deltas' device")
        keep_bits = _c(keep_bits)
    out = _out(ds[0], out_dtype)
    L = _lib.lib()
    ws = _workspace(L.mecha_ties_extended_workspace_bytes(n, numel, row_len), out.device)
    thr = _thresholds(n, out.device, return_thresholds)
    fl = _Flags(out.device, flags, check_finite)
    _lib.check(L.mecha_ties_sum_extended(_lib.ptr_array([d.data_ptr() for d in ds]), n, keep_bits.data_ptr() if keep_bits is not None else None,
                                         out.data_ptr(), numel, row_len, _DT[ds[0].dtype], _DT[out.dtype], float(k), int(bool(vote_sgn)),
                                         int(bool(model_stock)), float(cos_eps), float(probability), int(bool(rescale)),
                                         thr.data_ptr() if thr is not None else None, ws.data_ptr(), ws.numel(), fl.ptr, _stream()),
               "mecha_ties_sum_extended")
    fl.finish(None)
    return (out, thr) if return_thresholds else out


def ties_sum_extended(*models, k: float = 0.2, vote_sgn: bool = False, apply_stock: bool = False, apply_median: bool = False,
                      cos_eps: float = 1e-6, eps: float = 1e-6, maxiter: int = 100, ftol: float = 1e-20, out_dtype=None, flags=None,
                      check_finite=True):
    """ties_sum_extended (ties.py:68-94) with apply_median == False.  apply_stock == False (the default) APPLIES the
    model-stock factor (ties.py:90, restated literally)."""
    if not models:
        return 0
    if apply_median:
        raise NotImplementedError("the geometric median (ties.py:162-199) stays on the reference")
    ds = [_c(m) for m in models]
    _check_inputs(ds, "ties_sum_extended")
    if ds[0].numel() == 0:
        return ds[0]
    return _ties_extended(ds, None, k, vote_sgn, not apply_stock, cos_eps, 0.0, False, out_dtype, flags, check_finite)


def ties_sum_with_dropout(*deltas, probability=0.9, della_eps: float = 0.0, rescale: bool = True, k: float = 0.2,
                          vote_sgn: bool = False, apply_stock: bool = False, cos_eps: float = 1e-6, apply_median: bool = False,
                          eps: float = 1e-6, maxiter: int = 100, ftol: float = 1e-20, seed: Optional[int] = None,
                          out_dtype=None, flags=None, check_finite=True, keep_bits: Optional[torch.Tensor] = None):
    """DARE: ties_sum_with_dropout (ties.py:10-50).  In scope on MI355X: della_eps == 0 and apply_median == False.
    Note ties.py:90: the reference APPLIES the model-stock factor when apply_stock is False (its default); that path takes
    2..8 deltas of at least one dimension.  Masks follow the reference's CPU generator stream for seed >= 0 (pass
    ``keep_bits`` to supply your own)."""
    p = _alpha(probability)
    if not deltas or math.isclose(p, 1.0):
        return 0
    if not math.isclose(della_eps, 0.0) or apply_median:
        raise NotImplementedError("MI355X DARE path covers della_eps == 0 and apply_median == False (SURVEY §8 a15); "
                                  "DELLA ranks and the geometric median stay on the reference")
    ds = [_c(d) for d in deltas]
    _check_inputs(ds, "ties_sum_with_dropout")
    if ds[0].numel() == 0:
        return ds[0]
    n, numel = len(ds), ds[0].numel()
    if not apply_stock:
        seed = _dare_seed(seed)
        bits = keep_bits if keep_bits is not None else dare_draw_masks(seed, p, n, numel, ds[0].device)
        return _ties_extended(ds, bits, k, vote_sgn, True, cos_eps, p, rescale, out_dtype, flags, check_finite)
    if keep_bits is None:
        keep_bits = dare_draw_masks(_dare_seed(seed), p, n, numel, ds[0].device)
    if keep_bits.dtype != torch.int32 or tuple(keep_bits.shape) != (n, (numel + 31) // 32) or keep_bits.device != ds[0].device:
        raise ValueError("keep_bits must be int32 [n_models, ceil(numel/32)] on the deltas' device")
    keep_bits = _c(keep_bits)
    out = _out(ds[0], out_dtype)
    L = _lib.lib()
    ws = _workspace(L.mecha_ties_workspace_bytes(n, numel), out.device)
    fl = _Flags(out.device, flags, check_finite)
    _lib.check(L.mecha_ties_sum_with_dropout(_lib.ptr_array([d.data_ptr() for d in ds]), n, keep_bits.data_ptr(), out.data_ptr(), numel,
                                             _DT[ds[0].dtype], _DT[out.dtype], float(k), int(bool(vote_sgn)), p, int(bool(rescale)),
                                             None, ws.data_ptr(), ws.numel(), fl.ptr, _stream()), "mecha_ties_sum_with_dropout")
    fl.finish(None)
    return out


def model_stock(*deltas, cos_eps: float = 1e-6, out_dtype=None, flags=None, check_finite=True):
    """model_stock (ties.py:137-143): nan_to_num(t * (sum(deltas) / n)), t the Model-Stock factor per slice of the last dimension
    (ties.py:146-159).  2..8 deltas of at least one dimension (the reference raises for one delta and for 0-d keys)."""
    ds = [_c(d) for d in deltas]
    _check_inputs(ds, "model_stock")
    n = len(ds)
    if ds[0].dim() < 1:
        raise ValueError("model_stock takes the cosine along the last dimension: 0-d keys have none "
                         "(torch.nn.CosineSimilarity(dim=-1) raises in the reference)")
    if not 2 <= n <= 8:
        raise NotImplementedError(f"model_stock on MI355X takes 2..8 deltas (got {n}; the reference itself raises for 1)")
    out = _out(ds[0], out_dtype)
    if out.numel() == 0:
        return out
    row_len = ds[0].shape[-1]
    L = _lib.lib()
    ws = _workspace(L.mecha_model_stock_workspace_bytes(n, out.numel(), row_len), out.device)
    fl = _Flags(out.device, flags, check_finite)
    _lib.check(L.mecha_model_stock(_lib.ptr_array([d.data_ptr() for d in ds]), n, out.data_ptr(), out.numel(), row_len, _DT[ds[0].dtype],
                                   _DT[out.dtype], float(cos_eps), ws.data_ptr(), ws.numel(), fl.ptr, _stream()), "mecha_model_stock")
    fl.finish(None)
    return out


def dropout_draw_masks(seed, probability: float, n_models: int, shape, device) -> torch.Tensor:
    """The masks of ties.py:214-217, drawn on the host the way the reference draws them -- numpy's
    ``default_rng(seed).binomial(n=1, p=1 - probability, size=shape)`` once per delta, in order -- and bit-packed:
    int32 [n_models, ceil(numel / 32)], bit (i & 31) of word (i >> 5) = element i."""
    import numpy as np
    rng = np.random.default_rng(seed)
    numel = int(np.prod(shape, dtype=np.int64))
    words = (numel + 31) // 32
    packed = np.zeros((n_models, words * 4), dtype=np.uint8)
    for m in range(n_models):
        mask = rng.binomial(n=1, p=1 - probability, size=tuple(shape)).reshape(-1).astype(bool)
        b = np.packbits(mask, bitorder="little")
        packed[m, :b.size] = b
    return torch.from_numpy(packed.view("<u4").view(np.int32).reshape(n_models, words).copy()).to(device)


def dropout(*deltas, probability: float = 0.9, rescale: float = 1.0, overlap: float = 1.0, overlap_emphasis: float = 0.0,
            seed: Optional[int] = None, out_dtype=None, flags=None, check_finite=True, keep_bits: Optional[torch.Tensor] = None):
    """dropout, "n-supermario" (ties.py:202-238) at odd whole ``overlap`` (the default 1.0): every delta is masked by its own
    Bernoulli(1 - probability) mask, the kept values are averaged over the masks that kept the element, the result is divided
    by (1 - probability) ** rescale.  Other ``overlap`` values take the reference through ``torch.lerp`` on numpy arrays
    (ties.py:268-272), which raises: not supported here either."""
    if len(deltas) == 0:
        return 0
    if not (float(overlap).is_integer() and int(overlap) % 2 == 1):
        raise NotImplementedError("dropout on MI355X covers odd whole `overlap` (independent masks, the reference default); the "
                                  "overlapping-set sampler of ties.py:218-223 fails upstream (torch.lerp on numpy arrays)")
    ds = [_c(d) for d in deltas]
    _check_inputs(ds, "dropout")
    if len(ds) > _lib.MAX_MODELS:
        raise ValueError(f"dropout: at most {_lib.MAX_MODELS} deltas per call")
    probability = float(probability)
    out = _out(ds[0], out_dtype)
    numel = out.numel()
    if numel == 0:
        return out
    if keep_bits is None:
        keep_bits = dropout_draw_masks(seed, probability, len(ds), ds[0].shape, ds[0].device)
    if keep_bits.dtype != torch.int32 or tuple(keep_bits.shape) != (len(ds), (numel + 31) // 32) or keep_bits.device != ds[0].device:
        raise ValueError("keep_bits must be int32 [n_models, ceil(numel/32)] on the deltas' device")
    if probability == 1.0:                                          # ties.py:231-235
        rescalar = 1.0
    else:
        try:
            rescalar = (1.0 - probability) ** rescale
        except (OverflowError, ZeroDivisionError):
            rescalar = float("inf")
        rescalar = rescalar if math.isfinite(rescalar) else 1
    fl = _Flags(out.device, flags, check_finite)
    _lib.check(_lib.lib().mecha_dropout(_lib.ptr_array([d.data_ptr() for d in ds]), len(ds), _c(keep_bits).data_ptr(), out.data_ptr(), numel,
                                        _DT[ds[0].dtype], _DT[out.dtype], float(rescalar), fl.ptr, _stream()), "mecha_dropout")
    fl.finish(None)
    return out


# ------------------------------------------------------------------ lycoris.py
def _lora_network(up: torch.Tensor, down: torch.Tensor, alpha):
    """(up2d, down2d, rank, scale) of one network: up.view(out, -1), down.view(r, -1), alpha / r or 1 (lycoris.py:228-263)."""
    if up.dim() < 2 or down.dim() < 2:
        raise ValueError("lora_up / lora_down must be at least 2-D")
    rank = down.shape[0]
    up2, down2 = _c(up).reshape(up.shape[0], -1), _c(down).reshape(rank, -1)
    if up2.shape[1] != rank:
        raise ValueError(f"lora_up {tuple(up.shape)} does not match lora_down {tuple(down.shape)} (rank {rank}); "
                         "lora_mid / Tucker networks are outside the MI355X path")
    if alpha is None:
        scale = 1.0
    else:
        a = torch.as_tensor(alpha).detach().to("cpu", torch.float32)
        scale = float((a / rank).item())                    # 0-d fp32 tensor divided by a python int (lycoris.py:263)
    return up2, down2, rank, scale


def lora_merge(base: Optional[torch.Tensor], networks, target_shape=None, *, key=None, out_dtype=None, flags=None,
               check_finite=True) -> torch.Tensor:
    """The LoRA recipe of SURVEY.md §3.4 for one key, every network in one pass over the base tensor:
    ``for (up, down, alpha, weight) in networks: base = add_difference(base, compose(up, down, alpha), weight)``
    (lycoris.py:226-263 + linear.py:64-76).  ``base=None`` returns the (weighted sum of the) deltas in ``target_shape``."""
    nets = [_lora_network(up, down, alpha) + (float(w),) for up, down, alpha, w in networks]
    if not nets:
        raise ValueError("lora_merge needs at least one network")
    rows, cols = nets[0][0].shape[0], nets[0][1].shape[1]
    lora_dtype = nets[0][0].dtype
    for up2, down2, rank, _, _ in nets:
        if up2.shape[0] != rows or down2.shape[1] != cols or up2.dtype != lora_dtype or down2.dtype != lora_dtype:
            raise ValueError("all networks of one key must produce the same [out, in] delta and share a dtype")
        if not up2.is_cuda or not down2.is_cuda:
            raise RuntimeError("LoRA factors must be on the ROCm device; the MI355X backend has no CPU path")
    if lora_dtype not in (torch.float16, torch.bfloat16):
        raise NotImplementedError(f"LoRA factors of dtype {lora_dtype}: the MFMA path takes float16 / bfloat16 factors")
    if len(nets) > _lib.MAX_MODELS:
        raise ValueError(f"at most {_lib.MAX_MODELS} networks per call")
    if base is not None:
        base = _c(base)
        _check_inputs([base], "lora_merge")
        if base.numel() != rows * cols:
            raise ValueError(f"base has {base.numel()} elements, the networks produce {rows} x {cols}")
        shape = base.shape
    else:
        shape = tuple(target_shape) if target_shape is not None else (rows, cols)
    dev = nets[0][0].device
    out = torch.empty(shape, dtype=out_dtype or torch.float32, device=dev)
    if out.numel() != rows * cols:
        raise ValueError(f"target shape {tuple(shape)} does not hold {rows} x {cols} elements")
    n = len(nets)
    ranks = (ctypes.c_int32 * n)(*[t[2] for t in nets])
    scales = (ctypes.c_float * n)(*[t[3] for t in nets])
    weights = (ctypes.c_float * n)(*[t[4] for t in nets])
    fl = _Flags(dev, flags, check_finite)
    _lib.check(_lib.lib().mecha_lora_accumulate(
        base.data_ptr() if base is not None else None, out.data_ptr(), rows, cols, _DT[base.dtype] if base is not None else F32,
        _DT[out.dtype], _lib.ptr_array([t[0].data_ptr() for t in nets]), _lib.ptr_array([t[1].data_ptr() for t in nets]),
        ranks, scales, weights, n, _DT[lora_dtype], fl.ptr, _stream()), "mecha_lora_accumulate")
    fl.finish(key)
    return out


class LoraBatch:
    """The LoRA recipe for every key of a batch in ONE launch (mecha_batched_lora_accumulate): base and output keys live
    in two arenas at the same element offsets.  ``entries``: ``[(offset, rows, cols, [(up, down, alpha, weight), ...])]``
    with offsets in elements; the descriptor tables are built once and reused by every ``run``."""

    def __init__(self, entries, device):
        keys, nets, self._keep = [], [], []
        tile = 0
        self.lora_dtype = None
        for offset, rows, cols, networks in entries:
            begin = len(nets)
            for up, down, alpha, w in networks:
                if math.isclose(float(w), 0.0):
                    continue                                            # add_difference alpha == 0 returns `a` (linear.py:72-73)
                up2, down2, rank, scale = _lora_network(up, down, alpha)
                if up2.shape[0] != rows or down2.shape[1] != cols:
                    raise ValueError(f"a network of the key at offset {offset} does not produce {rows} x {cols}")
                if up2.dtype not in (torch.float16, torch.bfloat16) or down2.dtype != up2.dtype or (self.lora_dtype or up2.dtype) != up2.dtype:
                    raise NotImplementedError("LoRA factors of one batch must share one of float16 / bfloat16")
                self.lora_dtype = up2.dtype
                self._keep += [up2, down2]
                nets.append(_lib.LoraNet(up2.data_ptr(), down2.data_ptr(), rank, scale, float(w), 0))
            if len(nets) - begin > _lib.MAX_MODELS:
                raise ValueError(f"at most {_lib.MAX_MODELS} networks per key")
            keys.append(_lib.LoraKey(int(offset), int(rows), int(cols), tile, begin, len(nets) - begin))
            tile += ((cols + _lib.LORA_TILE - 1) // _lib.LORA_TILE) * ((rows + _lib.LORA_TILE - 1) // _lib.LORA_TILE)
        self.n_keys = len(keys)
        self.keys_host = (_lib.LoraKey * max(len(keys), 1))(*keys)
        self.nets_host = nets_host = (_lib.LoraNet * max(len(nets), 1))(*nets)
        self.keys_dev = torch.frombuffer(bytearray(bytes(self.keys_host)), dtype=torch.uint8).to(device)
        self.nets_dev = torch.frombuffer(bytearray(bytes(nets_host)), dtype=torch.uint8).to(device)

    def run(self, base_arena: Optional[torch.Tensor], out_arena: torch.Tensor, key_flags: Optional[torch.Tensor] = None) -> None:
        if self.n_keys == 0:
            return
        _lib.check(_lib.lib().mecha_batched_lora_accumulate(
            base_arena.data_ptr() if base_arena is not None else None, out_arena.data_ptr(), self.keys_dev.data_ptr(),
            ctypes.addressof(self.keys_host), self.n_keys, self.nets_dev.data_ptr(), ctypes.addressof(self.nets_host),
            _DT[base_arena.dtype] if base_arena is not None else F32, _DT[out_arena.dtype], _DT[self.lora_dtype or torch.float16],
            key_flags.data_ptr() if key_flags is not None else None, _stream()), "mecha_batched_lora_accumulate")


# ------------------------------------------------------------------ svd.py
def rotate_shape_2d(shape) -> tuple:
    """svd.py:49-53: conv k x k keys flatten to (out * in, kh * kw), everything else to (out, rest)."""
    shape = tuple(shape)
    is_conv = len(shape) == 4 and shape[2] * shape[3] != 1
    if is_conv:
        return shape[0] * shape[1], shape[2] * shape[3]
    rest = 1
    for s in shape[1:]:
        rest *= s
    return shape[0], rest


def rotate(a, b, alignment: float = 1.0, alpha=0.0, centralization: float = 1.0, stiefel_eps: float = 1e-8,
           stiefel_max_iters: int = 100, *, key=None, cache=None, out_dtype=None, flags=None, check_finite=True,
           return_sweeps: bool = False):
    """rotate (svd.py:10-86).  On the MI355X path: alignment in {-1, 0, 1}, scalar alpha, and the factor ``cache``
    (svd.py:55-79: a dict per key holding the transform in fp16 on the host and the centralization it was made for);
    the early-outs of svd.py:37-47 are kept.  Fractional alignment (matrix logarithms / eigendecompositions,
    svd.py:184-188, :216-236) and per-element alpha raise NotImplementedError."""
    al = _alpha(alpha)
    if math.isclose(alignment, 0.0) and math.isclose(al, 0.0):
        return cast(a, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
    if math.isclose(alignment, 1.0) and math.isclose(al, 1.0):
        return cast(b, out_dtype or torch.float32, key=key, flags=flags, check_finite=check_finite)
    a, b = _c(_resolve(a, key)), _c(_resolve(b, key))
    _check_inputs([a, b], "rotate")
    L = _lib.lib()
    if a.dim() > 1 and a.numel() > 0:
        word = torch.zeros(1, dtype=torch.int32, device=a.device)
        _lib.check(L.mecha_allclose_half(a.data_ptr(), b.data_ptr(), a.numel(), _DT[a.dtype], word.data_ptr(), _stream()), "mecha_allclose_half")
        same = int(word.item()) == 0
    else:
        same = True
    if a.dim() <= 1 or same:
        return weighted_sum(a, b, al, key=key, out_dtype=out_dtype, flags=flags, check_finite=check_finite)   # torch.lerp(a, b, alpha), svd.py:47
    whole = round(alignment)
    if not math.isclose(alignment, whole) or whole not in (-1, 0, 1):
        raise NotImplementedError("MI355X rotate covers alignment in {-1, 0, 1}; fractional powers of the transform "
                                  "(svd.py:184-188, :216-236) stay on the reference (SURVEY §8 a16)")
    n, p = rotate_shape_2d(a.shape)
    if cache is not None:                                       # svd.py:55-63
        cache = cache[key]
        if "centralization" in cache:
            if not math.isclose(cache["centralization"], centralization):
                cache.clear()
        else:
            cache["centralization"] = centralization
    t_floats = int(L.mecha_rotate_transform_floats(n, p))
    t_in = t_out = None
    if cache is not None:
        saved = cache.get("transform")
        if saved is not None:                                   # svd.py:74-75
            if saved.numel() != t_floats:
                raise ValueError(f"rotate: the cached transform of {key!r} has {saved.numel()} elements, this key needs {t_floats}")
            t_in = saved.to(device=a.device, dtype=torch.float32).contiguous()
        else:
            t_out = torch.empty(t_floats, dtype=torch.float32, device=a.device)
    out = _out(a, out_dtype)
    ws = _workspace(L.mecha_rotate_workspace_bytes(n, p), a.device)
    sweeps = ctypes.c_int(0)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(L.mecha_rotate_ex(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, p, _DT[a.dtype], _DT[out.dtype], float(centralization),
                                 int(whole), float(al), t_in.data_ptr() if t_in is not None else None,
                                 t_out.data_ptr() if t_out is not None else None, ws.data_ptr(), ws.numel(), ctypes.byref(sweeps),
                                 fl.ptr, _stream()), "mecha_rotate_ex")
    fl.finish(key)
    if t_out is not None:
        cache["transform"] = t_out.to(device="cpu", dtype=torch.float16)     # svd.py:79
    return (out, sweeps.value) if return_sweeps else out


def truncate_rank(a, rank: int = 8, use_approximate_basis: bool = True, approximate_basis_iters: int = 2, approximate_basis_seed=None, *,
                  key=None, cache=None, out_dtype=None, flags=None, check_finite=True, return_trials: bool = False):
    """truncate_rank (svd.py:89-143): the best rank-``rank`` approximation of ``a.flatten(start_dim=1)``, reshaped.  The MI355X path
    computes the exact (Eckart-Young) matrix -- what the reference returns for ``use_approximate_basis=False`` -- for either value of
    the flag: the reference's default is a randomised sketch of that same matrix (svd.py:126-127, :239-265; unseeded by default),
    so it has no parity target of its own, and ``approximate_basis_iters`` / ``approximate_basis_seed`` only steer that sketch.
    ``cache`` (svd.py:102-107, :118-142: the reference keeps u, s, vh in fp16 per key) is accepted and not used.  Early-outs as the
    reference: keys of < 2 dimensions and ``rank >= min(shape)`` return ``a`` (svd.py:99-100, :113-114), ``rank <= 0`` zeros (:115-116)."""
    a = _c(_resolve(a, key))
    _check_inputs([a], "truncate_rank")
    out_dt = out_dtype or torch.float32
    if a.dim() < 2:
        res = a if out_dt == a.dtype else cast(a, out_dt, key=key, flags=flags, check_finite=check_finite)
        return (res, 0) if return_trials else res
    m, n = a.shape[0], (a.numel() // a.shape[0] if a.shape[0] else 0)
    max_rank = min(m, n)
    target = min(max(round(rank), 0), max_rank)
    if target == max_rank:
        res = a if out_dt == a.dtype else cast(a, out_dt, key=key, flags=flags, check_finite=check_finite)
        return (res, 0) if return_trials else res
    if target == 0:
        res = torch.zeros_like(a, dtype=out_dt)
        return (res, 0) if return_trials else res
    L = _lib.lib()
    out = _out(a, out_dt)
    ws = _workspace(L.mecha_truncate_rank_workspace_bytes(m, n), a.device)
    trials = ctypes.c_int(0)
    fl = _Flags(a.device, flags, check_finite)
    _lib.check(L.mecha_truncate_rank(a.data_ptr(), out.data_ptr(), m, n, int(target), _DT[a.dtype], _DT[out.dtype], ws.data_ptr(), ws.numel(),
                                     ctypes.byref(trials), fl.ptr, _stream()), "mecha_truncate_rank")
    fl.finish(key)
    return (out, trials.value) if return_trials else out


def lora_delta(up: torch.Tensor, down: torch.Tensor, alpha=None, target_shape=None, *, out_dtype=None) -> torch.Tensor:
    """LoraAlgorithm.compose / compose_alpha (lycoris.py:226-263): (up @ down) * (alpha / r or 1), reshaped."""
    return lora_merge(None, [(up, down, alpha, 1.0)], target_shape, out_dtype=out_dtype)
