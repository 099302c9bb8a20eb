"""numpy front-end of the CPU oracle (oracle/mecha_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing under ``sd_mecha_amd/`` imports this module.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the reported CPU baseline.

Each function restates one reference symbol (cited ``file:line`` relative to
``/root/reference/sd_mecha``), including the Python-level shortcuts
(``math.isclose`` early-outs, empty-tensor and zero-model cases) that sit around
the torch ops.  Tensors are numpy arrays: fp32 for merge-space arithmetic,
``uint16`` bit patterns for fp16/bf16 storage.

Parity pinned: yes -- see tests/test_oracle_golden.py and tests/golden/gen_golden.py.
"""
from __future__ import annotations

import ctypes
import math
import os
import pathlib
import subprocess
from typing import Sequence

import numpy as np

F32, F16, BF16 = 0, 1, 2
DTYPE_CODE = {"float32": F32, "float16": F16, "bfloat16": BF16, "F32": F32, "F16": F16, "BF16": BF16}

_HERE = pathlib.Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libmecha_oracle.so"
_lib = None


def build(force: bool = False) -> pathlib.Path:
    """Compile oracle/libmecha_oracle.so with gcc (a no-op when it is up to date)."""
    src = _HERE / "mecha_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = ctypes.CDLL(str(_LIB_PATH))
        c_f, c_i64, c_p, c_d, c_i = ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double, ctypes.c_int
        L.orc_cast_to_f32.argtypes = [c_p, c_i, c_p, c_i64]
        L.orc_cast_from_f32.argtypes = [c_p, c_p, c_i, c_i64]
        L.orc_all_finite.argtypes = [c_p, c_i64]; L.orc_all_finite.restype = c_i
        L.orc_lerp.argtypes = [c_p, c_p, c_f, c_p, c_i64]
        L.orc_n_average.argtypes = [c_p, c_i, c_p, c_i64]
        L.orc_addcmul.argtypes = [c_p, c_p, c_f, c_p, c_i64]
        L.orc_sub.argtypes = [c_p, c_p, c_p, c_i64]
        L.orc_scale.argtypes = [c_p, c_f, c_p, c_i64]
        L.orc_slerp.argtypes = [c_p, c_p, c_f, c_p, c_i64]; L.orc_slerp.restype = c_i
        L.orc_perpendicular.argtypes = [c_p, c_p, c_p, c_i64]; L.orc_perpendicular.restype = c_i
        L.orc_ties_kth.argtypes = [c_d, c_i64]; L.orc_ties_kth.restype = c_i64
        L.orc_topk_threshold.argtypes = [c_p, c_i64, c_d]; L.orc_topk_threshold.restype = c_f
        L.orc_ties_sum.argtypes = [c_p, c_i, c_i64, c_d, c_i, c_p, c_p, c_p, c_p, c_p]
        L.orc_dare_masks.argtypes = [ctypes.c_uint64, c_f, c_i, c_i64, c_p]
        L.orc_apply_mask.argtypes = [c_p, c_p, c_p, c_i64]
        L.orc_div_scalar.argtypes = [c_p, c_f, c_p, c_i64]
        L.orc_lora_delta.argtypes = [c_p, c_p, c_i64, c_i64, c_i64, c_f, c_i, c_p]
        L.orc_lerp_tensor.argtypes = [c_p, c_p, c_p, c_p, c_i64]
        L.orc_addcmul_tensor.argtypes = [c_p, c_p, c_p, c_p, c_i64]
        _lib = L
    return _lib


def _p(a: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr_array(arrs: Sequence[np.ndarray]):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def isclose(a: float, b: float) -> bool:
    """math.isclose with the defaults the reference uses (rel_tol=1e-9, abs_tol=0)."""
    return math.isclose(float(a), float(b))


# ---------------------------------------------------------------- casts
def to_f32(a: np.ndarray, dtype) -> np.ndarray:
    """extensions/builtin/merge_methods/logistics.py:53 ``a.to(device, float32)`` (exact upcast)."""
    code = DTYPE_CODE[dtype] if isinstance(dtype, str) else dtype
    a = np.ascontiguousarray(a)
    out = np.empty(a.shape, dtype=np.float32)
    lib().orc_cast_to_f32(_p(a), code, _p(out), a.size)
    return out


def from_f32(a: np.ndarray, dtype) -> np.ndarray:
    """Root ``cast`` node (merging.py:157-158): one round-to-nearest-even to the output dtype."""
    code = DTYPE_CODE[dtype] if isinstance(dtype, str) else dtype
    a = _f32(a)
    out = np.empty(a.shape, dtype=np.float32 if code == F32 else np.uint16)
    lib().orc_cast_from_f32(_p(a), _p(out), code, a.size)
    return out


def all_finite(a: np.ndarray) -> bool:
    """logistics.py:139 / merging.py:299."""
    a = _f32(a)
    return bool(lib().orc_all_finite(_p(a), a.size))


# ------------------------------------------------------------ streaming merges
def lerp(a, b, w: float) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_lerp(_p(a), _p(b), float(np.float32(w)), _p(out), a.size)
    return out


def _alpha_tensor(alpha, like: np.ndarray) -> np.ndarray:
    """A full alpha tensor broadcast to the shape of `a` (torch broadcasting rules), fp32."""
    return np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), like.shape))


def weighted_sum(a, b, alpha=0.5) -> np.ndarray:
    """linear.py:7-23.  alpha: a scalar, or an array with more than one element (no 0 / 1 shortcut then, :16)."""
    if np.ndim(alpha) > 0 and np.size(alpha) > 1:
        a, b = _f32(a), _f32(b)
        w = _alpha_tensor(alpha, a)
        out = np.empty_like(a)
        lib().orc_lerp_tensor(_p(a), _p(b), _p(w), _p(out), a.size)
        return out
    alpha = float(np.float32(np.asarray(alpha).reshape(-1)[0] if np.ndim(alpha) > 0 else alpha))         # alpha arrives as a 0-d fp32 tensor (merging.py:632-633)
    if isclose(alpha, 0.0):
        return _f32(a)
    if isclose(alpha, 1.0):
        return _f32(b)
    return lerp(a, b, alpha)


def n_average(models: Sequence[np.ndarray]) -> np.ndarray:
    """linear.py:26-40."""
    assert len(models) > 0
    ms = [_f32(m) for m in models]
    out = np.empty_like(ms[0])
    lib().orc_n_average(_ptr_array(ms), len(ms), _p(out), out.size)
    return out


def add_difference(a, b, alpha=1.0) -> np.ndarray:
    """linear.py:64-76 (method body; the ``| a`` fallback of wrappers.py:9-39 is graph-level).  alpha: scalar or full tensor."""
    if np.ndim(alpha) > 0 and np.size(alpha) > 1:
        a, b = _f32(a), _f32(b)
        w = _alpha_tensor(alpha, a)
        out = np.empty_like(a)
        lib().orc_addcmul_tensor(_p(a), _p(b), _p(w), _p(out), a.size)
        return out
    alpha = float(np.float32(np.asarray(alpha).reshape(-1)[0] if np.ndim(alpha) > 0 else alpha))
    if isclose(alpha, 0.0):
        return _f32(a)
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_addcmul(_p(a), _p(b), alpha, _p(out), a.size)
    return out


def subtract(a, b) -> np.ndarray:
    """linear.py:96-101."""
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_sub(_p(a), _p(b), _p(out), a.size)
    return out


def scale(a, alpha: float = 1.0):
    """linear.py:79-93 (returns python 0.0 for alpha == 0, like the reference)."""
    alpha = float(np.float32(alpha))
    if isclose(alpha, 0.0):
        return 0.0
    if isclose(alpha, 1.0):
        return _f32(a)
    a = _f32(a)
    out = np.empty_like(a)
    lib().orc_scale(_p(a), alpha, _p(out), a.size)
    return out


def slerp(a, b, alpha: float = 0.5) -> np.ndarray:
    """linear.py:43-61."""
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_slerp(_p(a), _p(b), float(np.float32(alpha)), _p(out), a.size)
    return out


def perpendicular_component(a, b) -> np.ndarray:
    """linear.py:104-113."""
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_perpendicular(_p(a), _p(b), _p(out), a.size)
    return out


def add_perpendicular(a, b, c, alpha: float = 1.0) -> np.ndarray:
    """wrappers.py:42-64: a + alpha * perpendicular_component(a - c, b - c)."""
    a_diff = subtract(a, c)
    b_diff = subtract(b, c)
    perp = perpendicular_component(a_diff, b_diff)
    return add_difference(a, perp, alpha)


# ----------------------------------------------------------------- TIES
def ties_kth(k: float, numel: int) -> int:
    """ties.py:129."""
    return max(int((1 - k) * numel), 1)


def topk_threshold(a, k: float) -> np.float32:
    """ties.py:129-130: the kth-smallest |a| (1-indexed), as fp32."""
    a = _f32(a)
    return np.float32(lib().orc_topk_threshold(_p(a), a.size, float(k)))


def filter_top_k(a, k: float) -> np.ndarray:
    """ties.py:128-132."""
    a = _f32(a)
    thr = topk_threshold(a, k)
    return (a * (np.abs(a) >= thr).astype(np.float32)).astype(np.float32)


def ties_sum(deltas: Sequence[np.ndarray], k: float = 1.0, vote_sgn: bool = False, details: bool = False):
    """ties.py:98-125.  With ``details`` also returns thresholds, keep masks, elected sign, agree count."""
    if len(deltas) == 0:
        return 0
    ds = [_f32(d) for d in deltas]
    if ds[0].size == 0:
        return ds[0]
    n, N = ds[0].size, len(ds)
    assert N <= 64
    out = np.empty_like(ds[0])
    thr = np.empty(N, dtype=np.float32)
    if details:
        keep = np.empty((N,) + ds[0].shape, dtype=np.uint8)
        sign = np.empty(ds[0].shape, dtype=np.int8)
        count = np.empty(ds[0].shape, dtype=np.uint8)
        lib().orc_ties_sum(_ptr_array(ds), N, n, float(k), int(vote_sgn), _p(out), _p(thr), _p(keep), _p(sign), _p(count))
        return out, dict(thresholds=thr, keep=keep, sign=sign, count=count, kth=ties_kth(k, n))
    lib().orc_ties_sum(_ptr_array(ds), N, n, float(k), int(vote_sgn), _p(out), _p(thr), None, None, None)
    return out


def add_difference_ties(base, models: Sequence[np.ndarray], k: float = 0.2, alpha: float = 1.0,
                        vote_sgn: bool = False) -> np.ndarray:
    """Graph of wrappers.py:70-94 built from primitives (the wrapper itself raises upstream):
    add_difference(base, ties_sum(m_i - base ..., k), alpha)."""
    deltas = [subtract(m, base) for m in models]
    return add_difference(base, ties_sum(deltas, k=k, vote_sgn=vote_sgn), alpha)


# ----------------------------------------------------------------- DARE
TORCH_DEFAULT_SEED = 67280421310721   # a torch.Generator that was never seeded: what ties.py:29-31 uses for seed None / < 0


def dare_masks(seed: int, probability: float, n_models: int, shape) -> np.ndarray:
    """ties.py:29-33, :63-64 with della_eps == 0: torch CPU generator stream, one draw per element,
    deltas in order.  Returns uint8 [n_models, *shape] keep masks."""
    n = int(np.prod(shape, dtype=np.int64))
    masks = np.empty((n_models, n), dtype=np.uint8)
    if seed is None or seed < 0:
        seed = TORCH_DEFAULT_SEED
    lib().orc_dare_masks(int(seed) & 0xFFFFFFFFFFFFFFFF, float(np.float32(probability)), n_models, n, _p(masks))
    return masks.reshape((n_models,) + tuple(shape))


def model_stock_t(filtered: np.ndarray, cos_eps: float) -> np.ndarray:
    """ties.py:146-159 over the stacked filtered deltas [N, ...]: per-row (last dim) factor.
    torch.nn.CosineSimilarity(dim=-1, eps): x.y / (max(|x|, eps) * max(|y|, eps))  (ATen cosine_similarity)."""
    n = filtered.shape[0]
    f64 = filtered.astype(np.float64)
    cos_thetas = []
    for i in range(n - 1):
        x, y = f64[i], f64[i + 1]
        nx = np.maximum(np.sqrt((x * x).sum(-1)), cos_eps)
        ny = np.maximum(np.sqrt((y * y).sum(-1)), cos_eps)
        cos_thetas.append(((x / nx[..., None]) * (y / ny[..., None])).sum(-1))
    cos_theta = np.mean(np.stack(cos_thetas), axis=0).astype(np.float32)
    t = (np.float32(n) * cos_theta / (np.float32(1) + np.float32(n - 1) * cos_theta)).astype(np.float32)
    return t[..., None]


def ties_sum_extended(deltas: Sequence[np.ndarray], k: float = 0.2, vote_sgn: bool = False,
                      apply_stock: bool = False, cos_eps: float = 1e-6):
    """ties.py:68-94 with apply_median=False.  NOTE ties.py:90: ``t = 1.0 if apply_stock else
    get_model_stock_t(...)`` -- apply_stock=False APPLIES the factor (restated literally)."""
    if len(deltas) == 0:
        return 0
    ds = [_f32(d) for d in deltas]
    if ds[0].size == 0:
        return ds[0]
    _, det = ties_sum(ds, k=k, vote_sgn=vote_sgn, details=True)
    keep = det["keep"].astype(np.float32)
    stacked = np.stack(ds) * keep                                   # filter_top_k per delta
    agree = (np.sign(stacked) == det["sign"].astype(np.float32)[None]).astype(np.float32)
    filtered = (stacked * agree).astype(np.float32)
    counts = det["count"].astype(np.float32)
    acc = np.zeros_like(ds[0])
    for m in range(len(ds)):
        acc = (acc + filtered[m]).astype(np.float32)
    if apply_stock:
        with np.errstate(all="ignore"):
            res = (acc * np.float32(1.0) / counts)
    else:
        if ds[0].ndim == 0:
            raise ValueError("model stock factor needs at least 1 dim")
        t = model_stock_t(filtered, cos_eps)
        with np.errstate(all="ignore"):
            res = ((acc * t).astype(np.float32) / counts)
    with np.errstate(all="ignore"):
        res = np.asarray(res, dtype=np.float32)
    return np.nan_to_num(res, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)


def ties_sum_with_dropout(deltas: Sequence[np.ndarray], probability: float = 0.9, rescale: bool = True,
                          k: float = 0.2, vote_sgn: bool = False, apply_stock: bool = False,
                          cos_eps: float = 1e-6, seed=None, masks: np.ndarray | None = None):
    """ties.py:10-50 with della_eps == 0 and apply_median == False (the in-scope DARE path)."""
    probability = float(np.float32(probability))
    if len(deltas) == 0 or isclose(probability, 1.0):
        return 0
    ds = [_f32(d) for d in deltas]
    if masks is None:
        masks = dare_masks(seed, probability, len(ds), ds[0].shape)
    dropped = []
    for d, m in zip(ds, masks):
        out = np.empty_like(d)
        m = np.ascontiguousarray(m, dtype=np.uint8)
        lib().orc_apply_mask(_p(d), _p(m), _p(out), d.size)
        dropped.append(out)
    res = ties_sum_extended(dropped, k=k, vote_sgn=vote_sgn, apply_stock=apply_stock, cos_eps=cos_eps)
    if isinstance(res, int):
        return res
    rescalar = np.float32(1.0) if not rescale else np.float32(1.0) - np.float32(probability)
    out = np.empty_like(res)
    res = _f32(res)
    lib().orc_div_scalar(_p(res), float(rescalar), _p(out), res.size)
    return out


def model_stock(deltas: Sequence[np.ndarray], cos_eps: float = 1e-6) -> np.ndarray:
    """ties.py:137-143: nan_to_num(t * (sum(deltas) / len(deltas))); python's sum() adds in argument order from 0; nan_to_num with
    torch's defaults (nan -> 0, +-inf -> +-FLT_MAX)."""
    ds = [_f32(d) for d in deltas]
    acc = ds[0].copy()
    for d in ds[1:]:
        acc = (acc + d).astype(np.float32)
    w_avg = (acc / np.float32(len(ds))).astype(np.float32)
    t = model_stock_t(np.stack(ds), cos_eps)
    with np.errstate(all="ignore"):
        res = (t * w_avg).astype(np.float32)
    fmax = np.finfo(np.float32).max
    return np.nan_to_num(res, nan=0.0, posinf=fmax, neginf=-fmax).astype(np.float32)


def dropout(deltas: Sequence[np.ndarray], probability: float = 0.9, rescale: float = 1.0, seed=None, masks=None) -> np.ndarray:
    """ties.py:202-238 at overlap == 1 (independent masks from numpy's default_rng, :214-217)."""
    ds = [_f32(d) for d in deltas]
    if masks is None:
        rng = np.random.default_rng(seed)
        masks = [rng.binomial(n=1, p=1 - probability, size=ds[0].shape).astype(bool) for _ in ds]
    acc = np.zeros_like(ds[0])
    cnt = np.zeros(ds[0].shape, dtype=np.float32)
    for m, d in zip(masks, ds):
        acc = np.where(m, (acc + d).astype(np.float32), acc)
        cnt = cnt + m.astype(np.float32)
    if probability == 1.0:
        rescalar = 1.0
    else:
        rescalar = (1.0 - probability) ** rescale
        rescalar = rescalar if math.isfinite(rescalar) else 1
    res = (acc / np.maximum(cnt, np.float32(1))).astype(np.float32)
    out = np.empty_like(res)
    lib().orc_div_scalar(_p(np.ascontiguousarray(res)), float(np.float32(rescalar)), _p(out), res.size)
    return out


# ----------------------------------------------------------------- LoRA
def lora_delta(up, down, alpha=None, target_shape=None) -> np.ndarray:
    """extensions/builtin/lycoris.py:226-263: (up.view(out,-1) @ down.view(r,-1)) * (alpha / r or 1)."""
    up, down = _f32(up), _f32(down)
    rows, r = up.shape[0], down.shape[0]
    up2, down2 = up.reshape(rows, -1), down.reshape(r, -1)
    assert up2.shape[1] == r
    cols = down2.shape[1]
    out = np.empty((rows, cols), dtype=np.float32)
    if alpha is None:
        lib().orc_lora_delta(_p(up2), _p(down2), rows, r, cols, 1.0, 0, _p(out))
    else:
        s = np.float32(alpha) / np.float32(r)                       # lycoris.py:263  alpha / down.size(0)
        lib().orc_lora_delta(_p(up2), _p(down2), rows, r, cols, float(s), 1, _p(out))
    return out.reshape(target_shape) if target_shape is not None else out


# ----------------------------------------------------------------- rotate
def rotate_shape_2d(shape) -> tuple:
    """svd.py:49-53."""
    shape = tuple(shape)
    is_conv = len(shape) == 4 and int(np.prod(shape[-2:])) != 1
    if is_conv:
        return int(np.prod(shape[:2])), int(np.prod(shape[2:]))
    return int(shape[0]), int(np.prod(shape[1:]))


def orthogonal_procrustes(a2: np.ndarray, b2: np.ndarray):
    """svd.py:157-169 (cancel_reflection=False).  Returns ("full", R) or ("low", u, v).
    numpy/LAPACK in float64: the parity target is the rotation to 1e-3 relative Frobenius, not bits."""
    n, p = a2.shape
    a2, b2 = a2.astype(np.float64), b2.astype(np.float64)
    if n < p:
        b_q, b_r = np.linalg.qr(b2.T)                               # b.mH = Q R, Q: p x n
        u, _, vh = np.linalg.svd(a2.T @ b_r.T, full_matrices=False)
        return "low", u, (vh @ b_q.T).T                             # LowRankOrthogonalMatmul(u, vh @ b_q.mH); v = vh_arg.mH
    u, _, vh = np.linalg.svd(a2.T @ b2, full_matrices=False)
    return "full", u @ vh


def truncate_rank(a, rank: int = 8):
    """svd.py:89-143 with use_approximate_basis=False: u[:, :r] * s[:r] @ vh[:r] of the SVD of a.flatten(start_dim=1) (fp64 here;
    the reference runs LAPACK in the input's precision)."""
    a = _f32(a)
    if a.ndim < 2:                                              # svd.py:99-100
        return a
    a2 = a.reshape(a.shape[0], -1).astype(np.float64)           # svd.py:110
    max_rank = min(a2.shape)
    r = min(max(round(rank), 0), max_rank)                      # svd.py:111-112
    if r == max_rank:
        return a
    if r == 0:
        return np.zeros_like(a)
    u, s, vh = np.linalg.svd(a2, full_matrices=False)
    return ((u[:, :r] * s[:r]) @ vh[:r]).reshape(a.shape).astype(np.float32)   # svd.py:143


def rotate(a, b, alignment: float = 1.0, alpha: float = 0.0, centralization: float = 1.0, return_transform: bool = False):
    """svd.py:10-86 at alignment in {-1, 0, 1} -- the in-scope configurations."""
    a, b = _f32(a), _f32(b)
    alpha = float(np.float32(alpha))
    if isclose(alignment, 0.0) and isclose(alpha, 0.0):
        return a
    if isclose(alignment, 1.0) and isclose(alpha, 1.0):
        return b
    # torch.allclose(a.half(), b.half()): rtol 1e-5, atol 1e-8, evaluated in fp16
    if a.ndim <= 1 or np.allclose(a.astype(np.float16).astype(np.float32), b.astype(np.float16).astype(np.float32), rtol=1e-5, atol=1e-8):
        return lerp(a, b, alpha)
    if not (isclose(alignment, 1.0) or isclose(alignment, 0.0) or isclose(alignment, -1.0)):
        raise NotImplementedError("fractional alignment is out of scope (SURVEY §8 a16)")
    shape_2d = rotate_shape_2d(a.shape)
    a2 = a.reshape(shape_2d).astype(np.float64)
    b2 = b.reshape(shape_2d).astype(np.float64)
    a_mean = a2.mean(0) * centralization
    b_mean = b2.mean(0) * centralization
    a2 = a2 - a_mean
    b2 = b2 - b_mean
    tr = orthogonal_procrustes(a2, b2)

    def apply(x, t):
        if isclose(t, 0.0):
            return x
        if tr[0] == "full":
            return x @ (tr[1] if t > 0 else tr[1].T)
        u, v = tr[1], tr[2]
        if t > 0:
            return x - x @ v @ v.T + x @ u @ v.T
        return x - x @ u @ u.T + x @ v @ u.T

    if not isclose(alpha, 0.0):
        a2 = a2 + alpha * (apply(b2, -1.0) - a2)
    a2 = apply(a2, alignment)
    a2 = a2 + (a_mean + alignment * (b_mean - a_mean))
    out = a2.reshape(a.shape).astype(np.float32)
    if return_transform:
        return out, tr
    return out


def set_threads(n: int) -> None:
    """Number of OpenMP threads the C oracle uses (the cpu_baseline reports this as ``cores``)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass
