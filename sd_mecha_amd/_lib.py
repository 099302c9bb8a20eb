"""ctypes binding of libmecha_hip.so (the C ABI declared in include/mecha_hip.h).

The product path has no CPU fallback: if the library is missing this module raises at first use.
"""
from __future__ import annotations

import ctypes
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmecha_hip.so"

F32, F16, BF16 = 0, 1, 2
FLAG_INPUT_NONFINITE = 1
FLAG_OUTPUT_NONFINITE = 2
MAX_MODELS = 32
LORA_TILE = 128          # MECHA_LORA_TILE
ABI_VERSION = 1


class Segment(ctypes.Structure):
    """mecha_segment: a key's run of elements inside the per-model arenas."""
    _fields_ = [("offset", ctypes.c_int64), ("numel", ctypes.c_int64)]


class LoraKey(ctypes.Structure):
    """mecha_lora_key"""
    _fields_ = [("offset", ctypes.c_int64), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("tile_start", ctypes.c_int64),
                ("net_begin", ctypes.c_int32), ("net_count", ctypes.c_int32)]


class LoraNet(ctypes.Structure):
    """mecha_lora_net"""
    _fields_ = [("up", ctypes.c_void_p), ("down", ctypes.c_void_p), ("rank", ctypes.c_int32), ("scale", ctypes.c_float),
                ("weight", ctypes.c_float), ("reserved", ctypes.c_int32)]


class MechaError(RuntimeError):
    """A libmecha_hip entry point returned a negative status."""


_lib = None

c_p, c_i, c_i64, c_f, c_d, c_sz, c_u64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                          ctypes.c_double, ctypes.c_size_t, ctypes.c_uint64)

# name -> (restype, argtypes); every symbol include/mecha_hip.h declares
SIGNATURES = {
    "mecha_abi_version": (c_i, []),
    "mecha_last_error": (ctypes.c_char_p, []),
    "mecha_profile_enable": (c_i, [c_i]),
    "mecha_profile_kernel_count": (c_i, []),
    "mecha_profile_kernel_name": (ctypes.c_char_p, [c_i]),
    "mecha_profile_read": (c_i, [c_i, c_p, c_p, c_i]),
    "mecha_cast": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "mecha_weighted_sum": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_n_average": (c_i, [c_p, c_i, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "mecha_add_difference": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_weighted_sum_tensor": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "mecha_add_difference_tensor": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "mecha_subtract": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p]),
    "mecha_scale": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_reduce_workspace_bytes": (c_sz, [c_i64]),
    "mecha_slerp": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_sz, c_p, c_p]),
    "mecha_perpendicular_component": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_sz, c_p, c_p]),
    "mecha_add_perpendicular": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_sz, c_p, c_p]),
    "mecha_ties_workspace_bytes": (c_sz, [c_i, c_i64]),
    "mecha_host_pread_many": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i]),
    "mecha_host_pwrite_many": (c_i, [c_i, c_p, c_p, c_p, c_i64, c_i]),
    "mecha_host_h2d_many": (c_i, [c_p, c_p, c_p, c_i64, c_p]),
    "mecha_host_parse_safetensors_header": (c_i64, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mecha_host_mmap_write_many": (c_i, [c_i, c_p, c_p, c_p, c_i64, c_i]),
    "mecha_ties_select_stats": (c_i, [c_p, c_i, c_i, c_i64, c_p, c_p, c_p]),
    "mecha_ties_sum": (c_i, [c_p, c_i, c_p, c_i64, c_i, c_i, c_d, c_i, c_p, c_p, c_sz, c_p, c_p]),
    "mecha_add_difference_ties": (c_i, [c_p, c_p, c_i, c_p, c_i64, c_i, c_i, c_d, c_f, c_i, c_p, c_p, c_sz, c_p, c_p]),
    "mecha_add_difference_ties_noalloc": (c_i, [c_p, c_p, c_i, c_p, c_i64, c_i, c_i, c_d, c_f, c_i, c_p, c_p, c_p]),
    "mecha_ties_sum_with_dropout": (c_i, [c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_d, c_i, c_f, c_i, c_p, c_p, c_sz, c_p, c_p]),
    "mecha_ties_extended_workspace_bytes": (c_sz, [c_i, c_i64, c_i64]),
    "mecha_ties_sum_extended": (c_i, [c_p, c_i, c_p, c_p, c_i64, c_i64, c_i, c_i, c_d, c_i, c_i, c_f, c_f, c_i, c_p, c_p, c_sz, c_p, c_p]),
    "mecha_model_stock_workspace_bytes": (c_sz, [c_i, c_i64, c_i64]),
    "mecha_model_stock": (c_i, [c_p, c_i, c_p, c_i64, c_i64, c_i, c_i, c_f, c_p, c_sz, c_p, c_p]),
    "mecha_dropout": (c_i, [c_p, c_i, c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_dare_draw_masks_host": (c_i, [c_u64, c_f, c_i, c_i64, c_p]),
    "mecha_dare_draw_stream_host": (c_i, [c_u64, c_f, c_i64, c_p]),
    "mecha_add_difference_sub": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_batched_add_difference_sub": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_batched_reduce_workspace_bytes": (c_sz, [c_i, c_i64]),
    "mecha_batched_slerp": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_f, c_p, c_sz, c_p, c_p]),
    "mecha_batched_perpendicular_component": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_p, c_sz, c_p, c_p]),
    "mecha_batched_add_perpendicular": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_f, c_p, c_sz, c_p, c_p]),
    "mecha_batched_dare_workspace_bytes": (c_sz, [c_i, c_i, c_i64, c_i64]),
    "mecha_batched_add_difference_dare": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_d, c_f, c_i, c_f, c_i, c_p, c_i64,
                                                c_i, c_f, c_p, c_p, c_p, c_p, c_sz, c_p, c_p]),
    "mecha_lora_accumulate": (c_i, [c_p, c_p, c_i64, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "mecha_batched_lora_accumulate": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "mecha_rotate_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "mecha_rotate": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_f, c_p, c_sz, c_p, c_p, c_p]),
    "mecha_rotate_transform_floats": (c_i64, [c_i64, c_i64]),
    "mecha_truncate_rank_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "mecha_truncate_rank": (c_i, [c_p, c_p, c_i64, c_i64, c_i64, c_i, c_i, c_p, c_sz, c_p, c_p, c_p]),
    "mecha_rotate_ex": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_f, c_i, c_f, c_p, c_p, c_p, c_sz, c_p, c_p, c_p]),
    "mecha_allclose_half": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p]),
    "mecha_batched_rotate_workspace_bytes": (c_sz, [c_i64, c_i64, c_i]),
    "mecha_batched_rotate": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i64, c_i64, c_i, c_i, c_f, c_i, c_f, c_p, c_sz, c_p, c_p]),
    "mecha_batched_weighted_sum": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_batched_n_average": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i64, c_i, c_i, c_p, c_p]),
    "mecha_batched_add_difference": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_f, c_p, c_p]),
    "mecha_batched_ties_workspace_bytes": (c_sz, [c_i, c_i, c_i64]),
    "mecha_batched_add_difference_ties": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i64, c_i, c_i, c_d, c_f, c_i,
                                                c_p, c_p, c_sz, c_p, c_p]),
}


def lib() -> ctypes.CDLL:
    """Load libmecha_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise MechaError(
                f"{LIB_PATH} is missing: build it with `make -C sd_mecha_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback for the MI355X merge path."
            )
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header / library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        got = L.mecha_abi_version()
        if got != ABI_VERSION:
            raise MechaError(f"libmecha_hip ABI version {got}, binding expects {ABI_VERSION}")
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mecha_last_error().decode("utf-8", "replace")
        raise MechaError(f"{what} failed ({status}): {msg}")


def ptr_array(ptrs) -> ctypes.Array:
    """HOST array of device pointers (`const void *const *`)."""
    return (ctypes.c_void_p * len(ptrs))(*[int(p) for p in ptrs])


def profile_enable(on: bool) -> None:
    check(lib().mecha_profile_enable(int(on)), "mecha_profile_enable")


def profile_read(reset: bool = True) -> dict:
    """{kernel family: (total_ms, launches)} measured with HIP events on the launch streams."""
    L = lib()
    out = {}
    for i in range(L.mecha_profile_kernel_count()):
        ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
        check(L.mecha_profile_read(i, ctypes.byref(ms), ctypes.byref(n), int(reset)), "mecha_profile_read")
        if n.value:
            out[L.mecha_profile_kernel_name(i).decode()] = (ms.value, n.value)
    return out
