"""CPU: the C-ABI library loads and exports every symbol include/mecha_hip.h declares (no compute calls)."""
import ctypes
import pathlib
import re

ROOT = pathlib.Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "mecha_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mecha_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sd_mecha_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 20
    assert _lib.LIB_PATH.exists(), "build with `make -C sd_mecha_amd/csrc` or __graft_entry__.build()"
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in mecha_hip.h but not exported: {missing}"
    assert sorted(_lib.SIGNATURES) == syms, "the ctypes binding must cover exactly the declared symbols"


def test_abi_version_and_error_string():
    from sd_mecha_amd import _lib
    L = _lib.lib()
    assert L.mecha_abi_version() == _lib.ABI_VERSION
    assert isinstance(L.mecha_last_error(), bytes)


def test_host_side_argument_validation_needs_no_gpu():
    """Entry points validate on the host before touching the device: bad arguments return an error code."""
    from sd_mecha_amd import _lib
    L = _lib.lib()
    assert L.mecha_weighted_sum(None, None, None, -1, 2, 2, 0.5, None, None) < 0
    assert b"numel" in L.mecha_last_error() or b"NULL" in L.mecha_last_error()
    assert L.mecha_n_average(None, 0, None, 8, 2, 2, None, None) < 0
    assert L.mecha_ties_sum(None, 0, None, 8, 2, 2, 0.2, 0, None, None, 0, None, None) < 0
    assert L.mecha_cast(None, None, 0, 7, 2, None, None) < 0          # unknown dtype code


def test_dare_host_masks_match_reference_stream():
    """mecha_dare_draw_masks_host is host code: check it against the mask torch's CPU generator draws
    (seed 114514, p = 0.4, 16 elements) and against the oracle on a longer stream."""
    import numpy as np
    from sd_mecha_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    n_models, numel = 3, 1000
    words = (numel + 31) // 32
    bits = np.zeros((n_models, words), dtype=np.uint32)
    assert L.mecha_dare_draw_masks_host(114514, 0.4, n_models, numel, bits.ctypes.data) == 0
    unpacked = ((bits[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n_models, -1)[:, :numel].astype(np.uint8)
    want = O.dare_masks(114514, 0.4, n_models, (numel,))
    assert np.array_equal(unpacked, want)
    first16 = np.zeros((1, 1), dtype=np.uint32)
    assert L.mecha_dare_draw_masks_host(114514, 0.4, 1, 16, first16.ctypes.data) == 0
    assert [(int(first16[0, 0]) >> i) & 1 for i in range(16)] == [1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0, 1, 1]
