"""GPU parity of rotate (sd_mecha_amd/csrc/rotate.hip: f32-MFMA contractions + one-sided Jacobi SVD) against the
reference's svd.rotate outputs (tests/golden/rotate.npz) and against the oracle (numpy/LAPACK, float64) at larger sizes.
Bar (BASELINE.json north_star): <= 1e-3 relative Frobenius error.  Wide keys with centring are compared on the
component the reference determines (test_oracle_golden.rotate_rel_error, DESIGN.md "rotate: reference indeterminacy")."""
import pathlib
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from conftest import load_golden  # noqa: E402
from test_oracle_golden import f32_from_bits, rotate_rel_error  # noqa: E402

pytestmark = pytest.mark.gpu


def test_rotate_golden(oracle):
    import sd_mecha_amd as amd
    g = load_golden("rotate")
    for c in range(int(g["n_cases"])):
        pre = f"r{c}"
        a, b = f32_from_bits(g[pre + "_a"]), f32_from_bits(g[pre + "_b"])
        ta, tb = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()
        for cz in (1.0, 0.0):
            want = f32_from_bits(g[f"{pre}_out_alpha0.0_cz{cz}"]).reshape(a.shape)
            got = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=cz).cpu().numpy()
            rel = rotate_rel_error(oracle, a, b, got, want, cz)
            assert rel <= 1e-3, (pre, cz, rel)
        if a.ndim >= 2:
            with pytest.raises(NotImplementedError):
                amd.rotate(ta, tb, alignment=1.0, alpha=0.5)
    # early-outs (svd.py:37-47)
    x, y = torch.randn(8, 8, device="cuda"), torch.randn(8, 8, device="cuda")
    assert torch.equal(amd.rotate(x, y, alignment=0.0, alpha=0.0), x)
    assert torch.equal(amd.rotate(x, y, alignment=1.0, alpha=1.0), y)
    assert torch.equal(amd.rotate(x, x.clone(), alignment=1.0, alpha=0.0), x)              # allclose(a.half(), b.half()) -> lerp
    v, w = torch.randn(12, device="cuda"), torch.randn(12, device="cuda")
    assert torch.equal(amd.rotate(v, w, alignment=1.0, alpha=0.25), torch.lerp(v, w, 0.25))  # ndim <= 1 -> lerp


@pytest.mark.parametrize("shape,cz", [((1280, 320), 1.0), ((256, 256), 1.0), ((640, 640), 0.0), ((96, 512), 0.0), ((96, 512), 1.0),
                                      ((64, 32, 3, 3), 1.0), ((320, 4, 3, 3), 1.0), ((2048, 77), 1.0)])
def test_rotate_vs_oracle(oracle, shape, cz):
    import sd_mecha_amd as amd
    rng = np.random.default_rng(360 + len(shape))
    a = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=shape).astype(np.float32), "bfloat16"), "bfloat16")
    b = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=shape).astype(np.float32), "bfloat16"), "bfloat16")
    want = oracle.rotate(a, b, alignment=1.0, alpha=0.0, centralization=cz)
    ta, tb = torch.from_numpy(a).cuda().to(torch.bfloat16), torch.from_numpy(b).cuda().to(torch.bfloat16)
    got, sweeps = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=cz, return_sweeps=True)
    rel = rotate_rel_error(oracle, a, b, got.cpu().numpy(), want, cz)
    assert rel <= 1e-3, (shape, cz, rel, sweeps)
    assert 1 <= sweeps <= 200          # Newton-Schulz iterations (two polar solves for wide keys)
    # the transform is orthogonal on the rotated subspace: norms of centred rows are preserved (tall / square keys)
    n, p = oracle.rotate_shape_2d(shape)
    if n >= p and n > p + 1:
        a2 = a.reshape(n, p).astype(np.float64)
        g2 = got.cpu().numpy().reshape(n, p).astype(np.float64)
        b2 = b.reshape(n, p).astype(np.float64)
        ac = a2 - a2.mean(0) * cz
        gc = g2 - b2.mean(0) * cz
        assert abs(np.linalg.norm(gc) / np.linalg.norm(ac) - 1.0) < 1e-4
    out_bf16 = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=cz, out_dtype=torch.bfloat16)
    assert out_bf16.dtype == torch.bfloat16 and out_bf16.shape == ta.shape
