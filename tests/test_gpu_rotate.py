"""GPU parity of rotate (sd_mecha_amd/csrc/rotate.hip: f32-MFMA contractions + polar factor by a fixed schedule of optimal odd-quintic steps) against the
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
from test_oracle_golden import ROTATE_EXTRA_CASES, f32_from_bits, rotate_case_undetermined, rotate_rel_error, truncate_rank_cases  # noqa: E402

pytestmark = pytest.mark.gpu


def test_rotate_golden(oracle):
    import sd_mecha_amd as amd
    g = load_golden("rotate")
    for c in range(int(g["n_cases"])):
        pre = f"r{c}"
        a, b = f32_from_bits(g[pre + "_a"]), f32_from_bits(g[pre + "_b"])
        ta, tb = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()
        for alpha in (0.0, 0.5):
            for cz in (1.0, 0.0):
                want = f32_from_bits(g[f"{pre}_out_alpha{alpha}_cz{cz}"]).reshape(a.shape)
                got = amd.rotate(ta, tb, alignment=1.0, alpha=alpha, centralization=cz).cpu().numpy()
                rel = rotate_rel_error(oracle, a, b, got, want, cz, 1.0, alpha)
                assert rel <= 1e-3, (pre, alpha, cz, rel)
        for alignment, alpha, cz in ROTATE_EXTRA_CASES:         # alignment 0 and -1 (svd.py:84-85, :177-183, :209-215)
            if rotate_case_undetermined(oracle, a.shape, alignment, alpha, cz):
                continue
            want = f32_from_bits(g[f"{pre}_out_align{alignment}_alpha{alpha}_cz{cz}"]).reshape(a.shape)
            got = amd.rotate(ta, tb, alignment=alignment, alpha=alpha, centralization=cz).cpu().numpy()
            rel = rotate_rel_error(oracle, a, b, got, want, cz, alignment, alpha)
            assert rel <= 1e-3, (pre, alignment, alpha, cz, rel)
        # the factor cache (svd.py:55-79): the second call reads the transform the first one saved (fp16, on the host)
        cache = amd.rotate.create_cache()
        first = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=1.0, key="k", cache=cache)
        if a.ndim >= 2:
            assert cache["k"]["transform"].dtype == torch.float16 and cache["k"]["transform"].device.type == "cpu"
            assert cache["k"]["centralization"] == 1.0
            second, iters = amd.rotate(ta, tb, alignment=1.0, alpha=0.5, centralization=1.0, key="k", cache=cache, return_sweeps=True)
            assert iters == 0                                   # no Procrustes solve on a cache hit
            want = f32_from_bits(g[f"{pre}_out_cached_alpha0.5"]).reshape(a.shape)
            rel = rotate_rel_error(oracle, a, b, second.cpu().numpy(), want, 1.0, 1.0, 0.5)
            assert rel <= 1e-3, (pre, "cached", rel)
            again = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=1.0, key="k", cache=cache)
            rel = rotate_rel_error(oracle, a, b, again.cpu().numpy(), first.cpu().numpy(), 1.0)
            assert rel <= 1e-3
            amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=0.5, key="k", cache=cache)    # svd.py:59-61: invalidated
            assert "centralization" not in cache["k"] and "transform" in cache["k"]
            with pytest.raises(NotImplementedError):
                amd.rotate(ta, tb, alignment=0.5, alpha=0.0)
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
    print(f"rotate {shape} centralization {cz}: relative error {rel:.3e}")
    assert rel <= 1e-3, (shape, cz, rel, sweeps)
    assert sweeps in (14, 40)          # polar steps enqueued: 14 for the solve of a tall key, 2 x (6 + 14) for a wide key's two
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
    for alignment, alpha in ((1.0, 0.3), (0.0, 0.7), (-1.0, 0.0), (-1.0, 0.4)):
        if rotate_case_undetermined(oracle, shape, alignment, alpha, cz):
            continue
        want = oracle.rotate(a, b, alignment=alignment, alpha=alpha, centralization=cz)
        got = amd.rotate(ta, tb, alignment=alignment, alpha=alpha, centralization=cz)
        rel = rotate_rel_error(oracle, a, b, got.cpu().numpy(), want, cz, alignment, alpha)
        assert rel <= 1e-3, (shape, cz, alignment, alpha, rel)


# SDXL's real shapes: the centred square 1280 x 1280 (one zero singular value), the wide 1280 x 5120 and
# 640 x 2048, 3 x 3 convolutions as (out * in) x 9 with 102 400 and 1.6 M rows, and the 49408 x 1280 embedding.
@pytest.mark.parametrize("shape,cz", [((1280, 1280), 1.0), ((1280, 5120), 1.0), ((640, 2048), 1.0), ((640, 2048), 0.0),
                                      ((320, 320, 3, 3), 1.0), ((1280, 1280, 3, 3), 1.0), ((49408, 1280), 1.0)])
def test_rotate_sdxl_shapes_vs_oracle(oracle, shape, cz):
    import sd_mecha_amd as amd
    rng = np.random.default_rng(1280 + len(shape) + shape[1])
    a = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=shape).astype(np.float32), "bfloat16"), "bfloat16")
    b = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=shape).astype(np.float32), "bfloat16"), "bfloat16")
    want = oracle.rotate(a, b, alignment=1.0, alpha=0.0, centralization=cz)
    ta, tb = torch.from_numpy(a).cuda().to(torch.bfloat16), torch.from_numpy(b).cuda().to(torch.bfloat16)
    got = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=cz).cpu().numpy()
    rel = rotate_rel_error(oracle, a, b, got, want, cz)
    print(f"rotate {shape} centralization {cz}: relative error {rel:.3e}")
    assert rel <= 1e-3, (shape, cz, rel)
    n, p = oracle.rotate_shape_2d(shape)
    a2, g2, b2 = (x.reshape(n, p).astype(np.float64) for x in (a, got, b))
    ac, gc = a2 - a2.mean(0) * cz, g2 - b2.mean(0) * cz
    if n > p + 1:
        # tall: the centred output is the centred input times an orthogonal matrix -- Gram matrices agree
        assert abs(np.linalg.norm(gc) / np.linalg.norm(ac) - 1.0) < 1e-4
        rows = rng.choice(n, size=min(n, 512), replace=False)
        ga, gg = ac[rows] @ ac[rows].T, gc[rows] @ gc[rows].T
        assert np.linalg.norm(ga - gg) / np.linalg.norm(ga) < 1e-3
    else:
        # wide / square: the pass-through part.  Outside the row spaces of the centred a and b nothing may change except
        # along the reference's one free direction: (got - a) restricted to that complement has rank <= 1 ... and the HIP
        # path returns the minimum-norm member, for which it is zero
        _, sv, vh = np.linalg.svd(np.vstack([ac, b2 - b2.mean(0) * cz]), full_matrices=False)
        basis = vh[sv > sv[0] * 1e-6]
        moved = gc - ac
        outside = moved - (moved @ basis.T) @ basis
        assert np.linalg.norm(outside) / np.linalg.norm(ac) < 1e-3, (shape, np.linalg.norm(outside) / np.linalg.norm(ac))


def test_rotate_recipe_with_cache(oracle, tmp_path):
    """merge(rotate(a, b), cache={node: create_cache()}) (merging.py:199, svd.py:55-79): the second merge of the same node
    reuses the transforms of the first (a hyper-parameter sweep over alpha)."""
    import safetensors.torch
    import sd_mecha_amd as amd
    rng = np.random.default_rng(77)
    shapes = {"w.tall": (96, 40), "w.wide": (24, 80), "w.conv": (16, 8, 3, 3), "w.bias": (40,)}
    sds = []
    for i in range(2):
        sd = {k: torch.from_numpy(rng.normal(0, 0.02, size=s).astype(np.float32)).to(torch.bfloat16) for k, s in shapes.items()}
        safetensors.torch.save_file(sd, str(tmp_path / f"m{i}.safetensors"))
        sds.append({k: v.float().numpy() for k, v in sd.items()})
    caches = {}
    outs = []
    for alpha in (0.0, 0.5):
        node = amd.rotate(amd.model(tmp_path / "m0.safetensors"), amd.model(tmp_path / "m1.safetensors"), alignment=1.0, alpha=alpha)
        if not caches:
            cache = node.merge_method.create_cache()
        caches = {node: cache}
        res = amd.merge(node, cache=caches, output_dtype=torch.float32, merge_device="cuda")
        outs.append(res)
        for k in shapes:
            want = oracle.rotate(sds[0][k], sds[1][k], alignment=1.0, alpha=alpha, centralization=1.0)
            rel = rotate_rel_error(oracle, sds[0][k], sds[1][k], res[k].cpu().numpy(), want, 1.0, 1.0, alpha)
            assert rel <= 1e-3, (k, alpha, rel)
    assert set(cache) == {"w.tall", "w.wide", "w.conv"}          # 1-D keys take the lerp early-out before the cache (svd.py:46-47)
    assert all(v["transform"].dtype == torch.float16 for v in cache.values())


def _write_pair(tmp_path, tensors_a, tensors_b):
    import safetensors.torch
    for name, sd in (("a", tensors_a), ("b", tensors_b)):
        safetensors.torch.save_file({k: torch.from_numpy(v).to(torch.bfloat16) for k, v in sd.items()}, str(tmp_path / f"{name}.safetensors"))
    return tmp_path / "a.safetensors", tmp_path / "b.safetensors"


@pytest.mark.parametrize("alignment,alpha,cz", [(1.0, 0.0, 1.0), (1.0, 0.3, 1.0), (-1.0, 0.0, 0.0), (0.0, 0.6, 1.0)])
def test_batched_rotate_through_merge(oracle, tmp_path, alignment, alpha, cz):
    """merge(rotate(a, b)) takes the batched plan: keys grouped by 2-D shape, one launch sequence per group
    (mecha_batched_rotate), the svd.py:46-47 early-outs evaluated on the device.  Every key against the oracle."""
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    rng = np.random.default_rng(2024)
    shapes = {"t0": (320, 96), "t1": (320, 96), "t2": (320, 96), "sq0": (256, 256), "sq1": (256, 256), "w0": (96, 512), "w1": (96, 512),
              "w2": (96, 512), "c0": (32, 16, 3, 3), "c1": (32, 16, 3, 3), "bias": (320,), "same": (64, 48), "big": (640, 384)}
    bf = lambda x: oracle.to_f32(oracle.from_f32(x.astype(np.float32), "bfloat16"), "bfloat16")      # noqa: E731
    ta = {k: bf(rng.normal(0, 0.02, size=s)) for k, s in shapes.items()}
    tb = {k: bf(rng.normal(0, 0.02, size=s)) for k, s in shapes.items()}
    tb["same"] = ta["same"].copy()                                  # allclose(a.half(), b.half()) -> lerp (svd.py:46-47)
    pa, pb = _write_pair(tmp_path, ta, tb)
    node = amd.rotate(amd.model(pa), amd.model(pb), alignment=alignment, alpha=alpha, centralization=cz)
    res = amd.merge(node, output_dtype=torch.float32, merge_device="cuda")
    assert merging.last_stats["fused_keys"] == len(shapes) and merging.last_stats["generic_keys"] == 0
    for k, shape in shapes.items():
        want = oracle.rotate(ta[k], tb[k], alignment=alignment, alpha=alpha, centralization=cz)
        got = res[k].cpu().numpy()
        if len(shape) <= 1 or k == "same":
            assert np.array_equal(got, np.asarray(want, dtype=np.float32).reshape(shape)), k
            continue
        if rotate_case_undetermined(oracle, shape, alignment, alpha, cz):
            continue
        rel = rotate_rel_error(oracle, ta[k], tb[k], got, want, cz, alignment, alpha)
        assert rel <= 1e-3, (k, alignment, alpha, cz, rel)
    # the batched and the per-key entry point agree (same kernels; the wide keys' early exit is the only difference)
    per_key = amd.merge(node, output_dtype=torch.float32, merge_device="cuda", cache={node: node.merge_method.create_cache()})
    assert merging.last_stats["fused_keys"] == 0
    for k in shapes:
        d = np.linalg.norm(res[k].cpu().numpy().astype(np.float64) - per_key[k].cpu().numpy()) / max(np.linalg.norm(per_key[k].cpu().numpy()), 1e-30)
        assert d <= 2e-3, (k, d)


@pytest.mark.parametrize("polar_mode", ["", "both", "long"])
def test_batched_rotate_ill_conditioned_wide_keys(oracle, tmp_path, monkeypatch, polar_mode):
    """Wide keys whose b has singular values over three decades fail the optimistic 6-step polar schedule's convergence
    test and go on through the 14-step one, in the same batch as well-conditioned keys that leave after 6 steps."""
    import sd_mecha_amd as amd
    if polar_mode:                                                  # "both": every key falls through; "long": no short schedule
        monkeypatch.setenv("MECHA_ROT_POLAR", polar_mode)
    rng = np.random.default_rng(7)
    n, p = 64, 256
    ta, tb = {}, {}
    for i in range(4):
        u, _ = np.linalg.qr(rng.normal(size=(n, n)))
        v, _ = np.linalg.qr(rng.normal(size=(p, n)))
        sv = np.logspace(0, -3 if i % 2 else -0.3, n)              # odd keys: condition number 1e3
        ta[f"k{i}"] = rng.normal(0, 0.02, size=(n, p)).astype(np.float32)
        tb[f"k{i}"] = ((u * sv) @ v.T * 0.5).astype(np.float32)
    bf = lambda x: oracle.to_f32(oracle.from_f32(x, "bfloat16"), "bfloat16")      # noqa: E731
    ta = {k: bf(v) for k, v in ta.items()}
    tb = {k: bf(v) for k, v in tb.items()}
    pa, pb = _write_pair(tmp_path, ta, tb)
    res = amd.merge(amd.rotate(amd.model(pa), amd.model(pb), alignment=1.0, alpha=0.0, centralization=0.0), output_dtype=torch.float32,
                    merge_device="cuda")
    for k in ta:
        want = oracle.rotate(ta[k], tb[k], alignment=1.0, alpha=0.0, centralization=0.0)
        rel = rotate_rel_error(oracle, ta[k], tb[k], res[k].cpu().numpy(), want, 0.0)
        assert rel <= 1e-3, (k, rel)


@pytest.mark.parametrize("n,p", [(64, 256), (256, 1024)])
def test_rotate_borderline_conditioned_wide_keys(oracle, monkeypatch, n, p):
    """Fully centred wide keys whose b has a few singular values at 1e-3 .. 5e-3 of the largest -- just past what the optimistic
    6-step schedule is made for (POLAR_EASY_L), where its on-device convergence test decides: the default run (optimistic phase,
    deflated null direction) must agree with the 14-step-only run (MECHA_ROT_POLAR=long) and with the fp64 oracle, on both GEMM
    paths (64 rows: fp32 kernel; 256 rows: split-fp16 kernel)."""
    import sd_mecha_amd as amd
    rng = np.random.default_rng(n)
    for tail in ((1e-3, 2e-3, 5e-3), (3e-3,), (8e-3, 9e-3)):
        u, _ = np.linalg.qr(rng.normal(size=(n, n)))
        v, _ = np.linalg.qr(rng.normal(size=(p, n)))
        sv = np.linspace(1.0, 0.3, n)
        sv[-len(tail):] = tail
        a = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=(n, p)).astype(np.float32), "bfloat16"), "bfloat16")
        b = ((u * sv) @ v.T * 0.5).astype(np.float32)                      # fp32 input: the small singular values survive
        want = oracle.rotate(a, b, alignment=1.0, alpha=0.0, centralization=1.0)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        monkeypatch.delenv("MECHA_ROT_POLAR", raising=False)
        got = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=1.0).cpu().numpy()
        monkeypatch.setenv("MECHA_ROT_POLAR", "long")
        long = amd.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=1.0).cpu().numpy()
        monkeypatch.delenv("MECHA_ROT_POLAR")
        rel = rotate_rel_error(oracle, a, b, got, want, 1.0)
        rel_long = rotate_rel_error(oracle, a, b, long, want, 1.0)
        between = np.linalg.norm(got.astype(np.float64) - long) / np.linalg.norm(long)
        print(f"rotate {n} x {p}, tail {tail}: default {rel:.3e}, long schedule {rel_long:.3e}, default vs long {between:.3e}")
        assert rel <= 1e-3 and rel_long <= 1e-3, (tail, rel, rel_long)
        assert between <= 1e-3, (tail, between)


@pytest.mark.parametrize("shape", [(512, 256), (256, 512), (256, 256)])
@pytest.mark.parametrize("scale", [3e5, 1e-6])
def test_rotate_extreme_magnitudes(oracle, shape, scale):
    """Weights far outside fp16's range (|w| up to ~2.5e4 here; 65504 / 64 is where an unscaled split-fp16 image would overflow) and
    far below it: the split-kernel path scales each key's centred matrices by a power of two first (ROT_SCAL_T), so the result
    matches the oracle like any other key; inputs in fp32 so that the magnitudes are what they say."""
    import sd_mecha_amd as amd
    rng = np.random.default_rng(int(shape[0] + 10 * np.log10(scale)))
    a = (rng.normal(0, 0.02, size=shape) * scale).astype(np.float32)
    b = (rng.normal(0, 0.02, size=shape) * scale * 3).astype(np.float32)           # b on another scale than a
    want = oracle.rotate(a, b, alignment=1.0, alpha=0.0, centralization=1.0)
    got = amd.rotate(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), alignment=1.0, alpha=0.0, centralization=1.0).cpu().numpy()
    assert np.isfinite(got).all()
    rel = rotate_rel_error(oracle, a, b, got, want, 1.0)
    print(f"rotate {shape} x {scale:g}: relative error {rel:.3e}")
    assert rel <= 1e-3, (shape, scale, rel)


def test_full_config4_job_properties_and_sampled_keys(oracle):
    """BASELINE config 4 at FULL size: rotate(A, B, alignment=1, alpha=0) over every key of two SDXL-shaped bf16 checkpoints
    through merge() (files on tmpfs -> batched plan -> one fp32 file), checked where no CPU SVD of the whole job is affordable:
    * every strictly tall key (n > p: the full-rank path, svd.py:168-169) by what DEFINES its answer.  With centred
      A_c, B_c and O_c = out - mean(B): the rows of O_c have the norms of the rows of A_c (the transform is orthogonal), and
      O_c^T B_c is symmetric positive semidefinite (T = U V^T of A_c^T B_c = U S V^T gives T^T A_c^T B_c = V S V^T; among
      orthogonal T only the Procrustes solution makes that product symmetric PSD);
    * 1-d keys: the lerp at alpha = 0, bit for bit;
    * a spread of keys of every kind (square, wide, conv, tall) against the oracle with the reference's free terms split off."""
    import pathlib
    import shutil
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tools"))
    import bench_e2e
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    from sd_mecha_amd.shapes import load_manifest
    from sd_mecha_amd.streaming import InSafetensorsDict
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30 or shutil.disk_usage("/dev/shm").free < 40 << 30:
        pytest.skip("needs 40 GB of free HBM and of /dev/shm")
    manifest = [(k, s, d) for k, s, d in load_manifest("sdxl-sgm") if not d.startswith("int")]
    work = pathlib.Path("/dev/shm/mecha_test_config4")
    work.mkdir(parents=True, exist_ok=True)
    try:
        dev = torch.device("cuda", 0)
        pa, pb, po = work / "a.safetensors", work / "b.safetensors", work / "out.safetensors"
        bench_e2e.write_checkpoint(pa, manifest, 100, dev)
        bench_e2e.write_checkpoint(pb, manifest, 101, dev)
        amd.merge(amd.rotate(amd.model(str(pa)), amd.model(str(pb)), alignment=1.0, alpha=0.0), output=str(po), output_dtype=torch.float32,
                  total_buffer_size=8 << 30)
        assert merging.last_stats["generic_keys"] == 0
        a, b, o = InSafetensorsDict(pa), InSafetensorsDict(pb), InSafetensorsDict(po)
        assert set(o.keys()) == {k for k, _, _ in manifest}
        tall, worst_norm, worst_asym, worst_eig = 0, 0.0, 0.0, 0.0
        by_kind = {}
        for k, s, _ in manifest:
            if len(s) <= 1:
                assert torch.equal(o[k], a[k].float()), k                 # lerp(a, b, 0) = a (svd.py:46-47 for 1-d keys)
                continue
            n, p = oracle.rotate_shape_2d(s)
            by_kind.setdefault("conv" if len(s) == 4 and s[2] * s[3] != 1 else "tall" if n > p else "square" if n == p else "wide", []).append(k)
            if n <= p:
                continue
            tall += 1
            a2, b2, o2 = (x[k].to(dev).reshape(n, p).double() for x in (a, b, o))
            ac, bc, oc = a2 - a2.mean(0), b2 - b2.mean(0), o2 - b2.mean(0)
            rn = (oc.norm(dim=1) - ac.norm(dim=1)).norm() / ac.norm()
            m = oc.T @ bc
            asym = (m - m.T).norm() / m.norm()
            worst_norm, worst_asym = max(worst_norm, float(rn)), max(worst_asym, float(asym))
            assert rn <= 1e-5 and asym <= 1e-4, (k, s, float(rn), float(asym))
            if p <= 640 or tall % 16 == 0:
                ev = torch.linalg.eigvalsh((m + m.T) / 2)
                worst_eig = max(worst_eig, float(-ev[0] / ev[-1]))
                assert ev[0] >= -1e-5 * ev[-1], (k, s, float(ev[0]), float(ev[-1]))
        assert tall > 100, tall
        print(f"config 4 full size: {tall} tall keys, worst row-norm defect {worst_norm:.2e}, asymmetry {worst_asym:.2e}, negative eigenvalue {worst_eig:.2e}")
        for kind, keys in sorted(by_kind.items()):
            keys = sorted(keys, key=lambda k: int(np.prod(o[k].shape)))
            for k in {keys[0], keys[len(keys) // 2]}:
                if int(np.prod(o[k].shape)) > 8_000_000:
                    continue
                ta, tb = a[k].float().numpy(), b[k].float().numpy()
                want = oracle.rotate(ta, tb, alignment=1.0, alpha=0.0, centralization=1.0)
                rel = rotate_rel_error(oracle, ta, tb, o[k].numpy(), want, 1.0)
                assert rel <= 1e-3, (kind, k, o[k].shape, rel)
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ------------------------------------------------------------------ truncate_rank (svd.py:89-143, exact mode)
def _rel(got, want):
    return float(np.linalg.norm(got.astype(np.float64) - want.astype(np.float64))) / max(float(np.linalg.norm(want.astype(np.float64))), 1e-30)


def test_truncate_rank_golden(oracle):
    """Against the reference's own outputs (tests/golden/truncate_rank.npz): <= 1e-3 relative Frobenius error, in every input dtype
    the key may arrive in; both values of use_approximate_basis run the exact path."""
    import sd_mecha_amd as amd
    n = 0
    for name, a, r, want in truncate_rank_cases(oracle):
        for dt in (torch.float32, torch.bfloat16):
            ta = torch.from_numpy(a).cuda().to(dt)
            if not torch.equal(ta.float().cpu(), torch.from_numpy(a)):
                continue                                            # the golden input is not representable in this dtype
            got, trials = amd.truncate_rank(ta, rank=r, use_approximate_basis=bool(n & 1), return_trials=True)
            rel = _rel(got.cpu().numpy(), want)
            print(f"truncate_rank {name} {dt}: relative error {rel:.3e} after {trials} polar solves")
            assert got.dtype == torch.float32 and got.shape == ta.shape
            assert rel <= 1e-3, (name, dt, rel, trials)
            n += 1
    assert n >= 10
    x = torch.randn(6, 5, device="cuda")
    assert torch.equal(amd.truncate_rank(x, rank=5), x) and torch.equal(amd.truncate_rank(x, rank=99), x)       # svd.py:113-114
    assert not amd.truncate_rank(x, rank=0).any()                                                               # svd.py:115-116
    assert torch.equal(amd.truncate_rank(x[0], rank=2), x[0])                                                   # svd.py:99-100
    out = amd.truncate_rank(x.to(torch.bfloat16), rank=2, out_dtype=torch.bfloat16)
    assert out.dtype == torch.bfloat16


@pytest.mark.parametrize("shape,rank", [((1280, 1280), 8), ((1280, 1280), 640), ((640, 2048), 64), ((320, 320, 3, 3), 32), ((5120, 1280), 128),
                                        ((2048, 77), 16)])
def test_truncate_rank_sdxl_shapes_vs_oracle(oracle, shape, rank):
    """SDXL's shapes (conv keys flatten to out x (in kh kw), svd.py:110): against the fp64 oracle, and as properties -- the result has
    rank `rank` exactly where its singular values say so, and the residual is orthogonal to it."""
    import sd_mecha_amd as amd
    rng = np.random.default_rng(89 + rank)
    a = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=shape).astype(np.float32), "bfloat16"), "bfloat16")
    want = oracle.truncate_rank(a, rank)
    got, trials = amd.truncate_rank(torch.from_numpy(a).cuda().to(torch.bfloat16), rank=rank, return_trials=True)
    got = got.cpu().numpy()
    rel = _rel(got, want)
    print(f"truncate_rank {shape} rank {rank}: relative error {rel:.3e} after {trials} polar solves")
    assert rel <= 1e-3, (shape, rank, rel, trials)
    g2, a2 = got.reshape(shape[0], -1).astype(np.float64), a.reshape(shape[0], -1).astype(np.float64)
    sv = np.linalg.svd(g2, compute_uv=False)
    assert sv[rank] <= 1e-4 * sv[0] and sv[rank - 1] > 1e-2 * sv[0], (sv[rank - 1] / sv[0], sv[rank] / sv[0])
    resid = a2 - g2
    assert abs(np.vdot(resid, g2)) <= 1e-4 * np.linalg.norm(g2) ** 2                     # A - A_r is orthogonal to A_r


def test_truncate_rank_low_rank_and_clustered_inputs(oracle):
    """Inputs whose rank is below the request (the cut falls among zero singular values: the key comes back whole) and a spectrum
    with a tight cluster away from the cut."""
    import sd_mecha_amd as amd
    rng = np.random.default_rng(7)
    low = (rng.normal(size=(384, 12)) @ rng.normal(size=(12, 320))).astype(np.float32)
    got = amd.truncate_rank(torch.from_numpy(low).cuda(), rank=40).cpu().numpy()
    assert _rel(got, low) <= 1e-3
    u, _ = np.linalg.qr(rng.normal(size=(300, 300)))
    v, _ = np.linalg.qr(rng.normal(size=(260, 260)))
    sv = np.concatenate([np.full(5, 3.0) * (1 + 1e-5 * np.arange(5)), np.linspace(1.0, 0.01, 255)])
    a = ((u[:, :260] * sv) @ v.T).astype(np.float32)
    for rank in (5, 20):
        got = amd.truncate_rank(torch.from_numpy(a).cuda(), rank=rank).cpu().numpy()
        assert _rel(got, oracle.truncate_rank(a, rank)) <= 1e-3, rank
