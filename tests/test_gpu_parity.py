"""GPU parity: the HIP path (through the C ABI) against the committed golden fixtures (outputs of the real
reference) and against the CPU oracle on seeded inputs.  Bars (BASELINE.json north_star):
  * streaming merges: bit-exact in fp32; <= 1 ulp of the output dtype after the cast (checked: bit-exact);
  * TIES/DARE thresholds and keep masks: bit-exact; outputs <= 1 ulp of the output dtype;
  * reductions (slerp / perpendicular): <= 1 ulp of the output dtype, plus an fp32-level bound.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, ulp_distance_f32, ulp_distance_half

pytestmark = pytest.mark.gpu

TDT = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}


@pytest.fixture(scope="module")
def amd():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import sd_mecha_amd
    from sd_mecha_amd import _lib
    _lib.lib()          # fail loudly if the HIP library is missing
    return sd_mecha_amd


def dev(bits: np.ndarray, dtype: str) -> torch.Tensor:
    """numpy bit patterns (uint16 for half types, uint32/float32 for fp32) -> device tensor."""
    bits = np.ascontiguousarray(bits)
    if dtype == "float32":
        return torch.from_numpy(bits.view(np.float32).copy()).cuda()
    return torch.from_numpy(bits.view(np.int16).copy()).view(TDT[dtype]).cuda()


def host_f32(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def host_bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.float32:
        return t.numpy().view(np.uint32)
    return t.view(torch.int16).numpy().view(np.uint16)


def f32_from_bits(u32):
    return np.array(u32, dtype=np.uint32).view(np.float32)


def assert_f32_bits(t: torch.Tensor, want_u32, what):
    got = host_bits(t).reshape(-1)
    want = np.asarray(want_u32, dtype=np.uint32).reshape(-1)
    bad = got != want
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} fp32 bit mismatches"


# ------------------------------------------------------------------ golden fixtures
def test_golden_streaming_bit_exact(amd):
    g = load_golden("streaming")
    for c in range(int(g["n_cases"])):
        pre, dtype = f"s{c}", str(g[f"s{c}_dtype"])
        shape = tuple(int(x) for x in g[f"{pre}_shape"])
        a, b, cc = (dev(g[f"{pre}_{n}"], dtype).reshape(shape) for n in "abc")
        for alpha in (0.5, 0.3, 0.75, 0.0, 1.0):
            assert_f32_bits(amd.weighted_sum(a, b, alpha), g[f"{pre}_wsum_{alpha}"], f"{pre} wsum {alpha}")
            assert_f32_bits(amd.add_difference(a, b, alpha), g[f"{pre}_adddiff_{alpha}"], f"{pre} adddiff {alpha}")
        assert_f32_bits(amd.subtract(a, b), g[f"{pre}_sub"], f"{pre} sub")
        assert_f32_bits(amd.scale(a, 0.7), g[f"{pre}_scale_0.7"], f"{pre} scale")
        for n in (2, 3, 8):
            ms = [a, b, cc][:min(n, 3)] + [dev(g[f"{pre}_navg{n}_extra{j}"], dtype).reshape(shape) for j in range(max(0, n - 3))]
            assert_f32_bits(amd.n_average(*ms), g[f"{pre}_navg{n}"], f"{pre} navg{n}")
        # fused output cast == torch's .to(dtype) of the fp32 result
        assert np.array_equal(host_bits(amd.weighted_sum(a, b, 0.3, out_dtype=torch.bfloat16)).reshape(-1), g[f"{pre}_wsum_0.3_as_bf16"].reshape(-1))
        assert np.array_equal(host_bits(amd.weighted_sum(a, b, 0.3, out_dtype=torch.float16)).reshape(-1), g[f"{pre}_wsum_0.3_as_f16"].reshape(-1))


def test_golden_alpha_tensor_bit_exact(amd, oracle):
    """A full alpha tensor (block weights: per element, per row, per column) through mecha_weighted_sum_tensor /
    mecha_add_difference_tensor: the reference's bodies bit for bit, then a larger ragged case against the oracle."""
    g = load_golden("alpha_tensor")
    for c in range(int(g["n_cases"])):
        pre, dtype = f"a{c}", str(g[f"a{c}_dtype"])
        shape, ashape = tuple(int(x) for x in g[f"{pre}_shape"]), tuple(int(x) for x in g[f"{pre}_alpha_shape"])
        a, b = (dev(g[f"{pre}_{n}"], dtype).reshape(shape) for n in "ab")
        alpha = torch.from_numpy(f32_from_bits(g[f"{pre}_alpha"]).reshape(ashape).copy())          # a CPU tensor, as the reference passes it
        assert_f32_bits(amd.weighted_sum(a, b, alpha), g[f"{pre}_wsum"], f"{pre} wsum")
        assert_f32_bits(amd.add_difference(a, b, alpha), g[f"{pre}_adddiff"], f"{pre} adddiff")
    rng = np.random.default_rng(5)
    n = 3 * 1024 * 1024 + 5
    a32 = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16"), "bfloat16")
    b32 = oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16"), "bfloat16")
    w = rng.uniform(-0.5, 1.5, size=n).astype(np.float32)
    ta, tb = (torch.from_numpy(x).cuda().to(torch.bfloat16) for x in (a32, b32))
    tw = torch.from_numpy(w).cuda()
    for out_dtype, name in ((torch.float32, "float32"), (torch.bfloat16, "bfloat16")):
        got = host_bits(amd.weighted_sum(ta, tb, tw, out_dtype=out_dtype)).reshape(-1)
        want = oracle.from_f32(oracle.weighted_sum(a32, b32, w), name)
        assert np.array_equal(got, want.view(got.dtype).reshape(-1)), ("wsum", name)
        got = host_bits(amd.add_difference(ta, tb, tw, out_dtype=out_dtype)).reshape(-1)
        want = oracle.from_f32(oracle.add_difference(a32, b32, w), name)
        assert np.array_equal(got, want.view(got.dtype).reshape(-1)), ("adddiff", name)
    with pytest.raises(ValueError):
        amd.weighted_sum(ta[:8], tb[:8], torch.ones(3, 8))                 # would broadcast the KEY, not the weights


def test_golden_reductions(amd, oracle):
    g = load_golden("streaming")
    for c in range(int(g["n_cases"])):
        pre, dtype = f"s{c}", str(g[f"s{c}_dtype"])
        if f"{pre}_perp" not in g:
            continue
        shape = tuple(int(x) for x in g[f"{pre}_shape"])
        a, b, cc = (dev(g[f"{pre}_{n}"], dtype).reshape(shape) for n in "abc")
        checks = [(amd.slerp(a, b, al), g[f"{pre}_slerp_{al}"]) for al in (0.5, 0.3)]
        checks.append((amd.perpendicular_component(a, b), g[f"{pre}_perp"]))
        checks.append((amd.add_perpendicular(a, b, cc, 1.0), g[f"{pre}_addperp_1.0"]))
        for got_t, want_bits in checks:
            got = host_f32(got_t)
            want = f32_from_bits(want_bits).reshape(got.shape)
            assert np.abs(got - want).max() <= 2e-6 * (np.abs(want).max() + 1e-30)
            d = ulp_distance_half(oracle.from_f32(got, "bfloat16"), oracle.from_f32(want, "bfloat16"))
            assert d.max() <= 1


def test_golden_ties(amd, oracle):
    g = load_golden("ties")
    for c in range(int(g["n_cases"])):
        pre, dtype = f"t{c}", str(g[f"t{c}_dtype"])
        base = dev(g[pre + "_base"], dtype)
        models = [dev(m, dtype) for m in g[pre + "_models"]]
        deltas = [amd.subtract(m, base) for m in models]
        tag = "bf16" if dtype == "bfloat16" else "f16"
        for k in (0.2, 0.5, 1.0):
            out, thr = amd.ties_sum(*deltas, k=k, return_thresholds=True)
            assert np.array_equal(host_bits(thr), g[f"{pre}_k{k}_thr"].reshape(-1)), f"{pre} k={k}: thresholds differ"
            want = f32_from_bits(g[f"{pre}_k{k}_v0_out"]).reshape(out.shape)
            assert ulp_distance_f32(host_f32(out), want).max() <= 1
            n_vec = (want.size // 16) * 16       # torch sums the last numel%16 columns in another order (oracle header)
            assert np.array_equal(host_bits(out).reshape(-1)[:n_vec], g[f"{pre}_k{k}_v0_out"].reshape(-1)[:n_vec])
            # keep masks, observed through single-delta calls: out = d * keep
            keep_want = np.unpackbits(g[f"{pre}_k{k}_keep"], axis=1)[:, :want.size].astype(bool)
            for m, d in enumerate(deltas):
                single = host_f32(amd.ties_sum(d, k=k)).reshape(-1)
                dh = host_f32(d).reshape(-1)
                assert np.array_equal(single != 0, keep_want[m] & (dh != 0)), f"{pre} k={k} delta {m}: keep mask differs"
            # fused graph from the half-precision inputs, output cast included
            merged, thr2 = amd.add_difference_ties(base, *models, k=k, alpha=1.0, out_dtype=TDT[dtype], return_thresholds=True)
            assert np.array_equal(host_bits(thr2), g[f"{pre}_k{k}_thr"].reshape(-1))
            dist = ulp_distance_half(host_bits(merged).reshape(-1), g[f"{pre}_k{k}_merged_{tag}"].reshape(-1))
            assert dist.max() <= 1 and (dist > 0).mean() <= 0.002
        # vote_sgn=True: elected by an fp32 sum; compare with the oracle (same index-order sum)
        out_v = amd.ties_sum(*deltas, k=0.2, vote_sgn=True)
        want_v = oracle.ties_sum([host_f32(d) for d in deltas], k=0.2, vote_sgn=True)
        assert ulp_distance_f32(host_f32(out_v), want_v).max() <= 1


def test_golden_dare(amd, oracle):
    g = load_golden("dare")
    for c in range(int(g["n_cases"])):
        pre = f"d{c}"
        deltas = [dev(d, "float32") for d in g[pre + "_deltas"]]
        seed, p = int(g[pre + "_seed"]), float(g[pre + "_p"])
        numel = deltas[0].numel()
        bits = amd.dare_draw_masks(seed, p, len(deltas), numel, deltas[0].device)
        unpacked = ((bits.cpu().numpy().view(np.uint32)[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(len(deltas), -1)[:, :numel]
        assert np.array_equal(np.packbits(unpacked.astype(bool), axis=1), g[pre + "_masks"]), "bernoulli stream differs from torch"
        for rescale in (True, False):
            want = f32_from_bits(g[f"{pre}_stock1_rescale{int(rescale)}"]).reshape(deltas[0].shape)
            got = amd.ties_sum_with_dropout(*deltas, probability=p, rescale=rescale, k=0.5, apply_stock=True, seed=seed)
            assert ulp_distance_f32(host_f32(got), want).max() <= 1
        # apply_stock=False (the reference default) APPLIES the model-stock factor (ties.py:90); it goes through torch
        # reductions whose order is not fixed: same bound as the oracle's own golden test
        if deltas[0].dim() >= 1 and len(deltas) >= 2:
            for rescale in (True, False):
                want = f32_from_bits(g[f"{pre}_stock0_rescale{int(rescale)}"]).reshape(deltas[0].shape)
                got = amd.ties_sum_with_dropout(*deltas, probability=p, rescale=rescale, k=0.5, apply_stock=False, seed=seed)
                assert np.allclose(host_f32(got), want, rtol=2e-5, atol=1e-9), (pre, rescale)
        with pytest.raises(NotImplementedError):
            amd.ties_sum_with_dropout(*deltas, probability=p, k=0.5, apply_median=True, seed=seed)
    assert amd.ties_sum_with_dropout(probability=0.5) == 0
    assert amd.ties_sum_with_dropout(torch.zeros(4, device="cuda"), probability=1.0) == 0


def test_golden_model_stock_and_dropout(amd, oracle):
    """Row (f)4: model_stock (ties.py:137-159) and dropout (ties.py:202-238) on the HIP path against the reference's own bodies,
    and at a larger size against the oracle.  dropout is bit-exact: the masks are drawn on the host with numpy exactly as the
    reference draws them."""
    from test_oracle_golden import stock_close
    g = load_golden("stock_dropout")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}"
        deltas = [dev(d, "float32") for d in g[pre + "_deltas"]]
        want = f32_from_bits(g[pre + "_stock"]).reshape(deltas[0].shape)
        assert stock_close(host_f32(amd.model_stock(*deltas)), want), pre
        for seed, p, rescale in g["dropout_settings"]:
            got = amd.dropout(*deltas, probability=float(p), rescale=float(rescale), seed=int(seed))
            want = f32_from_bits(g[f"{pre}_dropout_s{int(seed)}"]).reshape(deltas[0].shape)
            assert ulp_distance_f32(host_f32(got), want).max() == 0, (pre, seed)
    rng = np.random.default_rng(5)
    shape = (640, 2560 + 8)
    f32 = [oracle.to_f32(oracle.from_f32(rng.normal(0, 0.02, size=shape).astype(np.float32), "bfloat16"), "bfloat16") for _ in range(4)]
    tens = [torch.from_numpy(x).cuda().to(torch.bfloat16) for x in f32]
    assert stock_close(host_f32(amd.model_stock(*tens)), oracle.model_stock(f32))
    got = amd.dropout(*tens, probability=0.9, seed=123, out_dtype=torch.bfloat16)
    want = oracle.from_f32(oracle.dropout(f32, probability=0.9, seed=123), "bfloat16")
    assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1), want.reshape(-1))
    assert amd.dropout() == 0
    with pytest.raises(NotImplementedError):
        amd.dropout(*tens, overlap=0.5)
    with pytest.raises(NotImplementedError):
        amd.model_stock(tens[0])


def test_upstream_known_answers_on_gpu(amd):
    """/root/reference/tests/merge_methods/test_ties.py:5-58, run through the HIP path."""
    models = [
        torch.tensor([[-1., 2., 3., 4.], [4., -3., 2., 1.], [3., 4., 1., -2.], [2., 1., -4., 3.]], device="cuda"),
        torch.tensor([[3., 4., 1., -2.], [2., 1., -4., 3.], [-1., 2., 3., 4.], [4., -3., 2., 1.]], device="cuda"),
    ]
    expected = torch.tensor([[3., 3., 3., 0.], [3., -3., 0., 3.], [3., 3., 3., 0.], [3., -3., 0., 3.]], device="cuda")
    actual = amd.ties_sum(*models, k=0.5)
    assert torch.allclose(actual, expected)
    actual2 = amd.ties_sum(*models, k=0.5, vote_sgn=True)
    assert not torch.allclose(actual, actual2)
    assert amd.ties_sum(torch.tensor([], device="cuda"), torch.tensor([], device="cuda"), k=0.5).numel() == 0
    assert amd.ties_sum(k=0.5) == 0.0


# ------------------------------------------------------------------ seeded inputs vs the oracle
SIZES = [1, 5, 8, 2049, 4096, 4097, 65536 + 3, (1 << 20) + 13]


@pytest.mark.parametrize("dtype", ["bfloat16", "float16", "float32"])
def test_streaming_vs_oracle(amd, oracle, dtype):
    rng = np.random.default_rng(7)
    for n in SIZES:
        raw = [rng.normal(0, 0.02, size=n).astype(np.float32) for _ in range(9)]
        bits = [r if dtype == "float32" else oracle.from_f32(r, dtype) for r in raw]
        f32 = [oracle.to_f32(b, dtype) for b in bits]
        t = [dev(b, dtype) for b in bits]
        for out_dtype in ("float32", "bfloat16", "float16"):
            od = TDT[out_dtype]
            cases = [
                (amd.weighted_sum(t[0], t[1], 0.37, out_dtype=od), oracle.weighted_sum(f32[0], f32[1], 0.37)),
                (amd.weighted_sum(t[0], t[1], 0.5, out_dtype=od), oracle.weighted_sum(f32[0], f32[1], 0.5)),
                (amd.add_difference(t[0], t[1], 0.8, out_dtype=od), oracle.add_difference(f32[0], f32[1], 0.8)),
                (amd.n_average(*t[:8], out_dtype=od), oracle.n_average(f32[:8])),
                (amd.n_average(*t, out_dtype=od), oracle.n_average(f32)),
                (amd.subtract(t[2], t[3], out_dtype=od), oracle.subtract(f32[2], f32[3])),
                (amd.scale(t[4], 1.7, out_dtype=od), oracle.scale(f32[4], 1.7)),
            ]
            for got, want in cases:
                want_bits = want.view(np.uint32) if out_dtype == "float32" else oracle.from_f32(want, out_dtype)
                assert np.array_equal(host_bits(got).reshape(-1), want_bits.reshape(-1)), (dtype, out_dtype, n)


def test_unaligned_views_take_the_scalar_path(amd, oracle):
    rng = np.random.default_rng(8)
    n = 10007
    raw = [oracle.from_f32(rng.normal(0, 0.02, size=n + 3).astype(np.float32), "bfloat16") for _ in range(3)]
    t = [dev(r, "bfloat16")[1:n + 1] for r in raw]            # 2-byte offset: not 16-byte aligned
    f32 = [oracle.to_f32(r, "bfloat16")[1:n + 1] for r in raw]
    assert t[0].data_ptr() % 16 != 0
    got = amd.n_average(*t, out_dtype=torch.bfloat16)
    assert np.array_equal(host_bits(got), oracle.from_f32(oracle.n_average(f32), "bfloat16"))
    got = amd.add_difference_ties(t[0], t[1], t[2], k=0.3, out_dtype=torch.bfloat16)
    want = oracle.from_f32(oracle.add_difference_ties(f32[0], f32[1:], k=0.3), "bfloat16")
    assert ulp_distance_half(host_bits(got), want).max() <= 1


# every model count has its own instance of the full-tile merge kernel (1..8 models; 9 stays with the general kernel)
@pytest.mark.parametrize("n_models,k", [(1, 0.5), (3, 0.2), (7, 0.2), (3, 1.0), (3, 0.0), (9, 0.35), (2, 0.3), (4, 0.2), (5, 0.2), (6, 0.5), (8, 0.2)])
def test_ties_vs_oracle(amd, oracle, n_models, k):
    rng = np.random.default_rng(1000 + n_models)
    for n in [1, 7, 4096, 4099, 300_000, (1 << 21) + 5]:
        base_b = oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16")
        models_b = [oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16") for _ in range(n_models)]
        if n >= 4096:
            models_b[0][: n // 3] = base_b[: n // 3]               # exact-zero deltas: massive ties at |d| = 0
        base, models = oracle.to_f32(base_b, "bfloat16"), [oracle.to_f32(m, "bfloat16") for m in models_b]
        deltas = [oracle.subtract(m, base) for m in models]
        want, det = oracle.ties_sum(deltas, k=k, details=True)
        want_merged = oracle.from_f32(oracle.add_difference(base, want, 1.0), "bfloat16")
        tb, tm = dev(base_b, "bfloat16"), [dev(m, "bfloat16") for m in models_b]
        got, thr = amd.add_difference_ties(tb, *tm, k=k, alpha=1.0, out_dtype=torch.bfloat16, return_thresholds=True)
        assert np.array_equal(host_bits(thr), det["thresholds"].view(np.uint32)), (n, "thresholds")
        assert np.array_equal(host_bits(got), want_merged), (n, "merged output")
        d32 = [dev(d, "float32") for d in deltas]
        got2, thr2 = amd.ties_sum(*d32, k=k, return_thresholds=True)
        assert np.array_equal(host_bits(thr2), det["thresholds"].view(np.uint32))
        assert np.array_equal(host_bits(got2), want.view(np.uint32)), (n, "ties_sum fp32 output")


@pytest.mark.parametrize("select", ["radix", "fail"])
def test_ties_selection_fallbacks(amd, oracle, monkeypatch, select):
    """The threshold selection has two pipelines (ties.hip header).  MECHA_TIES_SELECT=radix runs the radix select alone;
    =fail makes every bracket of the default pipeline miss, so each (key, delta) pair goes through the in-stream
    fallback.  Both must give the thresholds and outputs of the default path, i.e. the oracle's."""
    monkeypatch.setenv("MECHA_TIES_SELECT", select)
    rng = np.random.default_rng(77)
    for n, n_models, k in [(4099, 3, 0.2), (300_001, 3, 0.2), ((1 << 20) + 3, 5, 0.35)]:
        base_b = oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16")
        models_b = [oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16") for _ in range(n_models)]
        base, models = oracle.to_f32(base_b, "bfloat16"), [oracle.to_f32(m, "bfloat16") for m in models_b]
        want, det = oracle.ties_sum([oracle.subtract(m, base) for m in models], k=k, details=True)
        got, thr = amd.add_difference_ties(dev(base_b, "bfloat16"), *[dev(m, "bfloat16") for m in models_b], k=k, alpha=1.0,
                                           out_dtype=torch.bfloat16, return_thresholds=True)
        assert np.array_equal(host_bits(thr), det["thresholds"].view(np.uint32)), (select, n)
        assert np.array_equal(host_bits(got), oracle.from_f32(oracle.add_difference(base, want, 1.0), "bfloat16")), (select, n)


@pytest.mark.parametrize("levels", [3, 40, 5000])
def test_ties_heavy_ties_and_skewed_keys(amd, oracle, levels):
    """Deltas drawn from a few distinct magnitudes (heavy ties inside one histogram bin -> the bracket's list would
    overflow -> per-pair fallback), sorted keys (the stratified sample sees a ramp) and keys whose large values sit in
    one corner (the sample misses them)."""
    rng = np.random.default_rng(levels)
    n = 700_003
    grid = (rng.normal(0, 0.02, size=levels)).astype(np.float32)
    cases = {
        "quantised": [grid[rng.integers(0, levels, size=n)] for _ in range(3)],
        "sorted": [np.sort(rng.normal(0, 0.02, size=n).astype(np.float32)) for _ in range(2)],
        "corner": [np.concatenate([rng.normal(0, 1e-4, size=n - 5000), rng.normal(0, 1.0, size=5000)]).astype(np.float32) for _ in range(2)],
    }
    for name, deltas in cases.items():
        for k in (0.2, 0.006):
            want, det = oracle.ties_sum(deltas, k=k, details=True)
            got, thr = amd.ties_sum(*[dev(d, "float32") for d in deltas], k=k, return_thresholds=True)
            assert np.array_equal(host_bits(thr), det["thresholds"].view(np.uint32)), (name, k, "thresholds")
            assert np.array_equal(host_bits(got), want.view(np.uint32)), (name, k, "output")


def test_ties_count_pass_bracket_edges(amd, oracle):
    """The count pass bins with clamps that need room on both sides of the bracket (ties.hip, "Clamped binning"); brackets at the
    ends of the pattern range, over many octaves or one pattern wide take its compare-and-select form or the one-pattern cells.
    Thresholds and outputs stay bit-exact in every case, for fp32 deltas (integer clamps) and for bf16 models minus a base
    (clamp and |x| in one v_med3_f32), at a ragged key size."""
    rng = np.random.default_rng(77)
    n = 600_011                                                       # not a multiple of 8: the key's last position is ragged
    def mix(*parts):
        x = np.concatenate([np.asarray(p, dtype=np.float32) for p in parts])
        x = np.resize(x, n)
        rng.shuffle(x)
        return x * rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=n)
    cases = {
        # kth among denormal-range values above a mass of exact zeros: the bracket starts at (or a few patterns above) zero
        "near_zero": ([mix(np.zeros(n // 2), rng.uniform(1e-42, 3e-39, size=n // 4), rng.normal(0, 0.02, size=n // 4)) for _ in range(3)], (0.35, 0.6)),
        # kth close to the largest finite patterns: no room for clamp patterns below infinity
        # (only the first delta holds them: the sum of two would overflow)
        "near_max": ([mix(rng.uniform(3.0e38, float(np.finfo(np.float32).max), size=n // 3), rng.normal(0, 1.0, size=n - n // 3)),
                      mix(rng.normal(0, 1.0, size=n))], (0.2, 0.02)),
        # 60 octaves inside the bracket: cells wider than 2^18 patterns
        "wide_span": ([mix(np.exp(rng.uniform(np.log(1e-30), np.log(1e30), size=n))) for _ in range(2)], (0.5, 0.2)),
        # one magnitude only: the bracket is one pattern wide (cells of one pattern)
        "constant": ([mix(np.full(n, 0.0123)) for _ in range(3)], (0.2, 0.9)),
        # two neighbouring patterns
        "two_patterns": ([mix(np.full(n // 2, 0.5), np.full(n - n // 2, np.nextafter(np.float32(0.5), np.float32(1.0)))) for _ in range(2)], (0.5, 0.3)),
    }
    for name, (deltas, ks) in cases.items():
        for k in ks:
            want, det = oracle.ties_sum(deltas, k=k, details=True)
            got, thr = amd.ties_sum(*[dev(d, "float32") for d in deltas], k=k, return_thresholds=True)
            assert np.array_equal(host_bits(thr), det["thresholds"].view(np.uint32)), (name, k, "thresholds")
            assert np.array_equal(host_bits(got), want.view(np.uint32)), (name, k, "output")
    # models minus a base in bf16: deltas of a fine-tune (exact zeros, a few coarse levels) and ordinary ones, ragged size
    base = oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16")
    bf = oracle.to_f32(base, "bfloat16")
    steps = np.array([0.0, 0.0, 0.0, 2.0 ** -12, -(2.0 ** -12), 2.0 ** -11, 2.0 ** -9], dtype=np.float32)
    models = [oracle.from_f32(bf + steps[rng.integers(0, len(steps), size=n)], "bfloat16") for _ in range(2)]
    models.append(oracle.from_f32(bf + rng.normal(0, 0.003, size=n).astype(np.float32), "bfloat16"))
    f32 = [oracle.to_f32(m, "bfloat16") for m in models]
    for k in (0.2, 0.55, 0.9):
        ties, det = oracle.ties_sum([oracle.subtract(m, bf) for m in f32], k=k, details=True)
        want = oracle.from_f32(oracle.add_difference(bf, ties, 1.0), "bfloat16")
        got, thr = amd.add_difference_ties(dev(base, "bfloat16"), *[dev(m, "bfloat16") for m in models], k=k, alpha=1.0,
                                           out_dtype=torch.bfloat16, return_thresholds=True)
        assert np.array_equal(host_bits(thr), det["thresholds"].view(np.uint32)), ("finetune", k, "thresholds")
        assert np.array_equal(host_bits(got), want), ("finetune", k, "output")


@pytest.mark.parametrize("shape", [(5000,), (300, 1280), (64, 32, 3, 3), (7, 1), (1000, 9)])
def test_ties_sum_extended_model_stock_vs_oracle(amd, oracle, shape):
    """ties_sum_extended with the model-stock factor (ties.py:68-94, :146-159): per-row cosines along the last dimension
    (rows of 1, 3, 9, 1280 and 5000 elements: lanes that span several rows, rows that span several waves)."""
    rng = np.random.default_rng(len(shape) * 100 + shape[-1])
    for n_models in (2, 3, 5):
        common = rng.normal(0, 0.02, size=shape).astype(np.float32)
        deltas = [(common + rng.normal(0, 0.01, size=shape)).astype(np.float32) for _ in range(n_models)]      # correlated: cos away from 0
        want = oracle.ties_sum_extended(deltas, k=0.3, apply_stock=False)
        got = amd.ties_sum_extended(*[dev(d, "float32") for d in deltas], k=0.3, apply_stock=False)
        assert got.shape == tuple(shape)
        assert np.allclose(host_f32(got), want, rtol=2e-5, atol=1e-9), (shape, n_models, np.abs(host_f32(got) - want).max())
        # apply_stock=True is plain ties_sum
        plain = amd.ties_sum_extended(*[dev(d, "float32") for d in deltas], k=0.3, apply_stock=True)
        assert np.array_equal(host_bits(plain), host_bits(amd.ties_sum(*[dev(d, "float32") for d in deltas], k=0.3)))
    with pytest.raises(NotImplementedError):
        amd.ties_sum_extended(dev(deltas[0], "float32"), k=0.3)                              # one delta: the reference raises too
    with pytest.raises(ValueError):
        amd.ties_sum_extended(torch.tensor(1.0, device="cuda"), torch.tensor(2.0, device="cuda"))


def test_reductions_vs_oracle(amd, oracle):
    rng = np.random.default_rng(9)
    for n in [2, 9, 4097, (1 << 20) + 7]:
        bits = [oracle.from_f32(rng.normal(0, 0.02, size=n).astype(np.float32), "bfloat16") for _ in range(3)]
        f32 = [oracle.to_f32(b, "bfloat16") for b in bits]
        t = [dev(b, "bfloat16") for b in bits]
        cases = [
            (amd.slerp(t[0], t[1], 0.3, out_dtype=torch.bfloat16), oracle.slerp(f32[0], f32[1], 0.3)),
            (amd.perpendicular_component(t[0], t[1], out_dtype=torch.bfloat16), oracle.perpendicular_component(f32[0], f32[1])),
            (amd.add_perpendicular(t[0], t[1], t[2], 0.6, out_dtype=torch.bfloat16), oracle.add_perpendicular(f32[0], f32[1], f32[2], 0.6)),
        ]
        mag = np.abs(f32[0]) + np.abs(f32[1]) + np.abs(f32[2])
        for got, want in cases:
            d = ulp_distance_half(host_bits(got), oracle.from_f32(want, "bfloat16"))
            # <= 1 ulp of the output dtype; where the formula cancels (result << operands) an ulp of the tiny
            # result is below the fp32 rounding noise of the operands, so the floor is that noise instead
            noise = np.abs(host_f32(got) - want) <= 1e-6 * mag
            assert ((d <= 1) | noise).all() and (d > 0).mean() <= 0.01, n
    # NaN fallbacks: slerp of parallel vectors (sin(omega) = 0 -> 0/0) falls back to lerp; perp with a = 0 -> zeros
    a = torch.full((1000,), 0.5, device="cuda")
    got = amd.slerp(a, a * 2, 0.25)
    assert torch.equal(got, torch.lerp(a, a * 2, 0.25))
    z = torch.zeros(1000, device="cuda")
    assert torch.equal(amd.perpendicular_component(z, a), z)


def test_non_finite_inputs_raise_like_omit_non_finite(amd):
    a = torch.randn(5000, device="cuda").to(torch.bfloat16)
    b = a.clone()
    b[4321] = float("inf")
    with pytest.raises(amd.NonFiniteStateDictKeyError):
        amd.weighted_sum(a, b, 0.5, key="k")
    with pytest.raises(amd.StateDictKeyError):          # the subclass relation the executor relies on
        amd.n_average(a, b, a)
    with pytest.raises(amd.NonFiniteStateDictKeyError):
        amd.add_difference_ties(a, b, a, k=0.2)
    with pytest.raises(amd.NonFiniteStateDictKeyError):
        amd.omit_non_finite(b)
    # lazy shortcut: alpha == 0 never reads b (linear.py:18-19)
    assert torch.equal(amd.weighted_sum(a, b, 0.0, out_dtype=torch.bfloat16), a)
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    amd.weighted_sum(a, b, 0.5, flags=flags)
    assert int(flags.item()) & 1


def test_host_validation(amd):
    a = torch.randn(16, device="cuda")
    with pytest.raises(ValueError):
        amd.weighted_sum(a, torch.randn(17, device="cuda"))
    with pytest.raises(RuntimeError):
        amd.weighted_sum(a, torch.randn(16))
    with pytest.raises(ValueError):
        amd.n_average(a, a.to(torch.bfloat16))
    from sd_mecha_amd import _lib
    with pytest.raises(_lib.MechaError):
        amd.ties_sum(a, a, k=-0.5)               # the rank would exceed numel: torch.kthvalue raises in the reference
    # k > 1 is legal: kth = max(int((1 - k) * numel), 1) clamps the rank to 1 (ties.py:129), i.e. everything is kept
    b = torch.randn(16, device="cuda")
    assert torch.equal(amd.ties_sum(a, b, k=1.5), amd.ties_sum(a, b, k=1.0))


# ------------------------------------------------------------------ batched (scheduler) entry points
def _arena_setup(oracle, rng, numels, n_models, dtype="bfloat16", gap=8):
    from sd_mecha_amd import _lib
    offs, off = [], 0
    for n in numels:
        offs.append(off)
        off += (n + 7) // 8 * 8 + gap
    total = off
    arenas_bits = [oracle.from_f32(rng.normal(0, 0.02, size=total).astype(np.float32), dtype) for _ in range(n_models)]
    segs = (_lib.Segment * len(numels))(*[_lib.Segment(o, n) for o, n in zip(offs, numels)])
    segs_dev = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.int64).cuda()
    return offs, total, arenas_bits, segs, segs_dev


def test_batched_streaming_and_flags(amd, oracle):
    import ctypes
    from sd_mecha_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    numels = [5, 4096, 1, 70001, 320, 8, 123457]
    offs, total, arenas_bits, segs, segs_dev = _arena_setup(oracle, rng, numels, 4)
    arenas_bits[2][offs[3] + 70000] = 0x7F80                     # +inf in the LAST element of key 3, model 2
    arenas_bits[1][offs[4] + 320] = 0x7FC0                       # NaN in the GAP after key 4: must not flag any key
    arenas = [dev(a, "bfloat16") for a in arenas_bits]
    out = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    flags = torch.zeros(len(numels), dtype=torch.int32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.mecha_batched_n_average(_lib.ptr_array([a.data_ptr() for a in arenas]), 4, out.data_ptr(), segs_dev.data_ptr(),
                                         len(numels), total, _lib.BF16, _lib.BF16, flags.data_ptr(), stream), "batched n_average")
    fl = flags.cpu().numpy()
    assert [int(x) & 1 for x in fl] == [0, 0, 0, 1, 0, 0, 0]
    ob = host_bits(out)
    for s, (o, n) in enumerate(zip(offs, numels)):
        if s == 3:
            continue
        want = oracle.from_f32(oracle.n_average([oracle.to_f32(a[o:o + n], "bfloat16") for a in arenas_bits]), "bfloat16")
        assert np.array_equal(ob[o:o + n], want), s
    flags.zero_()
    _lib.check(L.mecha_batched_weighted_sum(arenas[0].data_ptr(), arenas[3].data_ptr(), out.data_ptr(), segs_dev.data_ptr(), len(numels),
                                            total, _lib.BF16, _lib.BF16, 0.25, flags.data_ptr(), stream), "batched weighted_sum")
    assert int(flags.sum().item()) == 0
    ob = host_bits(out)
    for o, n in zip(offs, numels):
        want = oracle.from_f32(oracle.weighted_sum(oracle.to_f32(arenas_bits[0][o:o + n], "bfloat16"), oracle.to_f32(arenas_bits[3][o:o + n], "bfloat16"), 0.25), "bfloat16")
        assert np.array_equal(ob[o:o + n], want)


@pytest.mark.parametrize("n_models", [3, 7])
def test_batched_ties(amd, oracle, n_models):
    import ctypes
    from sd_mecha_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(12)
    numels = [5, 4096, 1, 70001, 320, 8, 123457, 4097, 1280 * 1280]
    offs, total, arenas_bits, segs, segs_dev = _arena_setup(oracle, rng, numels, n_models + 1)
    arenas_bits[2][offs[4] + 7] = 0xFF80                        # -inf inside key 4
    arenas = [dev(a, "bfloat16") for a in arenas_bits]
    out = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    flags = torch.zeros(len(numels), dtype=torch.int32, device="cuda")
    thr = torch.zeros((len(numels), n_models), dtype=torch.float32, device="cuda")
    ws = torch.empty(L.mecha_batched_ties_workspace_bytes(n_models, len(numels), total), dtype=torch.uint8, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):      # second call: the workspace is reused without re-initialisation
        flags.zero_()
        _lib.check(L.mecha_batched_add_difference_ties(arenas[0].data_ptr(), _lib.ptr_array([a.data_ptr() for a in arenas[1:]]), n_models,
                                                       out.data_ptr(), segs_dev.data_ptr(), ctypes.addressof(segs), len(numels), total,
                                                       _lib.BF16, _lib.BF16, 0.2, 1.0, 0, thr.data_ptr(), ws.data_ptr(), ws.numel(),
                                                       flags.data_ptr(), stream), "batched ties")
        torch.cuda.synchronize()
    assert [int(x) & 1 for x in flags.cpu().numpy()] == [0, 0, 0, 0, 1, 0, 0, 0, 0]
    ob, th = host_bits(out), host_bits(thr)
    for s, (o, n) in enumerate(zip(offs, numels)):
        if s == 4:
            continue
        base = oracle.to_f32(arenas_bits[0][o:o + n], "bfloat16")
        models = [oracle.to_f32(a[o:o + n], "bfloat16") for a in arenas_bits[1:]]
        deltas = [oracle.subtract(m, base) for m in models]
        want, det = oracle.ties_sum(deltas, k=0.2, details=True)
        assert np.array_equal(th[s], det["thresholds"].view(np.uint32)), (s, "thresholds")
        assert np.array_equal(ob[o:o + n], oracle.from_f32(oracle.add_difference(base, want, 1.0), "bfloat16")), (s, "output")


# ------------------------------------------------------------------ full-size properties (BASELINE sizes)
def test_full_size_properties(amd):
    """At the largest key of the configs (SDXL embedding [49408, 1280] = 63.2 M elements) the oracle is too slow;
    check size-independent properties instead."""
    n = 49408 * 1280
    g = torch.Generator(device="cuda").manual_seed(5)
    a = (torch.randn(n, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    b = (torch.randn(n, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    c = (torch.randn(n, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    # idempotence: lerp(a, a, w) = fma(w, 0, a) = a exactly, so any average of copies of a is a
    assert torch.equal(amd.n_average(a, a, a, a, a, out_dtype=torch.bfloat16), a)
    assert torch.equal(amd.weighted_sum(a, a, 0.3, out_dtype=torch.bfloat16), a)
    # TIES with two identical models: both deltas agree, mean = (d + d) / 2 = d exactly (where kept)
    k = 0.2
    merged, thr = amd.add_difference_ties(a, b, b, k=k, out_dtype=torch.float32, return_thresholds=True)
    d = b.float() - a.float()
    assert thr[0] == thr[1]
    kth = max(int((1 - k) * n), 1)
    t = thr[0]
    below, at_or_below = int((d.abs() < t).sum().item()), int((d.abs() <= t).sum().item())
    assert below < kth <= at_or_below, "threshold is not the kth-smallest |delta|"
    want = torch.where(d.abs() >= t, a.float() + d, a.float())
    assert torch.equal(merged, want)
    # the rank property for independent deltas
    _, thr = amd.add_difference_ties(a, b, c, k=k, return_thresholds=True)
    for m, t in zip((b, c), thr):
        dm = (m.float() - a.float()).abs()
        assert int((dm < t).sum().item()) < kth <= int((dm <= t).sum().item())


def test_full_config2_job_sampled_keys_vs_oracle(amd, oracle):
    """BASELINE config 2 at FULL size -- every key of an SDXL-shaped checkpoint x4 (2 529 keys, 3.47 B elements per model) in ONE
    batched launch sequence, as bench.py and merge() run it -- checked where the oracle can follow: the thresholds of EVERY
    (key, delta) pair by their rank property (the kth-smallest |delta|, counted with torch on the device), and the output and
    flag word of a spread of keys (largest, smallest, ragged tails, the keys around them in the arena) bit for bit."""
    import ctypes
    import pathlib
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    import bench
    from sd_mecha_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free HBM")
    job = bench.Job("sdxl_ties", 0, torch.device("cuda", 0))
    L, lib = job.layout, job.lib
    n_keys, nd = len(L.keys), job.n_models - 1
    thr = torch.zeros((n_keys, nd), dtype=torch.float32, device="cuda")
    job.flags.zero_()
    st = lib.mecha_batched_add_difference_ties(
        job.arenas[0].data_ptr(), job.model_ptrs, nd, job.out.data_ptr(), job.segs_dev.data_ptr(), ctypes.addressof(job.segs_host), n_keys,
        L.total_elems, _lib.BF16, _lib.BF16, 0.2, 1.0, 0, thr.data_ptr(), job.ws.data_ptr(), job.ws.numel(), job.flags.data_ptr(),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "batched ties")
    torch.cuda.synchronize()
    assert int(job.flags.abs().sum().item()) == 0
    # rank property of every threshold (ties.py:128-132): #(|d| < t) < kth <= #(|d| <= t)
    base = job.arenas[0]
    bad = []
    for i, (o, n) in enumerate(zip(L.offsets, L.numels)):
        kth = max(int((1 - 0.2) * n), 1)
        b32 = base[o:o + n].float()
        for d in range(nd):
            dm = (job.arenas[1 + d][o:o + n].float() - b32).abs()
            t = thr[i, d]
            if not (int((dm < t).sum().item()) < kth <= int((dm <= t).sum().item())):
                bad.append((L.keys[i], d))
    assert not bad, bad[:5]
    # outputs of a spread of keys against the oracle
    order = sorted(range(n_keys), key=lambda i: L.numels[i])
    picks = set(order[:6] + order[-3:] + order[len(order) // 2 - 3:len(order) // 2 + 3] + [i for i in range(n_keys) if L.numels[i] % 4096][:6])
    for i in sorted(picks):
        o, n = L.offsets[i], L.numels[i]
        bits = [host_bits(a[o:o + n]) for a in job.arenas]
        b32 = oracle.to_f32(bits[0], "bfloat16")
        deltas = [oracle.subtract(oracle.to_f32(x, "bfloat16"), b32) for x in bits[1:]]
        want, det = oracle.ties_sum(deltas, k=0.2, details=True)
        assert np.array_equal(host_bits(thr[i]), det["thresholds"].view(np.uint32)), (L.keys[i], "thresholds")
        assert np.array_equal(host_bits(job.out[o:o + n]), oracle.from_f32(oracle.add_difference(b32, want, 1.0), "bfloat16")), (L.keys[i], n, "output")


@pytest.mark.parametrize("p", [0.9, 0.5])
def test_full_size_dare_job_thresholds_and_sampled_keys(amd, oracle, p):
    """The SDXL-shaped DARE job at FULL size (2 529 keys x 3 deltas, every key's masks cut from one Bernoulli stream,
    ties.py:29-33) in ONE batched launch sequence, as bench.py runs it.  p = 0.9: every kth value is one of the dropped zeros and
    the thresholds come from the masks' popcounts; p = 0.5: thresholds above zero, bracket select on delta * mask.  Every
    (key, delta) threshold by its rank property on the masked deltas (ties.py:128-132), counted with torch on the device;
    thresholds and outputs of a spread of keys bit for bit against the oracle."""
    import ctypes
    import pathlib
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    import bench
    from sd_mecha_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free HBM")
    job = bench.Job("sdxl_dare", 0, torch.device("cuda", 0), dare_p=p)
    L, lib = job.layout, job.lib
    n_keys, nd = len(L.keys), job.n_models - 1
    thr = torch.full((n_keys, nd), -1.0, dtype=torch.float32, device="cuda")
    job.flags.zero_()
    st = lib.mecha_batched_add_difference_dare(
        job.arenas[0].data_ptr(), job.model_ptrs, nd, job.out.data_ptr(), job.segs_dev.data_ptr(), ctypes.addressof(job.segs_host), n_keys,
        L.total_elems, _lib.BF16, _lib.BF16, 0.2, 1.0, 0, p, 1, job.mask_stream.data_ptr(), job.mask_bits, 0, 0.0, None, None, thr.data_ptr(),
        job.ws.data_ptr(), job.ws.numel(), job.flags.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "batched dare")
    torch.cuda.synchronize()
    assert int(job.flags.abs().sum().item()) == 0
    # bit s of the stream (word s >> 5, bit s & 31) keeps element s - d * numel of delta d
    shifts = torch.arange(32, dtype=torch.int32, device="cuda")
    keep = ((job.mask_stream[:, None] >> shifts) & 1).to(torch.bool).flatten()
    base = job.arenas[0]
    bad = []
    for i, (o, n) in enumerate(zip(L.offsets, L.numels)):
        kth = max(int((1 - 0.2) * n), 1)
        b32 = base[o:o + n].float()
        for d in range(nd):
            dm = ((job.arenas[1 + d][o:o + n].float() - b32) * keep[d * n:(d + 1) * n]).abs()
            t = thr[i, d]
            if not (int((dm < t).sum().item()) < kth <= int((dm <= t).sum().item())):
                bad.append((L.keys[i], d, float(t)))
    assert not bad, bad[:5]
    sizeable = torch.tensor(L.numels, device="cuda") >= 4096        # (a key of a few elements can keep more than its share)
    if p == 0.9:
        assert float(thr[sizeable].max()) == 0.0                    # 90 % dropped, kth at 80 %
    else:
        assert float(thr[sizeable].min()) > 0.0
    order = sorted(range(n_keys), key=lambda i: L.numels[i])
    picks = set(order[:4] + order[len(order) // 2 - 2:len(order) // 2 + 2] + [i for i in order if L.numels[i] <= 4_000_000][-2:]
                + [i for i in range(n_keys) if L.numels[i] % 4096][:4])
    for i in sorted(picks):
        o, n = L.offsets[i], L.numels[i]
        bits = [host_bits(a[o:o + n]) for a in job.arenas]
        b32 = oracle.to_f32(bits[0], "bfloat16")
        deltas = [oracle.subtract(oracle.to_f32(x, "bfloat16"), b32) for x in bits[1:]]
        masks = oracle.dare_masks(job.dare_seed, p, nd, (n,))
        _, det = oracle.ties_sum([dl * mk.astype(np.float32) for dl, mk in zip(deltas, masks)], k=0.2, details=True)
        assert np.array_equal(host_bits(thr[i]), det["thresholds"].view(np.uint32)), (L.keys[i], "thresholds")
        ties = oracle.ties_sum_with_dropout(deltas, probability=p, rescale=True, k=0.2, apply_stock=True, masks=masks)
        assert np.array_equal(host_bits(job.out[o:o + n]), oracle.from_f32(oracle.add_difference(b32, ties, 1.0), "bfloat16")), (L.keys[i], n, "output")


@pytest.mark.parametrize("workload", ["flux_navg", "flux_ties"])
def test_full_config3_stripe_vs_oracle(amd, oracle, workload):
    """BASELINE config 3 at one GPU's FULL share -- the keys LPT deals to rank 0 of 8 from the FLUX-dev diffuser (1.49 B
    elements), 8 bf16 checkpoints -- in ONE batched launch sequence, as bench.py and merge() run it.  n_average: every element
    within one bf16 ulp of the fp64 mean (plus the fp32 rounding of the running mean where the values cancel), a spread of keys bit for bit against the oracle.  TIES with 7 deltas: every
    (key, delta) threshold by its rank property, thresholds and outputs of a spread of keys bit for bit against the oracle."""
    import ctypes
    import pathlib
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    import bench
    from sd_mecha_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free HBM")
    job = bench.Job(workload, 0, torch.device("cuda", 0))
    L, lib = job.layout, job.lib
    n_keys, nd = len(L.keys), job.n_models - 1
    order = sorted(range(n_keys), key=lambda i: L.numels[i])
    picks = sorted(set(order[:4] + order[len(order) // 2 - 2:len(order) // 2 + 2] + [i for i in order if L.numels[i] <= 40_000_000][-2:]))
    if workload == "flux_navg":
        job.step()
        torch.cuda.synchronize()
        step = 1 << 26
        for o in range(0, L.total_elems, step):
            sl = slice(o, min(o + step, L.total_elems))
            mean = torch.zeros(sl.stop - sl.start, dtype=torch.float64, device="cuda")
            for a in job.arenas:
                mean += a[sl].double()
            mean /= job.n_models
            got = job.out[sl].double()
            # one ulp of a bf16 value of that magnitude, at most; where the eight values cancel, the fp32 rounding of the running mean
            ulp = torch.maximum(mean.abs(), got.abs()) * 2.0 ** -7 + 1e-7
            covered = torch.zeros(sl.stop - sl.start, dtype=torch.bool, device="cuda")
            for off, n in zip(L.offsets, L.numels):
                lo, hi = max(off, sl.start), min(off + n, sl.stop)
                if lo < hi:
                    covered[lo - sl.start:hi - sl.start] = True
            assert bool((((got - mean).abs() <= ulp) | ~covered).all()), o
        for i in picks:
            o, n = L.offsets[i], L.numels[i]
            f32 = [oracle.to_f32(host_bits(a[o:o + n]), "bfloat16") for a in job.arenas]
            assert np.array_equal(host_bits(job.out[o:o + n]), oracle.from_f32(oracle.n_average(f32), "bfloat16")), (L.keys[i], n)
        return
    thr = torch.zeros((n_keys, nd), dtype=torch.float32, device="cuda")
    job.flags.zero_()
    st = lib.mecha_batched_add_difference_ties(
        job.arenas[0].data_ptr(), job.model_ptrs, nd, job.out.data_ptr(), job.segs_dev.data_ptr(), ctypes.addressof(job.segs_host), n_keys,
        L.total_elems, _lib.BF16, _lib.BF16, 0.2, 1.0, 0, thr.data_ptr(), job.ws.data_ptr(), job.ws.numel(), job.flags.data_ptr(),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "batched ties")
    torch.cuda.synchronize()
    assert int(job.flags.abs().sum().item()) == 0
    base = job.arenas[0]
    bad = []
    for i, (o, n) in enumerate(zip(L.offsets, L.numels)):                # ties.py:128-132: #(|d| < t) < kth <= #(|d| <= t)
        kth = max(int((1 - 0.2) * n), 1)
        b32 = base[o:o + n].float()
        for d in range(nd):
            dm = (job.arenas[1 + d][o:o + n].float() - b32).abs()
            t = thr[i, d]
            if not (int((dm < t).sum().item()) < kth <= int((dm <= t).sum().item())):
                bad.append((L.keys[i], d))
    assert not bad, bad[:5]
    for i in picks:
        o, n = L.offsets[i], L.numels[i]
        bits = [host_bits(a[o:o + n]) for a in job.arenas]
        b32 = oracle.to_f32(bits[0], "bfloat16")
        deltas = [oracle.subtract(oracle.to_f32(x, "bfloat16"), b32) for x in bits[1:]]
        want, det = oracle.ties_sum(deltas, k=0.2, details=True)
        assert np.array_equal(host_bits(thr[i]), det["thresholds"].view(np.uint32)), (L.keys[i], "thresholds")
        assert np.array_equal(host_bits(job.out[o:o + n]), oracle.from_f32(oracle.add_difference(b32, want, 1.0), "bfloat16")), (L.keys[i], n, "output")
