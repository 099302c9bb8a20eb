"""GPU: the batched (whole-arena) forms of the rest of north_star's streaming list -- slerp, the add_perpendicular graph,
add_difference(a, b - c), add_difference(a, delta) and DARE -- called through the C ABI against the CPU oracle on ragged arenas
(keys of 1 .. 1.6 M elements, random bits in the gaps), and through merge() against the reference's own sd_mecha.merge outputs
(tests/golden/e2e_plans.npz).  Bars: one-fma merges bit for bit; reductions and TIES sums <= 1 ulp of the output dtype
(fp64 sums here, thread-count-dependent fp32 cascades in torch); DARE thresholds and masks bit for bit."""
import ctypes
import pathlib
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, ulp_distance_half

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from test_gpu_parity import _arena_setup, dev, host_bits, host_f32  # noqa: E402
from test_scheduler_cpu import check_plan_result, plan_models, plan_recipes, toy_models, PLAN_EXACT  # noqa: E402,F401

pytestmark = pytest.mark.gpu

NUMELS = [5, 4096, 1, 70001, 320, 8, 123457, 4097, 2048, 2049, 1280 * 1280]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("dtype,out_dtype", [("bfloat16", "bfloat16"), ("float16", "float32")])
def test_batched_reductions_vs_oracle(oracle, dtype, out_dtype):
    from sd_mecha_amd import _lib
    L = _lib.lib()
    code = {"bfloat16": _lib.BF16, "float16": _lib.F16, "float32": _lib.F32}
    tdt = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}
    rng = np.random.default_rng(21)
    offs, total, bits, segs, segs_dev = _arena_setup(oracle, rng, NUMELS, 3, dtype)
    bits[0][offs[4]:offs[4] + 320] = 0                              # key 4: a = 0 -> the NaN branches (lerp / zeros)
    bits[2][offs[4]:offs[4] + 320] = 0                              #        (and c = 0 so that a - c = 0 for add_perpendicular)
    bits[1][offs[6] + 123456] = 0x7C00 if dtype == "float16" else 0x7F80   # +inf in the last element of key 6, model b
    arenas = [dev(b, dtype) for b in bits]
    ws = torch.empty(L.mecha_batched_reduce_workspace_bytes(len(NUMELS), total), dtype=torch.uint8, device="cuda")
    n = len(NUMELS)

    def run(fn, *args):
        out = torch.full((total,), float("nan"), dtype=tdt[out_dtype], device="cuda")
        flags = torch.zeros(n, dtype=torch.int32, device="cuda")
        for _ in range(2):                                          # the second call reuses the workspace as it is
            flags.zero_()
            _lib.check(fn(*[a.data_ptr() for a in args[0]], out.data_ptr(), segs_dev.data_ptr(), ctypes.addressof(segs), n, total,
                          code[dtype], code[out_dtype], *args[1], ws.data_ptr(), ws.numel(), flags.data_ptr(), _stream()), fn.__name__)
        torch.cuda.synchronize()
        return out, flags.cpu().numpy()

    cases = {
        "slerp": run(L.mecha_batched_slerp, arenas[:2], (0.3,)),
        "perp": run(L.mecha_batched_perpendicular_component, arenas[:2], ()),
        "addperp": run(L.mecha_batched_add_perpendicular, arenas, (0.6,)),
    }
    for name, (out, fl) in cases.items():
        assert [int(x) & 1 for x in fl] == [int(s == 6) for s in range(n)], (name, fl)
        for s, (o, m) in enumerate(zip(offs, NUMELS)):
            if s == 6:
                continue
            f = [oracle.to_f32(b[o:o + m], dtype) for b in bits]
            want = {"slerp": lambda: oracle.slerp(f[0], f[1], 0.3), "perp": lambda: oracle.perpendicular_component(f[0], f[1]),
                    "addperp": lambda: oracle.add_perpendicular(f[0], f[1], f[2], 0.6)}[name]()
            got = out[o:o + m]
            mag = np.abs(f[0]) + np.abs(f[1]) + (np.abs(f[2]) if name == "addperp" else 0)
            if out_dtype == "float32":                              # fp32 out: a few fp32 ulps of the result, or of the operands where it cancels
                assert (np.abs(host_f32(got) - want) <= 4e-6 * np.abs(want) + 1e-6 * mag).all(), (name, s)
                continue
            d = ulp_distance_half(host_bits(got), oracle.from_f32(want, out_dtype))
            noise = np.abs(host_f32(got) - want) <= 1e-6 * mag      # where the formula cancels, fp32 operand noise is the floor
            assert ((d <= 1) | noise).all() and (m < 64 or (d > 0).mean() <= 0.02), (name, s, int(d.max()))
        # nothing outside the keys was written
        mask = np.ones(total, dtype=bool)
        for o, m in zip(offs, NUMELS):
            mask[o:o + m] = False
        assert torch.isnan(out[torch.from_numpy(mask).cuda()].float()).all(), name


def test_batched_add_difference_forms_bit_exact(oracle):
    from sd_mecha_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(22)
    offs, total, bits, segs, segs_dev = _arena_setup(oracle, rng, NUMELS, 3)
    bits[2][offs[3] + 5] = 0xFF80                                   # -inf in model c, key 3
    arenas = [dev(b, "bfloat16") for b in bits]
    n = len(NUMELS)
    out = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    flags = torch.zeros(n, dtype=torch.int32, device="cuda")
    _lib.check(L.mecha_batched_add_difference_sub(arenas[0].data_ptr(), arenas[1].data_ptr(), arenas[2].data_ptr(), out.data_ptr(),
                                                  segs_dev.data_ptr(), n, total, _lib.BF16, _lib.BF16, 0.6, flags.data_ptr(), _stream()), "addsub")
    assert [int(x) & 1 for x in flags.cpu().numpy()] == [int(s == 3) for s in range(n)]
    ob = host_bits(out)
    for s, (o, m) in enumerate(zip(offs, NUMELS)):
        if s == 3:
            continue
        f = [oracle.to_f32(b[o:o + m], "bfloat16") for b in bits]
        want = oracle.add_difference(f[0], oracle.subtract(f[1], f[2]), 0.6)
        assert np.array_equal(ob[o:o + m], oracle.from_f32(want, "bfloat16")), s
    flags.zero_()
    _lib.check(L.mecha_batched_add_difference(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(), segs_dev.data_ptr(), n, total,
                                              _lib.BF16, _lib.BF16, 0.25, flags.data_ptr(), _stream()), "adddiff")
    assert int(flags.sum().item()) == 0
    ob = host_bits(out)
    for o, m in zip(offs, NUMELS):
        f = [oracle.to_f32(b[o:o + m], "bfloat16") for b in bits[:2]]
        assert np.array_equal(ob[o:o + m], oracle.from_f32(oracle.add_difference(f[0], f[1], 0.25), "bfloat16"))
    # the per-key form and its refusal of alpha == 0 (the reference returns `a` without loading b or c)
    o, m = offs[6], NUMELS[6]
    one = torch.zeros(m, dtype=torch.bfloat16, device="cuda")
    _lib.check(L.mecha_add_difference_sub(arenas[0][o:].data_ptr(), arenas[1][o:].data_ptr(), arenas[2][o:].data_ptr(), one.data_ptr(), m,
                                          _lib.BF16, _lib.BF16, 0.6, None, _stream()), "addsub per key")
    f = [oracle.to_f32(b[o:o + m], "bfloat16") for b in bits]
    assert np.array_equal(host_bits(one), oracle.from_f32(oracle.add_difference(f[0], oracle.subtract(f[1], f[2]), 0.6), "bfloat16"))
    assert L.mecha_add_difference_sub(arenas[0].data_ptr(), arenas[1].data_ptr(), arenas[2].data_ptr(), one.data_ptr(), m, _lib.BF16, _lib.BF16,
                                      0.0, None, _stream()) < 0


@pytest.mark.parametrize("n_models,p,k,rescale,vote,scale", [(3, 0.9, 0.2, True, False, None), (2, 0.4, 0.5, False, False, None),
                                                              (7, 0.5, 1.0, True, False, None), (3, 0.3, 0.35, True, True, None),
                                                              (3, 0.9, 0.2, True, False, 1e-28), (3, 0.5, 0.6, True, False, 2e-37),
                                                              (2, 0.9, 1.0, True, False, 3e20), (3, 0.7, 0.3, False, False, "mixed")])
def test_batched_dare_vs_oracle(oracle, n_models, p, k, rescale, vote, scale):
    """Every key's masks are cut from ONE Bernoulli stream (the reference seeds a fresh generator per key, ties.py:29-31):
    thresholds of the masked deltas bit for bit, outputs bit for bit (vote_sgn: <= 1 ulp, the elected sign of an fp32 sum).
    `scale`: deltas far below 2^-57 (down into bf16's denormals), far above 2^59, or all of these beside ordinary ones inside
    one key -- the sums for which the full-tile kernel leaves its three-instruction divisions for IEEE ones."""
    from sd_mecha_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(23)
    numels = [5, 4096, 1, 70001, 320, 8, 123457, 4097, 640 * 640]
    offs, total, bits, segs, segs_dev = _arena_setup(oracle, rng, numels, n_models + 1)
    if scale is not None:                                           # base 0: the models ARE the deltas
        bits[0] = np.zeros_like(bits[0])
        for i in range(1, n_models + 1):
            f = oracle.to_f32(bits[i], "bfloat16") * np.float32(50.0)
            factor = (np.float32(scale) if scale != "mixed"
                      else rng.choice(np.array([1e-28, 2e-37, 1.0, 3e20], dtype=np.float32), size=f.size))
            bits[i] = oracle.from_f32((f * factor).astype(np.float32), "bfloat16")
    arenas = [dev(b, "bfloat16") for b in bits]
    n = len(numels)
    seed = 114514
    n_bits = n_models * max(numels)
    host = torch.empty((n_bits + 31) // 32 + 1, dtype=torch.int32)
    _lib.check(L.mecha_dare_draw_stream_host(seed, p, n_bits, host.data_ptr()), "draw")
    stream_dev = host.cuda()
    out = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    flags = torch.zeros(n, dtype=torch.int32, device="cuda")
    thr = torch.zeros((n, n_models), dtype=torch.float32, device="cuda")
    ws = torch.empty(L.mecha_batched_ties_workspace_bytes(n_models, n, total), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        flags.zero_()
        _lib.check(L.mecha_batched_add_difference_dare(arenas[0].data_ptr(), _lib.ptr_array([a.data_ptr() for a in arenas[1:]]), n_models,
                                                       out.data_ptr(), segs_dev.data_ptr(), ctypes.addressof(segs), n, total, _lib.BF16, _lib.BF16,
                                                       k, 0.8, int(vote), p, int(rescale), stream_dev.data_ptr(), n_bits, 0, 0.0, None, None, thr.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), flags.data_ptr(), _stream()), "batched dare")
        torch.cuda.synchronize()
    assert int(flags.sum().item()) == 0
    ob, th = host_bits(out), host_bits(thr)
    for s, (o, m) in enumerate(zip(offs, numels)):
        base = oracle.to_f32(bits[0][o:o + m], "bfloat16")
        deltas = [oracle.subtract(oracle.to_f32(b[o:o + m], "bfloat16"), base) for b in bits[1:]]
        masks = oracle.dare_masks(seed, p, n_models, (m,))
        masked = [d * mk.astype(np.float32) for d, mk in zip(deltas, masks)]
        _, det = oracle.ties_sum(masked, k=k, vote_sgn=vote, details=True)
        assert np.array_equal(th[s], det["thresholds"].view(np.uint32)), (s, "thresholds")
        ties = oracle.ties_sum_with_dropout(deltas, probability=p, rescale=rescale, k=k, vote_sgn=vote, apply_stock=True, masks=masks)
        want = oracle.from_f32(oracle.add_difference(base, ties, 0.8), "bfloat16")
        if vote:
            assert ulp_distance_half(ob[o:o + m], want).max() <= 1, s
        else:
            assert np.array_equal(ob[o:o + m], want), (s, "output")
    # a stream that is too short for the largest key is refused on the host
    assert L.mecha_batched_add_difference_dare(arenas[0].data_ptr(), _lib.ptr_array([a.data_ptr() for a in arenas[1:]]), n_models, out.data_ptr(),
                                               segs_dev.data_ptr(), ctypes.addressof(segs), n, total, _lib.BF16, _lib.BF16, k, 0.8, 0, p, 1,
                                               stream_dev.data_ptr(), n_bits - 1, 0, 0.0, None, None, None, ws.data_ptr(), ws.numel(), flags.data_ptr(), _stream()) < 0


@pytest.mark.parametrize("name", ["slerp", "slerp_zero", "slerp_nonfinite", "addperp", "addperp_zero", "addsub", "adddiff", "dare_nostock",
                                  "dare_nostock_norescale", "dare_default_seed", "dare_stock"])
@pytest.mark.parametrize("buffer", [1 << 31, 1 << 16])
def test_round3_plans_reproduce_reference_gpu(plan_models, name, buffer, tmp_path):   # noqa: F811
    """merge() on the HIP backend against the reference's own outputs; every key goes through the batched plan."""
    import safetensors.torch
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    recipe, kind = plan_recipes(amd, plan_models)[name]
    plan = merging.fuse(recipe)
    assert plan is not None and plan.kind == kind
    kw = dict(merge_device="cuda", output_dtype=torch.bfloat16, total_buffer_size=buffer)
    n_keys = 3 if name == "dare_stock" else 4
    res = amd.merge(recipe, **kw)
    check_plan_result(res, name, 0 if name in PLAN_EXACT else 1)
    assert merging.last_stats["generic_keys"] == 0 and merging.last_stats["fused_keys"] == n_keys
    out = tmp_path / "o.safetensors"
    amd.merge(recipe, output=out, **kw)
    check_plan_result(safetensors.torch.load_file(str(out)), name, 0 if name in PLAN_EXACT else 1)


def test_full_size_sdxl_plans_properties():
    """BASELINE-sized: every non-empty float key of two / three SDXL-shaped checkpoints in ONE arena per model through the
    batched slerp and add_perpendicular entry points (3.47 G elements per model); checked through properties that need no
    oracle at this size, on every key: |slerp(a, b, t)| = lerp(|a|, |b|, t) per key (linear.py:58: the result is rescaled to
    that norm) and perp(a - c, b - c) . (a - c) = 0 per key (linear.py:110), both by fp64 reductions on the device; sampled
    keys against the oracle."""
    from sd_mecha_amd import _lib
    from sd_mecha_amd.arena import plan_arena
    from sd_mecha_amd.shapes import load_manifest, numel
    from oracle import oracle as O
    O.build()
    L = _lib.lib()
    items = [(k, numel(s)) for k, s, dt in load_manifest("sdxl-sgm") if numel(s) > 0 and dt in ("float16", "bfloat16", "float32")]
    lay = plan_arena(items)
    total = lay.total_elems
    g = torch.Generator(device="cuda").manual_seed(3)
    arenas = []
    for _ in range(3):
        t = torch.empty(total, dtype=torch.bfloat16, device="cuda")
        step = 1 << 28
        for i in range(0, total, step):
            m = min(step, total - i)
            t[i:i + m] = (torch.randn(m, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        arenas.append(t)
    segs = lay.host_segments()
    segs_dev = lay.device_segments("cuda")
    n = len(lay.keys)
    ws = torch.empty(L.mecha_batched_reduce_workspace_bytes(n, total), dtype=torch.uint8, device="cuda")
    flags = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = torch.empty(total, dtype=torch.float32, device="cuda")
    offs = torch.tensor(lay.offsets, device="cuda")
    nums = torch.tensor(lay.numels, device="cuda")

    def per_key_sum(*xs):                                           # fp64 sum of the product of the arena-sized tensors, per key
        acc = torch.empty(n, dtype=torch.float64, device="cuda")
        for s_, (o, m) in enumerate(zip(lay.offsets, lay.numels)):   # (index_add_ into 2 531 slots serialises on its atomics)
            v = xs[0][o:o + m].double()
            for x in xs[1:]:
                v = v * x[o:o + m].double()
            acc[s_] = v.sum()
        return acc

    # ---- slerp
    _lib.check(L.mecha_batched_slerp(arenas[0].data_ptr(), arenas[1].data_ptr(), out.data_ptr(), segs_dev.data_ptr(), ctypes.addressof(segs), n,
                                     total, _lib.BF16, _lib.F32, 0.3, ws.data_ptr(), ws.numel(), flags.data_ptr(), _stream()), "slerp")
    torch.cuda.synchronize()
    assert int(flags.sum().item()) == 0
    na = per_key_sum(arenas[0], arenas[0]).sqrt()
    nb = per_key_sum(arenas[1], arenas[1]).sqrt()
    no = per_key_sum(out, out).sqrt()
    want = na + 0.3 * (nb - na)
    big = nums >= 64                                                # tiny keys (n = 1 .. 4 scalars) can be anti-parallel: omega ~ pi is ill-conditioned
    rel = ((no - want).abs() / want)[big]
    assert float(rel.max()) < 2e-5, float(rel.max())
    rng = np.random.default_rng(0)
    for s in [0, n // 3, n - 1] + list(rng.integers(0, n, size=5)):
        o, m = lay.offsets[s], lay.numels[s]
        if m > 4_000_000:
            continue
        a, b = (O.to_f32(host_bits(t[o:o + m]), "bfloat16") for t in arenas[:2])
        assert np.allclose(host_f32(out[o:o + m]), O.slerp(a, b, 0.3), rtol=1e-5, atol=1e-8), s
    # ---- add_perpendicular (alpha = 1: out - a = perp)
    flags.zero_()
    _lib.check(L.mecha_batched_add_perpendicular(arenas[0].data_ptr(), arenas[1].data_ptr(), arenas[2].data_ptr(), out.data_ptr(), segs_dev.data_ptr(),
                                                 ctypes.addressof(segs), n, total, _lib.BF16, _lib.F32, 1.0, ws.data_ptr(), ws.numel(),
                                                 flags.data_ptr(), _stream()), "addperp")
    torch.cuda.synchronize()
    assert int(flags.sum().item()) == 0
    x = arenas[0].float() - arenas[2].float()
    perp = out - arenas[0].float()
    dot = per_key_sum(perp, x)
    nx2 = per_key_sum(x, x)
    np2 = per_key_sum(perp, perp)
    cosine = (dot.abs() / (nx2.sqrt() * np2.sqrt()).clamp_(min=1e-30))[big]
    assert float(cosine.max()) < 1e-4, float(cosine.max())         # fp32 rounding of a + perp, then perp recovered by subtraction
    for s in [1, n // 2] + list(rng.integers(0, n, size=4)):
        o, m = lay.offsets[s], lay.numels[s]
        if m > 4_000_000:
            continue
        a, b, c = (O.to_f32(host_bits(t[o:o + m]), "bfloat16") for t in arenas)
        assert np.allclose(host_f32(out[o:o + m]), O.add_perpendicular(a, b, c, 1.0), rtol=1e-5, atol=1e-8), s
