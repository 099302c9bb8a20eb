"""GPU parity of the MFMA LoRA reconstruction (sd_mecha_amd/csrc/lora.hip) against the reference's own
LoraAlgorithm.compose / compose_alpha outputs (tests/golden/lora.npz) and against the oracle at larger sizes.
Tolerance: the reference's fp32 sgemm order is BLAS-defined, so deltas are compared at fp32 level (rtol 1e-5 on an
r-term dot product) and merged keys at <= 1 ulp of the output dtype."""
import numpy as np
import pytest
import torch

from conftest import load_golden, ulp_distance_half

pytestmark = pytest.mark.gpu

TDT = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def dev16(bits, dtype):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).view(TDT[dtype]).cuda()


def bits16(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def test_lora_golden():
    import sd_mecha_amd as amd
    g = load_golden("lora")
    for c in range(int(g["n_cases"])):
        pre = f"l{c}"
        up, down = dev16(g[pre + "_up"], "float16"), dev16(g[pre + "_down"], "float16")
        ts = tuple(int(x) for x in g[pre + "_target_shape"])
        want = np.array(g[pre + "_delta_alpha4"], dtype=np.uint32).view(np.float32).reshape(ts)
        got = amd.lora_delta(up, down, torch.tensor(4.0), ts).cpu().numpy()
        assert got.shape == ts and np.allclose(got, want, rtol=1e-5, atol=1e-9)
        want = np.array(g[pre + "_delta_noalpha"], dtype=np.uint32).view(np.float32).reshape(ts)
        got = amd.lora_delta(up, down, None, ts).cpu().numpy()
        assert np.allclose(got, want, rtol=1e-5, atol=1e-9)
        base = dev16(g[pre + "_base"], "bfloat16").reshape(ts)
        merged = amd.lora_merge(base, [(up, down, 4.0, 0.75)], out_dtype=torch.bfloat16)
        d = ulp_distance_half(bits16(merged).reshape(-1), g[pre + "_merged_0.75_bf16"].reshape(-1))
        assert d.max() <= 1 and (d > 0).mean() < 0.002


@pytest.mark.parametrize("ldt", ["float16", "bfloat16"])
def test_lora_many_networks_vs_oracle(oracle, ldt):
    """16 networks of mixed rank on one key, ragged sizes (not multiples of the 64x64 tile, rank not a multiple of 8); the last shape
    takes the fast kernel with partial tiles on both edges (72 rows, 72 columns: one 16-byte piece of a wave's row inside the key)."""
    import sd_mecha_amd as amd
    rng = np.random.default_rng(21)
    for rows, cols, ranks in [(640, 320 * 9, [32] * 16), (77, 131, [4, 8, 5, 16, 1]), (1280, 1280, [128, 64]), (200, 1096, [32, 64, 32])]:
        base_b = oracle.from_f32(rng.normal(0, 0.02, size=(rows, cols)).astype(np.float32), "bfloat16")
        acc = oracle.to_f32(base_b, "bfloat16")
        mag = np.abs(acc)
        nets = []
        for i, r in enumerate(ranks):
            up_b = oracle.from_f32(rng.normal(0, 0.05, size=(rows, r)).astype(np.float32), ldt)
            down_b = oracle.from_f32(rng.normal(0, 0.05, size=(r, cols)).astype(np.float32), ldt)
            alpha, w = float(r) / 2, 1.0 if i % 2 == 0 else 0.6
            delta = oracle.lora_delta(oracle.to_f32(up_b, ldt), oracle.to_f32(down_b, ldt), alpha)
            acc = oracle.add_difference(acc, delta, w)
            mag = mag + np.abs(delta)
            nets.append((dev16(up_b, ldt), dev16(down_b, ldt), alpha, w))
        got = amd.lora_merge(dev16(base_b, "bfloat16").reshape(rows, cols), nets, out_dtype=torch.bfloat16)
        d = ulp_distance_half(bits16(got), oracle.from_f32(acc, "bfloat16"))
        # <= 1 ulp of the output dtype; where base and deltas cancel (|result| << |terms|) the floor is the fp32
        # rounding noise of the terms, not an ulp of the tiny result
        noise = np.abs(got.float().cpu().numpy() - acc) <= 1e-6 * mag
        assert ((d <= 1) | noise).all() and (d > 0).mean() < 0.005, (rows, cols, d.max(), (d > 0).mean())
        # fp32 output, looser check of the accumulated value itself
        got32 = amd.lora_merge(dev16(base_b, "bfloat16").reshape(rows, cols), nets).cpu().numpy()
        assert np.allclose(got32, acc, rtol=2e-5, atol=1e-7)


def test_lora_batched_matches_per_key():
    """Every key of a batch in one launch (mecha_batched_lora_accumulate) gives, bit for bit, what the per-key entry
    point gives; gaps between keys stay untouched; a non-finite factor flags its own key only."""
    import sd_mecha_amd as amd
    from sd_mecha_amd import merge_methods as mm
    from sd_mecha_amd.arena import plan_arena
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(320, 320), (77, 131), (64, 32, 3, 3), (1280, 640), (5, 9), (640, 2560)]
    layout = plan_arena([(f"k{i}", int(np.prod(s))) for i, s in enumerate(shapes)])
    base = (torch.randn(layout.total_elems, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    out = torch.full((layout.total_elems,), 7.0, dtype=torch.bfloat16, device="cuda")
    entries, per_key = [], []
    for i, (s, off) in enumerate(zip(shapes, layout.offsets)):
        rows, cols = s[0], int(np.prod(s)) // s[0]
        nets = []
        for r in ([32, 32, 8], [4], [16, 5], [32] * 16, [3], [64, 32])[i]:
            up = (torch.randn(rows, r, device="cuda", generator=g) * 0.05).half()
            down = (torch.randn(r, cols, device="cuda", generator=g) * 0.05).half()
            nets.append((up, down, float(r) / 2, 0.0 if (i == 3 and len(nets) == 5) else 0.8))
        if i == 4:
            nets[0][0][2, 1] = float("inf")
        entries.append((off, rows, cols, nets))
        per_key.append(None if i == 4 else amd.lora_merge(base[off:off + rows * cols].view(rows, cols), nets, out_dtype=torch.bfloat16))
    flags = torch.zeros(len(shapes), dtype=torch.int32, device="cuda")
    batch = mm.LoraBatch(entries, torch.device("cuda"))
    for _ in range(2):
        flags.zero_()
        batch.run(base, out, flags)
    torch.cuda.synchronize()
    assert [int(f) & 1 for f in flags.cpu()] == [0, 0, 0, 0, 1, 0]
    covered = torch.zeros(layout.total_elems, dtype=torch.bool, device="cuda")
    for i, (s, off) in enumerate(zip(shapes, layout.offsets)):
        n = int(np.prod(s))
        covered[off:off + n] = True
        if per_key[i] is not None:
            assert torch.equal(out[off:off + n], per_key[i].reshape(-1)), i
    assert (out[~covered] == 7.0).all()


def test_lora_conv_shapes_and_validation():
    import sd_mecha_amd as amd
    up = torch.randn(32, 8, 1, 1, device="cuda").half()
    down = torch.randn(8, 16, 3, 3, device="cuda").half()
    base = torch.randn(32, 16, 3, 3, device="cuda").to(torch.bfloat16)
    got = amd.lora_merge(base, [(up, down, 4.0, 1.0)], out_dtype=torch.float32)
    want = base.float() + (up.float().view(32, 8) @ down.float().view(8, -1)).view(32, 16, 3, 3) * (4.0 / 8)
    assert got.shape == base.shape and torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert torch.equal(amd.lora_merge(base, [(up, down, 4.0, 0.0)], out_dtype=torch.bfloat16), base)     # weight 0: base untouched
    with pytest.raises(NotImplementedError):
        amd.lora_merge(base, [(up.float(), down.float(), 4.0, 1.0)])
    with pytest.raises(ValueError):
        amd.lora_merge(base, [(up, down[:4], 4.0, 1.0)])
    bad = up.clone()
    bad[3, 2] = float("nan")
    with pytest.raises(amd.NonFiniteStateDictKeyError):
        amd.lora_merge(base, [(bad, down, 4.0, 1.0)], key="k")


@pytest.mark.parametrize("ldt", ["float16", "bfloat16"])
def test_lora_fast_kernel_edges(oracle, ldt):
    """The LDS-DMA kernel (every rank a multiple of 32, cols % 8 == 0): partial tiles in both directions (clamped loads),
    ranks that differ between networks, a scale that is not a power of two (two-step fold), fp32 / fp16 base and output,
    and the finite checks of factors on edge tiles."""
    import sd_mecha_amd as amd
    rng = np.random.default_rng(5)
    for rows, cols, ranks, bdt in [(200, 136, [32, 64, 32], "bfloat16"), (129, 8, [32], "float16"), (72, 1000, [96, 32], "bfloat16"),
                                   (384, 264, [32] * 5, "float16")]:
        base_b = oracle.from_f32(rng.normal(0, 0.02, size=(rows, cols)).astype(np.float32), bdt)
        acc = oracle.to_f32(base_b, bdt)
        mag = np.abs(acc)
        nets = []
        for i, r in enumerate(ranks):
            up_b = oracle.from_f32(rng.normal(0, 0.05, size=(rows, r)).astype(np.float32), ldt)
            down_b = oracle.from_f32(rng.normal(0, 0.05, size=(r, cols)).astype(np.float32), ldt)
            alpha, w = (float(r) / 2, 0.6) if i % 2 == 0 else (7.0, 1.0)          # scale 0.5, or 7 / r (not a power of two)
            delta = oracle.lora_delta(oracle.to_f32(up_b, ldt), oracle.to_f32(down_b, ldt), alpha)
            acc = oracle.add_difference(acc, delta, w)
            mag = mag + np.abs(delta)
            nets.append((dev16(up_b, ldt), dev16(down_b, ldt), alpha, w))
        tbase = dev16(base_b, bdt).reshape(rows, cols)
        got = amd.lora_merge(tbase, nets, out_dtype=TDT[bdt])
        d = ulp_distance_half(bits16(got), oracle.from_f32(acc, bdt)) if bdt == "bfloat16" else None
        if d is not None:
            noise = np.abs(got.float().cpu().numpy() - acc) <= 1e-6 * mag
            assert ((d <= 1) | noise).all() and (d > 0).mean() < 0.005, (rows, cols, d.max(), (d > 0).mean())
        else:
            assert np.allclose(got.float().cpu().numpy(), acc, rtol=2e-3, atol=1e-6)
        got32 = amd.lora_merge(tbase.float(), nets).cpu().numpy()                # fp32 base and output
        assert np.allclose(got32, acc, rtol=2e-5, atol=1e-7)
        # a non-finite factor anywhere flags the key; the last row / last column sit in partial tiles
        for which, (i, j) in (("up", (rows - 1, ranks[0] - 1)), ("down", (ranks[0] - 1, cols - 1)), ("up", (0, 0)), ("down", (0, 0))):
            bad = [list(nw) for nw in nets]
            tgt = bad[0][0 if which == "up" else 1].clone()
            tgt[i, j] = float("inf")
            bad[0][0 if which == "up" else 1] = tgt
            with pytest.raises(amd.NonFiniteStateDictKeyError):
                amd.lora_merge(tbase, [tuple(nw) for nw in bad], out_dtype=TDT[bdt], key="k")


def test_lora_more_chunks_than_the_fast_table_holds():
    """The fast kernel keeps a key's k-chunk sequence in a table of 128 entries (csrc/lora.hip MAX_CHUNKS): a key whose networks
    add up to more 32-deep chunks takes the general kernel, one chunk fewer the fast one; both against fp64 on the host."""
    import sd_mecha_amd as amd
    g = torch.Generator().manual_seed(11)
    rows, cols = 160, 136
    base = (torch.randn(rows, cols, generator=g) * 0.02).to(torch.bfloat16)
    for ranks in ([2048, 2048, 32], [2048, 2016, 32]):                   # 129 chunks, 128 chunks
        nets, want = [], base.double()
        for r in ranks:
            up = (torch.randn(rows, r, generator=g) * 0.05).half()
            down = (torch.randn(r, cols, generator=g) * 0.05).half()
            nets.append((up.cuda(), down.cuda(), float(r) / 4, 0.5))
            want = want + (up.double() @ down.double()) * 0.25 * 0.5
        got = amd.lora_merge(base.cuda(), nets, out_dtype=torch.float32).cpu().double()
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (ranks, (got - want).abs().max())


def test_full_config5_job_every_key_by_linearity():
    """BASELINE config 5 at FULL size: an sdxl-kohya-shaped bf16 checkpoint + 16 rank-32 fp16 kohya LoRA files through merge()
    (files on tmpfs -> descriptor tables -> one batched launch per wave -> one fp32 file).  Every key the networks cover is
    checked by what the method is -- out - base = sum_i w_i (alpha_i / r_i) up_i . down_i (lycoris.py:226-263, linear.py:76) --
    with the products taken in fp64 on the device; every other key must be the base, bit for bit."""
    import pathlib
    import shutil
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tools"))
    import bench_e2e
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    from sd_mecha_amd.lora import lora_key_prefix
    from sd_mecha_amd.shapes import load_manifest
    from sd_mecha_amd.streaming import InSafetensorsDict
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30 or shutil.disk_usage("/dev/shm").free < 40 << 30:
        pytest.skip("needs 40 GB of free HBM and of /dev/shm")
    manifest = [(k, s, d) for k, s, d in load_manifest("sdxl-kohya") if not d.startswith("int")]
    work = pathlib.Path("/dev/shm/mecha_test_config5")
    work.mkdir(parents=True, exist_ok=True)
    try:
        dev = torch.device("cuda", 0)
        pb, po = work / "base.safetensors", work / "out.safetensors"
        bench_e2e.write_checkpoint(pb, manifest, 100, dev)
        loras = []
        for i in range(bench_e2e.N_LORAS):
            bench_e2e.write_lora(work / f"lora{i}.safetensors", manifest, 500 + i, dev)
            loras.append(work / f"lora{i}.safetensors")
        base_m = amd.model(str(pb))
        recipe = base_m
        for p in loras:
            recipe = amd.add_difference(recipe, amd.convert(amd.model(str(p)), base_m), alpha=1.0) | recipe
        amd.merge(recipe, output=str(po), output_dtype=torch.float32, total_buffer_size=8 << 30)
        assert merging.last_stats["generic_keys"] <= 3, merging.last_stats
        base, out = InSafetensorsDict(pb), InSafetensorsDict(po)
        nets = [InSafetensorsDict(p) for p in loras]
        assert set(out.keys()) == {k for k, _, _ in manifest}
        covered, worst = 0, 0.0
        for k, s, _ in manifest:
            prefix = lora_key_prefix(k, s)
            b = base[k].to(dev)
            o = out[k].to(dev)
            if prefix is None or f"{prefix}.lora_up.weight" not in nets[0]:
                assert torch.equal(o, b.float()), k
                continue
            covered += 1
            rows = s[0]
            want = b.double().reshape(rows, -1)
            mag = want.abs()
            for f in nets:
                up, down = f[f"{prefix}.lora_up.weight"].to(dev).double(), f[f"{prefix}.lora_down.weight"].to(dev).double().reshape(bench_e2e.LORA_RANK, -1)
                delta = (up @ down) * (16.0 / bench_e2e.LORA_RANK)
                want = want + delta
                mag = mag + delta.abs()
            err = ((o.double().reshape(rows, -1) - want).abs() / mag.clamp_min(1e-30)).max()
            worst = max(worst, float(err))
            assert err <= 4e-6, (k, s, float(err))                       # fp32 rounding of 17 terms of the size of `mag`
        assert covered > 700, covered
        print(f"config 5 full size: {covered} covered keys, worst |out - want| / (|base| + sum |delta|) = {worst:.2e}")
    finally:
        shutil.rmtree(work, ignore_errors=True)
