"""GPU: sd_mecha_amd.merge() end to end on the HIP backend -- mmap -> pinned staging -> H2D -> batched kernels -> D2H ->
safetensors -- against the reference's own sd_mecha.merge() outputs (tests/golden/e2e.npz)."""
import pathlib
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from test_scheduler_cpu import SHAPES, check_against_reference, toy_models  # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("buffer", [1 << 31, 1 << 18, 1 << 16])
def test_merge_matches_reference_on_gpu(toy_models, tmp_path, buffer):       # noqa: F811
    import safetensors.torch
    import sd_mecha_amd as amd
    m = [amd.model(p) for p in toy_models]
    kw = dict(merge_device="cuda", output_dtype=torch.bfloat16, total_buffer_size=buffer)
    check_against_reference(amd.merge(amd.weighted_sum(m[0], m[1], alpha=0.5), **kw), "wsum")
    check_against_reference(amd.merge(amd.weighted_sum(m[0], m[1], alpha=0.3), **{**kw, "output_dtype": torch.float16}), "wsum_f16")
    check_against_reference(amd.merge(amd.n_average(m[0], m[1], m[2]), **kw), "navg3")
    check_against_reference(amd.merge(amd.n_average(*m), **kw), "navg4")
    deltas = [amd.subtract(x, m[0]) for x in m[1:3]]
    check_against_reference(amd.merge(amd.add_difference(m[0], amd.ties_sum(*deltas, k=0.2), alpha=1.0), **kw), "ties", exact=False)
    deltas4 = [amd.subtract(x, m[0]) for x in m[1:4]]
    check_against_reference(amd.merge(amd.add_difference(m[0], amd.ties_sum(*deltas4, k=0.2), alpha=1.0) | m[0], **kw),
                            "ties4_fallback", exact=False)
    out = tmp_path / f"merged_{buffer}.safetensors"
    amd.merge(amd.n_average(*m), output=out, **kw)
    check_against_reference(safetensors.torch.load_file(str(out)), "navg4")


def test_truncate_rank_in_a_recipe(toy_models, oracle):                      # noqa: F811
    """truncate_rank (svd.py:89-143) inside merge(): a + truncate_rank(b - a, rank) key by key (the per-key executor), against the
    oracle's exact-mode result computed from the same bf16 inputs; keys of <= 1 dimension pass through (svd.py:99-100)."""
    import safetensors.torch
    import sd_mecha_amd as amd
    m = [amd.model(p) for p in toy_models[:2]]
    recipe = amd.add_difference(m[0], amd.truncate_rank(amd.subtract(m[1], m[0]), rank=3, use_approximate_basis=False), alpha=1.0)
    res = amd.merge(recipe, merge_device="cuda", output_dtype=torch.float32)
    a, b = (safetensors.torch.load_file(str(p)) for p in toy_models[:2])
    for k in SHAPES:
        af, bf = a[k].float().numpy(), b[k].float().numpy()
        want = (af + oracle.truncate_rank(bf - af, 3)).reshape(af.shape)
        got = res[k].cpu().numpy()
        assert got.shape == want.shape, k
        if want.size:
            assert np.linalg.norm(got.astype(np.float64) - want) <= 1e-3 * max(np.linalg.norm(want), 1e-30), k


def test_generic_path_runs_method_bodies_per_key(toy_models):               # noqa: F811
    """Recipes the fuser does not cover (here: a graph of primitives and a user-defined @merge_method) are evaluated key
    by key with the same bodies on device tensors; a graph of primitives must equal the fused pipeline bit for bit."""
    import sd_mecha_amd as amd
    m = [amd.model(p) for p in toy_models]
    kw = dict(merge_device="cuda", output_dtype=torch.bfloat16)
    from sd_mecha_amd.merging import fuse
    # same TIES graph, but with k passed so that the fuser still matches vs. an equivalent unfusable spelling
    fused = amd.merge(amd.add_difference(m[0], amd.ties_sum(amd.subtract(m[1], m[0]), amd.subtract(m[2], m[0]), k=0.2), alpha=1.0), **kw)
    generic_recipe = amd.add_difference(m[0], amd.scale(amd.ties_sum(amd.subtract(m[1], m[0]), amd.subtract(m[2], m[0]), k=0.2), 1.0), alpha=1.0)
    assert fuse(generic_recipe) is None
    generic = amd.merge(generic_recipe, **kw)
    for k in SHAPES:
        assert torch.equal(fused[k], generic[k]), k

    @amd.merge_method(identifier="test_double_minus", register=False)
    def double_minus(a, b, **kwargs):
        assert kwargs["key"] in SHAPES and a.is_cuda and a.dtype == torch.float32
        return 2 * a - b

    res = amd.merge(double_minus(m[0], m[1]), **kw)
    import safetensors.torch
    a, b = (safetensors.torch.load_file(str(p)) for p in toy_models[:2])
    for k in SHAPES:
        assert torch.equal(res[k], (2 * a[k].float() - b[k].float()).to(torch.bfloat16)), k
    # slerp through merge() (a batched plan since round 3) vs the per-key call: the same fp64 sums in another order
    from conftest import ulp_distance_half
    res = amd.merge(amd.slerp(m[0], m[1], alpha=0.3), **kw)
    want = amd.slerp(a["w.linear"].cuda(), b["w.linear"].cuda(), 0.3, out_dtype=torch.bfloat16).cpu()
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1)      # noqa: E731
    assert ulp_distance_half(bits(res["w.linear"]), bits(want)).max() <= 1


def test_pipeline_many_waves_medium_checkpoint(tmp_path):
    """A 40 M-parameter synthetic checkpoint set through the double-buffered pipeline with a 32 MB staging budget:
    output file must equal the single-wave result and, on sampled keys, the oracle."""
    import safetensors.torch
    import sd_mecha_amd as amd
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    shapes = {f"blk{i}.w": (512, 640 + 8 * i) for i in range(100)}
    shapes.update({f"blk{i}.b": (640 + i,) for i in range(100)})
    shapes["emb"] = (9000, 768)
    paths = []
    for mi in range(3):
        state = {k: torch.from_numpy(rng.normal(0, 0.02, size=s).astype(np.float32)).to(torch.bfloat16) for k, s in shapes.items()}
        p = tmp_path / f"m{mi}.safetensors"
        safetensors.torch.save_file(state, str(p))
        paths.append(p)
    m = [amd.model(p) for p in paths]
    recipe = amd.add_difference(m[0], amd.ties_sum(amd.subtract(m[1], m[0]), amd.subtract(m[2], m[0]), k=0.2), alpha=1.0)
    one = amd.merge(recipe, merge_device="cuda", output_dtype=torch.bfloat16)
    out = tmp_path / "merged.safetensors"
    amd.merge(recipe, merge_device="cuda", output_dtype=torch.bfloat16, total_buffer_size=32 << 20, output=out)
    from sd_mecha_amd import merging
    # 16 MiB per buffer set = 2.1 M elements per model and wave: `emb` (6.9 M elements) is larger than the window and crosses
    # it in row pieces; the pinned staging stays inside the 32 MiB budget (the reference grows its buffer, streaming.py:154-158)
    assert merging.last_stats["split_keys"] == 1 and merging.last_stats["pinned_bytes"] <= 32 << 20, merging.last_stats
    many = safetensors.torch.load_file(str(out))
    assert sorted(many) == sorted(shapes)
    for k in shapes:
        assert torch.equal(many[k], one[k]), k
    ins = [safetensors.torch.load_file(str(p)) for p in paths]
    for k in ("blk7.w", "blk99.b", "emb"):
        f32 = [O.to_f32(t[k].view(torch.int16).numpy().view(np.uint16), "bfloat16") for t in ins]
        want = O.from_f32(O.add_difference_ties(f32[0], f32[1:], k=0.2, alpha=1.0), "bfloat16")
        assert np.array_equal(many[k].view(torch.int16).numpy().view(np.uint16), want), k


def test_generic_path_threads_match_the_callers_thread(toy_models, tmp_path):     # noqa: F811
    """merge(threads=N): pool threads, each inside its own HIP stream (reference merging.py:171-179, :330-341), give the
    same tensors as threads=0, to a dict and to a file; a failing key still raises."""
    import safetensors.torch
    import sd_mecha_amd as amd
    m = [amd.model(p) for p in toy_models]
    kw = dict(merge_device="cuda", output_dtype=torch.bfloat16)
    recipe = amd.rotate(m[0], m[1], alignment=1.0, alpha=0.0)
    want = amd.merge(recipe, threads=0, **kw)
    got = amd.merge(recipe, threads=4, **kw)
    assert set(got) == set(want)
    for k in want:
        if k != "__metadata__":
            assert torch.equal(got[k], want[k]), k
    out = tmp_path / "rot.safetensors"
    amd.merge(recipe, threads=3, output=out, **kw)
    loaded = safetensors.torch.load_file(str(out))
    for k in want:
        if k != "__metadata__":
            assert torch.equal(loaded[k], want[k].cpu()), k

    @amd.merge_method(register=False)
    def boom(a, b, **kwargs):
        raise ValueError("no")
    with pytest.raises(ValueError):
        amd.merge(boom(m[0], m[1]), threads=2, **kw)


def test_inputs_preloaded_in_pinned_memory_merge_identically(toy_models):     # noqa: F811
    """SURVEY.md §8d's "from pinned RAM, no files" mode: with every input preloaded (streaming.preload_pinned) the fused pipeline
    skips its staging copies and enqueues the H2D copies straight from the pinned images; results are those of the file path."""
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging, streaming
    m = [amd.model(p) for p in toy_models]
    deltas = [amd.subtract(x, m[0]) for x in m[1:4]]
    recipe = amd.add_difference(m[0], amd.ties_sum(*deltas, k=0.2), alpha=1.0) | m[0]
    kw = dict(merge_device="cuda", output_dtype=torch.bfloat16)
    for buffer in (1 << 31, 1 << 16):
        want = amd.merge(recipe, total_buffer_size=buffer, **kw)
        assert merging.last_stats["source"] == "files"
        try:
            for p in toy_models:
                streaming.preload_pinned(p)
            got = amd.merge(recipe, total_buffer_size=buffer, **kw)
            # (a job with keys larger than the staging window moves those through the window: it keeps the staging path)
            want_source = "pinned images" if merging.last_stats["split_keys"] == 0 else "files"
            assert merging.last_stats["source"] == want_source and (want_source == "files" or merging.last_stats["stage"] < 0.01)
        finally:
            streaming.release_pinned()
        assert sorted(got) == sorted(want)
        for k in want:
            if k != "__metadata__":
                assert torch.equal(got[k], want[k]), k
    check_against_reference(got, "ties4_fallback", exact=False)
