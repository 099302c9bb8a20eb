"""CPU tests of the host side: safetensors reader/writer, arena planning, key sharding, the recipe fuser and the key
scheduler end to end (with the oracle as a test-double compute backend), including the world_size-2 path over gloo.
The expected outputs are the reference's own sd_mecha.merge() results (tests/golden/e2e.npz)."""
import json
import os
import pathlib
import struct
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))

SHAPES = {"w.linear": (128, 96), "w.conv": (32, 16, 3, 3), "w.bias": (128,), "w.scalar": ()}


def bf16(bits: np.ndarray, shape) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).view(torch.bfloat16).reshape(shape)


def tensor_bits(t: torch.Tensor) -> np.ndarray:
    t = t.contiguous()
    return t.view(torch.int16).numpy().view(np.uint16) if t.dtype != torch.float32 else t.numpy().view(np.uint32)


@pytest.fixture(scope="module")
def toy_models(tmp_path_factory):
    """The 4-model toy checkpoint set of the e2e goldens, written with the official safetensors package."""
    import safetensors.torch
    g = load_golden("e2e")
    d = tmp_path_factory.mktemp("toy")
    paths = []
    for m in range(4):
        state = {k: bf16(g[f"e_m{m}_{k}"], s) for k, s in SHAPES.items()}
        p = d / f"m{m}.safetensors"
        safetensors.torch.save_file(state, str(p))
        paths.append(p)
    return paths


def expected(name):
    g = load_golden("e2e")
    return {str(k): g[f"e_{name}_{k}"] for k in g[f"e_{name}_keys"]}


def check_against_reference(result: dict, name: str, exact=True):
    from conftest import ulp_distance_half
    want = expected(name)
    got = {k: v for k, v in result.items() if k != "__metadata__"}
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    for k, w in want.items():
        gb = tensor_bits(got[k]).reshape(-1)
        if exact:
            assert np.array_equal(gb, w.reshape(-1)), (name, k)
        else:
            assert ulp_distance_half(gb, w.reshape(-1)).max() <= 1, (name, k)
        assert tuple(got[k].shape) == SHAPES[k]


# ------------------------------------------------------------------ streaming.py
def test_reader_matches_official_loader(toy_models):
    import safetensors.torch
    from sd_mecha_amd.streaming import InSafetensorsDict
    from sd_mecha_amd import StateDictKeyError
    want = safetensors.torch.load_file(str(toy_models[0]))
    r = InSafetensorsDict(toy_models[0])
    assert sorted(r.keys()) == sorted(want)
    for k, t in want.items():
        assert torch.equal(r[k], t)
        assert bytes(r.raw(k)) == t.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()
        assert r.meta(k).shape == tuple(t.shape) and r.meta(k).dtype == t.dtype
    with pytest.raises(StateDictKeyError):
        r["nope"]
    with pytest.raises(StateDictKeyError):
        r["__metadata__"]
    r.close()
    with pytest.raises(ValueError):
        InSafetensorsDict(toy_models[0].with_suffix(".ckpt"))         # reference streaming.py:87-88


def test_writer_produces_valid_safetensors(tmp_path):
    import safetensors.torch
    from sd_mecha_amd.streaming import InSafetensorsDict, OutSafetensorsDict, TensorMetadata, compact_holes
    tensors = {"a": torch.randn(7, 3).to(torch.bfloat16), "b": torch.randn(5).half(), "c": torch.tensor(2.5), "e": torch.zeros(0, 4)}
    header = {k: TensorMetadata(t.shape, t.dtype) for k, t in tensors.items()}
    p = tmp_path / "out.safetensors"
    w = OutSafetensorsDict(p, header, "version 0.1.0\nmodel \"x\"")
    for k in ("c", "a", "e", "b"):                                       # completion order, not key order
        w[k] = tensors[k]
    w.close()
    back = safetensors.torch.load_file(str(p))                            # the official parser accepts the file
    assert sorted(back) == sorted(tensors)
    for k, t in tensors.items():
        assert torch.equal(back[k], t)
    raw = p.read_bytes()
    (hsize,) = struct.unpack("<Q", raw[:8])
    hdr = raw[8:8 + hsize].decode()
    assert hdr.endswith(" ") or hdr.endswith("}")                         # space padded (reference streaming.py:346-350)
    assert json.loads(hdr)["__metadata__"]["mecha_recipe"].startswith("version 0.1.0")
    r = InSafetensorsDict(p)
    assert r.file_metadata()["mecha_recipe"].startswith("version")
    r.close()
    # a sharded job that skipped a key leaves a hole; compaction restores a valid file
    p2 = tmp_path / "holes.safetensors"
    w = OutSafetensorsDict(p2, header, None)
    w["a"] = tensors["a"]
    w._cursor += 64                                                       # simulate an unused region
    w["b"] = tensors["b"]
    w.close()
    compact_holes(p2)
    back = safetensors.torch.load_file(str(p2))
    assert torch.equal(back["a"], tensors["a"]) and torch.equal(back["b"], tensors["b"])


# ------------------------------------------------------------------ arena.py
def test_arena_layout_and_lpt_sharding():
    from sd_mecha_amd.arena import ALIGN_ELEMS, plan_arena, shard_keys_lpt
    from sd_mecha_amd.shapes import load_manifest, numel
    items = [(k, numel(s)) for k, s, _ in load_manifest("sdxl-sgm")]
    assert len(items) == 2531 and sum(n for _, n in items) == 3_468_849_947
    L = plan_arena(items)
    assert all(o % ALIGN_ELEMS == 0 for o in L.offsets)
    ends = [o + n for o, n in zip(L.offsets, L.numels)]
    assert all(e <= o2 for e, o2 in zip(ends, L.offsets[1:])) and ends[-1] <= L.total_elems
    assert L.payload_elems == sum(n for _, n in items if n > 0)
    flux = [(k, numel(s)) for k, s, _ in load_manifest("flux-flux", ("diffuser",))]
    assert len(flux) == 780
    shards = shard_keys_lpt(flux, 8)
    assert sorted(k for s in shards for k, _ in s) == sorted(k for k, _ in flux)       # a partition
    loads = [sum(n for _, n in s) for s in shards]
    assert max(loads) / (sum(loads) / 8) < 1.02                                        # balanced within 2 %
    assert shard_keys_lpt(flux, 8) == shards                                           # deterministic on every rank


# ------------------------------------------------------------------ fuser
def test_fuser_recognises_the_benchmark_recipes(toy_models):
    import sd_mecha_amd as amd
    from sd_mecha_amd.merging import fuse
    m = [amd.model(p) for p in toy_models]
    p = fuse(amd.weighted_sum(m[0], m[1], alpha=0.3))
    assert p.kind == "wsum" and p.params["alpha"] == 0.3 and p.models == [m[0], m[1]]
    assert fuse(amd.weighted_sum(m[0], m[1])).params["alpha"] == 0.5
    assert fuse(amd.n_average(*m)).kind == "navg"
    deltas = [amd.subtract(x, m[0]) for x in m[1:]]
    t = fuse(amd.add_difference(m[0], amd.ties_sum(*deltas, k=0.2), alpha=1.0) | m[0])
    assert t.kind == "ties" and t.models == m and t.params == {"k": 0.2, "alpha": 1.0, "vote_sgn": False} and t.fallback is m[0]
    assert fuse(m[0] + amd.ties_sum(*deltas, k=0.2)).kind == "ties"                     # operator sugar: + is add_difference
    assert fuse(amd.add_difference(m[0], amd.ties_sum(amd.subtract(m[1], m[2]), k=0.2))) is None   # delta not against the base
    assert fuse(amd.slerp(m[0], m[1])).kind == "slerp"                                  # round 3: a batched plan of its own
    assert fuse(amd.slerp(m[0], amd.subtract(m[1], m[2]))) is None                      # generic per-key path
    text = amd.serialize(amd.add_difference(m[0], amd.ties_sum(*deltas, k=0.2), alpha=1.0))
    assert text.splitlines()[0] == "version 0.1.0" and 'merge "ties_sum"' in text and 'merge "add_difference" &0' in text


# ------------------------------------------------------------------ round 3: batched plans for the rest of north_star's streaming list
@pytest.fixture(scope="module")
def plan_models(toy_models, tmp_path_factory):
    """toy_models + a copy of m1 with an all-zero bias (the NaN branches of slerp / perpendicular_component) + three files
    without the 0-d key (the Model-Stock factor has no last dimension to take a cosine along)."""
    import safetensors.torch
    g, gp = load_golden("e2e"), load_golden("e2e_plans")
    d = tmp_path_factory.mktemp("plans")
    state = {k: bf16(g[f"e_m1_{k}"], s) for k, s in SHAPES.items()}
    state["w.bias"] = bf16(gp["p_z_w.bias"], SHAPES["w.bias"])
    safetensors.torch.save_file(state, str(d / "z.safetensors"))
    for m in range(3):
        safetensors.torch.save_file({k: bf16(g[f"e_m{m}_{k}"], s) for k, s in SHAPES.items() if len(s) >= 1}, str(d / f"s{m}.safetensors"))
    return {"m": list(toy_models), "z": d / "z.safetensors", "s": [d / f"s{m}.safetensors" for m in range(3)]}


def plan_recipes(amd, pm):
    """{golden name: (recipe, the plan kind fuse() must give it or None)} -- the recipes of gen_golden.gen_e2e_plans."""
    ms, z, ss = [amd.model(p) for p in pm["m"]], amd.model(pm["z"]), [amd.model(p) for p in pm["s"]]

    def addperp(a, b, c, alpha):
        return amd.add_difference(a, amd.perpendicular_component(amd.subtract(a, c), amd.subtract(b, c)), alpha=alpha)
    deltas = [amd.subtract(m, ms[0]) for m in ms[1:3]]
    sdeltas = [amd.subtract(m, ss[0]) for m in ss[1:]]
    return {
        "slerp": (amd.slerp(ms[0], ms[1], alpha=0.3), "slerp"),
        "slerp_zero": (amd.slerp(z, ms[2], alpha=0.6), "slerp"),
        "slerp_nonfinite": (amd.slerp(ms[0], ms[3], alpha=0.5) | ms[0], "slerp"),
        "addperp": (addperp(ms[0], ms[1], ms[2], 0.7), "addperp"),
        "addperp_zero": (addperp(z, ms[2], ms[1], 1.0), "addperp"),
        "addsub": (amd.add_difference(ms[0], amd.subtract(ms[1], ms[2]), alpha=0.6) | ms[0], "addsub"),
        "adddiff": (amd.add_difference(ms[0], amd.model(pm["m"][1], merge_space="delta"), alpha=0.25), "adddiff"),
        "dare_nostock": (amd.add_difference(ms[0], amd.ties_sum_with_dropout(*deltas, apply_stock=True, seed=114514, probability=0.4, k=0.5), alpha=1.0), "dare"),
        "dare_nostock_norescale": (amd.add_difference(ms[0], amd.ties_sum_with_dropout(*deltas, apply_stock=True, seed=7, probability=0.9, k=1.0, rescale=False), alpha=1.0), "dare"),
        "dare_default_seed": (amd.add_difference(ms[0], amd.ties_sum_with_dropout(*deltas, apply_stock=True, probability=0.5, k=0.2), alpha=1.0), "dare"),
        "dare_stock": (amd.add_difference(ss[0], amd.ties_sum_with_dropout(*sdeltas, seed=3, probability=0.4, k=0.5), alpha=0.8), "dare"),
    }


def check_plan_result(result: dict, name: str, max_ulp=1):
    from conftest import ulp_distance_half
    g = load_golden("e2e_plans")
    want = {str(k): g[f"p_{name}_{k}"] for k in g[f"p_{name}_keys"]}
    got = {k: v for k, v in result.items() if k != "__metadata__"}
    assert sorted(got) == sorted(want), (name, sorted(got), sorted(want))
    for k, w in want.items():
        d = ulp_distance_half(tensor_bits(got[k]).reshape(-1), w.reshape(-1))
        assert d.max() <= max_ulp, (name, k, int(d.max()), float((d > 0).mean()))
        assert tuple(got[k].shape) == SHAPES[k]


PLAN_EXACT = {"addsub", "adddiff"}            # one fma per element: bit for bit; the others hold reductions or TIES sums (<= 1 ulp)


@pytest.mark.parametrize("name", ["slerp", "slerp_zero", "slerp_nonfinite", "addperp", "addperp_zero", "addsub", "adddiff", "dare_nostock",
                                  "dare_nostock_norescale", "dare_default_seed", "dare_stock"])
@pytest.mark.parametrize("buffer", [1 << 31, 1 << 16])
def test_round3_plans_reproduce_reference_cpu(plan_models, name, buffer, tmp_path):
    """merge() on the batched plans of round 3 against the reference's own sd_mecha.merge outputs (tests/golden/e2e_plans.npz),
    scheduler on CPU with the oracle as the compute double: plan recognition, key routing (0-d and non-finite keys), fallbacks."""
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    from oracle_backend import OracleBackend
    recipe, kind = plan_recipes(amd, plan_models)[name]
    plan = merging.fuse(recipe)
    assert plan is not None and plan.kind == kind
    res = amd.merge(recipe, backend=OracleBackend(), merge_device="cpu", output_dtype=torch.bfloat16, total_buffer_size=buffer, threads=0)
    check_plan_result(res, name, 0 if name in PLAN_EXACT else 1)
    assert merging.last_stats["generic_keys"] == 0 and merging.last_stats["fused_keys"] == (3 if name == "dare_stock" else 4)


# ------------------------------------------------------------------ .mecha text (row f1)
def _mecha_recipes(amd, d):
    ms = [amd.model(d / f"m{i}.safetensors") for i in range(3)]
    ka, kb = (amd.model(d / f"{n}.safetensors", config="sdxl-kohya") for n in ("ka", "kb"))
    dl = [amd.subtract(m, ms[0]) for m in ms[1:]]
    return {
        "wsum": (amd.weighted_sum(ms[0], ms[1], alpha=0.5), {}),
        "wsum_default": (amd.weighted_sum(ms[0], ms[1]), {}),
        "navg": (amd.n_average(*ms), {}),
        "ties": (amd.add_difference(ms[0], amd.ties_sum(*dl, k=0.2), alpha=1.0) | ms[0], {}),
        "ties_vote_fallback_model": (amd.add_difference(ms[0], amd.ties_sum(*dl, k=0.2, vote_sgn=True), alpha=0.7), {"fallback_model": ms[0]}),
        "rotate": (amd.rotate(ms[0], ms[1], alignment=1.0, alpha=0.0), {}),
        "operators": (ms[0] + (ms[1] - ms[0]) * 0.5, {}),
        "slerp": (amd.slerp(ms[0], ms[1], alpha=0.3), {}),
        "known_config": (amd.weighted_sum(ka, kb, alpha=0.25), {}),
        "lora": (amd.add_difference(ka, amd.convert(amd.model(str(d / "lora0.safetensors")), ka), alpha=0.75) | ka, {}),
    }


def test_mecha_text_equals_the_references(tmp_path):
    """serialization.py:184-238: the recipe text this backend embeds in its outputs is, byte for byte, the text the reference
    embeds for the same recipe (tests/golden/mecha_text.json, from the real sd_mecha.merge); deserialize (:26-140) reads it
    back into a graph that serializes to the same text and fuses the same way."""
    import safetensors.torch
    import sd_mecha_amd as amd
    from sd_mecha_amd.merging import fuse
    from oracle_backend import OracleBackend
    want = json.loads((pathlib.Path(__file__).resolve().parent / "golden" / "mecha_text.json").read_text())
    g = torch.Generator().manual_seed(5)
    for i in range(3):
        safetensors.torch.save_file({"w.linear": torch.randn(12, 8, generator=g).to(torch.bfloat16), "w.bias": torch.randn(12, generator=g).to(torch.bfloat16)},
                                    str(tmp_path / f"m{i}.safetensors"))
    for n in ("ka", "kb"):
        safetensors.torch.save_file({"unet.out.2.weight": torch.randn(4, 320, 3, 3, generator=g).to(torch.bfloat16),
                                     "unet.out.2.bias": torch.randn(4, generator=g).to(torch.bfloat16)}, str(tmp_path / f"{n}.safetensors"))
    safetensors.torch.save_file({"lora_unet_out_2.lora_up.weight": (torch.randn(4, 2, generator=g) * 0.05).half(),
                                 "lora_unet_out_2.lora_down.weight": (torch.randn(2, 320, 3, 3, generator=g) * 0.05).half(),
                                 "lora_unet_out_2.alpha": torch.tensor(1.0).half()}, str(tmp_path / "lora0.safetensors"))
    recipes = _mecha_recipes(amd, tmp_path)
    assert sorted(recipes) == sorted(want)
    assert "convert_'sdxl-kohya_kohya_lora'_to_base" in want["lora"]            # the reference's identifier (lycoris.py:52)
    be = OracleBackend()
    for name, (recipe, extra) in recipes.items():
        expect = want[name].replace("<DIR>", str(tmp_path))
        if name not in ("rotate", "slerp"):                                  # the CPU double has no body for these two: text only
            res = amd.merge(recipe, backend=be, output_dtype=torch.bfloat16, threads=0, **extra)
            assert res["__metadata__"]["mecha_recipe"] == expect, name
        config = {"known_config": "sdxl-kohya"}.get(name)
        if name == "lora":                                                   # the deserialized conversion finds its base at merge time
            res2 = amd.merge(again_lora := amd.deserialize(expect), backend=be, output_dtype=torch.bfloat16, threads=0)
            assert all(torch.equal(res[k], res2[k]) for k in res if k != "__metadata__") and fuse(again_lora).kind == "lora"
        text = amd.serialize(recipe)
        assert text == expect, name                                          # standalone: the node's own config or "structural"
        again = amd.deserialize(expect)
        assert amd.serialize(again) == expect, name
        a, b = fuse(recipe), fuse(again)
        assert (a is None) == (b is None) and (a is None or a.kind == b.kind), name
        assert a is None or name == "lora" or a.params == b.params, name       # (a LoRA plan holds its file nodes, compared by identity)
        assert config is None or all(m.config == config for m in __import__("sd_mecha_amd").merging.collect_models(again))
    with pytest.raises(RuntimeError, match="bad recipe version"):
        amd.deserialize("version 9.9.9\nmodel \"x.safetensors\"")
    with pytest.raises(ValueError, match="line 2"):
        amd.deserialize("version 0.1.0\nfrobnicate 1 2")


def test_product_backend_has_no_cpu_path(toy_models):
    import sd_mecha_amd as amd
    m = [amd.model(p) for p in toy_models]
    with pytest.raises(RuntimeError, match="no CPU"):
        amd.merge(amd.weighted_sum(m[0], m[1]), merge_device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            amd.merge(amd.weighted_sum(m[0], m[1]), merge_device="cuda")


# ------------------------------------------------------------------ scheduler end to end (oracle as test double)
@pytest.mark.parametrize("buffer", [1 << 31, 1 << 18, 1 << 16])   # one wave / many small waves / keys larger than the window
def test_scheduler_reproduces_reference_merge(toy_models, tmp_path, buffer):
    import safetensors.torch
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    from oracle_backend import OracleBackend
    be = OracleBackend()
    m = [amd.model(p) for p in toy_models]
    kw = dict(backend=be, output_dtype=torch.bfloat16, total_buffer_size=buffer)
    check_against_reference(amd.merge(amd.weighted_sum(m[0], m[1], alpha=0.5), **kw), "wsum")
    if buffer == 1 << 16:
        # 32 KiB per buffer set = 5461 elements per model: w.linear (12288) and w.conv (4608 -- fits) ... the large key moves
        # through the window in row pieces, and the pinned staging stays inside the budget
        assert merging.last_stats["split_keys"] >= 1 and merging.last_stats["pinned_bytes"] <= 2 * (1 << 16), merging.last_stats
    check_against_reference(amd.merge(amd.weighted_sum(m[0], m[1], alpha=0.3), **{**kw, "output_dtype": torch.float16}), "wsum_f16")
    check_against_reference(amd.merge(amd.n_average(m[0], m[1], m[2]), **kw), "navg3")
    check_against_reference(amd.merge(amd.n_average(*m), **kw), "navg4")             # m3's bias is non-finite: key skipped
    deltas = [amd.subtract(x, m[0]) for x in m[1:3]]
    check_against_reference(amd.merge(amd.add_difference(m[0], amd.ties_sum(*deltas, k=0.2), alpha=1.0), **kw), "ties", exact=False)
    deltas4 = [amd.subtract(x, m[0]) for x in m[1:4]]
    res = amd.merge(amd.add_difference(m[0], amd.ties_sum(*deltas4, k=0.2), alpha=1.0) | m[0], **kw)
    check_against_reference(res, "ties4_fallback", exact=False)                        # skipped key comes from the base
    assert res["__metadata__"]["mecha_recipe"].startswith("version 0.1.0")
    # file output: same content, valid safetensors, recipe embedded
    out = tmp_path / f"merged_{buffer}.safetensors"
    assert amd.merge(amd.n_average(*m), output=out, **kw) is None
    check_against_reference(safetensors.torch.load_file(str(out)), "navg4")


def _rank_main(rank, world, paths, out_path, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    import sd_mecha_amd as amd
    from oracle_backend import OracleBackend
    m = [amd.model(p) for p in paths]
    amd.merge(amd.n_average(*m), backend=OracleBackend(), output_dtype=torch.bfloat16, output=out_path, total_buffer_size=1 << 16)
    dist.destroy_process_group()


def test_two_ranks_fill_one_file(toy_models, tmp_path):
    """N > 1: keys are dealt over ranks, every rank writes its own region of the one file, rank 0 writes the header and
    compacts the hole left by the skipped (non-finite) key.  No data-path collective."""
    import safetensors.torch
    import torch.multiprocessing as mp
    out = tmp_path / "two_ranks.safetensors"
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, [str(p) for p in toy_models], str(out), port), nprocs=2, join=True)
    check_against_reference(safetensors.torch.load_file(str(out)), "navg4")


def test_rotate_recipe_takes_the_batched_plan(toy_models, tmp_path):
    """rotate(model, model) is a fused plan: the scheduler stages both models' arenas and hands the backend every key with
    its shape (the HIP backend groups them by 2-D shape, svd.py:49-53).  Same values as the per-key path; with cache= the
    plan steps aside (svd.py:55-79 is a per-key feature)."""
    import sd_mecha_amd as amd
    from sd_mecha_amd import merging
    from oracle_backend import OracleBackend
    from oracle import oracle as O
    a, b = amd.model(toy_models[0]), amd.model(toy_models[1])
    node = amd.rotate(a, b, alignment=1.0, alpha=0.25)
    plan = merging.fuse(node)
    assert plan is not None and plan.kind == "rotate" and plan.params == {"alignment": 1, "alpha": 0.25, "centralization": 1.0}
    assert merging.fuse(amd.rotate(a, b, alignment=0.0, alpha=0.0)) is None          # returns `a` untouched (svd.py:39-40)
    assert merging.fuse(amd.rotate(a, b, alignment=0.5, alpha=0.0)) is None          # fractional alignment
    res = amd.merge(node, output_dtype=torch.float32, backend=OracleBackend())
    assert merging.last_stats["fused_keys"] == 4 and merging.last_stats["generic_keys"] == 0
    g = load_golden("e2e")
    for k, shape in SHAPES.items():
        x = O.to_f32(g[f"e_m0_{k}"], "bfloat16").reshape(shape)
        y = O.to_f32(g[f"e_m1_{k}"], "bfloat16").reshape(shape)
        want = O.rotate(x, y, alignment=1.0, alpha=0.25, centralization=1.0)
        assert np.allclose(res[k].numpy(), want, rtol=0, atol=1e-6 * max(1.0, float(np.abs(want).max()))), k


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (how the driver starts it) must start its own two ranks -- as a CHILD
    process, before anything touches a GPU -- which rendezvous and report.  --rehearse: over gloo, no GPU work."""
    import subprocess
    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--rehearse"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rehearsal"] and line["n_gpus"] == 2
    assert sorted((r["rank"], r["local_rank"]) for r in line["ranks"]) == [(0, 0), (1, 1)]


# ------------------------------------------------------------------ multi-rank host placement (round 3)
def test_numa_plan_and_host_thread_share(monkeypatch):
    """One process per GPU: a rank's staging team stays on one NUMA node (the GPU's own when sysfs knows it, else an even split)
    and takes its share of the cores; a single rank or a single node is left alone."""
    from sd_mecha_amd import merging, numa
    two = [list(range(0, 64)), list(range(64, 128))]
    allowed = list(range(128))
    assert numa.plan(0, 1, two, allowed) is None and numa.plan(0, 8, [allowed], allowed) is None
    shares = [numa.plan(r, 8, two, allowed) for r in range(8)]
    assert all(len(s) == 16 for s in shares) and sorted(c for s in shares for c in s) == allowed      # 4 ranks per node, 16 cores each
    assert all(set(s) <= set(two[r // 4]) for r, s in enumerate(shares))
    assert numa.plan(5, 8, two, allowed, device_node=0) == two[0]                                    # the GPU's own node wins
    assert numa.plan(0, 2, two, list(range(60, 70))) == [60, 61, 62, 63]                             # cgroup-restricted mask
    assert numa.plan(0, 2, two, [64, 65]) is None                                                    # nothing of node 0 is allowed
    assert numa._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    monkeypatch.delenv("MECHA_HOST_THREADS", raising=False)
    assert 1 <= merging.default_host_threads() <= 16
    monkeypatch.setenv("MECHA_HOST_THREADS", "3")
    assert merging.default_host_threads() == 3
