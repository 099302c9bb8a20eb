"""Config auto-detection by key-name overlap and alias resolution (sd_mecha_amd/model_configs.py; reference
graph_finalization.py:1024-1058, :591-624, :698 and merging.py:394-403).  The expected candidate statistics and choices
are the reference's own (tests/golden/configs.npz, made by gen_golden.py from its generate_configs_stats)."""
import pathlib
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))


def test_candidate_stats_match_reference():
    from sd_mecha_amd import model_configs as mc
    g = load_golden("configs")
    shipped = [str(s) for s in g["shipped"]]
    assert sorted(c.identifier for c in mc.get_all()) == sorted(shipped)
    for name in (str(n) for n in g["case_names"]):
        keys = [str(k) for k in g[f"{name}_keys"]]
        stats = mc.generate_configs_stats(keys)
        want_cand = [str(c) for c in g[f"{name}_candidates"]]
        assert sorted(stats) == sorted(want_cand), name
        for c, (n_match, sd_miss, cfg_miss) in zip(want_cand, g[f"{name}_stats"]):
            assert len(stats[c].intersection) == int(n_match), (name, c)
            assert abs(stats[c].state_dict_misses - sd_miss) < 1e-12 and abs(stats[c].config_misses - cfg_miss) < 1e-12, (name, c)
        best = mc.infer([keys])
        assert (best.identifier if best is not None else "") == str(g[f"{name}_best"]), name


def test_infer_intersects_inputs_and_honours_hints():
    from sd_mecha_amd import model_configs as mc
    g = load_golden("configs")
    sdxl = [str(k) for k in g["sdxl_full_keys"]]
    sd1 = [str(k) for k in g["sd1_no_ema_keys"]]
    flux = [str(k) for k in g["flux_full_keys"]]
    assert mc.infer([sdxl, sdxl]).identifier == "sdxl-sgm"
    # candidates of every input are intersected: sd1 + sdxl share sd1-ldm / sdxl-sgm as candidates, the averaged misses pick one
    both = mc.infer([sdxl, sd1])
    assert both is not None and both.identifier in ("sd1-ldm", "sdxl-sgm")
    assert mc.infer([sdxl, flux]) is None                       # no shared candidate, nothing explicit: structural
    assert mc.infer([sdxl], ["sdxl-sgm"]).identifier == "sdxl-sgm"
    with pytest.raises(TypeError):
        mc.infer([sdxl, flux], ["sdxl-sgm", None])              # an explicit config the other input cannot have
    cfg = mc.resolve("flux-flux")
    assert cfg.resolve_alias("double_blocks.0.img_attn.norm.key_norm.scale") == "model.diffusion_model.double_blocks.0.img_attn.norm.key_norm.scale"
    assert cfg.resolve_alias("nope") == "nope"
    assert len(cfg.components()["diffuser"]) == 780


def test_merge_detects_config_and_resolves_aliases(tmp_path):
    """Two FLUX-like files (tiny tensors under real key names; one uses the alias names a diffuser-only file has, plus a
    junk key): the merge runs over the config's keys, reads the alias-named tensors and writes canonical names."""
    import safetensors.torch
    import sd_mecha_amd as amd
    from oracle_backend import OracleBackend
    from sd_mecha_amd import model_configs as mc
    cfg = mc.resolve("flux-flux")
    canon = list(cfg.components()["diffuser"])[:60]
    rng = np.random.default_rng(3)
    a = {k: torch.from_numpy(rng.normal(0, 0.02, size=(4, 6)).astype(np.float32)).to(torch.bfloat16) for k in canon}
    b = {cfg.keys()[k].aliases[0]: torch.from_numpy(rng.normal(0, 0.02, size=(4, 6)).astype(np.float32)).to(torch.bfloat16) for k in canon}
    b["junk.weight"] = torch.zeros(3, dtype=torch.bfloat16)
    a["junk.weight"] = torch.ones(3, dtype=torch.bfloat16)
    safetensors.torch.save_file(a, str(tmp_path / "a.safetensors"))
    safetensors.torch.save_file(b, str(tmp_path / "b.safetensors"))
    out = amd.merge(amd.weighted_sum(amd.model(tmp_path / "a.safetensors"), amd.model(tmp_path / "b.safetensors"), alpha=0.5),
                    backend=OracleBackend(), output_dtype=torch.float32)
    got = {k for k in out if k != "__metadata__"}
    assert got == set(canon)                                     # canonical names; the junk key is not in the config
    for k in canon:
        want = torch.lerp(a[k].float(), b[cfg.keys()[k].aliases[0]].float(), 0.5)
        assert torch.equal(out[k], want), k
    # unknown key names: structural config, every shared key is merged
    c = {f"layer.{i}.weight": torch.full((2, 2), float(i), dtype=torch.bfloat16) for i in range(5)}
    safetensors.torch.save_file(c, str(tmp_path / "c.safetensors"))
    out = amd.merge(amd.weighted_sum(amd.model(tmp_path / "c.safetensors"), amd.model(tmp_path / "c.safetensors"), alpha=0.5),
                    backend=OracleBackend(), output_dtype=torch.float32)
    assert {k for k in out if k != "__metadata__"} == set(c)


def _alias_case(tmp_path):
    import safetensors.torch
    g = load_golden("e2e_alias")
    canon, aliases = [str(k) for k in g["canon"]], [str(k) for k in g["aliases"]]

    def bf16(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).view(torch.bfloat16).reshape(8, 16)
    a = {k: bf16(x) for k, x in zip(canon, g["a"])}
    b = {k: bf16(x) for k, x in zip(aliases, g["b"])}
    a["junk.weight"] = torch.ones(3, dtype=torch.bfloat16)
    b["junk.weight"] = torch.zeros(3, dtype=torch.bfloat16)
    safetensors.torch.save_file(a, str(tmp_path / "a.safetensors"))
    safetensors.torch.save_file(b, str(tmp_path / "b.safetensors"))
    return g, canon


def check_alias_merge(tmp_path, **kw):
    """The reference's own output for an alias-named FLUX-like pair (tests/golden/e2e_alias.npz): detected config, canonical
    output names, bit-identical tensors, and the config named in the embedded recipe."""
    import sd_mecha_amd as amd
    g, canon = _alias_case(tmp_path)
    out = amd.merge(amd.weighted_sum(amd.model(tmp_path / "a.safetensors"), amd.model(tmp_path / "b.safetensors"), alpha=0.3),
                    output_dtype=torch.bfloat16, **kw)
    text = out.pop("__metadata__")["mecha_recipe"]
    assert f"model_config={str(g['config_line'])} " in text
    assert sorted(out) == [str(k) for k in g["out_keys"]]
    for k, want in zip(canon, g["out"]):
        assert np.array_equal(out[k].cpu().contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1), want.reshape(-1)), k


def test_alias_merge_matches_reference_cpu(tmp_path):
    from oracle_backend import OracleBackend
    check_alias_merge(tmp_path, backend=OracleBackend())


@pytest.mark.gpu
def test_alias_merge_matches_reference_gpu(tmp_path):
    """Row (f)2 on the HIP path: an alias-named, auto-detected checkpoint pair through the fused batched kernels."""
    from sd_mecha_amd import merging
    check_alias_merge(tmp_path, merge_device="cuda")
    assert merging.last_stats["fused_keys"] == 60 and merging.last_stats["generic_keys"] == 0
