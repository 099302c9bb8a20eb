#!/usr/bin/env python3
"""Export key -> (shape, dtype) manifests of the reference's model configs (build container only).

The reference keeps these tables as YAML (sd_mecha/extensions/builtin/model_configs/*.yaml, loaded through
model_configs.resolve(id).keys()).  The manifests are data -- the shapes of SD1.5, SDXL
and FLUX checkpoints -- used to build synthetic checkpoints of the benchmark configs.

    python tools/export_shapes.py      # rewrites sd_mecha_amd/shapes/*.json.gz
"""
import gzip
import json
import pathlib
import sys
import types

ROOT = pathlib.Path(__file__).resolve().parent.parent
fw = types.ModuleType("fuzzywuzzy")
fwp = types.ModuleType("fuzzywuzzy.process")
fwp.extractOne = lambda q, choices, **k: (next(iter(choices), None), 0)
fw.process = fwp
sys.modules["fuzzywuzzy"] = fw
sys.modules["fuzzywuzzy.process"] = fwp
sys.path.insert(0, "/root/reference")

from sd_mecha.extensions import model_configs  # noqa: E402

OUT = ROOT / "sd_mecha_amd" / "shapes"
OUT.mkdir(exist_ok=True)

for ident in ("sd1-ldm", "sdxl-sgm", "flux-flux", "sdxl-kohya"):
    cfg = model_configs.resolve(ident)
    comps = {}
    for cname, comp in cfg.components().items():
        comps[cname] = [
            [k, list(m.shape) if m.shape is not None else None, str(m.dtype).replace("torch.", "") if m.dtype is not None else None,
             bool(m.optional), list(m.aliases)]
            for k, m in comp.keys().items()
        ]
    n_keys = sum(len(v) for v in comps.values())
    n_params = sum(int(__import__("math").prod(e[1])) for v in comps.values() for e in v if e[1] is not None)
    doc = {"identifier": ident, "n_keys": n_keys, "n_params": n_params, "components": comps}
    path = OUT / f"{ident}.json.gz"
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(doc, separators=(",", ":")).encode())
    print(ident, n_keys, "keys", f"{n_params / 1e9:.3f} B params", {c: len(v) for c, v in comps.items()}, path.stat().st_size, "bytes")
