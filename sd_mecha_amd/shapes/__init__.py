"""Key -> shape manifests of the model configs the benchmark configs use (data exported from the
reference's YAML tables by tools/export_shapes.py)."""
from __future__ import annotations

import gzip
import json
import math
import pathlib
from typing import List, Optional, Sequence, Tuple

_HERE = pathlib.Path(__file__).resolve().parent


def available() -> List[str]:
    return sorted(p.name[: -len(".json.gz")] for p in _HERE.glob("*.json.gz"))


def load_manifest(identifier: str, components: Optional[Sequence[str]] = None) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(key, shape, dtype)] of a model config, optionally restricted to some components
    (e.g. flux-flux ``("diffuser",)`` = the 780 keys / 11.9 B params of BASELINE config 3)."""
    path = _HERE / f"{identifier}.json.gz"
    if not path.exists():
        raise KeyError(f"no shape manifest for {identifier!r}; available: {available()}")
    doc = json.loads(gzip.decompress(path.read_bytes()))
    out = []
    for cname, keys in doc["components"].items():
        if components is not None and cname not in components:
            continue
        for key, shape, dtype, *_rest in keys:            # [key, shape, dtype, optional, aliases]
            if shape is None:
                continue
            out.append((key, tuple(int(s) for s in shape), dtype))
    return out


def numel(shape: Sequence[int]) -> int:
    return int(math.prod(shape))
