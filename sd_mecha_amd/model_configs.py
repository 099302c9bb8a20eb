"""Model configs and their auto-detection by key-name overlap.

The reference keeps one YAML table per architecture (extensions/builtin/model_configs/*.yaml) and, for inputs without
an explicit config, picks the config whose keys overlap the state dict's (graph_finalization.py:1024-1058: a config is a
candidate if it matches at least 10 % of the state dict's keys, :1036; candidates of several inputs are intersected and
their statistics averaged, :591-611, :1001-1013; the candidate with the fewest state-dict keys outside the config wins,
:698; with no candidate the config is structural = the keys the inputs share, :695-696).  With a known config the output
keys are the config's, and an input may hold a key under one of its aliases (merging.py:394-403).

The tables here are the manifests exported from the reference's YAML (tools/export_shapes.py): key, shape, dtype,
optional flag and aliases per component.  Only the configs of the benchmark matrix are shipped (sd1-ldm, sdxl-sgm,
sdxl-kohya, flux-flux); anything else resolves to the structural config, which is what this backend used before."""
from __future__ import annotations

import dataclasses
import gzip
import json
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Set, Tuple

from . import shapes


@dataclasses.dataclass(frozen=True)
class KeyMetadata:
    shape: Optional[Tuple[int, ...]]
    dtype: Optional[str]
    optional: bool
    aliases: Tuple[str, ...]


class ModelConfig:
    """extensions/model_configs.py:99-120, reduced: identifier, keys(), aliases(), resolve_alias()."""

    def __init__(self, identifier: str, keys: Mapping[str, KeyMetadata], components: Mapping[str, Sequence[str]]):
        self.identifier = identifier
        self._keys = dict(keys)
        self._components = {c: tuple(ks) for c, ks in components.items()}
        self._alias_to_key = {alias: k for k, m in self._keys.items() for alias in m.aliases}

    def keys(self) -> Mapping[str, KeyMetadata]:
        return self._keys

    def components(self) -> Mapping[str, Tuple[str, ...]]:
        return self._components

    def aliases(self) -> Mapping[str, Tuple[str, ...]]:
        return {k: m.aliases for k, m in self._keys.items()}

    def resolve_alias(self, key: str) -> str:
        return self._alias_to_key.get(key, key)

    def __repr__(self):
        return f"ModelConfig({self.identifier!r}, {len(self._keys)} keys)"


_cache: Dict[str, ModelConfig] = {}


def resolve(identifier: str) -> ModelConfig:
    """extensions/model_configs.py `resolve`: the config registered under this identifier (the YAML file's stem)."""
    if identifier not in _cache:
        path = shapes._HERE / f"{identifier}.json.gz"
        if not path.exists():
            raise ValueError(f"unknown model config: {identifier!r}; known: {shapes.available()}")
        doc = json.loads(gzip.decompress(path.read_bytes()))
        keys, components = {}, {}
        for cname, entries in doc["components"].items():
            names = []
            for key, shape, dtype, optional, *rest in entries:
                aliases = tuple(rest[0]) if rest else ()
                keys[key] = KeyMetadata(tuple(shape) if shape is not None else None, dtype, bool(optional), aliases)
                names.append(key)
            components[cname] = names
        _cache[identifier] = ModelConfig(identifier, keys, components)
    return _cache[identifier]


def get_all() -> List[ModelConfig]:
    return [resolve(i) for i in shapes.available()]


@dataclasses.dataclass
class ConfigMatchStats:
    """graph_finalization.py:993-1013."""
    intersection: Set[str]
    state_dict_misses: float = 0.0      # share of the state dict's keys the config does not know
    config_misses: float = 0.0          # share of the config's mandatory keys the state dict lacks
    count: int = 0

    def __or__(self, other: "ConfigMatchStats") -> "ConfigMatchStats":
        new_count = self.count + other.count
        common = self.intersection & other.intersection
        if new_count == 0:
            return ConfigMatchStats(common, 0.0, 0.0, new_count)
        sw, ow = self.count / new_count, other.count / new_count
        return ConfigMatchStats(common, self.state_dict_misses * sw + other.state_dict_misses * ow,
                                self.config_misses * sw + other.config_misses * ow, new_count)


def create_config_stats(config: ModelConfig, dict_keys: Optional[Set[str]] = None) -> ConfigMatchStats:
    """graph_finalization.py:1042-1058."""
    config_keys = set(config.keys())
    if not dict_keys:
        return ConfigMatchStats(config_keys)
    n = len(dict_keys)
    resolved = {config.resolve_alias(k) for k in dict_keys}
    matched = resolved & config_keys
    extra = resolved - config_keys
    missing = sum(1 for k, m in config.keys().items() if k not in resolved and not m.optional)
    return ConfigMatchStats(matched, len(extra) / n, missing / len(config.keys()), 1)


def generate_configs_stats(state_dict_keys: Iterable[str]) -> Dict[str, ConfigMatchStats]:
    """graph_finalization.py:1024-1039: the configs that match at least 10 % of the state dict's keys."""
    keys = set(state_dict_keys)
    if not keys:
        return {}
    out = {}
    for config in get_all():
        stats = create_config_stats(config, keys)
        if len(stats.intersection) >= len(keys) * 0.1:
            out[config.identifier] = stats
    return out


def infer(key_sets: Sequence[Iterable[str]], hints: Sequence[Optional[str]] = ()) -> Optional[ModelConfig]:
    """The config shared by the inputs of a merge, or None for the structural config.

    ``hints[i]`` is input i's explicit config identifier (``model(path, config=...)``) or None to infer it.  Candidates
    of every input are intersected (graph_finalization.py:591-624); mixing explicit configs, or an explicit config no
    inferred input can have, raises TypeError as the reference does (:604-610, :653-659)."""
    stats: Optional[Dict[str, ConfigMatchStats]] = None
    explicit: Set[str] = set()
    for i, keys in enumerate(key_sets):
        hint = hints[i] if i < len(hints) else None
        if hint is None:
            new = generate_configs_stats(keys)
        else:
            cfg = resolve(hint) if isinstance(hint, str) else hint
            explicit.add(cfg.identifier)
            new = {cfg.identifier: create_config_stats(cfg, set(keys))}
        if stats is None:
            stats = new
        else:
            common = set(stats) & set(new)
            if not common and explicit:
                raise TypeError("Incompatible model config constraints: candidate set intersection is empty.\n"
                                f"  Existing: {sorted(stats)}\n  New:      {sorted(new)}")
            stats = {c: stats[c] | new[c] for c in common}
    if not stats:
        return None
    best = min(stats.items(), key=lambda kv: kv[1].state_dict_misses)[0]
    return resolve(best)
