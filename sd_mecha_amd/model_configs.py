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


# ------------------------------------------------------------------ detection by key-name overlap
class _NameIndex:
    """Every name a shipped config knows a tensor by (its key, or one of the key's aliases) -> the configs that know it.
    Built once; a state dict is then matched against ALL configs in one pass over its keys (the reference walks the
    configs one by one and resolves every key for each, graph_finalization.py:1024-1058)."""

    def __init__(self, configs: Sequence[ModelConfig]):
        self.configs = list(configs)
        self.by_name: Dict[str, List[Tuple[int, str, bool]]] = {}          # name -> [(config index, canonical key, mandatory)]
        self.n_keys = [len(c.keys()) for c in self.configs]
        self.n_mandatory = [0] * len(self.configs)
        for ci, cfg in enumerate(self.configs):
            for key, meta in cfg.keys().items():
                mandatory = not meta.optional
                self.n_mandatory[ci] += mandatory
                for name in (key, *meta.aliases):
                    self.by_name.setdefault(name, []).append((ci, key, mandatory))


_index: Optional[_NameIndex] = None


def _name_index() -> _NameIndex:
    global _index
    if _index is None:
        _index = _NameIndex(get_all())
    return _index


@dataclasses.dataclass
class ConfigMatchStats:
    """How well one config explains the state dicts seen so far (graph_finalization.py:993-1013): the config keys they all
    hold, and -- averaged over the state dicts -- the share of a state dict's keys the config does not know and the share
    of the config's keys that are mandatory and absent.  Kept as sums over `count` state dicts."""
    intersection: Set[str]
    unknown_share_sum: float = 0.0
    absent_share_sum: float = 0.0
    count: int = 0

    @property
    def state_dict_misses(self) -> float:
        return self.unknown_share_sum / self.count if self.count else 0.0

    @property
    def config_misses(self) -> float:
        return self.absent_share_sum / self.count if self.count else 0.0

    def __or__(self, other: "ConfigMatchStats") -> "ConfigMatchStats":
        return ConfigMatchStats(self.intersection & other.intersection, self.unknown_share_sum + other.unknown_share_sum,
                                self.absent_share_sum + other.absent_share_sum, self.count + other.count)


def _match(names: Iterable[str], only: Optional[int] = None) -> Tuple[int, List[Optional[ConfigMatchStats]]]:
    """One pass over a state dict's key names: per config, the canonical keys they resolve to, how many of the names the
    config knows at all, how many mandatory keys they cover.  `only`: restrict to one config index."""
    idx = _name_index()
    n_cfg = len(idx.configs)
    hit: List[Set[str]] = [set() for _ in range(n_cfg)]
    known = [0] * n_cfg
    mandatory_hit = [0] * n_cfg
    total = 0
    for name in names:
        total += 1
        for ci, key, mandatory in idx.by_name.get(name, ()):
            known[ci] += 1
            if key not in hit[ci]:
                hit[ci].add(key)
                mandatory_hit[ci] += mandatory
    out: List[Optional[ConfigMatchStats]] = [None] * n_cfg
    for ci in range(n_cfg):
        if only is not None and ci != only:
            continue
        unknown = (total - known[ci]) / total if total else 0.0
        absent = (idx.n_mandatory[ci] - mandatory_hit[ci]) / idx.n_keys[ci] if idx.n_keys[ci] else 0.0
        out[ci] = ConfigMatchStats(hit[ci], unknown, absent, 1)
    return total, out


def create_config_stats(config: ModelConfig, dict_keys: Optional[Iterable[str]] = None) -> ConfigMatchStats:
    """The statistics of one given config for one state dict (graph_finalization.py:1042-1058); without a state dict, the
    config's own keys and no observation."""
    names = set(dict_keys) if dict_keys is not None else set()
    if not names:
        return ConfigMatchStats(set(config.keys()))
    idx = _name_index()
    ci = next(i for i, c in enumerate(idx.configs) if c.identifier == config.identifier)
    return _match(names, only=ci)[1][ci]


def generate_configs_stats(state_dict_keys: Iterable[str]) -> Dict[str, ConfigMatchStats]:
    """The candidate configs of a state dict: those whose keys cover at least 10 % of its key names
    (graph_finalization.py:1024-1039, the 10 % rule at :1036)."""
    names = set(state_dict_keys)
    if not names:
        return {}
    total, stats = _match(names)
    idx = _name_index()
    return {cfg.identifier: st for cfg, st in zip(idx.configs, stats) if st is not None and len(st.intersection) >= total * 0.1}


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
