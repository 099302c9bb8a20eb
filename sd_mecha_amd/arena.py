"""HBM layout of one key shard: every model's tensors for the shard live in ONE contiguous arena with the
SAME element offsets, so a whole shard is merged by one launch sequence (include/mecha_hip.h "batched").

Replaces the per-key tensors the reference's executor creates one future at a time (merging.py:223-229).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch

from . import _lib

ALIGN_ELEMS = 128          # key start alignment in elements: 256 B for fp16/bf16, 512 B for fp32


@dataclass
class ArenaLayout:
    keys: List[str]
    numels: List[int]
    offsets: List[int]         # in elements, multiples of ALIGN_ELEMS
    total_elems: int           # arena length in elements (gaps included)

    @property
    def payload_elems(self) -> int:
        return sum(self.numels)

    def host_segments(self):
        """ctypes array of mecha_segment (HOST copy)."""
        return (_lib.Segment * len(self.keys))(*[_lib.Segment(o, n) for o, n in zip(self.offsets, self.numels)])

    def device_segments(self, device) -> torch.Tensor:
        """The same table as an int64 [n, 2] DEVICE tensor (layout-compatible with mecha_segment[])."""
        t = torch.tensor([[o, n] for o, n in zip(self.offsets, self.numels)], dtype=torch.int64)
        return t.to(device)


def plan_arena(items: Sequence[Tuple[str, int]], align: int = ALIGN_ELEMS) -> ArenaLayout:
    """Lay keys out back to back, each start rounded up to `align` elements.  Keys with numel == 0 are
    not placed (the executor passes empty tensors through untouched, like the reference)."""
    assert align % 8 == 0
    keys, numels, offsets = [], [], []
    off = 0
    for key, n in items:
        if n <= 0:
            continue
        keys.append(key)
        numels.append(int(n))
        offsets.append(off)
        off = (off + int(n) + align - 1) // align * align
    return ArenaLayout(keys, numels, offsets, max(off, align))


def shard_keys_lpt(items: Sequence[Tuple[str, int]], n_shards: int) -> List[List[Tuple[str, int]]]:
    """Longest-processing-time-first assignment of keys to GPUs by element count:
    sort by size descending, give each key to the least-loaded shard.  Deterministic; no data-path
    collective is needed because keys are independent (merging.py:223-229)."""
    shards: List[List[Tuple[str, int]]] = [[] for _ in range(n_shards)]
    loads = [0] * n_shards
    order = sorted(range(len(items)), key=lambda i: (-items[i][1], i))
    for i in order:
        s = min(range(n_shards), key=lambda j: (loads[j], j))
        shards[s].append(items[i])
        loads[s] += items[i][1]
    return shards
