"""NUMA placement of one rank (one process per GPU).  No counterpart in the reference (single process, <= 4 threads).

With N ranks on one host every rank runs its own staging team (pread into pinned buffers) and writer; without placement the
threads of all ranks roam over every socket and the pinned buffers land wherever they were first touched.  ``bind_rank`` pins
the PROCESS (threads created afterwards inherit the mask: the staging team, the writer) to the cores of one NUMA node --
the GPU's own node when sysfs knows it, else an even split of the nodes over the local ranks -- so that the first touch of the
pinned buffers, and all staging traffic after it, stays on that node.  Plain os.sched_setaffinity + sysfs; nothing to install."""
from __future__ import annotations

import os
import pathlib
from typing import List, Optional

_bound: Optional[dict] = None


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def nodes(root: str = "/sys/devices/system/node") -> List[List[int]]:
    """cpus of every NUMA node that has any, in node order."""
    out = []
    base = pathlib.Path(root)
    if not base.is_dir():
        return out
    for d in sorted(base.glob("node[0-9]*"), key=lambda p: int(p.name[4:])):
        try:
            cpus = _parse_cpulist((d / "cpulist").read_text())
        except OSError:
            continue
        if cpus:
            out.append(cpus)
    return out


def gpu_node(pci_bus_id: Optional[str]) -> int:
    """NUMA node of a PCI device ("0000:c1:00.0"), -1 when sysfs does not say (virtual machines)."""
    if not pci_bus_id:
        return -1
    try:
        return int(pathlib.Path(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node").read_text().strip())
    except (OSError, ValueError):
        return -1


def plan(local_rank: int, local_world: int, node_cpus: List[List[int]], allowed: List[int], device_node: int = -1) -> Optional[List[int]]:
    """The cpus rank `local_rank` of `local_world` should run on, or None to leave the mask alone (one node, one rank, or a
    share that would be empty)."""
    if local_world <= 1 or len(node_cpus) <= 1:
        return None
    allowed_set = set(allowed)
    if 0 <= device_node < len(node_cpus):
        node = device_node
    else:
        node = local_rank * len(node_cpus) // local_world
    mine = [c for c in node_cpus[node] if c in allowed_set]
    # several ranks on one node share its cores evenly (contiguous slices)
    peers = [r for r in range(local_world) if (device_node >= 0 or r * len(node_cpus) // local_world == node)]
    if device_node < 0 and len(peers) > 1:
        i = peers.index(local_rank)
        per = len(mine) // len(peers)
        if per >= 2:
            mine = mine[i * per:(i + 1) * per]
    return mine if len(mine) >= 2 else None


def bind_rank(local_rank: int, local_world: int, pci_bus_id: Optional[str] = None) -> Optional[dict]:
    """Bind this process once; returns what was done (for the stats line) or None."""
    global _bound
    if _bound is not None or os.environ.get("MECHA_NUMA", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return _bound
    node_cpus = nodes()
    dev_node = gpu_node(pci_bus_id)
    cpus = plan(local_rank, local_world, node_cpus, sorted(os.sched_getaffinity(0)), dev_node)
    if cpus is None:
        return None
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    _bound = {"cpus": len(cpus), "first_cpu": cpus[0], "gpu_numa_node": dev_node, "numa_nodes": len(node_cpus)}
    return _bound
