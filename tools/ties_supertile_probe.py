#!/usr/bin/env python3
"""Does the Infinity Cache serve the merge pass's re-read when a TIES step runs group by group?

The SDXL x4 job of bench.py, but the batched entry point is called once per GROUP of whole keys whose inputs
total at most --group-mb megabytes (count -> resolve -> merge per group, so the merge pass re-reads what the count
pass of the same group has just read).  Prints the per-kernel HIP-event times summed over the groups.

    MECHA_TIES_COUNT_NT=0 python tools/ties_supertile_probe.py --group-mb 160
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sd_mecha_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group-mb", type=float, nargs="+", default=[0, 512, 256, 160, 96, 48])
    ap.add_argument("--workload", default="sdxl_ties")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    job = bench.Job(args.workload, 0, device)
    L, lib = job.layout, job.lib
    n = len(L.keys)
    for mb in args.group_mb:
        cap = int(mb * 1e6 / 2 / job.n_models) if mb > 0 else 1 << 62       # elements per group
        groups, start, acc = [], 0, 0
        for i in range(n):
            if acc and acc + L.numels[i] > cap:
                groups.append((start, i)); start, acc = i, 0
            acc += L.numels[i]
        groups.append((start, n))
        tables = []
        for a, b in groups:
            host = (_lib.Segment * (b - a))(*[_lib.Segment(L.offsets[i], L.numels[i]) for i in range(a, b)])
            dev = torch.tensor([[L.offsets[i], L.numels[i]] for i in range(a, b)], dtype=torch.int64).to(device)
            tables.append((host, dev, b - a, a))
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def step():
            for host, dev, cnt, a in tables:
                st = lib.mecha_batched_add_difference_ties(
                    job.arenas[0].data_ptr(), job.model_ptrs, job.n_models - 1, job.out.data_ptr(), dev.data_ptr(), ctypes.addressof(host), cnt,
                    L.total_elems, _lib.BF16, _lib.BF16, 0.2, 1.0, 0, None, job.ws.data_ptr(), job.ws.numel(),
                    job.flags[a:].data_ptr(), stream)
                _lib.check(st, "ties")
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        _lib.profile_enable(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        k = _lib.profile_read()
        _lib.profile_enable(False)
        print(json.dumps({"group_mb": mb, "groups": len(groups), "wall_ms_per_step": round(wall, 3), "count_nt": os.environ.get("MECHA_TIES_COUNT_NT", "1"),
                          "kernels_ms_per_step": {name: round(ms / args.steps, 3) for name, (ms, _) in k.items() if ms / args.steps > 0.02}}), flush=True)


if __name__ == "__main__":
    main()
