#!/usr/bin/env python3
"""bench.py -- merged state-dict GB/s of the per-key merge hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic checkpoints that are already resident in
HBM: every key of the workload merged once through the C ABI (batched entry points), per-key flag words
read back.  `value` = (input tensor bytes + output tensor bytes actually merged) / time, summed over ranks
(SURVEY.md §8d "Merged GB/s").  Keys are independent, so ranks never exchange data: each rank merges its own
stripe of the job (weak scaling, no collective on the data path); the only distributed calls are the
barrier around the timed region and the max-over-ranks of the elapsed time.

    python bench.py                                   # N=1, SDXL add_difference+ties_sum of 4 bf16 checkpoints (BASELINE config 2)
    python bench.py --workload flux_navg              # one 1/8 shard of FLUX n_average of 8 checkpoints (config 3, per-GPU share)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import ctypes
import json
import os
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sd_mecha_amd import _lib  # noqa: E402
from sd_mecha_amd.arena import plan_arena, shard_keys_lpt  # noqa: E402
from sd_mecha_amd.shapes import load_manifest, numel  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak

WORKLOADS = {
    # name: (manifest, components, n_models, recipe, shards)
    "sdxl_ties": ("sdxl-sgm", None, 4, "ties", 1),
    "sdxl_navg": ("sdxl-sgm", None, 4, "navg", 1),
    "sdxl_wsum": ("sdxl-sgm", None, 2, "wsum", 1),
    "flux_navg": ("flux-flux", ("diffuser",), 8, "navg", 8),
    "flux_ties": ("flux-flux", ("diffuser",), 8, "ties", 8),
}
DESCRIPTION = {
    "sdxl_ties": "SDXL (sdxl-sgm, 2531 keys, 3.469 B params) add_difference(m0, ties_sum(m1-m0, m2-m0, m3-m0, k=0.2), alpha=1.0) of 4 random-init bf16 checkpoints, bf16 out",
    "sdxl_navg": "SDXL (sdxl-sgm) n_average of 4 random-init bf16 checkpoints, bf16 out",
    "sdxl_wsum": "SDXL (sdxl-sgm) weighted_sum alpha=0.5 of 2 random-init bf16 checkpoints, bf16 out",
    "flux_navg": "FLUX-dev diffuser (780 keys, 11.9 B params) n_average of 8 random-init bf16 checkpoints, keys dealt over 8 shards (LPT), this rank's shard",
    "flux_ties": "FLUX-dev diffuser add_difference(m0, ties_sum(m1-m0 .. m7-m0, k=0.2)) of 8 random-init bf16 checkpoints, keys dealt over 8 shards (LPT), this rank's shard",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sdxl_ties", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time bound of the CPU baseline sample")
    return ap.parse_args()


class Job:
    """One rank's stripe: arenas in HBM + the launch sequence of one step."""

    def __init__(self, workload: str, rank: int, device):
        manifest, comps, n_models, recipe, shards = WORKLOADS[workload]
        items = [(k, numel(s)) for k, s, _ in load_manifest(manifest, comps)]
        if shards > 1:
            items = shard_keys_lpt(items, shards)[rank % shards]
        self.recipe, self.n_models, self.device = recipe, n_models, device
        self.layout = plan_arena(items)
        L = self.layout
        gen = torch.Generator(device=device).manual_seed(1234 + rank)
        self.arenas = []
        for _ in range(n_models):
            a = torch.empty(L.total_elems, dtype=torch.bfloat16, device=device)
            a.normal_(0.0, 0.02, generator=gen)            # random-init weights, N(0, 0.02^2)
            self.arenas.append(a)
        self.out = torch.zeros(L.total_elems, dtype=torch.bfloat16, device=device)
        self.segs_host = L.host_segments()
        self.segs_dev = L.device_segments(device)
        self.flags = torch.zeros(len(L.keys), dtype=torch.int32, device=device)
        self.lib = _lib.lib()
        self.ws = None
        if recipe == "ties":
            nbytes = self.lib.mecha_batched_ties_workspace_bytes(n_models - 1, len(L.keys), L.total_elems)
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.payload_bytes = L.payload_elems * 2 * (n_models + 1)       # bf16 in x n_models + bf16 out
        self.model_ptrs = _lib.ptr_array([a.data_ptr() for a in self.arenas[1:]])
        self.all_ptrs = _lib.ptr_array([a.data_ptr() for a in self.arenas])

    def step(self):
        L, lib = self.layout, self.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.flags.zero_()
        if self.recipe == "ties":
            st = lib.mecha_batched_add_difference_ties(
                self.arenas[0].data_ptr(), self.model_ptrs, self.n_models - 1, self.out.data_ptr(), self.segs_dev.data_ptr(),
                ctypes.addressof(self.segs_host), len(L.keys), L.total_elems, _lib.BF16, _lib.BF16, 0.2, 1.0, 0, None,
                self.ws.data_ptr(), self.ws.numel(), self.flags.data_ptr(), stream)
        elif self.recipe == "navg":
            st = lib.mecha_batched_n_average(self.all_ptrs, self.n_models, self.out.data_ptr(), self.segs_dev.data_ptr(), len(L.keys),
                                             L.total_elems, _lib.BF16, _lib.BF16, self.flags.data_ptr(), stream)
        else:
            st = lib.mecha_batched_weighted_sum(self.arenas[0].data_ptr(), self.arenas[1].data_ptr(), self.out.data_ptr(),
                                                self.segs_dev.data_ptr(), len(L.keys), L.total_elems, _lib.BF16, _lib.BF16, 0.5,
                                                self.flags.data_ptr(), stream)
        _lib.check(st, "batched merge")
        bad = int((self.flags & 1).sum().item())           # per-key skip flags come back to the host (one sync per batch)
        if bad:
            raise RuntimeError(f"{bad} keys flagged non-finite on finite synthetic inputs")


def cpu_baseline(job: Job, seconds: float):
    """The oracle (CPU restatement of the reference path: cast -> method bodies -> cast) timed on this box's host
    cores on a bounded sample of the same workload's keys (every 2nd key, biggest first, until the time bound)."""
    from oracle import oracle as O
    O.build()
    threads = max(1, min(16, os.cpu_count() or 1))
    O.set_threads(threads)
    L = job.layout
    order = sorted(range(len(L.keys)), key=lambda i: -L.numels[i])[::2]
    done_bytes, n_keys, t_total = 0, 0, 0.0
    for i in order:
        o, n = L.offsets[i], L.numels[i]
        bits = [a[o:o + n].view(torch.int16).cpu().numpy().view(np.uint16) for a in job.arenas]
        t0 = time.perf_counter()
        f32 = [O.to_f32(b, "bfloat16") for b in bits]
        if job.recipe == "ties":
            res = O.add_difference_ties(f32[0], f32[1:], k=0.2, alpha=1.0)
        elif job.recipe == "navg":
            res = O.n_average(f32)
        else:
            res = O.weighted_sum(f32[0], f32[1], 0.5)
        ok = all(O.all_finite(x) for x in f32)
        out_bits = O.from_f32(res, "bfloat16")
        t_total += time.perf_counter() - t0
        assert ok and out_bits.size == n
        done_bytes += n * 2 * (job.n_models + 1)
        n_keys += 1
        if t_total > seconds:
            break
    return {"value": round(done_bytes / t_total / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{n_keys} keys ({done_bytes / 1e9:.2f} GB of inputs+outputs, every 2nd key by size, largest first) "
                      f"of the same workload in {t_total:.1f} s; oracle/ C restatement with {threads} OpenMP threads"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"

    job = Job(args.workload, rank, device)
    for _ in range(args.warmup):
        job.step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    _lib.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernels = _lib.profile_read()
    _lib.profile_enable(False)

    total_bytes = torch.tensor([job.payload_bytes * args.steps], dtype=torch.float64, device=device)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_bytes, op=dist.ReduceOp.SUM)
    elapsed_max = float(t_max.item())
    value = float(total_bytes.item()) / elapsed_max / 1e9

    if rank == 0:
        # dominant kernel by summed time; its own algorithmic bytes per launch (DESIGN.md "kernels")
        elems = job.layout.payload_elems
        per_launch_bytes = {
            "streaming_ew": elems * 2 * (job.n_models + 1),
            "ties_hist1": elems * 2 * job.n_models, "ties_hist2": elems * 2 * job.n_models, "ties_hist3": elems * 2 * job.n_models,
            "ties_count": elems * 2 * job.n_models, "ties_merge": elems * 2 * (job.n_models + 1),
        }
        breakdown = {k: {"ms_total": round(ms, 3), "launches": n, "ms_avg": round(ms / n, 4), "ms_per_step": round(ms / args.steps, 4)}
                     for k, (ms, n) in kernels.items()}
        # a kernel family may take several launches per step (the count pass runs once per group of 4 models): the roofline
        # figure is the family's algorithmic bytes per step over its time per step
        dom = max((k for k in kernels if k in per_launch_bytes), key=lambda k: kernels[k][0])
        dom_ms = kernels[dom][0] / args.steps
        dom_launches_per_step = kernels[dom][1] / args.steps
        achieved = per_launch_bytes[dom] / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: they need rocprofv3, so they are collected in separate runs of this
        # same command and committed under profiles/ (how: the file's "_how"); None when this workload was not measured
        traffic = None
        traffic_file = pathlib.Path(__file__).resolve().parent / "profiles" / "r01_hbm_traffic.json"
        if traffic_file.exists():
            traffic = json.loads(traffic_file.read_text()).get(args.workload, {}).get(dom, {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "merged state-dict GB/s", "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "description": DESCRIPTION[args.workload], "keys": len(job.layout.keys),
                       "elements_per_rank": elems, "bytes_per_step_per_rank": job.payload_bytes,
                       "storage_dtype": "bf16 in / bf16 out", "parallelism": f"key stripes x{world}, no collective"},
            "hbm_frac_whole_step": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_launch_bytes[dom], "avg_launch_ms": round(dom_ms, 4),
                         "launches_per_step": dom_launches_per_step},
            "kernels": breakdown,
        }
        if job.recipe == "ties":
            failed, amb = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
            _lib.check(job.lib.mecha_ties_select_stats(job.ws.data_ptr(), job.n_models - 1, len(job.layout.keys), job.layout.total_elems,
                                                       ctypes.byref(failed), ctypes.byref(amb),
                                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ties stats")
            line["ties_select"] = {"pairs": len(job.layout.keys) * (job.n_models - 1), "pairs_fell_back_to_radix": failed.value,
                                   "elements_in_ambiguous_lists": amb.value}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(job, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
