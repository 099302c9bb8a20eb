#!/usr/bin/env python3
"""bench.py -- merged state-dict GB/s of the per-key merge hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic checkpoints that are already resident in
HBM: every key of the workload merged once through the C ABI (batched entry points), per-key flag words
read back.  `value` = (input tensor bytes + output tensor bytes actually merged) / time, summed over ranks
(SURVEY.md §8d "Merged GB/s").  Keys are independent, so ranks never exchange data: each rank merges its own
stripe of the job (weak scaling, no collective on the data path); the only distributed calls are the
barrier around the timed region and the max-over-ranks of the elapsed time.

After the timed loop the same workload also runs END TO END through `merge()` -- synthetic checkpoints written as
.safetensors files on tmpfs (shaped as SURVEY.md §8d says: integer config keys as int64 arange, empty keys present) ->
pread into pinned staging -> H2D -> kernels -> D2H -> one output file -- and the line carries `e2e`: wall seconds of merge()
from the first read to close(), merged GB/s, H2D / D2H GB/s and the fraction of the measured pinned hipMemcpyAsync rate the
slower direction of the host link was kept busy (`host_link_frac`).  With N ranks the ONE job is sharded over the ranks (keys
dealt LPT, every rank fills its region of the one file): the e2e leg is strong scaling, the HBM-resident `value` weak.

    python bench.py                                   # N=1, SDXL add_difference+ties_sum of 4 bf16 checkpoints (BASELINE config 2)
    python bench.py --workload flux_navg              # one 1/8 shard of FLUX n_average of 8 checkpoints (config 3, per-GPU share)
    python bench.py --gpus N                          # starts its own N ranks (torch.distributed.run) before any GPU call
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import ctypes
import json
import os
import pathlib
import shutil
import socket
import subprocess
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
    # round 3: the rest of north_star's streaming list as batched plans
    "sdxl_slerp": ("sdxl-sgm", None, 2, "slerp", 1),
    "sdxl_addperp": ("sdxl-sgm", None, 3, "addperp", 1),
    "sdxl_addsub": ("sdxl-sgm", None, 3, "addsub", 1),
    "sdxl_dare": ("sdxl-sgm", None, 4, "dare", 1),
}
DESCRIPTION = {
    "sdxl_ties": "SDXL (sdxl-sgm, 2531 keys, 3.469 B params) add_difference(m0, ties_sum(m1-m0, m2-m0, m3-m0, k=0.2), alpha=1.0) of 4 random-init bf16 checkpoints, bf16 out",
    "sdxl_navg": "SDXL (sdxl-sgm) n_average of 4 random-init bf16 checkpoints, bf16 out",
    "sdxl_wsum": "SDXL (sdxl-sgm) weighted_sum alpha=0.5 of 2 random-init bf16 checkpoints, bf16 out",
    "flux_navg": "FLUX-dev diffuser (780 keys, 11.9 B params) n_average of 8 random-init bf16 checkpoints, keys dealt over 8 shards (LPT), this rank's shard",
    "flux_ties": "FLUX-dev diffuser add_difference(m0, ties_sum(m1-m0 .. m7-m0, k=0.2)) of 8 random-init bf16 checkpoints, keys dealt over 8 shards (LPT), this rank's shard",
    "sdxl_slerp": "SDXL (sdxl-sgm) slerp(m0, m1, alpha=0.3) of 2 random-init bf16 checkpoints, per-key norms and dot products, bf16 out",
    "sdxl_addperp": "SDXL (sdxl-sgm) add_difference(m0, perpendicular_component(m0-m2, m1-m2), alpha=1.0) of 3 random-init bf16 checkpoints (wrappers.add_perpendicular), bf16 out",
    "sdxl_addsub": "SDXL (sdxl-sgm) add_difference(m0, m1-m2, alpha=1.0) of 3 random-init bf16 checkpoints (wrappers.add_difference with c), bf16 out",
    "sdxl_dare": "SDXL (sdxl-sgm) add_difference(m0, ties_sum_with_dropout(m1-m0, m2-m0, m3-m0, probability=0.9, k=0.2, apply_stock=True, seed=0), alpha=1.0) of 4 random-init bf16 checkpoints (wrappers.add_ties_with_dare), bf16 out",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sdxl_ties", choices=sorted(WORKLOADS))
    ap.add_argument("--inputs", default="random", choices=["random", "finetune"],
                    help="random: independent N(0, 0.02^2) checkpoints (SURVEY.md 8d).  finetune: model i = bf16(base + delta_i), delta_i ~ N(0, 0.001^2) "
                         "with 30 / 50 / 70 ... %% of its elements exactly zero: what TIES is run on in practice (few distinct |delta| levels, masses of ties and zeros)")
    ap.add_argument("--dare-p", type=float, default=0.9, help="sdxl_dare: drop probability (the workload's 0.9 settles every threshold from the masks; "
                    "below 1 - k = 0.8 the thresholds are positive and the radix-select passes run in full)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time bound of the CPU baseline sample")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end merge() leg (files on tmpfs -> file)")
    ap.add_argument("--e2e-dir", default="/dev/shm", help="where the e2e leg keeps its synthetic checkpoints (tmpfs)")
    ap.add_argument("--e2e-buffer-gib", type=float, default=8.0, help="merge(total_buffer_size=...) of the e2e leg, per rank")
    ap.add_argument("--e2e-reps", type=int, default=2)
    ap.add_argument("--rehearse-staging", action="store_true",
                    help="CPU only: N gloo ranks each run the real wave plan, staging reads (mecha_host_pread_many) and output writer of the "
                         "workload's stripe -- no device copies, no kernels -- and rank 0 reports the aggregate host-side GB/s: what the "
                         "host's DRAM and cores give N ranks before any GPU is involved")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher rehearsal: ranks rendezvous over gloo and report, no GPU work (CPU test of --gpus N)")
    return ap.parse_args()


class Job:
    """One rank's stripe: arenas in HBM + the launch sequence of one step."""

    def __init__(self, workload: str, rank: int, device, inputs: str = "random", dare_p: float = 0.9):
        manifest, comps, n_models, recipe, shards = WORKLOADS[workload]
        self.manifest = load_manifest(manifest, comps)
        items = [(k, numel(s)) for k, s, _ in self.manifest]
        if shards > 1:
            items = shard_keys_lpt(items, shards)[rank % shards]
            mine = {k for k, _ in items}
            self.manifest = [e for e in self.manifest if e[0] in mine]
        self.recipe, self.n_models, self.device = recipe, n_models, device
        self.layout = plan_arena(items)
        L = self.layout
        gen = torch.Generator(device=device).manual_seed(1234 + rank)
        self.arenas = []
        for m in range(n_models):
            a = torch.empty(L.total_elems, dtype=torch.bfloat16, device=device)
            a.normal_(0.0, 0.02, generator=gen)            # random-init weights, N(0, 0.02^2)
            if inputs == "finetune" and m > 0:
                # a fine-tune of model 0: small deltas, a share of them exactly zero, rounded to the storage dtype with the weight
                zero_frac = (0.3, 0.5, 0.7)[(m - 1) % 3]
                step = 1 << 27
                for i in range(0, L.total_elems, step):
                    j = min(i + step, L.total_elems)
                    delta = torch.randn(j - i, device=device, generator=gen) * 0.001
                    delta *= (torch.rand(j - i, device=device, generator=gen) >= zero_frac)
                    a[i:j] = (self.arenas[0][i:j].float() + delta).to(torch.bfloat16)
            self.arenas.append(a)
        self.out = torch.zeros(L.total_elems, dtype=torch.bfloat16, device=device)
        self.segs_host = L.host_segments()
        self.segs_dev = L.device_segments(device)
        self.flags = torch.zeros(len(L.keys), dtype=torch.int32, device=device)
        self.lib = _lib.lib()
        self.ws = None
        if recipe in ("ties", "dare"):
            nbytes = self.lib.mecha_batched_ties_workspace_bytes(n_models - 1, len(L.keys), L.total_elems)
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if recipe in ("slerp", "addperp"):
            self.ws = torch.empty(self.lib.mecha_batched_reduce_workspace_bytes(len(L.keys), L.total_elems), dtype=torch.uint8, device=device)
        if recipe == "dare":
            # the ONE Bernoulli stream every key's masks are cut from (ties.py:29-33), drawn on the host once per job
            self.dare_p, self.dare_seed = dare_p, 0
            self.mask_bits = (n_models - 1) * max(L.numels)
            host = torch.empty((self.mask_bits + 31) // 32 + 1, dtype=torch.int32, pin_memory=True)
            t0 = time.perf_counter()
            _lib.check(self.lib.mecha_dare_draw_stream_host(self.dare_seed, self.dare_p, self.mask_bits, host.data_ptr()), "draw stream")
            self.mask_draw_s = time.perf_counter() - t0
            self.mask_stream = host.to(device)
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
        elif self.recipe == "slerp":
            st = lib.mecha_batched_slerp(self.arenas[0].data_ptr(), self.arenas[1].data_ptr(), self.out.data_ptr(), self.segs_dev.data_ptr(),
                                         ctypes.addressof(self.segs_host), len(L.keys), L.total_elems, _lib.BF16, _lib.BF16, 0.3,
                                         self.ws.data_ptr(), self.ws.numel(), self.flags.data_ptr(), stream)
        elif self.recipe == "addperp":
            st = lib.mecha_batched_add_perpendicular(self.arenas[0].data_ptr(), self.arenas[1].data_ptr(), self.arenas[2].data_ptr(),
                                                     self.out.data_ptr(), self.segs_dev.data_ptr(), ctypes.addressof(self.segs_host), len(L.keys),
                                                     L.total_elems, _lib.BF16, _lib.BF16, 1.0, self.ws.data_ptr(), self.ws.numel(),
                                                     self.flags.data_ptr(), stream)
        elif self.recipe == "addsub":
            st = lib.mecha_batched_add_difference_sub(self.arenas[0].data_ptr(), self.arenas[1].data_ptr(), self.arenas[2].data_ptr(),
                                                      self.out.data_ptr(), self.segs_dev.data_ptr(), len(L.keys), L.total_elems, _lib.BF16,
                                                      _lib.BF16, 1.0, self.flags.data_ptr(), stream)
        elif self.recipe == "dare":
            st = lib.mecha_batched_add_difference_dare(
                self.arenas[0].data_ptr(), self.model_ptrs, self.n_models - 1, self.out.data_ptr(), self.segs_dev.data_ptr(),
                ctypes.addressof(self.segs_host), len(L.keys), L.total_elems, _lib.BF16, _lib.BF16, 0.2, 1.0, 0, self.dare_p, 1,
                self.mask_stream.data_ptr(), self.mask_bits, 0, 0.0, None, None, None, self.ws.data_ptr(), self.ws.numel(), self.flags.data_ptr(), stream)
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
        elif job.recipe == "slerp":
            res = O.slerp(f32[0], f32[1], 0.3)
        elif job.recipe == "addperp":
            res = O.add_perpendicular(f32[0], f32[1], f32[2], 1.0)
        elif job.recipe == "addsub":
            res = O.add_difference(f32[0], O.subtract(f32[1], f32[2]), 1.0)
        elif job.recipe == "dare":
            ties = O.ties_sum_with_dropout([O.subtract(m, f32[0]) for m in f32[1:]], probability=job.dare_p, k=0.2, apply_stock=True, seed=job.dare_seed)
            res = O.add_difference(f32[0], ties, 1.0)
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


# ------------------------------------------------------------------ end to end: files -> merge() -> file
def link_rate(device, nbytes=1 << 30, reps=4):
    """Pinned-memory hipMemcpyAsync rate of this GPU's host link, each direction on its own (the host-link roofline)."""
    h = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(nbytes, dtype=torch.uint8, device=device)
    out = {}
    for name, (dst, src) in {"h2d": (d, h), "d2h": (h, d)}.items():
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[name] = round(nbytes * reps / (time.perf_counter() - t0) / 1e9, 2)
    return out


def write_checkpoints(job: Job, work: pathlib.Path):
    """The job's synthetic checkpoints as .safetensors files: float keys in bf16 from the arenas already in
    HBM (one D2H of each arena into pinned memory, one scatter list per file), integer config keys (position_ids) as int64
    arange, empty keys present."""
    from sd_mecha_amd.streaming import OutSafetensorsDict, TensorMetadata
    L = job.layout
    at = {k: (o, n) for k, o, n in zip(L.keys, L.offsets, L.numels)}
    pinned = torch.empty(L.total_elems, dtype=torch.bfloat16, pin_memory=True)
    paths = []
    for m, arena in enumerate(job.arenas):
        header, float_keys, int_tensors = {}, [], {}
        for key, shape, dtype in job.manifest:
            if dtype.startswith("int"):
                n = numel(shape)
                int_tensors[key] = torch.arange(n, dtype=torch.int64).reshape(shape)
                header[key] = TensorMetadata(shape, torch.int64)
            else:
                header[key] = TensorMetadata(shape, torch.bfloat16)
                float_keys.append(key)
        p = work / f"m{m}.safetensors"
        sink = OutSafetensorsDict(p, header, None)
        pinned.copy_(arena, non_blocking=False)
        base = pinned.data_ptr()
        sink.put_many(float_keys, [header[k] for k in float_keys], [base + at[k][0] * 2 for k in float_keys], 16)
        for key, t in int_tensors.items():
            sink[key] = t
        sink.close()
        paths.append(p)
    return paths


def e2e_leg(job: Job, args, rank: int, world: int, dist, device):
    """merge() of the whole workload, first read to close(), on files the driver can see nothing of but their size."""
    import sd_mecha_amd as amd
    work = pathlib.Path(args.e2e_dir) / f"mecha_bench_{os.environ.get('MASTER_PORT', os.getpid())}"
    need = job.layout.payload_elems * 2 * (job.n_models + 1) * 1.05
    result = None
    try:
        ok = torch.ones(1, device=device)
        if rank == 0:
            work.mkdir(parents=True, exist_ok=True)
            if shutil.disk_usage(work).free < need:
                ok.zero_()
        if dist is not None:
            dist.broadcast(ok, 0)
        if not bool(ok.item()):
            return {"skipped": f"{args.e2e_dir} has less than {need / 1e9:.0f} GB free"}
        t0 = time.perf_counter()
        if rank == 0:
            write_checkpoints(job, work)
        if dist is not None:
            dist.barrier()
        gen_s = time.perf_counter() - t0
        models = [amd.model(str(work / f"m{m}.safetensors")) for m in range(job.n_models)]
        if job.recipe == "ties":
            recipe = amd.add_difference(models[0], amd.ties_sum(*[amd.subtract(m, models[0]) for m in models[1:]], k=0.2), alpha=1.0) | models[0]
        elif job.recipe == "navg":
            recipe = amd.n_average(*models)
        elif job.recipe == "slerp":
            recipe = amd.slerp(models[0], models[1], alpha=0.3)
        elif job.recipe == "addperp":
            recipe = amd.add_difference(models[0], amd.perpendicular_component(amd.subtract(models[0], models[2]), amd.subtract(models[1], models[2])), alpha=1.0)
        elif job.recipe == "addsub":
            recipe = amd.add_difference(models[0], amd.subtract(models[1], models[2]), alpha=1.0) | models[0]
        elif job.recipe == "dare":
            recipe = amd.add_difference(models[0], amd.ties_sum_with_dropout(*[amd.subtract(m, models[0]) for m in models[1:]], probability=job.dare_p,
                                                                             k=0.2, apply_stock=True, seed=job.dare_seed), alpha=1.0)
        else:
            recipe = amd.weighted_sum(models[0], models[1], alpha=0.5)
        from sd_mecha_amd import merging
        out = work / "out.safetensors"
        times, stats, all_stats = [], {}, []
        for _ in range(args.e2e_reps):
            if rank == 0 and out.exists():
                out.unlink()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            amd.merge(recipe, output=str(out), output_dtype=torch.bfloat16, merge_device=str(device),
                      total_buffer_size=int(args.e2e_buffer_gib * 2 ** 30))
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times.append(float(t.item()))
            stats = dict(merging.last_stats)
            all_stats.append(stats)
        link = link_rate(device)
        if rank == 0:
            # a user calls merge() once per process: the FIRST call is the honest number (it pays the first touch of the pinned
            # staging buffers and the first read of the input pages); the best later call is reported beside it as "warm"
            best = times[0]
            warm = min(times[1:]) if len(times) > 1 else None
            in_bytes = sum(numel(s) * (8 if d.startswith("int") else 2) for _, s, d in job.manifest) * job.n_models
            out_bytes = sum(numel(s) for _, s, _ in job.manifest) * 2
            # every rank has its own link: the slower direction, at the measured pinned rate, over the share of one rank
            bound = max(in_bytes / world / (link["h2d"] * 1e9), out_bytes / world / (link["d2h"] * 1e9))
            result = {
                "what": "merge() wall time, first read to close(): .safetensors on tmpfs -> pinned staging -> H2D -> kernels -> D2H -> one .safetensors on tmpfs",
                "seconds": round(best, 4), "all_seconds": [round(x, 4) for x in times], "merged_GBs": round((in_bytes + out_bytes) / best / 1e9, 2),
                "h2d_GBs": round(in_bytes / best / 1e9, 2), "d2h_GBs": round(out_bytes / best / 1e9, 2),
                "link_peak_GBs_per_gpu": link, "host_link_frac": round(bound / best, 4), "sink": "file:tmpfs",
                "seconds_is": "first merge() call of the process (cold)", "warm_seconds": None if warm is None else round(warm, 4),
                "host_link_frac_warm": None if warm is None else round(bound / warm, 4),
                "merged_GBs_warm": None if warm is None else round((in_bytes + out_bytes) / warm / 1e9, 2),
                "scaling": "strong" if world > 1 else "n/a", "n_gpus": world, "keys": len(job.manifest),
                "fused_keys_rank0": stats.get("fused_keys"), "per_key_path_keys_rank0": stats.get("generic_keys"),
                "total_buffer_size_gib_per_rank": args.e2e_buffer_gib, "generate_seconds": round(gen_s, 1),
                "phases_rank0_s": [{k: round(v, 3) for k, v in st.items() if isinstance(v, float)} for st in all_stats],
            }
    finally:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(work, ignore_errors=True)
    return result


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never does) and pass its exit code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def rehearse(args, rank, world):
    """No GPU: the ranks meet over gloo, agree on who they are, rank 0 prints what it saw."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0"))})
    dist.barrier()
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "ranks": seen}), flush=True)
    dist.destroy_process_group()


def rehearse_staging(args, rank, world):
    """Every rank stages and writes a whole stripe of the workload (weak scaling, like bench.py's GPU run); inputs are ONE set of
    files on tmpfs shared by the ranks (their pages are read from DRAM by every rank: 24 GB per FLUX stripe do not fit a cache),
    outputs one file per rank."""
    import torch.distributed as dist
    from sd_mecha_amd import merging, numa
    from sd_mecha_amd.streaming import InSafetensorsDict, OutSafetensorsDict, TensorMetadata
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    manifest_id, comps, n_models, recipe, shards = WORKLOADS[args.workload]
    manifest = [e for e in load_manifest(manifest_id, comps) if not e[2].startswith("int") and numel(e[1]) > 0]
    if shards > 1:
        mine = {k for k, _ in shard_keys_lpt([(k, numel(s)) for k, s, _ in manifest], shards)[0]}
        manifest = [e for e in manifest if e[0] in mine]
    work = pathlib.Path(args.e2e_dir) / f"mecha_rehearsal_{os.environ.get('MASTER_PORT', os.getpid())}"
    header = {k: TensorMetadata(s, torch.bfloat16) for k, s, _ in manifest}
    items = [(k, numel(s)) for k, s, _ in manifest]
    in_bytes = sum(n for _, n in items) * 2 * n_models
    out_bytes = sum(n for _, n in items) * 2
    bound = numa.bind_rank(int(os.environ.get("LOCAL_RANK", rank)), int(os.environ.get("LOCAL_WORLD_SIZE", world)), None) if world > 1 else None
    try:
        if rank == 0:
            work.mkdir(parents=True, exist_ok=True)
            block = torch.empty(1 << 26, dtype=torch.bfloat16).normal_(0.0, 0.02)       # 128 MiB of weights, reused for every tensor
            for m in range(n_models):
                sink = OutSafetensorsDict(work / f"m{m}.safetensors", header, None)
                keys = [k for k, _ in items if _ <= block.numel()]
                sink.put_many(keys, [header[k] for k in keys], [block.data_ptr()] * len(keys), 8)
                for k, n in items:
                    if n > block.numel():
                        sink[k] = block.repeat((n + block.numel() - 1) // block.numel())[:n].reshape(header[k].shape)
                sink.close()
        if world > 1:
            dist.barrier()
        models = [type("M", (), {"path": work / f"m{m}.safetensors"})() for m in range(n_models)]
        readers = {id(m): InSafetensorsDict(m.path) for m in models}
        plan = merging.FusedPlan("navg" if recipe not in ("wsum",) else "wsum", models, {"alpha": 0.5})
        backend = merging.StagingRehearsalBackend()
        times, stats = [], {}
        for rep in range(max(args.e2e_reps, 1)):
            out = work / f"out{rank}.safetensors"
            if out.exists():
                out.unlink()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            merging.last_stats.clear()
            sink = OutSafetensorsDict(out, header, None)
            merging._run_fused(plan, backend, readers, items, torch.bfloat16, header, sink, True, "cpu", torch.bfloat16,
                               int(args.e2e_buffer_gib * 2 ** 30), True, [])
            sink.close()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times.append(float(t.item()))
            stats = dict(merging.last_stats)
        if rank == 0:
            best = min(times)
            print(json.dumps({
                "rehearsal": "staging", "workload": args.workload, "n_ranks": world, "seconds": [round(x, 3) for x in times],
                "per_rank_bytes": {"staged_in": in_bytes, "written_out": out_bytes},
                "aggregate_staging_GBs": round(world * in_bytes / best / 1e9, 2), "aggregate_in_plus_out_GBs": round(world * (in_bytes + out_bytes) / best / 1e9, 2),
                "per_rank_staging_GBs": round(in_bytes / best / 1e9, 2),
                "host_threads_per_rank": merging.default_host_threads(), "cpus_visible": os.cpu_count(), "numa_binding_rank0": bound,
                "pinned_link_rate_needed_per_rank_GBs": 57.6,
                "phases_rank0_s": {k: round(v, 3) for k, v in stats.items() if isinstance(v, float)},
                "what": "real wave plan + staging preads (page cache -> staging buffers) + output writer per rank; no device copies, no kernels"}), flush=True)
        for r in readers.values():
            r.close()
    finally:
        if world > 1:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(work, ignore_errors=True)
        if world > 1:
            dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_staging:
        return rehearse_staging(args, rank, world)
    if args.rehearse:
        return rehearse(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"

    job = Job(args.workload, rank, device, args.inputs, args.dare_p)
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

    e2e = None if args.no_e2e else e2e_leg(job, args, rank, world, dist, device)

    if rank == 0:
        # dominant kernel by summed time; its own algorithmic bytes per launch (DESIGN.md "kernels")
        elems = job.layout.payload_elems
        per_launch_bytes = {
            "streaming_ew": elems * 2 * (job.n_models + 1),
            "ties_hist1": elems * 2 * job.n_models, "ties_hist2": elems * 2 * job.n_models, "ties_hist3": elems * 2 * job.n_models,
            "ties_count": elems * 2 * job.n_models, "ties_merge": elems * 2 * (job.n_models + 1),
            "reduce_partial": elems * 2 * job.n_models, "reduce_apply": elems * 2 * (job.n_models + 1),
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
        for name in ("r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):          # the newest collection that covers this kernel
            traffic_file = pathlib.Path(__file__).resolve().parent / "profiles" / name
            if traffic is None and traffic_file.exists():
                traffic = json.loads(traffic_file.read_text()).get(args.workload, {}).get(dom, {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "merged state-dict GB/s", "value": round(value, 2), "unit": "GB/s",
            # what `value` times (the contract: inputs resident in HBM when the timed region starts); SURVEY.md §8(d)'s own metric, wall
            # time of merge() from the first read to close(), is `e2e.merged_GBs` on this same line
            "value_is": "hbm_resident_loop", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "inputs": args.inputs, "description": DESCRIPTION[args.workload], "keys": len(job.layout.keys),
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
        # the whole step against its algorithm: every input read once, the output written once
        line["whole_step"] = {"algorithmic_bytes": job.payload_bytes, "ms": round(elapsed_max / args.steps * 1e3, 3),
                              "kernel_ms_sum": round(sum(v["ms_per_step"] for v in breakdown.values()), 3)}
        if job.recipe == "dare":
            line["dare_mask_stream"] = {"bits": job.mask_bits, "host_draw_seconds": round(job.mask_draw_s, 3),
                                        "note": "one stream serves every key (the reference seeds a fresh generator per key); drawn once per job, outside the timed steps"}
        if e2e is not None:
            line["e2e"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(job, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
