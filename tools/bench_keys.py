#!/usr/bin/env python3
"""Per-key methods on MI355X at the sizes of BASELINE configs 4 and 5, tensors resident in HBM:

    python tools/bench_keys.py --workload sdxl_rotate [--key-stride 8]    # svd.rotate(a, b, alignment=1, alpha=0) on every >=2-D key
    python tools/bench_keys.py --workload sdxl_lora                        # base + sum of 16 rank-32 fp16 LoRA deltas, every >=2-D key

One JSON line per run: merged GB/s (inputs + output of the merged keys over wall time), the MFMA roofline of the
dominant GEMM kernel from the library's HIP-event timers, and for rotate the Jacobi sweep counts and time.
bench.py stays the bench of record for the streaming / TIES path; this tool covers the two MFMA-bound rows."""
import argparse
import json
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from sd_mecha_amd import _lib  # noqa: E402
from sd_mecha_amd import merge_methods as mm  # noqa: E402
from sd_mecha_amd.shapes import load_manifest, numel  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
# the batched rotate's products run as three fp16 MFMA products each (split-fp16 images, rotate.hip gemm128h_kernel): a third of
# the dense f16 peak (~2.5 PFLOP/s) in fp32-equivalent flops
SPLIT_F16_PEAK_TFLOPS = 2500.0 / 3
F16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 / fp16 MFMA peak


def eligible(manifest):
    return [(k, tuple(s)) for k, s, d in manifest if not d.startswith("int") and len(s) >= 2 and numel(s) > 0]


def rotate_flops(n, p, iterations):
    """Flops of the GEMMs rotate.hip runs for one key: tall keys 2np^2 (A^T B) + 2np^2 (Ac R); wide keys five n x n x p
    products; plus 4 q^2 p per Newton-Schulz iteration (Gram matrix + update) on the q x p matrix, q = min(n, p)."""
    q = min(n, p)
    fixed = 4 * n * p * p if n >= p else 10 * n * n * p
    return fixed + iterations * 4 * q * q * p


def polar_step_flops(r, c):
    """One step X <- a X + (b G + c G^2) X on an r x c matrix as rotate.hip runs it: G = X X^T and G G computed on and above
    the diagonal only (128 x 128 tiles, outputs of 256 x 256 and more), the update in full."""
    nt = -(-r // 128)
    f = (nt + 1) / (2 * nt) if r >= 256 else 1.0
    return 2 * r * r * c * f + 2 * r * r * r * f + 2 * r * r * c


def rotate_on_split_kernel(n, p):
    """rotate.hip rotate_run: which keys run on the split-fp16 kernel at alignment 1 / alpha 0"""
    return n % 8 == 0 and p % 8 == 0 and min(n, p) >= 256


def rotate_flops_run(n, p, wide_keys_through_long_schedule=0.0):
    """GEMM flops of one key on the batched path (fp32-equivalent: a product counts once, however many MFMA passes it takes).
    Tall: A^T B, 14 polar steps on p x p, the application.  Wide: two polar solves of 6 steps + the Gram matrix of the
    convergence test (+ 14 steps for the given fraction of keys), and four n x n x p products on the split kernel
    (x (Z - Qb^T)^T is one product there), five on the fp32 one."""
    if n >= p:
        return 4 * n * p * p + 14 * polar_step_flops(p, p)
    nt = -(-n // 128)
    f = (nt + 1) / (2 * nt) if n >= 256 else 1.0
    per_solve = 6 * polar_step_flops(n, p) + 2 * n * n * p * f + wide_keys_through_long_schedule * 14 * polar_step_flops(n, p)
    return (8 if rotate_on_split_kernel(n, p) else 10) * n * n * p + 2 * per_solve


def run_rotate_batched(args, device):
    """The scheduler's form: both checkpoints' >= 2-D keys in two arenas, one mecha_batched_rotate launch sequence per 2-D
    shape (merging.HipBackend._run_rotate).  Times every shape group on its own (a sync per group) and the whole job."""
    import ctypes
    from sd_mecha_amd.arena import plan_arena
    from sd_mecha_amd.merging import HipBackend
    keys = eligible(load_manifest("sdxl-sgm"))[::args.key_stride]
    layout = plan_arena([(k, numel(s)) for k, s in keys])
    gen = torch.Generator(device=device).manual_seed(7)
    a = torch.empty(layout.total_elems, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
    b = torch.empty(layout.total_elems, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
    out = torch.empty(layout.total_elems, dtype=torch.bfloat16, device=device)
    backend = HipBackend(device)
    L = backend.lib
    groups = {}
    for i, (k, shape) in enumerate(keys):
        groups.setdefault(mm.rotate_shape_2d(shape), []).append(i)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    per_group = []
    t_total, flops_total, flops_split = 0.0, 0, 0
    for rep in range(args.warmup + 1):
        per_group, t_total, flops_total, flops_split = [], 0.0, 0, 0
        for (n, p), idx in sorted(groups.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1]):
            offs = torch.tensor([layout.offsets[i] for i in idx], dtype=torch.int64).to(device)
            flags = torch.zeros(len(idx), dtype=torch.int32, device=device)
            per_key = int(L.mecha_batched_rotate_workspace_bytes(n, p, 1))
            chunk = max(1, min(len(idx), HipBackend.ROTATE_WORKSPACE_BYTES // per_key))
            chunk = -(-len(idx) // -(-len(idx) // chunk))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            short_ok = 0
            for c0 in range(0, len(idx), chunk):
                cnt = min(chunk, len(idx) - c0)
                ws = backend._rotate_ws(int(L.mecha_batched_rotate_workspace_bytes(n, p, cnt)))
                _lib.check(L.mecha_batched_rotate(a.data_ptr(), b.data_ptr(), out.data_ptr(), offs[c0:].data_ptr(), cnt, n, p, _lib.BF16, _lib.BF16,
                                                  1.0, 1, 0.0, ws.data_ptr(), ws.numel(), flags[c0:].data_ptr(), stream), "mecha_batched_rotate")
                if n < p:                                       # the per-key done words of the last polar solve head the workspace
                    short_ok += int(ws[:4 * cnt].view(torch.int32).sum().item())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert int((flags & 1).sum().item()) == 0
            # flops actually run: tall keys one 14-step solve; wide keys two solves of 6 steps + the Gram matrix of the test,
            # + 14 more for the keys that failed it (counted from the second solve's words)
            fl = len(idx) * rotate_flops_run(n, p, (len(idx) - short_ok) / len(idx) if n < p else 0.0)
            per_group.append({"shape": [n, p], "keys": len(idx), "ms": round(dt * 1e3, 2), "tflops": round(fl / dt / 1e12, 1),
                              **({"converged_after_short_schedule": short_ok} if n < p else {})})
            t_total += dt
            flops_total += fl
            flops_split += fl if rotate_on_split_kernel(n, p) else 0
    done_bytes = layout.payload_elems * 2 * 3
    return {
        "metric": "merged state-dict GB/s", "value": round(done_bytes / t_total / 1e9, 3), "unit": "GB/s", "n_gpus": 1,
        "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sdxl_rotate", "description": "SDXL svd.rotate(a, b, alignment=1.0, alpha=0.0), every >=2-D key of 2 random-init bf16 checkpoints, "
                   "bf16 out, batched by 2-D shape (one launch sequence per shape)", "keys": len(keys), "key_stride": args.key_stride, "shape_groups": len(groups)},
        "seconds": round(t_total, 3), "keys_per_second": round(len(keys) / t_total, 2),
        "roofline": {"bound": "mfma", "kernel": "rotate_gemm_split_f16_mfma", "achieved": round(flops_total / t_total / 1e12, 2), "peak": round(SPLIT_F16_PEAK_TFLOPS, 1),
                     "unit": "TFLOP/s", "frac": round(flops_total / t_total / 1e12 / SPLIT_F16_PEAK_TFLOPS, 4), "traffic": None,
                     "algorithmic_flops": flops_total, "flops_on_split_kernel_frac": round(flops_split / max(flops_total, 1), 4),
                     "f32_mfma_peak": F32_MFMA_PEAK_TFLOPS,
                     "note": "fp32-equivalent GEMM flops actually run (short polar schedule where a key's convergence word allowed it) over the wall time of "
                     "all launch sequences, norms / centring / image conversions / reductions included; peak = dense f16 MFMA / 3 (every product is three "
                     "f16 MFMA products on split-fp16 operands); keys the split kernel does not take (a side < 256 or not a multiple of 8) run on "
                     "v_mfma_f32_32x32x2_f32"},
        "groups": per_group[:16],
    }


def run_rotate(args, device):
    if args.batched:
        return run_rotate_batched(args, device)
    keys = eligible(load_manifest("sdxl-sgm"))
    keys.sort(key=lambda kv: -numel(kv[1]))
    keys = keys[::args.key_stride]
    sweeps_all, done_bytes, flops, skipped = [], 0, 0, 0
    if args.threads > 1:
        # keys are independent and a key never synchronises with the host inside the Procrustes solve: worker threads, each on
        # its own HIP stream, overlap the latency-bound small GEMMs of different keys (the reference's executor does the same
        # with a torch.cuda.Stream per pool thread, merging.py:171-179, :330-341).  Inputs are made up front.
        import concurrent.futures
        import threading
        gen = torch.Generator(device=device).manual_seed(7)
        data = []
        for k, shape in keys:
            a = torch.empty(shape, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
            b = torch.empty(shape, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
            data.append((shape, a, b))
        torch.cuda.synchronize()
        local = threading.local()

        def work(item):
            shape, a, b = item
            if not hasattr(local, "stream"):
                local.stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(local.stream):
                out, sweeps = mm.rotate(a, b, alignment=1.0, alpha=0.0, out_dtype=torch.bfloat16, return_sweeps=True)
                local.stream.synchronize()
            return shape, sweeps

        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(max_workers=args.threads) as pool:
            results = list(pool.map(work, data))
        torch.cuda.synchronize()
        t_total = time.perf_counter() - t0
        for shape, sweeps in results:
            n, p = mm.rotate_shape_2d(shape)
            sweeps_all.append(sweeps)
            done_bytes += numel(shape) * 2 * 3
            flops += rotate_flops(n, p, sweeps)
        return {
            "metric": "merged state-dict GB/s", "value": round(done_bytes / t_total / 1e9, 3), "unit": "GB/s", "n_gpus": 1,
            "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sdxl_rotate", "description": "SDXL svd.rotate(a, b, alignment=1.0, alpha=0.0) per >=2-D key of 2 random-init bf16 checkpoints, bf16 out",
                       "keys": len(sweeps_all), "key_stride": args.key_stride, "threads": args.threads, "streams": args.threads},
            "seconds": round(t_total, 3), "keys_per_second": round(len(sweeps_all) / t_total, 2),
            "roofline": {"bound": "mfma", "kernel": "rotate_gemm_mfma", "achieved": round(flops / t_total / 1e12, 2),
                         "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flops / t_total / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": None, "algorithmic_flops": flops,
                         "note": "achieved = algorithmic GEMM flops over WALL time of the whole job (kernels of several keys overlap: no per-kernel timers)"},
            "newton_schulz": {"iterations_mean": round(sum(sweeps_all) / max(len(sweeps_all), 1), 2), "iterations_max": max(sweeps_all, default=0)},
        }
    gen = torch.Generator(device=device).manual_seed(7)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t_total = 0.0
    for k, shape in keys:
        n, p = mm.rotate_shape_2d(shape)
        a = torch.empty(shape, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
        b = torch.empty(shape, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, sweeps = mm.rotate(a, b, alignment=1.0, alpha=0.0, out_dtype=torch.bfloat16, return_sweeps=True)
        torch.cuda.synchronize()
        t_total += time.perf_counter() - t0
        sweeps_all.append(sweeps)
        done_bytes += numel(shape) * 2 * 3
        flops += rotate_flops(n, p, sweeps)
        del a, b, out
    kernels = _lib.profile_read()
    _lib.profile_enable(False)
    gemm_ms = kernels.get("rotate_gemm_mfma", (0.0, 1))[0]
    jac_ms = kernels.get("rotate_gram_norms", (0.0, 1))[0]
    achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None
    return {
        "metric": "merged state-dict GB/s", "value": round(done_bytes / t_total / 1e9, 3), "unit": "GB/s", "n_gpus": 1,
        "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sdxl_rotate", "description": "SDXL svd.rotate(a, b, alignment=1.0, alpha=0.0) per >=2-D key of 2 random-init bf16 checkpoints, bf16 out",
                   "keys": len(sweeps_all), "key_stride": args.key_stride, "keys_skipped_too_large": skipped},
        "seconds": round(t_total, 3), "keys_per_second": round(len(sweeps_all) / t_total, 2),
        "roofline": {"bound": "mfma", "kernel": "rotate_gemm_mfma", "achieved": round(achieved, 2) if achieved else None,
                     "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4) if achieved else None,
                     "traffic": None, "algorithmic_flops": flops, "kernel_ms_total": round(gemm_ms, 2)},
        "newton_schulz": {"norms_ms_total": round(jac_ms, 2), "iterations_mean": round(sum(sweeps_all) / max(len(sweeps_all), 1), 2),
                          "iterations_max": max(sweeps_all, default=0)},
        "kernels": {k: {"ms_total": round(ms, 2), "launches": n} for k, (ms, n) in kernels.items()},
    }


def run_lora(args, device):
    keys = eligible(load_manifest("sdxl-sgm"))
    keys = keys[::args.key_stride]
    n_nets, rank = args.nets, args.rank
    gen = torch.Generator(device=device).manual_seed(11)
    from sd_mecha_amd.arena import plan_arena
    layout = plan_arena([(k, numel(s)) for k, s in keys])
    base_arena = torch.empty(layout.total_elems, dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)
    out_arena = torch.zeros(layout.total_elems, dtype=torch.bfloat16, device=device)
    flags = torch.zeros(len(keys), dtype=torch.int32, device=device)
    job, entries = [], []
    for (k, shape), off in zip(keys, layout.offsets):
        rows, cols = shape[0], numel(shape) // shape[0]
        nets = []
        for _ in range(n_nets):
            up = torch.empty((rows, rank), dtype=torch.float16, device=device).normal_(0.0, 0.02, generator=gen)
            down = torch.empty((rank, cols), dtype=torch.float16, device=device).normal_(0.0, 0.02, generator=gen)
            nets.append((up, down, 16.0, 1.0))
        job.append((None, nets, rows, cols))
        entries.append((off, rows, cols, nets))
    batch = mm.LoraBatch(entries, device)

    def step():
        flags.zero_()
        batch.run(base_arena, out_arena, flags)
        if int((flags & 1).sum().item()) and not args.ignore_flags:
            raise RuntimeError("keys flagged non-finite on finite synthetic inputs")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / args.steps
    kernels = _lib.profile_read()
    _lib.profile_enable(False)
    flops = sum(2 * r * c * rank * n_nets for _, _, r, c in job)
    moved = sum(r * c * 2 * 2 + n_nets * (r + c) * rank * 2 for _, _, r, c in job)
    ms, launches = kernels.get("lora_mfma", (0.0, 1))
    per_step_ms = ms / args.steps
    achieved = flops / (per_step_ms * 1e-3) / 1e12 if per_step_ms else None
    traffic, traffic_note = None, None
    for name in ("r03_hbm_traffic.json",):                      # PMC counters of this same command, collected under rocprofv3 (profiles/README.md)
        f = ROOT / "profiles" / name
        if f.exists() and args.key_stride == 1 and (n_nets, rank) == (16, 32):
            e = json.loads(f.read_text()).get("sdxl_lora", {}).get("lora_mfma", {})
            traffic, traffic_note = e.get("hbm_bytes_per_launch"), e.get("_note")
    return {
        "metric": "merged state-dict GB/s", "value": round(moved / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sdxl_lora", "description": f"SDXL base (bf16) + sum of {n_nets} rank-{rank} fp16 LoRA deltas (alpha=16, weight 1.0) on every >=2-D key, bf16 out",
                   "keys": len(job), "key_stride": args.key_stride, "bytes_per_step": moved},
        "roofline": {"bound": "hbm", "kernel": "lora_mfma", "achieved": round(moved / (per_step_ms * 1e-3) / 1e9, 1) if per_step_ms else None,
                     "peak": 8000.0, "unit": "GB/s", "frac": round(moved / (per_step_ms * 1e-3) / 1e9 / 8000.0, 4) if per_step_ms else None,
                     "traffic": traffic, "traffic_note": traffic_note, "kernel_ms_per_step": round(per_step_ms, 3),
                     "mfma_TFLOPs": round(achieved, 2) if achieved else None, "mfma_frac_of_dense_f16_peak": round(achieved / F16_MFMA_PEAK_TFLOPS, 4) if achieved else None,
                     "note": "rank-32 reconstruction has 2*32/(2*2) = 16 flop per output byte: far below the MFMA ridge, so HBM bounds it"},
        "kernels": {k: {"ms_total": round(v[0], 2), "launches": v[1]} for k, v in kernels.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["sdxl_rotate", "sdxl_lora"])
    ap.add_argument("--key-stride", type=int, default=1, help="take every n-th eligible key (largest first for rotate)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nets", type=int, default=16, help="sdxl_lora: networks per key (BASELINE config 5: 16)")
    ap.add_argument("--rank", type=int, default=32, help="sdxl_lora: rank of every network (BASELINE config 5: 32)")
    ap.add_argument("--ignore-flags", action="store_true", help="timing experiments with deliberately broken kernels only")
    ap.add_argument("--threads", type=int, default=1, help="sdxl_rotate: worker threads, each with its own HIP stream")
    ap.add_argument("--batched", action="store_true", help="sdxl_rotate: one mecha_batched_rotate launch sequence per 2-D shape (the scheduler's form)")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs a GPU"
    device = torch.device("cuda", 0)
    line = run_rotate(args, device) if args.workload == "sdxl_rotate" else run_lora(args, device)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
