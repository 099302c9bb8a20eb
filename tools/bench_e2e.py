#!/usr/bin/env python3
"""End-to-end `merge()` timing: safetensors files on tmpfs -> pinned staging -> H2D -> HIP kernels -> D2H -> one output
file (SURVEY.md §8d "Merged GB/s, whole node": input bytes + output bytes over the wall time of merge() from the first
read to close()).  This is the host-link-inclusive number DESIGN.md quotes beside bench.py's HBM-resident `value`.

    python tools/bench_e2e.py --workload sdxl_ties --dir /dev/shm/mecha_e2e

Synthetic checkpoints (N(0, 0.02^2) rounded to bf16, shapes from the exported manifest) are written once into --dir and
removed afterwards.  Also measures the pinned-memory hipMemcpyAsync rate of this box, the roofline of the host link."""
import argparse
import json
import os
import pathlib
import shutil
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import sd_mecha_amd as amd  # noqa: E402
from sd_mecha_amd.shapes import load_manifest, numel  # noqa: E402
from sd_mecha_amd.streaming import OutSafetensorsDict, TensorMetadata  # noqa: E402

WORKLOADS = {
    # name: (manifest, components, n_models, recipe)
    "sdxl_ties": ("sdxl-sgm", None, 4, "ties"),
    "sdxl_navg": ("sdxl-sgm", None, 4, "navg"),
    "sdxl_wsum": ("sdxl-sgm", None, 2, "wsum"),
    "sd1_wsum": ("sd1-ldm", None, 2, "wsum"),
    # BASELINE config 5: one base (kohya key naming) + 16 kohya LoRA files, rank 32, fp16, alpha 16, weight 1.0
    "sdxl_lora": ("sdxl-kohya", None, 1, "lora"),
    # BASELINE config 3, one GPU's share: stripe 0 of the 8 key stripes (LPT) of the FLUX-dev diffuser, 8 checkpoints
    # BASELINE config 4: svd.rotate of model A onto B, every key through the per-key path (merge(threads=...))
    "sdxl_rotate": ("sdxl-sgm", None, 2, "rotate"),
    "flux_navg": ("flux-flux", ("diffuser",), 8, "navg"),
    "flux_ties": ("flux-flux", ("diffuser",), 8, "ties"),
}
STRIPES = {"flux_navg": 8, "flux_ties": 8}
N_LORAS, LORA_RANK = 16, 32


def write_lora(path, manifest, seed, device):
    from sd_mecha_amd.lora import lora_key_prefix
    gen = torch.Generator(device=device).manual_seed(seed)
    tensors = {}
    for k, s, _ in manifest:
        prefix = lora_key_prefix(k, s)
        if prefix is None:
            continue
        rows, cols = s[0], numel(s) // s[0]
        tensors[f"{prefix}.lora_up.weight"] = torch.empty((rows, LORA_RANK), dtype=torch.float16, device=device).normal_(0.0, 0.02, generator=gen).cpu()
        tensors[f"{prefix}.lora_down.weight"] = torch.empty((LORA_RANK,) + tuple(s[1:]), dtype=torch.float16, device=device).normal_(0.0, 0.02, generator=gen).cpu()
        tensors[f"{prefix}.alpha"] = torch.tensor(16.0, dtype=torch.float16)
    sink = OutSafetensorsDict(path, {k: TensorMetadata(tuple(t.shape), t.dtype) for k, t in tensors.items()}, None)
    for k, t in tensors.items():
        sink[k] = t
    sink.close()
    return sum(t.numel() * 2 for t in tensors.values())


def write_checkpoint(path, manifest, seed, device):
    header = {k: TensorMetadata(tuple(s), torch.bfloat16) for k, s, _ in manifest}
    sink = OutSafetensorsDict(path, header, None)
    gen = torch.Generator(device=device).manual_seed(seed)
    for k, s, _ in manifest:
        t = torch.empty(max(numel(s), 1), dtype=torch.bfloat16, device=device).normal_(0.0, 0.02, generator=gen)[:numel(s)]
        sink[k] = t.view(tuple(s)).cpu()
    sink.close()


def link_rate(device, nbytes=1 << 30, reps=5):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sdxl_ties", choices=sorted(WORKLOADS))
    ap.add_argument("--dir", default="/dev/shm/mecha_e2e")
    ap.add_argument("--buffer-gib", type=float, default=16.0, help="total_buffer_size of merge() (pinned staging budget)")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--source", default="files", choices=["files", "pinned"],
                    help="pinned: the inputs are preloaded into pinned host memory before the timed region (SURVEY.md 8d: isolates the "
                         "link from the filesystem); the H2D copies read them where they lie")
    ap.add_argument("--cold", action="store_true",
                    help="release torch's cached pinned blocks before every repetition: each one then pays the first touch of its staging "
                         "buffers like the first merge() of a process")
    ap.add_argument("--threads", type=int, default=None, help="merge(threads=...): worker threads of the per-key path (sdxl_rotate)")
    ap.add_argument("--sink", default="file", choices=["file", "none"],
                    help="file: one .safetensors on --dir; none: results come back to pinned host memory and are dropped "
                         "(the host-link pipeline alone: files -> staging -> H2D -> kernels -> D2H)")
    args = ap.parse_args()
    manifest_id, comps, n_models, recipe_kind = WORKLOADS[args.workload]
    manifest = [(k, s, d) for k, s, d in load_manifest(manifest_id, comps) if not d.startswith("int")]
    if args.workload in STRIPES:
        from sd_mecha_amd.arena import shard_keys_lpt
        mine = {k for k, _ in shard_keys_lpt([(k, numel(s)) for k, s, _ in manifest], STRIPES[args.workload])[0]}
        manifest = [(k, s, d) for k, s, d in manifest if k in mine]
    device = torch.device("cuda", 0)
    work = pathlib.Path(args.dir)
    work.mkdir(parents=True, exist_ok=True)
    try:
        t0 = time.perf_counter()
        paths = []
        for m in range(n_models):
            p = work / f"m{m}.safetensors"
            write_checkpoint(p, manifest, 100 + m, device)
            paths.append(p)
        lora_bytes = 0
        if recipe_kind == "lora":
            lora_paths = []
            for i in range(N_LORAS):
                p = work / f"lora{i}.safetensors"
                lora_bytes += write_lora(p, manifest, 500 + i, device)
                lora_paths.append(p)
        gen_s = time.perf_counter() - t0
        models = [amd.model(str(p)) for p in paths]
        if args.source == "pinned":
            from sd_mecha_amd import streaming
            for p in paths:
                streaming.preload_pinned(p)
        if recipe_kind == "lora":
            recipe = models[0]
            for p in lora_paths:
                recipe = amd.add_difference(recipe, amd.convert(amd.model(str(p)), models[0]), alpha=1.0) | recipe
        elif recipe_kind == "ties":
            deltas = [amd.subtract(m, models[0]) for m in models[1:]]
            recipe = amd.add_difference(models[0], amd.ties_sum(*deltas, k=0.2), alpha=1.0) | models[0]
        elif recipe_kind == "rotate":
            recipe = amd.rotate(models[0], models[1], alignment=1.0, alpha=0.0)
        elif recipe_kind == "navg":
            recipe = amd.n_average(*models)
        else:
            recipe = amd.weighted_sum(models[0], models[1], alpha=0.5)
        elems = sum(numel(s) for _, s, _ in manifest)
        moved = elems * 2 * (n_models + 1) + lora_bytes
        times = []
        for r in range(args.reps):
            out = work / "out.safetensors"
            if out.exists():
                out.unlink()
            if args.cold:
                torch._C._host_emptyCache()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            from sd_mecha_amd import merging
            amd.merge(recipe, output=str(out) if args.sink == "file" else merging.NullSink(), output_dtype=torch.bfloat16,
                      total_buffer_size=int(args.buffer_gib * 2 ** 30), threads=args.threads)
            times.append(time.perf_counter() - t0)
            print("phases (host seconds):", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in merging.last_stats.items()}, file=sys.stderr, flush=True)
        best = min(times)
        line = {
            "metric": "merged state-dict GB/s, end to end (files on tmpfs -> file on tmpfs)" if args.sink == "file" else
                      "merged state-dict GB/s, host-link pipeline (files on tmpfs -> pinned staging -> H2D -> kernels -> D2H -> pinned, no output file)",
            "workload": args.workload, "sink": args.sink, "source": args.source, "threads": args.threads,
            "value": round(moved / best / 1e9, 2), "unit": "GB/s", "seconds": [round(t, 3) for t in times],
            "bytes_moved": moved, "elements": elems, "keys": len(manifest), "n_models": n_models,
            "total_buffer_size_gib": args.buffer_gib, "host_link_GBs": link_rate(device),
            "link_bound_seconds": None, "cpu_count": os.cpu_count(), "generate_seconds": round(gen_s, 1),
        }
        lr = line["host_link_GBs"]
        line["link_bound_seconds"] = round(max(elems * 2 * n_models / (lr["h2d"] * 1e9), elems * 2 / (lr["d2h"] * 1e9)), 3)
        line["host_link_frac"] = round(line["link_bound_seconds"] / best, 4)      # 1.0 = the slower direction of the link busy all the time
        print(json.dumps(line), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
