#!/usr/bin/env python3
"""SURVEY.md §8d "CPU baseline beside it" (i): time the REAL reference (`sd_mecha.merge`, imported from /root/reference)
and this repo's CPU restatement (the oracle behind the key scheduler, tests/oracle_backend.py) on the same synthetic
checkpoints, in the build container.  Shows that the restatement `bench.py` times on the GPU box (`cpu_baseline`) is a
faithful stand-in for the reference's own CPU path.  Never runs on the GPU box (the reference cannot travel).

    python tools/cpu_reference_vs_oracle.py --config 1            # sd1-ldm without ema x2, weighted_sum alpha=0.5, fp16
    python tools/cpu_reference_vs_oracle.py --config 2 --stride 10  # every 10th sdxl-sgm key x4, TIES k=0.2, bf16
    python tools/cpu_reference_vs_oracle.py --config 3n --stride 40 # every 40th FLUX diffuser key x8, n_average (3t: TIES, 7 deltas)
    python tools/cpu_reference_vs_oracle.py --config 4 --stride 10  # every 10th sdxl-sgm key x2, svd.rotate(alignment=1, alpha=0)
    python tools/cpu_reference_vs_oracle.py --config 5 --stride 10  # every 10th sdxl-kohya key + 16 rank-32 kohya LoRA files
"""
import argparse
import json
import os
import pathlib
import shutil
import sys
import time
import types

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def import_reference():
    fw = types.ModuleType("fuzzywuzzy")
    fwp = types.ModuleType("fuzzywuzzy.process")
    fwp.extractOne = lambda q, choices, **k: (next(iter(choices), None), 0)     # only decorates error messages
    fw.process = fwp
    sys.modules["fuzzywuzzy"] = fw
    sys.modules["fuzzywuzzy.process"] = fwp
    sys.path.insert(0, "/root/reference")
    import sd_mecha
    from sd_mecha.extensions.builtin.merge_methods import linear, svd, ties
    return sd_mecha, linear, ties, svd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="1", choices=["1", "2", "3n", "3t", "4", "5"])
    ap.add_argument("--stride", type=int, default=1, help="take every n-th key (config 2: 10 = the 10 %% subset of SURVEY §8d)")
    ap.add_argument("--dir", default="/tmp/mecha_cpu_cmp")
    ap.add_argument("--loras", type=int, default=16, help="config 5: number of LoRA files (the reference's time grows steeply with it)")
    args = ap.parse_args()
    from sd_mecha_amd.shapes import load_manifest, numel
    from sd_mecha_amd.streaming import OutSafetensorsDict, TensorMetadata
    import sd_mecha_amd as amd
    from oracle_backend import OracleBackend

    n_loras = 0
    if args.config == "1":
        manifest = [(k, s, d) for k, s, d in load_manifest("sd1-ldm") if d.startswith("float") and not k.startswith("model_ema.")]
        n_models, dtype = 2, torch.float16
    elif args.config in ("3n", "3t"):
        manifest = [(k, s, d) for k, s, d in load_manifest("flux-flux", ("diffuser",)) if not d.startswith("int")]
        n_models, dtype = 8, torch.bfloat16
    elif args.config == "5":
        manifest = [(k, s, d) for k, s, d in load_manifest("sdxl-kohya") if d.startswith("float")]
        n_models, dtype, n_loras = 1, torch.bfloat16, args.loras
    else:
        manifest = [(k, s, d) for k, s, d in load_manifest("sdxl-sgm") if d.startswith("float")]
        n_models, dtype = (4 if args.config == "2" else 2), torch.bfloat16
    if args.stride > 1:                                     # a spread over the key sizes, not whatever sits at every n-th position
        manifest = sorted(manifest, key=lambda e: numel(e[1]))[args.stride // 2::args.stride]
    work = pathlib.Path(args.dir)
    work.mkdir(parents=True, exist_ok=True)
    try:
        gen = torch.Generator().manual_seed(5)
        paths = []
        for m in range(n_models):
            p = work / f"m{m}.safetensors"
            sink = OutSafetensorsDict(p, {k: TensorMetadata(tuple(s), dtype) for k, s, _ in manifest}, None)
            for k, s, _ in manifest:
                sink[k] = (torch.randn(max(numel(s), 1), generator=gen) * 0.02).to(dtype)[:numel(s)].view(tuple(s))
            sink.close()
            paths.append(p)
        elems = sum(numel(s) for _, s, _ in manifest)
        moved = elems * 2 * (n_models + 1)
        lora_paths = []
        if n_loras:
            import safetensors.torch
            from sd_mecha_amd.lora import lora_key_prefix
            for i in range(n_loras):
                sd = {}
                for k, s, _ in manifest:
                    prefix = lora_key_prefix(k, s)
                    if prefix is None:
                        continue
                    sd[f"{prefix}.lora_up.weight"] = (torch.randn(s[0], 32, generator=gen) * 0.02).half()
                    sd[f"{prefix}.lora_down.weight"] = (torch.randn((32,) + tuple(s[1:]), generator=gen) * 0.02).half()
                    sd[f"{prefix}.alpha"] = torch.tensor(16.0).half()
                safetensors.torch.save_file(sd, str(work / f"lora{i}.safetensors"))
                lora_paths.append(work / f"lora{i}.safetensors")
                moved += sum(t.numel() * 2 for t in sd.values())

        sd_mecha, linear, ties, svd = import_reference()
        from tqdm import tqdm

        class quiet(tqdm):
            def __init__(self, *a, **k):
                k["disable"] = True
                super().__init__(*a, **k)

        # a key subset is too small for the reference's config inference: name the config
        ref_config = {"1": "sd1-ldm", "2": "sdxl-sgm", "3n": "flux-flux", "3t": "flux-flux", "4": "sdxl-sgm", "5": "sdxl-kohya"}[args.config]
        rm = [sd_mecha.model(p, config=ref_config) for p in paths]
        if args.config == "1":
            ref_recipe = linear.weighted_sum(rm[0], rm[1], alpha=0.5)
        elif args.config == "3n":
            ref_recipe = linear.n_average(*rm)
        elif args.config == "4":
            ref_recipe = svd.rotate(rm[0], rm[1], alignment=1.0, alpha=0.0)
        elif args.config == "5":
            ref_recipe = rm[0]
            for lp in lora_paths:
                ref_recipe = linear.add_difference(ref_recipe, sd_mecha.convert(sd_mecha.model(lp), rm[0]), alpha=1.0) | ref_recipe
        else:
            ref_recipe = linear.add_difference(rm[0], ties.ties_sum(*[linear.subtract(x, rm[0]) for x in rm[1:]], k=0.2), alpha=1.0) | rm[0]
        out_ref = work / "out_ref.safetensors"
        t0 = time.perf_counter()
        sd_mecha.merge(ref_recipe, output=out_ref, merge_device="cpu", output_dtype=dtype, tqdm=quiet)
        t_ref = time.perf_counter() - t0

        om = [amd.model(str(p)) for p in paths]
        if args.config == "1":
            our_recipe = amd.weighted_sum(om[0], om[1], alpha=0.5)
        elif args.config == "3n":
            our_recipe = amd.n_average(*om)
        elif args.config == "4":
            our_recipe = amd.rotate(om[0], om[1], alignment=1.0, alpha=0.0)
        elif args.config == "5":
            our_recipe = om[0]
            for lp in lora_paths:
                our_recipe = amd.add_difference(our_recipe, amd.convert(amd.model(str(lp)), om[0]), alpha=1.0) | our_recipe
        else:
            our_recipe = amd.add_difference(om[0], amd.ties_sum(*[amd.subtract(x, om[0]) for x in om[1:]], k=0.2), alpha=1.0) | om[0]
        from oracle import oracle as O
        O.build()
        O.set_threads(os.cpu_count() or 1)
        out_our = work / "out_oracle.safetensors"
        t0 = time.perf_counter()
        amd.merge(our_recipe, output=str(out_our), merge_device="cpu", output_dtype=dtype, backend=OracleBackend(), total_buffer_size=2 ** 31)
        t_our = time.perf_counter() - t0

        # same bytes?  (the file layouts differ -- tensor order -- so compare tensor by tensor)
        from sd_mecha_amd.streaming import InSafetensorsDict
        a, b = InSafetensorsDict(out_ref), InSafetensorsDict(out_our)
        keys = sorted(set(a.keys()) & set(b.keys()))
        differing = 0
        worst = 0
        worst_rel = 0.0
        worst_abs = 0.0                                     # largest absolute difference anywhere
        worst_big = 0                                       # largest distance in ulps among elements that are not cancellation results
        for k in keys:
            x, y = a[k].view(torch.int16).numpy().astype(np.int32), b[k].view(torch.int16).numpy().astype(np.int32)
            # sign-aware distance in units of the output dtype: map the sign-magnitude patterns onto one ordered line
            x, y = np.where(x < 0, -(x & 0x7fff), x), np.where(y < 0, -(y & 0x7fff), y)
            d = np.abs(x - y)
            if d.size and d.max() > 0:
                differing += 1
                worst = max(worst, int(d.max()))
                xf, yf = a[k].float().flatten().numpy(), b[k].float().flatten().numpy()
                worst_abs = max(worst_abs, float(np.abs(xf - yf).max()))
                big = np.abs(xf) >= 1e-5                    # inputs are ~N(0, 0.02): below this a sum is the remainder of a cancellation
                if big.any():
                    worst_big = max(worst_big, int(d.flatten()[big].max()))
                if os.environ.get("MECHA_CMP_VERBOSE"):
                    i = int(d.argmax())
                    print(k, tuple(a[k].shape), "differing", int((d > 0).sum()), "worst", int(d.max()),
                          float(a[k].flatten()[i]), float(b[k].flatten()[i]), file=sys.stderr)
                if args.config == "4":                       # rotate: the bar is relative Frobenius error, not bits (LAPACK fp32 against fp64)
                    xf, yf = a[k].float().numpy().astype(np.float64), b[k].float().numpy().astype(np.float64)
                    worst_rel = max(worst_rel, float(np.linalg.norm(xf - yf) / max(np.linalg.norm(xf), 1e-300)))
        print(json.dumps({
            "config": args.config, "stride": args.stride, **({"loras": n_loras} if n_loras else {}), "keys": len(manifest), "elements": elems, "bytes_moved": moved,
            "reference_seconds": round(t_ref, 2), "reference_GBs": round(moved / t_ref / 1e9, 3),
            "oracle_scheduler_seconds": round(t_our, 2), "oracle_GBs": round(moved / t_our / 1e9, 3),
            "torch_threads": torch.get_num_threads(), "cpu_count": os.cpu_count(),
            "keys_in_both_outputs": len(keys), "keys_in_reference_output": len(list(a.keys())),
            "keys_with_any_bit_difference": differing, "worst_difference_in_output_ulps": worst,
            "worst_absolute_difference": worst_abs, "worst_difference_in_ulps_where_abs_reference_ge_1e-5": worst_big,
            **({"worst_relative_frobenius_difference": worst_rel} if args.config == "4" else {}),
        }), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
