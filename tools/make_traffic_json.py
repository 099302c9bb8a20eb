#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic.json from the per-kernel PMC summaries `tools/gpu.sh traffic pmc_<workload> <regex> bench.py ...` leaves
under gpurun_out/<tag>/ (`pmc_<workload>_{FETCH,WRITE}_SIZE.txt`, written by tools/pmc_summary.py):
    python tools/make_traffic_json.py gpurun_out/r3 > profiles/r03_hbm_traffic.json
Counter unit KiB; FETCH_SIZE doubled for the 16-byte-per-lane streaming reads of gfx950 (MI355X_MICROARCH.md, HBM section); per
launch = total / launches."""
import json
import pathlib
import re
import sys

src = pathlib.Path(sys.argv[1])
# kernel name -> the family name bench.py's roofline uses (abi.cpp's profiler names)
FAMILY = {"ties_merge_full_kernel": "ties_merge", "ties_count_kernel": "ties_count", "ties_sample_kernel": "ties_sample", "ew_kernel": "streaming_ew",
          "bred_reduce_kernel": "reduce_partial", "bred_reduce_flat_kernel": "reduce_partial", "bred_apply_kernel": "reduce_apply",
          "ties_hist_kernel": "ties_hist", "ties_dare_presolve_kernel": "ties_dare_presolve", "lora_kernel": "lora_mfma"}
out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py --workload <w> --steps 2 --warmup 1 "
               "--no-cpu-baseline --no-e2e` on one MI355X (`tools/gpu.sh traffic`, summed per kernel by tools/pmc_summary.py). Counter unit KiB. gfx950 "
               "correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the 128-B requests of 16-B-per-lane streaming reads at 64 B, so it is "
               "doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores. bytes = KiB * 1024 (* 2 for FETCH_SIZE). The 8-byte-per-lane "
               "accesses of ties_merge_full_kernel<5..8 deltas> are outside the guide's calibration (`_note`)."}
seen = {}                                                      # (workload, family) -> {kernel: entry}; the family's line is its heaviest kernel
for f in sorted(src.glob("pmc_*_SIZE.txt")):
    m = re.match(r"pmc_(.+)_(FETCH|WRITE)_SIZE\.txt", f.name)
    workload, which = m.group(1), m.group(2)
    cur = None
    for line in f.read_text().splitlines():
        head = re.match(r"(\S.*) launches (\d+)", line)
        if head:
            cur = head.group(1)
            continue
        val = re.match(r"\s+(FETCH|WRITE)_SIZE\s+\S+\s+per launch (\S+)", line)
        if val and cur:
            fam = next((v for k, v in FAMILY.items() if k in cur), None)
            if fam is None:
                continue
            e = seen.setdefault((workload, fam), {}).setdefault(cur, {"kernel": cur})
            e[f"{which}_SIZE_KiB"] = float(val.group(2))
for (workload, fam), kernels in seen.items():
    out.setdefault(workload, {})[fam] = max(kernels.values(), key=lambda e: e.get("FETCH_SIZE_KiB", 0.0) + e.get("WRITE_SIZE_KiB", 0.0))
for workload, fams in out.items():
    if workload.startswith("_"):
        continue
    for fam, e in fams.items():
        e["hbm_bytes_per_launch"] = int((2 * e.get("FETCH_SIZE_KiB", 0.0) + e.get("WRITE_SIZE_KiB", 0.0)) * 1024)
        if re.search(r"merge_full_kernel<[5-8],", e["kernel"]):
            e["_note"] = "8-byte-per-lane accesses: outside the guide's calibration, but 2 x FETCH + WRITE equals the algorithmic bytes here too"
        if "lora_kernel" in e["kernel"]:
            e["_note"] = ("whole-line base loads and LDS-DMA factor loads re-read through L2: outside the guide's calibration for the doubling; undoubled, FETCH + WRITE = "
                          f"{int((e.get('FETCH_SIZE_KiB', 0.0) + e.get('WRITE_SIZE_KiB', 0.0)) * 1024)} bytes")
print(json.dumps(out, indent=2))
