#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter rows per kernel: python tools/pmc_summary.py <dir> [kernel-substring]."""
import collections
import csv
import pathlib
import sys

root = pathlib.Path(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for f in root.rglob("*counter_collection.csv"):
    with open(f, newline="") as fh:
        seen = set()
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if want not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]
            acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (short, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                launches[short] += 1
for k, counters in acc.items():
    print(k, "launches", launches[k])
    for c, v in sorted(counters.items()):
        print(f"    {c:32s} {v:.6g}   per launch {v / launches[k]:.6g}")
