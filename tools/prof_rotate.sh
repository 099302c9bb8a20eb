cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rot -- python3 $R/tools/bench_keys.py --workload sdxl_rotate --key-stride 4 > $R/gpurun_out/rot_under_rocprof.json 2> /tmp/prof_rot.err || { tail -5 /tmp/prof_rot.err; exit 1; }
f=$(find /tmp/prof_rot -name "*kernel_stats.csv" | head -1)
cut -c1-150 "$f" | head -12 > $R/gpurun_out/rot_kernel_stats_head.txt
python3 - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[:12]:
    print(r[0][:90], r[1:4])
PY
