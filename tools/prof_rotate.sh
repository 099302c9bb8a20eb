# rocprofv3 kernel stats of the batched rotate bench (every >= 2-D SDXL key, one launch sequence per shape group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rot -- python3 $R/tools/bench_keys.py --workload sdxl_rotate --batched --warmup 0 > $R/gpurun_out/r2/rot_under_rocprof.json 2> /tmp/prof_rot.err || { tail -5 /tmp/prof_rot.err; exit 1; }
f=$(find /tmp/prof_rot -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r2/rot_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.reader(open("$f")))
for r in rows[:14]:
    print(r[0][:80], r[1:5])
PY
