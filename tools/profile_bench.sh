# rocprofv3 evidence for bench.py's default workload (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench run (kernel averages, to compare with bench.py's HIP-event timers)
#   2. FETCH_SIZE and WRITE_SIZE of the TIES kernels, one counter per run
# Only the summaries are kept under gpurun_out/ (the raw traces stay in /tmp).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2> /tmp/prof_stats.err || { echo "stats pass failed"; tail -5 /tmp/prof_stats.err; exit 1; }
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/prof/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex "ties_" --output-format csv -d /tmp/prof_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof_$c.log 2>&1 || { echo "pmc pass $c failed"; tail -5 /tmp/prof_$c.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/prof_$c ties_ > $R/gpurun_out/prof/pmc_$c.txt
done
