# Round-2 evidence run: the whole GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the same command, FETCH / WRITE
# counters of the streaming and TIES kernels (one counter per run), the HBM-resident lines of the other workloads.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/final
mkdir -p $O
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > $O/gpu_tests.txt 2>&1; rc=$?
tail -4 $O/gpu_tests.txt
[ $rc -ne 0 ] && { echo "gpu tests failed"; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -5 $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
for w in sdxl_navg flux_navg flux_ties sdxl_wsum; do
  timeout -k 10 300 python bench.py --workload $w --no-e2e --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail -5 $O/bench_$w.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2> /tmp/prof_stats.err || { echo "stats pass failed"; tail -5 /tmp/prof_stats.err; exit 1; }
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
for w in sdxl_ties flux_ties sdxl_navg; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-include-regex "ties_|ew_kernel" --output-format csv -d /tmp/prof_${w}_$c -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /tmp/prof_$c.log 2>&1 || { echo "pmc pass $w $c failed"; tail -5 /tmp/prof_$c.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/prof_${w}_$c > $O/pmc_${w}_$c.txt
done
done
python3 $R/tools/make_traffic_json.py $O > $O/hbm_traffic.json
python3 - <<PY
import json
for w in ("default", "sdxl_navg", "flux_navg", "flux_ties", "sdxl_wsum"):
    d = json.load(open("$O/bench_%s.json" % w))
    print(w, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], (d.get("e2e") or {}).get("seconds"), (d.get("e2e") or {}).get("host_link_frac"))
PY
