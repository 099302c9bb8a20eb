# round 2: GPU suite, default bench line, rocprofv3 kernel stats + FETCH/WRITE of the default bench (current kernels)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r2/gpu_tests.txt
if [ $rc -ne 0 ]; then echo "gpu tests failed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err; rc=$?
tail -c 1500 gpurun_out/r2/bench_default.json; tail -3 gpurun_out/r2/bench_default.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2/prof
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e > $R/gpurun_out/r2/prof/bench_under_rocprof.json 2> /tmp/prof_stats.err || { echo "stats pass failed"; tail -5 /tmp/prof_stats.err; exit 1; }
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r2/prof/kernel_stats.csv
head -8 "$f" | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex "ties_" --output-format csv -d /tmp/prof_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /tmp/prof_$c.log 2>&1 || { echo "pmc pass $c failed"; tail -5 /tmp/prof_$c.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/prof_$c ties_ > $R/gpurun_out/r2/prof/pmc_$c.txt
done
cat $R/gpurun_out/r2/prof/pmc_FETCH_SIZE.txt
