cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
for w in sdxl_ties flux_ties; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex "ties_" --output-format csv -d /tmp/prof_${w}_$c -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof_$c.log 2>&1 || { echo "pmc pass $c failed"; tail -5 /tmp/prof_$c.log; exit 1; }
  python3 $R/tools/pmc_summary.py /tmp/prof_${w}_$c ties_ > $R/gpurun_out/prof/pmc_${w}_$c.txt
done
done
