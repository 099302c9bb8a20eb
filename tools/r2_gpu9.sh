# TIES: parity tests, then the HBM-resident bench lines of the two TIES workloads
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/parity_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r2/parity_tests.txt
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for w in sdxl_ties flux_ties; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 --no-e2e --no-cpu-baseline > gpurun_out/r2/bench_$w.json 2> gpurun_out/r2/bench_$w.err; rc=$?
  if [ $rc -ne 0 ]; then echo "bench $w failed rc=$rc"; tail -5 gpurun_out/r2/bench_$w.err; exit $rc; fi
  python - <<PY
import json
d = json.load(open("gpurun_out/r2/bench_$w.json"))
print("$w", d["value"], "GB/s", d["ms_per_step"], "ms", d["roofline"]["kernel"], d["roofline"]["frac"], {k: v["ms_per_step"] for k, v in d["kernels"].items() if v["ms_per_step"] > 0.05})
PY
done
