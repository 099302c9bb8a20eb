# scheduler tests + the default bench line (HBM-resident loop + e2e leg)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_scheduler.py tests/test_gpu_lora.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/sched_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r2/sched_tests.txt
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py --no-cpu-baseline --e2e-reps 3 > gpurun_out/r2/bench_e2e.json 2> gpurun_out/r2/bench_e2e.err; rc=$?
python - <<PY
import json
d = json.load(open("gpurun_out/r2/bench_e2e.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["e2e"]))
PY
exit $rc
