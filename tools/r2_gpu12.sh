set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_lora.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/lora_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r2/lora_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_keys.py --workload sdxl_lora > gpurun_out/r2/lora_bench.json 2> gpurun_out/r2/lora_bench.err || { tail -5 gpurun_out/r2/lora_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r2/lora_bench.json"))
print(d.get("ms_per_step"), d["value"], d["roofline"])
PY
