set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_lora.py tests/test_lora_recipes.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/lora_tests.txt 2>&1; rc=$?
tail -15 gpurun_out/r2/lora_tests.txt
[ $rc -ne 0 ] && exit $rc
for n in 16 1 4; do
  timeout -k 10 300 python tools/bench_keys.py --workload sdxl_lora --nets $n --steps 5 > gpurun_out/r2/lora_bench_n$n.json 2> gpurun_out/r2/lora_bench_n$n.err || { tail -5 gpurun_out/r2/lora_bench_n$n.err; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/r2/lora_bench_n$n.json')); print('nets $n', {k: d[k] for k in d if k in ('ms_per_step','value','unit','roofline')})"
done
