set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_lora.py tests/test_lora_recipes.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/lora_tests.txt 2>&1; rc=$?
tail -3 gpurun_out/r2/lora_tests.txt
[ $rc -ne 0 ] && exit $rc
for run in ${RUNS:-2}; do
for n in 16 1; do
  MECHA_LORA_RUN=$run timeout -k 10 300 python tools/bench_keys.py --workload sdxl_lora --nets $n --steps 5 > gpurun_out/r2/lora_bench_n$n.json 2> gpurun_out/r2/lora_bench_n$n.err || { tail -5 gpurun_out/r2/lora_bench_n$n.err; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/r2/lora_bench_n$n.json')); print('run $run nets $n', d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done
done
