set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_lora.py tests/test_lora_recipes.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/lora_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r2/lora_tests.txt
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
timeout -k 10 400 python tools/bench_e2e.py --workload sdxl_lora --buffer-gib 8 --reps 3 > gpurun_out/r2/e2e_lora.json 2> gpurun_out/r2/e2e_lora.err; rc=$?
cat gpurun_out/r2/e2e_lora.json; tail -4 gpurun_out/r2/e2e_lora.err
exit $rc
