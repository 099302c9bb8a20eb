set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_lora_recipes.py tests/test_gpu_lora.py tests/test_gpu_scheduler.py -x -q --timeout 300 --timeout-method thread > $O/lora_tests.txt 2>&1; rc=$?
tail -3 $O/lora_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/bench_e2e.py --workload sdxl_lora --buffer-gib 8 --reps 4 > $O/e2e_sdxl_lora.json 2> $O/e2e_sdxl_lora.err || { tail -5 $O/e2e_sdxl_lora.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/e2e_sdxl_lora.json')); print('e2e sdxl_lora', d['seconds'])"
tail -2 $O/e2e_sdxl_lora.err | cut -c1-400
