set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_scheduler.py tests/test_gpu_rotate.py tests/test_lora_recipes.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/sched_tests.txt 2>&1; rc=$?
tail -4 gpurun_out/r2/sched_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_e2e.py --workload sdxl_rotate --reps 3 > gpurun_out/r2/e2e_rotate.json 2> gpurun_out/r2/e2e_rotate.err || { tail -5 gpurun_out/r2/e2e_rotate.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r2/e2e_rotate.json')); print('rotate', d['seconds'])"; tail -1 gpurun_out/r2/e2e_rotate.err
timeout -k 10 400 python bench.py --no-cpu-baseline --e2e-reps 3 --steps 3 > gpurun_out/r2/bench_e2e.json 2> gpurun_out/r2/bench_e2e.err || exit 1
python -c "import json; d=json.load(open('gpurun_out/r2/bench_e2e.json'))['e2e']; print('sdxl ties e2e', d['all_seconds'], d['host_link_frac'], d['phases_rank0_s'])"
