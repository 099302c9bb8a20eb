set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/ramp
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_scheduler.py tests/test_lora_recipes.py -x -q --timeout 300 --timeout-method thread > $O/tests.txt 2>&1; rc=$?
tail -3 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for ramp in 1 0; do
for w in sdxl_ties sdxl_lora flux_ties sdxl_rotate; do
  MECHA_WAVE_RAMP=$ramp timeout -k 10 400 python tools/bench_e2e.py --workload $w --buffer-gib 8 --reps 3 > $O/e2e_${w}_ramp$ramp.json 2> $O/e2e_${w}_ramp$ramp.err || { tail -5 $O/e2e_${w}_ramp$ramp.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/e2e_${w}_ramp$ramp.json')); print('ramp $ramp', '$w', d['seconds'], d['host_link_frac'])"
done
done
