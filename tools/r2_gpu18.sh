set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/nt
mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_hostio_cpu.py tests/test_gpu_scheduler.py -x -q --timeout 300 --timeout-method thread > $O/tests.txt 2>&1; rc=$?
tail -2 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for nt in 1 0 1 0; do
for w in sdxl_ties sdxl_lora; do
  MECHA_HOST_NT=$nt timeout -k 10 400 python tools/bench_e2e.py --workload $w --buffer-gib 8 --reps 3 > $O/e2e_${w}_nt$nt.json 2> $O/e2e_${w}_nt$nt.err || { tail -5 $O/e2e_${w}_nt$nt.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/e2e_${w}_nt$nt.json')); print('nt $nt', '$w', d['seconds'], d['host_link_frac'])"
  tail -1 $O/e2e_${w}_nt$nt.err | cut -c1-330
done
done
