set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_rotate.py -x -q --timeout 600 --timeout-method thread > $O/rotate_tests.txt 2>&1; rc=$?
tail -2 $O/rotate_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/bench_keys.py --workload sdxl_rotate --batched > $O/rotate_batched_x.json 2> $O/rotate_batched_x.err || { tail -3 $O/rotate_batched_x.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/rotate_batched_x.json')); print(d['seconds'], d['roofline']['frac']); [print('  ', g) for g in d['groups'][:7]]"
