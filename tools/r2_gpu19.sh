set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $O/parity_tests.txt 2>&1; rc=$?
tail -2 $O/parity_tests.txt
[ $rc -ne 0 ] && exit $rc
for w in sdxl_ties sdxl_ties flux_ties flux_ties; do
timeout -k 10 300 python bench.py --workload $w --no-e2e --no-cpu-baseline --steps 20 > $O/bench_count.json 2> $O/bench_count.err || { tail -5 $O/bench_count.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_count.json')); k=d['kernels']; print('$w', d['value'], d['ms_per_step'], {n: k[n]['ms_per_step'] for n in ('ties_merge','ties_count','ties_sample','ties_fixup')})"
done
