# rotate: GPU tests of the batched path, then the e2e rotate bench
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_rotate.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/rotate_tests.txt 2>&1; rc=$?
tail -25 gpurun_out/r2/rotate_tests.txt
if [ $rc -ne 0 ]; then echo "rotate tests failed rc=$rc"; exit $rc; fi
timeout -k 10 300 python tools/bench_e2e.py --workload sdxl_rotate --threads 4 --reps 2 > gpurun_out/r2/e2e_rotate.json 2> gpurun_out/r2/e2e_rotate.err; rc=$?
cat gpurun_out/r2/e2e_rotate.json; tail -5 gpurun_out/r2/e2e_rotate.err
exit $rc
