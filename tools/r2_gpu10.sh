set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 600 --timeout-method thread -k "full" > gpurun_out/r2/full_tests.txt 2>&1; rc=$?
tail -15 gpurun_out/r2/full_tests.txt
exit $rc
