set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py::test_full_config3_stripe_vs_oracle -x -q -s --timeout 600 --timeout-method thread --durations=5 > gpurun_out/r2/new_tests.txt 2>&1; rc=$?
tail -25 gpurun_out/r2/new_tests.txt
exit $rc
