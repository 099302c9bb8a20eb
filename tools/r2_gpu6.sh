set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_rotate.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/rotate_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r2/rotate_tests.txt
if [ $rc -ne 0 ]; then echo "rotate tests failed rc=$rc"; exit $rc; fi
bash tools/r2_gpu5.sh
