# round 2, first GPU call: the whole GPU suite, the default bench (with the new e2e leg), the stream placement sweep
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r2/gpu_tests.txt
if [ $rc -ne 0 ]; then echo "gpu tests failed rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err; rc=$?
tail -c 3000 gpurun_out/r2/bench_default.json; tail -3 gpurun_out/r2/bench_default.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
mkdir -p /tmp/sb && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/sb/stream_bench3 tools/stream_bench3.hip && timeout -k 10 300 /tmp/sb/stream_bench3 > gpurun_out/r2/stream_bench3.txt 2>&1; rc=$?
tail -30 gpurun_out/r2/stream_bench3.txt
exit $rc
