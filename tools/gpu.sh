#!/bin/bash
# The one GPU-box runner (replaces the per-experiment scripts of rounds 1-2).  Run from the repo root on the box:
#     gpurun --timeout 900 -- 'bash tools/gpu.sh <step> [args] && bash tools/gpu.sh <step> ...'
# Every step runs under its own time limit and writes under gpurun_out/<tag>/ (tag = $MECHA_TAG, default r3); a failing step
# ends the script with its exit code, so steps chained with && stop at the first failure.
#
#   tests [pytest args]             the GPU suite (-m gpu), one process
#   smoke                           __graft_entry__.smoke()
#   bench <name> [bench.py args]    python bench.py ... > <name>.json
#   keys  <name> [bench_keys args]  python tools/bench_keys.py ... > <name>.json
#   e2e   <name> [bench_e2e args]   python tools/bench_e2e.py ... > <name>.json
#   stats <name> <prog> [args]      rocprofv3 --kernel-trace --stats of `python3 <prog> args` -> <name>_kernel_stats.csv (+ the JSON line)
#   pmc   <name> <regex> <prog-and-args ...> -- <counters ...>
#                                   one rocprofv3 --pmc pass (<= 8 SQ_, 4 TCC_ counters per pass) over kernels matching <regex>;
#                                   the per-kernel sums are APPENDED to <name>.txt
#   traffic <name> <regex> <prog-and-args ...>
#                                   FETCH_SIZE and WRITE_SIZE, one pass each -> <name>_FETCH_SIZE.txt / <name>_WRITE_SIZE.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${MECHA_TAG:-r3}
mkdir -p "$O"
step=$1; shift
case "$step" in
  tests)
    cd "$R" && timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=8 "$@" > "$O/gpu_tests.txt" 2>&1
    rc=$?; tail -15 "$O/gpu_tests.txt"; exit $rc ;;
  smoke)
    cd "$R" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1
    rc=$?; tail -3 "$O/smoke.txt"; exit $rc ;;
  bench|keys|e2e)
    name=$1; shift
    prog=bench.py; [ "$step" = keys ] && prog=tools/bench_keys.py; [ "$step" = e2e ] && prog=tools/bench_e2e.py
    cd "$R" && timeout -k 10 600 python -u $prog "$@" > "$O/$name.json" 2> "$O/$name.err"
    rc=$?; [ $rc -ne 0 ] && tail -8 "$O/$name.err"; tail -c 1500 "$O/$name.json"; echo; exit $rc ;;
  stats)
    name=$1; prog=$2; shift 2
    cd /tmp && export TMPDIR=/tmp && rm -rf "/tmp/prof_$name"
    timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$name" -- python3 "$R/$prog" "$@" > "$O/${name}_under_rocprof.json" 2> "/tmp/prof_$name.err"
    rc=$?; [ $rc -ne 0 ] && { tail -5 "/tmp/prof_$name.err"; exit $rc; }
    cp "$(find "/tmp/prof_$name" -name '*kernel_stats.csv' | head -1)" "$O/${name}_kernel_stats.csv" && head -12 "$O/${name}_kernel_stats.csv" ;;
  pmc)
    name=$1; regex=$2; shift 2
    cmd=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
    cd /tmp && export TMPDIR=/tmp && rm -rf "/tmp/pmc_$name"
    timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex "$regex" --output-format csv -d "/tmp/pmc_$name" -- python3 "$R/${cmd[0]}" "${cmd[@]:1}" > "/tmp/pmc_$name.log" 2>&1
    rc=$?; [ $rc -ne 0 ] && { tail -5 "/tmp/pmc_$name.log"; exit $rc; }
    python3 "$R/tools/pmc_summary.py" "/tmp/pmc_$name" | tee -a "$O/$name.txt" ;;
  traffic)
    name=$1; regex=$2; shift 2
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf "/tmp/pmc_${name}_$c"
      timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex "$regex" --output-format csv -d "/tmp/pmc_${name}_$c" -- python3 "$R/$1" "${@:2}" > "/tmp/pmc_${name}_$c.log" 2>&1
      rc=$?; [ $rc -ne 0 ] && { tail -5 "/tmp/pmc_${name}_$c.log"; exit $rc; }
      python3 "$R/tools/pmc_summary.py" "/tmp/pmc_${name}_$c" | tee "$O/${name}_$c.txt"
    done ;;
  *) echo "unknown step: $step"; exit 2 ;;
esac
