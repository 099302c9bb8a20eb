# Round-2 evidence, second part (after the LoRA kernel rewrite): the whole GPU suite, smoke, the config-5 lines and their kernel stats
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/final2
mkdir -p $O
cd $R
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=8 > $O/gpu_tests.txt 2>&1; rc=$?
tail -14 $O/gpu_tests.txt
[ $rc -ne 0 ] && { echo "gpu tests failed"; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -5 $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
for n in 16 1; do
  timeout -k 10 300 python tools/bench_keys.py --workload sdxl_lora --nets $n --steps 5 > $O/lora_bench_n$n.json 2> $O/lora_bench_n$n.err || { tail -5 $O/lora_bench_n$n.err; exit 1; }
done
timeout -k 10 400 python tools/bench_e2e.py --workload sdxl_lora --buffer-gib 8 --reps 3 > $O/e2e_sdxl_lora.json 2> $O/e2e_sdxl_lora.err || { tail -5 $O/e2e_sdxl_lora.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lora -- python3 $R/tools/bench_keys.py --workload sdxl_lora --steps 5 > $O/lora_bench_under_rocprof.json 2> /tmp/prof_lora.err || { echo "stats pass failed"; tail -5 /tmp/prof_lora.err; exit 1; }
cp "$(find /tmp/prof_lora -name '*kernel_stats.csv' | head -1)" $O/lora_kernel_stats.csv
python3 - <<PY
import json
for n in (16, 1):
    d = json.load(open("$O/lora_bench_n%d.json" % n))
    print("lora nets", n, d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"])
d = json.load(open("$O/e2e_sdxl_lora.json")); print("e2e sdxl_lora", d["seconds"], d["host_link_frac"])
PY
head -4 $O/lora_kernel_stats.csv
