# batched rotate bench (kernel side), per shape group
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
timeout -k 10 400 python tools/bench_keys.py --workload sdxl_rotate --batched "$@" > gpurun_out/r2/rotate_batched.json 2> gpurun_out/r2/rotate_batched.err; rc=$?
tail -3 gpurun_out/r2/rotate_batched.err
[ $rc -ne 0 ] && exit $rc
python - <<PY
import json
d = json.load(open("gpurun_out/r2/rotate_batched.json"))
print(d["seconds"], "s", d["roofline"]["achieved"], "TFLOP/s", d["roofline"]["frac"])
for g in d["groups"]:
    print(g)
PY
