cd $GRAFT_REPO_ROOT
for b in 16 40; do
timeout -k 10 300 python tools/bench_e2e.py --workload sdxl_rotate --buffer-gib $b --reps 3 > /tmp/r.json 2> /tmp/r.err || { tail -5 /tmp/r.err; exit 1; }
python - <<PY
import json
d = json.load(open("/tmp/r.json")); print("buffer $b", d["seconds"], d["value"])
PY
tail -1 /tmp/r.err
done
