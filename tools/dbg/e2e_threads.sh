cd $GRAFT_REPO_ROOT
for t in 8 12 24 32; do
  MECHA_HOST_THREADS=$t timeout -k 10 300 python bench.py --no-cpu-baseline --e2e-reps 3 --steps 2 --warmup 1 > /tmp/b_$t.json 2>/dev/null || exit 1
  python - <<PY
import json
d = json.load(open("/tmp/b_$t.json"))["e2e"]
print("threads $t", d["all_seconds"], d["host_link_frac"], d["phases_rank0_s"])
PY
done
