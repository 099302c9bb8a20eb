cd $GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 200 python bench.py --workload sdxl_ties --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > /tmp/b.json 2>/dev/null || { echo fail; return; }
python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("$*", d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items() if k in ("ties_count",)})
PY
}
run X=0
run MECHA_COUNT_DBG=1
run MECHA_COUNT_GRID=1024
run MECHA_COUNT_GRID=4096
run MECHA_COUNT_GRID=8192
run MECHA_COUNT_GRID=16384
run MECHA_COUNT_DBG=1 MECHA_COUNT_GRID=8192
