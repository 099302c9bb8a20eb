# end-to-end: FLUX x8 stripes (config 3 at one GPU's share) and SDXL TIES without an output file
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
for w in flux_navg flux_ties; do
  timeout -k 10 400 python tools/bench_e2e.py --workload $w --buffer-gib 8 --reps 3 > gpurun_out/r2/e2e_$w.json 2> gpurun_out/r2/e2e_$w.err || { tail -5 gpurun_out/r2/e2e_$w.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/r2/e2e_$w.json"))
print("$w", d["seconds"], d["value"], d["host_link_frac"])
PY
  tail -1 gpurun_out/r2/e2e_$w.err
done
timeout -k 10 400 python tools/bench_e2e.py --workload sdxl_ties --buffer-gib 8 --reps 3 --sink none > gpurun_out/r2/e2e_sdxl_ties_nosink.json 2> gpurun_out/r2/e2e_sdxl_ties_nosink.err || exit 1
python - <<PY
import json
d = json.load(open("gpurun_out/r2/e2e_sdxl_ties_nosink.json"))
print("sdxl_ties nosink", d["seconds"], d["value"], d["host_link_frac"])
PY
