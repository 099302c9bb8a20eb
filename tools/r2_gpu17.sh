set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/buf
mkdir -p $O
cd $R
for g in 4 8 16 24; do
  timeout -k 10 400 python tools/bench_e2e.py --workload sdxl_ties --buffer-gib $g --reps 3 > $O/e2e_sdxl_ties_$g.json 2> $O/e2e_sdxl_ties_$g.err || { tail -5 $O/e2e_sdxl_ties_$g.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/e2e_sdxl_ties_$g.json')); print('buffer $g GiB', d['seconds'], d['host_link_frac'])"
  tail -1 $O/e2e_sdxl_ties_$g.err | cut -c1-330
done
