cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 1 > /tmp/s.json 2> /tmp/s.err || { tail -5 /tmp/s.err; exit 1; }
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.reader(open("$f")))
for r in rows[1:14]:
    print(r[0].replace("(anonymous namespace)::","")[:70], r[1], r[3][:9])
PY
