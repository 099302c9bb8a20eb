#!/bin/bash
# Kernel experiments: builds of ONE source file with tuning macros, linked into complete libraries under build_variants/
# (git-ignored; travels to the GPU box), and a box-side loop that benches each of them.
#
# here:        bash tools/variants.sh build ties.hip base "" v1 "-DMECHA_COUNT_BLOCKS_PER_CU=8" ...
# on the box:  bash tools/variants.sh run "<bench command that prints one JSON line>" base v1 ...
#              (copies each variant over the box's scratch copy of sd_mecha_amd/libmecha_hip.so, restores the default at the end)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/build_variants
C=$R/sd_mecha_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-bitwise-instead-of-logical"
case "$1" in
  build)
    src=$2; shift 2
    mkdir -p "$V"
    make -C "$C" -j8 > /dev/null || exit 1
    pids=()
    while [ $# -gt 1 ]; do
      name=$1; defs=$2; shift 2
      ( /opt/rocm/bin/hipcc $FLAGS $defs -c "$C/$src" -o "$V/$name.o" 2> "$V/$name.err" &&
        others=$(ls "$C"/build/*.o | grep -v "/$src.o") &&
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$V/lib_$name.so" $others "$V/$name.o" && rm -f "$V/$name.o" && echo "built $name" ) &
      pids+=($!)
    done
    rc=0; for p in "${pids[@]}"; do wait "$p" || rc=1; done
    grep -l "error" "$V"/*.err 2>/dev/null | while read -r f; do echo "== $f"; grep -m3 error "$f"; done
    exit $rc ;;
  run)
    cmd=$2; shift 2
    O=$R/gpurun_out/variants${VTAG:+_$VTAG}; mkdir -p "$O"
    cp "$R/sd_mecha_amd/libmecha_hip.so" /tmp/libmecha_default.so
    for name in "$@"; do
      cp "$V/lib_$name.so" "$R/sd_mecha_amd/libmecha_hip.so"
      ( cd "$R" && timeout -k 10 300 bash -c "$cmd" > "$O/$name.json" 2> "$O/$name.err" )
      rc=$?
      if [ $rc -ne 0 ]; then echo "$name: exit $rc"; tail -3 "$O/$name.err"; cp /tmp/libmecha_default.so "$R/sd_mecha_amd/libmecha_hip.so"; exit $rc; fi
      python3 - "$name" "$O/$name.json" <<'EOF'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
avg = {n: (v["ms_avg"] if "ms_avg" in v else v["ms_total"] / max(v["launches"], 1)) for n, v in k.items()}
print(sys.argv[1], "ms/step", d.get("ms_per_step"), {n: round(a, 4) for n, a in avg.items() if a > 0.05})
EOF
    done
    cp /tmp/libmecha_default.so "$R/sd_mecha_amd/libmecha_hip.so" ;;
  *) echo "usage: variants.sh build|run ..."; exit 2 ;;
esac
