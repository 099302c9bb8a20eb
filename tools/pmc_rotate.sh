# SQ counters of the rotate GEMM (gemm128_kernel) on every 4th SDXL key, batched; summaries only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
run() { n=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex gemm128 --output-format csv -d /tmp/pmc_rot/$n -- python3 $R/tools/bench_keys.py --workload sdxl_rotate --batched --warmup 0 --key-stride 4 > /tmp/pmc_rot_$n.log 2>&1 || { echo "pass $n failed"; tail -5 /tmp/pmc_rot_$n.log; }; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA
for d in sq1 sq2; do python3 $R/tools/pmc_summary.py /tmp/pmc_rot/$d gemm128; done > $R/gpurun_out/r2/pmc_rotate_summary.txt
cat $R/gpurun_out/r2/pmc_rotate_summary.txt
