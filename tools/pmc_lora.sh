# SQ / TCC counters of the batched LoRA kernel (SDXL x16 rank-32), one launch; summaries only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
run() { n=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex lora_kernel --output-format csv -d /tmp/pmc_lora/$n -- python3 $R/tools/bench_keys.py --workload sdxl_lora --warmup 0 --steps 1 > /tmp/pmc_lora_$n.log 2>&1 || { echo "pass $n failed"; tail -5 /tmp/pmc_lora_$n.log; }; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
for d in sq1 sq2 sq3 tcc tcp; do python3 $R/tools/pmc_summary.py /tmp/pmc_lora/$d lora_kernel; done > $R/gpurun_out/r2/pmc_lora_summary.txt
cat $R/gpurun_out/r2/pmc_lora_summary.txt
