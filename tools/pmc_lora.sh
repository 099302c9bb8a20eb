# PMC passes over the LoRA kernel of tools/bench_keys.py --workload sdxl_lora (one pass per run; summaries only come back)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() { n=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-include-regex lora_kernel --output-format csv -d /tmp/pmc_lora/$n -- python3 $R/tools/bench_keys.py --workload sdxl_lora --steps 1 --warmup 0 > /tmp/pmc_lora_$n.log 2>&1 || { echo "pass $n failed"; tail -5 /tmp/pmc_lora_$n.log; }; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
for d in sq1 sq2 tcc fetch write; do python3 $R/tools/pmc_summary.py /tmp/pmc_lora/$d lora_kernel; done > $R/gpurun_out/pmc_lora_summary.txt
