# SQ counters of the TIES count and merge kernels (bench.py default workload); summaries only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-include-regex "ties_count|ties_merge_full" --output-format csv -d /tmp/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_sq.log 2>&1 || { tail -5 /tmp/pmc_sq.log; exit 1; }
python3 $R/tools/pmc_summary.py /tmp/pmc_sq ties_ > $R/gpurun_out/prof/pmc_sq_ties.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-include-regex "ties_count|ties_merge_full" --output-format csv -d /tmp/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_sq2.log 2>&1 || { tail -5 /tmp/pmc_sq2.log; exit 1; }
python3 $R/tools/pmc_summary.py /tmp/pmc_sq2 ties_ >> $R/gpurun_out/prof/pmc_sq_ties.txt
