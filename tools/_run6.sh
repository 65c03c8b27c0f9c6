mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --dropout 0"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -f csv -d gpurun_out/pmc/p1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --kernel-trace -f csv -d gpurun_out/pmc/p2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA --kernel-trace -f csv -d gpurun_out/pmc/p3 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INST_LEVEL_SMEM --kernel-trace -f csv -d gpurun_out/pmc/p4 -- $B > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary_cur.txt
grep -A40 "block_bwd_attn_kernel" gpurun_out/pmc_summary_cur.txt | head -44
find gpurun_out/pmc -name "*.csv" -size +1M -delete
