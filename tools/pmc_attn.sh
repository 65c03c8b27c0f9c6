# SQ / LDS counters of the attention backward (and the forward) from one bench step: bash tools/pmc_attn.sh   (on the GPU box)
set -u
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline"
rm -rf gpurun_out/pmca; mkdir -p gpurun_out/pmca
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -f csv -d gpurun_out/pmca/a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace -f csv -d gpurun_out/pmca/b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace -f csv -d gpurun_out/pmca/c -- $B > /dev/null 2>&1
( python3 tools/pmc_summary.py gpurun_out/pmca/a; python3 tools/pmc_summary.py gpurun_out/pmca/b; python3 tools/pmc_summary.py gpurun_out/pmca/c ) | grep -A8 "bwd_attn\|fwd_hw\|bwd_mlp" > gpurun_out/pmc_attn.txt
find gpurun_out/pmca -name "*.csv" -size +512k -delete
cat gpurun_out/pmc_attn.txt
