# instruction census per dispatch (PMC): bash tools/pmc_insts.sh   (GPU box; prints per-wave-and-tile counts of the block kernels)
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/pmci; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline --no-traffic"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace -f csv -d $OUT/a -- $B > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -f csv -d $OUT/b -- $B > /dev/null 2>&1 || true
( python3 tools/pmc_summary.py $OUT/a; python3 tools/pmc_summary.py $OUT/b ) | grep -A9 "block_bwd_attn\|block_fwd_rs\|ln1mlp" > $OUT/summary.txt
find $OUT -name "*.csv" -size +256k -delete
cat $OUT/summary.txt
