# branch / scalar / total instruction census of the block kernels (PMC): bash tools/pmc_branch.sh   (GPU box)
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/pmcb; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline --no-traffic"
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace -f csv -d $OUT/a -- $B > /dev/null 2>&1 || true
python3 tools/pmc_summary.py $OUT/a | grep -A9 "block_bwd_attn\|block_fwd_rs\|ln1mlp" > $OUT/summary.txt
find $OUT -name "*.csv" -size +256k -delete
cat $OUT/summary.txt
