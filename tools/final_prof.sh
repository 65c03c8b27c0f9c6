# round-end evidence: rocprofv3 kernel stats of the default bench command, the default bench line, PMC passes
# (HBM traffic; SQ/LDS counters), other configurations, the box microbenchmark, the CU-contention sweep.
# Everything lands in gpurun_out/final/; `python tools/make_profiles.py $TAG` then copies the judged summaries into profiles/.
# usage (on the GPU box, through gpurun):  bash tools/final_prof.sh r03
set -eu
TAG="${1:?usage: final_prof.sh rNN}"
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/final
rm -rf "$OUT"; mkdir -p "$OUT"   # (on the box; the build container's own gpurun_out/final keeps files of earlier rounds that this pass does not write: make_profiles.py only takes files newer than the pass's bench_default.json minus an hour)
export MSST_ROUND="$TAG" MSST_RECORD=1 MSST_STRICT_PARITY=1   # strict tier: every recorded error within 1.5x the committed baseline (tests/util.py)
# parity measurements of the round (tests/util.py::record appends to gpurun_out/parity_$TAG.jsonl)
rm -f "gpurun_out/parity_$TAG.jsonl"
python3 -m pytest tests -m gpu -q --no-header 2>&1 | tail -3 > $OUT/gpu_tests.txt || true; cat $OUT/gpu_tests.txt
# box microbenchmark (MFMA / HBM / L2 / LDS access patterns)
unset MSST_STRICT_PARITY
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/peak_microbench.hip -o /tmp/peak_microbench && /tmp/peak_microbench > $OUT/peak_microbench.json
export TMPDIR=/tmp
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline --no-traffic --no-alt --no-probe"
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $pass --kernel-trace -f csv -d $OUT/pmc_$pass -- $B > /dev/null 2>&1 || true
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -f csv -d $OUT/pmc_sq1 -- $B > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace -f csv -d $OUT/pmc_sq2 -- $B > /dev/null 2>&1 || true
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE > $OUT/pmc_fetch.txt
python3 tools/pmc_summary.py $OUT/pmc_WRITE_SIZE > $OUT/pmc_write.txt
( python3 tools/pmc_summary.py $OUT/pmc_sq1; python3 tools/pmc_summary.py $OUT/pmc_sq2 ) > $OUT/pmc_sq.txt
grep -A1 "block_" $OUT/pmc_fetch.txt || true; grep -A1 "block_" $OUT/pmc_write.txt || true
find $OUT -name "*.csv" -size +512k -delete
# the per-launch HBM traffic table bench.py quotes (roofline.traffic, hbm_bound_kernels) comes from these passes: write it in
# place BEFORE the bench lines below are taken, so that the committed line and the committed table belong to the same kernels
python3 tools/make_profiles.py "$TAG" --traffic-only
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o "$TAG" -- python3 bench.py --no-cpu-baseline --no-traffic > $OUT/bench_under_rocprof.txt 2>&1 || true
find $OUT/stats -name "*kernel_trace.csv" -delete
timeout 900 python3 bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err || true
tail -1 $OUT/bench_default.json | cut -c1-600
timeout 600 python3 bench.py --no-cpu-baseline --no-traffic --no-alt --profile-all > $OUT/bench_profile_all.json 2>/dev/null || true
# other configurations (parity-test shapes and the batch sweep), one line each
( python3 bench.py --no-cpu-baseline --no-traffic --dropout 0 | tail -1
  python3 bench.py --no-cpu-baseline --no-traffic --bands 50 | tail -1
  python3 bench.py --no-cpu-baseline --no-traffic --batch 64 | tail -1
  python3 bench.py --no-cpu-baseline --no-traffic --batch 1024 --steps 5 | tail -1
  python3 bench.py --no-cpu-baseline --no-traffic --precision fp32 --steps 3 --warmup 1 | tail -1 ) > $OUT/bench_other_configs.jsonl 2>/dev/null || true
cut -c1-200 $OUT/bench_other_configs.jsonl
# CU contention: the step with N occupancy-probe workgroups held on a side stream, default grid and the DP grid
# static partition (single-GPU default), static partition on grids sized for 32 free CUs (round 3's DP choice; with and without a
# probe), and the dynamic tile queue (round 4's DP choice: no reservation)
( python3 bench.py --no-cpu-baseline --no-traffic --no-pipeline --no-profile --no-alt | tail -1
  for n in 8 16 32; do python3 bench.py --no-cpu-baseline --no-traffic --no-pipeline --no-profile --no-alt --cu-thief $n | tail -1; done
  python3 bench.py --no-cpu-baseline --no-traffic --no-pipeline --no-profile --no-alt --cu-thief 1 --thief-us 1 --thief-reserve 32 | tail -1
  for n in 8 16 32; do python3 bench.py --no-cpu-baseline --no-traffic --no-pipeline --no-profile --no-alt --cu-thief $n --thief-reserve 32 | tail -1; done
  python3 bench.py --no-cpu-baseline --no-traffic --no-pipeline --no-profile --no-alt --tile-queue | tail -1
  for n in 8 16 32 48; do python3 bench.py --no-cpu-baseline --no-traffic --no-pipeline --no-profile --no-alt --cu-thief $n --tile-queue | tail -1; done ) > $OUT/cu_contention.jsonl 2>/dev/null || true
python3 - << 'PY'
import json
for l in open("gpurun_out/final/cu_contention.jsonl"):
    d = json.loads(l); print(d.get("cu_thief"), d["ms_per_step"], d["value"])
PY
# data-parallel wiring on one GPU: a one-rank RCCL group, bucket hooks fired by the real backward (bench.py --force-dp)
rocprofv3 --kernel-trace -f csv -d $OUT/dp -- python3 bench.py --force-dp --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-pipeline --no-traffic --no-alt > $OUT/dp_bench.txt 2>&1 || true
python3 tools/dp_overlap.py $OUT/dp > $OUT/dp_overlap.txt 2>&1 || true; cat $OUT/dp_overlap.txt
find $OUT/dp -name "*.csv" -size +512k -delete
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gate_microbench.hip -o /tmp/gate_microbench && timeout 120 /tmp/gate_microbench > $OUT/gate_microbench.jsonl || true
# one launch per stack against one per block (the engine picks by tiles per workgroup), both forms alternating in one process
( python3 tools/fwd_ab.py | tail -1; python3 tools/fwd_ab.py --batch 64 | tail -1; python3 tools/fwd_ab.py --bands 50 | tail -1 ) > $OUT/fwd_stack_ab.txt 2>/dev/null || true
cat $OUT/fwd_stack_ab.txt
# q / k / v handed to the backward through HBM instead of recomputed: kernel-study library built next to the product one (tools/gate_qkv.py)
if [ -f maskedsst_amd/libmsst_lab.so ]; then timeout 300 python3 tools/gate_qkv.py 2>/dev/null | grep -v Warning > $OUT/gate_qkv.txt || true; cat $OUT/gate_qkv.txt; fi
cp "gpurun_out/parity_$TAG.jsonl" $OUT/parity_measured.jsonl 2>/dev/null || true
ls -la $OUT | head -40
