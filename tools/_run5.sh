mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof -o r1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/prof/bench_under_rocprof.txt 2>&1
ls -R gpurun_out/prof | head -30
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
head -25 "$f"
rm -f gpurun_out/prof/*kernel_trace.csv gpurun_out/prof/*.db
