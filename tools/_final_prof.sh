mkdir -p gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof2 -o r1b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/prof2/bench_under_rocprof.txt 2>&1
rm -f gpurun_out/prof2/*kernel_trace.csv
head -12 gpurun_out/prof2/r1b_kernel_stats.csv
timeout 600 python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.json | cut -c1-1500
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d gpurun_out/pmc_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d gpurun_out/pmc_w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_f | grep -A1 "block_" 
python3 tools/pmc_summary.py gpurun_out/pmc_w | grep -A1 "block_"
find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.csv" -size +1M -delete
