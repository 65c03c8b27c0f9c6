# per-kernel rocprofv3 stats of a short default bench run: bash tools/kstats.sh [bench args]   (on the GPU box)
set -eu; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
rm -rf gpurun_out/kstats; mkdir -p gpurun_out/kstats
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/kstats -o ks -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-profile "$@" > gpurun_out/kstats/out.txt 2>&1
tail -1 gpurun_out/kstats/out.txt | cut -c1-160
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/kstats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/kstats -name "*kernel_trace.csv" -delete
