set -eu; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
rm -rf gpurun_out/gaps; mkdir -p gpurun_out/gaps
rocprofv3 --kernel-trace --memory-copy-trace -f csv -d gpurun_out/gaps -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/gaps/out.txt 2>&1
tail -1 gpurun_out/gaps/out.txt | cut -c1-200
python3 tools/gaps.py gpurun_out/gaps | head -3
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/gaps/**/*kernel_trace.csv",recursive=True)[0]
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f)))
st=[e[0] for e in ev if "tokenize_fwd" in e[2]]
print("step-to-step (ms):",[round((b-a)/1e6,2) for a,b in zip(st[:-1],st[1:])])
m=glob.glob("gpurun_out/gaps/**/*memory_copy_trace.csv",recursive=True)
if m:
    rows=list(csv.DictReader(open(m[0])))
    print("memcpy rows",len(rows), rows[-1] if rows else None)
PY
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c1-200
rm -rf gpurun_out/gaps
