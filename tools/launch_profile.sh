# per-launch durations of the block kernels by position in the step (rocprofv3 kernel trace): bash tools/launch_profile.sh   (GPU box)
set -u; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/ltrace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-pipeline --no-traffic > $OUT/out.txt 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ltrace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    k = "fwd" if "block_fwd_rs" in n else "attn" if "block_bwd_attn" in n else "lnmlp" if "ln1mlp" in n else "adamw" if "adamw" in n else None
    if k: seq.append((k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# split into steps at adamw
steps, cur = [], []
for k, d in seq:
    if k == "adamw":
        steps.append(cur); cur = []
    else: cur.append((k, d))
steps = [s for s in steps if len(s) == 71][2:]   # 24 forwards + 24 attention backwards + 23 fused launches; drop the warmup steps
for kind in ("fwd", "attn", "lnmlp"):
    per = collections.defaultdict(list)
    for s in steps:
        for i, (k, d) in enumerate([x for x in s if x[0] == kind]): per[i].append(d)
    print(kind, " ".join("%.0f" % (sum(v) / len(v)) for i, v in sorted(per.items())))
PY
find $OUT -name "*.csv" -size +256k -delete
