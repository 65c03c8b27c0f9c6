"""Per-launch durations of the block kernels by mode (spatial / spectral blocks alternate) from a rocprofv3 --kernel-trace CSV:
python tools/per_mode.py gpurun_out/trace/**/t_kernel_trace.csv"""
import csv, sys, glob, collections
paths = [p for a in sys.argv[1:] for p in glob.glob(a, recursive=True)]
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
seq = collections.defaultdict(list)
for s, e, n, g in rows:
    for pat in ("block_bwd_attn", "block_fwd_rs", "block_bwd_ln1mlp"):
        if pat in n:
            seq[pat].append((e - s) / 1e3)
for pat, d in seq.items():
    ev, od = d[0::2], d[1::2]
    n = len(d)
    tail = d[n // 2:]   # second half: the timed steps, past the cold first one
    print(pat, "launches", n, "even-index avg %.1f us" % (sum(tail[0::2]) / max(1, len(tail[0::2]))), "odd-index avg %.1f us" % (sum(tail[1::2]) / max(1, len(tail[1::2]))),
          "first 8:", [round(x) for x in d[:8]], "last 8:", [round(x) for x in d[-8:]])
