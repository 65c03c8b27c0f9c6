"""Per-launch durations of the block kernels by stack from a rocprofv3 --kernel-trace CSV.  A step runs the 12 spatial blocks, then
the 12 spectral ones (and the backward in reverse): launches are grouped per step (steps end with the adamw launch) and printed in
launch order, with the mean of each half.
python tools/per_mode.py "gpurun_out/trace/**/*kernel_trace.csv" """
import csv, sys, glob
paths = [p for a in sys.argv[1:] for p in glob.glob(a, recursive=True)]
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [e for s, e, n in rows if "adamw" in n]
for pat, first, second in (("block_fwd_rs", "spatial", "spectral"), ("block_bwd_attn", "spectral", "spatial"), ("block_bwd_ln1mlp", "spectral", "spatial")):
    print(pat)
    for i in range(1, len(ends)):
        d = [(e - s) / 1e3 for s, e, n in rows if pat in n and ends[i - 1] <= s < ends[i]]
        if len(d) < 2:
            continue
        h = len(d) // 2
        print(f"  step {i}: {first} {sum(d[:h]) / h:6.1f} us, {second} {sum(d[h:]) / (len(d) - h):6.1f} us | " + " ".join(f"{x:.0f}" for x in d))
for i in range(1, len(ends)):
    span = (ends[i] - ends[i - 1]) / 1e6
    ks = sum(e - s for s, e, n in rows if ends[i - 1] <= s < ends[i]) / 1e6
    print(f"step {i}: span {span:.3f} ms, kernels {ks:.3f} ms, gaps {span - ks:.3f} ms")
