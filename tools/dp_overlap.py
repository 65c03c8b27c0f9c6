"""Summarise a rocprofv3 kernel trace of `bench.py --force-dp`: which collective kernels ran, on which queue, and which
msst backward kernels were running at the same time (overlap of the gradient all-reduce with the backward)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no kernel trace found"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qkey, "?") if qkey else "?") for r in rows)
coll = [e for e in ev if any(s in e[2].lower() for s in ("nccl", "rccl", "allreduce", "all_reduce"))]
ours = [e for e in ev if "msst" in e[2]]
print(f"kernels traced: {len(ev)}; msst kernels: {len(ours)}; collective kernels: {len(coll)}")
print("queues used by msst kernels:", sorted(set(e[3] for e in ours)))
if not coll:
    print("no collective kernel was launched: with ONE rank RCCL completes an in-place all-reduce without a device kernel, so a")
    print("1-GPU box cannot show the overlap; the calls themselves (ranges, order, once per bucket) are checked by")
    print("tests/test_gpu_scripts.py::test_dp_wiring_single_rank_rccl and the 2-rank gloo test.")
    sys.exit(0)
print("queues used by collective kernels:", sorted(set(e[3] for e in coll)))
tot = over = 0
by = collections.Counter()
for s, e, name, q in coll:
    tot += e - s
    for s2, e2, n2, q2 in ours:
        lo, hi = max(s, s2), min(e, e2)
        if hi > lo:
            over += hi - lo
            by[n2.split("(")[0][-48:]] += hi - lo
print(f"collective kernel time {tot / 1e3:.1f} us, of which {over / 1e3:.1f} us overlapped msst kernels")
for k, v in by.most_common(8):
    print(f"   {v / 1e3:9.1f} us under {k}")
