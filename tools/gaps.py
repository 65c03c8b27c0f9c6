"""kernel study: idle time between consecutive kernels of the bench step, from a rocprofv3 kernel trace csv."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# last full step: from the last tokenize_fwd to the following adamw
starts = [i for i, e in enumerate(ev) if "tokenize_fwd" in e[2]]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(ev)) if "adamw" in ev[i][2])
step = ev[i0:i1 + 1]
busy = sum(e[1] - e[0] for e in step)
span = step[-1][1] - step[0][0]
print("kernels in step", len(step), "span us", span / 1e3, "busy us", busy / 1e3, "idle us", (span - busy) / 1e3)
gaps = sorted(((step[i + 1][0] - step[i][1]) / 1e3, step[i][2][:40], step[i + 1][2][:40]) for i in range(len(step) - 1))
print("largest gaps (us):")
for g in gaps[-12:]: print("  %8.1f  %s -> %s" % g)
import collections
by = collections.defaultdict(list)
for i in range(len(step) - 1): by[(step[i][2][:28], step[i + 1][2][:28])].append((step[i + 1][0] - step[i][1]) / 1e3)
print("mean gap by kernel pair:")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:10]: print("  %-60s n=%3d mean %.1f us total %.1f" % (" -> ".join(k), len(v), sum(v) / len(v), sum(v)))
# time between steps
if len(starts) > 1:
    prev_end = ev[i0 - 1][1]
    print("gap before this step (us):", (step[0][0] - prev_end) / 1e3)
