"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0][-60:]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    if "msst" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
