#!/usr/bin/env python3
"""Register / spill / LDS / occupancy table of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py maskedsst_amd/csrc/msst_bwd5.hip [filter-substring] [extra hipcc flags ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:])
    sys.exit(1)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("msst::", "")}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} VGPR AGPR  vspill sspill  occ  scratch")
for c in rows:
    if flt and flt not in c["name"]:
        continue
    print(f"{c['name'][:70]:70s} {c.get('VGPRs','?'):>4s} {c.get('AGPRs','?'):>4s} {c.get('VGPRs Spill','?'):>6s} {c.get('SGPRs Spill','?'):>6s} "
          f"{c.get('Occupancy','?'):>4s} {c.get('ScratchSize','?'):>7s}")
