#!/usr/bin/env python3
"""Static instruction census of a kernel instance from the compiler's assembly (VERDICT r5 item 2 iii): how many of each class the
ISA carries -- MFMA, other VALU, LDS, vector memory, scalar ALU, scalar memory, branches, s_waitcnt, s_nop, barriers -- for the whole
kernel and for its tile loop (the largest loop by instruction count).  Static counts: every instruction once, whatever its role
branch -- the per-wave dynamic numbers come from the PMC passes (tools/pmc_branch.sh).
usage: python tools/isa_census.py maskedsst_amd/csrc/msst_bwd4.hip 'block_bwd_attn_r4_kernel<true, false, 2>' [hipcc flags ...]"""
import re
import subprocess
import sys
import tempfile

src, want = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-S",
           "--cuda-device-only", "-o", f.name, src] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-2000:])
    text = open(f.name).read()
names = re.findall(r"^(_Z\w+):\s*; @", text, re.M)
dem = {n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() for n in names}
hit = [n for n in names if want.replace(" ", "") in dem[n].replace(" ", "")]
if not hit:
    sys.exit("no kernel matches; have:\n" + "\n".join(sorted(set(dem.values()))))
name = hit[0]
body = text[text.index(name + ":"):]
body = body[:body.index(".Lfunc_end")].splitlines()


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def census(lines):
    c = {}
    for l in lines:
        l = l.strip()
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        k = cls(l.split()[0])
        c[k] = c.get(k, 0) + 1
    return c


# the tile loop: the longest span between a label and the last backward branch to it
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
order = ["mfma", "valu", "lds", "vmem", "salu", "smem", "branch", "waitcnt", "nop", "barrier", "other"]
print(dem[name])
for tag, c in (("whole kernel", census(body)), ("largest loop", census(body[best[1]:best[2] + 1]))):
    tot = sum(c.values())
    mf = max(1, c.get("mfma", 0))
    print(f"  {tag:13s} total {tot:6d} | " + " ".join(f"{k} {c.get(k, 0)}" for k in order if c.get(k, 0)) +
          f" | per MFMA: valu {c.get('valu', 0) / mf:.2f} lds {c.get('lds', 0) / mf:.2f} scalar-class {(c.get('salu', 0) + c.get('branch', 0) + c.get('waitcnt', 0) + c.get('nop', 0) + c.get('barrier', 0) + c.get('smem', 0)) / mf:.2f}")
