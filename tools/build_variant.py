"""kernel study (build container): a variant library next to the product one, recompiling only the named sources with extra flags and
linking them with the product's objects -- seconds instead of minutes per variant; run on the box with tools/ab_libs.sh / tools/with_lib.py.
usage: python tools/build_variant.py <tag> <source.hip>[,<source.hip>...] [flags...]   ->  maskedsst_amd/libmsst_<tag>.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskedsst_amd import build as B

tag, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build()   # product objects up to date
objdir = os.path.join(B.HERE, "build")
vdir = os.path.join(objdir, "var_" + tag)
os.makedirs(vdir, exist_ok=True)
objs = []
for s in B.SOURCES:
    if s in srcs:
        o = os.path.join(vdir, s + ".o")
        r = subprocess.run([B._hipcc()] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, s), "-o", o], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr)
    else:
        o = os.path.join(objdir, s + ".o")
    objs.append(o)
lib = os.path.join(B.HERE, f"libmsst_{tag}.so")
r = subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print(lib)
