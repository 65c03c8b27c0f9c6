"""Build libmsst.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

``python -m maskedsst_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles for
gfx950 without a GPU; the resulting ``maskedsst_amd/libmsst.so`` travels with the source tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsst.so")
SOURCES = ["msst_fwd.hip", "msst_fwd3.hip", "msst_bwd.hip", "msst_bwd3.hip", "msst_bwd4.hip", "msst_bwd5.hip", "msst_ln.hip", "msst_opt.hip", "msst_api.hip"]
HEADERS = ["msst_dev.h", "msst_kernels.h", os.path.join("..", "..", "include", "msst.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, extra_flags=(), lib=None, tag=None):
    """extra_flags: e.g. ("-DMSST_STAMPS",) for the kernel-study builds used by tools/stamps*.py.
    lib / tag: another output library and object directory (build/<tag>/) -- kernel-study variants built NEXT to the product library
    (tools/gate_qkv.py loads maskedsst_amd/libmsst_lab.so), so that a GPU call does not spend minutes rebuilding"""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objdir = os.path.join(HERE, "build", tag) if tag else os.path.join(HERE, "build")
    LIB = lib or globals()["LIB"]
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv or "--stamps" in sys.argv, verbose=True,
                extra_flags=("-DMSST_STAMPS",) if "--stamps" in sys.argv else ()))
