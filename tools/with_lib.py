"""kernel study: run a script of this repo (bench.py, tools/*.py) against ANOTHER build of the library -- a variant built next to the
product one in the build container (maskedsst_amd.build.build(lib=..., tag=...)), so that a GPU call compares builds without rebuilding.
usage: python tools/with_lib.py maskedsst_amd/libmsst_alt.so bench.py --steps 8 ..."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
sys.argv = sys.argv[2:]
from maskedsst_amd import _lib
_lib.LIB_PATH = lib
if script == "-m":   # python tools/with_lib.py <lib> -m pytest tests/...
    sys.argv = sys.argv[1:]
    runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
else:
    runpy.run_path(script, run_name="__main__")
