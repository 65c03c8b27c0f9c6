"""kernel study: cycle stamps of one wave of block_fwd_hw (MSST_DBG=8; stamps build, see tools/stamps.py).
Build with -DMSST_STAMPS -DMSST_F2_STAMPSEL=0x00c09 or 0x3f001 (subsets that do not spill) and -DMSST_F2_STAMP_TID=<thread>."""
import os, sys, ctypes
os.environ["MSST_DBG"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product
cfg = dict(bands=200, depth=1, B=256)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
eng = model.engine()
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
eng.lib.msst_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
eng.prep_weights()
x0 = eng.tokenize(x.cuda(), None)
for _ in range(2):
    acts, _ = eng.blocks_fwd(x0, save=True)
torch.cuda.synchronize()
s = buf.cpu().numpy()
names = ["tile start", "-", "-", "q m=0 start", "q m=1", "k m=0", "k m=1", "v mm=0", "v mm=1", "phase A done",
         "attention + O stores", "x requests + barrier (O complete)", "out-proj + exchange store", "barrier", "epilogue + LN2 (1 barrier inside)",
         "barrier", "MLP (1 barrier inside)", "end"]
prev = s[0]
for i, n in enumerate(names):
    if n == '-' or (i and s[i] == 0): continue   # (stamps not compiled in: MSST_F2_STAMPSEL)
    print(f"{n:18s} +{s[i]-prev:7d}  (t={s[i]-s[0]})")
    prev = s[i]
