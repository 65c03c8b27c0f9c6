"""kernel study: cycle stamps of one wave of block_fwd (MSST_DBG=8)."""
import os, sys, ctypes
# needs a stamps build first:  python -m maskedsst_amd.build --stamps   (rebuild with --force afterwards)
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
H = 8
t0 = s[0]
names = {0: "tile start", 1: "LN1 done", 2: "barrier"}
for h in range(H):
    names.update({3 + 8*h: f"h{h} start", 4 + 8*h: f"h{h} A done", 5 + 8*h: f"h{h} barrier", 6 + 8*h: f"h{h} softmax+P st",
                  7 + 8*h: f"h{h} O st", 8 + 8*h: f"h{h} outproj", 9 + 8*h: f"h{h} barrier2"})
names.update({3 + 8*H: "heads done", 80: "e: w1f issued", 81: "e: x1 + store", 82: "e: LN2 + xn2 st", 83: "e: w2f issue+MLP1", 84: "e: gelu + st", 85: "e: MLP2", 4 + 8*H: "epilogue done", 5 + 8*H: "final barrier"})
prev = t0
order = sorted(k for k in names if k < 68) + [80,81,82,83,84,85] + [4+8*H, 5+8*H]
for i in order:
    print(f"{names[i]:18s} +{s[i]-prev:7d}  (t={s[i]-t0})")
    prev = s[i]
