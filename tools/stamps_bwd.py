"""kernel study: cycle stamps of one wave of block_bwd_attn (MSST_DBG=8)."""
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
for _ in range(2):
    loss = model(x.cuda()); loss.backward()
    for p in model.parameters(): p.grad = None
torch.cuda.synchronize()
s = buf.cpu().numpy()
names = ["tile start", "LN1 done", "barrier", "phase A done", "barrier", "DA+S+softmax+P", "O + dO", "dP + ds", "barrier",
         "phase C done", "barrier", "dq st + LN1#2", "barrier", "dW loop", "vm0+barrier", "dx GEMM", "copy-out"]
prev = s[0]
for i, n in enumerate(names):
    print(f"{n:18s} +{s[i]-prev:7d}  (t={s[i]-s[0]})")
    prev = s[i]
