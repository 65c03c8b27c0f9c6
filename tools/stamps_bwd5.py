"""kernel study: cycle stamps of the eight waves of one workgroup of block_bwd_ln1mlp (msst_bwd5.hip), one mid-walk tile.
needs a stamps build first:  python -m maskedsst_amd.build --stamps"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product

cfg = dict(bands=200, depth=2, B=256)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
eng = model.engine()
masks = model.draw_masks(cfg["B"])
drop = (0.1, 5)
out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=drop)
dy = torch.randn_like(out["enc_out"]) * 1e-3
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
assert eng.lib.msst_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0, "build with --stamps"
for rep in range(2):
    buf.zero_()
    eng.blocks_bwd(out["acts"], out["x1s"], dy.clone(), drop=drop)   # the last fused launch (blocks 1 | 0) leaves its stamps
    torch.cuda.synchronize()
s = buf.cpu().numpy()
t0 = min(int(s[16 * w]) for w in range(8))
mn = ["pre", "gemm1", "gelu", "gemm2", "ln2bwd", "B1", "dW", "B2"]
ln = ["compute", "issue", "B1", "writeout", "B2"]
for w in range(8):
    names = mn if w < 4 else ln
    st = [int(s[16 * w + k]) for k in range(len(names) + 1)]
    d = [st[k + 1] - st[k] for k in range(len(names))]
    print(f"wave {w} ({'M' if w < 4 else 'L'}): start {st[0] - t0:6d} total {st[-1] - st[0]:6d} | " + " ".join(f"{n}:{v}" for n, v in zip(names, d)))
