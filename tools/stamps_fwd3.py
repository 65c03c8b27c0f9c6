"""kernel study: cycle stamps of the eight waves of one workgroup of block_fwd_rs (msst_fwd3.hip), one mid-walk tile.
needs a stamps build first:  python -m maskedsst_amd.build --stamps"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product

os.environ["MSST_DBG"] = "8"
cfg = dict(bands=200, depth=1, B=256)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
eng = model.engine()
eng.prep_weights()
x0 = eng.tokenize(x.cuda(), None)
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
assert eng.lib.msst_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0, "build with --stamps"
an = ["proj0", "j01", "B", "j23", "B", "proj1", "j01", "B", "j23", "B"]
rn = ["outproj1", "epi1", "B", "q1", "B", "outproj0", "mlp1", "B", "mlp2", "ln1", "B"]
for depth_i, label in ((0, "spatial"), (1, "spectral")):
    # run the two blocks one at a time: the last launch leaves its stamps
    from maskedsst_amd._lib import MODE_SPATIAL, MODE_SPECTRAL
    buf.zero_()
    acts, x1s = eng.blocks_fwd(x0, save=True, drop=(0.1, 5))
    torch.cuda.synchronize()
    s = buf.cpu().numpy()
    if depth_i == 0:
        continue   # (blocks_fwd runs both blocks: the stamps are the spectral block's; the spatial numbers come from a depth-1 spatial-only run below)
    t0 = min(int(s[16 * w]) for w in range(8))
    for w in range(8):
        names = an if w < 4 else rn
        st = [int(s[16 * w + k]) for k in range(len(names) + 1)]
        d = [st[k + 1] - st[k] for k in range(len(names))]
        print(f"{label} wave {w} ({'A' if w < 4 else 'R'}): start {st[0] - t0:6d} total {st[-1] - st[0]:6d} | " + " ".join(f"{n}:{v}" for n, v in zip(names, d)))
