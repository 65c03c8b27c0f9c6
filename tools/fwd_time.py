"""kernel study: the 24 encoder blocks forward only (one launch per block, everything the backward needs saved), us per block --
run per library build through tools/with_lib.py, the builds taken in turn inside ONE gpurun call (same box):
  python tools/with_lib.py maskedsst_amd/libmsst_<tag>.so tools/fwd_time.py [--batch 256] [--bands 200] [--reps 10]"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--bands", type=int, default=200)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--dropout", type=float, default=0.1)
a = ap.parse_args()
cfg = dict(bands=a.bands, depth=12, B=a.batch, dropout=a.dropout)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
model.train()
eng = model.engine()
eng.prep_weights()
x0 = eng.tokenize(x.cuda(), None)
drop = (a.dropout, 77) if a.dropout else (0.0, 0)
ts = []
for rep in range(a.reps + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    acts, x1s = eng.blocks_fwd(x0, save=True, drop=drop)
    e1.record()
    torch.cuda.synchronize()
    if rep >= 2:
        ts.append(e0.elapsed_time(e1) * 1e3 / 24)
    del acts, x1s
ts.sort()
from maskedsst_amd import _lib
print(f"{os.path.basename(_lib.LIB_PATH):28s} B={a.batch} bands={a.bands} X1_BF16={os.environ.get('MSST_X1_BF16', '1')} LSE={os.environ.get('MSST_LSE', '1')}: "
      f"{ts[len(ts) // 2]:.1f} us/block (min {ts[0]:.1f}, max {ts[-1]:.1f})")
