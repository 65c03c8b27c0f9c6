"""kernel study: the 24 encoder blocks forward with IEEE-half GEMM operands (MSST_FWD_HALF=1, the default) against bf16 operands (=0),
alternating in ONE process so that the box's clock cancels: us per block of both.
usage: python tools/fwd_half_ab.py [--batch 256] [--bands 200] [--reps 12]"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--bands", type=int, default=200)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--dropout", type=float, default=0.1)
a = ap.parse_args()
cfg = dict(bands=a.bands, depth=12, B=a.batch, dropout=a.dropout)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
model.train()
eng = model.engine()
eng.prep_weights()
x0 = eng.tokenize(x.cuda(), None)
drop = (a.dropout, 77) if a.dropout else (0.0, 0)
t = {"1": [], "0": []}
for rep in range(a.reps + 2):
    for flag in ("1", "0"):
        os.environ["MSST_FWD_HALF"] = flag
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        acts, x1s = eng.blocks_fwd(x0, save=True, drop=drop)
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            t[flag].append(e0.elapsed_time(e1) * 1e3 / 24)
        del acts, x1s
med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
print(f"B={a.batch} bands={a.bands}: half operands {med['1']:.1f} us/block (min {min(t['1']):.1f}), bf16 operands {med['0']:.1f} (min {min(t['0']):.1f}), ratio {med['1'] / med['0']:.4f}")
