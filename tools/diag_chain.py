"""diagnostic: chained vs unchained bf16 backward, each against the fp32-mode backward of the same dy (per-tensor rel-L2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product, rel_l2

cfg = dict(bands=int(os.environ.get("BANDS", "50")), depth=2, B=4)
drop = (0.0, 0)
res = {}
for prec, chain in (("fp32", "0"), ("bf16", "1"), ("bf16", "0")):
    model, params, x = build_product(cfg, precision=prec, device="cuda")
    eng = model.engine()
    masks = model.draw_masks(cfg["B"])
    out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=drop)
    torch.manual_seed(1)
    dy = torch.randn_like(out["enc_out"]) * 1e-3
    os.environ["MSST_BWD_CHAIN"] = chain
    eng.fp.grad.zero_()
    dx0 = eng.blocks_bwd(out["acts"], out["x1s"], dy.clone(), drop=drop)
    torch.cuda.synchronize()
    res[(prec, chain)] = (dx0.clone(), {n: eng.fp.view(n, eng.fp.grad).clone() for n, _ in eng.trainable()})
ref = res[("fp32", "0")]
a, b = res[("bf16", "1")], res[("bf16", "0")]
print("dx0: chain vs fp32 %.3e, unchained vs fp32 %.3e, chain vs unchained %.3e" % (rel_l2(a[0], ref[0]), rel_l2(b[0], ref[0]), rel_l2(a[0], b[0])))
rows = []
for n in ref[1]:
    if float(ref[1][n].abs().max()) == 0: continue
    rows.append((rel_l2(a[1][n], b[1][n]), rel_l2(a[1][n], ref[1][n]), rel_l2(b[1][n], ref[1][n]), n, ref[1][n].numel()))
rows.sort(reverse=True)
for r in rows[:12]:
    print("%.2e  chain-vs-fp32 %.2e  unchained-vs-fp32 %.2e  %s (%d)" % r)
