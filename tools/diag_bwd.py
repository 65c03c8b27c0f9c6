"""dev diagnostic: per-stage backward errors of the HIP path vs the oracle (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import oracle_cfg_from
from util import build_product, relerr
from oracle import simmim_forward
from maskedsst_amd.masking import inverse_csr

cfgs = [dict(bands=20, depth=1, B=2, heads=2), dict(bands=50, depth=2, B=4), dict(bands=200, depth=2, B=5)]
for prec in ("fp32", "bf16"):
    for cfg in cfgs:
        model, params, x = build_product(cfg, precision=prec, device="cuda")
        ocfg = oracle_cfg_from(cfg)
        masks = model.draw_masks(cfg["B"])
        for p in params.values():
            p.requires_grad_(True)
        ref = simmim_forward(params, x, ocfg, masks=masks)
        for k in ("enc_out", "tok_masked", "after_spatial"):
            ref[k].retain_grad()
        ref["loss"].backward()
        eng = model.engine()
        xc = x.cuda()
        out = eng.simmim_forward_stages(xc, masks[0], masks[1])
        T = ocfg.T
        ptr, pos = inverse_csr(masks[1].numpy(), T)
        dy = eng.head_bwd(out["enc_out"], out["dpred"], torch.from_numpy(ptr).cuda(), torch.from_numpy(pos).cuda())
        torch.cuda.synchronize()
        print(f"== {prec} {cfg}: loss {out['loss'].item():.6e} ref {ref['loss'].item():.6e}")
        print("  dy(enc_out)      ", relerr(dy, ref["enc_out"].grad))
        L = ocfg.depth
        # block-by-block: feed the oracle's gradient? (only end-to-end available) -> run full chain
        dx0 = eng.blocks_bwd(out["acts"], out["x1s"], dy.clone())
        torch.cuda.synchronize()
        print("  dx0(tok_masked)  ", relerr(dx0, ref["tok_masked"].grad))
        mask_u8 = masks[0].to(torch.uint8).cuda()
        eng.tokenize_bwd(xc, mask_u8, dx0)
        torch.cuda.synchronize()
        worst = []
        for (n, p) in eng.trainable():
            pass
        names = dict(model.named_parameters())
        for name, p in names.items():
            gr = params[name].grad
            if gr is None:
                continue
            # locate the flat gradient view of this parameter
            for fn, fp_ in eng.trainable():
                if fp_ is p:
                    g = eng.fp.view(fn, eng.fp.grad)
                    worst.append((relerr(g, gr), name))
                    break
        worst.sort(reverse=True)
        for e, n in worst[:12]:
            print(f"  {e:10.3e}  {n}")
