"""GPU parity, forward: every stage of the HIP path vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, oracle_cfg_from
from util import build_product, relerr

pytestmark = pytest.mark.gpu

CASES = [
    dict(bands=20, depth=1, B=2, heads=2),
    dict(bands=30, depth=1, B=3, heads=2, tube_masking=False),
    dict(bands=50, depth=2, B=4),
    dict(bands=50, depth=2, B=4, spectral_pos_embed=True),
    dict(bands=50, depth=2, B=4, to_pixels_per_spectral_block=False, mask_patch_size=1),
    dict(bands=200, depth=2, B=5),
]


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_forward_stages(cfg, prec, tol):
    from oracle import simmim_forward
    model, params, x = build_product(cfg, precision=prec, device="cuda")
    ocfg = oracle_cfg_from(cfg)
    masks = model.draw_masks(cfg["B"])
    with torch.no_grad():
        ref = simmim_forward(params, x, ocfg, masks=masks)
    out = model.engine().simmim_forward_stages(x.cuda(), masks[0], masks[1])
    torch.cuda.synchronize()
    for k in ["tok_embed", "tok_masked", "after_spatial", "enc_out", "pred"]:
        e = relerr(out[k], ref[k])
        assert e < tol, (k, e)
    l, lr = out["loss"].item(), ref["loss"].item()
    assert abs(l - lr) <= tol * abs(lr) + (1e-7 if prec == "fp32" else 1e-4), (l, lr)
    sgn = torch.sign(ref["pred"] - ref["target"])
    if prec == "fp32":
        mism = (out["dpred"].cpu() != sgn).float().mean().item()
        assert mism < 1e-3
