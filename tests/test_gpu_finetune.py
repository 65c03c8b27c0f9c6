"""GPU: classification path (row a17, BASELINE config 5): logits / CE loss / gradients of the bare encoder vs
the oracle and vs the fixtures captured from the reference (tools/make_golden.py: run_finetune_case)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, oracle_cfg_from, seed_all, fp_np
from util import relerr

pytestmark = pytest.mark.gpu


def build_encoder(cfg, precision):
    from maskedsst_amd import ViTSpatialSpectral
    seed_all(5)
    enc = ViTSpatialSpectral(
        image_size=8, spatial_patch_size=1, spectral_patch_size=10, num_classes=cfg["n_classes"], dim=96,
        depth=cfg["depth"], heads=8, mlp_dim=64, dropout=0.0, emb_dropout=0.0, channels=cfg["bands"],
        spectral_pos_embed=cfg["spectral_pos_embed"], spectral_pos=torch.arange(cfg["bands"] // 10),
        blockwise_patch_embed=True, precision=precision)
    x = torch.randn(cfg["B"], cfg["bands"], 8, 8)
    label = torch.randint(-1, cfg["n_classes"], (cfg["B"], 8, 8))
    return enc, x, label


@pytest.mark.parametrize("name", ["finetune_200b_L4_B2.npz", "finetune_50b_L2_B2_specpos.npz"])
def test_finetune_step_fp32(name):
    from oracle import classify_forward
    g = load_golden(name)
    cfg = g["cfg"]
    enc, x, label = build_encoder(cfg, "fp32")
    np.testing.assert_array_equal(label.numpy().astype(np.int8), g["label"])
    assert sum(p.numel() for p in enc.parameters()) == int(g["n_params"])
    params = {"encoder." + k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    ocfg = oracle_cfg_from(cfg)
    ref_logits = classify_forward(params, x, ocfg)
    ref_loss = F.cross_entropy(ref_logits, label, ignore_index=-1)
    ref_loss.backward()
    enc = enc.cuda()
    logits = enc(x.cuda())
    assert logits.shape == ref_logits.shape
    loss = F.cross_entropy(logits, label.cuda(), ignore_index=-1)
    loss.backward()
    torch.cuda.synchronize()
    assert relerr(logits, ref_logits) < 1e-4
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    bad = []
    for k, p in enc.named_parameters():
        e = relerr(p.grad, params["encoder." + k].grad)
        if not e < 3e-4:
            bad.append((k, e))
    assert not bad, bad
    gsq = sum(float((p.grad.double() ** 2).sum()) for p in enc.parameters())
    assert abs(gsq ** 0.5 - float(g["grad_l2"])) <= 1e-3 * float(g["grad_l2"])


def test_finetune_bf16_and_optimizer_step():
    from maskedsst_amd.optim import FusedAdamW
    cfg = dict(bands=50, depth=2, B=4, n_classes=20, spectral_pos_embed=False)
    enc, x, label = build_encoder(cfg, "bf16")
    enc = enc.cuda()
    opt = FusedAdamW(enc, lr=5e-4, weight_decay=5e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = F.cross_entropy(enc(x.cuda()), label.cuda(), ignore_index=-1)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    enc.eval()
    with torch.no_grad():
        out = enc(x.cuda())
    assert out.shape == (4, 20, 8, 8)
