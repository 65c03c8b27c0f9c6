"""Shared helpers for the parity tests: build the product model and the oracle parameters from
the same seed (parameters are drawn on the CPU in the reference order, then moved to the GPU)."""
import json
import os

import numpy as np
import torch

from conftest import ROOT, seed_all, oracle_cfg_from


def build_product(cfg, precision="fp32", device="cpu"):
    from maskedsst_amd import ViTSpatialSpectral, SimMIMSpatialSpectral
    seed_all(5)
    enc = ViTSpatialSpectral(
        image_size=cfg.get("image_size", 8), spatial_patch_size=1, spectral_patch_size=10, num_classes=cfg.get("n_classes", 8),
        dim=96, depth=cfg["depth"], heads=cfg.get("heads", 8), mlp_dim=64, dropout=0.0, emb_dropout=0.0,
        channels=cfg["bands"], spectral_pos_embed=cfg.get("spectral_pos_embed", False),
        spectral_pos=torch.arange(cfg["bands"] // 10), blockwise_patch_embed=True, spectral_only=False,
        precision=precision)
    model = SimMIMSpatialSpectral(
        encoder=enc, intermediate_losses=False, masking_ratio=cfg.get("masking_ratio", 0.7),
        mask_patch_size=cfg.get("mask_patch_size", 4),
        to_pixels_per_spectral_block=cfg.get("to_pixels_per_spectral_block", True),
        tube_masking=cfg.get("tube_masking", True))
    x = torch.randn(cfg["B"], cfg["bands"], cfg.get("image_size", 8), cfg.get("image_size", 8))
    if cfg.get("zero_pad_bands"):
        x[:, cfg["bands"] - cfg["zero_pad_bands"]:] = 0.0
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if device != "cpu":
        model = model.to(device)
    return model, params, x


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def record(test, **kv):
    """With MSST_RECORD=1 (tools/final_prof.sh sets it), append a measured error to gpurun_out/parity_$MSST_ROUND.jsonl
    (scratch; the round's copy lives in profiles/).  Ordinary test runs write nothing."""
    if os.environ.get("MSST_RECORD") != "1":
        return
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_%s.jsonl" % os.environ.get("MSST_ROUND", "dev")), "a") as f:
            f.write(json.dumps(dict(test=test, **kv)) + "\n")
    except OSError:
        pass
