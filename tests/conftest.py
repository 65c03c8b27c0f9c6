import os
import sys
import json

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    d = {k: z[k] for k in z.files}
    if "cfg" in d:
        d["cfg"] = json.loads(bytes(d["cfg"]).decode())
    if "names" in d:
        d["names"] = bytes(d["names"]).decode().split("\n")
    return d


def fp_np(t):
    """same fingerprint as tools/make_golden.py:fp"""
    import torch
    t = t.detach().to(torch.float64).reshape(-1)
    head = np.zeros(8, dtype=np.float64)
    n = min(8, t.numel())
    head[:n] = t[:n].numpy()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), float(t.numel())], head])


def oracle_cfg_from(cfg):
    from oracle import OracleConfig
    return OracleConfig(
        bands=cfg["bands"], depth=cfg["depth"], heads=cfg.get("heads", 8), image_size=cfg.get("image_size", 8),
        n_classes=cfg.get("n_classes", 8),
        spectral_pos_embed=cfg.get("spectral_pos_embed", False),
        masking_ratio=cfg.get("masking_ratio", 0.7), mask_patch_size=cfg.get("mask_patch_size", 4),
        tube_masking=cfg.get("tube_masking", True),
        to_pixels_per_spectral_block=cfg.get("to_pixels_per_spectral_block", True),
    )


def apply_qkv_scale(named_tensors, cfg):
    """The "peaky attention" fixtures (tools/make_golden.py::build, cfg key qkv_scale): every to_qkv.weight is multiplied by the
    factor AFTER the seeded construction, in place."""
    import torch
    s = cfg.get("qkv_scale")
    if not s:
        return
    with torch.no_grad():
        for n, p in named_tensors:
            if n.endswith("to_qkv.weight"):
                p.mul_(float(s))


def seed_all(seed=5):
    import random
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


@pytest.fixture
def golden():
    return load_golden
