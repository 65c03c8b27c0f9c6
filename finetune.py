#!/usr/bin/env python3
"""Finetune entry point (mirror of the reference's ``finetune.py`` for the ViTSpatialSpectral method) on the
MI355X-native kernels: build the encoder, optionally initialise it from a SimMIM checkpoint
(``load_checkpoint``), train the pixel-wise classification head with CE(ignore_index=-1).

The labelled GeoTIFF readers are out of scope: ``--synthetic`` (default) draws standardised random tiles and
random labels in {-1 .. n_classes-1}, which exercises the identical compute path (BASELINE config 5 checks the
logits / loss / gradients of that path against the CPU reference in tests/test_gpu_finetune.py)."""
import argparse
import random
import sys
import time

import numpy as np
import torch
import yaml

from maskedsst_amd import ViTSpatialSpectral
from maskedsst_amd.config import Dotdict
from maskedsst_amd.utils import get_spectral_pos_embedding, load_checkpoint, train_step

SEED = 5


def get_finetune_config(path, general_path, seed, device):
    """reference src/utils.py:337-364 (ViTSpatialSpectral branch; worldcover/dfc spectral positions)"""
    hp = yaml.safe_load(open(path))
    general = yaml.safe_load(open(general_path))
    hp.update(general["data"][hp["dataset"]])
    hp.update(general["transformer"])
    hp["seed"], hp["device"] = seed, device
    if hp["method_name"] != "ViTSpatialSpectral":
        raise NotImplementedError("only the ViTSpatialSpectral method is built (the DeepHyperX 'li' baseline is out of scope)")
    if hp["dataset"] == "houston2018":
        # the two sensors' band-centre tables live with the (out of scope) readers: the config carries the lookup's result
        hp["spectral_pos"] = torch.as_tensor(hp["spectral_pos"])
        assert len(hp["spectral_pos"]) == hp["n_bands"] // hp["band_patch_size"]
    else:
        hp["spectral_pos"] = get_spectral_pos_embedding(hp["dataset"], hp["n_bands"], hp["band_patch_size"])
    hp["patch_sub"] = 1 if (hp["pixelwise"] and hp["image_size"] % 2 == 0) else 0
    return Dotdict(hp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset", nargs="?", default="enmap", choices=["enmap", "houston2018"])   # reference finetune.py:42-46
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args()
    random.seed(SEED); np.random.seed(SEED); torch.manual_seed(SEED)
    if not torch.cuda.is_available():
        raise SystemExit("finetune.py needs an MI355X: maskedsst_amd has no CPU fallback")
    device = torch.device("cuda")
    config = get_finetune_config(f"configs/finetune_config_{args.dataset}.yaml", "configs/config.yaml", SEED, device)
    if args.batch_size:
        config.batch_size = args.batch_size
    if args.checkpoint:
        config.checkpoint_path = args.checkpoint
    model = ViTSpatialSpectral(
        image_size=config.image_size - config.patch_sub, spatial_patch_size=config.patch_size,
        spectral_patch_size=config.band_patch_size, num_classes=config.n_classes, dim=config.transformer_dim,
        depth=config.transformer_depth, heads=config.transformer_n_heads, mlp_dim=config.transformer_mlp_dim,
        dropout=config.transformer_dropout, emb_dropout=config.transformer_emb_dropout, channels=config.n_bands,
        spectral_pos=config.spectral_pos, spectral_pos_embed=config.spectral_pos_embed,
        blockwise_patch_embed=config.blockwise_patch_embed, spectral_only=config.spectral_only,
        pixelwise=config.pixelwise, pos_embed_len=config.pos_embed_len, precision=args.precision)
    if config.checkpoint_path is not None:
        model = load_checkpoint(config, model, "mlp_head", "cpu")
    model.to(device)
    if config.linear_eval:
        for n, p in model.named_parameters():
            p.requires_grad_("mlp_head" in n)
    head = [p for n, p in model.named_parameters() if "mlp_head" in n]
    body = [p for n, p in model.named_parameters() if "mlp_head" not in n]
    optimizer = torch.optim.Adam([{"params": body}, {"params": head, "lr": config.mlp_head_lr}], lr=config.lr,
                                 weight_decay=config.weight_decay)   # finetune.py:110-134 (two learning rates)
    criterion = torch.nn.CrossEntropyLoss(ignore_index=config.ignored_label)
    gen = torch.Generator().manual_seed(SEED)
    model.train()
    t0 = time.time()
    for step in range(1, args.steps + 1):
        img = torch.randn(config.batch_size, config.n_bands, 64, 64, generator=gen)
        if config.dataset == "houston2018":
            img[:, 48:] = 0.0   # 48 real bands zero padded to 50 (reference src/data_houston2018.py:268-269)
        label = torch.randint(-1, config.n_classes, (config.batch_size, 64, 64), generator=gen)
        loss, acc, _ = train_step(img, label, model, config, device, criterion, optimizer)
        if step % config.logging_freq == 0:
            print(f"step {step} loss {loss.item():.4f} acc {float(acc):.3f} {step * config.batch_size / (time.time() - t0):.1f} samples/s",
                  flush=True)


if __name__ == "__main__":
    main()
