#!/usr/bin/env python3
"""SimMIM pre-training entry point (mirror of the reference's ``pretrain.py``), on the MI355X-native
kernels.  Same config files / keys, same seeding, same loop shape (``model(img)`` ->
``loss.backward()`` -> ``optimizer.step()``), same checkpoint dictionary.

Differences from the reference script, all opt-in or forced by the environment:
  * data: the GeoTIFF readers are out of scope -> ``--synthetic`` (default) samples a pool of standardised
    random tiles shaped like ``EnMAPWorldCoverDataset`` output ([bands, 64, 64]) and takes the same random
    8x8 crop per batch (reference pretrain.py:99-107) through ``maskedsst_amd.data.SyntheticCubeLoader``
    (worker thread, pinned staging, asynchronous host->device copies);
  * wandb is optional (absent here); losses are printed every ``logging_freq`` steps;
  * ``--dp``: one process per GPU under ``python -m torch.distributed.run`` (RCCL gradient all-reduce);
  * optimizer: fused AdamW over the flat parameter buffer (``--torch-optim`` keeps torch.optim.AdamW
    with the reference's clamp hooks);
  * dropout: the four transformer dropout sites are fused into the kernels (stateless counter-based masks).
"""
import argparse
import os
import random
import time

import numpy as np
import torch

from maskedsst_amd import ViTSpatialSpectral, SimMIMSpatialSpectral
from maskedsst_amd.config import get_pretrain_config
from maskedsst_amd.data import SyntheticCubeLoader
from maskedsst_amd.optim import FusedAdamW, attach_data_parallel, dp_mean

SEED = 5


def save_checkpoint(save_dir, epoch, model, optimizer, config, losses, img):
    """the reference's checkpoint dictionary (pretrain.py:135-148): finetune.py / load_checkpoint read
    ``model_state_dict`` (keys ``mask_token``, ``encoder.*``, ``to_pixels.*``)"""
    os.makedirs(save_dir, exist_ok=True)
    cfg = {k: v for k, v in config.__dict__.items() if not isinstance(v, torch.device)}
    stats = {"losses": torch.stack(losses).cpu(), "config": cfg,
             "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
             "lr_current": optimizer.param_groups[0]["lr"],
             "input": img.detach().cpu(), "transformer_input": img.detach().cpu()}
    path = os.path.join(save_dir, f"model_{config.encoder_name}_ep{epoch}.pth")
    torch.save(stats, path)
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="configs/pretrain_config.yaml")
    ap.add_argument("--general-config", default="configs/config.yaml")
    ap.add_argument("--synthetic", action="store_true", default=True)
    ap.add_argument("--tiles", type=int, default=256, help="synthetic 64x64 tiles per epoch")
    ap.add_argument("--pool-tiles", type=int, default=64, help="distinct synthetic tiles held in host memory")
    ap.add_argument("--val-tiles", type=int, default=4, help="synthetic 64x64 validation tiles per rank (sliding-window validation)")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=None, help="override transformer_dropout of the config")
    ap.add_argument("--torch-optim", action="store_true")
    ap.add_argument("--save-dir", default=None)
    args = ap.parse_args()

    random.seed(SEED); np.random.seed(SEED); torch.manual_seed(SEED)
    if not torch.cuda.is_available():
        raise SystemExit("pretrain.py needs an MI355X: maskedsst_amd has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=device)

    config = get_pretrain_config(args.config, args.general_config, SEED, device)
    if args.depth is not None:
        config.transformer_depth = args.depth
    if args.batch_size is not None:
        config.batch_size = args.batch_size
    if args.epochs is not None:
        config.epoch = args.epochs
    if args.dropout is not None:
        config.transformer_dropout = config.transformer_emb_dropout = args.dropout
    assert config.encoder_name == "ViTSpatialSpectral", f"encoder {config.encoder_name} not available"

    spectral_pos = torch.arange(config.n_bands // config.band_patch_size)
    model = ViTSpatialSpectral(
        image_size=config.image_size, spatial_patch_size=config.patch_size,
        spectral_patch_size=config.band_patch_size, num_classes=config.n_classes,
        dim=config.transformer_dim, depth=config.transformer_depth, heads=config.transformer_n_heads,
        mlp_dim=config.transformer_mlp_dim, dropout=config.transformer_dropout,
        emb_dropout=config.transformer_emb_dropout, channels=config.n_bands,
        spectral_pos_embed=config.spectral_pos_embed, spectral_pos=spectral_pos,
        blockwise_patch_embed=config.blockwise_patch_embed, spectral_only=config.spectral_only,
        precision=args.precision)
    model = SimMIMSpatialSpectral(
        encoder=model, intermediate_losses=config.mim_intermediate_losses,
        masking_ratio=config.mim_masking_ratio, mask_patch_size=config.mim_mask_patch_size,
        to_pixels_per_spectral_block=config.to_pixels_per_spectral_block,
        tube_masking=config.tube_masking).to(device)
    config.model_params = sum(p.numel() for p in model.parameters())

    if args.torch_optim and world > 1:
        # torch.optim.AdamW reads p.grad; with the clamp hooks those are fresh tensors, not views of the flat buffer the
        # bucket reducer all-reduces, and the 1/world mean is applied inside FusedAdamW -- the combination would step every
        # rank on its local, unaveraged gradients.
        raise SystemExit("--torch-optim is single-process only: data parallel runs use the fused AdamW "
                         "(mean + clamp after the all-reduce, inside the optimizer launch)")
    if args.torch_optim:
        optimizer = torch.optim.AdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay)
        if config.clip_grad_norm:
            for p in model.parameters():
                p.register_hook(lambda grad: torch.clamp(grad, -1, 1))
    else:
        optimizer = FusedAdamW(model, lr=config.lr, weight_decay=config.weight_decay,
                               grad_clamp=1.0 if config.clip_grad_norm else 0.0)
    if config.scheduler == "cosine":       # reference src/utils.py:47-59
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=50, eta_min=0, last_epoch=-1)
    elif config.scheduler == "ReduceLROnPlateau":
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.9, patience=5)
    else:   # the stepping below tests the same two names: anything else would train at a constant rate without a word
        raise ValueError(f"unknown scheduler {config.scheduler!r} (reference src/utils.py:47-59 knows ReduceLROnPlateau and cosine)")
    reducer = attach_data_parallel(model) if world > 1 else None

    # synthetic standardised tiles [bands, 64, 64] (per-band N(0,1), like StandardizeEnMAP output), cropped per batch
    gen = torch.Generator().manual_seed(SEED + 1000 * rank)
    val_tiles = None
    if not config.skip_val:   # this rank's held-out tiles (resident: args.val_tiles x bands x 64 x 64 floats)
        val_tiles = torch.randn(args.val_tiles, config.n_bands, 64, 64, generator=gen).to(device)
    per_rank = config.batch_size // world
    steps_per_epoch = max(1, args.tiles // config.batch_size)

    step, losses, t0 = 0, [], time.time()
    for epoch in range(config.epoch):
        model.train()
        loader = SyntheticCubeLoader(per_rank, config.n_bands, image_size=config.image_size, pool_tiles=args.pool_tiles,
                                     steps=steps_per_epoch, seed=SEED + 1000 * rank + epoch, device=device)
        for img in loader:
            optimizer.zero_grad()
            loss = model(img)
            loss.backward()
            if reducer is not None:
                s = reducer.finish()
                if isinstance(optimizer, FusedAdamW):
                    optimizer.grad_scale = s
            optimizer.step()
            step += 1
            losses.append(loss.detach())
            if step % config.logging_freq == 0:
                recent = torch.stack(losses[-config.logging_freq:]).mean().item()
                if not np.isfinite(recent):
                    raise ValueError("Loss is NaN")
                if rank == 0:
                    print(f"epoch {epoch} step {step} loss {recent:.6e} lr {optimizer.param_groups[0]['lr']:.3e} "
                          f"{step * config.batch_size / (time.time() - t0):.1f} samples/s", flush=True)
            if args.max_steps and step >= args.max_steps:
                break
        loader.close()
        if epoch % config.model_save_freq == 0:     # reference pretrain.py:135-151
            if args.save_dir and rank == 0:
                save_checkpoint(args.save_dir, epoch, model, optimizer, config, losses, img)
            if epoch == 10 and config.model_save_freq == 1:
                config.model_save_freq = 10
        if not config.skip_val:
            # validation as in the reference (pretrain.py:155-197): eval mode, no_grad, every image_size x image_size window
            # of the held-out 64 x 64 tiles (stride = window size), mean loss -> ReduceLROnPlateau
            model.eval()
            val_losses = []
            with torch.no_grad():
                s_ = config.image_size
                for x0 in range(0, 64, s_):
                    for y0 in range(0, 64, s_):
                        val_losses.append(model(val_tiles[:, :, x0:x0 + s_, y0:y0 + s_].contiguous()))
            val_loss = torch.stack(val_losses).mean()
            # every rank validates on its own shard: the plateau scheduler must see the SAME number everywhere, or the
            # ranks cut the learning rate at different epochs and the replicas drift apart
            if config.scheduler == "ReduceLROnPlateau":
                scheduler.step(dp_mean(val_loss).item())
        if config.scheduler == "cosine":
            scheduler.step()
        if args.max_steps and step >= args.max_steps:
            break
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
