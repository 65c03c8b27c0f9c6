"""Host-side helpers mirroring the parts of reference ``src/utils.py`` that touch the model:
optimizer construction (``:36-59``), checkpoint hand-off pretrain -> finetune (``:276-313``), the
finetune training step (``:608-663``) and the Houston spectral-position lookup (``:415-429``)."""
import numpy as np
import torch


def get_optimizers(model, config, fused=True):
    """reference src/utils.py:36-59 (Adam / AdamW + ReduceLROnPlateau / cosine).  ``fused=True`` swaps
    AdamW for the one-launch FusedAdamW (same update rule)."""
    if config.optimizer == "Adam":
        optimizer = torch.optim.Adam(model.parameters(), lr=config.lr, weight_decay=config.weight_decay)
    elif config.optimizer == "AdamW":
        if fused:
            from .optim import FusedAdamW
            optimizer = FusedAdamW(model, lr=config.lr, weight_decay=config.weight_decay)
        else:
            optimizer = torch.optim.AdamW(model.parameters(), lr=config.lr, weight_decay=config.weight_decay)
    else:
        raise ValueError(f"unknown optimizer {config.optimizer}")
    if config.scheduler == "ReduceLROnPlateau":
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.9, patience=5)
    elif config.scheduler == "cosine":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=50, eta_min=0, last_epoch=-1)
    else:
        raise ValueError(f"unknown scheduler {config.scheduler}")
    return optimizer, scheduler


def get_pos_for_spectral_embedding(spectral_patch_depth, wavelengths, reference_wavelengths):
    """For every spectral block (``spectral_patch_depth`` consecutive bands, last block ragged) of a sensor with band
    centres ``wavelengths``: the index of the block of ``reference_wavelengths`` whose mean wavelength is closest
    (reference src/vit_spatial_spectral.py:767-800).  Used to address the spectral position table of a model pre-trained
    on the reference sensor with another sensor's bands (Houston2018 on an EnMAP model: [0, 3, 5, 7, 9])."""
    def block_means(w):
        w = np.asarray(w, dtype=np.float64)
        edges = np.arange(0, len(w), spectral_patch_depth)
        return np.add.reduceat(w, edges) / np.diff(np.append(edges, len(w)))
    bm, rm = block_means(wavelengths), block_means(reference_wavelengths)
    return [int(i) for i in np.abs(rm[None, :] - bm[:, None]).argmin(axis=1)]


def get_spectral_pos_embedding(dataset, n_bands, band_patch_size, wavelengths=None, reference_wavelengths=None):
    """reference src/utils.py:415-429: positions of a dataset's spectral tokens in the pre-training sensor's spectral
    sequence -- the identity for the EnMAP-derived label sets (worldcover / dfc), the nearest-block lookup for
    Houston2018 (its band-centre table and the reference sensor's are data the caller supplies; the readers that carry
    them are out of scope here).  Any other name raises, as in the reference (dataset: enmap is a pre-training set and
    has no finetune labels there either)."""
    if dataset in ("worldcover", "dfc"):
        return torch.arange(n_bands // band_patch_size)
    if dataset == "houston2018":
        if wavelengths is None or reference_wavelengths is None:
            raise ValueError("houston2018 needs the band-centre tables of both sensors")
        return get_pos_for_spectral_embedding(band_patch_size, wavelengths, reference_wavelengths)
    raise NotImplementedError(f"Unknown dataset {dataset=}")


def load_checkpoint(config, model, classifier_name="mlp_head", device="cpu", checkpoint=None):
    """Initialise a bare encoder from a SimMIM pre-training checkpoint, with the reference's semantics
    (src/utils.py:276-313): keys ``encoder.X`` are renamed to ``X``; every other key (``mask_token``,
    ``to_pixels.*``) is dropped; the checkpoint's classifier Linear is replaced by the freshly initialised
    one of ``model`` (its output shape differs); then a STRICT ``load_state_dict``.
    ``checkpoint`` may be an already loaded dict (else ``config.checkpoint_path`` is read)."""
    if checkpoint is None:
        checkpoint = torch.load(config.checkpoint_path, map_location=device, weights_only=False)
    src = checkpoint["model_state_dict"]
    weights = {k[len("encoder."):]: v for k, v in src.items() if k.startswith("encoder.")}
    linear_idx = 2 if getattr(model, "pixelwise", False) else 1
    head = getattr(model, classifier_name)[linear_idx]
    patch_sub = getattr(config, "patch_sub", 0)
    if patch_sub != 0 and weights.get("pos_embed") is not None:
        assert model.pos_embed.shape[1] == (config.image_size - patch_sub) ** 2
        weights["pos_embed"] = weights["pos_embed"][:, : model.pos_embed.shape[1], :]
    weights.pop(f"{classifier_name}.1.bias", None)
    weights.pop(f"{classifier_name}.1.weight", None)
    weights[f"{classifier_name}.{linear_idx}.bias"] = head.bias.detach().clone()
    weights[f"{classifier_name}.{linear_idx}.weight"] = head.weight.detach().clone()
    print(model.load_state_dict(weights))
    return model


def train_step(img, label, model, config, device, criterion, optimizer, acc_criterion=None):
    """reference src/utils.py:608-663 for the ViTSpatialSpectral method: optional random crop, forward,
    CE(ignore_index) loss, pixel accuracy on valid labels, backward, optimizer step."""
    patch_sub = getattr(config, "patch_sub", 0)
    if config.image_size != 64 and img.shape[-1] == 64:
        x, y = torch.randint(0, 64 - config.image_size - patch_sub, size=(2,))
        s = config.image_size - patch_sub
        img = img[:, :, x:x + s, y:y + s]
        label = label[:, x:x + s, y:y + s]
    img = img.to(device)
    label = label.to(device)
    optimizer.zero_grad()
    output = model(img)
    loss = criterion(output, label)
    if torch.isnan(loss):
        raise ValueError("Loss is NaN")
    pred = output.argmax(dim=1)
    valid = label != config.ignored_label
    acc = (pred[valid] == label[valid]).sum() / max(int(valid.sum()), 1)
    macro_acc = acc_criterion(pred[valid].to(int), label[valid]) if (acc_criterion is not None and valid.any()) else acc
    loss.backward()
    optimizer.step()
    return loss, acc, macro_acc
