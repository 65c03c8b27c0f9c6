"""``SimMIMSpatialSpectral`` -- drop-in mirror of the reference masked-image-modelling wrapper.

Mirrors reference ``src/vit_simmim_original.py:139-340`` (constructor, attributes, ``state_dict``
keys ``mask_token`` / ``encoder.*`` / ``to_pixels.*``, ``forward(img) -> scalar loss``).  The
forward/backward arithmetic runs in the HIP kernels of ``libmsst.so``; mask generation stays on the
host and is bit-exact with the reference (``maskedsst_amd/masking.py``).
"""
import weakref

import numpy as np
import torch
from torch import nn

from .masking import MaskGenerator, topk_masks, inverse_csr
from .vit_spatial_spectral import ViTSpatialSpectral


class BlockwiseToPixels(nn.Module):
    """Parameter container for the per-spectral-block pixel decoders (reference :9-40);
    block id of a token = token index // num_spatial_patches."""

    def __init__(self, dim, num_spectral_blocks, pixels_per_patch, precision="32-true"):
        super().__init__()
        self.pixels_per_patch = pixels_per_patch
        self.layers = nn.ModuleList([nn.Linear(dim, pixels_per_patch) for _ in range(num_spectral_blocks)])
        if precision == "16-mixed":
            self.dtype = torch.float16
        elif precision == "32-true":
            self.dtype = torch.float32

    def forward(self, x, block_indices):
        raise RuntimeError("BlockwiseToPixels is fused into the masked-L1 head kernel (no eager fallback)")


class SimMIMSpatialSpectral(nn.Module):
    def __init__(self, *, encoder, masking_ratio=0.5, mask_patch_size=1, tube_masking=False,
                 intermediate_losses=False, to_pixels_per_spectral_block=False, precision="32-true"):
        super().__init__()
        assert masking_ratio > 0 and masking_ratio < 1, "masking ratio must be kept between 0 and 1"
        if not isinstance(encoder, ViTSpatialSpectral):
            raise NotImplementedError("only the ViTSpatialSpectral encoder is accelerated")
        if intermediate_losses:
            # the reference only supports this with the legacy _V1 encoder (it would NameError here)
            raise NotImplementedError("intermediate_losses requires the legacy ViTSpatialSpectral_V1 encoder")
        self.masking_ratio = masking_ratio
        self.mask_patch_size = mask_patch_size
        self.intermediate_losses = intermediate_losses
        self.to_pixels_per_spectral_block = to_pixels_per_spectral_block
        self.tube_masking = tube_masking
        if self.mask_patch_size != 1:
            self.mask_generator = MaskGenerator(
                input_size=encoder.image_size, mask_patch_size=mask_patch_size,
                model_patch_size=encoder.patch_height, mask_ratio=self.masking_ratio)
        self.encoder = encoder
        encoder_dim = encoder.dim
        self.to_patch = encoder.to_patch_embedding.to_patch
        self.patch_to_emb = encoder.to_patch_embedding.embed
        self.pixel_values_per_patch = encoder.pixels_per_patch
        self.mask_token = nn.Parameter(torch.randn(encoder_dim))
        if self.to_pixels_per_spectral_block:
            self.to_pixels = BlockwiseToPixels(encoder_dim, encoder.num_spectral_patches,
                                               self.pixel_values_per_patch, precision=precision)
        else:
            self.to_pixels = nn.Linear(encoder_dim, self.pixel_values_per_patch)
        self._engine = None
        encoder._engine_owner = weakref.ref(self)
        # data-parallel placement of this process: masks are drawn for the GLOBAL batch
        self.dp_rank, self.dp_world = 0, 1
        self.last_masks = None

    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self.encoder, self)
        return self._engine

    # ------------------------------------------------------------------
    def draw_masks(self, batch):
        """Host-side masks for ``batch`` local samples (reference :252-282).  Under data parallel
        the global batch's masks are drawn on every rank and the local rows are sliced."""
        enc = self.encoder
        T = enc.num_patches
        num_masked = int(self.masking_ratio * T)
        gb = batch * self.dp_world
        if self.mask_patch_size == 1:
            bm, idx = topk_masks(gb, T, num_masked)
        elif self.tube_masking:   # closed form: only the local rows are materialised (RNG still advances globally)
            lo = self.dp_rank * batch
            return self.mask_generator.get_batch_tube_masked(gb, enc.num_spectral_patches, num_masked,
                                                             rows=(lo, lo + batch))
        else:
            bm, idx = self.mask_generator.get_batch(gb, enc.num_spectral_patches, num_masked)
        lo = self.dp_rank * batch
        return bm[lo:lo + batch], idx[lo:lo + batch]

    def forward(self, img, masks=None):
        """img [B, bands, H, W] -> scalar reconstruction loss (mean |pred - target| / num_masked)."""
        eng = self.engine()
        if masks is None:
            masks = self.draw_masks(img.shape[0])
        self.last_masks = masks
        return eng.simmim_loss(img, masks[0], masks[1])
