"""Mask generation oracle (test infrastructure only) -- loop-level numpy restatement.

Follows reference ``src/vit_simmim_original.py:343-416`` (``MaskGenerator``) and the
``mask_patch_size == 1`` top-k branch at ``:254-264``.  Draws from the numpy *global* RNG exactly
like the reference (``np.random.permutation``), so after ``np.random.seed(s)`` the masks are
bit-identical to the reference's.
"""
import numpy as np
import torch


class MaskGeneratorOracle:
    """reference src/vit_simmim_original.py:343-370."""

    def __init__(self, input_size=16, mask_patch_size=4, model_patch_size=1, mask_ratio=0.6):
        assert input_size % mask_patch_size == 0
        assert mask_patch_size % model_patch_size == 0
        self.rand_size = input_size // mask_patch_size
        self.scale = mask_patch_size // model_patch_size
        self.token_count = self.rand_size ** 2
        self.mask_count = int(np.ceil(self.token_count * mask_ratio))

    def __call__(self):
        # :356-364 -- one permutation draw per call
        mask_idx = np.random.permutation(self.token_count)[: self.mask_count]
        mask = np.zeros(self.token_count, dtype=int)
        mask[mask_idx] = 1
        mask = mask.reshape((self.rand_size, self.rand_size))
        return mask.repeat(self.scale, axis=0).repeat(self.scale, axis=1)

    @staticmethod
    def bool_mask_to_indices(flat_mask, batch, num_masked):
        """:372-382 -- NOTE the reference slices the row-major list of *column* indices of all
        true entries in chunks of ``num_masked``; when a row holds more (or fewer) than
        ``num_masked`` trues the chunks are misaligned with the rows.  Kept."""
        cols = np.nonzero(flat_mask)[1]
        out = np.empty((batch, num_masked), dtype=np.int64)
        for b in range(batch):
            out[b, :] = cols[num_masked * b: num_masked * (b + 1)]
        return out

    def get_batch(self, batch_size, channel_tokens, num_masked):
        """:384-402 -- independent draw per (sample, spectral block)."""
        m = np.stack([self().astype(bool) for _ in range(batch_size * channel_tokens)])
        flat = m.reshape(batch_size, channel_tokens, -1).reshape(batch_size, -1)
        return flat, self.bool_mask_to_indices(flat, batch_size, num_masked)

    def get_batch_tube_masked(self, batch_size, channel_tokens, num_masked):
        """:404-416 -- one draw per sample, repeated over the spectral blocks."""
        m = np.stack([self().astype(bool) for _ in range(batch_size)])  # [B, h, w]
        m = np.repeat(m[:, None], channel_tokens, axis=1)
        flat = m.reshape(batch_size, -1)
        return flat, self.bool_mask_to_indices(flat, batch_size, num_masked)


def make_masks(batch, num_spectral, num_spatial_sqrt, masking_ratio, mask_patch_size, tube_masking,
               model_patch_size=1):
    """Dispatch of reference src/vit_simmim_original.py:252-282.  Returns (bool [B,T], int64 [B,K])
    as torch CPU tensors."""
    T = num_spectral * num_spatial_sqrt ** 2
    num_masked = int(masking_ratio * T)
    if mask_patch_size == 1:
        idx = torch.rand(batch, T).topk(k=num_masked, dim=-1).indices
        bm = torch.zeros((batch, T)).scatter_(-1, idx, 1).bool()
        return bm, idx
    gen = MaskGeneratorOracle(
        input_size=num_spatial_sqrt * model_patch_size, mask_patch_size=mask_patch_size,
        model_patch_size=model_patch_size, mask_ratio=masking_ratio,
    )
    if tube_masking:
        bm, idx = gen.get_batch_tube_masked(batch, num_spectral, num_masked)
    else:
        bm, idx = gen.get_batch(batch, num_spectral, num_masked)
    return torch.from_numpy(bm), torch.from_numpy(idx)
