"""Host-side mask generation for SimMIM (bit-exact with the reference).

Mirrors ``MaskGenerator`` of reference ``src/vit_simmim_original.py:343-416`` and the
``mask_patch_size == 1`` top-k branch (``:254-264``).  Masks stay on the host on purpose: the
reference draws them from the numpy *global* RNG (``np.random.permutation``, one call per
sample -- or per (sample, spectral block) without tube masking) and from the torch CPU RNG, and the
parity contract is "mask indices bit-exact" (BASELINE.json north_star).  This implementation makes
exactly the same RNG calls in the same order and vectorises everything else.

Data-parallel runs: every rank seeds identically, generates the masks of the GLOBAL batch and
slices its own rows, which reproduces the single-process reference on the global batch including
the cross-row coupling of ``bool_mask_to_indices``.
"""
import numpy as np
import torch


class MaskGenerator:
    """Same constructor / attributes / call contract as the reference class (``:343-370``)."""

    def __init__(self, input_size=16, mask_patch_size=4, model_patch_size=1, mask_ratio=0.6):
        self.input_size = input_size
        self.mask_patch_size = mask_patch_size
        self.model_patch_size = model_patch_size
        self.mask_ratio = mask_ratio
        assert self.input_size % self.mask_patch_size == 0
        assert self.mask_patch_size % self.model_patch_size == 0
        self.rand_size = self.input_size // self.mask_patch_size
        self.scale = self.mask_patch_size // self.model_patch_size
        self.token_count = self.rand_size ** 2
        self.mask_count = int(np.ceil(self.token_count * self.mask_ratio))

    def __call__(self):
        return self._draw(1)[0].astype(int)

    def _draw(self, n):
        """n masks [n, H, W] (bool); one ``np.random.permutation`` call per mask, in order."""
        sel = np.empty((n, self.mask_count), dtype=np.int64)
        for i in range(n):
            sel[i] = np.random.permutation(self.token_count)[: self.mask_count]
        coarse = np.zeros((n, self.token_count), dtype=bool)
        coarse[np.arange(n)[:, None], sel] = True
        coarse = coarse.reshape(n, self.rand_size, self.rand_size)
        return coarse.repeat(self.scale, axis=1).repeat(self.scale, axis=2)

    @staticmethod
    def _indices(flat, batch, num_masked):
        """``bool_mask_to_indices`` (:372-382): the row-major list of column indices of all true
        entries, cut into consecutive chunks of ``num_masked`` -- misaligned with the rows whenever
        a row holds != num_masked trues (kept on purpose; raises like the reference if too few)."""
        cols = np.nonzero(flat)[1]
        if cols.shape[0] < batch * num_masked:
            raise RuntimeError(
                f"shape mismatch: {cols.shape[0]} masked entries for {batch} x {num_masked} indices")
        return cols[: batch * num_masked].reshape(batch, num_masked).astype(np.int64)

    def bool_mask_to_indices(self, masked_bool_mask, batch, num_masked, device=None):
        m = masked_bool_mask.cpu().numpy() if torch.is_tensor(masked_bool_mask) else np.asarray(masked_bool_mask)
        return torch.from_numpy(self._indices(m, batch, num_masked))

    def get_batch(self, batch_size, channel_tokens, num_masked, device=None):
        m = self._draw(batch_size * channel_tokens).reshape(batch_size, -1)
        return torch.from_numpy(m), torch.from_numpy(self._indices(m, batch_size, num_masked))

    def get_batch_tube_masked(self, batch_size, channel_tokens, num_masked, device=None, rows=None):
        """Tube masking (:404-416): one spatial mask per sample, repeated over the spectral tokens.

        Every row holds the same number of trues (R = channel_tokens * mask_count * scale^2), so the
        reference's row-major nonzero list is known in closed form: row r contributes
        ``c * HW + pos_r[j]`` for c-major, position-ascending order, at offsets [R r, R (r + 1)).  The
        index rows are cut from that list without materialising the [B, T] bool matrix first, and
        ``rows = (lo, hi)`` restricts both outputs to the local rows of a data-parallel rank while the
        RNG is still advanced for the whole global batch (one permutation per sample, in order)."""
        n = batch_size
        sel = np.empty((n, self.mask_count), dtype=np.int64)
        for i in range(n):
            sel[i] = np.random.permutation(self.token_count)[: self.mask_count]
        lo, hi = (0, n) if rows is None else rows
        hw = self.input_size // self.model_patch_size
        HW = hw * hw
        # positions (row-major in the hw x hw grid) covered by each coarse cell
        cell = np.arange(self.token_count)
        cy, cx = cell // self.rand_size, cell % self.rand_size
        dy, dx = np.meshgrid(np.arange(self.scale), np.arange(self.scale), indexing="ij")
        table = ((cy[:, None, None] * self.scale + dy) * hw + (cx[:, None, None] * self.scale + dx)).reshape(self.token_count, -1)
        per = table.shape[1] * self.mask_count                      # trues per spatial mask
        R = per * channel_tokens                                    # trues per row of the [B, T] mask
        if n * R < n * num_masked:
            raise RuntimeError(f"shape mismatch: {n * R} masked entries for {n} x {num_masked} indices")
        # index rows lo..hi-1 read list positions [num_masked lo, num_masked hi) -> source rows r0..r1-1
        r0, r1 = (num_masked * lo) // R, -(-(num_masked * hi) // R)
        need = np.arange(min(r0, lo), max(r1, hi))                 # rows whose positions are needed at all
        pos = np.sort(table[sel[need]].reshape(need.shape[0], per), axis=1)
        base = need[0]
        cols = (np.arange(channel_tokens)[None, :, None] * HW + pos[r0 - base:r1 - base, None, :]).reshape(-1)
        idx = cols[num_masked * lo - R * r0: num_masked * hi - R * r0].reshape(hi - lo, num_masked).astype(np.int64)
        m = np.zeros((hi - lo, HW), dtype=bool)
        m[np.arange(hi - lo)[:, None], pos[lo - base:hi - base]] = True
        m = np.tile(m, (1, channel_tokens))
        return torch.from_numpy(m), torch.from_numpy(idx)


def topk_masks(batch, num_patches, num_masked):
    """``mask_patch_size == 1`` branch (:254-264) on the torch CPU generator."""
    idx = torch.rand(batch, num_patches).topk(k=num_masked, dim=-1).indices
    bm = torch.zeros((batch, num_patches)).scatter_(-1, idx, 1).bool()
    return bm, idx


def inverse_csr(idx, num_tokens):
    """For the gather ``enc[b, idx[b, k]]`` build, per row, the token-sorted list of positions:
    ``ptr`` [B, T+1] and ``pos`` [B, K] (int32) such that the positions k with idx[b, k] == t are
    ``pos[b, ptr[b, t]:ptr[b, t+1]]``.  The backward of the gather is then a deterministic
    segmented sum (duplicates -- which the misaligned slicing does produce -- just make longer
    segments) instead of float atomics."""
    idx = np.asarray(idx)
    B, K = idx.shape
    key = idx.astype(np.uint16) if num_tokens < 65536 else idx     # 16-bit keys: numpy's stable sort is a radix sort
    pos = np.argsort(key, axis=1, kind="stable").astype(np.int32)
    flat = (np.arange(B, dtype=np.int64)[:, None] * (num_tokens + 1) + idx + 1).ravel()
    cnt = np.bincount(flat, minlength=B * (num_tokens + 1)).reshape(B, num_tokens + 1)
    ptr = np.cumsum(cnt, axis=1).astype(np.int32)
    return ptr, pos
