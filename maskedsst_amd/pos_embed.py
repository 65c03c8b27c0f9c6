"""Sin-cos position tables used to initialise ``pos_embed`` / ``channel_embed`` (init-time, host).

Same tables as reference ``src/pos_embed.py:16-63`` (which uses the ``np.float`` alias removed in
numpy >= 1.24 and therefore crashes on current numpy); computed here in float64 and built from
outer products of positions and inverse frequencies.
"""
import numpy as np


def _inv_freq(half_dim):
    return np.power(10000.0, -np.arange(half_dim, dtype=np.float64) / float(half_dim))


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """(M,) positions -> (M, embed_dim): first half sin, second half cos."""
    if embed_dim % 2:
        raise AssertionError("embed_dim must be even")
    ang = np.outer(np.asarray(pos, dtype=np.float64).reshape(-1), _inv_freq(embed_dim // 2))
    return np.hstack([np.sin(ang), np.cos(ang)])


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """(grid_size**2 [+1], embed_dim): first half encodes the column (w) index, second half the
    row (h) index -- the reference's ``np.meshgrid(grid_w, grid_h)`` ordering."""
    if embed_dim % 2:
        raise AssertionError("embed_dim must be even")
    rows, cols = np.divmod(np.arange(grid_size * grid_size), grid_size)
    emb = np.hstack([
        get_1d_sincos_pos_embed_from_grid(embed_dim // 2, cols.astype(np.float32)),
        get_1d_sincos_pos_embed_from_grid(embed_dim // 2, rows.astype(np.float32)),
    ])
    if cls_token:
        emb = np.vstack([np.zeros((1, embed_dim)), emb])
    return emb
