"""Sin-cos position tables (oracle; test infrastructure only).

Restates reference ``src/pos_embed.py:16-63`` with ``np.float64`` in place of the removed
``np.float`` alias (the reference crashes on numpy>=1.24).
"""
import numpy as np


def sincos_1d(embed_dim, pos):
    """reference src/pos_embed.py:45-63 -- (M,) positions -> (M, embed_dim) [sin | cos]."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    pos = np.asarray(pos).reshape(-1)
    out = np.einsum("m,d->md", pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim, grid_size):
    """reference src/pos_embed.py:16-42 -- (grid*grid, embed_dim); w-grid first (meshgrid order)."""
    assert embed_dim % 2 == 0
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    emb_h = sincos_1d(embed_dim // 2, grid[0])
    emb_w = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)
