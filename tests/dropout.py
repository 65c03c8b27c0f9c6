"""numpy restatement of the kernels' stateless dropout masks (maskedsst_amd/csrc/msst_dev.h: Drop /
drop_bits / drop4) so that tests can hand the CPU oracle exactly the masks the HIP kernels use."""
import numpy as np
import torch


def drop_bits(key, group):
    """msst_dev.h drop_bits: two 32-bit words = four 16-bit fields per element group"""
    with np.errstate(over="ignore"):
        x = (group.astype(np.uint32) ^ key).astype(np.uint32)
        x = (x * np.uint32(0x9E3779B1)).astype(np.uint32); x ^= x >> np.uint32(15)
        x = (x * np.uint32(0x85EBCA6B)).astype(np.uint32); x ^= x >> np.uint32(13)
        a = x.copy()
        b = (x * np.uint32(0xC2B2AE35)).astype(np.uint32); b ^= b >> np.uint32(16)
    return a, b


def keep_scaled(p, seed, layer, site, group, elem):
    """group, elem: integer arrays (element-group index and position 0..3 inside it) -> float32 array of
    0 or 1/(1-p') where p' = round(p*65536)/65536"""
    thr = np.uint32(int(p * 65536.0 + 0.5))
    scale = np.float32(1.0 / (1.0 - float(thr) / 65536.0))
    key = np.uint32(seed) ^ np.uint32((np.uint64(layer * 4 + site) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF))
    a, b = drop_bits(key, group)
    bits = np.where(elem == 0, a & np.uint32(0xffff), np.where(elem == 1, a >> np.uint32(16),
                    np.where(elem == 2, b & np.uint32(0xffff), b >> np.uint32(16))))
    return np.where(bits >= thr, scale, np.float32(0.0)).astype(np.float32)


def block_masks(p, seed, layer, mode, B, S, N, heads):
    """masks for one block in the oracle's sequence layout: mode 0 spatial [B*S, N, .], mode 1 spectral [B*N, S, .]"""
    L = N if mode == 0 else S
    nseq = B * S if mode == 0 else B * N
    T = S * N
    TS = 64 // L
    q = np.arange(nseq)
    pos = np.arange(L)
    if mode == 0:
        tok = q[:, None] * N + pos[None, :]
    else:
        b, n = q // N, q % N
        tok = b[:, None] * T + pos[None, :] * N + n[:, None]          # [nseq, L]
    out = {}
    for site, width in ((2, 96), (3, 64), (4, 96)):
        col = np.arange(width)
        grp = tok[:, :, None] * (width // 4) + col[None, None, :] // 4
        out[site] = torch.from_numpy(keep_scaled(p, seed, layer, site, grp, np.broadcast_to(col % 4, grp.shape)))
    tile, slot = q // TS, q % TS
    i = np.arange(L)
    qrow = slot[:, None] * L + i[None, :]                              # [nseq, L]
    key = qrow
    h = np.arange(heads)
    grp = (((tile[:, None, None, None] * heads + h[None, :, None, None]) * 64 + qrow[:, None, :, None]) * 16
           + key[:, None, None, :] // 4)
    el = np.broadcast_to(key[:, None, None, :] % 4, grp.shape)
    out[1] = torch.from_numpy(keep_scaled(p, seed, layer, 1, grp, el))
    return out


def make_drop_fn(p, seed, S, N, heads):
    return lambda layer, mode, B: block_masks(p, seed, layer, mode, B, S, N, heads)
