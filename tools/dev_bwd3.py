"""dev: round-3 attention backward (msst_bwd3.hip) against the round-2 kernel (MSST_DBG=32) and the template kernel
(MSST_DBG=16) on the same inputs; with a --stamps build (python -m maskedsst_amd.build --stamps) and `dump`, decodes the
LDS images one workgroup leaves after each phase and compares them with a torch restatement of that tile.

usage (GPU box):  python tools/dev_bwd3.py [dump] [time]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from util import build_product, rel_l2

R3 = dict(Q=0, K=8192, DO=16384, V=24576, P=32768, DS=40960, XN=49152, DA=61440, SMEM=73728)


def fz(r):
    return (((r >> 1) & 1) << 2) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 3) & 1)


def fz2(r):
    return (((r >> 3) & 1) << 1) | ((r >> 2) & 1)


def decode64(img, base):
    """[64][64] bf16 tile at byte offset base of an LDS image (uint8 array) -> float32"""
    out = np.zeros((64, 64), np.uint16)
    u16 = img.view(np.uint16)
    for r in range(64):
        for s in range(8):
            o = (base + r * 128 + ((s ^ fz(r)) << 4)) // 2
            out[r, 8 * s:8 * s + 8] = u16[o:o + 8]
    return bf16_to_f32(out)


def decode96(img, base):
    out = np.zeros((64, 96), np.uint16)
    u16 = img.view(np.uint16)
    for r in range(64):
        for s in range(12):
            sw = (s & ~3) | ((s & 3) ^ fz2(r))
            o = (base + r * 192 + (sw << 4)) // 2
            out[r, 8 * s:8 * s + 8] = u16[o:o + 8]
    return bf16_to_f32(out)


def bf16_to_f32(u16):
    return torch.from_numpy((u16.astype(np.uint32) << 16).view(np.float32).copy())


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max())


def tile_tokens(mode, tile, S, N, B):
    """token index (or -1) of the 64 rows of a tile (TileMap of msst_dev.h)"""
    T = S * N
    L = N if mode == 0 else S
    TS = 64 // L
    nseq = B * S if mode == 0 else B * N
    toks = []
    for r in range(64):
        s, p = r // L, r % L
        q = tile * TS + s
        if s >= TS or q >= nseq:
            toks.append(-1)
        elif mode == 0:
            toks.append(q * N + p)
        else:
            b, n = q // N, q % N
            toks.append(b * T + p * N + n)
    return toks, L


def main():
    do_dump = "dump" in sys.argv
    do_time = "time" in sys.argv
    cfg = dict(bands=200, depth=1, B=5)
    if "small" in sys.argv:
        cfg = dict(bands=50, depth=1, B=3)
    torch.manual_seed(1)
    model, params, x = build_product(cfg, precision="bf16", device="cuda")
    eng = model.engine()
    masks = model.draw_masks(cfg["B"])
    for drop in ((0.0, 0), (0.1, 777)):
        out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=drop)
        dy = torch.randn_like(out["enc_out"]) * 1e-3

        def run(flag):
            os.environ["MSST_DBG"] = str(flag)
            dx0 = eng.blocks_bwd(out["acts"], out["x1s"], dy.clone(), drop=drop)
            torch.cuda.synchronize()
            os.environ["MSST_DBG"] = "0"
            return dx0.clone(), eng.fp.grad.clone()

        dx_new, g_new = run(0)
        dx_r2, g_r2 = run(128)   # the one-head-per-workgroup kernel
        dx_t, g_t = run(16)
        print(f"== drop {drop}: dx new vs r2 {rel_l2(dx_new, dx_r2):.3e}  new vs template {rel_l2(dx_new, dx_t):.3e}  r2 vs template {rel_l2(dx_r2, dx_t):.3e}")
        worst = []
        for name, p in eng.trainable():
            a, b, c = eng.fp.view(name, g_new), eng.fp.view(name, g_r2), eng.fp.view(name, g_t)
            if float(c.abs().max()) == 0.0:
                continue
            worst.append((rel_l2(a, c), rel_l2(b, c), name))
        worst.sort(reverse=True)
        for e1, e2, n in worst[:8]:
            print(f"   new-vs-template {e1:.3e}   r2-vs-template {e2:.3e}   {n}")
        print("   any nan:", bool(torch.isnan(g_new).any()), bool(torch.isnan(dx_new).any()))

    if do_dump:
        dump_stages(eng, cfg, x, masks)
    if do_time:
        cfg = dict(bands=200, depth=1, B=256)
        model, params, x = build_product(cfg, precision="bf16", device="cuda")
        eng = model.engine()
        masks = model.draw_masks(cfg["B"])
        tdrop = (0.0, 0) if "nodrop" in sys.argv else (0.1, 5)
        out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=tdrop)
        dy = torch.randn_like(out["enc_out"]) * 1e-3
        res = {0: [], 128: []}
        for rnd in range(4):
            for flag in (0, 128):
                os.environ["MSST_DBG"] = str(flag)
                eng.lib.msst_profile_enable(1)
                for _ in range(8):
                    eng.blocks_bwd(out["acts"], out["x1s"], dy.clone(), drop=tdrop)
                torch.cuda.synchronize()
                n = eng.lib.msst_profile_kernels()
                tot = (ctypes.c_double * n)()
                cnt = (ctypes.c_long * n)()
                eng.lib.msst_profile_collect(tot, cnt)
                eng.lib.msst_profile_enable(0)
                d = {eng.lib.msst_profile_name(i).decode(): round(1e3 * tot[i] / max(cnt[i], 1), 1) for i in range(n) if cnt[i]}
                if rnd:
                    res[flag].append(d)
        for flag in (0, 128):
            print("flag", flag, {k: min(r[k] for r in res[flag]) for k in res[flag][0]})
        os.environ["MSST_DBG"] = "0"


def dump_stages(eng, cfg, x, masks):
    """needs a --stamps build; decodes the five LDS images of workgroup (chunk 0, head hsel), first tile"""
    from maskedsst_amd._lib import MODE_SPATIAL, MODE_SPECTRAL
    H = eng.enc.heads
    B, S, N = cfg["B"], eng.S, eng.N
    buf = torch.zeros(5 * R3["SMEM"] // 8 + 64, dtype=torch.int64, device="cuda")
    rc = eng.lib.msst_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
    if rc != 0:
        print("no --stamps build: skipping the LDS dumps")
        return
    out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=(0.0, 0))
    dy = torch.randn_like(out["enc_out"]) * 1e-3
    hsel = 3
    layers = eng._layers()
    for i in (1, 0):   # spectral block, then spatial block
        sname, l = layers[i]
        mode = MODE_SPATIAL if sname == "spatial" else MODE_SPECTRAL
        # run the backward down to block i so that dab / everything for block i is what the kernel saw
        os.environ["MSST_DBG"] = str(8 | (hsel << 8))
        acts, x1s = out["acts"], out["x1s"]
        ntok = B * S * N
        dx1 = torch.empty(ntok * 96, dtype=torch.float32, device="cuda")
        part = torch.empty(H * ntok * 96 * 2, dtype=torch.uint8, device="cuda")
        dab = torch.empty(ntok * 96, dtype=torch.bfloat16, device="cuda")
        from maskedsst_amd._lib import MLP_SLAB, ATTN_SLAB, LN1_SLAB
        slab = torch.empty(eng.grid_rows * (2 * MLP_SLAB + LN1_SLAB) + eng.attn_chunks * H * ATTN_SLAB, dtype=torch.float32, device="cuda")
        other = torch.empty_like(dy)
        from maskedsst_amd.engine import _p, _stream, _kernel_flags
        from maskedsst_amd import _lib
        g_in = dy.clone()
        buf.zero_()
        _lib.check(eng.lib.msst_block_bwd(
            ctypes.byref(eng._bw[i]), ctypes.byref(eng._bg[i]), _p(acts[i]), _p(x1s[i]), _p(g_in), _p(other),
            _p(dx1), _p(part), _p(slab), eng.grid_rows, eng.attn_chunks, mode, B, S, N, H,
            eng.prec | _kernel_flags() | (_lib.X1_BF16 if x1s[i].dtype == torch.bfloat16 else 0), 0.0, 0, i, _p(getattr(x1s[i], "_msst_xn", None)), _p(getattr(x1s[i], "_msst_lse", None)), _p(dab), _stream()), "msst_block_bwd")
        torch.cuda.synchronize()
        os.environ["MSST_DBG"] = "0"
        img = buf.cpu().numpy().view(np.uint8)
        imgs = [img[s * R3["SMEM"]:(s + 1) * R3["SMEM"]] for s in range(5)]
        toks, L = tile_tokens(mode, 0, S, N, B)
        xn_all = x1s[i]._msst_xn.reshape(-1, 96).float().cpu()
        da_all = dab.reshape(-1, 96).float().cpu()
        xn_t = torch.stack([xn_all[t] if t >= 0 else torch.zeros(96) for t in toks])
        da_t = torch.stack([da_all[t] if t >= 0 else torch.zeros(96) for t in toks])
        wqkv = rb(eng.fp.view(f"{sname}.{l}.wqkv", eng.fp.flat).reshape(3 * H * 64, 96).float().cpu())
        wout = rb(eng.fp.view(f"{sname}.{l}.wout", eng.fp.flat).reshape(96, H * 64).float().cpu())
        Wq, Wk, Wv = (wqkv[(w * H + hsel) * 64:(w * H + hsel) * 64 + 64] for w in range(3))
        Wo = wout[:, hsel * 64:hsel * 64 + 64]   # [m][d]
        print(f"---- block {sname}.{l} (mode {mode}, L {L}) head {hsel} tile 0")
        print("  XN ", err(decode96(imgs[0], R3["XN"]), xn_t), " DA ", err(decode96(imgs[0], R3["DA"]), da_t))
        q, k, v, dO = rb(xn_t @ Wq.T), rb(xn_t @ Wk.T), rb(xn_t @ Wv.T), rb(da_t @ Wo)
        print("  Q  ", err(decode64(imgs[1], R3["Q"]), q), " K ", err(decode64(imgs[1], R3["K"]), k),
              " V ", err(decode64(imgs[1], R3["V"]), v), " DO ", err(decode64(imgs[1], R3["DO"]), dO))
        s = (q @ k.T) * 0.125
        seq = torch.tensor([r // L if r < (64 // L) * L else -1 for r in range(64)])
        maskm = seq[:, None] == seq[None, :]
        s = torch.where(maskm, s, torch.full_like(s, -1e30))
        p = torch.softmax(s, dim=1)
        pad = seq < 0
        dp = dO @ v.T
        ds = p * (dp - (p * dp).sum(1, keepdim=True)) * 0.125
        gp, gds = decode64(imgs[2], R3["P"]), decode64(imgs[2], R3["DS"])
        valid = ~pad
        print("  P  ", err(gp[valid], rb(p)[valid]), " DS ", err(gds[valid], rb(ds)[valid]))
        p, ds = rb(p), rb(ds)
        p[pad] = gp[pad]; ds[pad] = gds[pad]   # padding queries: whatever the kernel produced (they multiply zero rows)
        dq, dk, dv = rb(ds @ k), rb(ds.T @ q), rb(p.T @ dO)
        print("  dqT", err(decode64(imgs[3], R3["K"]), dq.T), " dkT ", err(decode64(imgs[3], R3["Q"]), dk.T),
              " dvT ", err(decode64(imgs[3], R3["DO"]), dv.T))
        outr = dq @ Wq + dk @ Wk + dv @ Wv
        print("  OUT", err(decode96(imgs[4], R3["V"]), outr))


if __name__ == "__main__":
    main()
