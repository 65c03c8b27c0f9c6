"""kernel study (round 5, VERDICT r4 item 2: the 'saved q / k / v' column of the recompute table, measured on both sides).
Needs the kernel-study library next to the product one, built in the build container:
    python -c "from maskedsst_amd.build import build; build(force=True, extra_flags=('-DMSST_LAB','-DMSST_LAB_QKV'), lib='maskedsst_amd/libmsst_lab.so', tag='lab')"
With a scratch registered (msst_debug_stamps) that build's forward STORES every head's q / k / v operand fragments (24.5 KB per tile and
head: 1 GB per block at the bench shape) and its attention backward FETCHES 8 KB per q / k / v wave and tile from it by LDS-DMA instead
of running the three projections (garbage values: timing only).  Same process, same library, scratch on / off alternating:
prints us per launch of the forward and of the attention backward for both."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MSST_ALLOW_LAB"] = "1"
os.environ["MSST_FWD_STACK"] = "0"
import torch
from maskedsst_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "maskedsst_amd", "libmsst_lab.so")
from util import build_product

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(bands=200, depth=2, B=B)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
model.train()
eng = model.engine()
masks = model.draw_masks(B)
drop = (0.1, 5)
H = eng.enc.heads
tiles = max(int(eng.lib.msst_block_lse_floats(m, B, eng.S, eng.N, 1)) // 64 for m in (0, 1))
scratch = torch.empty(tiles * H * 24576 + 65536, dtype=torch.uint8, device="cuda")
print(f"scratch {scratch.numel() / 1e9:.2f} GB for {tiles} tiles x {H} heads", flush=True)
out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=drop)
dy = torch.randn_like(out["enc_out"]) * 1e-3


def timed(on):
    assert eng.lib.msst_debug_stamps(ctypes.c_void_p(scratch.data_ptr() if on else 0)) == 0
    eng.lib.msst_profile_enable(1)
    for _ in range(6):
        o = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=drop)
        eng.blocks_bwd(o["acts"], o["x1s"], dy.clone(), drop=drop)
    torch.cuda.synchronize()
    n = eng.lib.msst_profile_kernels()
    tot = (ctypes.c_double * n)(); cnt = (ctypes.c_long * n)()
    eng.lib.msst_profile_collect(tot, cnt)
    eng.lib.msst_profile_enable(0)
    return {eng.lib.msst_profile_name(i).decode(): 1e3 * tot[i] / max(cnt[i], 1) for i in range(n) if cnt[i]}


res = {0: [], 1: []}
for rnd in range(5):
    for on in (0, 1):
        d = timed(on)
        if rnd:
            res[on].append(d)
eng.lib.msst_debug_stamps(ctypes.c_void_p(0))
for k in ("block_fwd", "block_bwd_attn", "block_bwd_ln1mlp"):
    a = sorted(r[k] for r in res[0])[len(res[0]) // 2]; b = sorted(r[k] for r in res[1])[len(res[1]) // 2]
    print(f"{k}: recompute {a:.1f} us, q/k/v through HBM {b:.1f} us ({b - a:+.1f})")
