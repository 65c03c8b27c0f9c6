"""kernel study: cycle stamps of one wave of block_bwd_attn_r3 (msst_bwd3.hip), per wave role, spatial and spectral block.
needs a stamps build first:  python -m maskedsst_amd.build --stamps   (MSST_DBG = 8 | wave << 8 selects the stamped wave)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import build_product
from maskedsst_amd import _lib
from maskedsst_amd._lib import MODE_SPATIAL, MODE_SPECTRAL, MLP_SLAB, ATTN_SLAB, LN1_SLAB
from maskedsst_amd.engine import _p, _stream, _kernel_flags

cfg = dict(bands=200, depth=1, B=256)
model, params, x = build_product(cfg, precision="bf16", device="cuda")
eng = model.engine()
masks = model.draw_masks(cfg["B"])
drop = (0.1, 5)
out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1], drop=drop)
dy = torch.randn_like(out["enc_out"]) * 1e-3
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
assert eng.lib.msst_debug_stamps(ctypes.c_void_p(buf.data_ptr())) == 0, "build with --stamps"
names = ["top", "phase 1", "B1", "phase 2", "B2", "phase 3", "B3", "phase 4", "B4", "copy-out"]
IDX = [0, 2, 3, 4, 5, 6, 7, 8, 9, 10]
B, S, N, H = cfg["B"], eng.S, eng.N, eng.enc.heads
ntok = B * S * N
acts, x1s = out["acts"], out["x1s"]
dx1 = torch.empty(ntok * 96, dtype=torch.float32, device="cuda")
part = torch.empty(H * ntok * 96 * 2, dtype=torch.uint8, device="cuda")
dab = torch.empty(ntok * 96, dtype=torch.bfloat16, device="cuda")
slab = torch.empty(eng.grid_rows * (2 * MLP_SLAB + LN1_SLAB) + eng.attn_chunks * H * ATTN_SLAB, dtype=torch.float32, device="cuda")
other = torch.empty_like(dy)
for i in (0, 1):
    sname, l = eng._layers()[i]
    mode = MODE_SPATIAL if sname == "spatial" else MODE_SPECTRAL
    for wave in range(4):
        os.environ["MSST_DBG"] = str(8 | (wave << 8))
        for rep in range(2):
            buf.zero_()
            _lib.check(eng.lib.msst_block_bwd(
                ctypes.byref(eng._bw[i]), ctypes.byref(eng._bg[i]), _p(acts[i]), _p(x1s[i]), _p(dy), _p(other),
                _p(dx1), _p(part), _p(slab), eng.grid_rows, eng.attn_chunks, mode, B, S, N, H,
                eng.prec | _kernel_flags() | (_lib.X1_BF16 if x1s[i].dtype == torch.bfloat16 else 0), drop[0], drop[1], i, _p(getattr(x1s[i], "_msst_xn", None)), _p(getattr(x1s[i], "_msst_lse", None)), _p(dab), _stream()), "msst_block_bwd")
            torch.cuda.synchronize()
        s = buf.cpu().numpy()
        d = [int(s[IDX[k + 1]] - s[IDX[k]]) for k in range(len(IDX) - 1)]
        print(f"{sname} wave {'QKVO'[wave]}: total {int(s[10] - s[0]):6d} | " + " ".join(f"{n}:{v}" for n, v in zip(names[1:], d)))
os.environ["MSST_DBG"] = "0"
