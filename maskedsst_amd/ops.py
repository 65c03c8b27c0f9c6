"""Stand-alone ops of the hot path behind the C-ABI (``include/msst.h``), as autograd functions.

On the pre-training path every one of these runs fused into a larger kernel (``engine.py``); the functions here serve a
caller that needs the op by itself -- the reference's ``nn.LayerNorm`` of ``PreNorm`` / ``BlockwisePatchEmbedding``
(``vit_spatial_spectral.py:25,194-195``) -- and are the unit the fused kernels are checked against.  HIP only: a CPU
tensor raises (no eager fallback).
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        if not x.is_cuda:
            raise RuntimeError("maskedsst_amd.ops.layer_norm runs on an MI355X only (tensor is on %s); there is no CPU fallback" % x.device)
        D = x.shape[-1]
        if weight.shape != (D,) or bias.shape != (D,):
            raise ValueError("LayerNorm over the last axis: weight / bias must be [%d]" % D)
        xc = x.contiguous().float()
        w, b = weight.contiguous().float(), bias.contiguous().float()
        y = torch.empty_like(xc)
        rows = xc.numel() // D
        _lib.check(lib.msst_layernorm_fwd(_p(xc), _p(w), _p(b), _p(y), _p(None), _p(None), rows, D, float(eps), _stream()),
                   "msst_layernorm_fwd")
        ctx.save_for_backward(xc, w)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xc, w = ctx.saved_tensors
        D = xc.shape[-1]
        rows = xc.numel() // D
        dy = dy.contiguous().float()
        dx = torch.empty_like(xc)
        dg = torch.empty(D, dtype=torch.float32, device=xc.device)
        db = torch.empty(D, dtype=torch.float32, device=xc.device)
        slab = torch.empty(max(1, int(lib.msst_layernorm_bwd_slab(rows, D))), dtype=torch.float32, device=xc.device)
        _lib.check(lib.msst_layernorm_bwd(_p(xc), _p(w), _p(dy), _p(dx), _p(dg), _p(db), _p(slab), rows, D, ctx.eps, _stream()),
                   "msst_layernorm_bwd")
        return dx, dg, db, None


def layer_norm(x, weight, bias, eps=1e-5):
    """``F.layer_norm(x, (D,), weight, bias, eps)`` over the last axis (D <= 128) on the HIP kernels of msst_ln.hip."""
    return _LayerNormFn.apply(x, weight, bias, eps)
