"""Host-side engine: owns the flat parameter buffers, the operand-layout weight copies and the
launch sequence of the HIP kernels (through the C-ABI in ``libmsst.so``) for one model.

PyTorch is used for device memory (``torch.empty``), streams and autograd bookkeeping only; all
arithmetic of the hot path runs in the kernels under ``maskedsst_amd/csrc``.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import (MsstBlockWeights, MsstBlockGrads, MsstPrepJob, PREC_BF16, PREC_F32, MODE_SPATIAL,
                   MODE_SPECTRAL, MLP_SLAB, ATTN_SLAB, LN1_SLAB)
from .flat import FlatParams

D = 96
DH = 64
STACK_MAX_TILES = 12   # msst_block_fwd_stack is used for a stack whose workgroups hold at most this many tiles (crossover measured near 14; batch 256 at the EnMAP shape: 20 / 22)
MLP = 64


def _prec_of(name):
    name = (name or os.environ.get("MSST_PRECISION", "bf16")).lower()
    if name in ("fp32", "f32", "32-true", "float32"):
        return PREC_F32
    if name in ("bf16", "bf16-mixed", "bfloat16"):
        return PREC_BF16
    raise ValueError(f"unknown precision {name!r} (use 'bf16' or 'fp32')")


def _kernel_flags():
    """MSST_KERNEL_* selection flags (include/msst.h), read from the environment on the HOST side per call: the
    library itself never reads the environment.  MSST_DBG=16 selects the generic template kernels, 64 the 4-wave
    forward, 128 the one-head attention backward; 0 (default) the tuned kernels.  Only those selection bits pass (plus 8 and
    the wave-select bits 0x300 of the -DMSST_STAMPS kernel-study builds): MSST_X1_BF16 / MSST_BWD_DEFER_REDUCE share the
    field and are set by the engine's own logic, never from the environment."""
    v = int(os.environ.get("MSST_DBG", "0"))
    allowed = 16 | 64 | 128
    if v & 8:
        allowed |= 8 | 0x300
    return (v & allowed) << 8


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Engine:
    def __init__(self, encoder, mim):
        self.enc = encoder
        self.mim = mim
        self.lib = _lib.load()
        self.fp = FlatParams(encoder, mim)
        self.prec = _prec_of(getattr(encoder, "precision", None))
        self.max_grid = int(os.environ.get("MSST_MAX_GRID", "0"))
        # persistent grids of the backward: one row-wise workgroup per CU; tile chunks of the attention backward so that its
        # (chunks x heads or head pairs) grid fills every CU slot exactly once (64 x 4 two-head workgroups on 256 CUs)
        self.grid_rows = int(os.environ.get("MSST_BWD_GRID", "0")) or (self._cu_count() if torch.cuda.is_available() else 256)
        self.attn_chunks = int(os.environ.get("MSST_ATTN_CHUNKS", "0")) or \
            (self.default_attn_chunks() if torch.cuda.is_available() else 64)
        self.tok_chunks = int(os.environ.get("MSST_TOK_CHUNKS", "64"))
        self.bucket_hook = None  # callable(bucket_name, start, end) fired when a gradient bucket is complete
        self.tile_queue = False  # data parallel: the backward's persistent grids draw tiles from a queue (no static partition)
        self._tpw = {}   # (mode, batch) -> tiles per workgroup of a block forward
        self._wbuf = None
        self._jobs = None
        self._bw = None
        self._zero_mask = None

    # ------------------------------------------------------------------ setup
    @property
    def S(self):
        return self.enc.num_spectral_patches

    @property
    def N(self):
        return self.enc.num_spatial_patches

    @property
    def P(self):
        return self.enc.pixels_per_patch

    def _cu_count(self):
        try:
            return int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
        except Exception:
            return 256

    def _attn_wgs_per_chunk_and_cu(self):
        """(workgroups per tile chunk, workgroups per CU) of the bf16 attention backward this engine selects: the two-head
        kernel (even head count) launches H/2 workgroups of 512 threads per chunk, one per CU (156 KB of LDS); the one-head
        kernels H workgroups of 256 threads, two per CU."""
        H = max(1, int(self.enc.heads))
        return (H // 2, 1) if H % 2 == 0 else (H, 2)

    def default_attn_chunks(self, free_cus=0):
        per_chunk, per_cu = self._attn_wgs_per_chunk_and_cu()
        return max(1, (per_cu * max(8, self._cu_count() - free_cus)) // per_chunk)

    def reserve_cus(self, n):
        """Leave ``n`` of the chip's CUs free of backward workgroups (data parallel: RCCL's channels).  The persistent grids are
        derived from the device's CU count and from how many workgroups of the selected attention backward fit a CU;
        explicit MSST_ATTN_CHUNKS / MSST_BWD_GRID settings win."""
        total = self._cu_count()
        n = max(0, min(int(n), total - 8))
        if "MSST_ATTN_CHUNKS" not in os.environ:
            self.attn_chunks = self.default_attn_chunks(n)
        if "MSST_BWD_GRID" not in os.environ:
            self.grid_rows = total - n

    def queue_capable(self):
        """Will blocks_bwd run the kernels that can DRAW their tiles (msst_block_bwd_chain with a tile queue: the two-head attention
        backward + the fused LN1 / MLP launch)?  The static part of blocks_bwd's `chain` predicate: bf16 tuned kernels, no kernel
        selection flags, an even head count with at most four head pairs, chaining not switched off.  (fp32, odd or more than
        eight heads, MSST_DBG flags, MSST_BWD_CHAIN=0 fall back to msst_block_bwd: full static grids.)"""
        H = int(self.enc.heads)
        return (self.prec == PREC_BF16 and _kernel_flags() == 0 and H % 2 == 0 and H // 2 <= 4
                and os.environ.get("MSST_BWD_CHAIN", "1") != "0")

    def set_precision(self, name):
        prec = _prec_of(name)
        if prec != self.prec:
            self.prec = prec
            self._wbuf = None

    def _require_cuda(self, t):
        if not t.is_cuda:
            raise RuntimeError(
                "maskedsst_amd runs on an MI355X only (tensor is on %s); there is no CPU fallback" % t.device)

    def ensure(self):
        """flat buffers + operand-layout weight storage + prep job table (rebuilt if params moved)"""
        if self.fp.stale():
            self.fp.flatten()
            self._wbuf = None
        if self._wbuf is None:
            self._build_weight_storage()

    def _layers(self):
        """[(stack name, layer index)] in forward order"""
        L = self.enc.depth
        return [("spatial", l) for l in range(L)] + [("spectral", l) for l in range(L)]

    def _build_weight_storage(self):
        dev = self.fp.flat.device
        self._require_cuda(self.fp.flat)
        H = self.enc.heads
        inner = H * DH
        esz = 4 if self.prec == PREC_F32 else 2
        mats = [("wqkv", 3 * inner, D), ("wout", D, inner), ("w1", MLP, D), ("w2", D, MLP)]
        # bf16: three more copies in the 32-row x 16-k fragment packing of the round-3 attention backward (32x32x16 MFMAs)
        # (name, rows, cols, transpose, field, scale_rows): pack = 1; wqkvT32 carries dim_head^-0.5 in its q and k blocks
        mats32 = [("wqkv", 3 * inner, D, 0, "wqkv32", 0), ("wout", D, inner, 1, "woutT32", 0), ("wqkv", 3 * inner, D, 1, "wqkvT32", 2 * inner)] \
            if self.prec != PREC_F32 else []
        # bf16: and the four forward matrices once more as IEEE half (MSST_FWD_HALF: the fp16-operand forward), same fragment packing
        mats_h = [(name, r, c, name + "_h") for name, r, c in mats] if self.prec != PREC_F32 else []
        per_layer = sum(2 * r * c for _, r, c in mats) + sum(r * c for _, r, c, _, _, _ in mats32) + sum(r * c for _, r, c, _ in mats_h)
        layers = self._layers()
        self._wbuf = torch.empty(per_layer * len(layers) * esz, dtype=torch.uint8, device=dev)
        base = self._wbuf.data_ptr()
        jobs = (MsstPrepJob * ((8 + len(mats32) + len(mats_h)) * len(layers)))()
        self._bw = []
        self._bg = []
        off = 0
        j = 0
        maxel = 0
        for sname, l in layers:
            bw = MsstBlockWeights()
            bw.struct_bytes = ctypes.sizeof(MsstBlockWeights)
            for name, r, c in mats:
                src = self.fp.ptr(f"{sname}.{l}.{name}")
                for tr in (0, 1):
                    dst = base + off * esz
                    jobs[j].src, jobs[j].dst, jobs[j].rows, jobs[j].cols, jobs[j].transpose = src, dst, r, c, tr
                    setattr(bw, name + ("T" if tr else ""), dst)
                    off += r * c
                    j += 1
                    maxel = max(maxel, r * c)
            for name, r, c, tr, field, scale_rows in mats32:
                dst = base + off * esz
                jobs[j].src, jobs[j].dst, jobs[j].rows, jobs[j].cols = self.fp.ptr(f"{sname}.{l}.{name}"), dst, r, c
                jobs[j].transpose, jobs[j].pack = tr, 1
                jobs[j].scale_rows, jobs[j].scale = scale_rows, float(DH) ** -0.5
                dr, dk = (c, r) if tr else (r, c)   # destination rows x contraction length: whole 32 x 16 fragments only
                assert dr % 32 == 0 and dk % 16 == 0, (name, r, c, tr)
                setattr(bw, field, dst)
                off += r * c
                j += 1
            for name, r, c, field in mats_h:
                dst = base + off * esz
                jobs[j].src, jobs[j].dst, jobs[j].rows, jobs[j].cols = self.fp.ptr(f"{sname}.{l}.{name}"), dst, r, c
                jobs[j].transpose, jobs[j].pack = 0, _lib.PREP_HALF
                setattr(bw, field, dst)
                off += r * c
                j += 1
            for name in ("ln1_g", "ln1_b", "bo", "ln2_g", "ln2_b", "b1", "b2"):
                setattr(bw, name, self.fp.ptr(f"{sname}.{l}.{name}"))
            self._bw.append(bw)
            bg = MsstBlockGrads()
            for name in ("ln1_g", "ln1_b", "wqkv", "wout", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2"):
                setattr(bg, name, self.fp.ptr(f"{sname}.{l}.{name}", self.fp.grad))
            self._bg.append(bg)
        raw = bytes(jobs)
        self._jobs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self._njobs = j
        self._maxel = maxel

    def prep_weights(self):
        """fp32 master weights -> operand layout (one launch); call after every parameter update"""
        self.ensure()
        first = getattr(self, "_prep_flag", None) is None or self._prep_flag.device != self._jobs.device
        if first:
            self._prep_flag = torch.zeros(1, dtype=torch.int32, device=self._jobs.device)
            self._prep_checked = None
        _lib.check(self.lib.msst_prep_weights(_p(self._jobs), self._njobs, ctypes.sizeof(MsstPrepJob), self._maxel, self.prec,
                                              _p(self._prep_flag), _stream()), "msst_prep_weights")
        if self._prep_checked is not self._jobs:   # once per job table: did the kernel skip a malformed job? (one sync at setup)
            bad = int(self._prep_flag.item())
            if bad:
                raise _lib.MsstError(f"msst_prep_weights skipped malformed jobs (flags {bad}): operand copies are incomplete")
            self._prep_checked = self._jobs
        if self.prec == PREC_BF16:
            self._ln1_guard_launch()

    # ------------------------------------------------------------------ parameter guards of MSST_LN1_FROM_XN and MSST_FWD_HALF
    LN1_XN_MAX_RATIO = 12.0   # max |ln1_b / ln1_g|: xhat = (row - b) / g amplifies the bf16 rounding of the saved rows by 1 + |b / g| / |xhat|
    HALF_MAX_BOUND = 3.0e4    # bound on |q|, |k|, |v|, |attention output| and the MLP's hidden pre-activation (half's largest finite value: 65504)

    def _ln1_guard_launch(self):
        """Two numbers over every block, computed on the device from the flat parameter buffer (a [blocks, floats per block] view: the
        blocks' parameters lie a constant stride apart) and copied to pinned host memory WITHOUT a synchronisation -- read some steps later
        (parameters move by <= lr per step; both thresholds are orders of magnitude, not margins):
          * max |ln1_b / ln1_g| (inf when a gamma is 0): MSST_LN1_FROM_XN divides by gamma;
          * a bound on every half operand the forward produces (MSST_FWD_HALF): |W x| <= max_row ||W_row||_1 * max |x|, with
            |LayerNorm row| <= max |gamma| * sqrt(96) + max |beta| -- q / k / v (and the attention output, a convex combination of v
            rows) from Wqkv and LN1, the hidden pre-activation (>= |GELU output|) from W1, b1 and LN2."""
        if getattr(self, "_ln1_idx_key", None) is not self._jobs:
            layers = self._layers()
            names = ["ln1_g", "ln1_b", "wqkv", "ln2_g", "ln2_b", "w1", "b1"]
            seg = self.fp.segments
            base = [seg[f"{layers[0][0]}.{layers[0][1]}.{n}"][0] for n in names]
            starts = [min(seg[f"{s_}.{l}.{n}"][0] for n in names) for s_, l in layers]
            lo = min(starts)
            stride = None
            order = sorted(range(len(layers)), key=lambda i: starts[i])
            affine = True
            for a_, b_ in zip(order, order[1:]):
                d = starts[b_] - starts[a_]
                stride = d if stride is None else stride
                affine = affine and d == stride
            rel = [o - starts[0] for o in base]
            for i, (s_, l) in enumerate(layers):
                affine = affine and all(seg[f"{s_}.{l}.{n}"][0] - starts[i] == r for n, r in zip(names, rel))
            span = max(seg[f"{layers[0][0]}.{layers[0][1]}.{n}"][0] + seg[f"{layers[0][0]}.{layers[0][1]}.{n}"][1] for n in names) - starts[0]
            self._guard_view = (lo, stride if len(layers) > 1 else span, len(layers), dict(zip(names, rel)), span) if affine and (stride or 0) >= span else None
            self._ln1_idx_key = self._jobs
            self._ln1_host = torch.empty(2, dtype=torch.float32, pin_memory=True)
            self._ln1_ev = None
            self._ln1_ratio = None
            self._half_bound = None
            self._ln1_calls = 0
        if self._guard_view is None:
            return                                                     # (not our own flat layout: both switches stay off)
        self._ln1_calls += 1
        if self._ln1_ev is not None and self._ln1_ev.query():      # the copy launched some parameter updates ago has landed: adopt its values
            self._ln1_ratio, self._half_bound = float(self._ln1_host[0]), float(self._ln1_host[1])
            self._ln1_ev = None
        # fresh values every 8th parameter update are enough; the very first call synchronises once
        if self._ln1_ratio is not None and (self._ln1_ev is not None or self._ln1_calls % 8 != 1):
            return
        lo, stride, n, rel, span = self._guard_view
        H = self.enc.heads
        base = self.fp.flat[lo:]
        m = base.as_strided((n, span), (stride, 1), base.storage_offset())

        def t(name, *shape):
            k = int(np.prod(shape))
            return m[:, rel[name]:rel[name] + k].reshape((n,) + shape)
        g1, b1_, g2, b2_ = t("ln1_g", 96).abs(), t("ln1_b", 96).abs(), t("ln2_g", 96).abs(), t("ln2_b", 96).abs()
        ratio = (b1_ / g1).nan_to_num(nan=float("inf")).max()
        ln1_max = g1.amax(1) * (96 ** 0.5) + b1_.amax(1)
        ln2_max = g2.amax(1) * (96 ** 0.5) + b2_.amax(1)
        qkv = t("wqkv", 3 * H * DH, 96).abs().sum(-1).amax(1) * ln1_max
        hid = t("w1", MLP, 96).abs().sum(-1).amax(1) * ln2_max + t("b1", MLP).abs().amax(1)
        bound = torch.maximum(qkv, hid).nan_to_num(nan=float("inf")).max()
        self._ln1_host.copy_(torch.stack((ratio, bound)), non_blocking=True)
        self._ln1_ev = torch.cuda.Event()
        self._ln1_ev.record()
        if self._ln1_ratio is None:
            self._ln1_ev.synchronize()
            self._ln1_ratio, self._half_bound = float(self._ln1_host[0]), float(self._ln1_host[1])
            self._ln1_ev = None

    def ln1_xn_ok(self):
        r = getattr(self, "_ln1_ratio", None)
        return r is not None and r <= self.LN1_XN_MAX_RATIO

    def half_ok(self):
        """may the forward's GEMM operands be IEEE half?  (every one of them provably below HALF_MAX_BOUND; unknown -> no)"""
        b = getattr(self, "_half_bound", None)
        return b is not None and b <= self.HALF_MAX_BOUND

    # ------------------------------------------------------------------ forward pieces
    def tokenize(self, img, mask_u8=None, with_pos=True, emb_drop=(0.0, 0)):
        """img [B, C, H, W] fp32 cuda -> tokens [B, T, 96] (pos added, mask token substituted)"""
        self._require_cuda(img)
        self.ensure()
        B = img.shape[0]
        S, N, P = self.S, self.N, self.P
        T = S * N
        img = img.contiguous().float()
        out = torch.empty(B, T, D, dtype=torch.float32, device=img.device)
        if mask_u8 is None:
            if self._zero_mask is None or self._zero_mask.numel() < B * T:
                self._zero_mask = torch.zeros(B * T, dtype=torch.uint8, device=img.device)
            mask_u8 = self._zero_mask
        fp = self.fp
        if not with_pos:
            if getattr(self, "_zero_pos", None) is None or self._zero_pos.numel() < T * D:
                self._zero_pos = torch.zeros(T * D, dtype=torch.float32, device=img.device)
            pos_a, pos_b, split = self._zero_pos.data_ptr(), 0, 0
        elif self.enc.spectral_pos_embed:
            split = self.enc.pos_embed.shape[-1]
            pos_a, pos_b = fp.ptr("pos_embed"), fp.ptr("channel_embed")
        else:
            split = 0
            pos_a, pos_b = fp.ptr("pos_embedding"), 0
        mt = fp.ptr("mask_token") if self.mim is not None else fp.ptr("post_b")
        V = ctypes.c_void_p
        _lib.check(self.lib.msst_tokenize_fwd(
            _p(img), V(fp.ptr("pre_g")), V(fp.ptr("pre_b")), V(fp.ptr("embed.w.0")), V(fp.ptr("embed.b.0")),
            V(fp.ptr("post_g")), V(fp.ptr("post_b")), V(pos_a), V(pos_b), split, V(mt), _p(mask_u8), _p(out),
            B, S, N, P, emb_drop[0], emb_drop[1], _stream()), "msst_tokenize_fwd")
        return out

    def blocks_fwd(self, x0, save=True, drop=(0.0, 0)):
        """run the 2*depth fused blocks; returns (list of activations [x0 .. x_2L], list of x1)"""
        H = self.enc.heads
        acts = [x0]
        x1s = []
        flags = _kernel_flags()
        # bf16 x1 rows (MSST_X1_BF16): only the role-split forward writes them -- bf16, 8 heads, no kernel-selection flags;
        # MSST_X1_BF16=0 keeps fp32 rows.  The x1 tensor's dtype tells the backward which kind it holds.
        x1_bf16 = (save and self.prec == PREC_BF16 and H == 8 and flags == 0 and os.environ.get("MSST_X1_BF16", "1") != "0")
        want_lse = save and self.prec == PREC_BF16 and H == 8 and flags == 0 and os.environ.get("MSST_LSE", "1") != "0"
        # MSST_FWD_HALF (round 6): the role-split forward multiplies IEEE-half operands (11 significant bits, same MFMA rate) instead of
        # bf16 ones -- the bf16 forward's loss error against the fp32 reference is owned by the rounding of the weights
        # (tools/bf16_error_table.py: 2.7e-4 -> 7e-6 on the Houston-shape anchor).  MSST_FWD_HALF=0: bf16 operands.
        # (decided per launch by _half_flag)
        layers = self._layers()
        # Round 5: a whole stack (its blocks never mix tiles) as ONE launch of the role-split forward -- msst_block_fwd_stack; same
        # arithmetic, bit-identical outputs, no prologue + pipeline fill / drain per block.  MSST_FWD_STACK=0: one launch per block.
        # Measured (tools/fwd_ab.py): ahead when a workgroup holds few tiles (batch 64: -5.5 % forward time, Houston shape: -7.3 %),
        # behind when it holds many (batch 256, EnMAP shape: +1.3 %) -- MSST_FWD_STACK=1 / 0 force it on / off, otherwise by tiles per workgroup.
        want_stack = os.environ.get("MSST_FWD_STACK", "auto")
        stacked = self.prec == PREC_BF16 and H == 8 and flags == 0 and want_stack != "0"
        i0 = 0
        self.fwd_launch_blocks = []   # blocks carried by each block-forward launch of this call (bench.py normalises per-launch numbers with it)
        while i0 < len(layers):
            i1 = i0
            while i1 < len(layers) and layers[i1][0] == layers[i0][0] and i1 - i0 < 16:
                i1 += 1
            use = stacked and (want_stack == "1" or self._tiles_per_workgroup(layers[i0][0], x0.shape[0]) <= STACK_MAX_TILES)
            if use and self._fwd_stack(acts, x1s, i0, i1, save, drop, x1_bf16, want_lse):
                self.fwd_launch_blocks.append(i1 - i0)
            else:
                for i in range(i0, i1):
                    self._fwd_block(acts, x1s, i, save, drop, x1_bf16, want_lse, flags)
                self.fwd_launch_blocks += [1] * (i1 - i0)
            i0 = i1
        return acts, x1s

    def _tiles_per_workgroup(self, sname, B):
        """64-row tiles the busiest workgroup of a block forward walks (the library's own tiling: msst_block_lse_floats counts tiles x heads x 64)"""
        mode = MODE_SPATIAL if sname == "spatial" else MODE_SPECTRAL
        key = (mode, B)
        hit = self._tpw.get(key)
        if hit is None:
            tiles = int(self.lib.msst_block_tiles(mode, B, self.S, self.N))
            grid = max(1, min(tiles, self._cu_count(), self.max_grid if self.max_grid > 0 else tiles))
            hit = self._tpw[key] = -(-tiles // grid)
        return hit

    def _half_flag(self, flags):
        """MSST_FWD_HALF for a forward launch: the role-split kernel (bf16 mode, 8 heads, no kernel selection flags), unless MSST_FWD_HALF=0"""
        self.fwd_half = (self.prec == PREC_BF16 and self.enc.heads == 8 and flags == 0 and os.environ.get("MSST_FWD_HALF", "1") != "0"
                         and self.half_ok())
        return _lib.FWD_HALF if self.fwd_half else 0

    def _fwd_block(self, acts, x1s, i, save, drop, x1_bf16, want_lse, flags):
        """block i as its own launch (msst_block_fwd): appends its output to acts, its saved mid residual to x1s"""
        x = acts[-1]
        B = x.shape[0]
        S, N, H = self.S, self.N, self.enc.heads
        sname, l = self._layers()[i]
        y = torch.empty_like(x)
        x1 = (torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if x1_bf16 else torch.empty_like(x)) if save else None
        # bf16: the block also saves LN1(x) as it used it (bf16 rows), if the selected kernel can; the attention backward
        # then skips its own LN1.  The buffer rides on the x1 tensor object so that every caller keeps its (acts, x1s) pair.
        xn = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if (save and self.prec != PREC_F32) else None
        mode = MODE_SPATIAL if sname == "spatial" else MODE_SPECTRAL
        # ... and (role-split kernel) the softmax statistics of every (tile, head, row): MSST_LSE=0 keeps the backward's own softmax
        lse = None
        if xn is not None and want_lse:
            lse = torch.empty(int(self.lib.msst_block_lse_floats(mode, B, S, N, H)), dtype=torch.float32, device=x.device)
        wrote = ctypes.c_int(0)
        _lib.check(self.lib.msst_block_fwd(ctypes.byref(self._bw[i]), _p(x), _p(y), _p(x1), mode, B, S, N, H,
                                           self.prec | flags | (_lib.X1_BF16 if x1_bf16 else 0) | self._half_flag(flags), self.max_grid, drop[0], drop[1], i, _p(xn), _p(lse),
                                           ctypes.byref(wrote), _stream()),
                   "msst_block_fwd")
        if x1 is not None:
            x1._msst_xn = xn if (wrote.value & _lib.SAVED_XN) else None
            x1._msst_lse = lse if (wrote.value & _lib.SAVED_LSE) else None
            x1._msst_rstd = bool(wrote.value & _lib.SAVED_RSTD) and x1._msst_lse is not None   # rstd of LN1 rides in the tail of the statistics buffer
            x1._msst_half = bool(self.fwd_half)   # the statistics are those of half-operand scores: the backward renormalises (MSST_LSE_RENORM)
        acts.append(y)
        x1s.append(x1)

    def _fwd_stack(self, acts, x1s, i0, i1, save, drop, x1_bf16, want_lse):
        """blocks i0 .. i1 - 1 (one stack) through msst_block_fwd_stack, one launch; False when the library refuses the call (too many
        (tile, block) steps per workgroup for its step table, or per-block operands that are not a constant stride apart): the caller
        then launches block by block"""
        x0 = acts[-1]
        B = x0.shape[0]
        S, N, H = self.S, self.N, self.enc.heads
        dev = x0.device
        n = i1 - i0
        mode = MODE_SPATIAL if self._layers()[i0][0] == "spatial" else MODE_SPECTRAL
        # one allocation per kind, block j its j-th slice: the kernel addresses block j's operands as block 0's + j x a byte stride
        # (the weight copies and the flat parameters are laid out that way by _build_weight_storage / FlatParams)
        def slices(dtype, shape=None):
            t = torch.empty((n,) + tuple(shape if shape is not None else x0.shape), dtype=dtype, device=dev)
            return [t[j] for j in range(n)]
        ys = slices(x0.dtype)
        x1 = slices(torch.bfloat16 if x1_bf16 else torch.float32) if save else None
        xn = slices(torch.bfloat16) if save else None
        lse = slices(torch.float32, (int(self.lib.msst_block_lse_floats(mode, B, S, N, H)),)) if (save and want_lse) else None
        wv = (ctypes.POINTER(MsstBlockWeights) * n)(*[ctypes.pointer(self._bw[i0 + j]) for j in range(n)])
        VP = ctypes.c_void_p * n

        def arr(ts):
            return VP(*[t.data_ptr() for t in ts]) if ts is not None else None
        wrote = ctypes.c_int(0)
        rc = self.lib.msst_block_fwd_stack(wv, n, _p(x0), arr(ys), arr(x1), arr(xn), arr(lse), mode, B, S, N, H,
                                           self.prec | (_lib.X1_BF16 if x1_bf16 else 0) | self._half_flag(0), self.max_grid, drop[0], drop[1], i0,
                                           ctypes.byref(wrote), _stream())
        if rc == -2:   # MSST_ERR_UNSUPPORTED: nothing was launched
            return False
        _lib.check(rc, "msst_block_fwd_stack")
        for j in range(n):
            acts.append(ys[j])
            if save:
                x1[j]._msst_xn = xn[j] if (wrote.value & _lib.SAVED_XN) else None
                x1[j]._msst_lse = lse[j] if (lse is not None and (wrote.value & _lib.SAVED_LSE)) else None
                x1[j]._msst_rstd = bool(wrote.value & _lib.SAVED_RSTD) and x1[j]._msst_lse is not None
                x1[j]._msst_half = bool(self.fwd_half)
                x1s.append(x1[j])
            else:
                x1s.append(None)
        return True

    def head_fwd(self, y, img, idx32, want_pred=False):
        B, T, _ = y.shape
        S, N, P = self.S, self.N, self.P
        K = idx32.shape[1]
        dev = y.device
        dpred = torch.empty(B, K, P, dtype=torch.float32, device=dev)
        pred = torch.empty(B, K, P, dtype=torch.float32, device=dev) if want_pred else None
        partial = torch.empty(B * ((K + 63) // 64), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        per_block = 1 if hasattr(self.mim.to_pixels, "layers") else 0
        V = ctypes.c_void_p
        _lib.check(self.lib.msst_head_fwd(
            _p(y), _p(img), _p(idx32), V(self.fp.ptr("to_pixels.w.0")), V(self.fp.ptr("to_pixels.b.0")), per_block,
            _p(dpred), _p(pred), _p(partial), _p(loss), B, S, N, P, K, _stream()), "msst_head_fwd")
        return loss, dpred, pred

    # ------------------------------------------------------------------ backward pieces
    def _fire(self, bucket):
        if self.bucket_hook is not None:
            for name, start, end in self.fp.buckets:
                if name == bucket:
                    self.bucket_hook(name, start, end)

    def head_bwd(self, y, dpred, csr_ptr, csr_pos, gout=None):
        """-> dy [B, T, 96]; to_pixels grads land in the flat grad buffer"""
        B, T, _ = y.shape
        S, N, P = self.S, self.N, self.P
        K = dpred.shape[1]
        dev = y.device
        nchunk = max(1, min(B, int(os.environ.get("MSST_HEAD_CHUNKS", "0")) or 512 // max(S, 1)))
        dy = torch.empty_like(y)
        slab = torch.empty(S * nchunk * (P * 96 + P), dtype=torch.float32, device=dev)
        per_block = 1 if hasattr(self.mim.to_pixels, "layers") else 0
        gscale = 1.0 / (B * K * P) / K
        V = ctypes.c_void_p
        g = self.fp.grad
        _lib.check(self.lib.msst_head_bwd(
            _p(y), _p(dpred), _p(csr_ptr), _p(csr_pos), V(self.fp.ptr("to_pixels.w.0")), per_block, gscale, _p(gout),
            _p(dy), _p(slab), nchunk, V(self.fp.ptr("to_pixels.w.0", g)), V(self.fp.ptr("to_pixels.b.0", g)),
            B, S, N, P, K, _stream()), "msst_head_bwd")
        self._fire("head")
        return dy

    def _grad_stride(self, nlayers):
        """floats between the gradient tensors of consecutive backward calls (block i -> block i - 1), or None when the blocks'
        gradients are not laid out a constant, 16-byte-aligned stride apart (msst_block_bwd_reduce needs that)"""
        if nlayers < 2:
            return None
        fields = [n for n, _ in MsstBlockGrads._fields_]
        d0 = None
        for i in range(nlayers - 1, 0, -1):
            for f in fields:
                a, b = getattr(self._bg[i], f), getattr(self._bg[i - 1], f)
                if a is None or b is None:
                    return None
                d = b - a
                if d0 is None:
                    d0 = d
                if d != d0:
                    return None
        if d0 is None or d0 <= 0 or d0 % 16:
            return None
        return d0 // 4

    def blocks_bwd(self, acts, x1s, dy, drop=(0.0, 0)):
        """backward through the 2*depth blocks (reverse order); returns dx0"""
        B = dy.shape[0]
        S, N, H = self.S, self.N, self.enc.heads
        dev = dy.device
        ntok = B * S * N
        esz = 4 if self.prec == PREC_F32 else 2
        dx1 = torch.empty(ntok * 96, dtype=torch.float32, device=dev)
        part = torch.empty(H * ntok * 96 * esz, dtype=torch.uint8, device=dev)
        dab = torch.empty(ntok * 96, dtype=torch.bfloat16, device=dev) if self.prec != PREC_F32 else None
        # the bf16 MLP backward runs two workgroups per CU: up to 2 * grid_rows MLP slabs (msst_block_bwd lays the parts out);
        # the chained form adds one MLP slab per workgroup of the fused LN1 + MLP launch
        nslab = self.grid_rows * (3 * MLP_SLAB + LN1_SLAB) + self.attn_chunks * H * ATTN_SLAB
        slab = torch.empty(nslab, dtype=torch.float32, device=dev)
        layers = self._layers()
        flags = _kernel_flags()
        xns = [getattr(t, "_msst_xn", None) for t in x1s]
        lses = [getattr(t, "_msst_lse", None) for t in x1s]
        x1_bf16 = len(x1s) > 0 and all(t.dtype == torch.bfloat16 for t in x1s)
        if not x1_bf16 and any(t.dtype != torch.float32 for t in x1s):
            raise ValueError("saved x1 rows of mixed dtypes")
        x1flag = _lib.X1_BF16 if x1_bf16 else 0
        # Chained backward (msst_block_bwd_chain): the LN1 backward of block i and the MLP-half backward of block i - 1 are
        # one launch, dx of block i stays on chip.  bf16 tuned kernels with saved LN1 rows only; at most four d(LN1 out)
        # partials (one per head pair for an even head count, else one per head).
        nparts = H // 2 if H % 2 == 0 else H
        chain = (self.prec == PREC_BF16 and flags == 0 and dab is not None and all(t is not None for t in xns)
                 and nparts <= 4 and os.environ.get("MSST_BWD_CHAIN", "1") != "0" and len(layers) > 0
                 and ntok * 384 < 2 ** 31 - 16 and nparts * ntok * 192 < 2 ** 31 - 16)   # 32-bit buffer offsets in the fused launch
        # MSST_LN1_FROM_XN (round 6): the fused LN1 + MLP launch rebuilds xhat of LN1 from the saved bf16 LN1 rows and the saved rstd
        # instead of re-reading the fp32 block input (192 of 2304 bytes per token less) -- when the forward saved both for every block,
        # and the LN1 parameters allow the division by gamma (ln1_xn_ok: max |beta / gamma| <= 12, checked on the device a step behind)
        xnflag = 0
        if (chain and x1_bf16 and os.environ.get("MSST_LN1_XN", "1") != "0" and all(getattr(t, "_msst_rstd", False) for t in x1s)
                and self.ln1_xn_ok()):
            xnflag = _lib.LN1_FROM_XN
        self.last_bwd_ln1_from_xn = bool(xnflag)
        if chain:
            # dynamic tile queues (attach_data_parallel sets self.tile_queue; MSST_TILE_QUEUE=1 forces them): see include/msst.h
            queue = None
            if self.tile_queue or os.environ.get("MSST_TILE_QUEUE", "0") == "1":
                if getattr(self, "_queue_ws", None) is None or self._queue_ws.device != dev:
                    self._queue_ws = torch.zeros(64, dtype=torch.int32, device=dev)
                queue = self._queue_ws
            last = len(layers) - 1
            dx0 = torch.empty_like(dy)
            null_w = ctypes.POINTER(MsstBlockWeights)()
            null_g = ctypes.POINTER(MsstBlockGrads)()
            # Deferred slab reduction (msst_block_bwd_reduce, opt-in with MSST_BWD_DEFER=1): every call of a run of same-mode blocks
            # keeps its own slab set and ONE launch reduces the run -- 2 block reductions per step instead of 2 * depth.  Bit-identical
            # gradients; measured (LABNOTES round 4): the reductions drop from 566 to 400 us per EnMAP step, but the producers pay it
            # back -- their slab epilogues now write 1.5 GB of cold memory per step instead of the same 62 MB that the reduction just
            # read (attention backward +6 us per launch) -- so the per-call reduction stays the default.
            gstride = self._grad_stride(len(layers))
            defer = gstride is not None and os.environ.get("MSST_BWD_DEFER", "0") == "1"
            nslab_r = (nslab + 3) // 4 * 4
            if defer:
                slab = torch.empty(nslab_r * len(layers), dtype=torch.float32, device=dev)
            run_start = last
            for i in reversed(range(len(layers))):
                sname, l = layers[i]
                mode = MODE_SPATIAL if sname == "spatial" else MODE_SPECTRAL
                prev = i > 0
                y = last - i
                slab_i = slab[y * nslab_r:] if defer else slab
                _lib.check(self.lib.msst_block_bwd_chain(
                    ctypes.byref(self._bw[i]), ctypes.byref(self._bg[i]),
                    ctypes.byref(self._bw[i - 1]) if prev else null_w, ctypes.byref(self._bg[i - 1]) if prev else null_g,
                    _p(acts[i]), _p(x1s[i]), _p(x1s[i - 1]) if prev else _p(None), _p(dy) if i == last else _p(None),
                    _p(None) if prev else _p(dx0), _p(dx1), _p(part), _p(slab_i), self.grid_rows, self.attn_chunks, mode,
                    B, S, N, H, self.prec | x1flag | xnflag | (_lib.LSE_RENORM if getattr(x1s[i], "_msst_half", False) else 0) |
                    (_lib.BWD_DEFER_REDUCE if defer else 0), drop[0], drop[1], i, _p(xns[i]), _p(lses[i]), _p(dab),
                    1 if i == last else 0, _p(queue), _stream()),
                    "msst_block_bwd_chain")
                if not defer:
                    self._fire(f"{sname}.{l}")
                elif i == 0 or layers[i - 1][0] != sname:   # the run of this stack ends here: reduce it, then announce its blocks
                    count = run_start - i + 1
                    _lib.check(self.lib.msst_block_bwd_reduce(
                        ctypes.byref(self._bg[run_start]), ctypes.byref(self._bg[run_start - 1]) if run_start > 0 else null_g,
                        _p(slab[(last - run_start) * nslab_r:]), nslab_r, gstride, count, count if i > 0 else count - 1,
                        1 if run_start == last else 0, self.grid_rows, self.attn_chunks, mode, B, S, N, H, self.prec, _stream()),
                        "msst_block_bwd_reduce")
                    for j in range(run_start, i - 1, -1):
                        self._fire(f"{layers[j][0]}.{layers[j][1]}")
                    run_start = i - 1
            return dx0
        g = dy
        other = torch.empty_like(dy)
        ws = (dx1, part, slab, dab)
        for i in reversed(range(len(layers))):
            self.block_bwd_single(i, acts[i], x1s[i], g, other, drop=drop, ws=ws)
            g, other = other, g
            self._fire(f"{layers[i][0]}.{layers[i][1]}")
        return g

    def block_bwd_single(self, i, x, x1, dy, dx=None, drop=(0.0, 0), ws=None):
        """Backward of block i alone (msst_block_bwd: MLP half, attention half, LN1 backward, slab reduction): x = the block's input,
        x1 = its saved mid residual (with the LN1 rows / statistics its forward attached), dy = the gradient at its output -> dx;
        the block's parameter gradients land in the flat gradient buffer.  The unchained loop of blocks_bwd runs on it; the parity tests
        call it block by block with the ORACLE's activations and gradients (no error carried from block to block)."""
        B = dy.shape[0]
        S, N, H = self.S, self.N, self.enc.heads
        dev = dy.device
        ntok = B * S * N
        if ws is None:
            esz = 4 if self.prec == PREC_F32 else 2
            ws = (torch.empty(ntok * 96, dtype=torch.float32, device=dev), torch.empty(H * ntok * 96 * esz, dtype=torch.uint8, device=dev),
                  torch.empty(self.grid_rows * (3 * MLP_SLAB + LN1_SLAB) + self.attn_chunks * H * ATTN_SLAB, dtype=torch.float32, device=dev),
                  torch.empty(ntok * 96, dtype=torch.bfloat16, device=dev) if self.prec != PREC_F32 else None)
        dx1, part, slab, dab = ws
        if dx is None:
            dx = torch.empty_like(dy)
        sname, l = self._layers()[i]
        mode = MODE_SPATIAL if sname == "spatial" else MODE_SPECTRAL
        if x1.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("saved x1 rows must be fp32 or bf16")
        x1flag = _lib.X1_BF16 if x1.dtype == torch.bfloat16 else 0
        _lib.check(self.lib.msst_block_bwd(
            ctypes.byref(self._bw[i]), ctypes.byref(self._bg[i]), _p(x), _p(x1), _p(dy), _p(dx),
            _p(dx1), _p(part), _p(slab), self.grid_rows, self.attn_chunks, mode, B, S, N, H,
            self.prec | _kernel_flags() | x1flag | (_lib.LSE_RENORM if getattr(x1, "_msst_half", False) else 0),
            drop[0], drop[1], i, _p(getattr(x1, "_msst_xn", None)), _p(getattr(x1, "_msst_lse", None)), _p(dab), _stream()), "msst_block_bwd")
        return dx

    def tokenize_bwd(self, img, mask_u8, dx0, emb_drop=(0.0, 0), with_pos=True):
        """with_pos=False: gradients of the embedding / its two LayerNorms only (the position table and the mask
        token were applied outside the kernel, by the caller's own autograd ops)"""
        B = img.shape[0]
        S, N, P = self.S, self.N, self.P
        dev = img.device
        nchunk = max(1, min(B, self.tok_chunks))
        ss = N * 96 + 96 * P + 96 * 4 + 32
        slab = torch.empty(S * nchunk * ss + S * N * 96, dtype=torch.float32, device=dev)
        fp, g = self.fp, self.fp.grad
        V = ctypes.c_void_p
        if not with_pos:
            split, dpa, dpb = 0, 0, 0
        elif self.enc.spectral_pos_embed:
            split = self.enc.pos_embed.shape[-1]
            dpa, dpb = fp.ptr("pos_embed", g), fp.ptr("channel_embed", g)
        else:
            split = 0
            dpa, dpb = fp.ptr("pos_embedding", g), 0
        dmt = fp.ptr("mask_token", g) if (self.mim is not None and with_pos) else 0
        _lib.check(self.lib.msst_tokenize_bwd(
            _p(img), V(fp.ptr("pre_g")), V(fp.ptr("pre_b")), V(fp.ptr("embed.w.0")), V(fp.ptr("embed.b.0")),
            V(fp.ptr("post_g")), V(fp.ptr("post_b")), _p(mask_u8), _p(dx0), _p(slab), nchunk,
            V(fp.ptr("pre_g", g)), V(fp.ptr("pre_b", g)), V(fp.ptr("embed.w.0", g)), V(fp.ptr("embed.b.0", g)),
            V(fp.ptr("post_g", g)), V(fp.ptr("post_b", g)), V(dpa), V(dpb), split, V(dmt), B, S, N, P,
            emb_drop[0], emb_drop[1], _stream()), "msst_tokenize_bwd")
        if with_pos:
            self._fire("tokenizer")

    # ------------------------------------------------------------------ autograd entry (SimMIM loss)
    def trainable(self):
        """[(flat name, parameter)] of everything that receives a gradient in pre-training"""
        if self.fp.flat is None or getattr(self, "_trainable_key", None) != self.fp.version:
            groups, _ = self.fp._ordered()
            self._trainable = [(n, p) for _, g in groups for n, p in g]
            self._trainable_key = self.fp.version if self.fp.flat is not None else None
        return self._trainable

    def dropout_state(self):
        """(p, seed) for this forward: p = transformer dropout when the encoder is in training mode (the
        reference's nn.Dropout sites, vit_spatial_spectral.py:38,40,57,62), else 0.  A fresh 32-bit seed is
        drawn from the torch CPU generator per forward (reproducible under torch.manual_seed); the masks are a
        stateless function of (seed, layer, site, element) that the backward regenerates."""
        p = float(self.enc.dropout_p) if self.enc.training else 0.0
        if p <= 0.0:
            return 0.0, 0
        if not 0.0 < p < 1.0:
            raise ValueError(f"dropout probability {p} outside (0, 1)")
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        # data parallel: every rank draws the same seed from identically seeded generators; mix the rank in so that
        # the ranks' dropout masks are independent (as they are for the reference's per-process RNG streams)
        rank = int(getattr(self.mim, "dp_rank", 0)) if self.mim is not None else 0
        if rank:
            seed = (seed ^ (rank * 0x9E3779B1)) & 0x7FFFFFFF
        return p, seed

    def _upload(self, dev, *arrays):
        """Host arrays of one step -> device, without stalling the launch queue.

        A ``.to(device)`` from pageable memory is stream ordered AND host blocking: the host sat until the
        previous step had drained, and the GPU then idled while the next step's launches were issued (measured:
        2 ms per 41 ms step).  Going through pinned staging buffers keeps the copies asynchronous: they queue on
        the compute stream behind the previous step and the host keeps launching.  Two pinned sets alternate; a
        set is rewritten only after the copy that last read it has run (host waits on its event, i.e. the host
        runs at most two steps ahead).  The DEVICE tensors are fresh per call (caching allocator, stream ordered):
        the autograd path stashes them for its backward, and any number of further forwards (eval / no_grad ones
        included) may run between a forward and its backward without touching them."""
        if getattr(self, "_up", None) is None:
            self._up = {"sets": [None, None], "k": 0}
        up = self._up
        k = up["k"]
        up["k"] = 1 - k
        cur = up["sets"][k]
        sig = tuple((a.shape, a.dtype) for a in arrays)
        if cur is None or cur["sig"] != sig:
            cur = {"sig": sig,
                   "pin": [torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True) for a in arrays],
                   "ev": torch.cuda.Event()}
            up["sets"][k] = cur
        else:
            cur["ev"].synchronize()
        out = []
        for pin, a in zip(cur["pin"], arrays):
            pin.numpy()[...] = a
            d = torch.empty(pin.shape, dtype=pin.dtype, device=dev)
            d.copy_(pin, non_blocking=True)
            out.append(d)
        cur["ev"].record(torch.cuda.current_stream(dev))
        return out

    def simmim_loss(self, img, bool_mask, idx):
        """scalar loss attached to autograd (reference SimMIMSpatialSpectral.forward :203-340)"""
        self._require_cuda(img)
        self.ensure()
        dev = img.device
        T = self.S * self.N
        img = img.contiguous().float()
        from .masking import inverse_csr
        bm = bool_mask.cpu().numpy() if torch.is_tensor(bool_mask) else np.asarray(bool_mask)
        ix = idx.cpu().numpy() if torch.is_tensor(idx) else np.asarray(idx)
        ptr, pos = inverse_csr(ix, T)
        mask_u8, idx32, csr_ptr, csr_pos = self._upload(dev, bm.astype(np.uint8), ix.astype(np.int32), ptr, pos)
        names = [n for n, _ in self.trainable()]
        params = [p for _, p in self.trainable()]
        if not torch.is_grad_enabled() or not any(p.requires_grad for p in params):
            self.prep_weights()
            x0 = self.tokenize(img, mask_u8)
            acts, _ = self.blocks_fwd(x0, save=False, drop=self.dropout_state())
            loss, _, _ = self.head_fwd(acts[-1], img, idx32)
            return loss
        return _SimMIMLossFn.apply(self, names, self.dropout_state(), img, mask_u8, idx32, csr_ptr, csr_pos, *params)

    # ------------------------------------------------------------------ encoder-level API (inference)
    def _block_param_names(self):
        return [n for n, _ in self.trainable() if n.startswith(("spatial.", "spectral."))]

    def _embed_param_names(self):
        return [n for n, _ in self.trainable() if n.startswith(("embed.", "pre_", "post_"))]

    def transformer(self, tokens):
        """ViTSpatialSpectral.transformer_forward (reference :495-499): both stacks on [B, T, 96] tokens.  Attached to
        autograd when gradients are enabled, so a caller that keeps the reference's own SimMIMSpatialSpectral
        (vit_simmim_original.py:298) and swaps only the encoder trains through the HIP blocks: d(tokens) flows on to
        whatever produced them, the block parameters receive views of the flat gradient buffer."""
        self._require_cuda(tokens)
        self.ensure()
        drop = self.dropout_state()
        tokens = tokens.contiguous().float()
        by_name = dict(self.trainable())
        names = self._block_param_names()
        params = [by_name[n] for n in names]
        if torch.is_grad_enabled() and (tokens.requires_grad or any(p.requires_grad for p in params)):
            return _TransformerFn.apply(self, names, drop, tokens, *params)
        self.prep_weights()
        acts, _ = self.blocks_fwd(tokens, save=False, drop=drop)
        return acts[-1]

    def embed_patches(self, patches):
        """BlockwisePatchEmbedding.embed on patches [B, S, N, P] (no position / mask terms; reference :210-222),
        attached to autograd for the embedding parameters when gradients are enabled."""
        self._require_cuda(patches)
        self.ensure()
        B, S, N, P = patches.shape
        img = patches.detach().permute(0, 1, 3, 2).reshape(B, S * P, N).contiguous().float()
        by_name = dict(self.trainable())
        names = self._embed_param_names()
        params = [by_name[n] for n in names]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _EmbedFn.apply(self, names, img, *params)
        return self.tokenize(img, None, with_pos=False)

    def features(self, img):
        """forward_features (reference :518-534): tokenize + pos (+ embedding dropout in training mode) -> transformer.

        Differentiable like the reference's: when gradients are enabled and any encoder parameter requires one, the result is
        composed of the two autograd entry points (``embed_patches`` and ``transformer``) with the position add and the
        embedding dropout as ordinary torch ops in between -- exactly what the reference's own SimMIM wrapper does with this
        encoder -- so a custom head trained on these features trains the encoder too.  Otherwise (eval / no_grad) one fused
        tokenizer launch does embed + position + dropout."""
        enc = self.enc
        params = [q for _, q in self.trainable()]
        if torch.is_grad_enabled() and any(q.requires_grad for q in params):
            self._require_cuda(img)
            patches = enc.to_patch_embedding.to_patch(img.contiguous().float())
            tokens = self.embed_patches(patches).reshape(img.shape[0], -1, D)
            pos = enc.get_pos_embeddings() if enc.spectral_pos_embed else enc.pos_embedding[:, :tokens.shape[1]]
            tokens = tokens + pos
            if enc.training and float(enc.emb_dropout_p) > 0:
                tokens = torch.nn.functional.dropout(tokens, p=float(enc.emb_dropout_p), training=True)
            return self.transformer(tokens)
        pe = float(enc.emb_dropout_p) if enc.training else 0.0
        emb_drop = (pe, int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) if pe > 0 else (0.0, 0)
        x0 = self.tokenize(img, None, with_pos=True, emb_drop=emb_drop)
        with torch.no_grad():
            return self.transformer(x0)

    # ------------------------------------------------------------------ classification (row a17 / finetune.py)
    def cls_head_fwd(self, y):
        B = y.shape[0]
        nc = self.enc.num_classes
        logits = torch.empty(B, nc, self.N, dtype=torch.float32, device=y.device)
        fp = self.fp
        V = ctypes.c_void_p
        _lib.check(self.lib.msst_cls_head_fwd(
            _p(y), V(fp.ptr("mlp_head.0.weight")), V(fp.ptr("mlp_head.0.bias")), V(fp.ptr("mlp_head.1.weight")),
            V(fp.ptr("mlp_head.1.bias")), _p(logits), B, self.S, self.N, nc, _stream()), "msst_cls_head_fwd")
        return logits

    def cls_head_bwd(self, y, dlogits):
        B = y.shape[0]
        nc = self.enc.num_classes
        dy = torch.empty_like(y)
        slab = torch.empty(B * (nc * 97 + 192), dtype=torch.float32, device=y.device)
        fp, g = self.fp, self.fp.grad
        V = ctypes.c_void_p
        _lib.check(self.lib.msst_cls_head_bwd(
            _p(y), _p(dlogits), V(fp.ptr("mlp_head.0.weight")), V(fp.ptr("mlp_head.0.bias")),
            V(fp.ptr("mlp_head.1.weight")), _p(dy), _p(slab), V(fp.ptr("mlp_head.0.weight", g)),
            V(fp.ptr("mlp_head.0.bias", g)), V(fp.ptr("mlp_head.1.weight", g)), V(fp.ptr("mlp_head.1.bias", g)),
            B, self.S, self.N, nc, _stream()), "msst_cls_head_bwd")
        self._fire("cls_head")
        return dy

    def classify(self, img):
        """ViTSpatialSpectral.forward: logits [B, num_classes, H, W] (reference :536-564)."""
        self._require_cuda(img)
        self.ensure()
        img = img.contiguous().float()
        H = W = self.enc.num_spatial_patches_sqrt
        p = float(self.enc.dropout_p) if self.enc.training else 0.0
        pe = float(self.enc.emb_dropout_p) if self.enc.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (p > 0 or pe > 0) else 0
        drop, emb_drop = (p, seed), (pe, seed ^ 0x5bd1e995)
        params = [q for _, q in self.trainable()]
        if self.mim is not None or not torch.is_grad_enabled() or not any(q.requires_grad for q in params):
            if self.mim is not None and torch.is_grad_enabled() and any(q.requires_grad for q in params):
                raise NotImplementedError("train the classifier through a bare ViTSpatialSpectral (as finetune.py "
                                          "does), not through an encoder wrapped in SimMIMSpatialSpectral")
            self.prep_weights()
            x0 = self.tokenize(img, None, emb_drop=emb_drop)
            acts, _ = self.blocks_fwd(x0, save=False, drop=drop)
            return self.cls_head_fwd(acts[-1]).view(img.shape[0], -1, H, W)
        names = [n for n, _ in self.trainable()]
        out = _ClassifyFn.apply(self, names, drop, emb_drop, img, *params)
        return out.view(img.shape[0], -1, H, W)

    # ------------------------------------------------------------------ staged forward (tests / debugging)
    def simmim_forward_stages(self, img, bool_mask, idx, drop=(0.0, 0)):
        """Forward only, returning the intermediates the golden fixtures pin."""
        self._require_cuda(img)
        self.prep_weights()
        dev = img.device
        img = img.contiguous().float()
        mask_u8 = bool_mask.to(device=dev, dtype=torch.uint8).contiguous()
        idx32 = idx.to(device=dev, dtype=torch.int32).contiguous()
        tok_embed = self.tokenize(img, None, with_pos=False)
        x0 = self.tokenize(img, mask_u8)
        acts, x1s = self.blocks_fwd(x0, drop=drop)
        loss, dpred, pred = self.head_fwd(acts[-1], img, idx32, want_pred=True)
        L = self.enc.depth
        return dict(loss=loss, tok_embed=tok_embed, tok_masked=x0, after_spatial=acts[L], enc_out=acts[-1],
                    pred=pred, dpred=dpred, acts=acts, x1s=x1s)


class _SimMIMLossFn(torch.autograd.Function):
    """loss = SimMIM(img); gradients of every parameter come from the HIP backward kernels and are
    handed to autograd as views of the flat gradient buffer."""

    @staticmethod
    def forward(ctx, eng, names, drop, img, mask_u8, idx32, csr_ptr, csr_pos, *params):
        eng.prep_weights()
        x0 = eng.tokenize(img, mask_u8)
        acts, x1s = eng.blocks_fwd(x0, save=True, drop=drop)
        ctx.drop = drop
        loss, dpred, _ = eng.head_fwd(acts[-1], img, idx32)
        ctx.eng = eng
        ctx.names = names
        ctx.stash = (img, mask_u8, csr_ptr, csr_pos, acts, x1s, dpred)
        return loss

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        img, mask_u8, csr_ptr, csr_pos, acts, x1s, dpred = ctx.stash
        ctx.stash = None
        lo, hi = eng.fp.grad.data_ptr(), eng.fp.grad.data_ptr() + 4 * eng.fp.grad.numel()
        for _, p in eng.trainable():
            if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                raise RuntimeError(
                    "maskedsst_amd hands autograd views of its flat gradient buffer: drop the previous gradients "
                    "with optimizer.zero_grad(set_to_none=True) (the torch default) before the next backward; "
                    "in-place gradient accumulation across backward calls is not supported")
        gout = gout.contiguous().float()
        dy = eng.head_bwd(acts[-1], dpred, csr_ptr, csr_pos, gout)
        dx0 = eng.blocks_bwd(acts, x1s, dy, drop=ctx.drop)
        eng.tokenize_bwd(img, mask_u8, dx0)
        grads = tuple(eng.fp.view(n, eng.fp.grad) for n in ctx.names)
        return (None,) * 8 + grads


def _refuse_accumulation(eng, names):
    by_name = dict(eng.trainable())
    lo, hi = eng.fp.grad.data_ptr(), eng.fp.grad.data_ptr() + 4 * eng.fp.grad.numel()
    for n in names:
        p = by_name[n]
        if p.grad is not None and lo <= p.grad.data_ptr() < hi:
            raise RuntimeError("call optimizer.zero_grad(set_to_none=True) before the next backward "
                               "(maskedsst_amd hands autograd views of its flat gradient buffer)")


class _TransformerFn(torch.autograd.Function):
    """y = transformer_forward(tokens): the 2 * depth fused blocks with their HIP backward."""

    @staticmethod
    def forward(ctx, eng, names, drop, tokens, *params):
        eng.prep_weights()
        acts, x1s = eng.blocks_fwd(tokens, save=True, drop=drop)
        ctx.eng, ctx.names, ctx.drop = eng, names, drop
        ctx.stash = (acts[:-1], x1s)   # block inputs only: the output itself is not needed by the backward
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.eng
        acts, x1s = ctx.stash
        ctx.stash = None
        _refuse_accumulation(eng, ctx.names)
        dx0 = eng.blocks_bwd(acts, x1s, dy.contiguous().float().clone(), drop=ctx.drop)
        grads = tuple(eng.fp.view(n, eng.fp.grad) for n in ctx.names)
        return (None, None, None, dx0) + grads


class _EmbedFn(torch.autograd.Function):
    """tokens = BlockwisePatchEmbedding.embed(patches) (no position / mask terms) with its HIP backward."""

    @staticmethod
    def forward(ctx, eng, names, img, *params):
        ctx.eng, ctx.names = eng, names
        ctx.stash = (img,)
        return eng.tokenize(img, None, with_pos=False)

    @staticmethod
    def backward(ctx, dtok):
        eng = ctx.eng
        (img,) = ctx.stash
        ctx.stash = None
        _refuse_accumulation(eng, ctx.names)
        n = img.shape[0] * eng.S * eng.N
        if eng._zero_mask is None or eng._zero_mask.numel() < n:
            eng._zero_mask = torch.zeros(n, dtype=torch.uint8, device=img.device)
        eng.tokenize_bwd(img, eng._zero_mask, dtok.contiguous().float(), with_pos=False)
        grads = tuple(eng.fp.view(n_, eng.fp.grad) for n_ in ctx.names)
        return (None, None, None) + grads


class _ClassifyFn(torch.autograd.Function):
    """logits = encoder(img) for the classification path; backward through the same HIP kernels."""

    @staticmethod
    def forward(ctx, eng, names, drop, emb_drop, img, *params):
        eng.prep_weights()
        x0 = eng.tokenize(img, None, emb_drop=emb_drop)
        acts, x1s = eng.blocks_fwd(x0, save=True, drop=drop)
        logits = eng.cls_head_fwd(acts[-1])
        ctx.eng, ctx.names, ctx.drop, ctx.emb_drop = eng, names, drop, emb_drop
        ctx.stash = (img, acts, x1s)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        img, acts, x1s = ctx.stash
        ctx.stash = None
        lo, hi = eng.fp.grad.data_ptr(), eng.fp.grad.data_ptr() + 4 * eng.fp.grad.numel()
        for _, p in eng.trainable():
            if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                raise RuntimeError("call optimizer.zero_grad(set_to_none=True) before the next backward "
                                   "(maskedsst_amd hands autograd views of its flat gradient buffer)")
        dy = eng.cls_head_bwd(acts[-1], dlogits.contiguous().float())
        dx0 = eng.blocks_bwd(acts, x1s, dy, drop=ctx.drop)
        if eng._zero_mask is None or eng._zero_mask.numel() < img.shape[0] * eng.S * eng.N:
            eng._zero_mask = torch.zeros(img.shape[0] * eng.S * eng.N, dtype=torch.uint8, device=img.device)
        eng.tokenize_bwd(img, eng._zero_mask, dx0, emb_drop=ctx.emb_drop)
        grads = tuple(eng.fp.view(n, eng.fp.grad) for n in ctx.names)
        return (None,) * 5 + grads
