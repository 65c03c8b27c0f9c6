"""Flat parameter / gradient buffers.

The HIP kernels want the per-spectral-block ``nn.Linear`` weights of the reference
(``blockwise_embed.{i}``, ``to_pixels.layers.{i}``) as packed ``[S, out, in]`` arrays, the fused
AdamW wants ONE buffer, and data-parallel buckets want gradients laid out in the order in which
the backward produces them.  All three are met by making every ``nn.Parameter`` a *view* into one
flat fp32 buffer (the ``state_dict`` keys / shapes of the reference are untouched).

Layout (= backward completion order, so DP buckets are contiguous slices):
    to_pixels | spectral layers L-1..0 | spatial layers L-1..0 | tokenizer (embed, norms, pos,
    mask_token) | mlp_head (no gradient during pre-training: kept last, outside the AdamW/DP range)
A bare encoder (classification) puts mlp_head first instead (it is trained, and done first).
"""
from collections import OrderedDict

import torch


def _layer_params(layer):
    attn, ff = layer[0], layer[1]
    return [
        ("ln1_g", attn.norm.weight), ("ln1_b", attn.norm.bias),
        ("wqkv", attn.fn.to_qkv.weight), ("wout", attn.fn.to_out[0].weight), ("bo", attn.fn.to_out[0].bias),
        ("ln2_g", ff.norm.weight), ("ln2_b", ff.norm.bias),
        ("w1", ff.fn.net[0].weight), ("b1", ff.fn.net[0].bias),
        ("w2", ff.fn.net[3].weight), ("b2", ff.fn.net[3].bias),
    ]


class FlatParams:
    """Owns the flat buffers of one model (a SimMIM wrapper or a bare encoder)."""

    def __init__(self, encoder, mim=None):
        self.encoder = encoder
        self.mim = mim
        self.flat = None
        self.grad = None
        self.segments = OrderedDict()   # name -> (offset, numel, shape)
        self.buckets = []               # [(name, start, end)] in backward-completion order
        self.n_trainable = 0            # prefix length that receives gradients in pre-training
        self._ptrs = None
        self.version = 0                # bumped by every (re)flatten: caches keyed on the layout check it

    # ------------------------------------------------------------------
    def _ordered(self):
        enc, mim = self.encoder, self.mim
        groups = []  # (bucket name, [(name, param)])
        if mim is not None:
            tp = mim.to_pixels
            if hasattr(tp, "layers"):
                ws = [(f"to_pixels.w.{i}", l.weight) for i, l in enumerate(tp.layers)]
                bs = [(f"to_pixels.b.{i}", l.bias) for i, l in enumerate(tp.layers)]
            else:
                ws, bs = [("to_pixels.w.0", tp.weight)], [("to_pixels.b.0", tp.bias)]
            groups.append(("head", ws + bs))
        tr = enc.spatial_spectral_transformer
        stacks = [("spectral", tr[3]), ("spatial", tr[1])] if not enc.spectral_only else [("spectral", tr[1])]
        for sname, stack in stacks:
            for l in reversed(range(len(stack.layers))):
                groups.append((f"{sname}.{l}", [(f"{sname}.{l}.{n}", p) for n, p in _layer_params(stack.layers[l])]))
        pe = enc.to_patch_embedding
        tok = [(f"embed.w.{i}", l.weight) for i, l in enumerate(pe.blockwise_embed)]
        tok += [(f"embed.b.{i}", l.bias) for i, l in enumerate(pe.blockwise_embed)]
        tok += [("pre_g", pe.pre_norm.weight), ("pre_b", pe.pre_norm.bias),
                ("post_g", pe.post_norm.weight), ("post_b", pe.post_norm.bias)]
        if enc.spectral_pos_embed:
            tok += [("pos_embed", enc.pos_embed), ("channel_embed", enc.channel_embed)]
        else:
            tok += [("pos_embedding", enc.pos_embedding)]
        if mim is not None:
            tok += [("mask_token", mim.mask_token)]
        groups.append(("tokenizer", tok))
        head = [(f"mlp_head.{n}", p) for n, p in enc.mlp_head.named_parameters()]
        if mim is None:
            # bare encoder (classification / finetune.py): the head is trained and its gradients are
            # the first to complete in the backward
            return [("cls_head", head)] + groups, []
        return groups, head

    def stale(self):
        if self.flat is None:
            return True
        groups, head = self._ordered()
        ptrs = tuple(p.data_ptr() for _, g in groups for _, p in g) + tuple(p.data_ptr() for _, p in head)
        return ptrs != self._ptrs

    def flatten(self):
        """(Re)build the flat buffer from the current parameter values and re-point every
        parameter at its view.  Called lazily; a ``module.to(device)`` simply triggers a rebuild."""
        groups, head = self._ordered()
        allp = [p for _, g in groups for _, p in g] + [p for _, p in head]
        device = allp[0].device
        total = sum(p.numel() for p in allp)
        # segments are packed back to back (no padding): consecutive per-block Linear weights /
        # biases form the packed [S, out, in] / [S, out] arrays the kernels index directly
        def al(n):
            return n
        size = (total + 3) // 4 * 4
        flat = torch.zeros(size, dtype=torch.float32, device=device)
        grad = torch.zeros(size, dtype=torch.float32, device=device)
        self.segments.clear()
        self.buckets = []
        off = 0
        with torch.no_grad():
            for bname, g in groups + [("mlp_head", head)]:
                start = off
                for name, p in g:
                    n = p.numel()
                    flat[off:off + n].copy_(p.detach().reshape(-1).to(torch.float32))
                    p.data = flat[off:off + n].view(p.shape)
                    self.segments[name] = (off, n, tuple(p.shape))
                    off += al(n)
                if bname != "mlp_head":
                    self.buckets.append((bname, start, off))
                    self.n_trainable = off
        self.flat, self.grad = flat, grad
        self.version += 1
        self._ptrs = tuple(p.data_ptr() for p in allp)
        self.total = total
        return self

    def view(self, name, buf=None):
        off, n, shape = self.segments[name]
        return (self.flat if buf is None else buf)[off:off + n].view(shape)

    def ptr(self, name, buf=None):
        off, _, _ = self.segments[name]
        base = self.flat if buf is None else buf
        return base.data_ptr() + 4 * off

    def grad_views(self, params):
        """gradient views (into ``self.grad``) matching a list of (name) keys"""
        return [self.view(n, self.grad) for n in params]
