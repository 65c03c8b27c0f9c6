"""Functional plain-PyTorch fp32 restatement of the MaskedSST hot path (oracle; test
infrastructure only -- see ``oracle/__init__.py``).

Every function cites the reference lines it follows (paths relative to the reference repo).
Parameters live in an ordered ``dict[str, Tensor]`` keyed exactly like the reference
``SimMIMSpatialSpectral.state_dict()``; gradients come from autograd over this
restated forward.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
import math

import numpy as np
import torch
import torch.nn.functional as F

from .pos_embed import sincos_1d, sincos_2d
from .masking import make_masks


@dataclass
class OracleConfig:
    bands: int = 200
    image_size: int = 8
    spatial_patch: int = 1
    spectral_patch: int = 10
    dim: int = 96
    depth: int = 12
    heads: int = 8
    dim_head: int = 64
    mlp_dim: int = 64
    n_classes: int = 8
    spectral_pos_embed: bool = False
    spectral_pos: list = None
    masking_ratio: float = 0.7
    mask_patch_size: int = 4
    tube_masking: bool = True
    to_pixels_per_spectral_block: bool = True

    @property
    def S(self):  # spectral blocks        vit_spatial_spectral.py:326
        return self.bands // self.spectral_patch

    @property
    def Nsq(self):  # vit_spatial_spectral.py:324
        return self.image_size // self.spatial_patch

    @property
    def N(self):  # spatial tokens          vit_spatial_spectral.py:325
        return self.Nsq ** 2

    @property
    def T(self):  # vit_spatial_spectral.py:328
        return self.S * self.N

    @property
    def P(self):  # pixels per patch        vit_spatial_spectral.py:309-311
        return self.spectral_patch * self.spatial_patch ** 2

    @property
    def K(self):  # num_masked              vit_simmim_original.py:252
        return int(self.masking_ratio * self.T)


def _linear(out_f, in_f, bias=True):
    """Stock nn.Linear init == what the reference draws (same RNG stream under the same seed)."""
    lin = torch.nn.Linear(in_f, out_f, bias=bias)
    return lin.weight.detach().clone(), (lin.bias.detach().clone() if bias else None)


def init_params(cfg: OracleConfig, with_mim=True):
    """Draw parameters in the reference construction order:
    vit_spatial_spectral.py:194-208 (embed), :352-389 (pos), :85-97 x depth x 2 stacks,
    :481-493 (head); vit_simmim_original.py:192-201 (mask token, to_pixels).
    Returns an OrderedDict in ``state_dict()`` order."""
    D, S, N, P = cfg.dim, cfg.S, cfg.N, cfg.P
    inner = cfg.heads * cfg.dim_head
    enc = OrderedDict()
    pe = "encoder.to_patch_embedding."
    enc[pe + "pre_norm.weight"] = torch.ones(P)
    enc[pe + "pre_norm.bias"] = torch.zeros(P)
    enc[pe + "post_norm.weight"] = torch.ones(D)
    enc[pe + "post_norm.bias"] = torch.zeros(D)
    for i in range(S):
        w, b = _linear(D, P)
        enc[pe + f"blockwise_embed.{i}.weight"] = w
        enc[pe + f"blockwise_embed.{i}.bias"] = b
    pos = OrderedDict()
    if cfg.spectral_pos_embed:
        cdim = D // 3
        pdim = D - cdim
        sp = np.arange(S) if cfg.spectral_pos is None else np.asarray(cfg.spectral_pos)
        assert len(sp) == S
        pos["encoder.pos_embed"] = torch.from_numpy(sincos_2d(pdim, cfg.Nsq)).float().unsqueeze(0)
        pos["encoder.channel_embed"] = torch.from_numpy(sincos_1d(cdim, sp)).float().unsqueeze(0)
    else:
        pos["encoder.pos_embedding"] = torch.randn(1, cfg.T + 1, D)
    tr = OrderedDict()
    for stack in (1, 3):
        for l in range(cfg.depth):
            pre = f"encoder.spatial_spectral_transformer.{stack}.layers.{l}."
            wqkv, _ = _linear(3 * inner, D, bias=False)
            wo, bo = _linear(D, inner)
            w1, b1 = _linear(cfg.mlp_dim, D)
            w2, b2 = _linear(D, cfg.mlp_dim)
            tr[pre + "0.norm.weight"] = torch.ones(D)
            tr[pre + "0.norm.bias"] = torch.zeros(D)
            tr[pre + "0.fn.to_qkv.weight"] = wqkv
            tr[pre + "0.fn.to_out.0.weight"] = wo
            tr[pre + "0.fn.to_out.0.bias"] = bo
            tr[pre + "1.norm.weight"] = torch.ones(D)
            tr[pre + "1.norm.bias"] = torch.zeros(D)
            tr[pre + "1.fn.net.0.weight"] = w1
            tr[pre + "1.fn.net.0.bias"] = b1
            tr[pre + "1.fn.net.3.weight"] = w2
            tr[pre + "1.fn.net.3.bias"] = b2
    head = OrderedDict()
    head["encoder.mlp_head.0.weight"] = torch.ones(D)
    head["encoder.mlp_head.0.bias"] = torch.zeros(D)
    w, b = _linear(cfg.n_classes * cfg.spatial_patch ** 2, D)
    head["encoder.mlp_head.1.weight"] = w
    head["encoder.mlp_head.1.bias"] = b
    if not with_mim:  # bare encoder (finetune.py:67-85): no SimMIM draws (keys keep the 'encoder.' prefix)
        out = OrderedDict()
        for d in (pos, enc, tr, head):
            out.update(d)
        return out
    mim = OrderedDict()
    mask_token = torch.randn(D)
    if cfg.to_pixels_per_spectral_block:
        for i in range(S):
            w, b = _linear(P, D)
            mim[f"to_pixels.layers.{i}.weight"] = w
            mim[f"to_pixels.layers.{i}.bias"] = b
    else:
        w, b = _linear(P, D)
        mim["to_pixels.weight"] = w
        mim["to_pixels.bias"] = b
    out = OrderedDict()
    out["mask_token"] = mask_token
    # state_dict order: pos params are registered before the patch embedding's sub-module? No --
    # to_patch_embedding (a sub-module) is assigned first (:333), but nn.Module.state_dict lists a
    # module's OWN parameters before its children, so pos_* precede to_patch_embedding.*.
    for d in (pos, enc, tr, head, mim):
        out.update(d)
    return out


def param_names(cfg: OracleConfig):
    return list(init_params(cfg).keys())


def encoder_only(params):
    """Strip the 'encoder.' prefix and drop SimMIM-only keys (what load_checkpoint keeps,
    src/utils.py:281-285)."""
    return OrderedDict((k[len("encoder."):], v) for k, v in params.items() if k.startswith("encoder."))


# ----------------------------------------------------------------------------------------------
# forward pieces
# ----------------------------------------------------------------------------------------------

def layer_norm(x, w, b):
    """nn.LayerNorm, eps 1e-5 (vit_spatial_spectral.py:25,194-195)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def to_patches(img, cfg):
    """Rearrange 'b (c p0)(h p1)(w p2) -> b c (h w)(p0 p1 p2)' (vit_spatial_spectral.py:197-202)."""
    B = img.shape[0]
    p0, p1 = cfg.spectral_patch, cfg.spatial_patch
    h = w = cfg.Nsq
    x = img.reshape(B, cfg.S, p0, h, p1, w, p1).permute(0, 1, 3, 5, 2, 4, 6)
    return x.reshape(B, cfg.S, h * w, p0 * p1 * p1)


def pos_table(params, cfg):
    """[1, T, D] additive table: learned pos_embedding[:, :T] (vit_simmim_original.py:241) or
    concat(pos_embed, channel_embed) broadcast (vit_spatial_spectral.py:501-516)."""
    if cfg.spectral_pos_embed:
        pe = params["encoder.pos_embed"].unsqueeze(1).expand(-1, cfg.S, -1, -1)
        ce = params["encoder.channel_embed"].unsqueeze(2).expand(-1, -1, cfg.N, -1)
        return torch.cat((pe, ce), dim=-1).reshape(1, cfg.T, cfg.dim)
    return params["encoder.pos_embedding"][:, : cfg.T]


def encoder_embed(params, img, cfg):
    """BlockwisePatchEmbedding.embed (vit_spatial_spectral.py:210-222): LN(P) -> per-block
    Linear(P->D) -> stack (g n) -> LN(D).  Returns (patches [B,T,P] raw, tokens [B,T,D])."""
    pe = "encoder.to_patch_embedding."
    patches = to_patches(img, cfg)
    xn = layer_norm(patches, params[pe + "pre_norm.weight"], params[pe + "pre_norm.bias"])
    embeds = []
    for i in range(cfg.S):
        w = params[pe + f"blockwise_embed.{i}.weight"]
        b = params[pe + f"blockwise_embed.{i}.bias"]
        embeds.append(xn[:, i] @ w.t() + b)
    e = torch.stack(embeds, dim=1).reshape(img.shape[0], cfg.T, cfg.dim)
    e = layer_norm(e, params[pe + "post_norm.weight"], params[pe + "post_norm.bias"])
    return patches.reshape(img.shape[0], cfg.T, cfg.P), e


def attention(x, wqkv, wo, bo, heads, drop=None):
    """Attention.forward (vit_spatial_spectral.py:67-78): bias-free qkv, chunk q|k|v, head-major
    (h d) split, softmax(q k^T * dim_head^-0.5) v, merge, out projection.
    drop: optional dict of pre-scaled keep masks {1: [Bq,h,n,n], 2: [Bq,n,D]} standing in for the two
    nn.Dropout sites (:73-74 attention probabilities, :62 after to_out)."""
    Bq, n, _ = x.shape
    qkv = x @ wqkv.t()
    inner = wqkv.shape[0] // 3
    dh = inner // heads
    q, k, v = qkv.split(inner, dim=-1)
    q = q.reshape(Bq, n, heads, dh).transpose(1, 2)
    k = k.reshape(Bq, n, heads, dh).transpose(1, 2)
    v = v.reshape(Bq, n, heads, dh).transpose(1, 2)
    dots = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    attn = torch.softmax(dots, dim=-1)
    if drop is not None:
        attn = attn * drop[1]
    out = (attn @ v).transpose(1, 2).reshape(Bq, n, inner)
    out = out @ wo.t() + bo
    if drop is not None:
        out = out * drop[2]
    return out


def gelu_erf(x):
    """nn.GELU() default = exact erf form (vit_spatial_spectral.py:37)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def block(x, params, pre, heads, drop=None):
    """One Transformer layer: x = attn(LN(x)) + x; x = ff(LN(x)) + x (vit_spatial_spectral.py:100-104).
    drop (optional): pre-scaled keep masks for the four dropout sites (1, 2 in Attention; 3 after GELU :38;
    4 after the second Linear :40)."""
    h = layer_norm(x, params[pre + "0.norm.weight"], params[pre + "0.norm.bias"])
    x = attention(h, params[pre + "0.fn.to_qkv.weight"], params[pre + "0.fn.to_out.0.weight"],
                  params[pre + "0.fn.to_out.0.bias"], heads, drop) + x
    h = layer_norm(x, params[pre + "1.norm.weight"], params[pre + "1.norm.bias"])
    h = gelu_erf(h @ params[pre + "1.fn.net.0.weight"].t() + params[pre + "1.fn.net.0.bias"])
    if drop is not None:
        h = h * drop[3]
    o = h @ params[pre + "1.fn.net.3.weight"].t() + params[pre + "1.fn.net.3.bias"]
    if drop is not None:
        o = o * drop[4]
    return o + x


def transformer_forward(params, tokens, cfg, return_mid=False, drop_fn=None):
    """ViTSpatialSpectral.transformer_forward (vit_spatial_spectral.py:410-431,495-499):
    'b (c h w) d -> (b c)(h w) d' -> spatial stack -> '(b c)(h w) d -> (b h w) c d' -> spectral
    stack -> back to 'b (c h w) d'.  No final norm."""
    B = tokens.shape[0]
    S, N, D = cfg.S, cfg.N, cfg.dim
    x = tokens.reshape(B * S, N, D)
    # drop_fn(layer_index, mode, batch) -> dict of masks for that block (tests feed the kernels' own masks)
    for l in range(cfg.depth):
        x = block(x, params, f"encoder.spatial_spectral_transformer.1.layers.{l}.", cfg.heads,
                  drop_fn(l, 0, B) if drop_fn else None)
    mid = x.reshape(B, cfg.T, D)
    x = x.reshape(B, S, N, D).transpose(1, 2).reshape(B * N, S, D)
    for l in range(cfg.depth):
        x = block(x, params, f"encoder.spatial_spectral_transformer.3.layers.{l}.", cfg.heads,
                  drop_fn(cfg.depth + l, 1, B) if drop_fn else None)
    x = x.reshape(B, N, S, D).transpose(1, 2).reshape(B, cfg.T, D)
    return (x, mid) if return_mid else x


def simmim_forward(params, img, cfg, masks=None, drop_fn=None):
    """SimMIMSpatialSpectral.forward (vit_simmim_original.py:203-340).  ``masks`` =
    (bool [B,T], int64 [B,K]) or None to draw them like the reference does (numpy global RNG /
    torch CPU RNG).  Returns a dict with the loss and the intermediates the golden fixtures pin."""
    B = img.shape[0]
    patches, tok_embed = encoder_embed(params, img, cfg)                    # :207-227
    pos = pos_table(params, cfg)                                            # :236-242
    tokens = tok_embed + pos
    mask_tokens = params["mask_token"][None, None, :] + pos                 # :245-249
    K = cfg.K                                                               # :252
    if masks is None:
        masks = make_masks(B, cfg.S, cfg.Nsq, cfg.masking_ratio, cfg.mask_patch_size, cfg.tube_masking,
                           cfg.spatial_patch)
    bool_mask, idx = masks
    tok_masked = torch.where(bool_mask[..., None], mask_tokens, tokens)     # :285
    enc_out, mid = transformer_forward(params, tok_masked, cfg, return_mid=True, drop_fn=drop_fn)  # :298
    br = torch.arange(B)[:, None]
    enc_m = enc_out[br, idx]                                                # :314
    if cfg.to_pixels_per_spectral_block:                                    # :317-330 + :21-40
        blk = idx // cfg.N        # arange(S).repeat_interleave(N)[idx]
        W = torch.stack([params[f"to_pixels.layers.{i}.weight"] for i in range(cfg.S)])  # [S,P,D]
        bvec = torch.stack([params[f"to_pixels.layers.{i}.bias"] for i in range(cfg.S)])
        pred = torch.einsum("bkd,bkpd->bkp", enc_m, W[blk]) + bvec[blk]
    else:
        pred = enc_m @ params["to_pixels.weight"].t() + params["to_pixels.bias"]
    target = patches[br, idx]                                               # :335
    loss = (pred - target).abs().mean() / K                                 # :338
    return dict(loss=loss, tok_embed=tok_embed, tok_masked=tok_masked, after_spatial=mid,
                enc_out=enc_out, pred=pred, target=target, bool_mask=bool_mask, masked_indices=idx)


def classify_forward(params, img, cfg):
    """ViTSpatialSpectral.forward (vit_spatial_spectral.py:518-564, head :481-493), eval mode:
    embed + pos -> transformer -> mean over spectral axis -> LN -> Linear -> [B, n_classes, H, W]."""
    B = img.shape[0]
    _, tok = encoder_embed(params, img, cfg)
    x = transformer_forward(params, tok + pos_table(params, cfg), cfg)
    x = x.reshape(B, cfg.S, cfg.Nsq, cfg.Nsq, cfg.dim).mean(dim=1)
    x = layer_norm(x, params["encoder.mlp_head.0.weight"], params["encoder.mlp_head.0.bias"])
    x = x @ params["encoder.mlp_head.1.weight"].t() + params["encoder.mlp_head.1.bias"]
    p = cfg.spatial_patch
    x = x.reshape(B, cfg.Nsq, cfg.Nsq, p, p, cfg.n_classes).permute(0, 1, 3, 2, 4, 5)
    x = x.reshape(B, cfg.Nsq * p, cfg.Nsq * p, cfg.n_classes)
    return torch.moveaxis(x, -1, 1)
