"""``ViTSpatialSpectral`` -- drop-in mirror of the reference encoder's ``nn.Module`` surface.

Mirrors reference ``src/vit_spatial_spectral.py:256-564`` (constructor signature, attributes read
across the SimMIM seam, ``state_dict`` key schema, parameter draw order -- SURVEY.md 8b / 3.4).
The modules below are *parameter containers*: the compute of the accelerated configuration
(blockwise patch embedding, spatial -> spectral factorised attention) runs in hand-written HIP
kernels through ``libmsst.so``; there is no eager / CPU fallback and calling a container's
``forward`` raises.
"""
from functools import reduce
from operator import mul

import numpy as np
import torch
from torch import nn

from .pos_embed import get_1d_sincos_pos_embed_from_grid, get_2d_sincos_pos_embed


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


class _HipOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container of the fused MI355X path; "
            "call ViTSpatialSpectral / SimMIMSpatialSpectral instead (no eager fallback).")


class PreNorm(_HipOnly):
    """reference :22-29"""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class FeedForward(_HipOnly):
    """reference :32-44 (Linear, GELU(erf), Dropout, Linear, Dropout)"""

    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim), nn.Dropout(dropout),
        )


class Attention(_HipOnly):
    """reference :47-78"""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = (nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
                       if project_out else nn.Identity())


class Transformer(_HipOnly):
    """reference :81-104"""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)),
            ]))


class _Rearrange(nn.Module):
    """Placeholder for the einops ``Rearrange`` layers of the reference Sequential (:410-431): keeps
    the child indices (``spatial_spectral_transformer.1`` / ``.3``) of the state_dict schema.  The
    regrouping itself is done by strided addressing inside the kernels (no copies)."""

    def __init__(self, pattern):
        super().__init__()
        self.pattern = pattern

    def extra_repr(self):
        return self.pattern


class ToPatch(nn.Module):
    """``Rearrange('b (c p0)(h p1)(w p2) -> b c (h w)(p0 p1 p2)')`` (reference :197-202); a cheap
    view/permute kept callable because SimMIM exposes it as ``self.to_patch``."""

    def __init__(self, p0, p1, p2):
        super().__init__()
        self.p0, self.p1, self.p2 = p0, p1, p2

    def forward(self, x):
        b, C, H, W = x.shape
        c, h, w = C // self.p0, H // self.p1, W // self.p2
        x = x.reshape(b, c, self.p0, h, self.p1, w, self.p2).permute(0, 1, 3, 5, 2, 4, 6)
        return x.reshape(b, c, h * w, self.p0 * self.p1 * self.p2)


class BlockwisePatchEmbedding(nn.Module):
    """reference :178-229"""

    def __init__(self, num_channels, transformer_dim, patch_depth, patch_height, patch_width):
        super().__init__()
        assert num_channels % patch_depth == 0, \
            f"Number of channels {num_channels=} not divisible by patch_depth {patch_depth=}"
        self.patch_depth = patch_depth
        self.patch_height = patch_height
        self.patch_width = patch_width
        self.transformer_dim = transformer_dim
        self.patch_dim = reduce(mul, [patch_depth, patch_height, patch_width])
        self.num_blocks = num_channels // patch_depth
        self.pre_norm = nn.LayerNorm(self.patch_dim)
        self.post_norm = nn.LayerNorm(self.transformer_dim)
        self.to_patch = ToPatch(patch_depth, patch_height, patch_width)
        self.blockwise_embed = nn.ModuleList(
            [nn.Linear(self.patch_dim, self.transformer_dim) for _ in range(self.num_blocks)])
        self._owner = None  # set by ViTSpatialSpectral: the fused tokenizer lives in its engine

    def embed(self, patches):
        """tokens = LN(stack_i Linear_i(LN(patches[:, i]))) -- runs the HIP tokenizer (without
        position / mask terms) on patches laid out [B, S, N, P]."""
        return self._owner()._embed_patches(patches)

    def forward(self, x):
        return self.embed(self.to_patch(x))


class MoveAxis(nn.Module):
    def __init__(self, axes):
        super().__init__()
        self.axes = axes

    def forward(self, x):
        return torch.moveaxis(x, *self.axes)


class _HeadRearrange(nn.Module):
    """'b h w (p1 p2 num_classes) -> b (h p1) (w p2) num_classes' (reference :485-491)"""

    def __init__(self, p1, p2, num_classes):
        super().__init__()
        self.p1, self.p2, self.nc = p1, p2, num_classes

    def forward(self, x):
        b, h, w, _ = x.shape
        x = x.reshape(b, h, w, self.p1, self.p2, self.nc).permute(0, 1, 3, 2, 4, 5)
        return x.reshape(b, h * self.p1, w * self.p2, self.nc)


class ViTSpatialSpectral(nn.Module):
    """Same keyword-only constructor as reference :257-301.  Extra keyword ``precision``
    ('bf16' | 'fp32') selects the MFMA operand type of the fused kernels."""

    def __init__(self, *, image_size, spatial_patch_size, spectral_patch_size, num_classes, dim, depth,
                 heads, mlp_dim, spectral_pos_embed=True, pool="mean", blockwise_patch_embed=True,
                 channels=3, dim_head=64, dropout=0.0, emb_dropout=0.0,
                 spectral_pos=list(range(20)), spectral_only=False, spectral_mlp_head=False,
                 pixelwise=False, pos_embed_len=None, precision=None):
        super().__init__()
        image_height, image_width = pair(image_size)
        image_depth = channels
        self.patch_height, self.patch_width = pair(spatial_patch_size)
        self.patch_depth = spectral_patch_size
        self.image_size = image_size
        self.pixels_per_patch = reduce(mul, [self.patch_depth, self.patch_height, self.patch_width])
        self.spectral_pos = np.asarray(spectral_pos.tolist() if torch.is_tensor(spectral_pos) else spectral_pos)
        self.spectral_pos_embed = spectral_pos_embed
        self.blockwise_patch_embed = blockwise_patch_embed
        self.spectral_only = spectral_only
        self.spectral_mlp_head = spectral_mlp_head
        self.pixelwise = pixelwise
        assert (image_height % self.patch_height == 0 and image_width % self.patch_width == 0
                and image_depth % self.patch_depth == 0), \
            "Image dimensions must be divisible by the patch size."
        self.num_spatial_patches_sqrt = image_height // self.patch_height
        self.num_spatial_patches = self.num_spatial_patches_sqrt ** 2
        self.num_spectral_patches = image_depth // self.patch_depth
        self.num_patches = self.num_spatial_patches * self.num_spectral_patches
        assert pool in {"mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"

        # ---- what the fused HIP path covers; everything else fails loudly (no eager fallback) ----
        unsupported = []
        if not blockwise_patch_embed:
            unsupported.append("blockwise_patch_embed=False")
        if spectral_only:
            unsupported.append("spectral_only=True")
        if spectral_mlp_head:
            unsupported.append("spectral_mlp_head=True")
        if pixelwise:
            unsupported.append("pixelwise=True")
        if dim != 96 or dim_head != 64 or mlp_dim != 64:
            unsupported.append(f"dim/dim_head/mlp_dim={dim}/{dim_head}/{mlp_dim} (kernels are built for 96/64/64)")
        if self.patch_height != 1 or self.patch_width != 1:
            unsupported.append("spatial_patch_size != 1")
        if self.num_spatial_patches > 64 or self.num_spectral_patches > 64 or self.patch_depth > 16:
            unsupported.append("more than 64 spatial / spectral tokens per sequence or spectral patch > 16")
        if unsupported:
            raise NotImplementedError(
                "maskedsst_amd accelerates the shipped MaskedSST configuration only; unsupported: "
                + ", ".join(unsupported))
        self.dropout_p = float(dropout)
        self.emb_dropout_p = float(emb_dropout)
        self.heads = heads
        self.depth = depth
        self.precision = precision

        self.to_patch_embedding = BlockwisePatchEmbedding(
            channels, dim, self.patch_depth, self.patch_height, self.patch_width)

        if self.spectral_pos_embed:
            channel_embed_dim = dim // 3
            pos_embed_dim = dim - channel_embed_dim
            self.pos_embed = nn.Parameter(torch.zeros(1, self.num_spatial_patches, pos_embed_dim))
            p_embed = get_2d_sincos_pos_embed(pos_embed_dim, self.num_spatial_patches_sqrt, cls_token=False)
            self.pos_embed.data.copy_(torch.from_numpy(p_embed).float().unsqueeze(0))
            assert len(self.spectral_pos) == self.num_spectral_patches, \
                f"{self.spectral_pos.shape=}, {self.num_spectral_patches=}"
            self.channel_embed = nn.Parameter(torch.zeros(1, self.num_spectral_patches, channel_embed_dim))
            chan_embed = get_1d_sincos_pos_embed_from_grid(channel_embed_dim, self.spectral_pos)
            self.channel_embed.data.copy_(torch.from_numpy(chan_embed).float().unsqueeze(0))
        else:
            if pos_embed_len is not None:
                self.pos_embedding = nn.Parameter(torch.randn(1, pos_embed_len, dim))
            else:
                self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))

        self.dropout = nn.Dropout(emb_dropout)

        c, hw = self.num_spectral_patches, self.num_spatial_patches_sqrt
        self.spatial_spectral_transformer = nn.Sequential(
            _Rearrange("b (c h w) d -> (b c) (h w) d"),
            Transformer(dim, depth, heads, dim_head, mlp_dim, dropout),
            _Rearrange("(b c) (h w) d -> (b h w) c d"),
            Transformer(dim, depth, heads, dim_head, mlp_dim, dropout),
            _Rearrange("(b h w) c d -> b (c h w) d"),
        )
        self.pool = pool
        self.to_latent = nn.Identity()
        self.dim = dim
        num_out_pixels = self.patch_width * self.patch_height
        self.mlp_head = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, num_classes * num_out_pixels),
            _HeadRearrange(self.patch_height, self.patch_width, num_classes),
            MoveAxis((-1, 1)),
        )
        self.num_classes = num_classes

        import weakref
        self.to_patch_embedding._owner = weakref.ref(self)
        self._engine = None
        self._engine_owner = None  # a SimMIM wrapper installs its own engine (covers mask token + to_pixels)

    # ------------------------------------------------------------------
    def engine(self):
        if self._engine_owner is not None and self._engine_owner() is not None:
            return self._engine_owner().engine()
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self, None)
        return self._engine

    def get_pos_embeddings(self):
        """reference :501-516 -- [1, T, D] table (used for inspection; the kernels read the two
        factor tables directly)."""
        channel_embed = self.channel_embed.unsqueeze(2)
        pos_embed = self.pos_embed.unsqueeze(1)
        channel_embed = channel_embed.expand(-1, -1, pos_embed.shape[2], -1)
        pos_embed = pos_embed.expand(-1, channel_embed.shape[1], -1, -1)
        pos_channel = torch.cat((pos_embed, channel_embed), dim=-1)
        return pos_channel.reshape(1, self.num_patches, self.dim)

    def _embed_patches(self, patches):
        return self.engine().embed_patches(patches)

    def transformer_forward(self, x):
        """reference :495-499 -- both transformer stacks on tokens [B, T, D] (fused HIP blocks)."""
        return self.engine().transformer(x)

    def forward_features(self, img):
        """reference :518-534: tokenize + position + (emb dropout) + transformer."""
        return self.engine().features(img)

    def forward(self, img):
        """reference :536-564: features -> mean over the spectral axis -> LN -> Linear ->
        [B, num_classes, H, W]."""
        return self.engine().classify(img)
