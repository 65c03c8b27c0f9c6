"""maskedsst_amd -- MI355X-native (gfx950) implementation of the MaskedSST masked-pretraining hot path.

Drop-in mirrors of the reference modules (same constructor / attribute / state_dict surface):

    from maskedsst_amd import ViTSpatialSpectral, SimMIMSpatialSpectral

Compute runs in hand-written HIP kernels (``maskedsst_amd/csrc``) behind the C-ABI of
``include/msst.h`` (``libmsst.so``, loaded with ctypes).  There is no CPU or eager fallback.
"""
from .vit_spatial_spectral import ViTSpatialSpectral  # noqa: F401
from .vit_simmim_original import SimMIMSpatialSpectral, BlockwiseToPixels  # noqa: F401
from .masking import MaskGenerator  # noqa: F401

__all__ = ["ViTSpatialSpectral", "SimMIMSpatialSpectral", "BlockwiseToPixels", "MaskGenerator"]
