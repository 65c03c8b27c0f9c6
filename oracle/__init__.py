"""CPU oracle for the MaskedSST masked-pretraining hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch-fp32 / numpy restatement of the
reference algorithm (HSG-AIML/MaskedSST ``src/vit_spatial_spectral.py`` and
``src/vit_simmim_original.py``) used as the *checker* for the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``maskedsst_amd/`` does, and the product path raises when the HIP library is missing instead of
falling back to this code.

Pinning: the reference has no tests / golden vectors of its own, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
``tools/make_golden.py`` (which imports ``/root/reference``) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every fixture.
"""
from .model import (  # noqa: F401
    OracleConfig,
    init_params,
    encoder_embed,
    transformer_forward,
    simmim_forward,
    classify_forward,
    param_names,
)
from .masking import MaskGeneratorOracle, make_masks  # noqa: F401
from .pos_embed import sincos_2d, sincos_1d  # noqa: F401
