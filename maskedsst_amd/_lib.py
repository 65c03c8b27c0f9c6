"""ctypes binding of libmsst.so (the C-ABI declared in include/msst.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.  Build the
library with ``python -m maskedsst_amd.build`` (hipcc, gfx950).
"""
import ctypes
import os
from ctypes import c_int, c_int32, c_uint32, c_long, c_float, c_void_p, c_char_p, POINTER, Structure

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmsst.so")
HEADER_PATH = os.path.normpath(os.path.join(HERE, "..", "include", "msst.h"))

PREC_F32 = 0
PREC_BF16 = 1
MODE_SPATIAL = 0
MODE_SPECTRAL = 1
MLP_SLAB = 64 * 96 + 96 * 64 + 64 + 96 + 96 + 96
ATTN_SLAB = 3 * 64 * 96 + 96 * 64
LN1_SLAB = 288


class MsstError(RuntimeError):
    pass


class MsstPrepJob(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_int32), ("cols", c_int32),
                ("transpose", c_int32), ("pack", c_int32), ("scale_rows", c_int32), ("scale", c_float)]


class MsstBlockWeights(Structure):
    _fields_ = [("struct_bytes", ctypes.c_uint64)] + [(n, c_void_p) for n in (
        "wqkv", "wout", "w1", "w2", "wqkvT", "woutT", "w1T", "w2T",
        "ln1_g", "ln1_b", "bo", "ln2_g", "ln2_b", "b1", "b2", "wqkv32", "woutT32", "wqkvT32",
        "wqkv_h", "wout_h", "w1_h", "w2_h")]


class MsstBlockGrads(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "ln1_g", "ln1_b", "wqkv", "wout", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2")]


_P = c_void_p
BWD_DEFER_REDUCE = 512 << 8   # include/msst.h: MSST_BWD_DEFER_REDUCE
X1_BF16 = 1024 << 8           # include/msst.h: MSST_X1_BF16
SAVED_XN, SAVED_LSE, SAVED_RSTD = 1, 2, 4    # include/msst.h: MSST_SAVED_*
LN1_FROM_XN = 2048 << 8       # include/msst.h: MSST_LN1_FROM_XN
FWD_HALF = 4096 << 8          # include/msst.h: MSST_FWD_HALF
LSE_RENORM = 8192 << 8        # include/msst.h: MSST_LSE_RENORM
PREP_HALF = 256               # include/msst.h: MSST_PREP_HALF
_SIGS = {
    "msst_version": (c_int, []),
    "msst_last_error": (c_char_p, []),
    "msst_prep_weights": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "msst_tokenize_fwd": (c_int, [_P] * 9 + [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_uint32, _P]),
    "msst_cls_head_fwd": (c_int, [_P] * 6 + [c_int, c_int, c_int, c_int, _P]),
    "msst_cls_head_bwd": (c_int, [_P] * 11 + [c_int, c_int, c_int, c_int, _P]),
    "msst_block_lse_floats": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "msst_block_tiles": (c_long, [c_int, c_int, c_int, c_int]),
    "msst_block_fwd": (c_int, [POINTER(MsstBlockWeights), _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_float, c_uint32, c_int, _P, _P, POINTER(c_int), _P]),
    "msst_block_fwd_stack": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint32, c_int,
                                     POINTER(c_int), _P]),
    "msst_head_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                              c_int, _P]),
    "msst_head_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_float, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int,
                              c_int, c_int, _P]),
    "msst_block_bwd": (c_int, [POINTER(MsstBlockWeights), POINTER(MsstBlockGrads), _P, _P, _P, _P, _P, _P, _P,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint32, c_int, _P, _P, _P, _P]),
    "msst_block_bwd_chain": (c_int, [POINTER(MsstBlockWeights), POINTER(MsstBlockGrads), POINTER(MsstBlockWeights),
                                     POINTER(MsstBlockGrads), _P, _P, _P, _P, _P, _P, _P, _P,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint32, c_int, _P, _P, _P,
                                     c_int, _P, _P]),
    "msst_block_bwd_reduce": (c_int, [POINTER(MsstBlockGrads), POINTER(MsstBlockGrads), _P, ctypes.c_long, ctypes.c_long,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "msst_tokenize_bwd": (c_int, [_P] * 10 + [c_int] + [_P] * 8 + [c_int, _P, c_int, c_int, c_int, c_int, c_float,
                                  c_uint32, _P]),
    "msst_debug_stamps": (c_int, [_P]),
    "msst_debug_cu_thief": (c_int, [c_int, c_int, _P, _P]),
    "msst_debug_box_probe": (c_int, [_P, _P, c_long, _P]),
    "msst_profile_enable": (c_int, [c_int]),
    "msst_profile_select": (c_int, [ctypes.c_ulonglong]),
    "msst_profile_sample": (c_int, [c_int]),
    "msst_profile_kernels": (c_int, []),
    "msst_profile_name": (c_char_p, [c_int]),
    "msst_profile_collect": (c_int, [_P, _P]),
    "msst_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, c_float, _P]),
    "msst_layernorm_bwd_slab": (c_long, [c_long, c_int]),
    "msst_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_long, c_int, c_float, _P]),
    "msst_adamw": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, c_float, c_int,
                           c_float, c_float, _P]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGS)


def header_version(path=None):
    """MSST_VERSION of include/msst.h -- the revision this binding (and every struct layout in it) was written against"""
    import re
    with open(path or HEADER_PATH) as f:
        m = re.search(r"^#define\s+MSST_VERSION\s+(\d+)", f.read(), re.M)
    if not m:
        raise MsstError(f"no MSST_VERSION in {path or HEADER_PATH}")
    return int(m.group(1))


def load():
    """Load libmsst.so (once).  Raises MsstError when it is absent -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsstError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m maskedsst_amd.build). "
            "maskedsst_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise MsstError(f"libmsst.so does not export {name} (stale build? run python -m maskedsst_amd.build)")
        fn.restype = res
        fn.argtypes = args
    # a library built from another revision of the header (a stale .so after a checkout: the build is mtime based and the
    # prebuilt library travels with the tree) or a kernel-study build (-DMSST_LAB: msst_version() < 0, results wrong by
    # design) is refused, not loaded
    got, want = int(lib.msst_version()), header_version()
    if got == -want and os.environ.get("MSST_ALLOW_LAB") == "1":
        # kernel-study tools only (tools/exp_lab.sh): timing-only modes may compute wrong results; never set by the product, the tests or bench.py
        import warnings
        warnings.warn("maskedsst_amd: loading a -DMSST_LAB kernel-study build (results may be wrong by design)")
        got = want
    if got != want:
        raise MsstError(
            f"{LIB_PATH} reports msst_version() = {got}, include/msst.h says MSST_VERSION {want}: "
            + ("a kernel-study build (-DMSST_LAB) must not be loaded by the product; " if got < 0 else "stale library; ")
            + "rebuild with python -m maskedsst_amd.build --force")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().msst_last_error()
        raise MsstError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
