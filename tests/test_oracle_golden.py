"""Pin the CPU oracle against golden vectors captured from the reference (tools/make_golden.py).

Protocol mirrors the capture: seed 5 everywhere -> draw params in reference order -> x = randn ->
eval forward -> backward.  Masks / indices must be bit-exact; floats within fp32 round-off of the
reference's own CPU result.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, fp_np, oracle_cfg_from, seed_all, apply_qkv_scale
from oracle import init_params, simmim_forward, classify_forward

SIMMIM = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "simmim_*.npz")))
FAST = [n for n in SIMMIM if "L12" not in n and "B32" not in n]
SLOW = [n for n in SIMMIM if n not in FAST]


def run_oracle(g):
    cfg = oracle_cfg_from(g["cfg"])
    seed_all(5)
    params = init_params(cfg)
    apply_qkv_scale(params.items(), g["cfg"])
    B = g["cfg"]["B"]
    x = torch.randn(B, cfg.bands, cfg.image_size, cfg.image_size)
    if g["cfg"].get("zero_pad_bands"):
        x[:, cfg.bands - g["cfg"]["zero_pad_bands"]:] = 0.0
    for p in params.values():
        p.requires_grad_(True)
    out = simmim_forward(params, x, cfg)
    out["loss"].backward()
    return cfg, params, x, out


def check_fp(a, b, rtol, atol, what):
    # [sum, abssum, numel, first 8]
    assert a[2] == b[2], what
    n = a[2]
    # sums of n fp32-rounded numbers: allow per-element atol
    assert abs(a[0] - b[0]) <= rtol * abs(b[1]) + atol * n, (what, a[0], b[0])
    assert abs(a[1] - b[1]) <= rtol * abs(b[1]) + atol * n, (what, a[1], b[1])
    np.testing.assert_allclose(a[3:], b[3:], rtol=max(rtol, 1e-5) * 10, atol=atol * 10, err_msg=what)


def conditioning(gcfg):
    """How far two fp32 evaluations of the SAME formulas may sit apart on a fixture: 1 for the randomly initialised models.  The
    peaky-attention fixtures (to_qkv.weight x s, logits x s^2) amplify fp32 round-off through 24 blocks; measured oracle vs reference
    on the CPU -- x4: stages 1.3e-5 of max, gradients 2.6e-5 (still pinned at the ordinary bars, slices relative to the tensor's
    max); x8: loss 2e-5 / 1.2e-4, enc_out 8e-2 of max, a gradient tensor's abs-sum 0.6: only the loss, the masks and the
    parameters are held (tests/test_gpu_depth12.py::test_peaky_x8_conditioning says what that fixture is for)."""
    return {None: "exact", 4: "exact", 8: "loss-only"}[gcfg.get("qkv_scale")]


@pytest.mark.parametrize("name", FAST + SLOW)
def test_simmim_matches_reference(name):
    g = load_golden(name)
    cfg, params, x, out = run_oracle(g)
    cond = conditioning(g["cfg"])
    # input + parameter draw order: bit-exact
    np.testing.assert_array_equal(fp_np(x), g["x_fp"])
    assert list(params.keys()) == g["names"] or sorted(params.keys()) == sorted(g["names"])
    assert sum(p.numel() for p in params.values()) == int(g["n_params"])
    for k, p in params.items():
        np.testing.assert_array_equal(fp_np(p), g["p_fp/" + k], err_msg=k)
    # masks: bit-exact
    bits = np.packbits(out["bool_mask"].numpy().astype(np.uint8), axis=-1)
    np.testing.assert_array_equal(bits, g["bool_mask_bits"])
    np.testing.assert_array_equal(out["masked_indices"].numpy().astype(np.int16), g["masked_indices"])
    # loss + intermediates
    if cond == "loss-only":
        assert abs(out["loss"].item() - float(g["loss"])) <= 5e-4 * abs(float(g["loss"]))
        for k in ["tok_embed", "tok_masked", "target"]:      # everything in front of the first attention block is still exact
            check_fp(fp_np(out[k]), g["i_fp/" + k], 1e-5, 2e-6, k)
        return
    assert abs(out["loss"].item() - float(g["loss"])) <= 2e-6 * abs(float(g["loss"])) + 1e-10
    for k in ["tok_embed", "tok_masked", "after_spatial", "enc_out", "pred", "target"]:
        check_fp(fp_np(out[k]), g["i_fp/" + k], 1e-5, 2e-6, k)
        flat = out[k].detach().reshape(-1)
        stride = max(1, flat.numel() // 64)
        want = g["i_slice/" + k]
        np.testing.assert_allclose(flat[::stride][:64].numpy(), want, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(want).max())), err_msg=k)
    # gradients
    gsq = 0.0
    for k, p in params.items():
        if ("g_none/" + k) in g:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
            continue
        ref = g["g_fp/" + k]
        got = fp_np(p.grad)
        scale = max(ref[1] / max(ref[2], 1), 1e-12)
        assert abs(got[0] - ref[0]) <= 2e-3 * ref[1] + 1e-12, (k, got[0], ref[0])
        assert abs(got[1] - ref[1]) <= 2e-4 * ref[1] + 1e-12, (k, got[1], ref[1])
        np.testing.assert_allclose(got[3:], ref[3:], rtol=5e-3, atol=50 * scale * 1e-3, err_msg=k)
        gsq += float((p.grad.double() ** 2).sum())
    assert abs(gsq ** 0.5 - float(g["grad_l2"])) <= 1e-4 * float(g["grad_l2"])
    if "attn_stats" in g:       # the fixture really is peaky: mean row maximum of the softmax >= 0.5 in every sampled block
        assert g["attn_stats"][:, 2].min() >= 0.5 and g["attn_stats"][:, 0].min() > 4.0, g["attn_stats"]


@pytest.mark.parametrize("name", [n for n in SIMMIM if "tiny" in n])
def test_tiny_elementwise(name):
    """Full tensors: every intermediate and every gradient element."""
    g = load_golden(name)
    cfg, params, x, out = run_oracle(g)
    np.testing.assert_array_equal(x.numpy(), g["x"])
    for k, p in params.items():
        np.testing.assert_array_equal(p.detach().numpy(), g["sd/" + k], err_msg=k)
    for k in ["tok_embed", "tok_masked", "after_spatial", "enc_out", "pred", "target"]:
        np.testing.assert_allclose(out[k].detach().numpy(), g["full/" + k], rtol=1e-4, atol=2e-5, err_msg=k)
    for k, p in params.items():
        if ("grad/" + k) not in g:
            continue
        ref = g["grad/" + k]
        tol = 1e-4 * np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-3, atol=tol, err_msg=k)


def test_anchor_values():
    """Loss anchors recorded in SURVEY.md §8c."""
    assert abs(float(load_golden("simmim_200b_L2_B32.npz")["loss"]) - 1.245592372e-03) < 1e-9
    assert abs(float(load_golden("simmim_50b_L12_B8.npz")["loss"]) - 5.703608040e-03) < 1e-9
    assert abs(float(load_golden("simmim_200b_L12_B4.npz")["loss"]) - 1.377874170e-03) < 1e-9
    assert load_golden("simmim_200b_L2_B32.npz")["masked_indices"][1, :4].tolist() == [1184, 1185, 1186, 1187]
    assert load_golden("simmim_50b_L12_B8.npz")["masked_indices"][1, :4].tolist() == [288, 289, 290, 291]
    # parameter-count KATs (inference_example.ipynb:144 for 1,821,564)
    assert int(load_golden("simmim_200b_L2_B32.npz")["n_params"]) == 1_002_916
    assert int(load_golden("simmim_50b_L12_B8.npz")["n_params"]) == 5_071_086
    assert int(load_golden("simmim_200b_L12_B4.npz")["n_params"]) == 5_193_636
    assert int(load_golden("finetune_200b_L4_B2.npz")["n_params"]) == 1_821_564


@pytest.mark.parametrize("name", ["finetune_200b_L4_B2.npz", "finetune_50b_L2_B2_specpos.npz"])
def test_finetune_step(name):
    g = load_golden(name)
    cfg = oracle_cfg_from(g["cfg"])
    seed_all(5)
    params = init_params(cfg, with_mim=False)  # bare encoder: x / labels are drawn right after it
    B = g["cfg"]["B"]
    x = torch.randn(B, cfg.bands, 8, 8)
    label = torch.randint(-1, cfg.n_classes, (B, 8, 8))
    np.testing.assert_array_equal(label.numpy().astype(np.int8), g["label"])
    assert sum(p.numel() for p in params.values()) == int(g["n_params"])
    for p in params.values():
        p.requires_grad_(True)
    logits = classify_forward(params, x, cfg)
    loss = torch.nn.functional.cross_entropy(logits, label, ignore_index=-1)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    check_fp(fp_np(logits), g["logits_fp"], 1e-5, 2e-6, "logits")
    for k, p in params.items():
        ref = g["g_fp/" + k[len("encoder."):]]
        got = fp_np(p.grad)
        assert abs(got[1] - ref[1]) <= 5e-4 * ref[1] + 1e-12, (k, got[1], ref[1])
    gsq = sum(float((p.grad.double() ** 2).sum()) for p in params.values())
    assert abs(gsq ** 0.5 - float(g["grad_l2"])) <= 2e-4 * float(g["grad_l2"])
