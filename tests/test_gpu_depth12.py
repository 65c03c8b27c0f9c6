"""GPU parity at the BASELINE depth (configs 2 and 3: depth 12 per stack = 24 fused blocks) and at the full
benchmark size (B = 256, 200 bands).

Three legs per depth-12 fixture captured from the reference (tools/make_golden.py):
  * fp32 mode against the fixture itself (loss anchor of SURVEY 8c, stage slices, gradient fingerprints),
  * fp32 and bf16 mode against the CPU oracle on the same seeded inputs (full tensors, every gradient),
  * the bf16 kernels that bench.py times (block_fwd_hw, block_bwd_attn_bf16, block_bwd_mlp) are the ones exercised:
    nothing here sets MSST_DBG.
Every measured error is appended to gpurun_out/parity_r02.jsonl (scratch) so that the bars below can be quoted
next to the measurements in DESIGN.md; every bf16 bar is 3-4x the measured value (round 4; measured values in comments): a
bar at 2x one run trips on the next compiler or clock change.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, oracle_cfg_from, fp_np
from util import build_product, relerr, rel_l2, record

pytestmark = pytest.mark.gpu

L12 = ["simmim_200b_L12_B4.npz", "simmim_50b_L12_B8.npz", "simmim_50b_L12_B8_zeropad.npz"]
# Round 6: the same two shapes with PEAKY attention rows -- every to_qkv.weight x4 after construction, built by the
# reference's own modules (tools/make_golden.py peaky).  Logit std ~5, max |logit| 25-34, mean row maximum of the softmax 0.5-0.8 (the
# fixtures carry the measured `attn_stats`): the regime a trained model lives in, where a softmax / scale error moves every stage.
# (The scores go with the SQUARE of the factor.  x8 -- logit std 22, rows 0.9 one-hot -- and x16 are past what fp32 can pin: the
# reference and the oracle, two fp32 CPU evaluations of the same formulas, already differ by 8e-2 of max in enc_out and 0.6 in a
# gradient tensor at x8, and by 4e-3 ... 8e-3 in the LOSS at x16; test_peaky_x8_conditioning below keeps x8 as a robustness case.)
PEAKY = ["simmim_200b_L12_B4_qkv4.npz", "simmim_50b_L12_B8_qkv4.npz"]
PEAKY8 = ["simmim_200b_L12_B4_qkv8.npz", "simmim_50b_L12_B8_qkv8.npz"]
STAGES = ["tok_embed", "tok_masked", "after_spatial", "enc_out", "pred"]


def oracle_run(params, x, cfg, masks):
    from oracle import simmim_forward
    for p in params.values():
        p.requires_grad_(True)
    ref = simmim_forward(params, x, oracle_cfg_from(cfg), masks=masks)
    ref["tok_masked"].retain_grad()
    ref["loss"].backward()
    return ref


def check_masks_against_fixture(masks, g):
    bits = np.packbits(masks[0].numpy().astype(np.uint8), axis=-1)
    np.testing.assert_array_equal(bits, g["bool_mask_bits"])
    np.testing.assert_array_equal(masks[1].numpy().astype(np.int16), g["masked_indices"])


@pytest.mark.parametrize("name", L12 + PEAKY)
def test_depth12_fp32_vs_fixture_and_oracle(name):
    """fp32 MFMA mode, 24 blocks: the reference's own numbers (fixture) and the oracle's full tensors."""
    g = load_golden(name)
    cfg = g["cfg"]
    model, params, x = build_product(cfg, precision="fp32", device="cuda")
    np.testing.assert_array_equal(fp_np(x), g["x_fp"])
    masks = model.draw_masks(cfg["B"])
    check_masks_against_fixture(masks, g)
    out = model.engine().simmim_forward_stages(x.cuda(), masks[0], masks[1])
    torch.cuda.synchronize()
    # (1) the reference fixture: loss anchor + strided 64-element slices of every stage
    lf = float(g["loss"])
    assert abs(out["loss"].item() - lf) <= 1e-4 * abs(lf), (out["loss"].item(), lf)
    worst_slice = 0.0
    for k in STAGES:
        flat = out[k].detach().reshape(-1).cpu()
        stride = max(1, flat.numel() // 64)
        got, want = flat[::stride][:64].numpy(), g["i_slice/" + k]
        worst_slice = max(worst_slice, float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30)))
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-4 * float(np.abs(want).max()), err_msg=k)
    # (2) the oracle, full tensors
    ref = oracle_run(params, x, cfg, masks)
    stage_err = {k: relerr(out[k], ref[k]) for k in STAGES}
    assert max(stage_err.values()) < 1e-4, stage_err
    # (3) gradients: every tensor vs the oracle (element-wise) and vs the fixture fingerprints
    loss = model(x.cuda(), masks=masks)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - lf) <= 1e-4 * abs(lf)
    peaky = name in PEAKY
    got_grads = {pname: p.grad for pname, p in model.named_parameters()}
    if peaky:
        # The L1 gradient is sign(pred - target).  On the peaky fixtures fp32 round-off is amplified ~40x through the 24 blocks (stages
        # 2e-5 of max instead of 6e-7; oracle fp32 vs fp64: 1e-5), so a few of the 35 840 entries with pred ~= target flip sign between
        # two fp32 evaluations and every flip moves a gradient sum by 6e-5 of its scale (measured with the kernels' own signs:
        # mask_token 4.7e-3).  As in the bf16 tests the backward is therefore fed the ORACLE's sign pattern (= the reference's).
        from maskedsst_amd.masking import inverse_csr
        eng = model.engine()
        sgn = torch.sign(ref["pred"] - ref["target"]).detach().cuda().contiguous()
        flips = float((out["dpred"].cpu() != sgn.cpu()).float().mean())
        assert flips < 2e-3, flips
        ptr, pos = inverse_csr(masks[1].numpy(), eng.S * eng.N)
        dy = eng.head_bwd(out["enc_out"], sgn, torch.from_numpy(ptr).cuda(), torch.from_numpy(pos).cuda())
        dx0 = eng.blocks_bwd(out["acts"], out["x1s"], dy)
        eng.tokenize_bwd(x.cuda(), masks[0].to(torch.uint8).cuda(), dx0)
        torch.cuda.synchronize()
        flat = {id(p): n for n, p in eng.trainable()}
        got_grads = {pname: (eng.fp.view(flat[id(p)], eng.fp.grad) if id(p) in flat else None) for pname, p in model.named_parameters()}
    grad_tol, fp_tol = (5e-4, 1e-3) if peaky else (2e-4, 5e-4)
    worst, gsq = 0.0, 0.0
    for pname, p in model.named_parameters():
        gr = params[pname].grad
        if gr is None:
            assert ("g_none/" + pname) in g and p.grad is None, pname
            continue
        gg = got_grads[pname]
        e = relerr(gg, gr)
        worst = max(worst, e)
        assert e < grad_tol, (pname, e)
        fpr, got = g["g_fp/" + pname], fp_np(gg.cpu())
        assert abs(got[1] - fpr[1]) <= fp_tol * fpr[1] + 1e-12, (pname, got[1], fpr[1])
        gsq += float((gg.double() ** 2).sum())
    assert abs(gsq ** 0.5 - float(g["grad_l2"])) <= (1e-3 if peaky else 2e-4) * float(g["grad_l2"])
    record("depth12_fp32", fixture=name, loss=out["loss"].item(), loss_ref=lf, worst_slice=worst_slice,
           stage_err=stage_err, worst_grad=worst)


# bf16 bars: 3-4x the errors measured on MI355X (profiles/r0N_parity_measured.jsonl).
# Measured through 24 blocks: loss 0.7e-4 / 2.6e-4 / 2.3e-4 relative to the reference anchor, worst stage (max-norm)
# 3.5e-3, dx0 2.8e-3 rel-L2, worst parameter-gradient tensor 5.4e-3 rel-L2 (median 2.8e-3), 1 - cosine 2.6e-5.
BF16_BARS = {
    "simmim_200b_L12_B4.npz": dict(loss=2.8e-4, stage=1.2e-2, dx0=1e-2, grad=1.9e-2, cos=0.9999),
    "simmim_50b_L12_B8.npz": dict(loss=9e-4, stage=1.2e-2, dx0=1e-2, grad=1.9e-2, cos=0.9999),
    "simmim_50b_L12_B8_zeropad.npz": dict(loss=9e-4, stage=1.2e-2, dx0=1e-2, grad=1.9e-2, cos=0.9999),
    # peaky rows END TO END: a bf16 q / k pair (2^-9 relative each) moves a logit of 30 by ~0.06, i.e. a probability by 6 %, and the
    # x4 models amplify every error ~40x through their 24 blocks (fp32 kernels: 2e-5 of max at the encoder output instead of 6e-7).
    # Measured: loss 4e-5 / 2e-3, stages 12 % / 22 % of max, dx0 0.37 / 0.58 rel-L2, worst gradient tensor 0.60 / 0.67, cosine 0.96 /
    # 0.835 -- the conditioning of the MODEL in bf16, not a kernel error: block by block on the oracle's own activations the same
    # kernels sit at 0.5 % (forward) / 1.3 % (backward), test_blocks_teacher_forced_on_peaky_rows.  The numbers are recorded; the bars
    # here only catch a broken kernel (NaN, wrong masks, a wrong scale: cosine < 0.5).
    "simmim_200b_L12_B4_qkv4.npz": dict(loss=1e-2, stage=1.0, dx0=2.0, grad=3.0, cos=0.5),
    "simmim_50b_L12_B8_qkv4.npz": dict(loss=1e-2, stage=1.0, dx0=2.0, grad=3.0, cos=0.5),
}


@pytest.mark.parametrize("name", L12 + PEAKY)
def test_depth12_bf16_vs_oracle(name):
    """The benchmarked bf16 kernels through 24 blocks: loss vs the reference anchor, every stage vs the oracle, and --
    with the oracle's sign pattern fed to the backward (the L1 gradient is sign(pred - target); bf16 rounding flips
    the entries with pred ~= target) -- dx0 and every parameter gradient in relative L2."""
    from maskedsst_amd.masking import inverse_csr
    g = load_golden(name)
    cfg = g["cfg"]
    bars = BF16_BARS[name]
    model, params, x = build_product(cfg, precision="bf16", device="cuda")
    masks = model.draw_masks(cfg["B"])
    check_masks_against_fixture(masks, g)
    eng = model.engine()
    xc = x.cuda()
    out = eng.simmim_forward_stages(xc, masks[0], masks[1])
    torch.cuda.synchronize()
    ref = oracle_run(params, x, cfg, masks)
    lf = float(g["loss"])
    loss_err = abs(out["loss"].item() - lf) / abs(lf)
    stage_err = {k: relerr(out[k], ref[k]) for k in STAGES}
    stage_l2 = {k: rel_l2(out[k], ref[k]) for k in STAGES}
    # end to end (own sign pattern): loss + whole-gradient cosine
    loss = model(xc, masks=masks)
    loss.backward()
    torch.cuda.synchronize()
    ga, gb = [], []
    for pname, p in model.named_parameters():
        if params[pname].grad is not None:
            ga.append(p.grad.detach().double().cpu().reshape(-1))
            gb.append(params[pname].grad.double().reshape(-1))
    ga, gb = torch.cat(ga), torch.cat(gb)
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    # kernels with the oracle's sign pattern
    sgn = torch.sign(ref["pred"] - ref["target"]).detach().cuda().contiguous()
    ptr, pos = inverse_csr(masks[1].numpy(), eng.S * eng.N)
    dy = eng.head_bwd(out["enc_out"], sgn, torch.from_numpy(ptr).cuda(), torch.from_numpy(pos).cuda())
    dx0 = eng.blocks_bwd(out["acts"], out["x1s"], dy)
    eng.tokenize_bwd(xc, masks[0].to(torch.uint8).cuda(), dx0)
    torch.cuda.synchronize()
    dx0_err = rel_l2(dx0, ref["tok_masked"].grad)
    flat = {id(p): n for n, p in eng.trainable()}
    gerr = {}
    for pname, p in model.named_parameters():
        gr = params[pname].grad
        if gr is None:
            continue
        gerr[pname] = rel_l2(eng.fp.view(flat[id(p)], eng.fp.grad), gr)
    worst_name = max(gerr, key=gerr.get)
    record("depth12_bf16", fixture=name, fwd_half=bool(eng.fwd_half), loss_err=loss_err, stage_err=stage_err, stage_l2=stage_l2, cos=cos,
           dx0_err=dx0_err, worst_grad=gerr[worst_name], worst_grad_name=worst_name,
           median_grad=float(np.median(list(gerr.values()))))
    assert loss_err < bars["loss"], loss_err
    if eng.fwd_half and name in L12:
        # round 6: with IEEE-half GEMM operands in the forward (MSST_FWD_HALF, the default) the timed kernels meet north_star's loss
        # tolerance against the REFERENCE's anchors on all three depth-12 fixtures (bf16 operands: 0.8e-4 / 2.6e-4 / 2.3e-4)
        assert loss_err < 1e-4, loss_err
    assert max(stage_err.values()) < bars["stage"], stage_err
    assert cos > bars["cos"], cos
    assert dx0_err < bars["dx0"], dx0_err
    assert gerr[worst_name] < bars["grad"], (worst_name, gerr[worst_name])


# teacher-forced bars (bf16, per block; rel-L2): 3.5x the values measured on MI355X -- forward output 4.7e-3, input gradient 1.27e-2,
# worst parameter-gradient tensor of a block 1.45e-2 (3x the randomly initialised models': peaky rows); with dropout 0.1 (the
# kernels' masks fed to the oracle) the same within 10 %.  fp32 kernels: 8.9e-7 / 1.6e-6 / 1.9e-6 against bars of 1e-4 / 2e-4 / 2e-4.
TF_BARS = dict(y=1.6e-2, dx=4.5e-2, grad=5e-2)


@pytest.mark.parametrize("prec,p_drop", [("bf16", 0.0), ("bf16", 0.1), ("fp32", 0.0)], ids=["bf16", "bf16-drop0.1", "fp32"])
@pytest.mark.parametrize("name", PEAKY)
def test_blocks_teacher_forced_on_peaky_rows(name, prec, p_drop):
    """Round 6: the benchmarked bf16 block kernels on PEAKY attention rows, block by block.  Through 24 blocks
    of the x4 fixtures every rounding error is amplified ~40x (fp32 kernels: 2e-5 of max at the encoder output instead of 6e-7), so
    an end-to-end bf16 comparison says little (measured: 12-22 % of max, cosine 0.84-0.96 -- test_depth12_bf16_vs_oracle records
    it).  Here every block is fed the ORACLE's input for that block and, in the backward, the oracle's gradient at its output: nothing
    is carried from block to block, each of the 24 blocks (12 spatial with 64-token sequences, 12 spectral with short ones) is held
    to the oracle on its own -- forward output, input gradient and every parameter gradient of the block -- with logit std ~5 and
    row maxima of 0.5-0.8.  The role-split forward (with its saved LN1 rows, bf16 x1 rows and softmax statistics) and the two-head
    attention backward that consumes them are the kernels exercised (bf16); fp32: the template kernels at 1e-4 / 2e-4."""
    import oracle.model as om
    from oracle import simmim_forward
    from dropout import make_drop_fn
    g = load_golden(name)
    cfg = g["cfg"]
    model, params, x = build_product(cfg, precision=prec, device="cuda")
    masks = model.draw_masks(cfg["B"])
    check_masks_against_fixture(masks, g)
    ocfg = oracle_cfg_from(cfg)
    B, S, N, T = cfg["B"], ocfg.S, ocfg.N, ocfg.T
    drop = (p_drop, 911) if p_drop else (0.0, 0)
    for p in params.values():
        p.requires_grad_(True)
    ins = []
    real_block = om.block

    def recording_block(x_, params_, pre, heads, drop=None):
        x_.retain_grad()
        ins.append(x_)
        return real_block(x_, params_, pre, heads, drop)

    om.block = recording_block
    try:
        ref = simmim_forward(params, x, ocfg, masks=masks,
                             drop_fn=make_drop_fn(drop[0], drop[1], ocfg.S, ocfg.N, ocfg.heads) if p_drop else None)
        ref["enc_out"].retain_grad()
        ref["loss"].backward()
    finally:
        om.block = real_block
    L = cfg["depth"]
    assert len(ins) == 2 * L

    def to_tok(t, i):     # a block's rows in the oracle's sequence layout -> [B, T, 96] in token order b (c h w)
        t = t.detach()
        return (t.reshape(B, T, 96) if i < L else t.reshape(B, N, S, 96).transpose(1, 2).reshape(B, T, 96)).contiguous()

    eng = model.engine()
    eng.prep_weights()
    flat = {id(p): n for n, p in eng.trainable()}
    bf = prec == "bf16"
    worst = dict(y=0.0, dx=0.0, grad=0.0)
    worst_at = {}
    for i in range(2 * L):
        x_i = to_tok(ins[i], i).cuda()
        y_ref = to_tok(ins[i + 1], i + 1) if i + 1 < 2 * L else ref["enc_out"].detach()
        dy_ref = to_tok(ins[i + 1].grad, i + 1) if i + 1 < 2 * L else ref["enc_out"].grad
        dx_ref = to_tok(ins[i].grad, i)
        acts, x1s = [x_i], []
        eng._fwd_block(acts, x1s, i, True, drop, bf, bf, 0)
        e_y = rel_l2(acts[1], y_ref) if bf else relerr(acts[1], y_ref)
        eng.fp.grad.zero_()
        dx = eng.block_bwd_single(i, x_i, x1s[0], dy_ref.cuda().contiguous(), drop=drop)
        torch.cuda.synchronize()
        e_dx = rel_l2(dx, dx_ref) if bf else relerr(dx, dx_ref)
        stack, l = ("1", i) if i < L else ("3", i - L)
        pre = f"encoder.spatial_spectral_transformer.{stack}.layers.{l}."
        e_g = 0.0
        for pname, p in model.named_parameters():
            if pname.startswith(pre):
                e = (rel_l2 if bf else relerr)(eng.fp.view(flat[id(p)], eng.fp.grad), params[pname].grad)
                if e > e_g:
                    e_g, worst_at["grad"] = e, pname
        for k, e in (("y", e_y), ("dx", e_dx), ("grad", e_g)):
            worst[k] = max(worst[k], e)
    record("blocks_teacher_forced_peaky", fixture=name, prec=prec, p_drop=p_drop, worst_y=worst["y"], worst_dx=worst["dx"], worst_grad=worst["grad"],
           worst_grad_name=worst_at.get("grad", ""))
    if bf:
        assert worst["y"] < TF_BARS["y"] and worst["dx"] < TF_BARS["dx"] and worst["grad"] < TF_BARS["grad"], worst
    else:
        assert worst["y"] < 1e-4 and worst["dx"] < 2e-4 and worst["grad"] < 2e-4, worst


@pytest.mark.parametrize("name", PEAKY8)
def test_peaky_x8_conditioning(name):
    """to_qkv.weight x8: logit std ~22, max |logit| 100-130, softmax rows 0.9 one-hot -- past the point where fp32 pins anything
    element-wise (the reference and the oracle differ by 8e-2 of max in enc_out on the CPU; tests/test_oracle_golden.py holds those
    numbers).  What is still well defined is checked: masks bit-exact, the fp32-mode loss against the reference's to 1e-3 (reference vs
    oracle: 2e-5 / 1.2e-4), every stage finite and no further from the oracle than the oracle is from the reference (x4 margin), and
    the bf16 kernels -- exp2 of scores up to 130 * log2 e, saved lse included -- finite with a loss within 5 %."""
    g = load_golden(name)
    cfg = g["cfg"]
    model, params, x = build_product(cfg, precision="fp32", device="cuda")
    masks = model.draw_masks(cfg["B"])
    check_masks_against_fixture(masks, g)
    eng = model.engine()
    out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1])
    torch.cuda.synchronize()
    lf = float(g["loss"])
    loss_err = abs(out["loss"].item() - lf) / lf
    from oracle import simmim_forward
    with torch.no_grad():
        ref = simmim_forward(params, x, oracle_cfg_from(cfg), masks=masks)
    stage_err = {k: relerr(out[k], ref[k]) for k in STAGES}
    eng.set_precision("bf16")
    model.zero_grad(set_to_none=True)
    loss = model(x.cuda(), masks=masks)
    loss.backward()
    torch.cuda.synchronize()
    bf16_loss_err = abs(loss.item() - lf) / lf
    finite = all(torch.isfinite(p.grad).all().item() for p in model.parameters() if p.grad is not None)
    record("peaky_x8_conditioning", fixture=name, loss_err=loss_err, stage_err=stage_err, bf16_loss_err=bf16_loss_err)
    assert loss_err < 5e-3, loss_err          # measured 4e-5 / 1.3e-3
    assert all(torch.isfinite(out[k]).all() for k in STAGES)
    assert max(stage_err.values()) < 0.4, stage_err
    assert finite and bf16_loss_err < 5e-2, bf16_loss_err


@pytest.mark.parametrize("name", ["simmim_50b_L12_B8.npz", "simmim_200b_L12_B4.npz"])
def test_half_vs_bf16_operand_forward(name, monkeypatch):
    """MSST_FWD_HALF (round 6): the role-split forward with IEEE-half GEMM operands against the same kernel with bf16 operands
    (MSST_FWD_HALF=0), both against the oracle and the reference's loss anchor.  tools/bf16_error_table.py predicts it on the CPU: the
    bf16 forward's loss error is systematic and owned by the rounding of the weights (-2.7e-4 / +0.8e-4 on these fixtures); with half
    operands 7e-6.  Required: half meets 1e-4 on the anchor and is closer to the oracle at every stage."""
    g = load_golden(name)
    cfg = g["cfg"]
    model, params, x = build_product(cfg, precision="bf16", device="cuda")
    masks = model.draw_masks(cfg["B"])
    from oracle import simmim_forward
    with torch.no_grad():
        ref = simmim_forward(params, x, oracle_cfg_from(cfg), masks=masks)
    eng = model.engine()
    lf = float(g["loss"])
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MSST_FWD_HALF", flag)
        out = eng.simmim_forward_stages(x.cuda(), masks[0], masks[1])
        torch.cuda.synchronize()
        assert eng.fwd_half == (flag == "1")
        res[flag] = dict(loss_err=abs(out["loss"].item() - lf) / lf, **{k: rel_l2(out[k], ref[k]) for k in ("after_spatial", "enc_out", "pred")})
    monkeypatch.delenv("MSST_FWD_HALF")
    record("half_vs_bf16_operand_forward", fixture=name, half=res["1"], bf16=res["0"])
    assert res["1"]["loss_err"] < 1e-4, res
    assert res["0"]["loss_err"] < 9e-4, res
    for k in ("after_spatial", "enc_out", "pred"):
        assert res["1"][k] < res["0"][k], (k, res)


def test_full_size_b256_bf16():
    """The bench.py launch shape itself: B = 256 cubes of 8x8x200, depth 12, bf16 (5120 tiles per launch, the persistent
    grids wrap 10-20 times).  The forward is per-sample independent given a sample's bool-mask row, so the stages of
    samples [0:4] and [252:256] must equal the oracle run on just those eight samples; the step must be bit-reproducible
    run to run (fixed-order reductions everywhere); the bf16 loss must agree with the fp32-mode loss of the same launch."""
    from oracle import simmim_forward
    cfg = dict(bands=200, depth=12, B=256)
    model, params, x = build_product(cfg, precision="bf16", device="cuda")
    masks = model.draw_masks(cfg["B"])
    eng = model.engine()
    xc = x.cuda()
    out = eng.simmim_forward_stages(xc, masks[0], masks[1])
    torch.cuda.synchronize()
    sel = torch.tensor([0, 1, 2, 3, 252, 253, 254, 255])
    # index rows of a sub-batch are only used by the head (not compared here); any valid indices do
    sub_idx = torch.arange(eng.S * eng.N)[None, :masks[1].shape[1]].repeat(len(sel), 1)
    with torch.no_grad():
        ref = simmim_forward(params, x[sel], oracle_cfg_from(cfg), masks=(masks[0][sel], sub_idx))
    stage_err = {k: relerr(out[k][sel.cuda()], ref[k]) for k in ["tok_embed", "tok_masked", "after_spatial", "enc_out"]}
    assert max(stage_err.values()) < 1e-2, stage_err   # measured 2.8e-3
    assert torch.isfinite(out["loss"])
    del out

    def step():
        for p in model.parameters():
            p.grad = None
        loss = model(xc, masks=masks)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), eng.fp.grad.clone()

    l1, g1 = step()
    l2, g2 = step()
    assert torch.equal(l1, l2) and torch.equal(g1, g2), "bf16 step is not bit-reproducible"
    assert torch.isfinite(g1).all()
    eng.set_precision("fp32")
    with torch.no_grad():
        l32 = model(xc, masks=masks)
    torch.cuda.synchronize()
    eng.set_precision("bf16")
    loss_err = abs(l1.item() - l32.item()) / abs(l32.item())
    record("full_size_b256", stage_err=stage_err, loss_bf16=l1.item(), loss_fp32=l32.item(), loss_err=loss_err)
    assert loss_err < 1.4e-4, (l1.item(), l32.item())   # measured 3.8e-5


def test_adamw_parameters_after_5_steps():
    """The parameters after the 5 reference AdamW steps (fixture `p_fp_after/*`, captured from the reference and
    unused in round 1): lr 0.008 moves every weight by ~0.04 in 5 steps, so an optimiser / gradient error of 1 % of
    a step shows up at the 1e-4 level of the parameter abs-sums."""
    from maskedsst_amd.optim import FusedAdamW
    g = load_golden("adamw_traj_200b_L2_B32.npz")
    model, _, x = build_product(dict(bands=200, depth=2, B=32), precision="fp32", device="cuda")
    opt = FusedAdamW(model, lr=0.008, weight_decay=0.05, grad_clamp=1.0)
    x = x.cuda()
    model.train()
    for _ in range(5):
        opt.zero_grad()
        loss = model(x)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    worst = 0.0
    for pname, p in model.named_parameters():
        ref = g["p_fp_after/" + pname]
        got = fp_np(p.detach().cpu())
        assert got[2] == ref[2], pname
        e = abs(got[1] - ref[1]) / (ref[1] + 1e-30)
        worst = max(worst, e)
        assert e < 2e-3, (pname, got[1], ref[1])
        # first 8 elements: each moved by up to 5 * lr; compare to 5 % of that travel
        np.testing.assert_allclose(got[3:], ref[3:], atol=0.05 * 5 * 0.008, rtol=0, err_msg=pname)
    record("adamw_params_after_5", worst_abs_sum_err=worst)
