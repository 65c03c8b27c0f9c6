#!/usr/bin/env python3
"""Where the bf16 loss error comes from -- a CPU experiment on the oracle, no GPU needed.

The bf16 forward (maskedsst_amd/csrc/msst_fwd3.hip) rounds to bf16 at ten points per block and nowhere else (the residual stream, the
LayerNorm statistics, the softmax and every accumulation are fp32): LN1 rows, Wqkv, q / k / v, the (dropped) probabilities P, the
attention output O, Wout, LN2 rows, W1, the GELU output, W2.  This script replays the ORACLE's forward (oracle/model.py, eval mode) with
those roundings emulated by round-to-nearest-even at exactly those points, on the two depth-12 reference fixtures, and prints
  * the loss error of the all-bf16 emulation against the reference anchor (what the kernels measure: 0.8e-4 / 2.4e-4),
  * the same with ONE point at a time kept in fp32 (who owns the error), and with ONLY one point in bf16,
each as the relative loss error on the fixture's own input AND as its RMS over `--draws` other random inputs (a single loss error is one
draw of a zero-mean quantity: a point whose removal moves it by less than the draw-to-draw scatter owns nothing).
usage: python tools/bf16_error_table.py [--draws 6] [--json out.jsonl]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle.model as om
from oracle import init_params, simmim_forward
from oracle.masking import make_masks
from conftest import load_golden, oracle_cfg_from, seed_all

POINTS = ["ln1_rows", "wqkv", "qkv", "p", "o", "wout", "ln2_rows", "w1", "gelu_out", "w2"]


SPLIT = False
LOW = torch.bfloat16   # --dtype f16: the same ten points rounded to IEEE half instead (11 significant bits; the forward's operands are
                       # LayerNorm rows, weights, probabilities and bounded activations: all inside half's range)


def bf(t):
    return t.to(LOW).to(torch.float32)


def make_block(on):
    """oracle.model.block with bf16 rounding at the points in `on`"""
    r = lambda name, t: bf(t) if name in on else t

    def block(x, params, pre, heads, drop=None):
        h = r("ln1_rows", om.layer_norm(x, params[pre + "0.norm.weight"], params[pre + "0.norm.bias"]))
        wqkv, wo, bo = params[pre + "0.fn.to_qkv.weight"], params[pre + "0.fn.to_out.0.weight"], params[pre + "0.fn.to_out.0.bias"]
        Bq, n, _ = h.shape
        inner = wqkv.shape[0] // 3
        wr = r("wqkv", wqkv)
        if "wqk" in on or "wv" in on:      # (--split: the q | k rows and the v rows of to_qkv.weight rounded separately)
            wr = torch.cat((bf(wqkv[:2 * inner]) if "wqk" in on else wqkv[:2 * inner], bf(wqkv[2 * inner:]) if "wv" in on else wqkv[2 * inner:]))
        qkv = r("qkv", h @ wr.t())
        dh = inner // heads
        q, k, v = (t.reshape(Bq, n, heads, dh).transpose(1, 2) for t in qkv.split(inner, dim=-1))
        attn = r("p", torch.softmax((q @ k.transpose(-1, -2)) * (dh ** -0.5), dim=-1))
        o = r("o", (attn @ v).transpose(1, 2).reshape(Bq, n, inner))
        x = o @ r("wout", wo).t() + bo + x
        h2 = r("ln2_rows", om.layer_norm(x, params[pre + "1.norm.weight"], params[pre + "1.norm.bias"]))
        g = r("gelu_out", om.gelu_erf(h2 @ r("w1", params[pre + "1.fn.net.0.weight"]).t() + params[pre + "1.fn.net.0.bias"]))
        return g @ r("w2", params[pre + "1.fn.net.3.weight"]).t() + params[pre + "1.fn.net.3.bias"] + x
    return block


def losses(name, draws):
    g = load_golden(name)
    cfg = oracle_cfg_from(g["cfg"])
    seed_all(5)
    params = init_params(cfg)
    B = g["cfg"]["B"]
    xs = [torch.randn(B, cfg.bands, 8, 8)]
    masks = [None]
    gen = torch.Generator().manual_seed(1234)
    for _ in range(draws):
        xs.append(torch.randn(B, cfg.bands, 8, 8, generator=gen))
    real = om.block
    out = {}
    variants = [("fp32", set())] + [("all_bf16", set(POINTS))] + [("fp32:" + p, set(POINTS) - {p}) for p in POINTS] + \
               [("only:" + p, {p}) for p in POINTS]
    if SPLIT:
        variants = [("fp32", set()), ("only:wq+wk", {"wqk"}), ("only:wv", {"wv"}),
                    ("all but wv (the QK path and every other point rounded, Wv exact)", (set(POINTS) - {"wqkv"}) | {"wqk"})]
    state = (np.random.get_state(), torch.get_rng_state())
    for tag, on in variants:
        om.block = make_block(on)
        ls = []
        try:
            for x in xs:
                np.random.set_state(state[0]); torch.set_rng_state(state[1])      # the same masks for every variant
                with torch.no_grad():
                    ls.append(float(simmim_forward(params, x, cfg)["loss"]))
        finally:
            om.block = real
        out[tag] = ls
    return float(g["loss"]), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=6)
    ap.add_argument("--json", default="")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--split", action="store_true", help="only: to_qkv.weight's q | k rows against its v rows")
    a = ap.parse_args()
    global LOW, SPLIT
    SPLIT = a.split
    LOW = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    torch.set_num_threads(8)
    rows = []
    for name in ("simmim_50b_L12_B8.npz", "simmim_200b_L12_B4.npz"):
        anchor, out = losses(name, a.draws)
        ref = out["fp32"]
        assert abs(ref[0] - anchor) <= 2e-6 * anchor, (ref[0], anchor)      # the emulation with nothing rounded IS the oracle
        print(f"== {name}: anchor {anchor:.9e}")
        print(f"{'variant':18s} {'loss err (fixture input)':>26s} {'RMS over draws':>16s}")
        for tag, ls in out.items():
            if tag == "fp32":
                continue
            e0 = (ls[0] - ref[0]) / ref[0]
            rms = float(np.sqrt(np.mean([((l - r) / r) ** 2 for l, r in zip(ls[1:], ref[1:])]))) if len(ls) > 1 else float("nan")
            print(f"{tag:18s} {e0:+26.3e} {rms:16.3e}")
            rows.append(dict(fixture=name, dtype=a.dtype, variant=tag, loss_rel_err=e0, rms_over_draws=rms, draws=a.draws))
    if a.json:
        with open(a.json, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
