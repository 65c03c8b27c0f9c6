#!/usr/bin/env python3
"""bench.py -- SimMIM pre-training throughput of the MI355X-native hot path.

Metric (BASELINE.json): pretrain samples/s (hyperspectral cubes, full fwd + bwd + AdamW step) on
synthetic EnMAP-shape cubes (8x8x200 bands, depth 12 per stack, dim 96, heads 8, mlp 64; mask ratio
0.7 / mask patch 4 / tube masking) -- BASELINE.json configs[2]/[3]; per-GPU batch 256 (global 2048 at
8 GPUs, weak scaling).  One process per GPU; for N>1 launch with
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` (RCCL over xGMI).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      the dominant kernel's algorithmic FLOP/s (DESIGN.md "work model") from HIP events
                recorded around its launches inside the timed steps, vs the dense bf16 MFMA peak
  cpu_baseline  the CPU oracle (plain PyTorch fp32 restatement of the reference, oracle/) timed on
                the host cores of this box on a bounded sample of the same workload (rank 0, N=1)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3     # fp32-input MFMA peak
D, INNER_PER_HEAD, MLP = 96, 64, 64


def flops_model(S, N, depth, heads):
    """algorithmic forward FLOPs per sample: per token and block
    2*D*3I (qkv) + 2*I*D (out) + 4*D*M (mlp) + 4*I*Lseq (QK^T and PV)."""
    I = heads * INNER_PER_HEAD
    T = S * N
    lin = 6 * D * I + 2 * I * D + 4 * D * MLP
    attn_lin = 6 * D * I + 2 * I * D
    per = {
        "block_fwd": lambda Lseq: lin + 4 * I * Lseq,
        "block_bwd_attn": lambda Lseq: 2 * (attn_lin + 4 * I * Lseq),
        "block_bwd_mlp": lambda Lseq: 2 * (4 * D * MLP),
    }
    fwd = T * depth * (2 * lin + 4 * I * (N + S))
    return fwd, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--bands", type=int, default=200)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="transformer dropout in training mode (reference configs/config.yaml:23-24 ships 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-all", action="store_true",
                    help="event pairs around every kernel in the timed region (default: only the dominant kernel, "
                         "found during warmup; ~150 event pairs per step cost ~5 %% of the step)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from maskedsst_amd import ViTSpatialSpectral, SimMIMSpatialSpectral, _lib
    from maskedsst_amd.optim import FusedAdamW, attach_data_parallel

    SEED = 5
    import random
    random.seed(SEED); np.random.seed(SEED); torch.manual_seed(SEED)
    S = args.bands // 10
    enc = ViTSpatialSpectral(
        image_size=8, spatial_patch_size=1, spectral_patch_size=10, num_classes=8, dim=96, depth=args.depth,
        heads=args.heads, mlp_dim=64, dropout=args.dropout, emb_dropout=args.dropout, channels=args.bands,
        spectral_pos_embed=False,
        spectral_pos=torch.arange(S), blockwise_patch_embed=True, spectral_only=False, precision=args.precision)
    model = SimMIMSpatialSpectral(encoder=enc, masking_ratio=0.7, mask_patch_size=4, tube_masking=True,
                                  to_pixels_per_spectral_block=True).to(dev)
    model.train()
    opt = FusedAdamW(model, lr=0.008, weight_decay=0.05, grad_clamp=1.0)
    reducer = attach_data_parallel(model) if dist.is_initialized() else None

    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(SEED + rank)
    img = torch.randn(B, args.bands, 8, 8, generator=g).to(dev)   # synthetic cubes, resident in HBM

    def step():
        opt.zero_grad()
        loss = model(img)
        loss.backward()
        if reducer is not None:
            opt.grad_scale = reducer.finish()
        opt.step()
        return loss

    lib = _lib.load()
    prof = not args.no_profile
    nk = lib.msst_profile_kernels()

    def collect():
        tot = (ctypes.c_double * nk)()
        cnt = (ctypes.c_long * nk)()
        lib.msst_profile_collect(tot, cnt)
        lib.msst_profile_enable(0)
        return {lib.msst_profile_name(i).decode(): dict(id=i, total_ms=tot[i], launches=cnt[i], avg_us=1e3 * tot[i] / cnt[i])
                for i in range(nk) if cnt[i]}

    # warmup doubles as the survey pass: every kernel is timed there, the timed region then carries event
    # pairs only around the dominant MFMA kernel (the one the roofline object is about)
    lib.msst_profile_select(ctypes.c_ulonglong(~0 & (2 ** 64 - 1)))
    if prof and args.warmup > 0:
        lib.msst_profile_enable(1)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    survey = collect() if (prof and args.warmup > 0) else {}
    mfma_kernels = ("block_fwd", "block_bwd_mlp", "block_bwd_attn")
    if prof and not args.profile_all:
        cand = {k: v for k, v in survey.items() if k in mfma_kernels}
        dom_name = max(cand, key=lambda k: cand[k]["total_ms"]) if cand else "block_bwd_attn"
        dom_id = [i for i in range(nk) if lib.msst_profile_name(i).decode() == dom_name][0]
        lib.msst_profile_select(ctypes.c_ulonglong(1 << dom_id))
    if world > 1:
        dist.barrier()
    if prof:
        lib.msst_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernels = collect() if prof else {}
    lib.msst_profile_select(ctypes.c_ulonglong(2 ** 64 - 1))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        N = 64
        fwd_flops, per = flops_model(S, N, args.depth, args.heads)
        samples = B * world * args.steps
        value = samples / elapsed
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": "pretrain samples/sec", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"SimMIM pretrain step (fwd+bwd+AdamW), EnMAP-shape cubes 8x8x{args.bands}, "
                                   f"depth {args.depth}x2, dim 96, heads {args.heads}, mlp 64, mask 0.7/4/tube, dropout {args.dropout}",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            "step_mfma_frac": round(value / world * 3 * fwd_flops / (peak * 1e12), 4),
            "gflop_per_sample_step": round(3 * fwd_flops / 1e9, 3),
            "final_loss": final_loss,
        }
        if kernels:
            # dominant kernel among the MFMA kernels; algorithmic FLOPs per launch = tokens * per-token FLOPs
            # (half the launches are spatial blocks, Lseq = N; half spectral, Lseq = S -> use the mean)
            cand = {k: v for k, v in kernels.items() if k in per}
            dom = max(cand, key=lambda k: cand[k]["total_ms"])
            ntok = B * S * N
            fl = ntok * 0.5 * (per[dom](N) + per[dom](S))
            avg_s = cand[dom]["avg_us"] * 1e-6
            achieved = fl / avg_s / 1e12
            traffic = None   # HBM bytes per launch from the committed PMC passes of this same command (profiles/)
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                if B == 256 and args.bands == 200 and args.depth == 12 and args.precision == "bf16":
                    traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                               "avg_launch_us": round(cand[dom]["avg_us"], 2),
                               "algorithmic_gflop_per_launch": round(fl / 1e9, 3)}
            out["kernels"] = {k: {"avg_us": round(v["avg_us"], 2), "launches": v["launches"],
                                  "share": round(v["total_ms"] / (1e3 * elapsed), 4)} for k, v in kernels.items()}
            if survey and not args.profile_all:   # untimed warmup steps, every kernel bracketed
                out["warmup_survey"] = {k: {"avg_us": round(v["avg_us"], 2), "launches": v["launches"]}
                                        for k, v in survey.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def cpu_baseline(args):
    """The CPU oracle (oracle/: plain-PyTorch fp32 restatement of the reference, pinned to the
    reference by tests/golden) doing the same training step (fwd + autograd bwd + torch AdamW) on
    this box's host cores, on a bounded sample (a few small batches)."""
    from oracle import OracleConfig, init_params, simmim_forward
    torch.manual_seed(5); np.random.seed(5)
    cfg = OracleConfig(bands=args.bands, depth=args.depth, heads=args.heads)
    params = init_params(cfg)
    for p in params.values():
        p.requires_grad_(True)
    opt = torch.optim.AdamW([p for p in params.values()], lr=0.008, weight_decay=0.05)
    Bc = args.cpu_batch
    x = torch.randn(Bc, args.bands, 8, 8)
    threads = torch.get_num_threads()

    def one():
        opt.zero_grad()
        out = simmim_forward(params, x, cfg)
        out["loss"].backward()
        for p in params.values():
            if p.grad is not None:
                p.grad.clamp_(-1, 1)
        opt.step()

    one()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > 12.0 or n >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * Bc / dt, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of batch {Bc} (same model/config, fwd+bwd+AdamW, torch {torch.__version__} CPU, "
                      f"{threads} threads of {os.cpu_count()} logical cores)"}


if __name__ == "__main__":
    main()
