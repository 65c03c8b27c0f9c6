"""Data-parallel path on CPU: 2 processes over gloo drive the SAME BucketReducer / mask-slicing code
the GPU path uses; per-rank gradients come from the oracle (the checker), the reduced result must
equal the single-process oracle gradient on the global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import oracle_cfg_from, seed_all
from util import build_product

CFG = dict(bands=30, depth=1, B=4, heads=2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_grads(params, x, ocfg, masks):
    from oracle import simmim_forward
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = simmim_forward(ps, x, ocfg, masks=masks)
    out["loss"].backward()
    return out["loss"].item(), {k: (v.grad if v.grad is not None else None) for k, v in ps.items()}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from maskedsst_amd.flat import FlatParams
        from maskedsst_amd.optim import BucketReducer
        model, params, x = build_product(CFG)           # identical on every rank (same seed)
        ocfg = oracle_cfg_from(CFG)
        fp = FlatParams(model.encoder, model).flatten()  # CPU flat buffers, same layout as on the GPU
        model.dp_rank, model.dp_world = rank, world
        b = CFG["B"] // world
        seed_all(21)
        masks = model.draw_masks(b)                     # global draw, local rows
        xs = x[rank * b:(rank + 1) * b]
        loss, grads = _oracle_grads(params, xs, ocfg, masks)
        key_of = {id(p): k for k, p in model.named_parameters()}
        flat_names = {}
        groups, _ = fp._ordered()
        for bname, g in groups:
            for n, p in g:
                flat_names[n] = key_of[id(p)]
        red = BucketReducer(fp.grad, fp.buckets, bucket_bytes=64 << 10)
        # emulate the backward: fill bucket by bucket in completion order, announcing each
        for bname, g in groups:
            for n, p in g:
                gk = grads[flat_names[n]]
                fp.view(n, fp.grad).copy_(gk if gk is not None else torch.zeros_like(p))
            red.bucket_ready(bname)
        scale = red.finish()
        assert abs(scale - 1.0 / world) < 1e-12
        t = torch.tensor([loss], dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            q.put((t.item() / world, (fp.grad[: fp.n_trainable] * scale).numpy().copy(), dict(flat_names),
                   {n: fp.segments[n] for n in flat_names}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss_dp, flat, names, segs = q.get(timeout=240)
    flat = torch.from_numpy(flat)   # sent by value (numpy): the worker may exit before a shared-fd tensor is rebuilt
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, global batch, same seeds
    model, params, x = build_product(CFG)
    ocfg = oracle_cfg_from(CFG)
    seed_all(21)
    masks = model.draw_masks(CFG["B"])
    loss, grads = _oracle_grads(params, x, ocfg, masks)
    assert abs(loss_dp - loss) <= 1e-6 * abs(loss)
    for n, key in names.items():
        off, num, shape = segs[n]
        g = grads[key]
        got = flat[off:off + num].view(shape)
        if g is None:
            assert float(got.abs().max()) == 0.0
            continue
        assert float((got - g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-12, n


def test_reducer_coalesces_contiguous_buckets():
    """single process: no process group -> world 1 -> reducer is a no-op that returns scale 1"""
    from maskedsst_amd.optim import BucketReducer
    flat = torch.arange(10, dtype=torch.float32)
    red = BucketReducer(flat, [("a", 0, 4), ("b", 4, 10)])
    red.bucket_ready("a")
    red.bucket_ready("b")
    assert red.finish() == 1.0 and torch.equal(flat, torch.arange(10, dtype=torch.float32))
