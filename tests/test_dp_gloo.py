"""Data-parallel path on CPU: 2 processes over gloo drive the SAME code the GPU path uses --
``attach_data_parallel`` + ``BucketReducer`` + global-mask slicing + the ENGINE's own backward orchestration
(``head_bwd`` -> ``blocks_bwd`` -> ``tokenize_bwd``, which fire the bucket hooks in their real order).  Only the
C-ABI library is replaced by a stand-in whose ``msst_*_bwd`` entry points fill the flat gradient buffer with the
oracle's gradients of the rank's shard (there is no GPU here); the reduced result must equal the single-process
oracle gradient on the global batch, and every element must have been all-reduced exactly once.
A second leg checks that the plateau scheduler sees the same validation loss on every rank (``dp_mean``)."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import oracle_cfg_from, seed_all
from util import build_product

CFG = dict(bands=30, depth=2, B=4, heads=2)


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


class StandInLib:
    """What the engine calls during a backward, without a GPU: each ``msst_*_bwd`` writes the oracle gradients of the
    parameters that kernel is responsible for into the flat gradient buffer (the real kernels write the same slots)."""

    def __init__(self, eng, grads_by_flat_name):
        self.eng, self.grads = eng, grads_by_flat_name
        self.block_calls = []
        self.filled = {}
        self.chain_calls = 0
        self.layers = eng._layers()
        self.next_block = len(self.layers) - 1

    def _fill(self, prefix):
        fp = self.eng.fp
        for n in fp.segments:
            if n.startswith(prefix) and n in self.grads:
                fp.view(n, fp.grad).copy_(self.grads[n])

    def msst_head_bwd(self, *a):
        self._fill("to_pixels.")
        return 0

    ATTN = ("ln1_g", "ln1_b", "wqkv", "wout", "bo")          # what the attention half + LN1 backward of a block write
    MLP = ("ln2_g", "ln2_b", "w1", "b1", "w2", "b2")          # ... and its MLP half

    def _fill_half(self, i, names):
        sname, l = self.layers[i]
        for n in names:
            self._fill(f"{sname}.{l}.{n}")
        self.filled.setdefault(f"{sname}.{l}", set()).add("attn" if names is self.ATTN else "mlp")

    def msst_block_bwd(self, *a):
        sname, l = self.layers[self.next_block]   # the engine walks the blocks in reverse
        self._fill_half(self.next_block, self.ATTN)
        self._fill_half(self.next_block, self.MLP)
        self.next_block -= 1
        self.block_calls.append(f"{sname}.{l}")
        return 0

    def msst_block_bwd_chain(self, *a):
        """include/msst.h: the call for block i runs [its MLP half when `first`] -> its attention half -> its LN1 backward FUSED with
        the MLP half of block i - 1 (w_prev): block i's gradients are complete when this call returns, the MLP-half gradients
        of block i - 1 are written one call EARLY.  This is the orchestration data parallel actually uses (bf16)."""
        (w, g, w_prev, g_prev, x, x1, x1_prev, dy, dx, dx1, part, slab, grid_rows, nchunk, mode, B, S, N, H, prec, p, seed,
         layer, xn, lse, dab, first, queue, stream) = a
        assert layer == self.next_block, (layer, self.next_block)
        has_prev = bool(ctypes.cast(w_prev, ctypes.c_void_p).value)
        assert has_prev == (layer > 0) and bool(first) == (layer == len(self.layers) - 1)
        assert bool(queue.value) == bool(self.eng.tile_queue), "data parallel: the chained calls must carry the tile queue"
        assert xn.value and dab.value
        if first:
            self._fill_half(layer, self.MLP)
        self._fill_half(layer, self.ATTN)
        if has_prev:
            self._fill_half(layer - 1, self.MLP)
        sname, l = self.layers[layer]
        self.next_block -= 1
        self.block_calls.append(f"{sname}.{l}")
        self.chain_calls += 1
        return 0

    def msst_tokenize_bwd(self, *a):
        for pre in ("embed.", "pre_", "post_", "pos_", "channel_embed", "mask_token"):
            self._fill(pre)
        return 0

    def msst_last_error(self):
        return b""


def _worker(rank, world, port, q, precision="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import maskedsst_amd.engine as engine_mod
        from maskedsst_amd.optim import attach_data_parallel, dp_mean
        model, params, x = build_product(CFG, precision=precision)   # identical on every rank (same seed)
        ocfg = oracle_cfg_from(CFG)
        eng = model.engine()
        eng._require_cuda = lambda t: None               # CPU tensors stand in for device memory in this test only
        engine_mod._stream = lambda: ctypes.c_void_p(0)
        red = attach_data_parallel(model, bucket_bytes=64 << 10)   # the product's own wiring: hook + rank + world
        assert (model.dp_rank, model.dp_world) == (rank, world)
        chained = precision == "bf16"
        # fp32 has no queued kernels: attach_data_parallel must keep the static partition with reserved CUs there
        assert eng.tile_queue == chained and eng.queue_capable() == chained
        sent = []
        flush = red._flush

        def counting_flush():
            if red.pending is not None:
                sent.append(red.pending)
            flush()
        red._flush = counting_flush

        b = CFG["B"] // world
        seed_all(21)
        masks = model.draw_masks(b)                     # global draw, local rows
        xs = x[rank * b:(rank + 1) * b]
        loss, grads = _oracle_grads(params, xs, ocfg, masks)
        key_of = {id(p): k for k, p in model.named_parameters()}
        flat_names = {n: key_of[id(p)] for n, p in eng.trainable()}
        by_flat = {n: grads[k] for n, k in flat_names.items() if grads[k] is not None}
        lib = StandInLib(eng, by_flat)
        eng.lib = lib
        # the backward exactly as _SimMIMLossFn.backward drives it (engine.py), on placeholder activations
        T, K = eng.S * eng.N, masks[1].shape[1]
        acts = [torch.zeros(b, T, 96) for _ in range(2 * CFG["depth"] + 1)]
        x1s = [torch.zeros(b, T, 96) for _ in range(2 * CFG["depth"])]
        if chained:   # the bf16 forward leaves LN1(x) rows on the x1 tensors: what makes blocks_bwd take the chained path
            for t in x1s:
                t._msst_xn = torch.zeros(b, T, 96, dtype=torch.bfloat16)
        # a block's bucket may only be announced when BOTH halves of its gradients have been written
        announce = eng.bucket_hook

        def checked_hook(name, start, end):
            if name.startswith(("spatial.", "spectral.")):
                assert lib.filled.get(name) == {"attn", "mlp"}, (name, lib.filled.get(name))
            announce(name, start, end)
        eng.bucket_hook = checked_hook
        from maskedsst_amd.masking import inverse_csr
        ptr, pos = inverse_csr(masks[1].numpy(), T)
        dy = eng.head_bwd(acts[-1], torch.zeros(b, K, eng.P), torch.from_numpy(ptr), torch.from_numpy(pos))
        dx0 = eng.blocks_bwd(acts, x1s, dy)
        eng.tokenize_bwd(xs, masks[0].to(torch.uint8), dx0)
        scale = red.finish()
        assert abs(scale - 1.0 / world) < 1e-12
        # the engine announced the buckets in backward-completion order, each exactly once
        L = CFG["depth"]
        assert lib.block_calls == [f"spectral.{l}" for l in reversed(range(L))] + [f"spatial.{l}" for l in reversed(range(L))]
        assert lib.chain_calls == (2 * L if chained else 0)
        covered = sorted(sent)
        assert covered[0][0] == 0 and covered[-1][1] == eng.fp.n_trainable
        for (s0, e0), (s1, e1) in zip(covered, covered[1:]):
            assert e0 == s1, ("gap or overlap between all-reduced ranges", covered)
        t = torch.tensor([loss], dtype=torch.float64)
        dist.all_reduce(t)
        # ---- plateau scheduler under DP: every rank must see the same validation loss ----
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5, patience=0)
        local_val = [1.0, 2.0 if rank == 0 else 0.5, 1.4 if rank == 0 else 1.5]   # rank 1 alone would see an improvement at step 2
        for v in local_val:
            sch.step(dp_mean(torch.tensor(v)).item())
        lrs = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(lrs, torch.tensor([opt.param_groups[0]["lr"]], dtype=torch.float64))
        assert all(float(l) == float(lrs[0]) for l in lrs), lrs
        if rank == 0:
            q.put((t.item() / world, (eng.fp.grad[: eng.fp.n_trainable] * scale).numpy().copy(), dict(flat_names),
                   {n: eng.fp.segments[n] for n in flat_names}, len(sent), float(lrs[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("precision", ["fp32", "bf16"], ids=["unchained-fp32", "chained-bf16-tile-queue"])
def test_two_rank_gloo_matches_single_process(precision):
    """fp32: msst_block_bwd per block (static grids, reserved CUs).  bf16: the orchestration data parallel really uses --
    msst_block_bwd_chain with the tile queue, MLP-half gradients of block i - 1 written by the call for block i."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, precision)) for r in range(2)]
    for p in procs:
        p.start()
    loss_dp, flat, names, segs, nsent, lr = q.get(timeout=240)
    flat = torch.from_numpy(flat)   # sent by value (numpy): the worker may exit before a shared-fd tensor is rebuilt
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nsent >= 2                # 64 KB buckets: the gradient went out in several overlappable pieces
    assert lr == 0.25                # means 1.0, 1.25, 1.45: two plateau cuts on every rank
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


def test_dp_dropout_seeds_differ_by_rank():
    """ranks seed torch identically (pretrain.py:24); the dropout seed mixes the rank in"""
    model, _, _ = build_product(dict(bands=20, depth=1, B=2, heads=2, dropout=0.1))
    model.encoder.dropout_p = 0.1
    model.train()
    eng = model.engine()
    seeds = []
    for rank in (0, 1, 2):
        model.dp_rank = rank
        torch.manual_seed(7)
        seeds.append(eng.dropout_state()[1])
    assert len(set(seeds)) == 3 and all(0 <= s < 2 ** 31 for s in seeds)
