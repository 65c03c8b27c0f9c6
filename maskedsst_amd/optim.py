"""Fused AdamW + data-parallel gradient reduction for maskedsst_amd models.

``FusedAdamW`` is a ``torch.optim.Optimizer`` (so LR schedulers such as the reference's
``ReduceLROnPlateau``, ``src/utils.py:47-50``, drive it unchanged) whose ``step`` is ONE launch of
``msst_adamw`` over the flat parameter buffer instead of ~350 small tensor updates.  It reproduces
``torch.optim.AdamW`` (reference ``src/utils.py:41-44``) including the "parameters without a
gradient are skipped" rule: ``encoder.mlp_head.*`` receives no gradient in pre-training and lies
outside the updated range.  ``grad_clamp`` applies the reference's per-parameter gradient hook
``clamp(grad, -1, 1)`` (``pretrain.py:71-73``) inside the kernel, after the data-parallel mean.

``BucketReducer`` all-reduces slices of the flat gradient buffer as soon as the backward has
completed them (buckets are contiguous because the flat layout follows backward order), on the
process group's own stream (``torch.distributed`` NCCL backend == RCCL on ROCm), overlapping the
remaining backward kernels.  It only needs a flat tensor and a bucket list, so the gloo tests
exercise the same code on CPU tensors.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def dp_mean(value, group=None):
    """Mean of a scalar tensor over the data-parallel ranks (identity in a single process).  Used for every number
    that steers replicated state -- e.g. the validation loss fed to ReduceLROnPlateau (reference pretrain.py:194-197):
    the all-reduce of the gradients keeps the WEIGHTS equal only while every rank applies the same learning rate."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    v = value.detach().clone().float().reshape(1)
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return (v / dist.get_world_size(group)).reshape(())


class BucketReducer:
    def __init__(self, flat_grad, buckets, group=None, bucket_bytes=4 << 20, average_in_optimizer=True):
        """buckets: [(name, start, end)] in the order the backward completes them."""
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # MSST_FORCE_DP=1: issue the collectives even with a single rank (exercises the RCCL path on a 1-GPU box)
        self.force = dist.is_initialized() and os.environ.get("MSST_FORCE_DP", "0") == "1"
        self.bucket_bytes = bucket_bytes
        self.average_in_optimizer = average_in_optimizer
        self.order = [b[0] for b in buckets]
        self.range = {b[0]: (b[1], b[2]) for b in buckets}
        self.reset()

    def reset(self):
        self.handles = []
        self.pending = None  # (start, end) of completed-but-unsent contiguous range
        self.done = set()

    def bucket_ready(self, name, start=None, end=None):
        """called (in backward order) when a bucket's gradients are final"""
        if self.world == 1 and not self.force:
            return
        s, e = self.range[name]
        self.done.add(name)
        if self.pending is None:
            self.pending = (s, e)
        elif self.pending[1] == s:
            self.pending = (self.pending[0], e)
        else:  # non-contiguous: flush what we have, start a new run
            self._flush()
            self.pending = (s, e)
        # the LAST bucket of the backward (the tokenizer's) goes out from its own hook: left to finish(), its all-reduce would
        # start only when the host gets there and sit fully exposed in front of the optimizer step
        if (self.pending[1] - self.pending[0]) * self.flat.element_size() >= self.bucket_bytes or name == self.order[-1]:
            self._flush()

    def _flush(self):
        if self.pending is None:
            return
        s, e = self.pending
        self.pending = None
        if e > s:
            h = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append(h)

    def finish(self):
        """flush the tail and wait for every outstanding all-reduce; returns the scale still to be
        applied to the gradients (1/world when the optimizer does the averaging)."""
        if self.world > 1 or self.force:
            self._flush()
            for h in self.handles:
                h.wait()
            if not self.average_in_optimizer:
                lo = min(self.range[n][0] for n in self.order)
                hi = max(self.range[n][1] for n in self.order)
                self.flat[lo:hi].mul_(1.0 / self.world)
        self.handles = []
        self.done = set()
        return (1.0 / self.world) if self.average_in_optimizer else 1.0


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_clamp=0.0):
        params = [p for p in model.parameters()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.grad_clamp = float(grad_clamp)
        self.grad_scale = 1.0
        self._m = None
        self._v = None
        self._step = 0
        self._flat_id = None
        self._checked_flat = None

    def _state(self, eng):
        eng.ensure()
        flat = eng.fp.flat
        if self._m is None or self._flat_id != flat.data_ptr():
            # (re)allocate moments; a re-flatten (e.g. after .to()) carries the old moments over
            m = torch.zeros_like(flat)
            v = torch.zeros_like(flat)
            if self._m is not None and self._m.numel() == m.numel():
                m.copy_(self._m)
                v.copy_(self._v)
            self._m, self._v = m, v
            self._flat_id = flat.data_ptr()
        return flat, eng.fp.grad, self._m, self._v

    def zero_grad(self, set_to_none=True):
        # The backward overwrites the flat gradient buffer and hands autograd VIEWS of it, so the
        # .grad attributes must be dropped (not zeroed, not kept): a surviving .grad would make
        # autograd accumulate a view onto itself.
        return super().zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        eng = self.model.engine()
        flat, grad, m, v = self._state(eng)
        # The launch updates the whole trainable prefix of the flat buffer.  torch.optim.AdamW skips parameters
        # without a gradient; a frozen (requires_grad=False) or unused parameter inside the prefix would instead
        # receive weight decay plus whatever the gradient buffer last held -- refuse rather than diverge silently.
        # (attribute reads only on the per-step path; the pointer-range check runs on the first step and after a re-flatten)
        named = eng.trainable()
        full = self._checked_flat != grad.data_ptr()
        lo, hi = grad.data_ptr(), grad.data_ptr() + 4 * grad.numel()
        for name, p in named:
            if not p.requires_grad:
                raise RuntimeError(f"FusedAdamW updates every trainable parameter in one launch; {name!r} has "
                                   "requires_grad=False -- use torch.optim.AdamW for partially frozen models")
            if p.grad is None or (full and not (lo <= p.grad.data_ptr() < hi)):
                raise RuntimeError(f"FusedAdamW.step(): parameter {name!r} has no gradient from the HIP backward of "
                                   "this step (call loss.backward() first; gradients must be the flat-buffer views)")
        self._checked_flat = grad.data_ptr()
        # cheap per-step guard on one sentinel per end of the buffer: a later step whose .grad is not a flat-buffer view any
        # more (e.g. hooks that replace gradients) must not be applied from stale buffer contents
        for name, p in (named[0], named[-1]):
            if not (lo <= p.grad.data_ptr() < hi):
                raise RuntimeError(f"FusedAdamW.step(): the gradient of {name!r} is no view of the flat gradient buffer")
        g = self.param_groups[0]
        self._step += 1
        n = eng.fp.n_trainable
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        _lib.check(eng.lib.msst_adamw(P(flat), P(grad), P(m), P(v), n, float(g["lr"]), float(g["betas"][0]),
                                      float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step,
                                      self.grad_clamp, float(self.grad_scale), stream), "msst_adamw")
        return None

    def state_dict(self):
        d = super().state_dict()
        # clones: an in-memory snapshot must not keep changing as training continues
        d["fused"] = dict(step=self._step, m=None if self._m is None else self._m.clone(),
                          v=None if self._v is None else self._v.clone())
        return d

    def load_state_dict(self, state_dict):
        """Restores lr / betas / ... through torch and the fused moments + step count saved by ``state_dict``
        (a resumed run continues the bias correction where it stopped instead of restarting at step 0)."""
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused", None)
        # validate everything BEFORE touching any state: a failed load must leave the optimizer as it was
        if fused is None:
            raise KeyError("state dict has no 'fused' entry: it was not produced by FusedAdamW.state_dict()")
        eng = self.model.engine()
        flat, _, m, v = self._state(eng)
        if fused["m"] is not None and (fused["m"].numel() != m.numel() or fused["v"] is None or fused["v"].numel() != v.numel()):
            raise ValueError(f"fused moments have {fused['m'].numel()} elements, the model needs {m.numel()}")
        super().load_state_dict(state_dict)
        self._step = int(fused["step"])
        if fused["m"] is not None:
            m.copy_(fused["m"].to(m.device))
            v.copy_(fused["v"].to(v.device))
        else:
            m.zero_()
            v.zero_()


def attach_data_parallel(model, group=None, bucket_bytes=4 << 20):
    """Make ``model`` (a SimMIMSpatialSpectral) data parallel over ``group``: masks are drawn for
    the global batch, gradient buckets are all-reduced as the backward completes them.  Returns the
    reducer; call ``reducer.finish()`` after ``loss.backward()`` and hand its return value to
    ``optimizer.grad_scale``."""
    eng = model.engine()
    eng.ensure()
    model.dp_rank = dist.get_rank(group) if dist.is_initialized() else 0
    model.dp_world = dist.get_world_size(group) if dist.is_initialized() else 1
    red = BucketReducer(eng.fp.grad, eng.fp.buckets, group=group, bucket_bytes=bucket_bytes)
    eng.bucket_hook = red.bucket_ready
    # RCCL's channel workgroups need CUs of their own while the backward runs.  The backward's persistent grids fill every CU
    # with ONE workgroup each and are statically partitioned: a workgroup that finds no CU runs BEHIND the others and doubles
    # that launch.  Measured with bench.py --cu-thief (N probe workgroups, each holding a whole CU under the backward;
    # profiles/r03_dp_cu_contention.jsonl, two-head attention backward): step 26.46 ms undisturbed; 33.6-33.7 ms (+27 %) with
    # 8, 16 or 32 probes on the full grids; with 16 CUs left free 28.1 ms (+6 %) for 8 / 16 probes but 34.7 ms (+31 %) for 32;
    # with 32 CUs left free 27.6-27.7 ms (+4.5 %) for all three.  Hence 32 (RCCL's default channel count on this node is below
    # that); the r04 sweep (profiles/r04_dp_cu_contention.jsonl) adds the no-probe run on the reserved grids.
    # Round 4: under data parallel the attention backward and the fused LN1 + MLP launch no longer partition their tiles
    # statically but DRAW them (engine.tile_queue -> msst_block_bwd_chain's tile_queue): a workgroup whose CU is held by a
    # channel workgroup simply draws fewer tiles, so nothing has to be reserved (MSST_DP_RESERVE_CUS, default 0 now; the static
    # partition with 32 reserved CUs is MSST_DP_TILE_QUEUE=0 MSST_DP_RESERVE_CUS=32).  Cost: the gradients are no longer
    # bit-reproducible from run to run (fp32 summation order follows the draw order).
    # The queue only exists on the chained bf16 path (Engine.queue_capable): an fp32 run, an odd head count, MSST_DBG kernel
    # selections or MSST_BWD_CHAIN=0 fall back to msst_block_bwd with full static grids -- those keep round 3's reservation.
    if model.dp_world > 1 or os.environ.get("MSST_FORCE_DP", "0") == "1":
        eng.tile_queue = os.environ.get("MSST_DP_TILE_QUEUE", "1") != "0" and eng.queue_capable()
        reserve = int(os.environ.get("MSST_DP_RESERVE_CUS", "0" if eng.tile_queue else "32"))
        if reserve > 0:
            eng.reserve_cus(reserve)
    return red
