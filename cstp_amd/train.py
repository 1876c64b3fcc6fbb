"""Training steps on the GPU.  ``FineTuneStep``: main_ft_mp.py:199-212.  ``PretrainStep``, one CSTP pre-training step: the sequence of main_byol.py:60-91 --
    model(clip_1, clip_2, o_type) -> 6x CrossEntropy -> loss_weight sum -> zero_grad ->
    backward (DDP all-reduces gradients on RCCL) -> clip_grad_norm_(18) -> SGD step
-- with every scalar left on the device (the reference's seven .item() syncs per step are
deferred to ``StepOutput.to_host()``, called when the driver prints/logs)."""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from . import ops

CLIP_VALUE = 18  # main_byol.py:89


@dataclass
class StepOutput:
    loss_total: torch.Tensor
    loss_byol: torch.Tensor
    ce: Sequence[torch.Tensor]          # spa, tem, pb_1, pb_2, rot_1, rot_2
    grad_norm: Optional[torch.Tensor]
    ntxent: Optional[torch.Tensor]
    logits: Sequence[torch.Tensor]

    def to_host(self):
        vals = torch.stack([self.loss_total.detach(), self.loss_byol.detach()] + [c.detach() for c in self.ce]).tolist()
        return {"loss": vals[0], "loss_byol": vals[1], "loss_pred_spa": vals[2], "loss_pred_tem": vals[3],
                "loss_pred_pb": (vals[4] + vals[5]) / 2, "loss_pred_rot": (vals[6] + vals[7]) / 2}


class LaggedScalars:
    """The per-iteration log scalars of main_byol.py:75-84 without a host sync on the step's critical path.

    The reference all-reduces the total loss and calls ``.item()`` seven times per iteration.  Here ``push`` enqueues,
    behind step i, one all-reduce (mean) of the total loss and ONE device->pinned-host copy of the eight scalars, records
    an event, and returns the record of step i-1, whose event completed long ago: the driver prints iteration i-1 while
    the GPU runs iteration i+1's kernels.  ``flush`` returns the last record at the end of the epoch."""

    KEYS = ("loss", "loss_byol", "loss_pred_spa", "loss_pred_tem", "loss_pred_pb", "loss_pred_rot")

    def __init__(self, device, world_size=1):
        self.world = int(world_size)
        self.device = device
        self._slots = [torch.empty(8, dtype=torch.float32).pin_memory() if torch.device(device).type == "cuda"
                       else torch.empty(8, dtype=torch.float32) for _ in range(2)]
        self._pending = None          # (slot index, event or None, tag)
        self._n = 0

    def _read(self, pend):
        slot, ev, tag = pend
        if ev is not None:
            ev.synchronize()
        v = self._slots[slot].tolist()
        return tag, {"loss": v[0], "loss_byol": v[1], "loss_pred_spa": v[2], "loss_pred_tem": v[3],
                     "loss_pred_pb": (v[4] + v[5]) / 2, "loss_pred_rot": (v[6] + v[7]) / 2}

    def push(self, out: "StepOutput", tag=None):
        prev = self._read(self._pending) if self._pending is not None else None
        total = out.loss_total.detach().reshape(1).clone()
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            dist.all_reduce(total, op=dist.ReduceOp.SUM)          # main_byol.py:22-26 reduce_mean
            total = total / self.world
        vals = torch.cat([total, out.loss_byol.detach().reshape(1)] + [c.detach().reshape(1) for c in out.ce])
        slot = self._n & 1
        self._n += 1
        self._slots[slot].copy_(vals, non_blocking=True)
        ev = None
        if vals.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._pending = (slot, ev, tag)
        return prev

    def flush(self):
        prev = self._read(self._pending) if self._pending is not None else None
        self._pending = None
        return prev


def normalise_loss_weight(w):
    if isinstance(w, (int, float)):
        w = [float(w)]
    w = list(w)
    if len(w) != 5:
        raise ValueError("--loss_weight needs 5 values (byol, spa, tem, pb, rot) for task loss_com, got %r" % (w,))
    return w


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks of one flat tensor (the whole gradient arena): a single RCCL all-reduce over
    xGMI instead of DDP's per-bucket copies (grad -> bucket -> grad costs ~4.7 ms/step at R18 on one MI355X,
    more than the collective itself).  SUM then scale: gloo has no AVG."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / dist.get_world_size())
    return flat


class StagedAllReduce:
    """The gradient all-reduce of DistributedDataParallel (models/model.py:97-103: per-bucket reduction OVERLAPPED with the
    backward pass) on the flat gradient arena: the arena is reduced in the slices the backward pass completes --
    [projector | predictor | heads], conv5, conv4, conv3, conv2, stem (ByolBase.grad_stage_slices) -- each started from an
    autograd hook the moment its gradients are enqueued, on the weight-gradient side stream (which is made to wait for the
    main stream's gradients of that slice), so the main stream's data-gradient chain never waits for a collective and
    only the last slice (the stem: 0.04 % of the arena) is reduced after backward has finished.  ``finish()`` reduces
    whatever has not been started (everything, if no hook fired), waits for all slices and leaves the MEAN over ranks in
    the arena.  On two ranks the result equals the one-piece reduce bit for bit (a + b in either order); on more ranks a
    ring's summation order depends on the element's position in the message, as it does between DDP's buckets."""

    def __init__(self, model, flat_grad, staged=None):
        inner = model.module if hasattr(model, "module") else model
        # CSTP_STAGED_ALLREDUCE=0 selects the one-piece reduce after backward (round-3 ADVICE: kept selectable until the staged
        # path has been verified bit-equal to it on RCCL with two ranks; gloo: tests/test_dist_gloo.py)
        if staged is None:
            staged = os.environ.get("CSTP_STAGED_ALLREDUCE", "1") != "0"
        slices = inner.grad_stage_slices() if (staged and hasattr(inner, "grad_stage_slices")) else None
        self.flat = flat_grad
        self.slices = slices or [(0, flat_grad.numel())]
        self.inner = inner if slices is not None else None
        self._works, self._next, self._armed = [], 0, False

    @staticmethod
    def active() -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def begin(self):
        """Arm the reducer for ONE backward pass: the model's stage callback points at this object only between begin() and
        finish()/abort(), so a backward outside a step (a second step object on the same model, a debug or evaluation
        backward, a step that raised half-way) queues no collective that nobody waits for."""
        self._drain()
        self._works, self._next, self._armed = [], 0, True
        if self.inner is not None:
            self.inner._grad_stage_cb = self.stage_done

    def _disarm(self):
        self._armed = False
        if self.inner is not None and getattr(self.inner, "_grad_stage_cb", None) == self.stage_done:
            self.inner._grad_stage_cb = None

    def _drain(self):
        for w in self._works:
            w.wait()
        self._works = []

    def abort(self):
        """The step failed between begin() and finish(): stop listening and wait for what was already issued."""
        self._disarm()
        self._drain()

    def _reduce(self, k):
        off, n = self.slices[k]
        t = self.flat[off:off + n]
        if t.is_cuda:
            # the slice's gradients were written on the main stream (data-gradient chain, BatchNorm) and on the weight-gradient
            # side stream: the collective is issued from the side stream once that stream has also seen the main stream's work
            main = torch.cuda.current_stream(t.device)
            side = ops._side_stream(t.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

    def stage_done(self, i):
        """autograd hook: stages 0..i are complete (a later stage's hook may fire first when an earlier one had no hook).
        A no-op when the reducer is not armed, and a stage index never moves the cursor backwards."""
        if not self._armed or not self.active():
            return
        while self._next <= i and self._next < len(self.slices) - 1:
            self._reduce(self._next)
            self._next += 1

    def finish(self):
        self._disarm()
        if not self.active():
            return self.flat
        while self._next < len(self.slices):
            self._reduce(self._next)
            self._next += 1
        self._drain()                     # the current stream waits for every collective
        self.flat.mul_(1.0 / dist.get_world_size())
        return self.flat


def sync_buffers(model) -> bool:
    """Rank 0's BN running statistics / counters -> every rank: what DistributedDataParallel(broadcast_buffers=True), the
    reference's wrap (models/model.py:97-103), does at the start of EVERY forward, train or eval.  The steps below call
    it at the top of each step; the drivers call it before validation and before a checkpoint is written."""
    inner = model.module if hasattr(model, "module") else model
    fn = getattr(inner, "broadcast_buffers_", None)
    return bool(fn()) if fn is not None else False


class PretrainStep:
    """``flat_allreduce`` (default on when the model is DDP-wrapped and its gradients live in the flat arena):
    DDP stays the launch surface, but its per-bucket gradient reducer is bypassed (``no_sync``) in favour of one
    all-reduce of the gradient arena after backward -- same result (mean of per-rank gradients), no bucket copies.
    Under ``no_sync`` DDP also stops broadcasting its buffers after the first forward (``require_forward_param_sync``
    goes False), so the step issues that broadcast itself: rank 0's BN buffers overwrite the other ranks' at the top
    of every step (``sync_buffers``: two collectives over the flat buffer arenas), as DDP's default does at each
    forward of the reference.  ``cross_entropy`` is injectable so that the control flow can be driven on CPU
    (tests/test_dist_gloo.py); the product default is the HIP kernel."""

    def __init__(self, model, optimizer, loss_weight, task="loss_com", clip_grad_norm=True, ntxent=None,
                 ntxent_weight=0.0, flat_allreduce=True, cross_entropy=None):
        self.model, self.optimizer, self.task = model, optimizer, task
        self._ce = cross_entropy if cross_entropy is not None else ops.cross_entropy
        self.w = normalise_loss_weight(loss_weight)
        self.clip = bool(clip_grad_norm)
        self.ntxent, self.ntxent_weight = ntxent, float(ntxent_weight)
        self._inner = model.module if hasattr(model, "module") else model
        arenas = getattr(self._inner, "_arenas", None)
        self._flat_grad = arenas["grad"] if (flat_allreduce and arenas is not None and hasattr(model, "no_sync")) else None
        self._reducer = StagedAllReduce(model, self._flat_grad) if self._flat_grad is not None else None
        # weight gradients may bypass autograd's accumulation (side stream, ops._Conv3d.backward) when the gradients live in
        # the arena and no DDP reducer hook waits for them: a property of THIS step's model (ops.direct_wgrad_params)
        self._direct = arenas is not None and (self._flat_grad is not None or not hasattr(model, "no_sync"))
        ops.mark_direct_grad(self._inner, arenas, self._direct)
        self._tiles_shared = False
        # the weight packs of a step from three launches (ops.PackPlan): models whose parameters live in the flat arenas on a HIP
        # device; CSTP_PACK_PLAN=0 keeps the per-call packs
        self._packs = None
        if arenas is not None and arenas["param"].is_cuda and os.environ.get("CSTP_PACK_PLAN", "1") != "0":
            tgt = arenas.get("target")
            rng = (tgt.data_ptr(), tgt.data_ptr() + tgt.numel() * tgt.element_size()) if tgt is not None else None
            self._packs = ops.PackPlan(rng)

    def __call__(self, clip_1, clip_2, spa, tem, pb, rot_1, rot_2) -> StepOutput:
        if self._flat_grad is not None:
            sync_buffers(self.model)                  # DDP's per-forward buffer broadcast (see the class docstring)
        sync_ctx = self.model.no_sync() if self._flat_grad is not None else contextlib.nullcontext()
        if self._reducer is not None:
            self._reducer.begin()
        plan = self._packs
        if plan is not None:
            ops.pack_plan = plan
            plan.tick()
            plan.armed = True
            plan.replay("online")          # every online / predictor / head FORWARD weight pack of this step: one launch
            self._pack_side = None
            if plan.state == "replay" and "online_d" in plan.tables and clip_1.is_cuda:
                # ... and their data-gradient packs beside the forward pass, on the weight-gradient side stream
                main = torch.cuda.current_stream(clip_1.device)
                side = ops._side_stream(clip_1.device)
                side.wait_stream(main)     # (the optimizer step that wrote the weights ran on the main stream)
                with torch.cuda.stream(side):
                    plan.replay("online_d")
                self._pack_side = side
        try:
            with sync_ctx:
                out = self._forward_backward(clip_1, clip_2, spa, tem, pb, rot_1, rot_2)
        except BaseException:
            if self._reducer is not None:
                self._reducer.abort()
            if plan is not None:
                plan.armed = False
                plan.invalidate()
            raise
        if plan is not None:
            if plan.state == "record":
                plan.finish_record(clip_1.device)
            plan.armed = False
        if self._reducer is not None:
            self._reducer.finish()
        gnorm = self.optimizer.clip_grad_norm_(CLIP_VALUE) if self.clip else None
        self.optimizer.step()
        out.grad_norm = gnorm
        if not self._tiles_shared:
            # every layer geometry has been seen (and tuned, if the persisted table lacked it) once: all ranks adopt rank
            # 0's tiles so that the same layer runs the same kernel everywhere
            self._tiles_shared = True
            if any(p.is_cuda for p in self._inner.parameters()):      # (CPU / stub models never load the HIP library)
                ops.share_tune_table()
        return out

    def _forward_backward(self, clip_1, clip_2, spa, tem, pb, rot_1, rot_2) -> StepOutput:
        w = self.w
        loss_byol, logits = self.model(clip_1, clip_2, o_type=self.task)
        loss_byol = loss_byol.mean()
        xe = self._ce
        ce = [xe(logits[0], spa), xe(logits[1], tem), xe(logits[2], pb), xe(logits[3], pb), xe(logits[4], rot_1),
              xe(logits[5], rot_2)]
        loss_total = (w[0] * loss_byol + w[1] * ce[0] + w[2] * ce[1] + w[3] * ce[2] + w[3] * ce[3]
                      + w[4] * ce[4] + w[4] * ce[5])
        nt = None
        objective = loss_total
        if self.ntxent is not None and self.ntxent_weight != 0.0:
            z1, z2 = self._inner.last_projections
            nt = self.ntxent(z1, z2)
            # every rank holds the same global loss; DDP will average the per-rank gradients
            objective = loss_total + (self.ntxent_weight * self.ntxent.ddp_scale) * nt
        self.optimizer.zero_grad()
        if getattr(self, "_pack_side", None) is not None:      # the data-gradient weight packs (ops.PackPlan) must have landed
            torch.cuda.current_stream(clip_1.device).wait_stream(self._pack_side)
            self._pack_side = None
        objective.backward()
        return StepOutput(loss_total.detach(), loss_byol.detach(), [c.detach() for c in ce], None,
                          None if nt is None else nt.detach(), [l.detach() for l in logits])


class FineTuneStep:
    """One supervised step of main_ft_mp.py:199-212: ``outputs = model(inputs, o_type=task)`` ->
    ``nn.CrossEntropyLoss()`` -> ``zero_grad`` -> ``backward`` -> ``optimizer.step()`` (no gradient clipping here).
    Returns (loss, outputs) as DEVICE tensors; the caller derives accuracy from ``outputs`` as the reference does.
    Under DDP the per-bucket reducer is bypassed (``no_sync``) and only the trainable runs of the flat gradient arena
    are all-reduced -- for ft_fc that is the 52 K-float classifier instead of the 33 M-float encoder -- and the BN
    buffers are broadcast from rank 0 at the top of each step (see PretrainStep)."""

    def __init__(self, model, optimizer, task, flat_allreduce=True, cross_entropy=None):
        self._ce = cross_entropy if cross_entropy is not None else ops.cross_entropy
        if task not in ("ft_fc", "ft_all"):
            raise ValueError("o_type %r: the classifier forward serves 'ft_fc' / 'ft_all' (r21d_byol.py:394)" % (task,))
        self.model, self.optimizer, self.task = model, optimizer, task
        inner = model.module if hasattr(model, "module") else model
        arenas = getattr(inner, "_arenas", None)
        self._flat = flat_allreduce and arenas is not None and hasattr(model, "no_sync") and hasattr(optimizer, "_plan")
        self._g = arenas["grad"] if arenas is not None else None
        ops.mark_direct_grad(inner, arenas, arenas is not None and (self._flat or not hasattr(model, "no_sync")))
        self._tiles_shared = False

    def __call__(self, inputs, targets):
        if self._flat:
            sync_buffers(self.model)
        sync_ctx = self.model.no_sync() if self._flat else contextlib.nullcontext()
        with sync_ctx:
            outputs = self.model(inputs, o_type=self.task)
            loss = self._ce(outputs, targets)
            self.optimizer.zero_grad()
            loss.backward()
        if self._flat:
            for off, n, _, _ in self.optimizer._plan():
                allreduce_mean_(self._g[off:off + n])
        self.optimizer.step()
        if not self._tiles_shared:            # all ranks adopt rank 0's tuned tiles (see PretrainStep)
            self._tiles_shared = True
            if self._flat and self._g is not None and self._g.is_cuda:
                ops.share_tune_table()
        return loss.detach(), outputs.detach()
