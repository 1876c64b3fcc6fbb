"""Training steps on the GPU.  ``FineTuneStep``: main_ft_mp.py:199-212.  ``PretrainStep``, one CSTP pre-training step: the sequence of main_byol.py:60-91 --
    model(clip_1, clip_2, o_type) -> 6x CrossEntropy -> loss_weight sum -> zero_grad ->
    backward (DDP all-reduces gradients on RCCL) -> clip_grad_norm_(18) -> SGD step
-- with every scalar left on the device (the reference's seven .item() syncs per step are
deferred to ``StepOutput.to_host()``, called when the driver prints/logs)."""
from __future__ import annotations

import contextlib
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


class PretrainStep:
    """``flat_allreduce`` (default on when the model is DDP-wrapped and its gradients live in the flat arena):
    DDP stays the launch surface and still broadcasts the BN buffers at each forward, but its per-bucket
    gradient reducer is bypassed (``no_sync``) in favour of one all-reduce of the gradient arena after
    backward.  Same result (mean of per-rank gradients), no bucket copies."""

    def __init__(self, model, optimizer, loss_weight, task="loss_com", clip_grad_norm=True, ntxent=None,
                 ntxent_weight=0.0, flat_allreduce=True):
        self.model, self.optimizer, self.task = model, optimizer, task
        self.w = normalise_loss_weight(loss_weight)
        self.clip = bool(clip_grad_norm)
        self.ntxent, self.ntxent_weight = ntxent, float(ntxent_weight)
        self._inner = model.module if hasattr(model, "module") else model
        arenas = getattr(self._inner, "_arenas", None)
        self._flat_grad = arenas["grad"] if (flat_allreduce and arenas is not None and hasattr(model, "no_sync")) else None
        # weight gradients may bypass autograd's accumulation (side stream, ops._Conv3d.backward) when the gradients live in
        # the arena and no DDP reducer hook waits for them
        ops.DIRECT_WGRAD = arenas is not None and (self._flat_grad is not None or not hasattr(model, "no_sync"))

    def __call__(self, clip_1, clip_2, spa, tem, pb, rot_1, rot_2) -> StepOutput:
        sync_ctx = self.model.no_sync() if self._flat_grad is not None else contextlib.nullcontext()
        with sync_ctx:
            out = self._forward_backward(clip_1, clip_2, spa, tem, pb, rot_1, rot_2)
        if self._flat_grad is not None:
            allreduce_mean_(self._flat_grad)
        gnorm = self.optimizer.clip_grad_norm_(CLIP_VALUE) if self.clip else None
        self.optimizer.step()
        out.grad_norm = gnorm
        return out

    def _forward_backward(self, clip_1, clip_2, spa, tem, pb, rot_1, rot_2) -> StepOutput:
        w = self.w
        loss_byol, logits = self.model(clip_1, clip_2, o_type=self.task)
        loss_byol = loss_byol.mean()
        ce = [ops.cross_entropy(logits[0], spa), ops.cross_entropy(logits[1], tem), ops.cross_entropy(logits[2], pb),
              ops.cross_entropy(logits[3], pb), ops.cross_entropy(logits[4], rot_1), ops.cross_entropy(logits[5], rot_2)]
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
        objective.backward()
        return StepOutput(loss_total.detach(), loss_byol.detach(), [c.detach() for c in ce], None,
                          None if nt is None else nt.detach(), [l.detach() for l in logits])


class FineTuneStep:
    """One supervised step of main_ft_mp.py:199-212: ``outputs = model(inputs, o_type=task)`` ->
    ``nn.CrossEntropyLoss()`` -> ``zero_grad`` -> ``backward`` -> ``optimizer.step()`` (no gradient clipping here).
    Returns (loss, outputs) as DEVICE tensors; the caller derives accuracy from ``outputs`` as the reference does.
    Under DDP the per-bucket reducer is bypassed (``no_sync``) and only the trainable runs of the flat gradient arena
    are all-reduced -- for ft_fc that is the 52 K-float classifier instead of the 33 M-float encoder."""

    def __init__(self, model, optimizer, task, flat_allreduce=True):
        if task not in ("ft_fc", "ft_all"):
            raise ValueError("o_type %r: the classifier forward serves 'ft_fc' / 'ft_all' (r21d_byol.py:394)" % (task,))
        self.model, self.optimizer, self.task = model, optimizer, task
        inner = model.module if hasattr(model, "module") else model
        arenas = getattr(inner, "_arenas", None)
        self._flat = flat_allreduce and arenas is not None and hasattr(model, "no_sync") and hasattr(optimizer, "_plan")
        self._g = arenas["grad"] if arenas is not None else None
        ops.DIRECT_WGRAD = arenas is not None and (self._flat or not hasattr(model, "no_sync"))

    def __call__(self, inputs, targets):
        sync_ctx = self.model.no_sync() if self._flat else contextlib.nullcontext()
        with sync_ctx:
            outputs = self.model(inputs, o_type=self.task)
            loss = ops.cross_entropy(outputs, targets)
            self.optimizer.zero_grad()
            loss.backward()
        if self._flat:
            for off, n, _, _ in self.optimizer._plan():
                allreduce_mean_(self._g[off:off + n])
        self.optimizer.step()
        return loss.detach(), outputs.detach()
