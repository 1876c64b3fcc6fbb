"""Flat-arena SGD + gradient-norm clipping for the CSTP pre-training step.

Semantics follow torch.optim.SGD(momentum, weight_decay; dampening 0, no nesterov) and
torch.nn.utils.clip_grad_norm_(params, 18) exactly as main_byol.py:86-91,228-232 use them --
including weight decay on BN gamma/beta and biases -- but the work is three streaming kernels
over the model's flat parameter/gradient arenas instead of ~170 x 3 tiny launches:
    sumsq(grad arena) -> clip coefficient (stays on the device, no host sync) -> fused
    scale + weight-decay + momentum + update.
"""
from __future__ import annotations

import torch

from . import ops


class FlatSGD(torch.optim.Optimizer):
    """Drop-in for ``optim.SGD(parameters, lr, momentum, weight_decay)`` on a model whose
    parameters were re-homed by ``R21DBYOL.flatten_parameters()``."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, arenas=None):
        if arenas is None:
            raise ValueError("FlatSGD needs arenas=model.flatten_parameters()")
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._p, self._g = arenas["param"], arenas["grad"]
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        if n != self._p.numel():
            raise ValueError("parameter list does not match the flat arena (%d vs %d floats)" % (n, self._p.numel()))
        dev = self._p.device
        self._buf = torch.zeros_like(self._p)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lr_host = None
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._coef = torch.ones(1, dtype=torch.float32, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._clip_pending = False
        self._steps = 0

    # main_byol.py:86 -- gradients are arena views, so "zero" (not set-to-None) keeps them in place
    def zero_grad(self, set_to_none: bool = False):
        self._g.zero_()

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """clip_grad_norm_(model.parameters(), max_norm): returns the total norm as a DEVICE scalar.
        The scaling itself is folded into the next step() (which also writes the scaled gradients
        back, so .grad holds what the reference's in-place clip leaves there)."""
        ops.grad_sumsq(self._g, self._sumsq)
        ops.clip_coef(self._sumsq, max_norm, self._coef, self._norm)
        self._clip_pending = True
        return self._norm

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        lr = float(g["lr"])
        if lr != self._lr_host:
            self._lr_dev.fill_(lr)
            self._lr_host = lr
        ops.sgd_step_(self._p, self._g, self._buf, self._lr_dev, g["momentum"], g["weight_decay"],
                      self._coef if self._clip_pending else None, self._steps == 0, True)
        self._clip_pending = False
        self._steps += 1

    # checkpoint wire format of torch.optim.SGD (main_byol.py:132-140 saves optimizer.state_dict())
    def state_dict(self):
        sd = super().state_dict()
        state, off = {}, 0
        for i, p in enumerate(self.param_groups[0]["params"]):
            if self._steps > 0:
                state[i] = {"momentum_buffer": self._buf[off:off + p.numel()].view_as(p).clone()}
            off += (p.numel() + 3) // 4 * 4
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        off = 0
        loaded = False
        for i, p in enumerate(self.param_groups[0]["params"]):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None and st.get("momentum_buffer") is not None:
                self._buf[off:off + p.numel()].view_as(p).copy_(st["momentum_buffer"])
                loaded = True
            off += (p.numel() + 3) // 4 * 4
        self._steps = 1 if loaded else 0
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "momentum", "weight_decay"):
                if k in sg:
                    g[k] = sg[k]
