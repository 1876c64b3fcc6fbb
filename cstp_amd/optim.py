"""Flat-arena optimizers + gradient-norm clipping for the CSTP training loops.

Semantics follow torch.optim.SGD(momentum, weight_decay; dampening 0, no nesterov), torch.optim.Adam / AdamW
(betas, eps 1e-8, no amsgrad) and torch.nn.utils.clip_grad_norm_(params, 18) exactly as main_byol.py:86-91,228-232 and
main_ft_mp.py:133-147,210-212 use them -- including weight decay on BN gamma/beta and biases, and including the
fine-tune parameter list of one group per tensor with frozen tensors at lr 0.0 / requires_grad False
(r21d_byol.py:10-35), which torch skips because their .grad is None.  The work is streaming kernels over the model's
flat parameter/gradient arenas instead of ~170 x 3 tiny launches:
    sumsq(grad arena) -> clip coefficient (stays on the device, no host sync) -> fused scale + weight-decay +
    momentum + update, one launch per RUN of consecutive trainable tensors that share hyper-parameters
    (pre-training / ft_all / scratch: one run = the whole arena; ft_fc: one run = classify.weight|bias).
"""
from __future__ import annotations

import torch

from . import ops


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class _FlatOptimizer(torch.optim.Optimizer):
    """Shared plumbing: maps every parameter to its offset in the flat arena, groups them into runs."""

    _hyper = ()          # group keys that must match for two tensors to share a launch

    def __init__(self, params, defaults, arenas):
        if arenas is None:
            raise ValueError("%s needs arenas=model.flatten_parameters()" % type(self).__name__)
        super().__init__(params, defaults)
        self._p, self._g = arenas["param"], arenas["grad"]
        base, n = self._p.data_ptr(), self._p.numel()
        self._slots = []   # (group index, offset, padded length, numel, param) in state_dict index order
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                off = (p.data_ptr() - base) // 4
                inside = p.device == self._p.device and off >= 0 and off + p.numel() <= n and (p.data_ptr() - base) % 16 == 0
                if not inside and not p.requires_grad:
                    # a frozen tensor outside the trainable arena (the EMA target network of a pre-training model, which
                    # optim.SGD(model.parameters()) of main_byol.py:228 lists too): it keeps its INDEX in param_groups and
                    # in the state dict -- torch gives it no state, its .grad being None -- and is never touched
                    self._slots.append((gi, -1, 0, p.numel(), p))
                    continue
                if not inside:
                    raise ValueError("parameter of shape %s does not live in the flat arena -- build the optimizer from "
                                     "the parameters of a model whose flatten_parameters() produced `arenas`"
                                     % (tuple(p.shape),))
                self._slots.append((gi, off, _pad4(p.numel()), p.numel(), p))
        self._runs_key, self._runs = None, []
        self._lr_cache = {}
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=self._p.device)
        self._coef = torch.ones(1, dtype=torch.float32, device=self._p.device)
        self._norm = torch.zeros(1, dtype=torch.float32, device=self._p.device)
        self._clip_pending = False

    # main_byol.py:86 / main_ft_mp.py:210 -- gradients are arena views, so "zero" (not set-to-None) keeps them in place
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

    def _lr_tensor(self, lr: float) -> torch.Tensor:
        t = self._lr_cache.get(lr)
        if t is None:
            if len(self._lr_cache) > 64:
                self._lr_cache.clear()
            t = torch.full((1,), lr, dtype=torch.float32, device=self._p.device)
            self._lr_cache[lr] = t
        return t

    def _plan(self):
        """Runs of arena-adjacent trainable tensors with identical hyper-parameters: [(offset, length, group)]."""
        key = tuple(tuple(g[k] for k in self._hyper) for g in self.param_groups) + \
            tuple(s[4].requires_grad for s in self._slots)
        if key == self._runs_key:
            return self._runs
        runs = []
        for gi, off, plen, _, p in sorted(self._slots, key=lambda s: s[1]):
            if not p.requires_grad or off < 0:      # torch: .grad is None -> skipped (no decay, no momentum)
                continue
            g = self.param_groups[gi]
            hp = tuple(g[k] for k in self._hyper)
            if runs and runs[-1][0] + runs[-1][1] == off and runs[-1][3] == hp:
                runs[-1][1] += plen
            else:
                runs.append([off, plen, g, hp])
        self._runs_key, self._runs = key, runs
        return runs


class FlatSGD(_FlatOptimizer):
    """Drop-in for ``optim.SGD(parameters, lr, momentum, weight_decay)`` on a model whose parameters were
    re-homed by ``R21DBYOL.flatten_parameters()``; ``params`` may be ``model.parameters()`` or the per-tensor
    group list of ``get_fine_tuning_parameters``."""

    _hyper = ("lr", "momentum", "weight_decay")

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, arenas=None):
        # frozen tensors stay listed (state-dict indices = torch.optim.SGD(model.parameters())'s); _plan skips them
        params = list(params)
        # the remaining torch.optim.SGD group keys ride along (fixed at what this kernel implements) so that a
        # state_dict written here loads into torch.optim.SGD -- i.e. into the reference -- and steps there
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False,
                                      maximize=False, foreach=None, differentiable=False, fused=None), arenas)
        self._buf = torch.zeros_like(self._p)
        self._steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        for off, n, g, _ in self._plan():
            sl = slice(off, off + n)
            ops.sgd_step_(self._p[sl], self._g[sl], self._buf[sl], self._lr_tensor(float(g["lr"])), g["momentum"],
                          g["weight_decay"], self._coef if self._clip_pending else None, self._steps == 0, True)
        self._clip_pending = False
        self._steps += 1

    # checkpoint wire format of torch.optim.SGD (main_byol.py:132-140 saves optimizer.state_dict())
    def state_dict(self):
        sd = super().state_dict()
        state = {}
        for i, (_, off, _, numel, p) in enumerate(self._slots):
            if self._steps > 0 and p.requires_grad and off >= 0:
                state[i] = {"momentum_buffer": self._buf[off:off + numel].view_as(p).clone()}
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        loaded = False
        for i, (_, off, _, numel, p) in enumerate(self._slots):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None and st.get("momentum_buffer") is not None and off >= 0:
                self._buf[off:off + numel].view_as(p).copy_(st["momentum_buffer"])
                loaded = True
        self._steps = 1 if loaded else 0
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "momentum", "weight_decay"):
                if k in sg:
                    g[k] = sg[k]


class FlatAdam(_FlatOptimizer):
    """``optim.Adam(parameters, lr, weight_decay)`` / ``optim.AdamW(parameters, lr, betas, weight_decay)`` of
    main_ft_mp.py:139-147 (``decoupled=True`` is AdamW)."""

    _hyper = ("lr", "betas", "eps", "weight_decay")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False, arenas=None):
        params = list(params)       # frozen tensors stay listed, as in FlatSGD
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=bool(decoupled)), arenas)
        self.decoupled = bool(decoupled)
        self._m = torch.zeros_like(self._p)
        self._v = torch.zeros_like(self._p)
        self._steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        if self._clip_pending:
            self._g.mul_(self._coef)
            self._clip_pending = False
        self._steps += 1
        for off, n, g, _ in self._plan():
            sl = slice(off, off + n)
            ops.adam_step_(self._p[sl], self._g[sl], self._m[sl], self._v[sl], self._lr_tensor(float(g["lr"])),
                           g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.decoupled, self._steps)

    def state_dict(self):
        sd = super().state_dict()
        state = {}
        for i, (_, off, _, numel, p) in enumerate(self._slots):
            if self._steps > 0 and p.requires_grad and off >= 0:
                state[i] = {"step": torch.tensor(float(self._steps)),
                            "exp_avg": self._m[off:off + numel].view_as(p).clone(),
                            "exp_avg_sq": self._v[off:off + numel].view_as(p).clone()}
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        steps = 0
        for i, (_, off, _, numel, p) in enumerate(self._slots):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None and "exp_avg" in st and off >= 0:
                self._m[off:off + numel].view_as(p).copy_(st["exp_avg"])
                self._v[off:off + numel].view_as(p).copy_(st["exp_avg_sq"])
                steps = max(steps, int(float(st.get("step", 0))))
        self._steps = steps
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]


def build_optimizer(opts, parameters, arenas):
    """main_byol.py:227-244 / main_ft_mp.py:132-147: --optimizer sgd | adam | adamw."""
    if opts.optimizer == "sgd":
        return FlatSGD(parameters, lr=opts.learning_rate, momentum=opts.momentum, weight_decay=opts.weight_decay,
                       arenas=arenas)
    if opts.optimizer == "adamw":
        return FlatAdam(parameters, lr=opts.learning_rate, betas=(0.9, 0.99), weight_decay=opts.weight_decay,
                        decoupled=True, arenas=arenas)
    if opts.optimizer == "adam":
        return FlatAdam(parameters, lr=opts.learning_rate, weight_decay=opts.weight_decay, arenas=arenas)
    raise ValueError("unknown --optimizer %r (sgd / adam / adamw)" % (opts.optimizer,))
