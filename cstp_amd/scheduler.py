"""Per-epoch cosine schedule with linear warm-up and restarts, as main_byol.py:252-269 drives
/root/reference/scheduler/cosine_anneal.py:6-88: the optimiser's lr is forced to ``min_lr`` at
construction (so epoch 1 trains at min_lr), climbs linearly for ``warmup_steps`` epochs to
``max_lr``, then follows a half cosine back to ``min_lr``; on restart ``max_lr`` shrinks by gamma.
Plain Python state machine (no _LRScheduler inheritance needed for this driver)."""
from __future__ import annotations

import math


class CosineAnnealingWarmupRestarts:
    def __init__(self, optimizer, first_cycle_steps, cycle_mult=1.0, max_lr=0.1, min_lr=0.001, warmup_steps=0, gamma=1.0,
                 last_epoch=-1):
        if not warmup_steps < first_cycle_steps:
            raise AssertionError("warmup_steps must be smaller than first_cycle_steps")
        self.optimizer = optimizer
        self.first_cycle_steps, self.cycle_mult = first_cycle_steps, cycle_mult
        self.base_max_lr = self.max_lr = max_lr
        self.min_lr, self.warmup_steps, self.gamma = min_lr, warmup_steps, gamma
        self.cur_cycle_steps = first_cycle_steps
        self.cycle = 0
        self.step_in_cycle = last_epoch
        self.last_epoch = last_epoch
        self.step()                      # what _LRScheduler.__init__ does
        for group in self.optimizer.param_groups:   # ... after which the reference pins lr to min_lr
            group["lr"] = self.min_lr

    def _lr(self):
        s, w = self.step_in_cycle, self.warmup_steps
        if s == -1:
            return self.min_lr
        if s < w:
            return (self.max_lr - self.min_lr) * s / w + self.min_lr
        return self.min_lr + (self.max_lr - self.min_lr) * (1 + math.cos(math.pi * (s - w) / (self.cur_cycle_steps - w))) / 2

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
            self.step_in_cycle += 1
            if self.step_in_cycle >= self.cur_cycle_steps:
                self.cycle += 1
                self.step_in_cycle -= self.cur_cycle_steps
                self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps
        elif epoch >= self.first_cycle_steps:
            if self.cycle_mult == 1.0:
                self.step_in_cycle = epoch % self.first_cycle_steps
                self.cycle = epoch // self.first_cycle_steps
            else:
                n = int(math.log(epoch / self.first_cycle_steps * (self.cycle_mult - 1) + 1, self.cycle_mult))
                self.cycle = n
                self.step_in_cycle = epoch - int(self.first_cycle_steps * (self.cycle_mult ** n - 1) / (self.cycle_mult - 1))
                self.cur_cycle_steps = self.first_cycle_steps * self.cycle_mult ** n
        else:
            self.cur_cycle_steps = self.first_cycle_steps
            self.step_in_cycle = epoch
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        self.last_epoch = math.floor(epoch)
        lr = self._lr()
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class ReduceLROnPlateau:
    """``optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', patience=opts.lr_patience)`` as the fine-tune driver
    uses it (main_ft_mp.py:153, stepped with the epoch's validation loss at :279): factor 0.1, relative threshold 1e-4,
    cooldown 0, min_lr 0, eps 1e-8 -- torch's defaults, restated as a plain state machine so it drives any optimizer
    exposing ``param_groups`` (every group's lr is scaled, so the frozen groups at lr 0.0 stay at 0.0)."""

    def __init__(self, optimizer, mode="min", factor=0.1, patience=10, threshold=1e-4, threshold_mode="rel", cooldown=0,
                 min_lr=0.0, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        if mode not in ("min", "max"):
            raise ValueError("mode " + mode + " is unknown!")
        if threshold_mode not in ("rel", "abs"):
            raise ValueError("threshold mode " + threshold_mode + " is unknown!")
        self.optimizer, self.mode, self.factor, self.patience = optimizer, mode, factor, patience
        self.threshold, self.threshold_mode, self.cooldown, self.eps = threshold, threshold_mode, cooldown, eps
        self.min_lrs = [min_lr] * len(optimizer.param_groups)
        self.best = math.inf if mode == "min" else -math.inf
        self.num_bad_epochs = 0
        self.cooldown_counter = 0
        self.last_epoch = 0

    def _is_better(self, a):
        if self.mode == "min":
            return a < (self.best * (1.0 - self.threshold) if self.threshold_mode == "rel" else self.best - self.threshold)
        return a > (self.best * (1.0 + self.threshold) if self.threshold_mode == "rel" else self.best + self.threshold)

    def step(self, metrics):
        current = float(metrics)
        self.last_epoch += 1
        if self._is_better(current):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            for i, group in enumerate(self.optimizer.param_groups):
                old_lr = float(group["lr"])
                new_lr = max(old_lr * self.factor, self.min_lrs[i])
                if old_lr - new_lr > self.eps:
                    group["lr"] = new_lr
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)
