"""Warm-up + cosine learning-rate schedule with the iterated-learning re-warm-up every `reset_steps`
(reference prototype/lr_scheduler/scheduler.py:7-38, 68-94, 239-255).  Host-side fp64 arithmetic, one value per
parameter group; step(i) sets group['lr'] for iteration i (the solver calls it with the 1-based step)."""
import math

import torch


class CosineLRScheduler:
    def __init__(self, optimizer, max_iter, min_lr, base_lr, warmup_lr, warmup_steps, last_iter=0, reset_steps=0):
        if not isinstance(optimizer, torch.optim.Optimizer):
            raise TypeError("%s is not an Optimizer" % type(optimizer).__name__)
        assert warmup_steps >= 2 or warmup_steps == 0
        if warmup_steps == 0:
            assert base_lr == warmup_lr
        self.optimizer = optimizer
        self.max_iter, self.min_lr = max_iter, min_lr
        self.base_lr, self.warmup_lr, self.warmup_steps, self.reset_steps = base_lr, warmup_lr, warmup_steps, reset_steps
        for group in optimizer.param_groups:
            group.setdefault("initial_lr", group["lr"])
        self.base_lrs = [group["initial_lr"] for group in optimizer.param_groups]
        self.last_iter = last_iter

    def _scale(self):
        """multiplier applied to each group's initial lr at self.last_iter"""
        it, ws = self.last_iter, self.warmup_steps
        ratio = (it - ws) / (self.max_iter - ws)
        cosine = self.min_lr + (self.warmup_lr - self.min_lr) * (1 + math.cos(math.pi * ratio)) / 2
        outer = cosine / self.base_lr
        if ws >= 2:
            ramp = (self.warmup_lr - self.base_lr) / (ws - 1)
            if it < ws:
                return (ramp * (it - 1) + self.base_lr) / self.base_lr
            if self.reset_steps > 0 and it % self.reset_steps < ws:
                return outer * ((ramp * (it % self.reset_steps - 1) + self.base_lr) / self.warmup_lr)
        return outer

    def get_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self, this_iter=None):
        self.last_iter = self.last_iter + 1 if this_iter is None else this_iter
        s = self._scale()
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = s * base


Cosine = CosineLRScheduler
