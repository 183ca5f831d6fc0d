"""Drop-in for the part of the reference's `metrics` module that the hot path's Trainer uses: the `Metrics` accumulator
protocol (metrics.py:6-20: name / update / get / reset) and `Dice` (metrics.py:101-133) -- threshold 0.5, per sample and
channel 2*sum(p*g)/sum(p+g) (NaN -> 1), mean over the batch, accumulated over update() calls.  The counting runs on the
device (ru_dice_counts) and the running sum stays there: `update` never synchronises, `get()` copies C-1 numbers to the host.  `update(ground, predict)` keeps the reference's
argument order (train.py:304)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class Metrics(object):
    def __init__(self, name):
        self.name = name
        self.accumulator = 0.0
        self.samples = 0.0

    def update(self, ground, predict):
        self.samples = self.samples + 1

    def get(self):
        return self.accumulator / self.samples

    def reset(self):
        self.accumulator = 0.0
        self.samples = 0.0


class Dice(Metrics):
    def __init__(self, name="Dice", input_index=0, target_index=0, classes=4):
        super(Dice, self).__init__(name)
        self.input_index = input_index
        self.target_index = target_index
        self.classes = classes

    def update(self, ground, predict):
        pred = predict[self.input_index].detach()
        gr = ground[self.target_index].detach()
        assert gr.shape == pred.shape
        counts = ops.dice_counts(pred.cuda(), gr.cuda())                        # [N,C,2] int64, stays on the device: no host sync per update
        nacc = self.classes - 1
        if nacc <= int(counts.shape[1]) and nacc <= 64:
            # ratio (float32, 0/0 -> NaN -> 1), batch mean and accumulation (float64) in one tiny launch: metrics.py:124-130 without a
            # dozen ATen passes over 12 numbers per training step
            if not isinstance(self.accumulator, torch.Tensor):
                self.accumulator = torch.full((nacc,), float(self.accumulator), dtype=torch.float64, device=counts.device)
            ops.dice_accumulate(counts, self.accumulator, nacc)
        else:
            cf = counts.to(torch.float32)
            r = ((2 * cf[..., 0]) / cf[..., 1]).to(torch.float64)               # float32 division like metrics.py:126 (0/0 -> NaN)
            r = torch.where(torch.isnan(r), torch.ones_like(r), r)              # metrics.py:127
            self.accumulator = self.accumulator + r[:, :nacc].mean(dim=0)
        self.samples += 1

    def get(self):
        acc = self.accumulator
        if isinstance(acc, torch.Tensor):
            acc = acc.cpu().numpy()                                             # the one device -> host copy, when the value is asked for
        return acc / self.samples
