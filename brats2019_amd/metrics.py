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
        counts = ops.dice_counts(pred.cuda(), gr.cuda()).to(torch.float32)      # [N,C,2], stays on the device: no host sync per update
        r = ((2 * counts[..., 0]) / counts[..., 1]).to(torch.float64)           # float32 division like metrics.py:126 (0/0 -> NaN)
        r = torch.where(torch.isnan(r), torch.ones_like(r), r)                  # metrics.py:127
        self.accumulator = self.accumulator + r[:, : self.classes - 1].mean(dim=0)
        self.samples += 1

    def get(self):
        acc = self.accumulator
        if isinstance(acc, torch.Tensor):
            acc = acc.cpu().numpy()                                             # the one device -> host copy, when the value is asked for
        return acc / self.samples
