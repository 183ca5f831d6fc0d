"""Drop-in for the part of the reference's `metrics` module that the hot path's Trainer uses: the `Metrics` accumulator
protocol (metrics.py:6-20: name / update / get / reset) and `Dice` (metrics.py:101-133) -- threshold 0.5, per sample and
channel 2*sum(p*g)/sum(p+g) (NaN -> 1), mean over the batch, accumulated over update() calls.  The counting runs on the
device (ru_dice_counts); only N*C pairs of integers reach the host.  `update(ground, predict)` keeps the reference's
argument order (train.py:304)."""
from __future__ import annotations

import numpy as np

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
        counts = ops.dice_counts(pred.cuda(), gr.cuda()).cpu().numpy().astype(np.float32)     # [N,C,2]
        with np.errstate(invalid="ignore", divide="ignore"):
            r = (2 * counts[..., 0] / counts[..., 1]).astype(np.float64)    # float32 division like metrics.py:126
        r[np.isnan(r)] = 1
        self.accumulator = self.accumulator + r[:, : self.classes - 1].mean(axis=0)
        self.samples += 1
