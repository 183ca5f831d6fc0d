"""Un-pickling compatibility: reference checkpoints are whole-module pickles whose classes are named
`model.UNet`, `model.Residual`, `model.conv`, `model.Trilinear` and `train.TrainingState`
(train.py:320-324).  `install_aliases()` registers this package's modules under those top-level names
(only if nothing else already owns them) so `torch.load(..., weights_only=False)` resolves them here."""
from __future__ import annotations

import sys


def install_aliases(force=False):
    from . import loss, metrics, model, train
    for name, mod in (("model", model), ("train", train), ("loss", loss), ("metrics", metrics)):
        if force or name not in sys.modules:
            sys.modules[name] = mod
