"""Entry point with the reference's `test.py --name --models_path` contract (test.py:17-19,64-176): load the best checkpoint
of <models_path>/<name>/ and segment cases.  The reference hard-codes its NIfTI input/output directories (test.py:68-69)
and needs nibabel / skimage, which this image lacks; here a case is a `.npy` array [4,D,H,W] (t1, t1ce, t2, flair stacked as
loader_helper.read_multimodal does) and the result a uint8 `.npy` label volume {0,1,2,4}.

    python -m brats2019_amd.test --name brain-tumor-segmentation-0002 --models_path ./models --input case.npy --output seg.npy
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import inference, train

parser = argparse.ArgumentParser(description="PyTorch BraTS2019 (MI355X HIP engine)")
parser.add_argument("--name", default="test", type=str, help="Name of the experiment")
parser.add_argument("--models_path", default="/models", type=str, help="Path to models folder")
parser.add_argument("--input", default=None, type=str, nargs="+", help=".npy case(s) [4,D,H,W]; default: a synthetic 128^3 case")
parser.add_argument("--output", default=None, type=str, help="output .npy (single case) or directory")
parser.add_argument("--precision", default="bf16x3", choices=["bf16x3", "f32"])


def main(argv=None):
    opt = parser.parse_args(argv)
    print(torch.__version__)
    print(opt)
    trainer = train.Trainer(name=opt.name, models_root=opt.models_path, rewrite=False, connect_tb=False)
    trainer.load_best()
    trainer.state.cuda = True
    net = trainer.model.module if hasattr(trainer.model, "module") else trainer.model
    net.set_precision(opt.precision)
    net.cuda()
    cases = opt.input
    if not cases:
        rng = np.random.default_rng(0)
        img = np.zeros((4, 128, 128, 128), np.float32)
        img[:, 8:120, 8:120, 8:120] = rng.random((4, 112, 112, 112)).astype(np.float32) + 0.05
        cases = [("synthetic", img)]
    else:
        cases = [(os.path.splitext(os.path.basename(c))[0], np.load(c)) for c in cases]
    for name, image in cases:
        labels, (wt, tc, et) = inference.predict_case(net, image)
        if opt.output:
            dst = opt.output if opt.output.endswith(".npy") and len(cases) == 1 else os.path.join(opt.output, name + ".npy")
            os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
            np.save(dst, labels)
        print(name, labels.shape, labels.dtype, wt, tc, et)


if __name__ == "__main__":
    main()
