"""Device-side training input pipeline -- the part of the reference's `dataloader` module that feeds the hot path
(dataloader.py:67-216 `SimpleReader`, :218-286 `FullReader`): per-channel z-score, random crop around the lesion, affine
zoom 0.7-1.3 (scipy affine_transform order 1 / reflect, restated in the kernel), flips, D<->H transpose, intensity gain /
bias and the WT/TC/ET targets.  At > 100 volumes/s per GPU the reference's CPU path (scipy on 4 x 128^3 float64 per patch)
cannot keep 8 GPUs fed; here the raw case lives in HBM and ONE kernel (`ru_augment_patch`) produces a patch.

What stays on the host, on purpose: file IO (nibabel is not part of this path -- cases are handed over as arrays), the
bounding-box cache (dataloader.py:99-116, once per case) and the random draws, which reproduce the reference's order from
the same global generators (`numpy.random`, `random`), so equal seeds give equal patches (tests/test_dataloader.py).
"""
from __future__ import annotations

import ctypes as C
import random

import numpy as np
import torch

from . import _lib as L


def _bbox3(mask):
    """loader_helper.py:105-129"""
    nz = np.nonzero(mask)
    if nz[0].size == 0:
        return np.array([[-1, -1, -1], [0, 0, 0]])
    return np.array([[a.min() for a in nz], [a.max() for a in nz]])


def label_bbox(label, patch_size):
    """dataloader.py:104-116: where patch centres may fall (lesion box +- 50 voxels, clipped so the patch fits)."""
    bbox = _bbox3(np.asarray(label) > 0).astype(np.float64)
    shape = np.array(np.asarray(label).shape)
    bbox[0] = np.maximum(bbox[0] - 50, np.array(patch_size) / 2.0 + 1)
    bbox[1] = np.minimum(bbox[1] + 50, shape - np.array(patch_size) / 2.0 - 1)
    return bbox


def zscore_stats(image):
    """(mean, std) float64 numpy per channel of a device tensor [C,D,H,W], dataloader.py:124-132: count over x > 0, sums over all."""
    L.require_gpu()
    if not image.is_cuda:
        raise RuntimeError("brats2019_amd.dataloader: expected a ROCm device tensor (HIP-only path)")
    image = image.contiguous().float()
    c = int(image.shape[0])
    v = image.numel() // c
    lib = L.load()
    stats = torch.empty((c, 3), dtype=torch.float64, device=image.device)
    ws = L.workspace(lib.ru_zscore_workspace_bytes(c, v), image.device)
    L.check(lib.ru_zscore_stats(L.f32(image), L.ptr(stats), c, v, L.ptr(ws), ws.numel(), L.stream()), "ru_zscore_stats")
    s = stats.cpu().numpy()
    mean = s[:, 1] / s[:, 0]
    std = np.sqrt(s[:, 2] / s[:, 0] - mean * mean)
    return mean, std


def remap_labels(label):
    """Label contract of the pipeline: {0, 1, 2, 3} with 3 = enhancing tumour.  The reference gets there in its reader
    (loader_helper.read_multimodal: `annotation[annotation == 4] = 3`, loader_helper.py:30) and `np.eye(4)[label]` fails loudly on a raw
    4 if that step is skipped; the device kernel only recognises 1, 2, 3, so raw BraTS labels {0, 1, 2, 4} are remapped HERE (4 -> 3)
    and anything else is refused instead of silently dropping every ET voxel from the WT / TC / ET targets."""
    lab = np.ascontiguousarray(label)
    bad = ~np.isin(lab, (0, 1, 2, 3, 4))
    if bad.any():
        raise ValueError("label volume holds values outside {0,1,2,3,4}: %s" % np.unique(lab[bad])[:8])
    lab = lab.astype(np.uint8)
    lab[lab == 4] = 3
    return lab


class DeviceCase(object):
    """One multimodal case resident in HBM: raw modalities [C,D,H,W] float32, label [D,H,W] uint8 in {0,1,2,3} (raw BraTS 4 is
    remapped to 3, see remap_labels), z-score constants, centre box."""

    def __init__(self, image, label, patch_size, device="cuda"):
        L.require_gpu()
        label = remap_labels(label)
        self.image = torch.as_tensor(np.ascontiguousarray(image, dtype=np.float32)).to(device)
        self.label = torch.as_tensor(label).to(device)
        self.patch_size = tuple(int(p) for p in patch_size)
        self.mean, self.std = zscore_stats(self.image)
        self.bbox = label_bbox(label, self.patch_size)


def draw_augment_params(bbox, patch_size, channels=4):
    """The draws of SimpleReader.__getitem__ (dataloader.py:141-199), in its order, from the same global generators."""
    center = np.random.rand(3)
    center = center * (bbox[1] - bbox[0]) + bbox[0]
    left_bottom = (center - np.array(patch_size) / 2.0).astype(np.int32)
    random.random()                                      # sigma / alpha of the disabled elastic transform (:157-158)
    random.random()
    scale = [0.7 + random.random() * 0.6 for _ in range(3)]
    flips = [random.random() > 0.5 for _ in range(3)]
    transpose = random.random() > 0.5
    gain = np.random.uniform(0.9, 1.1, size=(channels, 1, 1, 1)).reshape(-1)
    bias = np.random.uniform(-0.2, 0.2, size=(channels, 1, 1, 1)).reshape(-1)
    return dict(crop_lo=left_bottom, scale=np.array(scale), flips=flips, transpose=transpose, gain=gain, bias=bias)


def _arr(ctype, values):
    return (ctype * len(values))(*values)


def augment_patch(case, p, patch_size=None):
    """(data [C,Q0,Q1,P2], target [3,Q0,Q1,P2]) float32 device tensors for explicit parameters `p` (see draw_augment_params)."""
    patch = tuple(int(v) for v in (patch_size or case.patch_size))
    c, d, h, w = (int(v) for v in case.image.shape)
    flags = sum(1 << i for i, f in enumerate(p["flips"]) if f) | (8 if p["transpose"] else 0)
    out_sp = (patch[1], patch[0], patch[2]) if p["transpose"] else patch
    data = torch.empty((c,) + out_sp, dtype=torch.float32, device=case.image.device)
    target = torch.empty((3,) + out_sp, dtype=torch.float32, device=case.image.device)
    lib = L.load()
    L.check(lib.ru_augment_patch(L.f32(case.image), L.ptr(case.label),
                                 _arr(C.c_float, [float(v) for v in case.mean]), _arr(C.c_float, [float(1.0 / v) for v in case.std]),
                                 c, d, h, w, _arr(C.c_int, [int(v) for v in p["crop_lo"]]), _arr(C.c_int, list(patch)),
                                 _arr(C.c_double, [float(v) for v in p["scale"]]), flags,
                                 _arr(C.c_float, [float(v) for v in p["gain"]]), _arr(C.c_float, [float(v) for v in p["bias"]]),
                                 L.f32(data), L.f32(target), L.stream()), "ru_augment_patch")
    return data, target


class SimpleReader(torch.utils.data.Dataset):
    """dataloader.py:67-216 over in-memory cases: `cases` is a list of (image [C,D,H,W], label [D,H,W]) arrays (or of callables
    returning such a pair -- the place for a NIfTI reader).  Items are ([data], [target]) like the reference's, on the device."""

    def __init__(self, cases, patch_size, images_in_epoch=4000, patches_from_single_image=1, device="cuda"):
        super(SimpleReader, self).__init__()
        self.cases = list(cases)
        self.patch_size = tuple(patch_size)
        self.images_in_epoch = images_in_epoch
        self.patches_from_single_image = patches_from_single_image
        self.device = device
        self.real_length = len(self.cases)
        self.patches_from_current_image = self.patches_from_single_image + 1     # first item loads (the reference's constructor + first item do)
        self.current_image_index = 0
        self.case = None

    def _load(self, index):
        if self.patches_from_current_image > self.patches_from_single_image or self.case is None:      # dataloader.py:119-121
            self.patches_from_current_image = 0
            self.current_image_index = index
            src = self.cases[index]
            image, label = src() if callable(src) else src
            self.case = DeviceCase(image, label, self.patch_size, self.device)
        self.patches_from_current_image += 1

    def __getitem__(self, index):
        index = index % self.real_length
        self._load(index)
        p = draw_augment_params(self.case.bbox, self.patch_size, int(self.case.image.shape[0]))
        data, target = augment_patch(self.case, p)
        return [data], [target]

    def __len__(self):
        return int(self.images_in_epoch)


class FullReader(torch.utils.data.Dataset):
    """dataloader.py:218-286: whole case, zero-padded to multiples of 16, z-scored, hard WT/TC/ET targets."""

    def __init__(self, cases, device="cuda"):
        super(FullReader, self).__init__()
        self.cases = list(cases)
        self.device = device

    def __getitem__(self, index):
        src = self.cases[index]
        image, label = src() if callable(src) else src
        image, label = np.asarray(image), np.asarray(label)
        new_shape = tuple(int(np.ceil(s / 16.0) * 16) for s in image.shape[1:])      # loader_helper.closest_to_k
        img = np.zeros((image.shape[0],) + new_shape, np.float32)
        lab = np.zeros(new_shape, np.float32)
        img[(slice(None),) + tuple(slice(0, s) for s in image.shape[1:])] = image
        lab[tuple(slice(0, s) for s in label.shape)] = label
        case = DeviceCase(img, lab, new_shape, self.device)
        p = dict(crop_lo=(0, 0, 0), scale=(1.0, 1.0, 1.0), flips=(False, False, False), transpose=False,
                 gain=np.ones(img.shape[0]), bias=np.zeros(img.shape[0]))
        data, target = augment_patch(case, p, new_shape)
        return [data], [target]

    def __len__(self):
        return len(self.cases)
