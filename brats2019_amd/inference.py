"""Inference driver around the forward pass: the per-case pipeline of the reference's test.py (:82-164), on the device from the upload
of the case to the download of its label volume (kernels in csrc/inference.hip) --

  ru_case_bbox      bounding box of the non-zero voxels (test.py:47-49,85-87); its 6 integers are the only values that visit the host
                    in between: they fix the tensor shapes of everything downstream;
  ru_case_stats     non-zero z-score moments of the crop (test.py:103-111), float64;
  ru_case_prepare   crop + zero-pad to x16 (test.py:92-99) + z-score (:113) + the FOUR test-time flips (:115-120) written as ONE batch;
  model             one batch-4 forward instead of four forward calls with host round trips (:124-132);
  ru_tta_merge_box  un-flip + average + un-pad + threshold + per-class counts (:134-144); ru_compose_labels (:146-159);
  ru_cc_reject      26-connected components (skimage.morphology.label) + rejection of regions below ratio 0.1 (:51-62,162-164);
  ru_paste_labels   paste into the full volume (:167-168).

The numpy functions below (`get_bbox`, `prepare_case`, `reject_small_regions`, `postprocess_labels`) are the host restatement the
device pipeline is tested against; `predict_case` does not call them.  NIfTI reading/writing (nibabel) is out of scope; `predict_case`
takes and returns arrays.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

TTA_FLIPS = ((), (1,), (2,), (1, 2))          # test.py:117-120, axes of the [C,D,H,W] volume


def closest_to_k(n, k=16):
    """loader_helper.py:99-103."""
    return n if n % k == 0 else (n // k + 1) * k


def get_bbox(image):
    """test.py:47-49 / loader_helper.py:105-129: union bounding box of the non-zero voxels of every modality."""
    lo, hi = [], []
    for d in image:
        nz = np.nonzero(d)
        if nz[0].size == 0:
            lo.append([-1, -1, -1]); hi.append([0, 0, 0])
        else:
            lo.append([a.min() for a in nz]); hi.append([a.max() for a in nz])
    return np.stack([np.min(lo, axis=0), np.max(hi, axis=0)], axis=0)


def prepare_case(image):
    """test.py:85-113 -> (normalised padded crop [C,D,H,W] float32, bbox, pad_left, pad_right)."""
    bbox = get_bbox(image)
    crop = image[:, bbox[0, 0]:bbox[1, 0], bbox[0, 1]:bbox[1, 1], bbox[0, 2]:bbox[1, 2]]
    old = np.array(crop.shape[1:])
    new = np.array([closest_to_k(int(i), 16) for i in old])
    diff = new - old
    left = diff // 2
    right = diff - left
    x = np.pad(crop, ((0, 0),) + tuple((int(left[i]), int(right[i])) for i in range(3)), mode="constant", constant_values=0)
    n = (x > 0).sum(axis=(1, 2, 3))
    mean = np.sum(x / n[:, None, None, None], axis=(1, 2, 3))
    mean2 = np.sum(np.square(x) / n[:, None, None, None], axis=(1, 2, 3))
    std = np.sqrt(mean2 - mean * mean)
    x = (x - mean.reshape(-1, 1, 1, 1)) / std.reshape(-1, 1, 1, 1)
    return x.astype(np.float32), bbox, left, right


def reject_small_regions(connectivity, ratio=0.25):
    """test.py:51-62."""
    out = connectivity.copy()
    unique, counts = np.unique(connectivity, return_counts=True)
    nonzero = connectivity.size - counts.max()
    for u, c in zip(unique, counts):
        if c < ratio * nonzero:
            out[out == u] = 0
    return out


def predict_tta(model, x, pad_left=(0, 0, 0), pad_right=(0, 0, 0), want_mean=False):
    """x: [C,D,H,W] float32 (numpy or tensor), extents divisible by 8.  One batch-4 forward over the four flips, then the
    device-side merge; the padding is removed BEFORE the labels are composed (test.py:140-159: the ET > 32 rule counts
    un-padded voxels).  Returns (labels uint8 [d,h,w] device tensor, counts [3] device tensor, mean probs or None)."""
    xt = torch.as_tensor(x, dtype=torch.float32).cuda()
    batch = torch.stack([torch.flip(xt, dims=list(ax)) if ax else xt for ax in TTA_FLIPS], dim=0).contiguous()
    model.eval()
    if hasattr(model, "freeze_params"):
        model.freeze_params(True)                       # constant weights: packed once, reused by every later forward (dropped by .train() / load_state_dict)
    with torch.no_grad():
        probs = model([batch])[0]                       # [4,3,D,H,W]
    mask, counts, mean = ops.tta_merge(probs, TTA_FLIPS, want_mean=want_mean)
    d, h, w = [int(v) for v in mask.shape[1:]]
    if any(int(v) for v in pad_left) or any(int(v) for v in pad_right):
        mask = mask[:, int(pad_left[0]):d - int(pad_right[0]), int(pad_left[1]):h - int(pad_right[1]),
                    int(pad_left[2]):w - int(pad_right[2])].contiguous()
        counts = mask.sum(dim=(1, 2, 3), dtype=torch.int64)
        if mean is not None:
            mean = mean[:, int(pad_left[0]):d - int(pad_right[0]), int(pad_left[1]):h - int(pad_right[1]), int(pad_left[2]):w - int(pad_right[2])]
    labels = ops.compose_labels(mask, counts, et_min=32)
    return labels, counts, mean


def postprocess_labels(labels):
    """test.py:162-164 on the host (26-connected components, ratio 0.1)."""
    import scipy.ndimage as ndi
    comp, _ = ndi.label(labels > 0, structure=np.ones((3, 3, 3), dtype=bool))
    clusters = reject_small_regions(comp, 0.1)
    labels = labels.copy()
    labels[clusters == 0] = 0
    return labels


def prepare_case_device(image):
    """test.py:85-120 on the device.  image: [C,D,H,W] float32 device tensor.  Returns (batch [4,C,Dp,Hp,Wp] = the four test-time flips
    of the padded, normalised crop; lo, size = the crop box; pad_left; padded extents)."""
    boxes = ops.case_bbox(image)                                     # [C,6] on the host: the only device -> host copy before the labels
    lo = boxes[:, :3].min(axis=0)
    hi = boxes[:, 3:].max(axis=0)                                    # test.py:87 uses the max INDEX as an exclusive slice end
    size = hi - lo
    if (lo < 0).any() or (size <= 0).any():
        raise ValueError("predict_case: a modality without non-zero voxels (or a one-voxel-thick box) gives an empty crop (test.py:85-87)")
    padded = np.array([closest_to_k(int(v), 16) for v in size])
    left = (padded - size) // 2
    stats = ops.case_stats(image, lo, size)
    batch = ops.case_prepare(image, stats, lo, size, left, padded, TTA_FLIPS)
    return batch, lo, size, left, padded


def predict_case_device(model, image):
    """The per-case pipeline of test.py:82-168 with every array on the device: image [4,D,H,W] device tensor -> (uint8 device label volume
    [D,H,W] with values {0,1,2,4}, int64 device tensor of the (wt, tc, et) voxel counts)."""
    image = image.contiguous().float()
    batch, lo, size, left, _padded = prepare_case_device(image)
    model.eval()
    if hasattr(model, "freeze_params"):
        model.freeze_params(True)
    with torch.no_grad():
        probs = model([batch])[0]                                    # [4,3,Dp,Hp,Wp]
    mask, counts, _ = ops.tta_merge_box(probs, TTA_FLIPS, left, size)
    labels = ops.compose_labels(mask, counts, et_min=32)
    ops.cc_reject(labels, 0.1)
    return ops.paste_labels(labels, image.shape[1:], lo), counts


def predict_case(model, image):
    """Full per-case pipeline of test.py:82-168 for one multimodal volume `image` [4,D,H,W] (numpy or tensor): one upload, the device
    pipeline above, one download.  Returns (uint8 label volume [D,H,W] with values {0,1,2,4}, (wt, tc, et) voxel counts)."""
    img = torch.as_tensor(np.asarray(image) if not isinstance(image, torch.Tensor) else image, dtype=torch.float32).cuda()
    full, counts = predict_case_device(model, img)
    return full.cpu().numpy(), tuple(int(v) for v in counts.cpu().tolist())
