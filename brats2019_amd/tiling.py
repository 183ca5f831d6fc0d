"""Sliding-window helpers with the semantics of loader_helper.py:34-97 (`get_indices`, `copy`, `copy_back`),
kept on whatever device the tensors live on (torch slicing = device memory plumbing; no host round trips,
unlike the reference's per-tile .cuda()/.cpu(), train.py:165-171)."""
from __future__ import annotations

import math

import torch


def get_indices(position, center_shape, border):
    """loader_helper.py:34-40: the tile is centre block `position` grown by `border` on every side."""
    lo = [p * c - b for p, c, b in zip(position, center_shape, border)]
    hi = [(p + 1) * c + b for p, c, b in zip(position, center_shape, border)]
    return lo, hi


def copy(data, tile_shape, index_min, index_max):
    """loader_helper.py:42-60: zero-padded extract data[:, :, min:max] -> [N,C,*tile_shape]."""
    tile = torch.zeros(tuple(data.shape[:2]) + tuple(tile_shape), dtype=torch.float32, device=data.device)
    src, dst = [], []
    for a in range(3):
        lo, hi = max(int(index_min[a]), 0), min(int(index_max[a]), int(data.shape[2 + a]))
        src.append(slice(lo, hi))
        dst.append(slice(lo - int(index_min[a]), hi - int(index_min[a])))
    tile[:, :, dst[0], dst[1], dst[2]] = data[:, :, src[0], src[1], src[2]]
    return tile


def copy_back(data, tile, center_shape, index_min, index_max, border):
    """loader_helper.py:82-97: paste the tile's centre block into `data`, clipped at the volume end."""
    src, dst = [], []
    for a in range(3):
        lo = int(index_min[a]) + int(border[a])
        hi = min(int(index_max[a]) - int(border[a]), int(data.shape[2 + a]))
        dst.append(slice(lo, hi))
        src.append(slice(int(border[a]), int(border[a]) + (hi - lo)))
    data[:, :, dst[0], dst[1], dst[2]] = tile[:, :, src[0], src[1], src[2]].to(data.device)


def grid_for(shape, center_shape):
    """train.py:158: number of centre blocks per axis."""
    return [int(math.ceil(s / c)) for s, c in zip(shape, center_shape)]
