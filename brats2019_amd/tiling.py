"""Sliding-window helpers with the semantics of loader_helper.py:34-97 (`get_indices`, `copy`, `copy_back`) as device kernels:
`copy_tiles` cuts T zero-padded tiles straight into the batch tensor of one forward (ru_tile_gather), `copy_back_tiles` pastes their
centre blocks (ru_tile_scatter) -- one launch each per batch of tiles, no host round trips (the reference does .cuda()/.cpu() per tile,
train.py:165-171) and no per-tile slice copies.  `copy` / `copy_back` keep the reference's single-tile signatures on top of them."""
from __future__ import annotations

import math

import torch

from . import ops


def get_indices(position, center_shape, border):
    """loader_helper.py:34-40: the tile is centre block `position` grown by `border` on every side."""
    lo = [p * c - b for p, c, b in zip(position, center_shape, border)]
    hi = [(p + 1) * c + b for p, c, b in zip(position, center_shape, border)]
    return lo, hi


def copy_tiles(data, tile_shape, index_mins):
    """T tiles of `data` [N,C,D,H,W] in one launch -> [T*N, C, *tile_shape] (rows t*N .. t*N+N-1 = tile t, i.e. torch.cat of the
    reference's per-tile `copy` results along the batch axis)."""
    return ops.tile_gather(data, tile_shape, [tuple(int(v) for v in lo) for lo in index_mins])


def copy_back_tiles(data, tiles, center_shape, index_mins, border):
    """paste the centre blocks of T tiles (layout of `copy_tiles`) into `data` in place, clipped at the volume end."""
    return ops.tile_scatter(data, tiles, [tuple(int(v) for v in lo) for lo in index_mins], border, center_shape)


def _device_kernels_apply(data, tile_shape):
    # ru_tile_gather / ru_tile_scatter move float4 pieces of device rows: a host tensor or a tile width that is not a multiple of 4
    # takes the index arithmetic of loader_helper.py:42-97 on torch slices instead (host-side helper use, e.g. a CPU label volume)
    return data.is_cuda and data.dtype == torch.float32 and int(tile_shape[-1]) % 4 == 0


def copy(data, tile_shape, index_min, index_max):
    """loader_helper.py:42-60: zero-padded extract data[:, :, min:max] -> [N,C,*tile_shape]."""
    assert all(int(b) - int(a) == int(t) for a, b, t in zip(index_min, index_max, tile_shape))
    if _device_kernels_apply(data, tile_shape):
        return copy_tiles(data, tile_shape, [index_min])
    # float32 whatever the volume's dtype, as loader_helper.py:43 allocates it (an integer label volume yields float tiles there too)
    out = torch.zeros(tuple(data.shape[:2]) + tuple(int(t) for t in tile_shape), dtype=torch.float32, device=data.device)
    src, dst = [slice(None), slice(None)], [slice(None), slice(None)]
    for a, b, n in zip(index_min, index_max, data.shape[2:]):
        lo, hi = max(int(a), 0), min(int(b), int(n))
        src.append(slice(lo, hi))
        dst.append(slice(lo - int(a), hi - int(a)))
    out[tuple(dst)] = data[tuple(src)]
    return out


def copy_back(data, tile, center_shape, index_min, index_max, border):
    """loader_helper.py:82-97: paste the tile's centre block into `data`, clipped at the volume end."""
    if _device_kernels_apply(data, tile.shape[2:]):
        copy_back_tiles(data, tile.to(data.device), center_shape, [index_min], border)
        return
    src, dst = [slice(None), slice(None)], [slice(None), slice(None)]
    for a, c, b, n in zip(index_min, center_shape, border, data.shape[2:]):
        first = int(a) + int(b)                              # first voxel of the centre block in the volume (negative: clamped, loader_helper.py:86)
        lo, hi = max(first, 0), min(first + int(c), int(n))
        dst.append(slice(lo, max(hi, lo)))
        src.append(slice(int(b) + lo - first, int(b) + lo - first + max(hi - lo, 0)))
    data[tuple(dst)] = tile.to(data.device)[tuple(src)].to(data.dtype)


def grid_for(shape, center_shape):
    """train.py:158: number of centre blocks per axis."""
    return [int(math.ceil(s / c)) for s, c in zip(shape, center_shape)]
