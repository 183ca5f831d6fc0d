"""Thin tensor-level wrappers over the op entry points of libresunet_hip.so.  They allocate outputs and
scratch with torch (device memory plumbing) and pass raw pointers through the C-ABI; every function
documents the reference call site it stands in for.  Used by loss.py / train.py and by the per-op parity
tests; the network itself goes through engine.py (one C call per forward/backward)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

LEAKY_SLOPE = 1e-2


def _dims5(x):
    if x.dim() != 5:
        raise ValueError("expected an NCDHW tensor, got shape %s" % (tuple(x.shape),))
    return [int(v) for v in x.shape]


def _prep(t):
    L.require_gpu()
    if t is None:
        return None
    return t.contiguous().float() if (not t.is_contiguous() or t.dtype != torch.float32) else t


def conv3d(x, w, bias=None, precision="f32"):
    """nn.Conv3d forward as built at model.py:72-73,336,348 (k=3,s=1,p=1), :361-363 (k=2,s=2), :393,401 (k=1).
    `precision`: "f32" (exact) or "bf16x3" (split-bf16, 3 MFMA products; k=3 only)."""
    x, w, bias = _prep(x), _prep(w), _prep(bias)
    lib = L.load()
    n, cin, d, h, wd = _dims5(x)
    cout, k = int(w.shape[0]), int(w.shape[2])
    if int(w.shape[1]) != cin:
        raise ValueError("weight expects %d input channels, input has %d" % (int(w.shape[1]), cin))
    out_sp = (d, h, wd) if k != 2 else (d // 2, h // 2, wd // 2)
    y = torch.empty((n, cout) + out_sp, dtype=torch.float32, device=x.device)
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, cin, cout, d, h, wd, k), x.device)
    L.check(lib.ru_conv3d_fwd_p(L.f32(x), L.f32(w), L.ptr(bias, True), L.f32(y), n, cin, cout, d, h, wd, k, L.PRECISIONS[precision],
                                L.ptr(ws), ws.numel(), L.stream()), "ru_conv3d_fwd")
    return y


def conv3d_bwd_data(dy, w, in_spatial, precision="f32"):
    """Data gradient of the conv above; `in_spatial` = (D,H,W) of the conv INPUT."""
    dy, w = _prep(dy), _prep(w)
    lib = L.load()
    n = int(dy.shape[0])
    cout, cin, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
    d, h, wd = [int(v) for v in in_spatial]
    dx = torch.empty((n, cin, d, h, wd), dtype=torch.float32, device=dy.device)
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, cin, cout, d, h, wd, k), dy.device)
    L.check(lib.ru_conv3d_bwd_data_p(L.f32(dy), L.f32(w), L.f32(dx), n, cin, cout, d, h, wd, k, L.PRECISIONS[precision],
                                     L.ptr(ws), ws.numel(), L.stream()), "ru_conv3d_bwd_data")
    return dx


def conv3d_bwd_weight(x, dy, k, with_bias=False, precision="f32"):
    """Weight (and bias) gradient of the conv above."""
    x, dy = _prep(x), _prep(dy)
    lib = L.load()
    n, cin, d, h, wd = _dims5(x)
    cout = int(dy.shape[1])
    dw = torch.empty((cout, cin, k, k, k), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if with_bias else None
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, cin, cout, d, h, wd, k), x.device)
    L.check(lib.ru_conv3d_bwd_weight_p(L.f32(x), L.f32(dy), L.f32(dw), L.ptr(db, True), n, cin, cout, d, h, wd, k, L.PRECISIONS[precision],
                                       L.ptr(ws), ws.numel(), L.stream()), "ru_conv3d_bwd_weight")
    return (dw, db) if with_bias else dw


def group_norm(x, gamma, beta, groups=8, eps=1e-5, slope=1.0, residual=None):
    """nn.GroupNorm(8, C) (model.py:95-96,338) + LeakyReLU(slope) (model.py:93-94; 1.0 = none) + optional
    residual add (model.py:115).  Returns (y, mean[N*G], rstd[N*G])."""
    x, gamma, beta, residual = _prep(x), _prep(gamma), _prep(beta), _prep(residual)
    lib = L.load()
    n, c = int(x.shape[0]), int(x.shape[1])
    v = x.numel() // (n * c)
    y = torch.empty_like(x)
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = L.workspace(lib.ru_groupnorm_workspace_bytes(n, c, v), x.device)
    L.check(lib.ru_groupnorm_fwd(L.f32(x), L.f32(gamma), L.f32(beta), L.ptr(residual, True), L.f32(y), L.f32(mean), L.f32(rstd),
                                 n, c, v, groups, eps, slope, L.ptr(ws), ws.numel(), L.stream()), "ru_groupnorm_fwd")
    return y, mean, rstd


def group_norm_bwd(x, gamma, beta, mean, rstd, dy, groups=8, slope=1.0):
    """Backward of lrelu(GN(x)); returns (dx, dgamma, dbeta)."""
    x, gamma, beta, dy = _prep(x), _prep(gamma), _prep(beta), _prep(dy)
    lib = L.load()
    n, c = int(x.shape[0]), int(x.shape[1])
    v = x.numel() // (n * c)
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    ws = L.workspace(lib.ru_groupnorm_workspace_bytes(n, c, v), x.device)
    L.check(lib.ru_groupnorm_bwd(L.f32(x), L.f32(gamma), L.f32(beta), L.f32(mean), L.f32(rstd), L.f32(dy), L.f32(dx), L.f32(dgamma),
                                 L.f32(dbeta), n, c, v, groups, slope, L.ptr(ws), ws.numel(), L.stream()), "ru_groupnorm_bwd")
    return dx, dgamma, dbeta


def leaky_relu(x, slope=LEAKY_SLOPE):
    x = _prep(x)
    y = torch.empty_like(x)
    L.check(L.load().ru_leaky_relu_fwd(L.f32(x), L.f32(y), x.numel(), slope, L.stream()), "ru_leaky_relu_fwd")
    return y


def leaky_relu_bwd(y, dy, slope=LEAKY_SLOPE):
    y, dy = _prep(y), _prep(dy)
    dx = torch.empty_like(y)
    L.check(L.load().ru_leaky_relu_bwd(L.f32(y), L.f32(dy), L.f32(dx), y.numel(), slope, L.stream()), "ru_leaky_relu_bwd")
    return dx


def upsample2x(x):
    """model.Trilinear(scale=2) (model.py:7-14)."""
    x = _prep(x)
    n, c, d, h, w = _dims5(x)
    y = torch.empty((n, c, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    L.check(L.load().ru_upsample2x_trilinear_fwd(L.f32(x), L.f32(y), n, c, d, h, w, L.stream()), "ru_upsample2x_trilinear_fwd")
    return y


def upsample2x_bwd(dy):
    dy = _prep(dy)
    n, c, d2, h2, w2 = _dims5(dy)
    dx = torch.empty((n, c, d2 // 2, h2 // 2, w2 // 2), dtype=torch.float32, device=dy.device)
    L.check(L.load().ru_upsample2x_trilinear_bwd(L.f32(dy), L.f32(dx), n, c, d2 // 2, h2 // 2, w2 // 2, L.stream()),
            "ru_upsample2x_trilinear_bwd")
    return dx


def sigmoid(x):
    x = _prep(x)
    y = torch.empty_like(x)
    L.check(L.load().ru_sigmoid_fwd(L.f32(x), L.f32(y), x.numel(), L.stream()), "ru_sigmoid_fwd")
    return y


def criterion_sums(p, g, bg_weight=1e-2):
    """Phase 1 of the criterion (loss.py:76-79,114-115): float64 device tensor [2C+1] =
    (sum p*g per class, sum p^2+g per class, BCE log-sum) over THIS shard, no epsilons."""
    p, g = _prep(p), _prep(g)
    if p.shape != g.shape:
        raise AssertionError("prediction/target shape mismatch")      # loss.py:71,107 assert
    lib = L.load()
    n, c = int(p.shape[0]), int(p.shape[1])
    v = p.numel() // (n * c)
    sums = torch.empty(2 * c + 1, dtype=torch.float64, device=p.device)
    ws = L.workspace(lib.ru_criterion_workspace_bytes(n, c, v), p.device)
    L.check(lib.ru_criterion_sums(L.f32(p), L.f32(g), L.ptr(sums), n, c, v, bg_weight, L.ptr(ws), ws.numel(), L.stream()),
            "ru_criterion_sums")
    return sums


def criterion_grad(p, g, sums, count, w_dice=0.5, w_bce=0.5, bg_weight=1e-2, priority=1.0):
    """Phase 2: d(w_dice*Dice + w_bce*BCE)/dp from GLOBAL sums / element count."""
    p, g = _prep(p), _prep(g)
    n, c = int(p.shape[0]), int(p.shape[1])
    v = p.numel() // (n * c)
    dp = torch.empty_like(p)
    L.check(L.load().ru_criterion_grad(L.f32(p), L.f32(g), L.ptr(sums), float(count), w_dice, w_bce, bg_weight, priority,
                                       L.f32(dp), n, c, v, L.stream()), "ru_criterion_grad")
    return dp


def criterion_losses(sums, count, priority=1.0, w_dice=0.5, w_bce=0.5):
    """float64 DEVICE tensor [3] = (w_dice*dice + w_bce*bce, dice, bce) from (global) sums -- one launch, no host sync."""
    c = (sums.numel() - 1) // 2
    out = torch.empty(3, dtype=torch.float64, device=sums.device)
    L.check(L.load().ru_criterion_value_device(L.ptr(sums), c, float(count), float(priority), float(w_dice), float(w_bce), L.ptr(out), L.stream()),
            "ru_criterion_value_device")
    return out


def criterion_value(sums, count, priority=1.0):
    """(dice, bce) as float64 0-dim DEVICE tensors from (global) sums -- no host sync."""
    out = criterion_losses(sums, count, priority)
    return out[1], out[2]                                                            # loss.py:114-122, loss.py:79


def adam_amsgrad_step(w, g, m, v, vmax, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam(amsgrad=True) update on flat float32 buffers (main.py:133-137)."""
    L.check(L.load().ru_adam_amsgrad_step(L.f32(w), L.f32(g), L.f32(m), L.f32(v), L.f32(vmax), w.numel(), lr, betas[0], betas[1],
                                          eps, weight_decay, int(step), L.stream()), "ru_adam_amsgrad_step")


TTA_FLIP_BITS = {1: 1, 2: 2, 3: 4}      # axis of a [C,D,H,W] array -> flip bit (D, H, W)


def tta_merge(probs, flip_axes, want_mean=False):
    """Un-flip + average K predictions of flipped copies (test.py:134-138), threshold at 0.5 (test.py:144).
    probs: [K,C,D,H,W]; flip_axes: per copy the tuple of [C,D,H,W] axes that copy was flipped along.
    Returns (mask uint8 [C,D,H,W], counts uint64-as-int64 [C], mean or None)."""
    probs = _prep(probs)
    k, c, d, h, w = [int(v) for v in probs.shape]
    flips = 0
    for i, axes in enumerate(flip_axes):
        for ax in axes:
            flips |= TTA_FLIP_BITS[ax] << (3 * i)
    mask = torch.empty((c, d, h, w), dtype=torch.uint8, device=probs.device)
    counts = torch.empty(c, dtype=torch.int64, device=probs.device)
    mean = torch.empty((c, d, h, w), dtype=torch.float32, device=probs.device) if want_mean else None
    L.check(L.load().ru_tta_merge(L.f32(probs), k, flips, L.ptr(mean, True), L.ptr(mask), L.ptr(counts), c, d, h, w, L.stream()), "ru_tta_merge")
    return mask, counts, mean


def compose_labels(mask, counts, et_min=32):
    """test.py:153-159: labels {0,1,2,4} from the WT/TC/ET masks; ET only if more than `et_min` ET voxels."""
    v = mask.numel() // int(mask.shape[0])
    labels = torch.empty(tuple(mask.shape[1:]), dtype=torch.uint8, device=mask.device)
    L.check(L.load().ru_compose_labels(L.ptr(mask), L.ptr(counts), int(et_min), L.ptr(labels), v, L.stream()), "ru_compose_labels")
    return labels


def _ints(v):
    import ctypes
    v = [int(x) for x in v]
    return (ctypes.c_int * len(v))(*v)


def _flip_bits(flip_axes):
    flips = 0
    for i, axes in enumerate(flip_axes):
        for ax in axes:
            flips |= TTA_FLIP_BITS[ax] << (3 * i)
    return flips


def tile_gather(data, tile_shape, origins):
    """loader_helper.copy (:42-60) for T tiles in one launch: data [N,C,D,H,W] -> [T*N, C, *tile_shape], tile t = the zero-padded
    block starting at origins[t] (= get_indices' index_min; may be negative)."""
    data = _prep(data)
    n, c, d, h, w = _dims5(data)
    t = len(origins)
    td, th, tw = (int(v) for v in tile_shape)
    tiles = torch.empty((t * n, c, td, th, tw), dtype=torch.float32, device=data.device)
    L.check(L.load().ru_tile_gather(L.f32(data), L.f32(tiles), n, c, d, h, w, t, _ints([v for o in origins for v in o]), td, th, tw, L.stream()),
            "ru_tile_gather")
    return tiles


def tile_scatter(out, tiles, origins, border, center):
    """loader_helper.copy_back (:82-97) for T tiles in one launch: the centre block of tile t goes to out[..., origins[t] + border ...]."""
    tiles = _prep(tiles)
    if not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32):
        raise ValueError("tile_scatter: `out` must be a contiguous float32 device tensor (it is written in place)")
    n, c, d, h, w = _dims5(out)
    t = len(origins)
    td, th, tw = (int(v) for v in tiles.shape[2:])
    if int(tiles.shape[0]) != t * n or int(tiles.shape[1]) != c:
        raise ValueError("tile_scatter: tiles %s do not match %d tiles of a %s volume" % (tuple(tiles.shape), t, tuple(out.shape)))
    L.check(L.load().ru_tile_scatter(L.f32(tiles), L.f32(out), n, c, d, h, w, t, _ints([v for o in origins for v in o]), td, th, tw,
                                     _ints(border), _ints(center), L.stream()), "ru_tile_scatter")
    return out


def case_bbox(image):
    """test.py:47-49 on the device: per modality {min z, y, x, max z, y, x} of the non-zero voxels -> int64 numpy [C,6] (one small copy to
    the host: the crop extents fix the shapes of everything downstream); all-zero modality: {-1,-1,-1, 0,0,0} as loader_helper.bbox3."""
    image = _prep(image)
    c, d, h, w = (int(v) for v in image.shape)
    box = torch.empty((c, 6), dtype=torch.int32, device=image.device)
    L.check(L.load().ru_case_bbox(L.f32(image), L.ptr(box), c, d, h, w, L.stream()), "ru_case_bbox")
    b = box.cpu().numpy().astype("int64")
    empty = b[:, 3] < 0
    b[empty, :3] = -1
    b[empty, 3:] = 0
    return b


def case_stats(image, lo, size):
    """float64 device tensor [C,3] = count(x > 0), sum x, sum x^2 over the crop box (test.py:103-111)."""
    image = _prep(image)
    c, d, h, w = (int(v) for v in image.shape)
    lib = L.load()
    stats = torch.empty((c, 3), dtype=torch.float64, device=image.device)
    ws = L.workspace(lib.ru_case_workspace_bytes(c, d, h, w), image.device)
    L.check(lib.ru_case_stats(L.f32(image), L.ptr(stats), c, d, h, w, _ints(lo), _ints(size), L.ptr(ws), ws.numel(), L.stream()), "ru_case_stats")
    return stats


def case_prepare(image, stats, lo, size, pad_left, padded, flip_axes):
    """[K,C,*padded] = the K test-time flips of the crop, zero-padded and z-scored (test.py:85-120) -- the batch the network takes."""
    image = _prep(image)
    c, d, h, w = (int(v) for v in image.shape)
    k = len(flip_axes)
    batch = torch.empty((k, c) + tuple(int(v) for v in padded), dtype=torch.float32, device=image.device)
    L.check(L.load().ru_case_prepare(L.f32(image), L.ptr(stats), L.f32(batch), c, d, h, w, _ints(lo), _ints(size), _ints(pad_left), _ints(padded),
                                     k, _flip_bits(flip_axes), L.stream()), "ru_case_prepare")
    return batch


def tta_merge_box(probs, flip_axes, lo, size, want_mean=False):
    """tta_merge restricted to the box [lo, lo+size) of the padded prediction (test.py:134-144 with the padding removed):
    (mask uint8 [C,*size], counts int64 [C], mean or None)."""
    probs = _prep(probs)
    k, c, d, h, w = [int(v) for v in probs.shape]
    size = tuple(int(v) for v in size)
    mask = torch.empty((c,) + size, dtype=torch.uint8, device=probs.device)
    counts = torch.empty(c, dtype=torch.int64, device=probs.device)
    mean = torch.empty((c,) + size, dtype=torch.float32, device=probs.device) if want_mean else None
    L.check(L.load().ru_tta_merge_box(L.f32(probs), k, _flip_bits(flip_axes), L.ptr(mean, True), L.ptr(mask), L.ptr(counts), c, d, h, w,
                                      _ints(lo), _ints(size), L.stream()), "ru_tta_merge_box")
    return mask, counts, mean


def cc_reject(labels, ratio=0.1):
    """test.py:162-164 in place on a uint8 device label volume [D,H,W]: 26-connected components of labels > 0, components smaller than
    ratio * (voxels - most frequent label's voxels) are zeroed (test.py:51-62)."""
    if not (labels.is_cuda and labels.is_contiguous() and labels.dtype == torch.uint8 and labels.dim() == 3):
        raise ValueError("cc_reject: contiguous uint8 device tensor [D,H,W]")
    d, h, w = (int(v) for v in labels.shape)
    lib = L.load()
    ws = L.workspace(lib.ru_cc_workspace_bytes(d, h, w), labels.device)
    L.check(lib.ru_cc_reject(L.ptr(labels), d, h, w, float(ratio), L.ptr(ws), ws.numel(), L.stream()), "ru_cc_reject")
    return labels


def paste_labels(lab, full_shape, lo):
    """test.py:167-168: uint8 device volume `full_shape`, zero except the box at `lo`, which holds `lab`."""
    if not (lab.is_cuda and lab.is_contiguous() and lab.dtype == torch.uint8 and lab.dim() == 3):
        raise ValueError("paste_labels: contiguous uint8 device tensor [d,h,w]")
    full = torch.empty(tuple(int(v) for v in full_shape), dtype=torch.uint8, device=lab.device)
    d, h, w = (int(v) for v in full.shape)
    L.check(L.load().ru_paste_labels(L.ptr(lab), L.ptr(full), d, h, w, _ints(lo), _ints(lab.shape), L.stream()), "ru_paste_labels")
    return full


def dice_counts(pred, target):
    """metrics.Dice.update counting step (metrics.py:116-127): int64 device tensor [N,C,2] = (#(p>.5 & g>.5), #(p>.5) + #(g>.5))."""
    pred, target = _prep(pred), _prep(target)
    assert pred.shape == target.shape                                  # metrics.py:112
    n, c = int(pred.shape[0]), int(pred.shape[1])
    v = pred.numel() // (n * c)
    counts = torch.empty((n, c, 2), dtype=torch.int64, device=pred.device)
    L.check(L.load().ru_dice_counts(L.f32(pred), L.f32(target), L.ptr(counts), n, c, v, L.stream()), "ru_dice_counts")
    return counts


def dice_accumulate(counts, acc, nacc):
    """metrics.py:124-130 on the device: acc[c] += batch mean of 2*num/den (float32 ratio, NaN -> 1; float64 mean), c < nacc."""
    n, c = int(counts.shape[0]), int(counts.shape[1])
    assert counts.dtype == torch.int64 and counts.is_contiguous() and acc.dtype == torch.float64 and acc.numel() >= nacc
    L.check(L.load().ru_dice_accumulate(L.ptr(counts), L.ptr(acc), n, c, int(nacc), L.stream()), "ru_dice_accumulate")
    return acc


# ---------------------------------------------------------------------- engine-internal voxel-major layout (tests / probes)
def to_c16(x):
    """NCDHW [N,C,D,H,W] -> C16 storage [N,C/16,D,H,W,16] (device kernel ru_layout_convert)."""
    x = _prep(x)
    n, c, d, h, w = _dims5(x)
    y = torch.empty((n, c // 16, d, h, w, 16), dtype=torch.float32, device=x.device)
    L.check(L.load().ru_layout_convert(L.f32(x), L.f32(y), n, c, d * h * w, 1, L.stream()), "ru_layout_convert")
    return y


def from_c16(x):
    x = _prep(x)
    n, cb, d, h, w, _ = (int(v) for v in x.shape)
    y = torch.empty((n, cb * 16, d, h, w), dtype=torch.float32, device=x.device)
    L.check(L.load().ru_layout_convert(L.f32(x), L.f32(y), n, cb * 16, d * h * w, 0, L.stream()), "ru_layout_convert")
    return y


def upsample2x_c16(x, out_slope=1.0):
    """Trilinear x2 on a C16 tensor [N,C/16,D,H,W,16]; LeakyReLU(out_slope) fused on the output (1 = none)."""
    x = _prep(x)
    n, cb, d, h, w, _ = (int(v) for v in x.shape)
    y = torch.empty((n, cb, 2 * d, 2 * h, 2 * w, 16), dtype=torch.float32, device=x.device)
    L.check(L.load().ru_upsample2x_trilinear_fwd_l(L.f32(x), L.f32(y), n, cb * 16, d, h, w, float(out_slope), L.stream()),
            "ru_upsample2x_trilinear_fwd_l")
    return y


def upsample2x_bwd_c16(dy):
    dy = _prep(dy)
    n, cb, d2, h2, w2, _ = (int(v) for v in dy.shape)
    dx = torch.empty((n, cb, d2 // 2, h2 // 2, w2 // 2, 16), dtype=torch.float32, device=dy.device)
    L.check(L.load().ru_upsample2x_trilinear_bwd_l(L.f32(dy), L.f32(dx), n, cb * 16, d2 // 2, h2 // 2, w2 // 2, L.stream()),
            "ru_upsample2x_trilinear_bwd_l")
    return dx


def to_split_c16(x_c16):
    """fp32 voxel-major [N,CB,D,H,W,16] -> the SPLIT form of the same storage size that gn_bwd_apply16 publishes and the gradient
    convolutions read: per voxel and block 64 bytes = [hi bf16 ch0-7 | hi ch8-15 | lo ch0-7 | lo ch8-15], hi = bf16(v), lo = bf16(v - hi)."""
    hi = x_c16.to(torch.bfloat16)
    lo = (x_c16 - hi.to(torch.float32)).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32)


def conv3d_layout(x, w, bias=None, in_c16=False, out_c16=False, few_channels=False, in_split=False, exact_f32=False, activations=False, gradient=False):
    """3x3x3 split-bf16 convolution on tensors in NCDHW or C16 storage (x: 5-D NCDHW or 6-D C16); few_channels: NCDHW input
    with Cin <= 4 through the 4-channel tap-pair kernel; activations: x is an activation tensor (a forward convolution), which lets the
    shapes that have the kernel take the fp16 + MX-fp8 product scheme (conv3_mx.hpp), as the engine's forward convolutions do; gradient: x is a
    gradient tensor -- 16 -> 16 voxel-major shapes take the gradient-operand form of the scheme, as the engine's 16-channel data-gradient convolutions do."""
    x, w, bias = _prep(x), _prep(w), _prep(bias)
    if in_c16:
        n, cb, d, h, wd, _ = (int(v) for v in x.shape)
        cin = cb * 16
    else:
        n, cin, d, h, wd = _dims5(x)
    cout = int(w.shape[0])
    y = torch.empty((n, cout // 16, d, h, wd, 16) if out_c16 else (n, cout, d, h, wd), dtype=torch.float32, device=x.device)
    lib = L.load()
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, cin, cout, d, h, wd, 3) + (n * d * h * wd * 16 + 65536 if few_channels else 0)
                     + (n * d * h * wd * 64 + 65536 if gradient else 0), x.device)
    L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), L.ptr(bias, True), L.f32(y), n, cin, cout, d, h, wd,
                                int(in_c16) | (int(out_c16) << 1) | (int(few_channels) << 2) | (int(in_split) << 3) | (int(exact_f32) << 4) | (int(activations) << 5) | (int(gradient) << 6),
                                L.ptr(ws), ws.numel(), L.stream()), "ru_conv3d_fwd_l")
    return y


def conv3d_bwd_weight_layout(x, dy, x_c16=False, dy_c16=False):
    """3x3x3 split-bf16 weight gradient on tensors in NCDHW or C16 storage."""
    x, dy = _prep(x), _prep(dy)
    if x_c16:
        n, cb, d, h, wd, _ = (int(v) for v in x.shape)
        cin = cb * 16
    else:
        n, cin, d, h, wd = _dims5(x)
    cout = int(dy.shape[1]) * 16 if dy_c16 else int(dy.shape[1])
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    lib = L.load()
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, cin, cout, d, h, wd, 3), x.device)
    L.check(lib.ru_conv3d_bwd_weight_l(L.f32(x), L.f32(dy), L.f32(dw), n, cin, cout, d, h, wd, int(x_c16) | (int(dy_c16) << 1),
                                       L.ptr(ws), ws.numel(), L.stream()), "ru_conv3d_bwd_weight_l")
    return dw
