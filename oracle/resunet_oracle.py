"""CPU ORACLE for the ResUNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product path
(``brats2019_amd``) never imports it and fails loudly when the HIP library is missing.

What it is: a from-scratch restatement, in plain ``torch.nn.functional`` calls and closed-form
numpy, of the one path the build accelerates -- ``model.UNet.forward`` and the Dice/BCE criterion
of lachinov/brats2019.  The arithmetic of that path lives in a third-party dependency that is not
vendored under /root/reference: **PyTorch** (README pins torch 1.2.0; this image has
2.10.0+rocm7.0 CPU).  The op semantics used here (cross-correlation Conv3d, GroupNorm with biased
variance and eps 1e-5, trilinear ``align_corners=False`` scale 2, LeakyReLU 0.01) are identical
between those versions for these arguments, so the oracle is "torch CPU executing the reference's
op sequence".

Parity pin: the reference has no tests and no golden vectors, so this oracle is
pinned by ``tests/golden/*.npz`` -- outputs of the reference's own ``model.py`` / ``loss.py``
imported in the build container by ``tests/golden/make_golden.py`` (committed).
``tests/test_oracle_golden.py`` checks every function below against those files.

Each function cites the reference lines (into /root/reference) it follows.
"""
from __future__ import annotations

from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

LEAKY_SLOPE = 1e-2      # model.py:93-94,352
GN_GROUPS = 8           # model.py:95-96,338
GN_EPS = 1e-5           # nn.GroupNorm default
IN_CHANNELS = 4         # model.py:336 (hard-coded in_channels=4)

# the only configuration the reference ships (main.py:56-59, export_onnx_group_norm.py:23-26)
DEFAULT_CFG = dict(depth=4, encoder_layers=[1, 2, 2, 4], decoder_layers=[1, 1, 1, 1],
                   number_of_channels=[16, 32, 64, 128], number_of_outputs=3)


# --------------------------------------------------------------------------------------
# parameter inventory (model.py:309-404) -- key order == reference state_dict() order
# --------------------------------------------------------------------------------------
def _residual_keys(prefix, cin_down, c):
    """Keys of one ``Residual`` (model.py:81-97): downsample first (registered first when it is a
    Module, model.py:87), then conv1, conv2, norm1, norm2."""
    keys = []
    if cin_down is not None:
        keys.append((prefix + "downsample.0.weight", (c, cin_down, 2, 2, 2)))      # model.py:361-363
    keys += [
        (prefix + "conv1.conv1.weight", (c, c, 3, 3, 3)),                          # model.py:72-73,89
        (prefix + "conv2.conv1.weight", (c, c, 3, 3, 3)),                          # model.py:91
        (prefix + "norm1.weight", (c,)), (prefix + "norm1.bias", (c,)),            # model.py:95
        (prefix + "norm2.weight", (c,)), (prefix + "norm2.bias", (c,)),            # model.py:96
    ]
    return keys


def state_dict_spec(depth=4, encoder_layers=(1, 2, 2, 4), decoder_layers=(1, 1, 1, 1),
                    number_of_channels=(16, 32, 64, 128), number_of_outputs=3):
    """Ordered ``[(key, shape)]`` exactly as ``model.UNet(...).state_dict()`` yields it.

    Registration order in ``UNet.__init__`` (model.py:320-357): encoder_convs, upsampling,
    decoder_convs, decoder_convs1x1, (empty lists), conv_input, norm_input, conv_first, conv_output.
    """
    ch = list(number_of_channels)
    spec = []
    for i in range(depth - 1):                                                      # model.py:374-377
        for j in range(encoder_layers[i + 1]):
            spec += _residual_keys("encoder_convs.%d.%d." % (i, j), ch[i] if j == 0 else None, ch[i + 1])
    for i in range(depth - 1):                                                      # model.py:397-404
        spec.append(("upsampling.%d.1.weight" % i, (ch[i], ch[i + 1], 1, 1, 1)))
    for i in range(depth):                                                          # model.py:379-395
        for j in range(decoder_layers[i]):
            spec += _residual_keys("decoder_convs.%d.%d." % (i, j), None, ch[i])
    for i in range(depth):
        spec.append(("decoder_convs1x1.%d.weight" % i, (ch[i], 2 * ch[i], 1, 1, 1)))
    spec.append(("conv_input.weight", (ch[0], IN_CHANNELS, 3, 3, 3)))               # model.py:336
    spec += [("norm_input.weight", (ch[0],)), ("norm_input.bias", (ch[0],))]        # model.py:338
    for j in range(encoder_layers[0]):                                              # model.py:340-345
        spec += _residual_keys("conv_first.%d." % j, None, ch[0])
    spec.append(("conv_output.weight", (number_of_outputs, ch[0], 3, 3, 3)))        # model.py:348
    spec.append(("conv_output.bias", (number_of_outputs,)))
    return spec


def make_params(seed=1337, **cfg):
    """Synthetic weights per SURVEY.md section 8(d): Conv3d ~ N(0, sqrt(2/((1+0.01^2) fan_in))) (the
    distribution of weight_init.py:23), conv bias ~ N(0,1) (weight_init.py:27), GN gamma ~ U(.5,1.5),
    beta ~ U(-.5,.5); drawn from one ``numpy.random.default_rng(seed)`` in state-dict key order.
    Returns ``OrderedDict[str, np.ndarray(float32)]``."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for key, shape in state_dict_spec(**cfg):
        if len(shape) == 5:
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            std = math.sqrt(2.0 / ((1.0 + LEAKY_SLOPE ** 2) * fan_in))
            v = rng.standard_normal(shape) * std
        elif key.endswith("conv_output.bias"):
            v = rng.standard_normal(shape)
        elif key.endswith(".weight"):
            v = rng.uniform(0.5, 1.5, shape)
        else:
            v = rng.uniform(-0.5, 0.5, shape)
        out[key] = v.astype(np.float32)
    return out


def projection_vectors(name, numel, count=3):
    """Seeded random directions r_j ~ N(0, I) for the gradient-projection pins of the golden fixtures (tests/golden/make_golden.py stores
    <g, r_j> of every parameter gradient of the reference; the GPU tests form the same inner products): data generation only."""
    import zlib
    return [np.random.default_rng([zlib.crc32(name.encode()), j]).standard_normal(numel).astype(np.float32) for j in range(count)]


def make_input(n, d, h, w, seed=1337):
    """x ~ N(0,1) float32 ``[n,4,d,h,w]`` (real inputs are per-channel z-scored, test.py:103-113)."""
    rng = np.random.default_rng(seed + 1)
    return rng.standard_normal((n, IN_CHANNELS, d, h, w)).astype(np.float32)


def make_target(n, d, h, w, seed=1337):
    """Nested WT>=TC>=ET binary masks from one uniform field (dataloader.py:208-212 nesting)."""
    rng = np.random.default_rng(seed + 2)
    u = rng.random((n, 1, d, h, w))
    return np.concatenate([u > 0.70, u > 0.80, u > 0.90], axis=1).astype(np.float32)


def to_torch(params, requires_grad=False):
    out = OrderedDict()
    for k, v in params.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone() if isinstance(v, np.ndarray) else v.clone()
        out[k] = t.requires_grad_(requires_grad)
    return out


# --------------------------------------------------------------------------------------
# ops (each the stock torch op the reference calls at the cited line)
# --------------------------------------------------------------------------------------
def conv3x3x3(x, w, bias=None):
    """model.py:72-73 (`conv`), :336 (conv_input), :348 (conv_output, with bias): k=3, s=1, p=1."""
    return F.conv3d(x, w, bias, stride=1, padding=1)


def conv2x2x2_s2(x, w):
    """model.py:361-363: downsample Conv3d(kernel_size=2, stride=2, bias=False)."""
    return F.conv3d(x, w, None, stride=2, padding=0)


def conv1x1x1(x, w):
    """model.py:393,401: 1x1x1 Conv3d, bias=False."""
    return F.conv3d(x, w, None)


def group_norm(x, gamma, beta):
    """model.py:95-96,338: nn.GroupNorm(8, C), eps 1e-5, affine."""
    return F.group_norm(x, GN_GROUPS, gamma, beta, GN_EPS)


def leaky_relu(x):
    """model.py:93-94,352: LeakyReLU(1e-2)."""
    return F.leaky_relu(x, LEAKY_SLOPE)


def trilinear_up2(x):
    """model.py:12-14: F.interpolate(scale_factor=2, mode='trilinear') (align_corners=False)."""
    return F.interpolate(x, scale_factor=2, mode="trilinear")


def residual(p, prefix, x):
    """model.py:99-117 Residual.forward: x=down(x)?; LReLU(GN(conv)) twice; x + out (no act after)."""
    if prefix + "downsample.0.weight" in p:
        x = conv2x2x2_s2(x, p[prefix + "downsample.0.weight"])
    out = conv3x3x3(x, p[prefix + "conv1.conv1.weight"])
    out = leaky_relu(group_norm(out, p[prefix + "norm1.weight"], p[prefix + "norm1.bias"]))
    out = conv3x3x3(out, p[prefix + "conv2.conv1.weight"])
    out = leaky_relu(group_norm(out, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"]))
    return x + out


def unet_forward(p, x, depth=4, encoder_layers=(1, 2, 2, 4), decoder_layers=(1, 1, 1, 1), taps=None, **_):
    """model.py:407-433 UNet.forward (without its gc.collect()).  ``x``: ``[N,4,D,H,W]`` tensor;
    returns sigmoid probabilities ``[N,n_out,D,H,W]``.  ``taps`` (optional dict) receives named
    intermediates for per-layer checks."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    c = conv3x3x3(x, p["conv_input.weight"])                                   # :412
    c = group_norm(c, p["norm_input.weight"], p["norm_input.bias"])            # :413 (no activation)
    tap("norm_input", c)
    for j in range(encoder_layers[0]):                                         # :414
        c = residual(p, "conv_first.%d." % j, c)
    tap("conv_first", c)
    skips = []
    for i in range(depth - 1):                                                 # :416-418
        skips.append(c)
        for j in range(encoder_layers[i + 1]):
            c = residual(p, "encoder_convs.%d.%d." % (i, j), c)
        tap("enc%d" % i, c)
    for i in reversed(range(depth - 1)):                                       # :420-426
        c = conv1x1x1(trilinear_up2(c), p["upsampling.%d.1.weight" % i])       # :399-402
        c = leaky_relu(c)                                                      # :422
        c = torch.cat([skips[i], c], dim=1)                                    # :424 skip first
        c = conv1x1x1(c, p["decoder_convs1x1.%d.weight" % i])                  # :425
        for j in range(decoder_layers[i]):                                     # :426
            c = residual(p, "decoder_convs.%d.%d." % (i, j), c)
        tap("dec%d" % i, c)
    logits = conv3x3x3(c, p["conv_output.weight"], p["conv_output.bias"])     # :429
    tap("logits", logits)
    return torch.sigmoid(logits)                                               # :431


def dice_loss_joint(pred, gt, priority=1.0):
    """loss.py:105-122: sums over batch and space jointly (dim=(0,2))."""
    n, c = pred.shape[:2]
    pr = pred.reshape(n, c, -1)
    g = gt.reshape(n, c, -1)
    inter = (pr * g).sum(dim=(0, 2)) + 1e-6
    union = (pr ** 2 + g).sum(dim=(0, 2)) + 2e-6
    return priority * (1.0 - torch.mean(2.0 * inter / union))


def bce_loss(pred, gt, bg_weight=1.0):
    """loss.py:70-79."""
    loss = gt * torch.log(pred + 1e-6) + bg_weight * (1.0 - gt) * torch.log((1.0 + 1e-6) - pred)
    return -torch.mean(loss)


def criterion(pred, gt, bg_weight=1e-2, priority=1.0):
    """train.py:203-205 with the criterion list of main.py:126-128: (Dice + BCE(bg 1e-2)) / 2."""
    return (dice_loss_joint(pred, gt, priority) + bce_loss(pred, gt, bg_weight)) / 2.0


def forward_backward(params_np, x_np, g_np, bg_weight=1e-2, threads=None, **cfg):
    """One training forward + criterion + backward (train.py:201-210) on CPU via autograd.
    Returns (probs ndarray, loss float, OrderedDict of grad ndarrays; dead params -> None)."""
    if threads:
        torch.set_num_threads(threads)
    p = to_torch(params_np, requires_grad=True)
    x = torch.from_numpy(x_np)
    g = torch.from_numpy(g_np)
    probs = unet_forward(p, x, **cfg)
    loss = criterion(probs, g, bg_weight)
    loss.backward()
    grads = OrderedDict((k, None if v.grad is None else v.grad.numpy()) for k, v in p.items())
    return probs.detach().numpy(), float(loss), grads


# --------------------------------------------------------------------------------------
# closed forms in float64 numpy -- independent cross-checks of the
# formulas the HIP kernels implement; small sizes only
# --------------------------------------------------------------------------------------
def np_trilinear_matrix(n):
    """Appendix A5: the 2n x n per-axis matrix of scale-2 linear interpolation, align_corners=False
    (model.py:13)."""
    m = np.zeros((2 * n, n))
    for k in range(n):
        m[2 * k, max(k - 1, 0)] += 0.25
        m[2 * k, k] += 0.75
        m[2 * k + 1, k] += 0.75
        m[2 * k + 1, min(k + 1, n - 1)] += 0.25
    return m


def np_trilinear_up2(x):
    x = np.asarray(x, np.float64)
    for ax in (-3, -2, -1):
        m = np_trilinear_matrix(x.shape[ax])
        x = np.moveaxis(np.tensordot(m, np.moveaxis(x, ax, 0), axes=(1, 0)), 0, ax)
    return x


def np_trilinear_up2_bwd(dy):
    dy = np.asarray(dy, np.float64)
    for ax in (-3, -2, -1):
        m = np_trilinear_matrix(dy.shape[ax] // 2).T
        dy = np.moveaxis(np.tensordot(m, np.moveaxis(dy, ax, 0), axes=(1, 0)), 0, ax)
    return dy


def np_group_norm(x, gamma, beta, groups=GN_GROUPS, eps=GN_EPS):
    """Appendix A3 forward; returns (y, mean[n,g], rstd[n,g])."""
    x = np.asarray(x, np.float64)
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mu = xg.mean(-1)
    var = ((xg - mu[..., None]) ** 2).mean(-1)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = ((xg - mu[..., None]) * rstd[..., None]).reshape(x.shape)
    bshape = (1, c) + (1,) * (x.ndim - 2)
    return xh * np.asarray(gamma, np.float64).reshape(bshape) + np.asarray(beta, np.float64).reshape(bshape), mu, rstd


def np_group_norm_bwd(x, gamma, dy, groups=GN_GROUPS, eps=GN_EPS):
    """Appendix A3 backward; returns (dx, dgamma, dbeta)."""
    x = np.asarray(x, np.float64)
    dy = np.asarray(dy, np.float64)
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mu = xg.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(((xg - mu) ** 2).mean(-1, keepdims=True) + eps)
    xh = ((xg - mu) * rstd)
    bshape = (1, c) + (1,) * (x.ndim - 2)
    dyt = (dy * np.asarray(gamma, np.float64).reshape(bshape)).reshape(n, groups, -1)
    dx = rstd * (dyt - dyt.mean(-1, keepdims=True) - xh * (dyt * xh).mean(-1, keepdims=True))
    xh = xh.reshape(x.shape)
    red = (0,) + tuple(range(2, x.ndim))
    return dx.reshape(x.shape), (dy * xh).sum(red), dy.sum(red)


def np_dice_bce_sums(p, g, bg_weight=1e-2):
    """Appendix A7 phase 1: per-class (I_c, U_c) WITHOUT epsilons and the BCE sum, float64."""
    p32 = np.asarray(p, np.float32)
    p = p32.astype(np.float64)
    g = np.asarray(g, np.float64)
    red = (0,) + tuple(range(2, p.ndim))
    inter = (p * g).sum(red)
    union = (p * p + g).sum(red)
    # the two log arguments are formed in float32 exactly as loss.py:76-77 forms them (matters only
    # at saturated p: float32(1+1e-6) - 1 = 9.54e-7, not 1e-6)
    a = (p32 + np.float32(1e-6)).astype(np.float64)
    b = (np.float32(1.0 + 1e-6) - p32).astype(np.float64)
    bce = (g * np.log(a) + bg_weight * (1 - g) * np.log(b)).sum()
    return inter, union, bce


def np_criterion_from_sums(inter, union, bce, count, priority=1.0):
    """loss value from GLOBAL sums (epsilons added after any cross-rank reduce)."""
    dice = priority * (1.0 - np.mean(2.0 * (inter + 1e-6) / (union + 2e-6)))
    return 0.5 * (dice + (-bce / count)), dice, -bce / count


def np_criterion_grad(p, g, inter, union, count, bg_weight=1e-2, priority=1.0):
    """Appendix A7: d[(Dice+BCE)/2]/dp given GLOBAL sums and GLOBAL element count."""
    p32 = np.asarray(p, np.float32)
    p = p32.astype(np.float64)
    g = np.asarray(g, np.float64)
    c = p.shape[1]
    bshape = (1, c) + (1,) * (p.ndim - 2)
    i_c = (inter + 1e-6).reshape(bshape)
    u_c = (union + 2e-6).reshape(bshape)
    d_dice = -priority * (2.0 / c) * (g * u_c - 2.0 * p * i_c) / (u_c * u_c)
    a = (p32 + np.float32(1e-6)).astype(np.float64)          # float32-formed, see np_dice_bce_sums
    b = (np.float32(1.0 + 1e-6) - p32).astype(np.float64)
    d_bce = -(g / a - bg_weight * (1 - g) / b) / count
    return 0.5 * (d_dice + d_bce)


def np_adam_amsgrad_step(w, g, m, v, vmax, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-6):
    """torch.optim.Adam(amsgrad=True, weight_decay=wd) single-tensor update (main.py:133-137:
    lr 2e-5, wd 1e-6, amsgrad) -- L2 decay added to the gradient.  ``step`` is 1-based.
    float64 in, float64 out; returns (w, m, v, vmax)."""
    g = g + weight_decay * w
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    vmax = np.maximum(vmax, v)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(vmax) / math.sqrt(bc2) + eps
    return w - (lr / bc1) * m / denom, m, v, vmax


def step_lr(base_lr, step, step_size=16000, gamma=0.5):
    """StepLR(step_size=16000, gamma=0.5) stepped per iteration (main.py:139-142, train.py:222-223)."""
    return base_lr * gamma ** (step // step_size)


# --------------------------------------------------------------------------------------
# evaluation yardstick (metrics.py:108-133) and tiling helpers (loader_helper.py:34-97)
# --------------------------------------------------------------------------------------
def dice_metric(pred, gt):
    """metrics.Dice.update semantics (metrics.py:108-133): threshold 0.5; per sample and channel
    2*sum(p*g)/sum(p+g) over space, NaN -> 1; mean over the batch.  Returns [C]."""
    p = (np.asarray(pred) > 0.5).astype(np.float64)
    g = (np.asarray(gt) > 0.5).astype(np.float64)
    red = tuple(range(2, p.ndim))
    num = (p * g).sum(red).astype(np.float32)          # the reference's sums are float32 tensors (exact: counts < 2^24)
    den = (p + g).sum(red).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = (2 * num / den).astype(np.float64)          # float32 division (metrics.py:126), stored into a float64 array
    d[np.isnan(d)] = 1.0
    return d.mean(axis=0)


def tile_indices(position, center_shape, border):
    """loader_helper.py:34-40 get_indices: tile = centre block `position` grown by `border`."""
    index_min = [p * c - b for p, c, b in zip(position, center_shape, border)]
    index_max = [(p + 1) * c + b for p, c, b in zip(position, center_shape, border)]
    return index_min, index_max


def tile_copy(data, tile_shape, index_min, index_max):
    """loader_helper.py:42-80 copy: zero-padded extract of data[..., min:max] into a tile."""
    tile = np.zeros(tuple(data.shape[:2]) + tuple(tile_shape), data.dtype)
    src, dst = [], []
    for a in range(3):
        lo = max(index_min[a], 0)
        hi = min(index_max[a], data.shape[2 + a])
        src.append(slice(lo, hi))
        dst.append(slice(lo - index_min[a], hi - index_min[a]))
    tile[(slice(None), slice(None)) + tuple(dst)] = data[(slice(None), slice(None)) + tuple(src)]
    return tile


def tile_copy_back(data, tile, center_shape, index_min, index_max, border):
    """loader_helper.py:82-97 copy_back: paste the tile's centre block into data (clipped)."""
    src, dst = [], []
    for a in range(3):
        lo = index_min[a] + border[a]
        hi = min(index_max[a] - border[a], data.shape[2 + a])
        dst.append(slice(lo, hi))
        src.append(slice(border[a], border[a] + (hi - lo)))
    data[(slice(None), slice(None)) + tuple(dst)] = tile[(slice(None), slice(None)) + tuple(src)]
    return data


# --------------------------------------------------------------------------------------
# inference driver around the forward pass (test.py:47-164) -- restated for the section 8(f) rank-1 row
# --------------------------------------------------------------------------------------
def closest_to_k(n, k=16):
    """loader_helper.py:99-103: round n up to a multiple of k."""
    return n if n % k == 0 else (n // k + 1) * k


def bbox3(img):
    """loader_helper.py:105-129: [[min],[max]] indices of the non-zero voxels per axis ([[-1,-1,-1],[0,0,0]] when empty)."""
    nz = np.nonzero(np.asarray(img))
    if nz[0].size == 0:
        return np.array([[-1, -1, -1], [0, 0, 0]])
    return np.array([[a.min() for a in nz], [a.max() for a in nz]])


def get_bbox(data):
    """test.py:47-49: union of the per-modality bounding boxes (max index is used as an EXCLUSIVE slice end at :87)."""
    boxes = np.stack([bbox3(d) for d in data], axis=0)
    return np.stack([boxes[:, 0].min(axis=0), boxes[:, 1].max(axis=0)], axis=0)


def pad_to_multiple(image, k=16):
    """test.py:92-99: symmetric zero padding of [C,D,H,W] to multiples of k; returns (padded, pad_left, pad_right)."""
    old = np.array(image.shape[1:])
    new = np.array([closest_to_k(int(i), k) for i in old])
    diff = new - old
    left = diff // 2
    right = diff - left
    padded = np.pad(image, ((0, 0),) + tuple((int(left[i]), int(right[i])) for i in range(3)), mode="constant", constant_values=0)
    return padded, left, right


def zscore_nonzero(x):
    """test.py:103-113: per-channel mean/std over the voxels > 0 (moments accumulated as sum(x / n)), applied to ALL voxels."""
    x = np.asarray(x)
    n = (x > 0).sum(axis=(1, 2, 3))
    mean = np.sum(x / n[:, None, None, None], axis=(1, 2, 3))
    mean2 = np.sum(np.square(x) / n[:, None, None, None], axis=(1, 2, 3))
    std = np.sqrt(mean2 - mean * mean)
    return (x - mean.reshape(-1, 1, 1, 1)) / std.reshape(-1, 1, 1, 1)


TTA_FLIPS = ((), (1,), (2,), (1, 2))      # test.py:117-120: axes of the [C,D,H,W] array that are reversed


def tta_inputs(x):
    """test.py:115-120: the four flipped copies of a [C,D,H,W] volume."""
    return [np.ascontiguousarray(np.flip(x, axis=ax)) if ax else x for ax in TTA_FLIPS]


def tta_merge(outputs):
    """test.py:134-138: un-flip each prediction and average: sum(outputs) / len(outputs) (left-to-right float32 sum)."""
    un = [np.flip(o, axis=ax) if ax else o for o, ax in zip(outputs, TTA_FLIPS)]
    acc = un[0]
    for o in un[1:]:
        acc = acc + o
    return acc / len(un)


def compose_labels(prob):
    """test.py:144-159: threshold 0.5; label 2 = WT, overwritten by 1 = TC, overwritten by 4 = ET if more than 32 ET voxels."""
    m = np.asarray(prob) > 0.5
    wt, tc, et = m[0], m[1], m[2]
    out = np.zeros(wt.shape, np.uint8)
    out[wt] = 2
    out[tc] = 1
    if et.sum() > 32:
        out[et] = 4
    return out, (int(wt.sum()), int(tc.sum()), int(et.sum()))


def reject_small_regions(connectivity, ratio=0.25):
    """test.py:51-62: zero every connected region (label of `connectivity`, background included) whose voxel count is below
    ratio * (all voxels - voxels of the most frequent label)."""
    out = connectivity.copy()
    unique, counts = np.unique(connectivity, return_counts=True)
    nonzero = connectivity.size - counts.max()
    for u, c in zip(unique, counts):
        if c < ratio * nonzero:
            out[out == u] = 0
    return out


def label_components(mask):
    """skimage.morphology.label default (test.py:162): full connectivity (26-neighbourhood in 3-D); here scipy.ndimage."""
    import scipy.ndimage as ndi
    lab, _ = ndi.label(mask, structure=np.ones((3, 3, 3), dtype=bool))
    return lab


def postprocess(prob):
    """test.py:144-164 on the un-padded probability volume [3,D,H,W] -> uint8 labels."""
    labels, vols = compose_labels(prob)
    clusters = reject_small_regions(label_components(labels > 0), 0.1)
    labels[clusters == 0] = 0
    return labels, vols


# ====================================================================== training input pipeline (dataloader.py:100-216)
# The reference's SimpleReader draws its parameters from the global `random` / `numpy.random` streams and transforms with
# numpy + scipy.ndimage.affine_transform (third-party, scipy; image has 1.15.3).  Restated with EXPLICIT parameters; the closed
# form of affine_transform(order=1, mode='reflect') with a diagonal matrix -- linear interpolation on the half-sample-symmetric
# extension, input coordinate = scale * output index -- was checked against scipy here to 2e-16.
import random as _pyrandom


def zscore_positive(image):
    """dataloader.py:124-134 (and :259-266): per channel, the voxel COUNT is over x > 0 but the sums run over ALL voxels;
    every voxel is normalised.  float64 like the reference (num_voxels is int64).  Returns (normalised, mean, std)."""
    image = np.asarray(image)
    nv = (image > 0).sum(axis=(1, 2, 3))
    mean = np.sum(image / nv[:, None, None, None], axis=(1, 2, 3))
    mean2 = np.sum(np.square(image) / nv[:, None, None, None], axis=(1, 2, 3))
    std = np.sqrt(mean2 - mean * mean)
    return (image - mean.reshape(-1, 1, 1, 1)) / std.reshape(-1, 1, 1, 1), mean, std


def label_bbox(label, patch_size):
    """dataloader.py:104-116 (__cache): bounding box of label > 0, grown by 50 voxels, clipped so a patch fits."""
    bbox = bbox3(np.asarray(label) > 0).astype(np.float64)
    shape = np.array(np.asarray(label).shape)
    low = np.array(patch_size) / 2.0 + 1
    high = shape - np.array(patch_size) / 2.0 - 1
    bbox[0] = np.maximum(bbox[0] - 50, low)
    bbox[1] = np.minimum(bbox[1] + 50, high)
    return bbox


def draw_augment_params(bbox, patch_size, channels=4):
    """dataloader.py:141-199: the draws of SimpleReader.__getitem__, in its order, from the same global generators
    (`numpy.random`: crop centre, intensity gain / bias; `random`: the unused sigma and alpha, the three zoom factors,
    three flips, the transpose)."""
    center = np.random.rand(3)
    center = center * (bbox[1] - bbox[0]) + bbox[0]
    left_bottom = (center - np.array(patch_size) / 2.0).astype(np.int32)
    _pyrandom.random()                                   # sigma (:157), only used by the commented-out elastic transform
    _pyrandom.random()                                   # alpha (:158)
    scale = [0.7 + _pyrandom.random() * 0.6 for _ in range(3)]
    flips = [_pyrandom.random() > 0.5 for _ in range(3)]
    transpose = _pyrandom.random() > 0.5
    gain = np.random.uniform(0.9, 1.1, size=(channels, 1, 1, 1))
    bias = np.random.uniform(-0.2, 0.2, size=(channels, 1, 1, 1))
    return dict(crop_lo=left_bottom, scale=np.array(scale), flips=flips, transpose=transpose,
                gain=gain.reshape(-1), bias=bias.reshape(-1))


def reflect_index(i, n):
    """half-sample symmetric extension (d c b a | a b c d | d c b a): index of the source sample"""
    i = np.mod(i, 2 * n)
    return np.where(i < n, i, 2 * n - 1 - i)


def zoom_linear_reflect(vol, scale):
    """scipy.ndimage.affine_transform(vol, (1, sx, sy, sz), order=1, mode='reflect') for vol [C, D, H, W]; float64."""
    vol = np.asarray(vol, np.float64)
    _, D, H, W = vol.shape
    idx, wgt = [], []
    for n, s in zip((D, H, W), scale):
        x = np.arange(n, dtype=np.float64) * s
        i0 = np.floor(x).astype(np.int64)
        t = x - i0
        idx.append((reflect_index(i0, n), reflect_index(i0 + 1, n)))
        wgt.append((1.0 - t, t))
    out = np.zeros_like(vol)
    for a in range(2):
        for b in range(2):
            for c in range(2):
                w = wgt[0][a][:, None, None] * wgt[1][b][None, :, None] * wgt[2][c][None, None, :]
                out += w[None] * vol[:, idx[0][a]][:, :, idx[1][b]][:, :, :, idx[2][c]]
    return out


def augment_patch(image_norm, label, crop_lo, patch_size, scale, flips, transpose, gain, bias):
    """dataloader.py:147-205 with explicit parameters: crop, zoom (data and one-hot label), flips, transpose, intensity,
    WT/TC/ET targets.  Returns (data [C,...], target [3,...]) float32."""
    lo, ps = [int(v) for v in crop_lo], [int(v) for v in patch_size]
    sl = tuple(slice(l, l + p) for l, p in zip(lo, ps))
    data = np.asarray(image_norm)[(slice(None),) + sl]
    lab = np.asarray(label)[sl]
    onehot = np.eye(4)[lab.astype(np.int32)].transpose((3, 0, 1, 2))
    data = zoom_linear_reflect(data, scale)
    onehot = zoom_linear_reflect(onehot, scale)
    for ax, f in enumerate(flips):
        if f:
            data = np.flip(data, axis=ax + 1)
            onehot = np.flip(onehot, axis=ax + 1)
    if transpose:
        data = data.transpose((0, 2, 1, 3))
        onehot = onehot.transpose((0, 2, 1, 3))
    data = data * np.asarray(gain).reshape(-1, 1, 1, 1) + np.asarray(bias).reshape(-1, 1, 1, 1)
    wt = onehot[1:].sum(axis=0, keepdims=True)
    tc = onehot[[1, 3]].sum(axis=0, keepdims=True)
    et = onehot[3, None]
    return np.ascontiguousarray(data, np.float32), np.concatenate([wt, tc, et], axis=0).astype(np.float32)


def full_volume_item(image, label, k=16):
    """dataloader.py:243-283 (FullReader.__getitem__): zero-pad to multiples of k, z-score, hard WT/TC/ET targets."""
    image, label = np.asarray(image), np.asarray(label)
    new_shape = tuple(closest_to_k(i, k) for i in image.shape[1:])
    img = np.zeros((image.shape[0],) + new_shape, np.float32)
    lab = np.zeros(new_shape, np.float32)
    img[(slice(None),) + tuple(slice(0, s) for s in image.shape[1:])] = image
    lab[tuple(slice(0, s) for s in label.shape)] = label
    norm, _, _ = zscore_positive(img)
    onehot = np.eye(4)[lab.astype(np.int32)].transpose((3, 0, 1, 2))
    tgt = np.concatenate([onehot[1:].sum(axis=0, keepdims=True), onehot[[1, 3]].sum(axis=0, keepdims=True), onehot[3, None]], axis=0)
    return norm.astype(np.float32), tgt.astype(np.float32)


def make_dataloader_case(seed=77, shape=(48, 56, 40)):
    """Synthetic multimodal case for the input-pipeline fixtures: 4 positive modalities inside an ellipsoid "brain" (zeros
    outside, like skull-stripped BraTS data) and a nested 2 / 1 / 3 label blob.  Inputs only."""
    r = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    c = [s / 2.0 for s in shape]
    brain = ((zz - c[0]) / (0.42 * shape[0])) ** 2 + ((yy - c[1]) / (0.42 * shape[1])) ** 2 + ((xx - c[2]) / (0.42 * shape[2])) ** 2 < 1.0
    image = (np.abs(r.standard_normal((4,) + tuple(shape))) * 120 + 40).astype(np.float32) * brain[None]
    d2 = ((zz - c[0] - 2) / (0.15 * shape[0])) ** 2 + ((yy - c[1] + 3) / (0.16 * shape[1])) ** 2 + ((xx - c[2] - 1) / (0.14 * shape[2])) ** 2
    label = np.zeros(shape, np.float32)
    label[d2 < 1.0] = 2
    label[d2 < 0.5] = 1
    label[d2 < 0.2] = 3
    return image, label
