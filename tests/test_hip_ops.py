"""GPU parity, op level: every entry point of libresunet_hip.so is called through the C-ABI (brats2019_amd.ops
-> ctypes) and compared (a) with the reference's own outputs in tests/golden/ops.npz and (b) with the CPU
oracle on further seeded shapes (ragged extents, channel counts of every network level).

Tolerances (float32 path, exact-f32 MFMA; only the summation ORDER differs from the CPU):
forward tensors  |d| <= 1e-5 + 1e-5*|ref|   (K <= 3456 products per output)
reductions over >= 1e3 voxels (weight grads, GN parameter grads) relative 2e-4 of the tensor's max."""
import numpy as np
import pytest
import torch

from oracle import resunet_oracle as O

pytestmark = pytest.mark.gpu

T = torch.from_numpy


def dev(a):
    return T(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, ref, rtol=1e-5, atol=1e-5, name=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    tol = atol + rtol * np.abs(ref)
    bad = err > tol
    assert not bad.any(), "%s: %d/%d mismatches, max err %.3e (ref %.3e) at %s" % (
        name, bad.sum(), bad.size, err.max(), np.abs(ref).max(), np.unravel_index(err.argmax(), err.shape))


def close_rel_max(got, ref, rel=2e-4, name=""):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref).max()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e" % (name, err, scale)


@pytest.fixture(scope="module")
def ops():
    from brats2019_amd import ops as m
    return m


@pytest.mark.parametrize("tag,bias", [("c3_4_16", False), ("c3_16_16", False), ("c3_32_32", False), ("c3_16_3b", True)])
def test_conv3_golden(golden, ops, tag, bias):
    g = golden("ops")
    x, w, dy = dev(g[tag + "_x"]), dev(g[tag + "_w"]), dev(g[tag + "_dy"])
    b = dev(g[tag + "_b"]) if bias else None
    close(ops.conv3d(x, w, b), g[tag + "_y"], name=tag + " fwd")
    close(ops.conv3d_bwd_data(dy, w, x.shape[2:]), g[tag + "_dx"], name=tag + " dgrad")
    if bias:
        dw, db = ops.conv3d_bwd_weight(x, dy, 3, with_bias=True)
        close_rel_max(db, g[tag + "_db"], name=tag + " db")
    else:
        dw = ops.conv3d_bwd_weight(x, dy, 3)
    close_rel_max(dw, g[tag + "_dw"], name=tag + " wgrad")


@pytest.mark.parametrize("tag,k", [("c2s2_16_32", 2), ("c1_32_16", 1), ("c1_128_64", 1)])
def test_small_convs_golden(golden, ops, tag, k):
    g = golden("ops")
    x, w, dy = dev(g[tag + "_x"]), dev(g[tag + "_w"]), dev(g[tag + "_dy"])
    close(ops.conv3d(x, w), g[tag + "_y"], name=tag + " fwd")
    close(ops.conv3d_bwd_data(dy, w, x.shape[2:]), g[tag + "_dx"], name=tag + " dgrad")
    close_rel_max(ops.conv3d_bwd_weight(x, dy, k), g[tag + "_dw"], name=tag + " wgrad")


CONV3_CASES = [  # (N, Cin, Cout, D, H, W): every level's channel pair, ragged / non-multiple-of-tile extents, W % 4 != 0
    (1, 4, 16, 12, 20, 36), (2, 16, 16, 9, 17, 33), (1, 32, 32, 8, 16, 32), (1, 64, 64, 8, 8, 16),
    (1, 128, 128, 4, 8, 16), (1, 16, 3, 7, 9, 18), (1, 8, 8, 6, 6, 6), (2, 3, 16, 5, 8, 16), (1, 16, 32, 4, 4, 4),
    (1, 32, 32, 32, 32, 32),
]


@pytest.mark.parametrize("case", CONV3_CASES)
def test_conv3_vs_oracle(ops, case):
    n, cin, cout, d, h, w = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, cin, d, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)
    dy = rng.standard_normal((n, cout, d, h, w)).astype(np.float32)
    xt, wtt = T(x).requires_grad_(True), T(wt).requires_grad_(True)
    y = O.conv3x3x3(xt, wtt)
    y.backward(T(dy))
    close(ops.conv3d(dev(x), dev(wt)), y, name="fwd %s" % (case,))
    close(ops.conv3d_bwd_data(dev(dy), dev(wt), (d, h, w)), xt.grad, atol=2e-5, name="dgrad %s" % (case,))
    close_rel_max(ops.conv3d_bwd_weight(dev(x), dev(dy), 3), wtt.grad.numpy(), name="wgrad %s" % (case,))


@pytest.mark.parametrize("case", [(2, 16, 32, 8, 12, 16, 2), (1, 64, 128, 4, 4, 8, 2), (1, 8, 16, 2, 6, 6, 2),
                                  (1, 64, 32, 5, 7, 9, 1), (2, 32, 16, 8, 8, 8, 1), (1, 16, 8, 3, 3, 3, 1)])
def test_small_convs_vs_oracle(ops, case):
    n, cin, cout, d, h, w, k = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, cin, d, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k, k)) / np.sqrt(k ** 3 * cin)).astype(np.float32)
    xt, wtt = T(x).requires_grad_(True), T(wt).requires_grad_(True)
    y = O.conv2x2x2_s2(xt, wtt) if k == 2 else O.conv1x1x1(xt, wtt)
    dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(dy))
    close(ops.conv3d(dev(x), dev(wt)), y, name="fwd %s" % (case,))
    close(ops.conv3d_bwd_data(dev(dy), dev(wt), (d, h, w)), xt.grad, atol=2e-5, name="dgrad %s" % (case,))
    close_rel_max(ops.conv3d_bwd_weight(dev(x), dev(dy), k), wtt.grad.numpy(), name="wgrad %s" % (case,))


@pytest.mark.parametrize("tag", ["gn16", "gn128"])
def test_group_norm_golden(golden, ops, tag):
    g = golden("ops")
    x, gamma, beta, dy = dev(g[tag + "_x"]), dev(g[tag + "_gamma"]), dev(g[tag + "_beta"]), dev(g[tag + "_dy"])
    y, mean, rstd = ops.group_norm(x, gamma, beta, slope=1.0)
    close(y, g[tag + "_y"], atol=2e-6, name=tag + " y")
    close(mean, g[tag + "_mean"].ravel(), atol=1e-6, name=tag + " mean")
    close(rstd, g[tag + "_rstd"].ravel(), rtol=1e-5, name=tag + " rstd")
    dx, dgam, dbet = ops.group_norm_bwd(x, gamma, beta, mean, rstd, dy, slope=1.0)
    close(dx, g[tag + "_dx"], rtol=1e-4, atol=5e-6, name=tag + " dx")
    close_rel_max(dgam, g[tag + "_dgamma"], name=tag + " dgamma")
    close_rel_max(dbet, g[tag + "_dbeta"], name=tag + " dbeta")


def test_group_norm_act_residual_vs_oracle(ops):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 32, 6, 10, 14)) * 2 + 0.5).astype(np.float32)
    res = rng.standard_normal(x.shape).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, 32).astype(np.float32), rng.uniform(-0.5, 0.5, 32).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    xt, gt, bt, rt = (T(a).requires_grad_(True) for a in (x, gamma, beta, res))
    y = rt + O.leaky_relu(O.group_norm(xt, gt, bt))
    y.backward(T(dy))
    yh, mean, rstd = ops.group_norm(dev(x), dev(gamma), dev(beta), slope=0.01, residual=dev(res))
    close(yh, y, atol=3e-6, name="gn+lrelu+res")
    dx, dgam, dbet = ops.group_norm_bwd(dev(x), dev(gamma), dev(beta), mean, rstd, dev(dy), slope=0.01)
    close(dx, xt.grad, rtol=1e-4, atol=5e-6, name="dx")
    close_rel_max(dgam, gt.grad.numpy(), name="dgamma")
    close_rel_max(dbet, bt.grad.numpy(), name="dbeta")


def test_leaky_relu_sigmoid_golden(golden, ops):
    g = golden("ops")
    y = ops.leaky_relu(dev(g["lrelu_x"]))
    assert np.array_equal(y.cpu().numpy(), g["lrelu_y"])                       # bit-exact elementwise
    assert np.array_equal(ops.leaky_relu_bwd(y, dev(g["lrelu_dy"])).cpu().numpy(), g["lrelu_dx"])
    close(ops.sigmoid(dev(g["sig_x"])), g["sig_y"], rtol=1e-6, atol=1e-7, name="sigmoid")


def test_trilinear_golden_and_ragged(golden, ops):
    g = golden("ops")
    close(ops.upsample2x(dev(g["up_x"])), g["up_y"], atol=1e-6, name="up fwd")
    close(ops.upsample2x_bwd(dev(g["up_dy"])), g["up_dx"], atol=1e-5, name="up bwd")
    rng = np.random.default_rng(9)
    for shape in [(1, 2, 1, 1, 1), (1, 3, 2, 1, 5), (2, 4, 8, 8, 8)]:
        x = rng.standard_normal(shape).astype(np.float32)
        xt = T(x).requires_grad_(True)
        y = O.trilinear_up2(xt)
        dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(T(dy))
        close(ops.upsample2x(dev(x)), y, atol=1e-6, name="up fwd %s" % (shape,))
        close(ops.upsample2x_bwd(dev(dy)), xt.grad, atol=1e-5, name="up bwd %s" % (shape,))


def test_criterion_golden(golden, ops):
    g = golden("loss")
    p, gt = dev(g["p"]), dev(g["g"])
    count = float(p.numel())
    for tag, wd, wb, bgw in (("dice", 1.0, 0.0, 1.0), ("bce", 0.0, 1.0, 1e-2), ("bce_w1", 0.0, 1.0, 1.0), ("crit", 0.5, 0.5, 1e-2)):
        sums = ops.criterion_sums(p, gt, bgw)
        dice, bce = ops.criterion_value(sums, count)
        loss = float(wd * dice + wb * bce)
        assert abs(loss - float(g[tag + "_loss"])) < 2e-6, (tag, loss, float(g[tag + "_loss"]))
        dp = ops.criterion_grad(p, gt, sums, count, wd, wb, bgw, 1.0)
        close(dp, g[tag + "_dp"], rtol=2e-5, atol=1e-9, name=tag + " dp")
    # closed-form sums vs oracle
    inter, union, bce = O.np_dice_bce_sums(g["p"], g["g"], 1e-2)
    s = ops.criterion_sums(p, gt, 1e-2).cpu().numpy()
    np.testing.assert_allclose(s[:3], inter, rtol=1e-6)
    np.testing.assert_allclose(s[3:6], union, rtol=1e-6)
    np.testing.assert_allclose(s[6], bce, rtol=1e-5)


def test_loss_modules_autograd(golden):
    from brats2019_amd import loss as L
    g = golden("loss")
    for tag, crit in (("dice", [L.Dice_loss_joint(index=0, priority=1)]), ("bce", [L.BCE_Loss(index=0, bg_weight=1e-2)]),
                      ("crit", [L.Dice_loss_joint(), L.BCE_Loss(bg_weight=1e-2)]), ("crit", [L.FusedCriterion()])):
        p = dev(g["p"]).requires_grad_(True)
        vals = [c([p], [dev(g["g"])]) for c in crit]
        lv = sum(vals) / len(vals)                       # train.py:203-205
        lv.backward()
        assert abs(float(lv) - float(g[tag + "_loss"])) < 2e-6
        close(p.grad, g[tag + "_dp"], rtol=2e-5, atol=1e-9, name=tag + " module dp")


def test_adam_golden(golden, ops):
    g = golden("adam")
    w = dev(g["w0"])
    m, v, vmax = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
    for i in range(g["grads"].shape[0]):
        ops.adam_amsgrad_step(w, dev(g["grads"][i]), m, v, vmax, i + 1, float(g["lrs"][i]), weight_decay=1e-6)
        close(w, g["traj"][i], rtol=1e-6, atol=1e-7, name="adam step %d" % i)


@pytest.mark.parametrize("tag", ["res16", "res32d"])
def test_residual_module_golden(golden, tag):
    """model.Residual built from the per-op autograd functions (stand-alone use of the building blocks)."""
    from brats2019_amd import model as M
    g = golden("residual")
    params = {k[len(tag) + 3:]: v for k, v in g.items() if k.startswith(tag + "_p_")}
    down = None
    c = params["conv1.conv1.weight"].shape[0]
    if "downsample.0.weight" in params:
        cin = params["downsample.0.weight"].shape[1]
        down = torch.nn.Sequential(torch.nn.Conv3d(cin, c, kernel_size=2, stride=2, bias=False))
    blk = M.Residual(in_channels=c, out_channels=c, stride=1, downsample=down)
    blk.load_state_dict({k: T(v) for k, v in params.items()})
    blk.cuda()
    x = dev(g[tag + "_x"]).requires_grad_(True)
    y = blk(x)
    y.backward(dev(g[tag + "_dy"]))
    close(y, g[tag + "_y"], atol=2e-5, name=tag + " y")
    close(x.grad, g[tag + "_dx"], rtol=1e-4, atol=2e-5, name=tag + " dx")
    for k, p in blk.named_parameters():
        close_rel_max(p.grad, g["%s_g_%s" % (tag, k)], rel=3e-4, name="%s grad %s" % (tag, k))


# ---------------------------------------------------------------- split-bf16 (bf16x3) convolution path
# hi+lo operands carry 16 mantissa bits -> each product is accurate to ~2^-16; over K <= 3456 products of random sign the
# output error is ~2^-16 * sqrt(K) * rms(x) * rms(w): we hold max|err| <= 1e-4 * max|ref| (plain bf16 operands: ~5e-3).
@pytest.mark.parametrize("case", [(1, 4, 16, 12, 20, 32), (2, 16, 16, 9, 17, 32), (1, 32, 32, 8, 16, 32), (1, 64, 64, 8, 8, 16),
                                  (1, 128, 128, 4, 8, 16), (1, 16, 3, 7, 9, 20), (1, 8, 8, 6, 6, 8), (2, 3, 16, 5, 8, 16),
                                  (1, 16, 16, 32, 32, 32), (1, 24, 40, 6, 10, 12)])
def test_conv3_bf16x3_vs_oracle(ops, case):
    n, cin, cout, d, h, w = case
    rng = np.random.default_rng(sum(case) + 1)
    x = rng.standard_normal((n, cin, d, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    dy = rng.standard_normal((n, cout, d, h, w)).astype(np.float32)
    xt, wtt = T(x).requires_grad_(True), T(wt)
    y = O.conv3x3x3(xt, wtt, T(b))
    y.backward(T(dy))
    close_rel_max(ops.conv3d(dev(x), dev(wt), dev(b), precision="bf16x3"), y.detach().numpy(), rel=1e-4, name="bf16x3 fwd %s" % (case,))
    close_rel_max(ops.conv3d_bwd_data(dev(dy), dev(wt), (d, h, w), precision="bf16x3"), xt.grad.numpy(), rel=1e-4, name="bf16x3 dgrad %s" % (case,))
    wtt2 = T(wt).requires_grad_(True)
    O.conv3x3x3(T(x), wtt2).backward(T(dy))
    close_rel_max(ops.conv3d_bwd_weight(dev(x), dev(dy), 3, precision="bf16x3"), wtt2.grad.numpy(), rel=2e-4, name="bf16x3 wgrad %s" % (case,))


def test_conv3_bf16x3_is_not_plain_bf16(ops):
    """The lo terms matter: the result must be far closer to fp32 than single-bf16 operands could be."""
    rng = np.random.default_rng(77)
    x = rng.standard_normal((1, 16, 8, 16, 32)).astype(np.float32)
    wt = (rng.standard_normal((16, 16, 3, 3, 3)) / np.sqrt(432)).astype(np.float32)
    ref = O.conv3x3x3(T(x), T(wt)).numpy().astype(np.float64)
    got = ops.conv3d(dev(x), dev(wt), precision="bf16x3").cpu().numpy()
    plain = O.conv3x3x3(T(x).bfloat16().float(), T(wt).bfloat16().float()).numpy()
    e_split, e_plain = np.abs(got - ref).max(), np.abs(plain - ref).max()
    assert e_split < e_plain / 50, (e_split, e_plain)


def test_conv3_bf16x3_ragged_w_falls_back_to_f32(ops):
    rng = np.random.default_rng(78)
    x = rng.standard_normal((1, 16, 4, 6, 18)).astype(np.float32)        # W % 4 != 0
    wt = (rng.standard_normal((16, 16, 3, 3, 3)) / np.sqrt(432)).astype(np.float32)
    close(ops.conv3d(dev(x), dev(wt), precision="bf16x3"), O.conv3x3x3(T(x), T(wt)), name="fallback")


def test_dice_metric_matches_reference(golden):
    """metrics.Dice (metrics.py:101-133) vs the value the reference's own class produced (tests/golden/loss.npz)."""
    from brats2019_amd import metrics as MT
    g = golden("loss")
    m = MT.Dice(classes=4)
    m.update([dev(g["g"])], [dev(g["p"])])
    np.testing.assert_array_equal(m.get(), g["metric_dice"])
    m.update([dev(g["g"])], [dev(g["g"])])                 # perfect prediction -> 1 in every class; mean of the two updates
    np.testing.assert_allclose(m.get(), (g["metric_dice"] + 1.0) / 2, rtol=0, atol=1e-12)
    empty = torch.zeros(1, 3, 4, 4, 4).cuda()
    m.reset()
    m.update([empty], [empty])                              # 0/0 -> NaN -> 1 (metrics.py:128)
    assert np.array_equal(m.get(), np.ones(3))
