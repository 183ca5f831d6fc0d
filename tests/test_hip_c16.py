"""Voxel-major working layout ("C16", include/resunet_hip.h): the layout-aware kernels must give the same numbers as
their NCDHW forms -- compared here through the C-ABI hooks, on the same seeded inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda()


def test_layout_roundtrip():
    from brats2019_amd import ops
    x = _rand(2, 32, 6, 10, 12)
    c = ops.to_c16(x)
    ref = x.view(2, 2, 16, 6, 10, 12).permute(0, 1, 3, 4, 5, 2).contiguous()
    assert torch.equal(c, ref)
    assert torch.equal(ops.from_c16(c), x)


@pytest.mark.parametrize("in16,out16", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("shape", [(4, 16, 16, 32, 32, 64), (1, 32, 32, 64, 64, 64), (2, 16, 16, 8, 128, 128)])
def test_conv3_layouts_equal_ncdhw(shape, in16, out16):
    """same split-bf16 arithmetic, different storage: identical K-order per output -> bit-equal results.  The last shape
    (N * tiles per z-layer a multiple of 256) takes the z-walk tile order of the persistent kernel; every shape is also
    checked against the exact-f32 kernel, which has its own tiling."""
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    x = _rand(n, cin, d, h, w, seed=1)
    wt = _rand(cout, cin, 3, 3, 3, seed=2) * 0.1
    ref = ops.conv3d(x, wt, precision="bf16x3")
    y = ops.conv3d_layout(ops.to_c16(x) if in16 else x, wt, in_c16=in16, out_c16=out16)
    if out16:
        y = ops.from_c16(y)
    exact = ops.conv3d(x, wt, precision="f32")
    if in16 and out16 and cin >= 32:
        # voxel-major on both sides with 32+ channels: the Winograd-z kernel (conv3_wz.hpp) -- other products, same 2^-17 class of error
        assert float((y - ref).abs().max()) <= 1e-4 * float(exact.abs().max())
    else:
        assert torch.equal(y, ref), float((y - ref).abs().max())
    assert float((y - exact).abs().max()) <= 2e-4 * float(exact.abs().max())


HEAD_SHAPES = [(2, 3, 32, 32, 64), (1, 3, 128, 128, 128), (4, 4, 22, 36, 40), (2, 1, 17, 60, 72), (2, 2, 8, 128, 128)]


@pytest.mark.parametrize("shape", HEAD_SHAPES, ids=["%dx16-%d_%dx%dx%d" % s for s in HEAD_SHAPES])
def test_conv3_head_form_equals_the_sixteen_column_kernel_class(shape):
    """The head convolution (model.py:348: 16 -> 3 channels, voxel-major in, NCDHW out) packs (row tap, output channel) pairs into the 16 matrix
    columns (conv3_sb_common.hpp, sb_head_shape: 150 instead of 336 MFMAs per tile).  Same products, another summation order: held to a float64
    convolution of the same operands within the bar of the 16-column kernel (RU_HEAD_FORM=0) on the same inputs, and the two must differ
    (the shape really took the other kernel).  Shapes: full tiles, the 128^3 level, ragged extents on every face, 1 / 2 / 3 / 4 output channels."""
    import os
    from brats2019_amd import ops
    n, cout, d, h, w = shape
    x = _rand(n, 16, d, h, w, seed=21)
    x = torch.where(x > 0, x, 0.01 * x) * 1.3 + 0.1
    wt = _rand(cout, 16, 3, 3, 3, seed=22) * float((2.0 / (16 * 27)) ** 0.5)
    b = _rand(cout, seed=23)
    ref = torch.nn.functional.conv3d(x.double().cpu(), wt.double().cpu(), b.double().cpu(), padding=1)
    scale = float(ref.abs().max())
    res = {}
    for tag, env in (("head", "1"), ("wide", "0")):
        os.environ["RU_HEAD_FORM"] = env
        try:
            y = ops.conv3d_layout(ops.to_c16(x), wt, bias=b, in_c16=True, out_c16=False)
        finally:
            os.environ.pop("RU_HEAD_FORM", None)
        res[tag] = (float((y.double().cpu() - ref).abs().max()) / scale, y)
    print("  %s: max error / max |y|  head form %.2e  16-column kernel %.2e" % (shape, res["head"][0], res["wide"][0]))
    assert not torch.equal(res["head"][1], res["wide"][1]), "the shape did not take the head-form kernel"
    assert res["head"][0] <= 3e-5 and res["head"][0] <= 1.3 * res["wide"][0] + 1e-6, res
    assert torch.equal(res["wide"][1], ops.conv3d(x, wt, b, precision="bf16x3"))


WZ_SHAPES = [(2, 32, 32, 32, 32, 32), (1, 64, 64, 32, 32, 32), (1, 128, 128, 16, 32, 32), (2, 64, 64, 22, 20, 24), (3, 32, 64, 16, 24, 40), (1, 64, 32, 64, 28, 36)]


@pytest.mark.parametrize("shape", WZ_SHAPES, ids=["%dx%d-%d_%dx%dx%d" % s for s in WZ_SHAPES])
def test_conv3_winograd_z_against_float64(shape):
    """conv3_wz32_kernel / conv3_wz32mx_kernel (Winograd F(2,3) along z, (y, x) taps direct; the voxel-major 3x3x3 forward convolutions of the 32..128-channel levels:
    model.py:72-73 as used by model.py:89-91) against a float64 convolution of the same fp32 operands: max error / output RMS within 6e-5
    (CPU emulation: 2.7e-5, the direct split-bf16 kernel 2.4e-5 -- profiles/r05_winograd_gate.txt), and within 1.5x of what the DIRECT kernel
    (RU_WZ=0) leaves on the same inputs.  Shapes: full tiles at the three deep levels, ragged extents (H, W not multiples of the tile,
    several samples per workgroup run, zero padding on every face), Cin != Cout in both directions."""
    import os
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    x = _rand(n, cin, d, h, w, seed=11)
    x = torch.where(x > 0, x, 0.01 * x) * 1.3 + 0.1                     # an activated GroupNorm output
    wt = _rand(cout, cin, 3, 3, 3, seed=12) * float((2.0 / (cin * 27)) ** 0.5)
    ref = torch.nn.functional.conv3d(x.double().cpu(), wt.double().cpu(), padding=1)
    rms = float(ref.pow(2).mean().sqrt())
    res = {}
    # the three-product Winograd-z kernel (conv3_wz32_kernel on 32x32x16 MFMAs) and the direct kernel (RU_WZ=0).  (The first matrix form, conv3_wz_kernel on
    # 16x16x32 MFMAs, lives in devtools builds only since round 6.)
    for tag, env in (("wz32", "1"), ("direct", "0")):
        os.environ["RU_WZ"] = env
        try:
            y = ops.from_c16(ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=True))
        finally:
            os.environ.pop("RU_WZ", None)
        res[tag] = (float((y.double().cpu() - ref).abs().max()) / rms, y)
    assert not torch.equal(res["wz32"][1], res["direct"][1]), "the shape did not take the Winograd-z kernel"
    assert res["wz32"][0] <= 6e-5 and res["wz32"][0] <= 1.5 * res["direct"][0] + 1e-6, res
    # round 6: the same kernel with fp16 + MX-fp8 products (conv3_wz32mx_kernel), what the engine's forward convolutions of these levels run -- a launch that
    # declares its input an activation tensor; RU_MX=0 keeps the three-product kernel, bit for bit
    y_mx = ops.from_c16(ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=True, activations=True))
    os.environ["RU_MX"] = "0"
    try:
        y_off = ops.from_c16(ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=True, activations=True))
    finally:
        os.environ.pop("RU_MX", None)
    e_mx = float((y_mx.double().cpu() - ref).abs().max()) / rms
    print("  %s: max error / rms  winograd-z three bf16 products %.2e  direct %.2e  winograd-z fp16 + MX-fp8 %.2e" % (shape, res["wz32"][0], res["direct"][0], e_mx))
    assert torch.equal(y_off, res["wz32"][1]) and not torch.equal(y_mx, res["wz32"][1]), "RU_MX / the activation flag did not switch the product scheme"
    assert e_mx <= 1.2e-4 and e_mx <= 4 * res["direct"][0] + 1e-6, (e_mx, res)


MX_SHAPES = [(2, 16, 16, 64, 64, 64), (1, 16, 16, 40, 60, 72), (3, 16, 32, 32, 32, 48), (1, 16, 16, 128, 128, 128), (2, 16, 16, 30, 44, 40)]


@pytest.mark.parametrize("shape", MX_SHAPES, ids=["%dx%d-%d_%dx%dx%d" % s for s in MX_SHAPES])
def test_conv3_mx_against_float64(shape):
    """conv3_mx_kernel (the 16-channel level's forward convolutions, model.py:72-73 as used by model.py:89-91: f16 * f16 on v_mfma_f32_16x16x32_f16 plus both
    cross terms in e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4) against a float64 convolution of the same fp32 operands: max error / output RMS within 1.2e-4
    (CPU emulation tools/mx_gate.py: ~2x the three-product kernel's error) and within 4x of what the three-product kernel leaves on the same inputs.
    Shapes: whole tiles, ragged extents with zero padding on every face and several samples per workgroup run, two output-channel groups, the z-walk shape
    of the network (1 x 128^3).  RU_MX=0 and a launch that does not declare its input an activation tensor both keep the three-product kernel, bit for bit."""
    import os
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    x = _rand(n, cin, d, h, w, seed=21)
    x = torch.where(x > 0, x, 0.01 * x) * 1.3 + 0.1                     # an activated GroupNorm output
    wt = _rand(cout, cin, 3, 3, 3, seed=22) * float((2.0 / (cin * 27)) ** 0.5)
    ref = torch.nn.functional.conv3d(x.double().cpu(), wt.double().cpu(), padding=1)
    rms = float(ref.pow(2).mean().sqrt())
    x16 = ops.to_c16(x)
    y_mx = ops.from_c16(ops.conv3d_layout(x16, wt, in_c16=True, out_c16=True, activations=True))
    y_sb = ops.from_c16(ops.conv3d_layout(x16, wt, in_c16=True, out_c16=True))
    os.environ["RU_MX"] = "0"
    try:
        y_off = ops.from_c16(ops.conv3d_layout(x16, wt, in_c16=True, out_c16=True, activations=True))
    finally:
        os.environ.pop("RU_MX", None)
    e_mx = float((y_mx.double().cpu() - ref).abs().max()) / rms
    e_sb = float((y_sb.double().cpu() - ref).abs().max()) / rms
    print("  %s: max error / rms  fp16 + MX-fp8 %.2e  three bf16 products %.2e" % (shape, e_mx, e_sb))
    assert torch.equal(y_off, y_sb), "RU_MX=0 did not keep the three-product kernel"
    assert not torch.equal(y_mx, y_sb), "the shape did not take conv3_mx_kernel"
    assert e_mx <= 1.2e-4 and e_mx <= 4 * e_sb + 1e-6, (e_mx, e_sb)


MXG_SHAPES = [(2, 16, 16, 64, 64, 64), (1, 16, 16, 40, 60, 72), (1, 16, 16, 128, 128, 128), (2, 16, 16, 30, 44, 40)]


@pytest.mark.parametrize("shape", MXG_SHAPES, ids=["%dx%d-%d_%dx%dx%d" % s for s in MXG_SHAPES])
def test_conv3_gradient_operand_against_float64(shape):
    """conv3_mx_kernel<GRAD> (the 16-channel level's data-gradient convolutions: bf16 * bf16 main term, both cross terms in e4m3 with ONE exponent per voxel
    carried beside the tensor) against a float64 convolution of the same fp32 operands.  The input is a gradient-like tensor: magnitudes around 1e-6 that change by
    decades from voxel to voxel and from channel to channel -- what a fixed scale could not cover.  Relative L2 error within 2.5e-4 (CPU emulation: 8e-5; one bf16
    product 2.3e-3, three 4e-6), max error within 5e-4 of the largest output (the outputs span decades too); RU_MXG=0 keeps the three-product kernel bit for bit; an all-zero input gives zeros."""
    import os
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    g = torch.Generator().manual_seed(41)
    mag = torch.exp(2.5 * torch.randn(n, 1, d, h, w, generator=g)) * 1e-6
    chan = torch.exp(1.5 * torch.randn(1, cin, 1, 1, 1, generator=g))
    x = (torch.randn(n, cin, d, h, w, generator=g) * mag * chan).float().cuda()
    wt = _rand(cout, cin, 3, 3, 3, seed=42) * float((2.0 / (cin * 27)) ** 0.5)
    ref = torch.nn.functional.conv3d(x.double().cpu(), wt.double().cpu(), padding=1)
    x16 = ops.to_c16(x)
    y_g = ops.from_c16(ops.conv3d_layout(x16, wt, in_c16=True, out_c16=True, gradient=True))
    y_sb = ops.from_c16(ops.conv3d_layout(x16, wt, in_c16=True, out_c16=True))
    os.environ["RU_MXG"] = "0"
    try:
        y_off = ops.from_c16(ops.conv3d_layout(x16, wt, in_c16=True, out_c16=True, gradient=True))
    finally:
        os.environ.pop("RU_MXG", None)
    rel = lambda y: float((y.double().cpu() - ref).norm() / ref.norm())
    e_g, e_sb = rel(y_g), rel(y_sb)
    e_max = float((y_g.double().cpu() - ref).abs().max() / ref.abs().max())
    print("  %s: relative L2 error  bf16 + MX-fp8 (per-voxel exponent) %.2e  three bf16 products %.2e;  max error / max |y| %.2e" % (shape, e_g, e_sb, e_max))
    assert torch.equal(y_off, y_sb), "RU_MXG=0 did not keep the three-product kernel"
    assert not torch.equal(y_g, y_sb), "the shape did not take conv3_mx_kernel<GRAD>"
    assert e_g <= 2.5e-4 and e_max <= 5e-4, (e_g, e_max)
    z = ops.conv3d_layout(torch.zeros_like(x16), wt, in_c16=True, out_c16=True, gradient=True)
    assert float(z.abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(1, 16, 16, 16, 16, 16), (1, 32, 32, 8, 16, 16), (2, 16, 3, 32, 32, 32)], ids=["small16", "small32", "head"])
def test_conv3_activation_flag_falls_back_to_three_products(shape):
    """Shapes that have no fp16 + MX-fp8 kernel (grids below one persistent workgroup per CU: the one-stage kernel; the 3-channel head form) must ignore the
    activation flag: bit-identical to the launch without it."""
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    x = _rand(n, cin, d, h, w, seed=31)
    wt = _rand(cout, cin, 3, 3, 3, seed=32) * 0.05
    out16 = cout % 16 == 0
    a = ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=out16, activations=True)
    b = ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=out16)
    assert torch.equal(a, b)


def test_conv3_mx_saturates_instead_of_nan():
    """Activations beyond the e4m3 range of the cross terms (|x| > 448, and residuals beyond 448 / 2^11) must degrade the cross terms, never poison the output:
    the staging waves run with MODE.FP16_OVFL, under which v_cvt_pk_fp8_f32 / v_cvt_pk_f16_f32 saturate (tools/mx_ovfl_probe.hip; without it the e4m3
    conversion returns NaN above 464).  A tensor with a few 1e3 .. 6e4 outliers: finite everywhere, and never worse than ONE fp16 product would be (2^-11 of the
    largest output: the main term stays exact to fp16 rounding, what saturates is the correction to it)."""
    from brats2019_amd import ops
    n, cin, cout, d, h, w = 1, 16, 16, 32, 64, 64
    x = _rand(n, cin, d, h, w, seed=23)
    flat = x.view(-1)
    idx = torch.arange(0, flat.numel(), 40009, device=x.device)
    flat[idx] = torch.linspace(1e3, 6e4, idx.numel(), device=x.device) * torch.where(idx % 2 == 0, 1.0, -1.0)
    wt = _rand(cout, cin, 3, 3, 3, seed=24) * float((2.0 / (cin * 27)) ** 0.5)
    ref = torch.nn.functional.conv3d(x.double().cpu(), wt.double().cpu(), padding=1)
    y = ops.from_c16(ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=True, activations=True))
    assert bool(torch.isfinite(y).all())
    rel = float((y.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    print("  outliers: max error / max |y| %.2e" % rel)
    assert rel <= 2.0 ** -11, rel


@pytest.mark.parametrize("x16,dy16", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("shape", [(2, 16, 16, 16, 32, 32), (1, 32, 32, 16, 16, 32), (2, 16, 32, 8, 16, 16)])
def test_wgrad3_layouts_equal_ncdhw(shape, x16, dy16):
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    x = _rand(n, cin, d, h, w, seed=3)
    dy = _rand(n, cout, d, h, w, seed=4)
    ref = ops.conv3d_bwd_weight(x, dy, 3, precision="bf16x3")
    ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    dw = ops.conv3d_bwd_weight_layout(ops.to_c16(x) if x16 else x, ops.to_c16(dy) if dy16 else dy, x_c16=x16, dy_c16=dy16)
    if x16 and dy16:
        # both voxel-major: the transpose-read kernel (wgrad_tr.hip) sums the same split-bf16 products in another order
        exact = torch.nn.grad.conv3d_weight(x.double(), (cout, cin, 3, 3, 3), dy.double(), padding=1).float()
        tol = 2e-5 * float(exact.abs().max())
        assert float((dw - exact).abs().max()) < tol and float((ref - exact).abs().max()) < tol
    else:
        assert torch.equal(dw, ref), float((dw - ref).abs().max())


@pytest.mark.parametrize("shape", [(3, 24, 40, 48), (2, 16, 128, 128)])
def test_engine_c16_flow_matches_ncdhw_flow_multi_sample(shape):
    """whole network, several samples -- a ragged shape (one-stage conv kernel) and a wide one that takes the persistent kernels at
    the first level with every option on (z-walk tile order, statistics partials per workgroup and sample, GroupNorm-backward sums
    in the data-gradient epilogue across a sample boundary, 4-channel stem / head kernels): the split-bf16 engine (voxel-major working layout, fused decoder / stride-2
    paths, per-workgroup GroupNorm partials) against the f32 engine (NCDHW throughout): probabilities within 1e-4, every
    parameter gradient within 2e-2 in relative L2 norm.  The gradient bar is loose on purpose: LeakyReLU is kinked, so the
    ~1e-5 relative difference of the activations flips the branch of a ~1e-5 fraction of the units, each changing its gradient by
    O(1): ~sqrt(1e-5) in relative L2, 3e-3 .. 9e-3 measured on every shape tried (tools/gradcmp.py), cubic ones included."""
    from brats2019_amd import model as M
    torch.manual_seed(3)
    net = M.UNet(4, [1, 2, 2, 4], [1, 1, 1, 1], [16, 32, 64, 128], 3).cuda()
    n, d, hh, ww = shape
    x = _rand(n, 4, d, hh, ww, seed=9)
    tgt = (_rand(n, 3, d, hh, ww, seed=10) > 0.3).float()
    res = {}
    for prec in ("f32", "bf16x3"):
        net.set_precision(prec)
        net.zero_grad()
        p = net([x])[0]
        ((p - tgt) ** 2).mean().backward()
        res[prec] = (p.detach().clone(), {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None})
    pa, ga = res["f32"]
    pb, gb = res["bf16x3"]
    assert float((pa - pb).abs().max()) < 1e-4
    assert ga.keys() == gb.keys() and len(ga) > 80
    worst = max((float((ga[n] - gb[n]).norm() / (ga[n].norm() + 1e-30)), n) for n in ga)
    print("worst relative L2 gradient difference: %.2e (%s)" % worst)
    for n in ga:
        assert float((ga[n] - gb[n]).norm()) < 2e-2 * float(ga[n].norm()) + 1e-12, n


@pytest.mark.parametrize("cin,out16", [(4, True), (3, True), (4, False), (1, True)])
def test_conv3_few_input_channels(cin, out16):
    """4-channel tap-pair kernel (network input / head gradient) against the 16-channel-padded split-bf16 kernel and the exact
    convolution: same products, another summation order -> 2e-5 of the largest output"""
    from brats2019_amd import ops
    n, cout, d, h, w = 2, 16, 32, 32, 64
    x = _rand(n, cin, d, h, w, seed=5)
    wt = _rand(cout, cin, 3, 3, 3, seed=6) * 0.2
    exact = torch.nn.functional.conv3d(x.double(), wt.double(), padding=1).float()
    ref = ops.conv3d(x, wt, precision="bf16x3")
    y = ops.conv3d_layout(x, wt, out_c16=out16, few_channels=True)
    if out16:
        y = ops.from_c16(y)
    tol = 2e-5 * float(exact.abs().max())
    assert float((y - exact).abs().max()) < tol and float((ref - exact).abs().max()) < tol


@pytest.mark.parametrize("shape", [(2, 16, 4, 6, 8), (1, 32, 9, 5, 12), (1, 16, 1, 1, 4), (1, 16, 17, 3, 20)])
def test_upsample_c16_equals_ncdhw(shape):
    """Trilinear x2 and its transpose on C16 tensors (cell-pair / z-marching kernels) against the NCDHW kernels, which are
    themselves checked against the reference's F.interpolate in test_hip_ops.py (same nesting z(y(x)); the compiler contracts
    the multiply-adds differently, so the forward agrees to an ulp or two rather than bit for bit).  Shapes cover odd extents, a single plane / row and a z extent that is not a multiple of the marching chunk."""
    from brats2019_amd import ops
    n, c, d, h, w = shape
    x = _rand(n, c, d, h, w, seed=3)
    y_ref = ops.upsample2x(x)
    y = ops.from_c16(ops.upsample2x_c16(ops.to_c16(x)))
    assert float((y - y_ref).abs().max()) <= 1e-6
    y_act = ops.from_c16(ops.upsample2x_c16(ops.to_c16(x), out_slope=0.01))
    assert torch.equal(y_act, torch.where(y > 0, y, y * 0.01))
    dy = _rand(n, c, 2 * d, 2 * h, 2 * w, seed=4)
    dx_ref = ops.upsample2x_bwd(dy)
    dx = ops.from_c16(ops.upsample2x_bwd_c16(ops.to_c16(dy)))
    assert float((dx - dx_ref).abs().max()) <= 1e-5


def test_conv3_voxel_major_volume_beyond_32bit_block_offsets():
    """Maximum sizes: the persistent kernel addresses a 16-channel block of its voxel-major input with 32-bit byte offsets (buffer
    loads), so a volume of 2^25 voxels or more -- 64 bytes per voxel: 2 GiB per block -- must take the one-stage kernel instead of
    wrapping around.  1 x 16 x 336 x 320 x 320 (34.4 M voxels) against the exact-f32 kernel, and the far corner against a direct sum."""
    from brats2019_amd import ops
    d, h, w = 336, 320, 320
    assert d * h * w * 64 >= 2 ** 31
    x = _rand(1, 16, d, h, w, seed=11)
    wt = _rand(16, 16, 3, 3, 3, seed=12) * 0.1
    y = ops.from_c16(ops.conv3d_layout(ops.to_c16(x), wt, in_c16=True, out_c16=True))
    exact = ops.conv3d(x, wt, precision="f32")
    assert float((y - exact).abs().max()) <= 2e-4 * float(exact.abs().max())
    # last voxel of the volume, channel 5: taps beyond the far faces are zero padding
    patch = x[0, :, d - 2:, h - 2:, w - 2:].double()
    direct = float((patch * wt[5, :, :2, :2, :2].double()).sum())
    assert abs(float(y[0, 5, d - 1, h - 1, w - 1]) - direct) <= 1e-4 * max(1.0, abs(direct))


def _split_form(xc16):
    """[N][C/16][D][H][W][16] fp32 -> the split form gn_bwd_apply16 publishes: per voxel and 16-channel block 64 bytes =
    [hi bf16 ch0-7 | hi ch8-15 | lo ch0-7 | lo ch8-15], hi = bf16_rne(v), lo = bf16_rne(v - hi); returned as float32 words"""
    hi = xc16.to(torch.bfloat16)
    lo = (xc16 - hi.float()).to(torch.bfloat16)
    packed = torch.cat([hi, lo], dim=-1).contiguous()                 # [..., 32] bf16 = 64 bytes
    return packed.view(torch.float32), hi.float() + lo.float()


@pytest.mark.parametrize("shape", [(2, 32, 32, 32, 32, 32), (1, 64, 64, 32, 32, 32), (2, 64, 64, 22, 20, 24), (4, 16, 16, 32, 32, 64)],
                         ids=["wz32", "wz64", "wz64-ragged", "direct16"])
def test_conv3_split_form_input(shape):
    """The data-gradient convolutions of the engine read their input in SPLIT form (gn_bwd_apply16's hi / lo packets; flags bit 3): the direct kernels
    copy the packets (LDS-DMA at 32+ channels) and must give the convolution of hi + lo -- held to a float64 convolution of exactly those values.  (The
    Winograd-z route for such inputs, round 5's RU_WZ=2, is retired: measured slower twice.)"""
    import os
    from brats2019_amd import ops
    n, cin, cout, d, h, w = shape
    x = _rand(n, cin, d, h, w, seed=21)
    wt = _rand(cout, cin, 3, 3, 3, seed=22) * float((2.0 / (cin * 27)) ** 0.5)
    xs, xjoined = _split_form(ops.to_c16(x))
    ref = torch.nn.functional.conv3d(ops.from_c16(xjoined).double().cpu(), wt.double().cpu(), padding=1)
    rms = float(ref.pow(2).mean().sqrt())
    y = ops.from_c16(ops.conv3d_layout(xs, wt, in_c16=True, out_c16=True, in_split=True))
    err = float((y.double().cpu() - ref).abs().max()) / rms
    print("  %s split-form input: max error / rms %.2e" % (shape, err))
    assert err <= 6e-5, err


F32C_SHAPES = [(1, 16, 16, 32, 32, 64, True, True), (1, 32, 32, 16, 32, 32, True, True), (2, 64, 32, 10, 20, 24, True, True), (1, 128, 128, 16, 16, 16, True, True),
               (2, 16, 3, 12, 20, 24, True, False), (2, 4, 16, 12, 20, 24, False, True), (1, 4, 16, 64, 64, 64, False, True)]


@pytest.mark.parametrize("shape", F32C_SHAPES, ids=["%dx%d-%d_%dx%dx%d_%d%d" % s for s in F32C_SHAPES])
def test_conv3_exact_f32_voxel_major_equals_ncdhw_kernel_class(shape):
    """conv3_f32c_kernel (exact-f32 MFMA on voxel-major tensors: the exact-f32 INFERENCE forward, BASELINE configs[1]) against a float64 convolution
    and against the NCDHW exact-f32 kernel on the same inputs: both are k-ordered fmaf chains over the same products in different orders, so they agree
    to f32 round-off and sit equally close to float64 (measured 0.8e-6 .. 2.3e-6 of the output maximum, growing with K = 27 Cin, the NCDHW kernel
    0.7e-6 .. 2.6e-6; bars 4e-6 and 1.3x the NCDHW kernel's error).  Shapes: the three tile sizes ((4,8), (2,8), (2,4)), several input
    chunks, ragged extents with several samples, the head form (voxel-major in, NCDHW out, 3 channels + bias) and the stem form (4 NCDHW channels in)."""
    from brats2019_amd import ops
    n, cin, cout, d, h, w, in16, out16 = shape
    x = _rand(n, cin, d, h, w, seed=41)
    wt = _rand(cout, cin, 3, 3, 3, seed=42) * float((2.0 / (cin * 27)) ** 0.5)
    bias = None if out16 else _rand(cout, seed=43)
    ref64 = torch.nn.functional.conv3d(x.double().cpu(), wt.double().cpu(), None if bias is None else bias.double().cpu(), padding=1)
    ncdhw = ops.conv3d(x, wt, bias, precision="f32")
    y = ops.conv3d_layout(ops.to_c16(x) if in16 else x, wt, bias, in_c16=in16, out_c16=out16, exact_f32=True)
    if out16:
        y = ops.from_c16(y)
    scale = float(ref64.abs().max())
    e_new, e_old = float((y.double().cpu() - ref64).abs().max()) / scale, float((ncdhw.double().cpu() - ref64).abs().max()) / scale
    print("  %s: max error / max |y|  voxel-major f32 %.2e  NCDHW f32 %.2e" % (shape, e_new, e_old))
    assert e_new <= 4e-6 and e_new <= 1.3 * e_old + 5e-7 and float((y - ncdhw).abs().max()) <= 5e-6 * scale
