#!/usr/bin/env python3
"""Numerics gate for a cheaper product scheme of the 3x3x3 FORWARD convolutions (round-5 verdict, item 1a).

Today every product of a split-bf16 convolution is three bf16 MFMA products (hi*hi + lo*hi + hi*lo).  The candidate: the main term in
fp16 (`v_mfma_f32_16x16x32_f16`: same rate as bf16, 11-bit significands, so the residual lo = v - fp16(v) is 2^-12 |v| instead of
2^-9 |v|) and BOTH cross terms in OCP e4m3 on the MX-scaled `v_mfma_scale_f32_16x16x128_f8f6f4` (twice the bf16 rate):

    x*w ~= f16(x)*f16(w) + 2^-SX * e4m3(x_lo * 2^SX) * 2^-SWH * e4m3(w * 2^SWH) + e4m3(x) * 2^-SWL * e4m3(w_lo * 2^SWL)

with ONE power-of-two scale per operand class for the whole tensor (the instruction's E8M0 block scales are then constants: a staged LDS element
is shared by output voxels that pair it with different taps, so a per-(row, K-block) scale cannot depend on the data).

CPU emulation on the oracle, 1 x 128^3, seeded weights, EVERY 3x3x3 convolution replaced; operand rounding emulated exactly, accumulation in
float64 (so only the operand rounding shows), against the exact-float32-operand network.  Prints max |dp|, flips of the > 0.5 mask and how many
of them lie outside the 1e-3 band around the threshold.

    python tools/mx_gate.py [size]      # size 128: ~6 minutes on 8 cores; 64: 1 minute
Result: profiles/r06_mx_gate.txt."""
import sys, os
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import resunet_oracle as ro

SX, SWH, SWL = 11, 8, 19      # l8 = e4m3(x_lo * 2^11) (|x| <= 448), wh8 = e4m3(w * 2^8) (|w| <= 1.75), wl8 = e4m3(w_lo * 2^19) (|w| <= 1.75): SX + SWH == SWL, so ONE pair of hardware scales serves both cross terms


def bf(v):
    return v.to(torch.bfloat16).to(torch.float32)


def f16(v):
    return v.to(torch.float16).to(torch.float32)


def e4m3(v):                   # saturating conversion (the hardware conversion saturates; torch's produces NaN above 448)
    return v.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def conv64(a, b):
    return F.conv3d(a.double(), b.double(), None, stride=1, padding=1)


def scheme_exact(x, w):
    return conv64(x, w)


def scheme_sb3(x, w):
    xh, wh = bf(x), bf(w)
    xl, wl = bf(x - xh), bf(w - wh)
    return conv64(xh, wh) + conv64(xl, wh) + conv64(xh, wl)


def scheme_bf16_fp8(x, w):     # round 2's emulation: bf16 main term, e4m3 cross terms (lands at 9e-4)
    xh, wh = bf(x), bf(w)
    xl, wl = x - xh, w - wh
    return conv64(xh, wh) + conv64(e4m3(xl * 2.0 ** 11), e4m3(w * 2.0 ** SWH)) * 2.0 ** -(11 + SWH) + conv64(e4m3(x), e4m3(wl * 2.0 ** 17)) * 2.0 ** -17


def scheme_f16_fp8(x, w):      # the candidate
    xh, wh = f16(x), f16(w)
    xl, wl = x - xh, w - wh
    return (conv64(xh, wh) + conv64(e4m3(xl * 2.0 ** SX), e4m3(w * 2.0 ** SWH)) * 2.0 ** -(SX + SWH)
            + conv64(e4m3(x), e4m3(wl * 2.0 ** SWL)) * 2.0 ** -SWL)


def scheme_f16_fp8_bf8lo(x, w):    # variant: the lo operands in e5m2 (more range, 3-bit significands) -- shows what the significand width buys
    e5m2 = lambda v: v.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float32)
    xh, wh = f16(x), f16(w)
    xl, wl = x - xh, w - wh
    return (conv64(xh, wh) + conv64(e5m2(xl * 2.0 ** SX), e4m3(w * 2.0 ** SWH)) * 2.0 ** -(SX + SWH)
            + conv64(e4m3(x), e5m2(wl * 2.0 ** SWL)) * 2.0 ** -SWL)


def scheme_f16_only(x, w):     # one fp16 product
    return conv64(f16(x), f16(w))


def scheme_f16_two(x, w):      # fp16 main + ONE cross pair folded: f16(x) * f16(w) + e4m3 cross terms on the activations only
    xh, wh = f16(x), f16(w)
    return conv64(xh, wh) + conv64(e4m3((x - xh) * 2.0 ** SX), e4m3(w * 2.0 ** SWH)) * 2.0 ** -(SX + SWH)


SCHEMES = [("exact f32 operands", scheme_exact), ("bf16 hi*hi + lo*hi + hi*lo (the library)", scheme_sb3), ("bf16 main + e4m3 cross terms (round 2)", scheme_bf16_fp8),
           ("fp16 main + e4m3 cross terms (candidate)", scheme_f16_fp8), ("fp16 main + e5m2 lo operands", scheme_f16_fp8_bf8lo),
           ("fp16 main + activation cross term only", scheme_f16_two), ("one fp16 product", scheme_f16_only)]


def forward(params, x, scheme):
    def conv3(xx, w, bias=None):
        y = scheme(xx, w).float()
        return y if bias is None else y + bias.view(1, -1, 1, 1, 1)
    keep = ro.conv3x3x3
    ro.conv3x3x3 = conv3
    try:
        with torch.no_grad():
            return ro.unet_forward(params, x)
    finally:
        ro.conv3x3x3 = keep


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    torch.set_num_threads(os.cpu_count())
    params = ro.to_torch(ro.make_params(seed=1337))
    x = torch.from_numpy(ro.make_input(1, S, S, S, seed=1337))
    print("1 x %d^3, seeded weights (oracle.make_params(1337)); every 3x3x3 convolution replaced; float64 accumulation" % S)
    ref = None
    for name, fn in SCHEMES:
        p = forward(params, x, fn)
        if ref is None:
            ref = p
            print("%-48s reference (%d voxels x 3 classes)" % (name, S ** 3))
            continue
        d = (p - ref).abs()
        flips = (p > 0.5) != (ref > 0.5)
        outside = flips & ((ref - 0.5).abs() >= 1e-3)
        print("%-48s max|dp| %.2e   rms %.2e   mask flips %d (outside the 1e-3 band: %d)" % (name, d.max().item(), d.pow(2).mean().sqrt().item(), int(flips.sum()), int(outside.sum())))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
