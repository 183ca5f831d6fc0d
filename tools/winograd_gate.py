#!/usr/bin/env python3
"""Numerics gate for a Winograd F(2x2x2, 3x3x3) form of the 3x3x3 convolution in split-bf16 arithmetic (round-4 verdict, item 2).

CPU emulation in torch: the direct convolution with split-bf16 operands (hi = bf16(v), lo = bf16(v - hi); hi*hi + lo*hi + hi*lo, fp32
accumulate -- what conv3_sb2_kernel computes) against the Winograd form with the SAME split applied AFTER the input / weight transforms
(transforms in fp32, 64 transformed-domain products per 8 outputs instead of 216), both against a float64 reference, at the channel
counts of the network's deep levels.  Prints max and RMS error relative to the RMS of the output.

    python tools/winograd_gate.py          # ~1 minute on 8 cores
Result (profiles/r04_winograd_gate.txt): the Winograd form carries ~1.9x the error of the direct split-bf16 conv (4.1e-5 vs 2.2e-5 max,
8.5e-6 vs 4.5e-6 RMS) -- the gate PASSES.  Why no kernel was built on it: DESIGN.md section 7."""
import torch


def bf(v):
    return v.to(torch.bfloat16).to(torch.float32)


def split(v):
    hi = bf(v)
    return hi, bf(v - hi)


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def xform(t, M, dt):
    M = M.to(dt)
    t = torch.einsum('ai,...ijk->...ajk', M, t)
    t = torch.einsum('bj,...ajk->...abk', M, t)
    return torch.einsum('ck,...abk->...abc', M, t)


def run(Cin, Cout, S, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Cin, S, S, S, generator=g, dtype=torch.float64)
    x = torch.where(x > 0, x, 0.01 * x) * 1.3 + 0.1                     # an activated GroupNorm output
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (Cin * 27)) ** 0.5
    x32, w32 = x.float().double(), w.float().double()
    conv = lambda a, b: torch.nn.functional.conv3d(a[None], b, padding=1)[0]
    ref = conv(x32, w32)
    xh, xl = split(x32.float())
    wh, wl = split(w32.float())
    out = {"direct_sb3": conv(xh, wh) + conv(xl, wh) + conv(xh, wl), "direct_f32": conv(x32.float(), w32.float()), "direct_bf16": conv(xh, wh)}
    xp = torch.nn.functional.pad(x32.float(), (1, 1, 1, 1, 1, 1))
    t = xp.unfold(1, 4, 2).unfold(2, 4, 2).unfold(3, 4, 2)              # [Cin, T, T, T, 4, 4, 4]
    V, U = xform(t, BT, torch.float32), xform(w32.float(), G, torch.float32)
    Vh, Vl = split(V)
    Uh, Ul = split(U)
    mm = lambda a, b: torch.einsum('cxyzijk,ocijk->oxyzijk', a, b)
    for name, M in (("wino_sb3", mm(Vh, Uh) + mm(Vl, Uh) + mm(Vh, Ul)), ("wino_f32", mm(V, U)), ("wino_bf16", mm(Vh, Uh))):
        out[name] = xform(M, AT, torch.float32).permute(0, 1, 4, 2, 5, 3, 6).reshape(Cout, S, S, S)
    # round 5: F(2,3) along z ONLY ((y, x) taps direct: 18 transformed-domain "taps" per output pair instead of 27 -- 1.5x fewer products), the
    # form conv3_wz_kernel computes: U = B^T d along z (4 transformed planes per pair of output planes), G = G g along dz, out = A^T M
    tz = xp.unfold(1, 4, 2)                                             # [Cin, T, S+2, S+2, 4]
    Vz = torch.einsum('ai,cthwi->ctahw', BT.float(), tz)                # [Cin, T, 4, S+2, S+2]
    Uz = torch.einsum('ai,ocijk->ocajk', G.float(), w32.float())        # [Cout, Cin, 4, 3, 3]
    Vzh, Vzl = split(Vz)
    Uzh, Uzl = split(Uz)

    def mmz(a, b):                                                      # per transformed plane xi: a 2-D 3x3 convolution over (y, x)
        T = a.shape[1]
        o = []
        for xi in range(4):
            o.append(torch.nn.functional.conv2d(a[:, :, xi].permute(1, 0, 2, 3), b[:, :, xi]))    # [T, Cout, S, S]
        return torch.stack(o, 2)                                        # [T, Cout, 4, S, S]
    for name, M in (("wz_sb3", mmz(Vzh, Uzh) + mmz(Vzl, Uzh) + mmz(Vzh, Uzl)), ("wz_f32", mmz(Vz, Uz)), ("wz_bf16", mmz(Vzh, Uzh))):
        o = torch.einsum('pa,tcahw->ctphw', AT.float(), M)              # [Cout, T, 2, S, S]
        out[name] = o.reshape(Cout, S, S, S)
    rms = ref.pow(2).mean().sqrt()
    return {k: (((v.double() - ref).abs().max() / rms).item(), ((v.double() - ref).pow(2).mean().sqrt() / rms).item()) for k, v in out.items()}


if __name__ == "__main__":
    torch.manual_seed(0)
    for Cin, Cout, S in ((32, 32, 16), (64, 64, 16), (128, 128, 8)):
        print("Cin %d Cout %d %d^3" % (Cin, Cout, S))
        for k, (mx, rm) in run(Cin, Cout, S).items():
            print("   %-12s max/rms %.3e   rmse/rms %.3e" % (k, mx, rm))
