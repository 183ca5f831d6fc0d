#!/usr/bin/env python3
"""Derived per-kernel-family table for profiles/: for the 3x3x3 conv (forward kernel) and its weight gradient at the four level shapes of
the training step (batch 4): average launch time, algorithmic HBM GB/s, counter HBM traffic, MFMA-busy %, roofline fractions.

Runs ON THE GPU BOX (tools/family_table.py > gpurun_out/<tag>_family_table.txt).  Every number comes from rocprofv3 on tools/conv_probe.py:
one --kernel-trace pass for the time, separate --pmc passes (never combined with tracing) for SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE,
FETCH_SIZE and WRITE_SIZE.  Counter conventions (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts a
128-byte request of a wide streaming read as 64 bytes, so it is doubled; GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for the
kernel's cycles); MFMA-busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)."""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "conv_probe.py")
SHAPES = [(16, 128), (32, 64), (64, 32), (128, 16)]
N = 4
BF16_PEAK, HBM_PEAK = 2.5e15, 8.0e12


def run(tag, extra, args, more_env=None):
    out = "/tmp/ft_%s" % tag
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp", **(more_env or {}))
    subprocess.run(["rocprofv3"] + extra + ["--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, PROBE] + args,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return out


def kernel_rows(out, pattern, want):
    rows = []
    for f in glob.glob(out + "/**/*%s.csv" % pattern, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if want(r)]
    return rows


def main():
    lines = ["# kernel family table (tools/family_table.py): batch %d, voxel-major tensors, split-bf16 x3; peaks: %.1f PF dense bf16, %.1f TB/s HBM" % (N, BF16_PEAK / 1e15, HBM_PEAK / 1e12),
             "# kind   shape        kernel                          avg_us  alg_GB  alg_GB/s  hbm_frac  counter_GB (fetch x2 + write)  MFMA_busy%%  exec_MFMA_TF  exec_frac"]
    # "conv fwd": the kernel the engine runs at that shape (round 5: conv3_wz32_kernel -- Winograd F(2,3) along z on 32x32x16 MFMAs -- at 32..128 channels);
    # "conv dir": the direct persistent kernel at the same shapes (RU_WZ=0), kept beside it for comparison
    # round 6: "conv fwd" = the engine's forward kernels on activations (flags 35: conv3_mx_kernel at 16 channels, conv3_wz32mx_kernel at 32..128: fp16 + MX-fp8 products);
    # "conv 3p" = the three-bf16-product kernels of the same shapes (flags 3: conv3_sb2_kernel / conv3_wz32_kernel; the data-gradient convolutions run the direct ones)
    for kind, kpats, flags, more_env in (("conv fwd", ("conv3_mx_kernel", "conv3_wz32mx_kernel", "conv3_sb2_kernel", "conv3_wz32_kernel"), "35", {}),
                                         ("conv 3p", ("conv3_sb2_kernel", "conv3_wz32_kernel"), "3", {}), ("conv dir", ("conv3_sb2_kernel",), "3", {"RU_WZ": "0"}),
                                         ("wgrad", ("wgrad3_tz_kernel",), "3", {})):
        for c, size in SHAPES:
            if kind == "conv dir" and c < 32:
                continue
            args = ["fwd" if kind.startswith("conv") else "wgrad", "bf16x3", str(N), str(c), str(size), "6", flags]
            want = lambda r, kpats=kpats: any(k in r.get("Kernel_Name", r.get("Name", "")) for k in kpats)
            tr = kernel_rows(run("t", ["--kernel-trace"], args, more_env), "kernel_trace", want)
            if not tr:
                continue
            durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr)
            us = sum(durs[1:]) / max(1, len(durs) - 1) if len(durs) > 1 else durs[0]
            name = tr[0]["Kernel_Name"].replace("void ", "").split("(")[0]
            cnt = {}
            for cs in (["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"]):
                for r in kernel_rows(run("c", ["--pmc"] + cs, args, more_env), "counter_collection", want):
                    cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            mean = lambda k: sum(cnt[k]) / len(cnt[k]) if cnt.get(k) else float("nan")
            vox = N * size ** 3
            alg_bytes = 2.0 * c * vox * 4                                                    # conv: x + y; weight gradient: x + dy (fp32)
            flops = 2.0 * 27 * c * c * vox
            # executed MFMA products per algorithmic one: direct kernels 3 x 28/27 (one phantom tap); Winograd-z on 16x16x32 MFMAs 3 x 40/54 (four
            # transformed planes x 10 tap slots, one of them phantom, per two output planes); on 32x32x16 MFMAs 3 x 36/54 (one K-step per tap)
            # (round 6, in bf16-MFMA time units: conv3_mx_kernel 28 per 27 taps -- 14 fp16 K-steps of two taps + 7 x 2; conv3_wz32mx_kernel 19 per 9 taps of a
            # transformed plane, 4 planes per two output planes: 4 x 19 / 54)
            unit = (28 / 13.5 if "conv3_mx" in name else (4 * 19 / 27.0 if "conv3_wz32mx" in name else (3 * 36 / 54 if "conv3_wz32" in name else (3 * 40 / 54 if "conv3_wz" in name else 3 * 28 / 27))))
            exec_tf = flops * unit / (us * 1e-6) / 1e12
            counter_gb = (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024 / 1e9
            busy = 100.0 * mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * mean("GRBM_GUI_ACTIVE") / 8.0)
            if kind == "conv fwd" and c == 16 and "--json" in sys.argv:
                import json
                dst = sys.argv[sys.argv.index("--json") + 1]
                old = json.load(open(dst)) if os.path.exists(dst) else {}
                hist = old.get("history", {})
                if "hbm_bytes_per_launch" in old:
                    hist["previous file (%s)" % old.get("source", "?").split(";")[-1].strip()] = {k: old[k] for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_per_launch", "avg_launch_us_in_trace", "mfma_busy_pct") if k in old}
                json.dump({"kernel": name + " (3x3x3 conv 16->16 forward, batch 4 x 128^3, voxel-major tensors)",
                           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/family_table.py --json) on tools/conv_probe.py fwd bf16x3 4 16 128 6 35; %s" % os.environ.get("RU_ROUND_TAG", "round 6"),
                           "FETCH_SIZE_KB": mean("FETCH_SIZE"), "WRITE_SIZE_KB": mean("WRITE_SIZE"),
                           "correction": "FETCH_SIZE x2 for 16-byte-per-lane streaming reads on gfx950 (guide); WRITE_SIZE uncorrected",
                           "hbm_bytes_per_launch": int((2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024), "algorithmic_bytes_per_launch": int(alg_bytes),
                           "avg_launch_us_in_trace": round(us, 1), "mfma_busy_pct": round(busy, 1), "history": hist}, open(dst, "w"), indent=1)
            lines.append("%-8s %3dch %3d^3  %-30s %7.1f  %6.3f  %8.0f  %8.3f  %10.3f  %27.1f  %12.0f  %9.3f" % (
                kind, c, size, name[:30], us, alg_bytes / 1e9, alg_bytes / (us * 1e-6) / 1e9, alg_bytes / (us * 1e-6) / HBM_PEAK, counter_gb, busy,
                exec_tf, exec_tf * 1e12 / BF16_PEAK))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
