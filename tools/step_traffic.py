#!/usr/bin/env python3
"""Sum the FETCH_SIZE / WRITE_SIZE counter passes of tools/step_traffic.sh per kernel -> the HBM bytes one training step moves.
usage: step_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ rocprofv3 outputs> <steps profiled> [out.txt]

Units and corrections (MI355X_MICROARCH.md, HBM section): the counters are in KB; on gfx950 FETCH_SIZE reports HALF of the bytes of wide
(16 B per lane) coalesced streaming reads -- every kernel of this library streams with 16-byte loads, so the read column is FETCH_SIZE x 2
(raw column kept); WRITE_SIZE is taken as reported.  The fused lower bound of the step (only the convolutions' inputs and outputs touch
HBM, fp32: SURVEY 8(d)) is printed beside the total."""
import collections
import csv
import glob
import os
import sys

src, steps = sys.argv[1], int(sys.argv[2])
OUT = sys.argv[3] if len(sys.argv) > 3 else None
agg = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "calls": 0})
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(src, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "")
            cut = name.find("(")
            name = name[:cut] if cut > 0 else name
            agg[name][counter] += float(r["Counter_Value"])
            if counter == "FETCH_SIZE":
                agg[name]["calls"] += 1
rows = []
for name, d in agg.items():
    rd_raw, wr = d["FETCH_SIZE"] * 1024 / steps, d["WRITE_SIZE"] * 1024 / steps
    rows.append((name, d["calls"] / steps, rd_raw, 2 * rd_raw, wr, 2 * rd_raw + wr))
rows.sort(key=lambda r: -r[5])
tot = [sum(r[i] for r in rows) for i in range(2, 6)]
# fused lower bound of one step at batch 4 x 128^3 (bench.family_bounds: conv / weight-gradient passes, input + output fp32)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bound = None
try:
    sys.argv = [sys.argv[0]]
    import bench
    fb = bench.family_bounds(4, 128, "bf16x3")
    bound = sum(f["gbytes"] for k, f in fb.items() if k != "groupnorm")
except Exception as e:                                     # noqa: BLE001
    bound = None
lines = ["# HBM traffic of one training step (batch 4 x 128^3, bf16x3), per kernel, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over %d steps" % steps,
         "# read_GB = FETCH_SIZE x 2 (gfx950: 16-byte-per-lane streaming reads are tallied at half), write_GB = WRITE_SIZE; per STEP",
         "# %-88s %8s %10s %10s %10s %10s" % ("kernel", "calls", "fetch_raw", "read_GB", "write_GB", "total_GB")]
for name, calls, rr, rd, wr, t in rows:
    if t < 1e6:
        continue
    lines.append("%-90s %8.1f %10.3f %10.3f %10.3f %10.3f" % (name[:90], calls, rr / 1e9, rd / 1e9, wr / 1e9, t / 1e9))
lines.append("%-90s %8s %10.3f %10.3f %10.3f %10.3f" % ("TOTAL per step", "", tot[0] / 1e9, tot[1] / 1e9, tot[2] / 1e9, tot[3] / 1e9))
if bound is not None:
    lines.append("# fused lower bound (conv / weight-gradient / 1x1 passes read input + write output once, fp32; bench.family_bounds): %.2f GB per step = %.2f ms at 8 TB/s"
                 % (bound, bound / 8.0))
    lines.append("# measured / bound = %.2f" % (tot[3] / 1e9 / bound))
    ab = bench.achievable_bounds(fb)
    lines.append("# achievable bound (+ the GroupNorm passes no fusion removes for fp32 tensors: forward residual pass of every block, backward apply below the "
                 "16-channel level): %.2f GB -> measured / bound = %.2f" % (ab["gbytes_achievable"], tot[3] / 1e9 / ab["gbytes_achievable"]))
    lines.append("# achievable bound, all (+ the y re-read of every GroupNorm's backward sums, + the (y, d) reads of the 16-channel level's apply inside its "
                 "weight gradient): %.2f GB -> measured / bound = %.2f" % (ab["gbytes_achievable_all"], tot[3] / 1e9 / ab["gbytes_achievable_all"]))
text = "\n".join(lines) + "\n"
if OUT:
    open(OUT, "w").write(text)
else:
    sys.stdout.write(text)
