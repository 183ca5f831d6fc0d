#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 --kernel-trace CSV: usage gap_report.py <kernel_trace.csv> [min_gap_us]
Prints busy time, span, and the largest gaps with the kernels on either side (host-side bubbles show up here)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60]))
rows.sort()
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
half = rows[len(rows) // 2:]                      # second half of the run (the Trainer leg of tools/trainer_probe.py)
for name, part in (("first half", rows[:len(rows) // 2]), ("second half", half)):
    busy = sum(e - s for s, e, _ in part) / 1e6
    span = (part[-1][1] - part[0][0]) / 1e6
    gaps = [((part[i + 1][0] - part[i][1]) / 1e3, part[i][2], part[i + 1][2]) for i in range(len(part) - 1)]
    big = sorted([g for g in gaps if g[0] >= ming], reverse=True)
    print("%s: %d kernels, busy %.2f ms, span %.2f ms, idle %.2f ms; %d gaps >= %.0f us (sum %.2f ms)" % (name, len(part), busy, span, span - busy, len(big), ming, sum(g[0] for g in big) / 1e3))
    import collections
    agg = collections.Counter()
    tot = collections.Counter()
    for g, a, b in big:
        agg[(a, b)] += 1
        tot[(a, b)] += g
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
        print("   %8.1f us total, %3d x   after %-50s before %s" % (v, agg[k], k[0], k[1]))
