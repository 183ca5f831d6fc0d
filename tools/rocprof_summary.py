#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db or *_kernel_stats.csv) into the
per-kernel text summary committed under profiles/.  Usage: rocprof_summary.py <results.db|stats.csv> [out.txt]"""
import csv
import sqlite3
import sys


def rows_from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    return [(r[0], int(r[1]), float(r[2]) * 1e3, float(r[3]) * 1e3, float(r[4])) for r in
            cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels")]   # us -> ns


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    return out


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    name = name[:cut] if cut > 0 else name
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    src = sys.argv[1]
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    rows.sort(key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    lines = ["# kernel, calls, total_ms, avg_us, pct   (source: %s; total GPU kernel time %.2f ms)" % (src.split("/")[-1], total / 1e6)]
    for name, calls, tot, avg, _pct in rows:
        lines.append("%-90s %6d %10.3f %10.2f %6.2f" % (short(name), calls, tot / 1e6, avg / 1e3, 100.0 * tot / total))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
