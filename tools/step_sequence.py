#!/usr/bin/env python3
"""Per-launch view of one training step from a rocprofv3 --kernel-trace CSV: the launches of the last K steps are
aligned by position in the step and averaged.  Usage: step_sequence.py <x_kernel_trace.csv> <steps K> [out.txt]
(the trace must end with K identical steps, e.g. `bench.py --steps K --warmup W --no-extras`)."""
import csv
import sys


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    k = int(sys.argv[2])
    names = [short(r["Kernel_Name"]) for r in rows]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    start = [int(r["Start_Timestamp"]) for r in rows]
    # a step ends with the last adam_kernel launch of a run of adam launches
    ends = [i for i in range(len(names)) if "adam" in names[i] and (i + 1 == len(names) or "adam" not in names[i + 1])]
    ends = ends[-(k + 1):]                      # (bench.py's probe steps at the end launch the same kernels: they count as steps here)
    per = min(ends[j + 1] - ends[j] for j in range(len(ends) - 1))
    # a step is the `per` launches that END at its last adam launch (bench.py reads the loss between the timed region and the probe
    # steps: one blit kernel that belongs to no step)
    lines = ["# launches per step %d; averaged over %d steps; columns: index, kernel, avg_us, gap_before_us" % (per, k)]
    tot = 0.0
    for p in range(per):
        idx = [ends[j + 1] - per + 1 + p for j in range(k)]
        assert len(set(names[i] for i in idx)) == 1
        d = sum(dur[i] for i in idx) / k
        gap = sum((start[i] - (start[i - 1] + dur[i - 1] * 1e3)) / 1e3 for i in idx) / k
        tot += d
        lines.append("%4d %-60s %9.1f %8.1f" % (p, names[idx[0]], d, gap))
    lines.append("# sum of kernel time per step %.3f ms" % (tot / 1e3))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
