#!/usr/bin/env python3
"""Where a step of the phase leaves the device idle: reads the kernel trace of scripts/gpu_trace.sh, takes the kernels [--lo, --hi) of one step (without --lo: lists the long
idle stretches, which is where steps begin), prints per stream the busy time and, over all streams together, the union busy time
and every stretch in which no kernel ran for more than --gap microseconds, with what ended before it and what started after it.
usage: python scripts/trace_gaps.py gpurun_out/<tag>/kernel_trace.csv [--gap 100] [--step -1] [--rows]"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name.split("::")[-1][:28]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--gap", type=float, default=100.0)
    ap.add_argument("--lo", type=int, default=-1, help="first kernel of the stretch to look at (without: lists the long idle stretches, i.e. where steps begin)")
    ap.add_argument("--hi", type=int, default=-1)
    ap.add_argument("--rows", action="store_true", help="print every kernel of the step")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    if a.lo >= 0:
        lo, hi = a.lo, (a.hi if a.hi > 0 else len(rows))
    else:
        # a step = the kernels between two launches of the first kernel the phase's step launches with this period
        end = rows[0][1]
        gaps = []
        for i, r in enumerate(rows[1:], 1):
            if r[0] - end > 1.5e6:
                gaps.append(i)
            end = max(end, r[1])
        for i in gaps:
            print("idle > 1.5 ms before kernel %d (%.2f ms after the first)" % (i, (rows[i][0] - rows[0][0]) / 1e6))
        return
    step = rows[lo:hi]
    t0 = step[0][0]
    t1 = max(r[1] for r in step)
    print("step of %d kernels, %.2f ms from first start to last end" % (len(step), (t1 - t0) / 1e6))
    busy = 0
    end = t0
    idle = []
    last = None
    for r in step:
        if r[0] > end:
            if (r[0] - end) / 1e3 >= a.gap:
                idle.append((end, r[0], last, r))
            busy += r[1] - r[0]
            end = r[1]
            last = r
        else:
            if r[1] > end:
                busy += r[1] - end
                end = r[1]
                last = r
    print("union busy %.2f ms, idle %.2f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
    per = {}
    for r in step:
        d = per.setdefault(r[2], [0, 0, r[0], r[1]])
        d[0] += r[1] - r[0]
        d[1] += 1
        d[3] = max(d[3], r[1])
    for s, d in sorted(per.items()):
        print("  stream %3d: %4d kernels, %.2f ms of kernels, active %.2f .. %.2f ms" % (s, d[1], d[0] / 1e6, (d[2] - t0) / 1e6, (d[3] - t0) / 1e6))
    print("idle stretches >= %.0f us: %d, together %.2f ms" % (a.gap, len(idle), sum(b - e for e, b, _, _ in idle) / 1e6))
    for e, b, p, n in idle:
        print("  %7.3f .. %7.3f ms (%5.0f us): after %s [s%d], before %s [s%d]" % ((e - t0) / 1e6, (b - t0) / 1e6, (b - e) / 1e3, p[3] if p else "-", p[2] if p else -1, n[3], n[2]))
    if a.rows:
        for r in step:
            print("%8.3f %8.3f s%-3d %-28s %d" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
