#!/usr/bin/env python3
"""Timeline of one search step from a rocprofv3 --kernel-trace CSV (development tool).

    python scripts/gap_report.py <dir with *_kernel_trace.csv> [anchor kernel substring] [kernel a step must contain]

Prints, for a step in the middle of the timed region (the 9 consecutive steps with the shortest span; anchored on the query preparation kernel), every kernel's start relative to
the step start, its duration and the idle gap in front of it, then the average step period over those steps.
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "prep_queries8_kernel"
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel trace under " + root)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    must = sys.argv[3] if len(sys.argv) > 3 else "i8_tile_kernel<0"
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    starts = [s for j, s in enumerate(starts[:-1]) if any(must in r[2] for r in rows[s:starts[j + 1]])]
    if len(starts) < 10:
        sys.exit("too few steps in the trace")
    # the timed region = the 9 consecutive steps with the shortest span (bench.py runs its validation, latency and drop-in calls
    # BEHIND the timed steps since round 3: the last steps of the trace are those, spaced by host work)
    span = lambda i: rows[starts[i + 8]][0] - rows[starts[i]][0]
    i0 = min(range(len(starts) - 8), key=span)
    a, b = starts[i0 + 4], starts[i0 + 5]
    t0 = rows[a][0]
    prev_end = rows[a - 1][1]
    print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel")
    for s, e, name in rows[a:b]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {name[:90]}")
        prev_end = e
    per = [(rows[starts[i + 1]][0] - rows[starts[i]][0]) / 1e3 for i in range(i0, i0 + 8)]
    busy = sum(e - s for s, e, _ in rows[a:b]) / 1e3
    print(f"step period (8 timed steps): {sum(per) / len(per):.1f} us; kernels busy in the shown step: {busy:.1f} us")


if __name__ == "__main__":
    main()
