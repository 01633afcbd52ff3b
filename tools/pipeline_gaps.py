#!/usr/bin/env python3
"""Where a pipelined step goes, from a rocprofv3 kernel trace of bench.py (one line per kernel launch with start / end
timestamps and the queue it ran on): per queue, the kernels of consecutive batches, their durations, the gap between the end
of one kernel and the start of the next on the same queue, how many kernels are running at a time, and the busy fraction.

usage:  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --emulate 8:0 --cpu-queries 0
        python3 tools/pipeline_gaps.py DIR [last-N-launches]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np


def short(name):
    for key in ("prep", "match", "scan", "tail"):
        if "knn_cells_%s_kernel" % key in name:
            return key
    return None


def main():
    d = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Queue_Id"]))
    rows.sort()
    rows = rows[-last:]          # the timed region and the single launches behind it are the end of the run; see `window`
    # the pipelined phase: the longest stretch in which >= 3 queues are active
    t0, t1 = rows[0][0], rows[-1][1]
    by_q = defaultdict(list)
    for s, e, k, q in rows:
        by_q[q].append((s, e, k))
    print("kernels of the pruned chain in the window: %d on %d queues, %.3f ms" % (len(rows), len(by_q), (t1 - t0) / 1e6))
    dur = defaultdict(list)
    gap = defaultdict(list)
    for q, lst in by_q.items():
        for i, (s, e, k) in enumerate(lst):
            dur[k].append((e - s) / 1e3)
            if i + 1 < len(lst):
                gap["%s->%s" % (k, lst[i + 1][2])].append((lst[i + 1][0] - e) / 1e3)
    print("durations (us): median / p10 / p90   [launches]")
    for k in ("prep", "match", "scan", "tail"):
        if dur[k]:
            a = np.array(dur[k])
            print("  %-6s %7.2f %7.2f %7.2f   [%d]" % (k, np.median(a), np.percentile(a, 10), np.percentile(a, 90), len(a)))
    print("gap between a kernel's end and the next kernel's start on the SAME queue (us): median / p10 / p90")
    for k in sorted(gap):
        a = np.array(gap[k])
        print("  %-14s %7.2f %7.2f %7.2f   [%d]" % (k, np.median(a), np.percentile(a, 10), np.percentile(a, 90), len(a)))
    # concurrency: time-weighted number of chain kernels running
    ev = []
    for s, e, k, q in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    run, prev, hist = 0, ev[0][0], defaultdict(float)
    for t, dlt in ev:
        hist[run] += t - prev
        prev = t
        run += dlt
    tot = sum(hist.values())
    print("kernels running at a time (share of the window): " + "  ".join("%d: %.1f %%" % (n, 100 * v / tot) for n, v in sorted(hist.items())))
    nb = sum(1 for r in rows if r[2] == "scan")
    print("batches in the window: %d -> %.2f us per batch; sum of the four kernels' median durations %.1f us, of the median gaps %.1f us"
          % (nb, (t1 - t0) / 1e3 / max(nb, 1), sum(np.median(dur[k]) for k in dur if dur[k]),
             sum(np.median(gap[g]) for g in gap if g in ("prep->match", "match->scan", "scan->tail", "tail->prep", "prep->scan"))))


if __name__ == "__main__":
    main()
