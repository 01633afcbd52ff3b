#!/usr/bin/env python3
"""Where the pruned scan's launch goes: five wall-clock stamps per wave (entry, LDS filled, items done, own records
re-ranked, exit) of the LAST scan launch, from a private build of the library with -DKNN_SCAN_TIMELINE
(tools/build_timeline_lib.sh -> tools/libknn_timeline.so; the product library carries no stamps).

usage (GPU box):  KNN_MI355X_LIB=$PWD/tools/libknn_timeline.so python bench.py [--emulate N:r] --scan-stamps /tmp/s.npz ...
                  python tools/scan_timeline.py /tmp/s.npz
"""
import sys

import numpy as np


def show(tag, st):
    st = st.astype(np.int64)
    st = st[st[:, 0] != 0]
    st = st[(st > 0).all(axis=1)]
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0   # 100 MHz
    print("%s: %d waves stamped; us after the first wave's entry: min / median / p90 / max" % (tag, len(st)))
    for i, nm in enumerate(["entry", "filled", "items done", "re-ranked", "exit"]):
        c = us[:, i]
        print("  %-11s %7.2f %7.2f %7.2f %7.2f" % (nm, c.min(), np.median(c), np.percentile(c, 90), c.max()))
    d = np.diff(us, axis=1)
    print("  per-wave durations (us): min / median / p90 / max")
    for i, nm in enumerate(["fill", "items", "re-rank", "end"]):
        c = d[:, i]
        print("  %-11s %7.2f %7.2f %7.2f %7.2f" % (nm, c.min(), np.median(c), np.percentile(c, 90), c.max()))


if __name__ == "__main__":
    z = np.load(sys.argv[1])
    for tag in z.files:
        show(tag, z[tag])
