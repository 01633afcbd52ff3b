#!/usr/bin/env python3
"""Copies what tools/collect_profiles.sh gathered (gpurun_out/<dir>) into profiles/ under the round's prefix:
bench lines, rocprofv3 kernel-stats CSVs, PMC traffic (tools/pmc_traffic.py), SQ counters per kernel, text records.
usage: publish_profiles.py gpurun_out/r04_final r04"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


for path in sorted(glob.glob(os.path.join(src, "*_bench.json"))):
    shutil.copy(path, os.path.join(dst, f"{tag}_{os.path.basename(path)}"))
for name in ("build_trace", "cells_trace", "dropin_timing", "ingest_timing", "distribution_check"):
    shutil.copy(os.path.join(src, name + ".txt"), os.path.join(dst, f"{tag}_{name}.txt"))
for name in ("deepk_sq_counters", "scan_timeline_final"):
    if os.path.exists(os.path.join(src, name + ".txt")):
        shutil.copy(os.path.join(src, name + ".txt"), os.path.join(dst, f"{tag}_{name}.txt"))
for d, out in (("kt_c3", "c3_kernel_stats"), ("kt_c2", "c2_kernel_stats"), ("kt_c5", "c5_kernel_stats"),
               ("kt_c3_serial", "c3_serial_kernel_stats"),
               ("kt_2097152_serial", "16_1024_2097152_serial_kernel_stats"),
               ("kt_c3_rank_0_of_8", "c3_rank_0_of_8_kernel_stats"),
               ("kt_c3_rank_0_of_8_serial", "c3_rank_0_of_8_serial_kernel_stats"),
               ("kt_k1024", "1024_16384_65536_kernel_stats"),
               ("kt_clusters64", "clusters64_kernel_stats"), ("kt_heavy_tail", "heavy_tail_kernel_stats")):
    shutil.copy(one(f"{d}/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_{out}.csv"))
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), f"{tag}_c3",
                       os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), os.path.join(src, "pmc_l2")], cwd=root)

lines = ["SQ counters of the C3 bench's kernels, one batch at a time (`bench.py --steps 5 --warmup 1 --cpu-queries 0 --serial` under rocprofv3 --pmc, two passes);",
         "per launch, summed over the chip.  WAVE_CYCLES / WAIT_* / ACTIVE_INST_* count quad-cycles, VALU_MFMA_BUSY_CYCLES cycles (32 per MFMA).", ""]
for d in ("pmc_sq1", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        if not any(t in k for t in ("scan", "match", "prep", "rerank", "cells_exact")):
            continue
        lines.append("%-46s " % k + "  ".join("%s=%.3g" % (a[3:], sum(v) / len(v)) for a, v in sorted(acc[k].items())))
    lines.append("")
open(os.path.join(dst, f"{tag}_c3_sq_counters.txt"), "w").write("\n".join(lines))
print("published", src, "->", dst)
