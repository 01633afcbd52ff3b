#!/usr/bin/env python3
"""Copies what tools/collect_profiles_r05.sh gathered (gpurun_out/r05_final) into profiles/ under the round's prefix: bench
lines, rocprofv3 kernel-stats CSVs, PMC traffic per workload key (tools/pmc_traffic.py), SQ counters, text records.
usage: publish_profiles_r05.py [gpurun_out/r05_final] [r05]"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r05_final")
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
dst = os.path.join(root, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


for path in sorted(glob.glob(os.path.join(src, "*_bench.json"))):
    shutil.copy(path, os.path.join(dst, f"{tag}_{os.path.basename(path)}"))
for name in ("build_trace", "cells_trace", "dropin_timing", "ingest_timing", "distribution_check", "distribution_check_one_frame", "distribution_check_cell_frames", "deepk_sq_counters", "scan_timeline",
             "pipeline_gaps_rank_0_of_8", "fuzz"):
    p = os.path.join(src, name + ".txt")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}.txt"))
for d, out in (("kt_c3", "c3_kernel_stats"), ("kt_c2", "c2_kernel_stats"), ("kt_c5", "c5_kernel_stats"),
               ("kt_c3_serial", "c3_serial_kernel_stats"), ("kt_2097152_serial", "16_1024_2097152_serial_kernel_stats"),
               ("kt_c3_rank_0_of_8", "c3_rank_0_of_8_kernel_stats"), ("kt_c3_rank_0_of_8_serial", "c3_rank_0_of_8_serial_kernel_stats"),
               ("kt_20_1024_16777216_serial", "20_1024_16777216_serial_kernel_stats"), ("kt_clusters64", "clusters64_kernel_stats"), ("kt_heavy_tail", "heavy_tail_kernel_stats"), ("kt_build", "build_kernel_stats")):
    hits = glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, f"{tag}_{out}.csv"))
for w, args in (("c3", ""), ("c3_rank_0_of_8", "--emulate 8:0"), ("c4_1gpu", "--workload c4"), ("k20", "--workload 20,1024,16777216")):
    if os.path.isdir(os.path.join(src, f"pmc_{w}_fetch")):
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), f"{tag}_{w}",
                               os.path.join(src, f"pmc_{w}_fetch"), os.path.join(src, f"pmc_{w}_write"), os.path.join(src, f"pmc_{w}_l2"),
                               "--bench-json", os.path.join(src, f"pmc_{w}_fetch.json"),
                               "--cmd", f"bench.py --steps 5 --warmup 1 --cpu-queries 0 {args}".strip()], cwd=root,
                              stdout=subprocess.DEVNULL)

lines = ["SQ counters of the C3 bench's kernels, one batch at a time (`bench.py --steps 5 --warmup 1 --cpu-queries 0 --serial` under rocprofv3 --pmc, two passes);",
         "per launch, summed over the chip.  WAVE_CYCLES / WAIT_* / ACTIVE_INST_* count quad-cycles, VALU_MFMA_BUSY_CYCLES cycles (32 per MFMA).", ""]
for d in ("pmc_sq1", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        if not any(t in k for t in ("scan", "match", "prep", "tail")):
            continue
        lines.append("%-46s " % k + "  ".join("%s=%.3g" % (a[3:], sum(v) / len(v)) for a, v in sorted(acc[k].items())))
    lines.append("")
if len(lines) > 4:
    open(os.path.join(dst, f"{tag}_c3_sq_counters.txt"), "w").write("\n".join(lines))
print("published", src, "->", dst)
