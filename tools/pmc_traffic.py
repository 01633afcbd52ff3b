#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT/MISS) per kernel into
profiles/<tag>_pmc_traffic.json.  HBM bytes per launch follow MI355X_MICROARCH.md §HBM:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B stores.

usage: pmc_traffic.py <tag> <fetch_dir> <write_dir> [<l2_dir>] [--key KEY | --bench-json FILE] [--cmd "bench.py ..."]

KEY = the workload key bench.py prints as roofline.pmc_key (shape of the rank's shard, shard mode, rank of N): bench.py quotes
a traffic file only for a run with the same key and the same kernel sources (round 4 quoted the whole-set C3 file for an
emulated rank of eight).  --bench-json: the JSON line one of the passes printed; the key is read from it.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname):
    rows = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(dirname, "*", "*counter_collection.csv")):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0]
                rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    argv = list(sys.argv[1:])
    key, cmd = None, "bench.py --steps 5 --warmup 1 --cpu-queries 0"
    for flag in ("--key", "--bench-json", "--cmd"):
        if flag in argv:
            at = argv.index(flag)
            val = argv[at + 1]
            del argv[at:at + 2]
            if flag == "--key":
                key = val
            elif flag == "--cmd":
                cmd = val
            else:
                with open(val) as f:
                    key = json.loads([ln for ln in f if ln.lstrip().startswith("{")][-1])["roofline"]["pmc_key"]
    tag, fetch_dir, write_dir = argv[0:3]
    l2_dir = argv[3] if len(argv) > 3 else None
    fetch, write = load(fetch_dir), load(write_dir)
    l2 = load(l2_dir) if l2_dir else {}
    out = {}
    for name in sorted(set(fetch) | set(write)):
        f = fetch.get(name, {}).get("FETCH_SIZE", [])
        w = write.get(name, {}).get("WRITE_SIZE", [])
        # steady-state launches: drop the first (cold) one when there are several
        fs = f[1:] if len(f) > 2 else f
        ws = w[1:] if len(w) > 2 else w
        favg = sum(fs) / len(fs) if fs else 0.0
        wavg = sum(ws) / len(ws) if ws else 0.0
        ent = {"launches": len(f), "FETCH_SIZE_KiB_avg": favg, "WRITE_SIZE_KiB_avg": wavg,
               "hbm_read_bytes_per_launch": 2.0 * favg * 1024.0, "hbm_write_bytes_per_launch": wavg * 1024.0,
               "hbm_bytes_per_launch": 2.0 * favg * 1024.0 + wavg * 1024.0}
        if name in l2:
            h = sum(l2[name].get("TCC_HIT_sum", [0.0]))
            m = sum(l2[name].get("TCC_MISS_sum", [0.0]))
            ent["l2_hit_rate"] = h / (h + m) if h + m else None
        out[name] = ent
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")
    # bench.py quotes roofline.traffic from this file only while the kernels it was measured on are the ones running
    import hashlib
    h = hashlib.sha256()
    for name in ("knn_filter.hip", "knn_cells.hip", "knn_filter_dev.h", "knn_exact.hip", "knn_exact_dev.h", "knn_common.h"):
        with open(os.path.join(root, "multicore_hw2_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    with open(path, "w") as fo:
        json.dump({"note": "rocprofv3 --pmc, separate passes; read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 correction, "
                           "MI355X_MICROARCH.md HBM section); command: " + cmd,
                   "workload_key": key,
                   "kernel_source_sha256": h.hexdigest(),
                   "kernels": out}, fo, indent=1)
    for k, v in out.items():
        print(f"{k[:60]:60s} n={v['launches']:3d} read {v['hbm_read_bytes_per_launch']/1e6:10.2f} MB write {v['hbm_write_bytes_per_launch']/1e6:9.2f} MB"
              + (f"  L2 hit {v['l2_hit_rate']:.3f}" if v.get('l2_hit_rate') is not None else ""))


if __name__ == "__main__":
    main()
