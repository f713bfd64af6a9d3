#!/usr/bin/env python3
"""Builds profiles/pmc_summary.json and the kernel-stats CSV from the rocprofv3 output that
scripts/profile_r01.sh leaves under gpurun_out/ (prof_trace/, prof_pmc_*/), for the headline workload.

HBM bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 + WRITE_SIZE [KiB] x 1024: gfx950 tallies a 128-byte read
request as 64 bytes in FETCH_SIZE (cross-check: TCC_EA0_RDREQ_sum x 128 B)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
KEY = sys.argv[1] if len(sys.argv) > 1 else "CRC32-1048576-64"
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "crc_windows_kernel"
ALG_BYTES = int(sys.argv[3]) if len(sys.argv) > 3 else 64 * 256 * (1 << 20) + 64 * 256 * 4


def counters():
    agg = collections.defaultdict(list)
    kernel = None
    for f in glob.glob(os.path.join(OUT, "prof_pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                kernel = r["Kernel_Name"]
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return kernel, {k: {"launches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(agg.items())}


def kernel_stats():
    for f in glob.glob(os.path.join(OUT, "prof_trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", "r01_kernel_stats_crc32_1MiB_64blocks.csv"))
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Name"]:
                return float(r["AverageNs"]), int(r["Calls"])
    return None, 0


def main():
    kernel, c = counters()
    avg_ns, calls = kernel_stats()
    fetch = c["FETCH_SIZE"]["avg"] * 1024 * 2
    write = c["WRITE_SIZE"]["avg"] * 1024
    rec = {
        "command": "rocprofv3 --kernel-trace --stats / --pmc <one block per run> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
        "kernel": kernel, "kernel_avg_ns_profiled": avg_ns, "kernel_calls_profiled": calls,
        "algorithmic_bytes_per_launch": ALG_BYTES, "hbm_bytes_per_launch": round(fetch + write),
        "fetch_bytes": round(fetch), "write_bytes": round(write),
        "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 tallies 128-B read requests at 64 B; cross-check: "
                      "TCC_EA0_RDREQ_sum x 128 B = %d); WRITE_SIZE [KiB] x 1024" % round(c.get("TCC_EA0_RDREQ_sum", {"avg": 0})["avg"] * 128),
        "counters": c,
    }
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        allrec = json.load(open(path))
    except (OSError, ValueError):
        allrec = {}
    allrec[KEY] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("kernel", "kernel_avg_ns_profiled", "algorithmic_bytes_per_launch", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
