#!/usr/bin/env python3
"""Turns the rocprofv3 output of scripts/profile_config.sh (gpurun_out/prof_<TAG>_*) into an entry of
profiles/pmc_summary.json and a kernel-stats CSV under profiles/.

    scripts/summarize_pmc.py TAG KEY KERNEL_SUBSTRING ALGORITHMIC_BYTES [WARMUP_STEPS [WAVE_BLOCKS [LAUNCHES_PER_STEP]]]

KEY is "<type>-<bpc>-<blocks>" as bench.py looks it up.  HBM bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 +
WRITE_SIZE [KiB] x 1024: gfx950 tallies a 128-byte read request as 64 bytes in FETCH_SIZE (cross-check:
TCC_EA0_RDREQ_sum x 128 B).  The entry is stamped with the hash of the kernel sources it was taken from; bench.py
ignores entries whose stamp differs from the sources of the library it runs.  WAVE_BLOCKS (digests): wavefronts x
64-byte blocks per window of one launch; SQ_INSTS_VALU divided by it is the vector instructions per block.
LAUNCHES_PER_STEP: a step that the launcher cuts into several launches of the same kernel (small-window CRC compute
holds a wavefront's digests in registers and a launch is as long as those last); counters and times of that many
consecutive launches are added up, so that every figure of the entry is per step, as bench.py's are."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
TAG, KEY, KERNEL, ALG_BYTES = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
WARMUP = int(sys.argv[5]) if len(sys.argv) > 5 else 0
WAVE_BLOCKS = int(sys.argv[6]) if len(sys.argv) > 6 else 0
PER_STEP = int(sys.argv[7]) if len(sys.argv) > 7 else 1


def per_step(values):
    """Sums of PER_STEP consecutive launches (an incomplete last group is dropped)."""
    return [sum(values[i:i + PER_STEP]) for i in range(0, len(values) - PER_STEP + 1, PER_STEP)]


def one_run_only(files):
    """gpurun merges a call's output into what earlier calls left behind, under file names that carry a process id: two
    files of one kind in one pass's directory are two runs, and a summary over both would describe neither."""
    seen = {}
    for f in files:
        d = os.path.dirname(f)
        if d in seen:
            sys.exit("summarize_pmc: %s and %s are from different runs; delete gpurun_out/prof_* and profile again" % (seen[d], f))
        seen[d] = f
    return files


def counters():
    agg = collections.defaultdict(list)
    kernel = None
    for f in one_run_only(sorted(glob.glob(os.path.join(OUT, "prof_%s_pmc_*" % TAG, "**", "*counter_collection.csv"), recursive=True))):
        rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            kernel = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg = {k: per_step(v) for k, v in agg.items()}
    return kernel, {k: {"launches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(agg.items()) if v}


def kernel_times():
    """(average over all profiled launches as rocprofv3 --stats reports it, calls, average without the first WARMUP
    launches from the kernel trace)."""
    avg_ns, calls, warm = None, 0, None
    for f in one_run_only(glob.glob(os.path.join(OUT, "prof_%s_trace" % TAG, "**", "*kernel_stats.csv"), recursive=True)):
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.csv" % (os.environ.get("ROUND", "r03"), TAG)))
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Name"] and int(r["Calls"]) > calls * PER_STEP:   # the variant the run is about: the one launched most
                avg_ns, calls = float(r["AverageNs"]) * PER_STEP, int(r["Calls"]) // PER_STEP
    for f in one_run_only(glob.glob(os.path.join(OUT, "prof_%s_trace" % TAG, "**", "*kernel_trace.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = per_step([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows])[WARMUP:]
        if d:
            warm = sum(d) / len(d)
    return avg_ns, calls, warm


def main():
    from hadoop_ozone_amd import native
    kernel, c = counters()
    avg_ns, calls, warm_ns = kernel_times()
    rec = {
        "command": "scripts/profile_config.sh (rocprofv3 --kernel-trace --stats; --pmc, one block per run) -- python3 bench.py "
                   "--no-cpu-baseline --no-ceiling ...",
        "source_hash": native.source_hash(),
        "kernel": kernel, "kernel_avg_ns_profiled": avg_ns, "kernel_calls_profiled": calls,
        "kernel_avg_ns_after_warmup": warm_ns, "warmup_launches_excluded": WARMUP, "launches_per_step": PER_STEP,
        "algorithmic_bytes_per_launch": ALG_BYTES, "counters": c,
    }
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = c["FETCH_SIZE"]["avg"] * 1024 * 2
        write = c["WRITE_SIZE"]["avg"] * 1024
        rec.update({
            "hbm_bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write),
            "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 tallies 128-B read requests at 64 B; cross-check: "
                          "TCC_EA0_RDREQ_sum x 128 B = %d); WRITE_SIZE [KiB] x 1024"
                          % round(c.get("TCC_EA0_RDREQ_sum", {"avg": 0})["avg"] * 128)})
    if WAVE_BLOCKS and "SQ_INSTS_VALU" in c:
        rec["valu_per_block_lane"] = round(c["SQ_INSTS_VALU"]["avg"] / WAVE_BLOCKS, 1)
        rec["wave_blocks_per_launch"] = WAVE_BLOCKS
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]["avg"]:
        rec["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"]["avg"] / c["SQ_LDS_IDX_ACTIVE"]["avg"], 5)
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        allrec = json.load(open(path))
    except (OSError, ValueError):
        allrec = {}
    allrec[KEY] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    print(json.dumps({k: rec.get(k) for k in ("kernel", "kernel_avg_ns_profiled", "kernel_avg_ns_after_warmup",
                                               "algorithmic_bytes_per_launch", "hbm_bytes_per_launch", "valu_per_block_lane",
                                               "lds_bank_conflict_frac", "source_hash")}))


if __name__ == "__main__":
    main()
