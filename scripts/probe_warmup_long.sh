#!/bin/bash
# Does a bench line move when the untimed launches in front of it cover 100-250 ms instead of 25 ms?  (The occupancy
# experiment found the first 100-200 ms of a process's GPU work reading 3-5 % low.)
set -u
line() { python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%s warmup=%s: %.0f GB/s  frac %.4f" % (sys.argv[1], d["warmup"], d["value"], d["roofline"]["frac"]))' "$1"; }
for w in 10 100; do
  timeout -k 10 170 python bench.py --warmup $w --no-cpu-baseline 2>/dev/null | line "headline" || exit 1
  for bpc in 512 1024 16384; do
    timeout -k 10 170 python bench.py --type CRC32 --bpc $bpc --warmup $w --no-cpu-baseline 2>/dev/null | line "CRC32 bpc=$bpc" || exit 1
  done
  timeout -k 10 170 python bench.py --type CRC32C --bpc 16384 --mode verify --warmup $w --no-cpu-baseline 2>/dev/null | line "CRC32C verify 16 KiB" || exit 1
done
