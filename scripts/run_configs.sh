#!/bin/bash
# The bench lines of every benchmark configuration on one GPU, with their CPU baselines.
set -u
run() { echo "== bench.py $*"; timeout -k 10 400 python -u bench.py "$@" 2>/dev/null || exit 1; }
run
run --type CRC32C --bpc 16384 --mode verify
run --type SHA256 --blocks 128 --steps 3 --warmup 1
run --type MD5 --blocks 128 --steps 3 --warmup 1
run --type SHA256 --bpc 16384 --blocks 32 --steps 20 --warmup 10 --no-cpu-baseline
run --type MD5 --bpc 16384 --blocks 32 --steps 20 --warmup 10 --no-cpu-baseline
