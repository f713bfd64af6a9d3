#!/bin/bash
# CRC32 throughput at windows of 128 B to 8 KiB, bench.py defaults (16 GiB per launch); the environment
# (OZCS_CRC_ROWS, OZCS_CRC_LPW, OZCS_CRC_FLAT ...) selects the kernel variant.
for b in ${BPCS:-128 256 512 1024 2048 4096 8192}; do
  timeout -k 10 120 python -u bench.py --no-cpu-baseline --steps 20 --warmup 10 --type CRC32 --bpc $b 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bpc=$b', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done
