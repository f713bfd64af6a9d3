#!/bin/bash
# A/B of two library builds over the small CRC window sizes.  usage: scripts/ab_small_windows.sh <other libozcs.so>
for bpc in ${BPCS:-128 256 512 1024 2048 4096 8192 16384}; do for lib in "" "$1"; do
  OZCS_LIB=$lib timeout -k 10 120 python -u bench.py --no-cpu-baseline --type CRC32 --bpc $bpc --steps 20 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bpc=$bpc lib=${lib:-default}', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done; done
