#!/bin/bash
# CRC32 at every window size from 128 B to 1 MiB, compute and verify, 16 GiB per launch, twenty launches after ten;
# columns: window, mode, GB/s, kernel ms average, kernel ms minimum, sampled digests / verdicts correct.
for b in 128 256 512 1024 2048 4096 8192 16384 32768 65536 131072 262144 524288 1048576; do
  for m in compute verify; do
    timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-ceiling --steps 20 --warmup 10 --type CRC32 --bpc $b --mode $m 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bpc=$b $m', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'], r['check']['ok'])" || exit 1
  done
done
