#!/bin/bash
# The headline workload at 1 / 4 / 16 / 64 / 256 resident blocks of 256 MiB (0.25 ... 64 GiB per launch), random data,
# and 64 blocks of zeros: CRC32 at 1 MiB windows, bench.py's own timing (40 launches after 10 warm-ups: short launches need the longer run-in, the first ones run at lower clocks).
for b in 1 4 16 64 256; do
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --blocks $b --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('blocks=$b GiB=%.2f' % ($b/4), 'GB/s', r['value'], 'kernel_ms_avg', r['roofline']['kernel_ms_avg'], 'kernel_ms_min', r['roofline']['kernel_ms_min'], 'frac', r['roofline']['frac'])" || exit 1
done
