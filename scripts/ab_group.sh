#!/bin/bash
# Interleaved A/B of wavefronts-per-window on the headline workload: ROUNDS passes over the variants.
for r in $(seq 1 ${ROUNDS:-3}); do for g in ${GROUPS_:-1 2 4 8}; do
  OZCS_CRC_GROUP=$g timeout -k 10 120 python -u bench.py --no-cpu-baseline --blocks ${BLOCKS:-32} --steps 30 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('round $r G=$g', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done; done
