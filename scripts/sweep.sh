#!/bin/bash
# A/B of CRC kernel tuning knobs on the headline workload (each line: group, prefetch depth, GB/s, kernel ms).
for g in ${GROUPS_:-1 2 4 8 16}; do for pf in ${PFS:-2 4}; do
  OZCS_CRC_GROUP=$g OZCS_CRC_PF=$pf timeout -k 10 120 python -u bench.py --no-cpu-baseline --blocks ${BLOCKS:-32} --steps 20 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('G=$g PF=$pf', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done; done
