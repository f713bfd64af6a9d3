#!/bin/bash
# Wavefronts per window of the 1 MiB-window CRC kernel against the batch size (256 MiB blocks per launch); "launcher" is
# what crc_choose_group picks.  Two rounds, alternating.
for round in 1 2; do for blocks in ${BLOCKS:-4 16 32 64 128}; do for g in "" 2 4 8 16; do
  OZCS_CRC_GROUP=$g timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-ceiling --blocks $blocks --steps 30 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('round $round blocks=$blocks group=${g:-launcher}', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'], r['check']['ok'])" || exit 1
done; done; done
