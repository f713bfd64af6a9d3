#!/bin/bash
# (lanes per window, stripes per load group) sweep of the gapless-layout CRC kernel; OZCS_CRC_FLAT = lanes*8 + group.
for bpc in ${BPCS:-4096 2048 1024 512 256}; do for flat in ${FLATS:-20 33 34 65 66 129}; do
  OZCS_CRC_FLAT=$flat timeout -k 10 120 python -u bench.py --no-cpu-baseline --type CRC32C --bpc $bpc --steps 20 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bpc=$bpc lanes=$((flat/8)) group=$((flat%8))', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done; done
