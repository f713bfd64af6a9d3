#!/bin/bash
# Lanes-per-window sweep of the small-window CRC kernel (64 = wavefront-per-window kernel).
# NOAL=1 also times the any-alignment code path on the same (aligned) data.
for bpc in ${BPCS:-16384 4096 1024 512}; do for lpw in ${LPWS:-4 8 16 32 64}; do for noal in 0 ${NOAL:+1}; do
  OZCS_CRC_NO_ALIGNED=$noal OZCS_CRC_LPW=$lpw timeout -k 10 120 python -u bench.py --no-cpu-baseline --type CRC32C --bpc $bpc --steps 20 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bpc=$bpc LPW=$lpw noal=$noal', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done; done; done
