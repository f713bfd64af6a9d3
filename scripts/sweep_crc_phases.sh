#!/bin/bash
# CRC32 at power-of-two windows from 4 to 64 KiB: lanes per window x (no phases | four phases), 16 GiB per launch.
# Usage (GPU box): [BPCS=..] [LANES=..] [UNRS=..] bash scripts/sweep_crc_phases.sh > gpurun_out/crc_phases_sweep.log
set -u
for bpc in ${BPCS:-4096 8192 16384 32768 65536}; do
  for lanes in ${LANES:-8 16 32}; do
   for unr in ${UNRS:-1}; do
    for ph in 1 4; do
      r=$(OZCS_CRC_FLAT=$((lanes * 8 + unr)) OZCS_CRC_LPW=$lanes OZCS_CRC_PHASES=$ph timeout -k 10 120 python bench.py --type CRC32 --bpc $bpc --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null) || exit 1
      echo "bpc=$bpc lanes=$lanes stripes_per_group=$unr phases=$ph $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("GB/s=%.0f kernel_ms=%.4f" % (d["roofline"]["achieved"], d["roofline"]["kernel_ms_avg"]))')"
    done
   done
  done
done
