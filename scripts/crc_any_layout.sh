#!/bin/bash
# The CRC kernel for layouts that are not gapless (ragged last windows, strides, unaligned bases) on the bench's own
# gapless batch: OZCS_CRC_NO_FLAT=1 forbids the gapless-layout kernel, so the rates show what the per-lane window
# geometry costs.  Columns: window, mode, GB/s with the gapless kernel, GB/s without.
for b in ${SIZES:-256 1024 4096 16384 65536 262144}; do
  for m in compute verify; do
    a=$(timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-ceiling --steps 20 --warmup 10 --type CRC32 --bpc $b --mode $m 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['check']['ok'])") || exit 1
    n=$(OZCS_CRC_NO_FLAT=1 timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-ceiling --steps 20 --warmup 10 --type CRC32 --bpc $b --mode $m 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['check']['ok'])") || exit 1
    echo "bpc=$b $m gapless: $a any layout: $n"
  done
done
