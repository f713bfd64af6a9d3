#!/bin/bash
# 64-byte against whole-line loader steps of the one-wavefront MD5 / SHA-256 kernels, with enough launches in front of
# the timed ones for the chip to reach the clock it holds (a 2 ms launch after an idle second runs far below it).
set -u
for t in MD5 SHA256; do
  v=OZCS_MD5_LINES; [ $t = SHA256 ] && v=OZCS_SHA_LINES
  for bpc in 4096 16384 65536 262144; do
    for lines in 0 1; do
      r=$(env $v=$lines OZCS_SHA_SPLIT=0 timeout -k 10 120 python bench.py --type $t --bpc $bpc --blocks 32 --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null) || exit 1
      echo "$t bpc=$bpc lines=$lines $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("GB/s=%.0f valu_frac=%.3f kernel_ms_avg=%.4f min=%.4f check=%s" % (d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel_ms_min"], d["check"]["ok"]))')"
    done
  done
done
