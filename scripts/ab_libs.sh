#!/bin/bash
# A/B of library builds over CRC window sizes: the product library first, then every build named on the command line
# (tools/tune/libozcs_<name>.so from scripts/build_tune.sh).  Timing-only builds give wrong digests, so the sampled
# check of bench.py is off.  usage: BPCS="256 512" scripts/ab_libs.sh <name> ...
for bpc in ${BPCS:-256 512 1024 4096}; do for name in default "$@"; do
  lib=""; [ "$name" != default ] && lib=$PWD/tools/tune/libozcs_$name.so
  OZCS_LIB=$lib timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-check --no-ceiling --type ${TYPE:-CRC32} --bpc $bpc --steps 20 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('bpc=$bpc lib=$name', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
done; done
