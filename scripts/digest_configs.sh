#!/bin/bash
# SHA-256 / MD5 throughput: every SIMD busy (16 KiB windows, 8 GiB) and the 32 GiB @ 1 MiB configuration.
run() {
  timeout -k 10 170 python -u bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', r['value'], r['roofline']['kernel_ms_avg'])" || exit 1
}
for t in MD5 SHA256; do
  run --type $t --bpc 16384 --blocks 32 --steps 20 --warmup 10
  run --type $t --bpc 65536 --blocks 32 --steps 20 --warmup 10
  run --type $t --blocks 128 --steps 2 --warmup 1
done
