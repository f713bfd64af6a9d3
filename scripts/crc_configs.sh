#!/bin/bash
# CRC throughput at the window sizes of the reference's own benchmark (BenchMarkCRCStreaming.java:69:
# 512, 1024, 2048, 4096, 32768, 1048576) plus the 16 KiB verify configuration, bench.py defaults (16 GiB).
run() {
  timeout -k 10 120 python -u bench.py --no-cpu-baseline --steps 20 --warmup 10 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])" || exit 1
}
run --type CRC32
run --type CRC32C
run --type CRC32C --bpc 16384 --mode verify
for b in 512 1024 2048 4096 16384 32768 65536 131072 262144 524288; do run --type CRC32 --bpc $b; done
