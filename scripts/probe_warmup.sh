#!/bin/bash
# Does the measured rate of the short digest launches depend on how many launches precede the timed ones?
set -u
for args in "--type MD5 --bpc 16384 --blocks 32" "--type SHA256 --bpc 16384 --blocks 32" "--type CRC32 --bpc 512" "--type MD5 --blocks 128"; do
  for sw in "3 1" "20 10" "60 30"; do
    set -- $sw
    echo "== $args steps=$1 warmup=$2"
    timeout -k 10 200 python -u bench.py $args --steps $1 --warmup $2 --no-cpu-baseline --no-ceiling 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j['roofline']
print(j['value'], j['ms_per_step'], r['bound'], round(r['frac'], 3), r.get('kernel_ms_avg'), r.get('kernel_ms_min'))" || exit 1
  done
done
