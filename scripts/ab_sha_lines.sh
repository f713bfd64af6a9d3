for lines in 0 1; do
  OZCS_SHA_LINES=$lines OZCS_SHA_SPLIT=0 timeout -k 10 170 python tests/variant_check.py 11 digests > /dev/null 2>&1 && echo "lines=$lines parity ok" || echo "lines=$lines PARITY FAIL"
  for bpc in 4096 16384 65536; do
    r=$(OZCS_SHA_LINES=$lines timeout -k 10 120 python bench.py --type SHA256 --bpc $bpc --blocks 32 --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null) || exit 1
    echo "lines=$lines bpc=$bpc $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("GB/s=%.0f valu_frac=%.3f" % (d["value"], d["roofline"]["frac"]))')"
  done
done
