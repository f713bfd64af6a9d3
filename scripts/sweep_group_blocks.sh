for b in 4 16 32; do for g in 1 2 4 8 16; do
  echo "blocks=$b group=$g $(OZCS_CRC_GROUP=$g timeout -k 10 100 python -u bench.py --no-cpu-baseline --blocks $b --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])")"
done; done
