#!/bin/bash
# rocprofv3 passes over one bench.py configuration: kernel trace/stats, then FETCH_SIZE and WRITE_SIZE in runs
# of their own.  usage: TAG=name scripts/profile_config.sh <bench.py arguments>; output under gpurun_out/prof_<TAG>_*
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${TAG:-cfg}
BENCH="python3 $PWD/bench.py --no-cpu-baseline $*"
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- $BENCH > $OUT/prof_${TAG}_trace.log 2>&1 || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_${TAG}_$pass -- $BENCH > $OUT/prof_${TAG}_$pass.log 2>&1 || exit 1
done
echo done
