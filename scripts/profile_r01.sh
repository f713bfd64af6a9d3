#!/bin/bash
# Profiles the headline bench with rocprofv3 on the GPU box: a kernel-trace/stats pass, then PMC passes
# (one counter block per run).  Output under gpurun_out/prof_*; summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
BENCH="python3 $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- $BENCH > $OUT/prof_trace.log 2>&1 || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_pmc_$name -- $BENCH > $OUT/prof_pmc_$name.log 2>&1 || exit 1
done
echo done
