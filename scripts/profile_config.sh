#!/bin/bash
# rocprofv3 passes over one bench.py configuration on the GPU box: a kernel-trace/stats pass, then PMC passes, one
# counter block per run (never combined with a trace).  usage: TAG=name scripts/profile_config.sh <bench.py arguments>
# Output under gpurun_out/prof_<TAG>_*; scripts/summarize_pmc.py turns it into profiles/ entries.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${TAG:-headline}
# no child processes under the profiler (its preloaded library has initialised the GPU before python starts): the
# bare-read ceiling, which bench.py measures in a child, is switched off here
BENCH="python3 $PWD/bench.py --no-cpu-baseline --no-ceiling $*"
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- $BENCH > $OUT/prof_${TAG}_trace.log 2>&1 || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
            "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_${TAG}_pmc_$name -- $BENCH > $OUT/prof_${TAG}_pmc_$name.log 2>&1 || exit 1
done
echo done $TAG
