#!/bin/bash
# PMC passes over one digest bench configuration (arguments are passed to bench.py), one counter block per run.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${TAG:-digest}
BENCH="python3 $PWD/bench.py --no-cpu-baseline $*"
cd /tmp
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_${TAG}_$name -- $BENCH > $OUT/prof_${TAG}_$name.log 2>&1 || exit 1
done
echo done
