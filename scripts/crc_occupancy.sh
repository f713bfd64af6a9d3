#!/bin/bash
# Small CRC windows with one and two workgroups per CU (tools/crc_occupancy_bench.hip), every window size the gapless
# small-window kernel serves.  Run on the GPU box; output is the record kept under profiles/.
set -u
for bpc in ${SIZES:-128 256 512 1024 2048 4096}; do
  timeout -k 10 120 tools/crc_occupancy_bench $bpc ${GIB:-8} || exit 1
done
