#!/bin/bash
# Copy streams of the pinned pipe (OZCS_PIPE_COPY_STREAMS = 1 / 2 / 4): CRC32 only and CRC32 : MD5 = 3 : 1 at 1 MiB windows,
# in-order delivery and any-order harvest.
for n in ${STREAMS:-1 2 4 2 1}; do
  echo "== copy streams $n"
  OZCS_PIPE_COPY_STREAMS=$n timeout -k 10 200 python -u scripts/bench_pipeline.py copy-streams 2>/dev/null || exit 1
done
