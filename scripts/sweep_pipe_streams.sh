#!/bin/bash
# Digest-stream pool of the pinned pipe: number of streams x stream priority, CRC32:MD5 3:1 at 1 MiB windows.
for prio in 1 0; do for n in 1 2 4 8; do
  echo "== digest streams $n, low priority $prio"
  OZCS_PIPE_DIGEST_STREAMS=$n OZCS_PIPE_DIGEST_LOW_PRIO=$prio timeout -k 10 120 python -u scripts/bench_pipeline.py mix || exit 1
done; done
