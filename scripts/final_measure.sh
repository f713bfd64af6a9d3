#!/bin/bash
# The full measurement record on one GPU: bench lines of every configuration, CRC window-size table, pipeline, per-call rates,
# then rocprofv3 trace + PMC passes of the headline, the 16 KiB verify, the SHA-256 32 GiB and two small-window
# configurations.  Output under gpurun_out/; scripts/collect_round.sh copies the summaries into profiles/.
set -u
O=gpurun_out
R=${ROUND:-r03}
export ROUND=$R
# Launches of a few milliseconds are preceded by ten untimed ones: after an idle second the chip runs the first launches of a
# sequence well below the clock it then holds (MD5 at 16 KiB windows: 4.1 TB/s over launches 2-4, 5.4 over launches 11-30).
# rocprofv3 passes first, and their summary: every bench line after them carries the measured traffic of these very sources
# (SKIP_PROFILES=1: the passes ran in an earlier call and profiles/pmc_summary.json already holds their summary)
if [ "${SKIP_PROFILES:-0}" != 1 ]; then
  bash scripts/final_profiles.sh > $O/final_profiles.log 2>&1 || exit 1
  bash scripts/collect_round.sh > $O/collect.log 2>&1 || exit 1
fi
bash scripts/run_configs.sh > $O/${R}_configs.log 2>&1 || exit 1
bash scripts/crc_configs.sh > $O/${R}_crc_window_sizes.log 2>&1 || exit 1
timeout -k 10 400 python -u scripts/bench_pipeline.py 2>/dev/null > $O/${R}_pipeline_end_to_end.log || exit 1
timeout -k 10 300 python -u scripts/bench_host_sync.py 2>/dev/null > $O/${R}_host_sync_calls.log || exit 1
timeout -k 10 300 python -u scripts/bench_latency.py 2>/dev/null > $O/${R}_call_latency.log || exit 1
timeout -k 10 300 python -u scripts/bench_small_batch.py 2>/dev/null > $O/${R}_small_chunk_batches.log || exit 1
timeout -k 10 200 python -u bench.py > $O/${R}_bench_default.json 2>/dev/null || exit 1
echo final measure done
