#!/bin/bash
# Copies the measurement summaries (files named <round>_*) from gpurun_out/ into profiles/ and rebuilds
# profiles/pmc_summary.json from the rocprofv3 passes (run here after scripts/final_measure.sh ran on the GPU box).
set -eu
R=${ROUND:-r03}
export ROUND=$R
for f in ${R}_configs.log ${R}_crc_window_sizes.log ${R}_pipeline_end_to_end.log ${R}_host_sync_calls.log ${R}_call_latency.log \
         ${R}_small_chunk_batches.log ${R}_bench_default.json; do
  if [ -f gpurun_out/$f ]; then cp gpurun_out/$f profiles/$f; fi   # the bench line is taken last, after a first collect
done
MIB=1048576
H=$((64*256*MIB)); 
python scripts/summarize_pmc.py headline CRC32-1048576-64 crc_windows_kernel $((H + 64*256*4)) 10
python scripts/summarize_pmc.py cfg3 CRC32C-16384-64 crc_flat_kernel $((H + H/16384*4 + H/4194304*8)) 11
python scripts/summarize_pmc.py cfg4 SHA256-1048576-128 sha256_split_windows_kernel $((2*H + 2*H/MIB*32)) 1 $((2*H/MIB/64*16385))
python scripts/summarize_pmc.py md51m MD5-1048576-128 digest_windows_kernel $((2*H + 2*H/MIB*16)) 1 $((2*H/MIB/64*16385))
python scripts/summarize_pmc.py sha16k SHA256-16384-32 digest_windows_kernel $((H/2 + H/2/16384*32)) 10 $((H/2/16384/64*257))
python scripts/summarize_pmc.py md516k MD5-16384-32 digest_windows_kernel $((H/2 + H/2/16384*16)) 10 $((H/2/16384/64*257))
# 512-byte windows: 32 Mi windows a step, 4 Mi windows a launch (8 collectors x 8 passes per wavefront): 8 launches a step
python scripts/summarize_pmc.py flat512 CRC32-512-64 crc_flat_kernel $((H + H/512*4)) 10 0 8
python scripts/summarize_pmc.py flat16k CRC32-16384-64 crc_flat_kernel $((H + H/16384*4)) 10
