#!/bin/bash
# rocprofv3 trace + PMC passes of the headline, the 16 KiB verify, the SHA-256 32 GiB, two 16 KiB digest and two small-window
# configurations (first part of scripts/final_measure.sh; output under gpurun_out/prof_<tag>_*).
set -u
O=gpurun_out
export ROUND=r03
TAG=headline bash scripts/profile_config.sh --steps 30 --warmup 10 > $O/prof_headline.log 2>&1 || exit 1
TAG=cfg3 bash scripts/profile_config.sh --type CRC32C --bpc 16384 --mode verify --steps 20 --warmup 10 > $O/prof_cfg3.log 2>&1 || exit 1
TAG=cfg4 bash scripts/profile_config.sh --type SHA256 --blocks 128 --steps 3 --warmup 1 > $O/prof_cfg4.log 2>&1 || exit 1
TAG=md51m bash scripts/profile_config.sh --type MD5 --blocks 128 --steps 3 --warmup 1 > $O/prof_md51m.log 2>&1 || exit 1
TAG=sha16k bash scripts/profile_config.sh --type SHA256 --bpc 16384 --blocks 32 --steps 20 --warmup 10 > $O/prof_sha16k.log 2>&1 || exit 1
TAG=md516k bash scripts/profile_config.sh --type MD5 --bpc 16384 --blocks 32 --steps 20 --warmup 10 > $O/prof_md516k.log 2>&1 || exit 1
TAG=flat512 bash scripts/profile_config.sh --type CRC32 --bpc 512 --steps 20 --warmup 10 > $O/prof_flat512.log 2>&1 || exit 1
TAG=flat16k bash scripts/profile_config.sh --type CRC32 --bpc 16384 --steps 20 --warmup 10 > $O/prof_flat16k.log 2>&1 || exit 1
echo profiles done
