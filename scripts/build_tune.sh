#!/bin/bash
# Builds a variant of the library with extra compiler flags (timing-only switches such as -DOZCS_TUNE_NOLOAD, or an
# A/B form of a kernel) as tools/tune/libozcs_<name>.so, for scripts/ab_small_windows.sh and friends (OZCS_LIB).
# usage: scripts/build_tune.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
out=tools/tune/$name
mkdir -p $out
src=hadoop-ozone_amd/csrc
for f in crc_kernels digest_kernels ozcs_api ozcs_pipe; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fconstexpr-steps=100000000 -fvisibility=hidden \
    --offload-arch=gfx950 $flags -c $src/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tune/libozcs_$name.so $out/*.o -lpthread
rm -rf $out
echo built tools/tune/libozcs_$name.so
