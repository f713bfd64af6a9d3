#!/bin/bash
# Differential fuzzing on the GPU box: tests/variant_check.py (random batch shapes, every digest against the oracle, verify
# with one flipped bit) over many seeds, with the launcher's own kernel choice and with every forced variant.
# usage: SEEDS="100 101 ..." scripts/fuzz_parity.sh   (prints one line per run; exit 1 at the first mismatch)
set -u
run() {  # env assignments..., then: seed [mode]
  out=$(env "${@:1:$#-2}" timeout -k 10 170 python tests/variant_check.py "${@: -2:1}" "${@: -1}" 2>&1) || { echo "$out" | tail -20; echo "FAILED: $*"; exit 1; }
  echo "$out" | tail -1
}
for seed in ${SEEDS:-101 102 103 104 105 106 107 108}; do
  run X=1 $seed crc
  run X=1 $seed digests
  run X=1 $seed pow2
  for g in 1 2 4 8 16; do run OZCS_CRC_LPW=64 OZCS_CRC_GROUP=$g $seed crc; done
  for l in 4 8 16 32; do run OZCS_CRC_LPW=$l $seed crc; run OZCS_CRC_LPW=$l OZCS_CRC_NO_FLAT=1 $seed crc; done
  for l in 8 16 32; do run OZCS_CRC_LPW=$l OZCS_CRC_PHASES=4 $seed pow2; done
  run OZCS_CRC_LPW=8 OZCS_CRC_NO_ALIGNED=1 $seed crc
  for h in 1 -1; do run OZCS_CRC_HOLD=$h $seed crc; run OZCS_CRC_HOLD=$h $seed pow2; done
  run OZCS_SHA_SPLIT=0 OZCS_MD5_LINES=0 OZCS_SHA_LINES=0 $seed digests
  run OZCS_SHA_SPLIT=0 OZCS_MD5_LINES=1 OZCS_SHA_LINES=1 $seed digests
  run OZCS_SHA_SPLIT=1 OZCS_MD5_LINES=1 OZCS_SHA_LINES=1 $seed digests
done
echo "fuzz done"
