// Launch interface of the CRC32 / CRC32C window kernels (crc_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ozcs {

enum { kModeCompute = 0, kModeVerify = 1 };

// A batch of equally laid out chunks: chunk c occupies [data + c*chunk_stride, +chunk_len) and is
// cut into wpc = ceil(chunk_len / bpc) windows from its offset 0, the last possibly shorter
// (Checksum.java:169-175).  Window g = c*wpc + i.
struct CrcLaunch {
  const uint8_t* data;
  uint64_t chunk_len;
  uint64_t chunk_stride;
  uint64_t n_windows;  // n_chunks * wpc
  uint64_t wpc;
  int wpc_shift;  // log2(wpc) when wpc is a power of two, else -1; filled in by crc_launch
  uint32_t bpc;
  int poly_idx;  // 0 = CRC32, 1 = CRC32C
  // compute: 4 big-endian bytes per window at sums + 4*g
  uint8_t* sums;
  // verify: chunk c's stored list starts at expected + c*expected_stride and holds n_expected
  // digests; computed[i] is compared with stored[start_index + i].  first_mismatch[c] receives the
  // smallest failing i (atomicMin; the caller presets it to all ones = "none").
  const uint8_t* expected;
  uint64_t expected_stride;
  uint64_t n_expected;
  uint64_t start_index;
  unsigned long long* first_mismatch;
};

// Wavefronts that share one window (1, 2, 4, 8 or 16).
int crc_choose_group(uint64_t n_windows, uint32_t bpc, int num_cus);

hipError_t crc_launch(const CrcLaunch& p, int mode, int group, int num_cus, hipStream_t stream);

}  // namespace ozcs
