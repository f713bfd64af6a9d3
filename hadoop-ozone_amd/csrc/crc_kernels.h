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

// One window on its own: `len` bytes at `data`, digest to `out` (device pointers).  A list of these describes a
// batch of unrelated chunks (own sizes, own window sizes) for one launch.
struct WindowDesc {
  const uint8_t* data;
  uint8_t* out;
  uint32_t len;
  uint32_t reserved;
};

// CRC of n described windows; total_bytes (the sum of their lengths) guides the choice of lanes per window.
hipError_t crc_launch_desc(const WindowDesc* d_desc, uint32_t n, uint64_t total_bytes, int poly_idx, int num_cus,
                           hipStream_t stream);

// Wavefronts that share one window (1, 2, 4, 8 or 16).
int crc_choose_group(uint64_t n_windows, uint32_t bpc, int num_cus);

hipError_t crc_launch(const CrcLaunch& p, int mode, int group, int num_cus, hipStream_t stream);

}  // namespace ozcs
