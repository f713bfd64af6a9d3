// Launch interface of the SHA-256 / MD5 window kernels (digest_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crc_kernels.h"  // WindowDesc

namespace ozcs {

enum { kAlgoSha256 = 0, kAlgoMd5 = 1 };

// Same batch layout as CrcLaunch (crc_kernels.h): chunk c at data + c*chunk_stride, wpc windows of
// bpc bytes per chunk, the last possibly shorter; window g = c*wpc + i.
struct DigestLaunch {
  const uint8_t* data;
  uint64_t chunk_len;
  uint64_t chunk_stride;
  uint64_t n_windows;
  uint64_t wpc;
  uint32_t bpc;
  int algo;  // kAlgoSha256 (32-byte digests) or kAlgoMd5 (16-byte digests)
  uint8_t* sums;
  const uint8_t* expected;
  uint64_t expected_stride;
  uint64_t n_expected;
  uint64_t start_index;
  unsigned long long* first_mismatch;
};

hipError_t digest_launch(const DigestLaunch& p, int mode, int num_cus, hipStream_t stream);

// Unrelated chunks hashed by one launch (compute only).  The job list lies in device-visible memory (device
// memory, or pinned host memory through its device alias), sorted by first_window, and may hold any number of
// chunks: the batched calls and the pinned pipe hash everything of one algorithm in one launch.
struct DigestJob {
  const uint8_t* data;  // device pointer of the chunk
  uint64_t len;
  uint8_t* sums;        // device pointer of its digests
  uint32_t bpc;
  uint32_t first_window;  // index of the chunk's first window in the launch
};
hipError_t digest_launch_jobs(const DigestJob* d_jobs, uint32_t n_jobs, uint32_t n_windows, int algo, int num_cus,
                              hipStream_t stream);

// SHA-256 / MD5 of n described windows (crc_kernels.h), the list in device memory: one lane per window.
hipError_t digest_launch_desc(const WindowDesc* d_desc, uint32_t n, int algo, hipStream_t stream);

}  // namespace ozcs
