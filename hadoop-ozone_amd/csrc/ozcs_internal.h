// Internal state shared by the C-ABI translation units (not installed; the ABI is include/ozone_checksum.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace ozcs {

struct DeviceInfo {
  int id = 0;
  int num_cus = 0;
  // this device's copy of the library's file-scope verdict array (ozcs_verify_device_noalloc), resolved once
  // at init with the device current; null if the lookup failed there
  unsigned long long* noalloc_verdicts = nullptr;
};

// Scratch of one in-flight host-pointer call: a stream and grow-only device buffers.
struct HostCtx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* d_data = nullptr;
  size_t data_cap = 0;
  void* d_sums = nullptr;
  size_t sums_cap = 0;
  void* d_aux = nullptr;
  size_t aux_cap = 0;
  // pinned host staging of the batched call (small chunks and window descriptors; digests on the way back)
  void* h_stage = nullptr;
  size_t h_stage_cap = 0;
  void* h_sums = nullptr;
  size_t h_sums_cap = 0;
  // second copy stream of the batched call's large gathers (every other staged piece goes through it: two copy
  // queues fill the link, one does not quite), created on first use
  hipStream_t side = nullptr;
  hipEvent_t side_done = nullptr;
  // phase timing of sampled calls (PhaseTimer): events created on first use
  hipEvent_t phase[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Times the three phases of a host-pointer call (copies in, kernels, copy back) with events on the context's
// stream, for one call in OZCS_STATS_SAMPLE_EVERY; see ozcs_stats in the ABI header.
class PhaseTimer {
 public:
  PhaseTimer(HostCtx* c, uint64_t bytes);
  void mark(int i);   // i = 0..3, in stream order
  void finish();      // after the stream has been synchronised and the call has succeeded

 private:
  HostCtx* c_;
  uint64_t bytes_;
  bool on_;
};

struct Global {
  std::mutex init_mu;
  std::atomic<bool> inited{false};
  std::vector<DeviceInfo> devs;
  std::mutex mu;  // guards free_ctx
  std::vector<HostCtx*> free_ctx;
  std::atomic<uint64_t> rr{0};
};

Global& global();
int ensure_init();
const DeviceInfo* find_device(int device);

class DeviceGuard {
 public:
  explicit DeviceGuard(int dev);
  ~DeviceGuard();
  bool ok() const { return ok_; }

 private:
  int prev_;
  bool ok_;
};

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail_hip(hipError_t e, const char* what);

HostCtx* acquire_ctx(int device);
void release_ctx(HostCtx* c);
int ctx_reserve(void** buf, size_t* cap, size_t need);
// memcpy into pinned memory, shared with the gather pool's helper threads when large
void pooled_copy(void* dst, const void* src, size_t n);

int launch_windows(const DeviceInfo& dev, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                   uint64_t n_chunks, uint64_t chunk_stride, int mode, void* d_sums, const void* d_expected,
                   uint64_t expected_stride, uint64_t n_expected, uint64_t start_index,
                   unsigned long long* d_first_mismatch, hipStream_t stream);

extern std::atomic<uint64_t> s_calls, s_bytes, s_windows, s_h2d, s_d2h, s_launches;

// Counts a call's outcome for ozcs_get_stats and returns it unchanged.  OZCS_EBUSY is flow control; so is the
// OZCS_EEMPTY of the pipe's "next" calls (empty_is_flow).
int tally(int rc, bool empty_is_flow = false);

}  // namespace ozcs
