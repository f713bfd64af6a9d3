// C ABI of the library (include/ozone_checksum.h): lifetime, the synchronous host-pointer calls,
// the batched host call and the device-resident calls.  All compute goes to the HIP kernels; there
// is no CPU fallback in this library -- without a usable GPU every call returns OZCS_ENODEV and the
// caller keeps its own path (as NativeCRC32Wrapper.isAvailable() gates the reference's native CRC,
// NativeCRC32Wrapper.java:40-42).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ozone_checksum.h"
#include "crc_kernels.h"
#include "digest_kernels.h"
#include "ozcs_internal.h"

// ---- verdict words of ozcs_verify_device_noalloc: a file-scope device array instead of an allocation ----
// Every GPU the library runs on holds its own copy of this array at its own address until the process ends.
// In host code the name below is only a handle for hipGetSymbolAddress: the address kernels and copies use is
// the per-device pointer resolved in ozcs_init (DeviceInfo::noalloc_verdicts), never the name.
static __device__ __align__(256) unsigned long long scratch[OZCS_NOALLOC_MAX_CHUNKS];

extern "C" {
static int init_impl(uint32_t device_mask);
static int verify_batch_impl(ozcs_vchunk* v, uint32_t n_chunks);
}

namespace ozcs {

thread_local std::string t_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_err = buf;
  return code;
}

int fail_hip(hipError_t e, const char* what) {
  return fail(OZCS_EHIP, "%s: %s", what, hipGetErrorString(e));
}

Global& global() {
  static Global g;
  return g;
}

std::atomic<uint64_t> s_calls{0}, s_bytes{0}, s_windows{0}, s_h2d{0}, s_d2h{0}, s_launches{0};
std::atomic<uint64_t> s_errors{0}, s_unavailable{0};
std::atomic<uint64_t> s_timed_calls{0}, s_timed_bytes{0}, s_h2d_us{0}, s_kernel_us{0}, s_d2h_us{0};

int tally(int rc, bool empty_is_flow) {
  if (rc >= 0 || rc == OZCS_EBUSY || (empty_is_flow && rc == OZCS_EEMPTY)) return rc;
  s_errors.fetch_add(1, std::memory_order_relaxed);
  if (rc == OZCS_ENODEV || rc == OZCS_EHIP || rc == OZCS_ENOMEM) s_unavailable.fetch_add(1, std::memory_order_relaxed);
  return rc;
}

// ---- phase timing of the host-pointer calls, one call in OZCS_STATS_SAMPLE_EVERY ----
// Four events on the call's stream: before the copies in, after them, after the kernels, after the copy back.
// The first call of the process is timed (so that a process that makes a handful of calls has figures at all),
// then every OZCS_STATS_SAMPLE_EVERY-th, counted over all threads.  Events are created on first use and live with
// the pooled context.
static std::atomic<uint64_t> s_phase_seq{0};
PhaseTimer::PhaseTimer(HostCtx* c, uint64_t bytes) : c_(c), bytes_(bytes), on_(false) {
  if (s_phase_seq.fetch_add(1, std::memory_order_relaxed) % OZCS_STATS_SAMPLE_EVERY != 0) return;
  for (hipEvent_t& ev : c->phase)
    if (!ev && hipEventCreate(&ev) != hipSuccess) {
      ev = nullptr;
      (void)hipGetLastError();
      return;
    }
  on_ = true;
}
void PhaseTimer::mark(int i) {
  if (on_ && hipEventRecord(c_->phase[i], c_->stream) != hipSuccess) {
    (void)hipGetLastError();
    on_ = false;
  }
}
// After the stream has been synchronised.
void PhaseTimer::finish() {
  if (!on_) return;
  float ms[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++)
    if (hipEventElapsedTime(&ms[i], c_->phase[i], c_->phase[i + 1]) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
  s_timed_calls.fetch_add(1, std::memory_order_relaxed);
  s_timed_bytes.fetch_add(bytes_, std::memory_order_relaxed);
  s_h2d_us.fetch_add((uint64_t)(ms[0] * 1000.0f + 0.5f), std::memory_order_relaxed);
  s_kernel_us.fetch_add((uint64_t)(ms[1] * 1000.0f + 0.5f), std::memory_order_relaxed);
  s_d2h_us.fetch_add((uint64_t)(ms[2] * 1000.0f + 0.5f), std::memory_order_relaxed);
}

// Keeps the calling thread's current device unchanged across a call.
DeviceGuard::DeviceGuard(int dev) : prev_(-1), ok_(false) {
  if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
  ok_ = (prev_ == dev) || hipSetDevice(dev) == hipSuccess;
  if (!ok_) (void)hipGetLastError();  // do not leave a sticky error for the caller's own HIP code
}
DeviceGuard::~DeviceGuard() {
  if (prev_ >= 0) (void)hipSetDevice(prev_);
}

const DeviceInfo* find_device(int device) {
  Global& g = global();
  for (const DeviceInfo& d : g.devs)
    if (d.id == device) return &d;
  return nullptr;
}

int ensure_init() {
  Global& g = global();
  if (g.inited.load(std::memory_order_acquire)) return OZCS_OK;
  return init_impl(0);
}

// ---- scratch contexts for the host-pointer calls: pooled per device, grow-only buffers ----

HostCtx* acquire_ctx(int device) {
  Global& g = global();
  {
    std::lock_guard<std::mutex> lk(g.mu);
    for (size_t i = 0; i < g.free_ctx.size(); i++) {
      if (g.free_ctx[i]->device == device) {
        HostCtx* c = g.free_ctx[i];
        g.free_ctx.erase(g.free_ctx.begin() + i);
        return c;
      }
    }
  }
  HostCtx* c = new HostCtx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  return c;
}

void release_ctx(HostCtx* c) {
  Global& g = global();
  std::lock_guard<std::mutex> lk(g.mu);
  g.free_ctx.push_back(c);
}

static void destroy_ctx(HostCtx* c) {
  DeviceGuard dg(c->device);
  if (c->d_data) (void)hipFree(c->d_data);
  if (c->d_sums) (void)hipFree(c->d_sums);
  if (c->d_aux) (void)hipFree(c->d_aux);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_sums) (void)hipHostFree(c->h_sums);
  for (hipEvent_t ev : c->phase)
    if (ev) (void)hipEventDestroy(ev);
  if (c->side_done) (void)hipEventDestroy(c->side_done);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int ctx_reserve(void** buf, size_t* cap, size_t need) {
  if (need <= *cap) return OZCS_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *cap = 0;
  size_t want = need + need / 4;
  want = (want + 4095) & ~(size_t)4095;
  hipError_t e = hipMalloc(buf, want);
  if (e != hipSuccess) {
    e = hipMalloc(buf, need);
    want = need;
  }
  if (e != hipSuccess) return fail(OZCS_ENOMEM, "hipMalloc(%zu): %s", need, hipGetErrorString(e));
  *cap = want;
  return OZCS_OK;
}

// Presets the per-chunk verdict words to "no mismatch" (all ones).  A kernel rather than hipMemsetAsync: under
// stream capture the runtime's memset node left other values in the words (seen on ROCm 7.2), and a kernel
// is what the rest of the call enqueues anyway.
__global__ void preset_verdicts_kernel(unsigned long long* v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = ~0ull;
}
static hipError_t preset_verdicts(void* v, uint64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  preset_verdicts_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(static_cast<unsigned long long*>(v), n);
  return hipGetLastError();
}

// Resolves this device's address of the array.  Caller has made `device` current.
static unsigned long long* resolve_noalloc_verdicts(int device) {
  unsigned long long* ptr = nullptr;
  const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(::scratch));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fail(OZCS_EHIP, "hipGetSymbolAddress(verdict array) on device %d: %s", device, hipGetErrorString(e));
    return nullptr;
  }
  return ptr;
}

// One workgroup: the first chunk whose verdict word is not "none" and that word -> summary[0..1] (-1, -1 if all
// chunks passed).
__global__ __launch_bounds__(1024) void summarize_verdicts_kernel(const unsigned long long* v, uint32_t n,
                                                                  long long* summary) {
  __shared__ unsigned int first;
  if (threadIdx.x == 0) first = 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    if (v[i] != ~0ull) {
      atomicMin(&first, i);
      break;  // this thread's later chunks have larger indices
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    summary[0] = first == 0xFFFFFFFFu ? -1ll : (long long)first;
    summary[1] = first == 0xFFFFFFFFu ? -1ll : (long long)v[first];
  }
}

// Grow-only pinned host buffer.
static int host_reserve(void** buf, size_t* cap, size_t need) {
  if (need <= *cap) return OZCS_OK;
  if (*buf) (void)hipHostFree(*buf);
  *buf = nullptr;
  *cap = 0;
  size_t want = need + need / 4;
  want = (want + 4095) & ~(size_t)4095;
  hipError_t e = hipHostMalloc(buf, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    e = hipHostMalloc(buf, need, hipHostMallocDefault);
    want = need;
  }
  if (e != hipSuccess) return fail(OZCS_ENOMEM, "hipHostMalloc(%zu): %s", need, hipGetErrorString(e));
  *cap = want;
  return OZCS_OK;
}

// ---- one launch over a uniform chunk batch, device pointers ----

int launch_windows(const DeviceInfo& dev, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                   uint64_t n_chunks, uint64_t chunk_stride, int mode, void* d_sums, const void* d_expected,
                   uint64_t expected_stride, uint64_t n_expected, uint64_t start_index,
                   unsigned long long* d_first_mismatch, hipStream_t stream) {
  const uint64_t wpc = ozcs_num_checksums(chunk_len, bpc);
  const uint64_t n_windows = wpc * n_chunks;
  if (n_windows == 0) return OZCS_OK;
  hipError_t e;
  if (type == OZCS_CRC32 || type == OZCS_CRC32C) {
    CrcLaunch p{};
    p.data = static_cast<const uint8_t*>(d_data);
    p.chunk_len = chunk_len;
    p.chunk_stride = chunk_stride;
    p.n_windows = n_windows;
    p.wpc = wpc;
    p.bpc = bpc;
    p.poly_idx = (type == OZCS_CRC32C) ? 1 : 0;
    p.sums = static_cast<uint8_t*>(d_sums);
    p.expected = static_cast<const uint8_t*>(d_expected);
    p.expected_stride = expected_stride;
    p.n_expected = n_expected;
    p.start_index = start_index;
    p.first_mismatch = d_first_mismatch;
    const int group = crc_choose_group(n_windows, bpc, dev.num_cus);
    e = crc_launch(p, mode, group, dev.num_cus, stream);
  } else {
    DigestLaunch p{};
    p.data = static_cast<const uint8_t*>(d_data);
    p.chunk_len = chunk_len;
    p.chunk_stride = chunk_stride;
    p.n_windows = n_windows;
    p.wpc = wpc;
    p.bpc = bpc;
    p.algo = (type == OZCS_SHA256) ? kAlgoSha256 : kAlgoMd5;
    p.sums = static_cast<uint8_t*>(d_sums);
    p.expected = static_cast<const uint8_t*>(d_expected);
    p.expected_stride = expected_stride;
    p.n_expected = n_expected;
    p.start_index = start_index;
    p.first_mismatch = d_first_mismatch;
    e = digest_launch(p, mode, dev.num_cus, stream);
  }
  if (e != hipSuccess) return fail_hip(e, "kernel launch");
  s_launches.fetch_add(1, std::memory_order_relaxed);
  s_bytes.fetch_add(chunk_len * n_chunks, std::memory_order_relaxed);
  s_windows.fetch_add(n_windows, std::memory_order_relaxed);
  return OZCS_OK;
}

static int check_type_bpc(int type, uint32_t bpc) {
  if (type < OZCS_NONE || type > OZCS_MD5) return fail(OZCS_EINVAL, "Unrecognized ChecksumType: %d", type);
  if (type != OZCS_NONE && bpc == 0) return fail(OZCS_EINVAL, "bytesPerChecksum must be positive");
  return OZCS_OK;
}

// ---- gathers: caller memory into pinned staging, range by range, and on to the device ----
constexpr uint64_t kStagePiece = 8ull << 20;            // largest byte range of a gather, and of one copy to the device
constexpr size_t kGatherThreads = 4;                    // the caller and three helpers
constexpr uint64_t kGatherRangeMin = 256ull << 10;      // smallest byte range one gather part copies and sends
constexpr uint64_t kGatherNtMinBytes = 8ull << 20;      // staged data from which the gather writes around the cache
constexpr uint64_t kLonePinnedSumsMax = 1ull << 20;     // a lone call's digests return through pinned memory up to this size

// ---- helper threads of the gathers (host copies of caller memory into pinned staging) ----
// A caller publishes a job of n parts and works through the parts itself; helpers that are awake claim parts beside
// it.  Nobody waits for a helper to wake up: once every part is claimed, the caller only waits for the parts that
// helpers have already taken.  With no helper at all (threads not to be had, a forked child) the caller does
// every part.  Helpers are started on first use and joined by ozcs_shutdown and at process exit.
class GatherPool {
 public:
  static GatherPool& get() {
    static GatherPool pool;
    return pool;
  }
  ~GatherPool() { stop(); }

  void run(size_t n_parts, const std::function<void(size_t)>& fn) {
    if (n_parts == 0) return;
    if (n_parts == 1) return fn(0);
    Job job{&fn, n_parts, 0, 0};
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!started_) {
        started_ = true;
        for (size_t t = 0; t + 1 < kGatherThreads; t++) {
          try {
            helpers_.emplace_back([this] { helper_main(); });
          } catch (...) {
            break;
          }
        }
      }
      jobs_.push_back(&job);
    }
    work_.notify_all();
    for (;;) {
      size_t idx;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (job.next >= job.n) break;
        idx = claim(&job);
      }
      fn(idx);
      std::lock_guard<std::mutex> lk(mu_);
      job.done++;
    }
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return job.done == job.n; });
  }

  void stop() {
    std::vector<std::thread> joining;
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      joining.swap(helpers_);
    }
    work_.notify_all();
    for (std::thread& t : joining) t.join();
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = false;
    started_ = false;
  }

 private:
  struct Job {
    const std::function<void(size_t)>* fn;
    size_t n, next, done;
  };
  // under mu_: the next part of a job; a job with no part left is taken off the list
  size_t claim(Job* job) {
    const size_t idx = job->next++;
    if (job->next == job->n) jobs_.erase(std::find(jobs_.begin(), jobs_.end(), job));
    return idx;
  }
  void helper_main() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      work_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
      if (stop_) return;
      Job* job = jobs_.front();
      const size_t idx = claim(job);
      lk.unlock();
      try {
        (*job->fn)(idx);
      } catch (...) {  // a part reports through what it writes; nothing may escape a helper
      }
      lk.lock();
      if (++job->done == job->n) done_.notify_all();  // the job lives on its caller's stack until then
    }
  }

  std::mutex mu_;
  std::condition_variable work_, done_;
  std::vector<Job*> jobs_;
  std::vector<std::thread> helpers_;
  bool started_ = false, stop_ = false;
};

// Copy of one chunk into pinned staging with non-temporal stores: the staged bytes are read next by the copy engine,
// not by a core, and the C library's memcpy takes ordinary stores below a few MiB -- every destination line is then
// read into the cache before it is overwritten, which halves what a gathering thread moves when chunks are small.
// dst is 16-byte aligned (staging offsets are multiples of 256).  The caller fences before the bytes are handed on.
static void gather_copy(uint8_t* dst, const uint8_t* src, size_t n, bool streaming) {
  if (!streaming || n < 256 || (reinterpret_cast<uintptr_t>(dst) & 15)) {
    memcpy(dst, src, n);
    return;
  }
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32));
    const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
  }
  if (i < n) memcpy(dst + i, src + i, n - i);
}

static bool gather_streaming_enabled() {  // OZCS_GATHER_NT=0: the C library's memcpy, for comparison
  static const bool on = [] {
    const char* v = getenv("OZCS_GATHER_NT");
    return !(v && v[0] == '0');
  }();
  return on;
}

// Size of the byte ranges a staged region of `bytes` is cut into: an eighth of it, between kGatherRangeMin and
// kStagePiece, in whole 4 KiB.
static uint64_t gather_range(uint64_t bytes) {
  uint64_t range = (bytes / 8 + 4095) & ~(uint64_t)4095;
  if (range < kGatherRangeMin) range = kGatherRangeMin;
  if (range > kStagePiece) range = kStagePiece;
  return range;
}

// Caller memory into pinned memory that the copy engine reads next (a pipe slot): from kPooledCopyMin on, in parts
// shared with the pool's helpers and with streaming stores; one thread's memcpy carries 30-37 GB/s, the link 57.
constexpr size_t kPooledCopyMin = 1u << 20, kPooledCopyPart = 512u << 10;
void pooled_copy(void* dst, const void* src, size_t n) {
  if (n < kPooledCopyMin || (reinterpret_cast<uintptr_t>(dst) & 15)) {
    memcpy(dst, src, n);
    return;
  }
  const size_t parts = (n + kPooledCopyPart - 1) / kPooledCopyPart;
  uint8_t* d = static_cast<uint8_t*>(dst);
  const uint8_t* s = static_cast<const uint8_t*>(src);
  const bool streaming = gather_streaming_enabled();
  GatherPool::get().run(parts, [&](size_t r) {
    const size_t lo = r * kPooledCopyPart, hi = lo + kPooledCopyPart < n ? lo + kPooledCopyPart : n;
    gather_copy(d + lo, s + lo, hi - lo, streaming);
    if (streaming) _mm_sfence();
  });
}

// Device for a single host-pointer call.  A caller thread keeps the device it was first given (threads are dealt
// out round-robin): the callers are long-lived writer / reader / scanner threads, and staying put lets each reuse
// the pooled staging buffers of its device instead of growing a set on every device in turn.
static int pick_device() {
  Global& g = global();
  const size_t n = g.devs.size();
  thread_local size_t mine = (size_t)-1;
  if (mine == (size_t)-1) mine = (size_t)g.rr.fetch_add(1, std::memory_order_relaxed);
  return g.devs[mine % n].id;
}

// Host-pointer compute of one chunk on one device, synchronous.
static int compute_host_on(int device, int type, uint32_t bpc, const void* data, uint64_t len, void* sums_out,
                           uint64_t n, size_t ds) {
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail(OZCS_ENODEV, "device %d is not initialised", device);
  DeviceGuard dg(device);
  if (!dg.ok()) return fail(OZCS_ENODEV, "cannot select device %d", device);
  HostCtx* c = acquire_ctx(device);
  if (!c) return fail(OZCS_EHIP, "cannot create a stream on device %d", device);
  // digests come back through pinned memory when they are few: a copy into the caller's pageable buffer goes through the
  // runtime's own staging and a wait inside the call
  const bool pinned_sums = n * ds <= kLonePinnedSumsMax;
  int rc = ctx_reserve(&c->d_data, &c->data_cap, len);
  if (rc == OZCS_OK) rc = ctx_reserve(&c->d_sums, &c->sums_cap, n * ds);
  if (rc == OZCS_OK && pinned_sums) rc = host_reserve(&c->h_sums, &c->h_sums_cap, n * ds);
  hipError_t e = hipSuccess;
  PhaseTimer timer(c, len);
  timer.mark(0);
  if (rc == OZCS_OK) {
    e = hipMemcpyAsync(c->d_data, data, len, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync H2D");
  }
  timer.mark(1);
  if (rc == OZCS_OK)
    rc = launch_windows(*dev, type, bpc, c->d_data, len, 1, len, kModeCompute, c->d_sums, nullptr, 0, 0, 0, nullptr,
                        c->stream);
  timer.mark(2);
  if (rc == OZCS_OK) {
    e = hipMemcpyAsync(pinned_sums ? c->h_sums : sums_out, c->d_sums, n * ds, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync D2H");
  }
  timer.mark(3);
  e = hipStreamSynchronize(c->stream);
  if (rc == OZCS_OK && e != hipSuccess) rc = fail_hip(e, "hipStreamSynchronize");
  if (rc == OZCS_OK) {
    if (pinned_sums) memcpy(sums_out, c->h_sums, n * ds);
    timer.finish();
    s_h2d.fetch_add(len, std::memory_order_relaxed);
    s_d2h.fetch_add(n * ds, std::memory_order_relaxed);
  }
  release_ctx(c);
  return rc;
}

// One device's share [lo, hi) of a chunk list, in groups of at most kBatchStageBytes of data.
//
// A copy call and a launch per chunk cost some 25 us per chunk whatever its size, and a copy from pageable
// memory runs at half the link's rate.  So chunks (of up to kStagedChunkBytes: larger ones arrive here cut into
// pieces, except those of even larger windows, which take a copy of their own) are gathered into one pinned
// staging buffer, range by range, each range copied to the device while the next are gathered (the caller's
// thread and the gather pool's helpers share the ranges).  Launches follow the last copy, as few as the shapes
// allow: CRC chunks of one shape that follow each other share one uniform launch; lone small chunks (at most
// kSmallChunkBytes and kSmallChunkWindows windows: small keys, the tail chunks of files) get a descriptor per
// window (start, length, where the digest goes) and ONE launch per checksum type works through the descriptors;
// large SHA-256 / MD5 chunks go through the job-list kernel so that their few windows per chunk add up to full
// wavefronts.  Digests come back in one copy when they fit the pinned return buffer.
// Per-chunk status and n_sums are filled in.
constexpr uint64_t kBatchStageBytes = 512ull << 20;  // also the most pinned memory one in-flight batch call holds
constexpr uint64_t kStagedChunkBytes = 128ull << 20;  // larger chunks are hashed as pieces of whole windows of at most this
constexpr uint64_t kSmallChunkBytes = 1ull << 20;
constexpr uint64_t kSmallChunkWindows = 64;
constexpr uint64_t kPinnedSumsMax = 32ull << 20;
constexpr uint64_t kSplitCallBytes = 32ull << 20;  // ozcs_compute on at least this much goes through the batched path
constexpr uint64_t kMaxLaunchWindows = 0x7fffffffull;  // window indices of a digest job launch are 32-bit
constexpr uint64_t kSideCopyMinBytes = 32ull << 20;     // staged data from which a second copy stream is worth its event

// Wall-clock trace of one batch group's host phases, for tuning builds only (-DOZCS_TUNE_TRACE): one stderr line per group.
struct BatchTrace {
#ifdef OZCS_TUNE_TRACE
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  std::string line;
  void lap(const char* what) {
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof buf, " %s=%lldus", what, (long long)std::chrono::duration_cast<std::chrono::microseconds>(now - last).count());
    line += buf;
    last = now;
  }
  ~BatchTrace() {
    fprintf(stderr, "[ozcs batch]%s total=%lldus\n", line.c_str(),
            (long long)std::chrono::duration_cast<std::chrono::microseconds>(last - t0).count());
  }
#else
  void lap(const char*) {}
#endif
};

static bool batch_side_copy_enabled() {
  static const bool on = [] {
    const char* v = getenv("OZCS_BATCH_SIDE_COPY");
    return !(v && v[0] == '0');
  }();
  return on;
}

static void batch_on_device(int device, ozcs_chunk* chunks, uint32_t lo, uint32_t hi) {
  auto fail_all = [&](uint32_t a, uint32_t b, int code) {
    for (uint32_t i = a; i < b; i++) { chunks[i].status = code; chunks[i].n_sums = 0; }
  };
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail_all(lo, hi, fail(OZCS_ENODEV, "device %d is not initialised", device));
  DeviceGuard dg(device);
  if (!dg.ok()) return fail_all(lo, hi, fail(OZCS_ENODEV, "cannot select device %d", device));
  HostCtx* c = acquire_ctx(device);
  if (!c) return fail_all(lo, hi, fail(OZCS_EHIP, "cannot create a stream on device %d", device));
  // run: on the first of consecutive CRC chunks of one shape (type, window size, length) that are laid out back to back,
  // the number of chunks one launch covers; 0 on the chunks that follow it and on everything that has no launch of its
  // own (digest chunks, and small chunks, which share a launch through window descriptors unless they form a run)
  struct Item { uint32_t idx; uint64_t data_off, sums_off, n; size_t ds; bool staged, small; uint32_t run; };
  const int kTypes[4] = {OZCS_CRC32, OZCS_CRC32C, OZCS_SHA256, OZCS_MD5};
  uint32_t i = lo;
  while (i < hi) {
    // ---- pick a group that fits the staging budget, validating as we go ----
    BatchTrace trace;
    std::vector<Item> items;
    uint64_t group_bytes = 0;
    uint32_t j = i;
    for (; j < hi; j++) {
      ozcs_chunk& ch = chunks[j];
      ch.n_sums = 0;
      ch.status = check_type_bpc(ch.type, ch.bpc);
      if (ch.status != OZCS_OK || ch.type == OZCS_NONE || ch.len == 0) continue;
      const size_t ds = ozcs_digest_size(ch.type);
      const uint64_t n = ozcs_num_checksums(ch.len, ch.bpc);
      if (!ch.data || !ch.sums) { ch.status = fail(OZCS_EINVAL, "null pointer in chunk %u", j); continue; }
      if (ch.sums_cap < n * ds) { ch.status = fail(OZCS_EINVAL, "sums_cap too small in chunk %u", j); continue; }
      if (n > kMaxLaunchWindows && (ch.type == OZCS_SHA256 || ch.type == OZCS_MD5)) {
        ch.status = fail(OZCS_EINVAL, "chunk %u has more windows than one launch can number", j);
        continue;
      }
      if (!items.empty() && group_bytes + ch.len > kBatchStageBytes) break;
      items.push_back(Item{j, 0, 0, n, ds, ch.len <= kStagedChunkBytes, ch.len <= kSmallChunkBytes && n <= kSmallChunkWindows, 0});
      group_bytes += (ch.len + 255) & ~(uint64_t)255;
    }
    const uint32_t group_end = j;
    if (!items.empty()) {
      // ---- layout: staged chunks first (one contiguous region), then the others; digests in the same order.
      // CRC chunks of one shape that follow each other get their digests back to back, so that one launch covers
      // the whole run (a launch per 4 MiB chunk keeps a fraction of the chip busy for 15-25 us each, after the
      // last copy has landed). ----
      uint64_t data_bytes = 0, sums_bytes = 0, n_desc = 0, n_jobs = 0, staged_bytes = 0;
      std::vector<Item*> order;
      order.reserve(items.size());
      for (int pass = 0; pass < 2; pass++)
        for (Item& it : items)
          if (it.staged == (pass == 0)) order.push_back(&it);
      Item* head = nullptr;
      for (Item* it : order) {  // runs first: whether a small chunk goes through the descriptor list depends on them
        const ozcs_chunk& ch = chunks[it->idx];
        const bool crc = ch.type == OZCS_CRC32 || ch.type == OZCS_CRC32C;
        if (head && crc && chunks[head->idx].type == ch.type && chunks[head->idx].bpc == ch.bpc &&
            chunks[head->idx].len == ch.len) {
          head->run++;
        } else {
          head = crc ? it : nullptr;
          if (head) head->run = 1;
        }
      }
      uint32_t left = 0;  // chunks still to come in the run being laid out
      for (Item* it : order) {
        const ozcs_chunk& ch = chunks[it->idx];
        const bool crc = ch.type == OZCS_CRC32 || ch.type == OZCS_CRC32C;
        it->data_off = data_bytes;
        data_bytes += (ch.len + 255) & ~(uint64_t)255;
        if (it->staged) staged_bytes = data_bytes;
        if (left) {  // inside a run: digests follow the previous chunk's
          left--;
          it->small = false;
        } else {
          if (it->small && it->run == 1) it->run = 0;  // a lone small chunk: one window descriptor per window
          if (it->run > 1) it->small = false;          // equal small chunks: the uniform launch serves them better
          sums_bytes = it->small ? ((sums_bytes + 31) & ~(uint64_t)31) : ((sums_bytes + 255) & ~(uint64_t)255);
          left = it->run ? it->run - 1 : 0;
        }
        if (it->small) n_desc += it->n;
        else if (!crc) n_jobs++;
        it->sums_off = sums_bytes;
        sums_bytes += it->n * it->ds;
      }
      sums_bytes = (sums_bytes + 255) & ~(uint64_t)255;
      // the pinned staging buffer and the device's auxiliary buffer both hold: window descriptors, then digest jobs
      const uint64_t desc_bytes = n_desc * sizeof(WindowDesc), aux_bytes = desc_bytes + n_jobs * sizeof(DigestJob);
      const bool pinned_sums = sums_bytes <= kPinnedSumsMax;
      trace.lap("layout");
      int rc = ctx_reserve(&c->d_data, &c->data_cap, data_bytes);
      if (rc == OZCS_OK) rc = ctx_reserve(&c->d_sums, &c->sums_cap, sums_bytes);
      if (rc == OZCS_OK && aux_bytes) rc = ctx_reserve(&c->d_aux, &c->aux_cap, aux_bytes);
      if (rc == OZCS_OK && staged_bytes + aux_bytes) rc = host_reserve(&c->h_stage, &c->h_stage_cap, staged_bytes + aux_bytes);
      if (rc == OZCS_OK && pinned_sums) rc = host_reserve(&c->h_sums, &c->h_sums_cap, sums_bytes);
      hipError_t e = hipSuccess;
      uint8_t* dd = static_cast<uint8_t*>(c->d_data);
      uint8_t* dsums = static_cast<uint8_t*>(c->d_sums);
      // Phase times of a group: all copies in are queued before the first kernel of the group except the small
      // descriptor and job lists, which are counted with the kernels they precede.
      PhaseTimer timer(c, group_bytes);
      timer.mark(0);
      if (rc == OZCS_OK && staged_bytes) {
        // ---- gather the staged chunks into pinned memory and copy it on, part by part.
        // The staged region is cut into byte ranges (a chunk may straddle ranges); whoever takes a range copies the
        // chunk bytes that fall into it and queues the range's copy to the device, so the link works while the rest
        // is still being gathered.  One thread's copy feeds about half of what the link takes: the ranges are
        // shared with the gather pool's helpers.  Ranges may be queued in any order: the kernels come after all of
        // them. ----
        uint8_t* hs = static_cast<uint8_t*>(c->h_stage);
        std::vector<const Item*> staged;
        for (const Item* it : order)
          if (it->staged) staged.push_back(it);
        hipStream_t side = nullptr;
        if (staged_bytes >= kSideCopyMinBytes && batch_side_copy_enabled()) {
          if (!c->side && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
          if (c->side && !c->side_done && hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming) != hipSuccess)
            c->side_done = nullptr;
          if (c->side && c->side_done) side = c->side;
        }
        const bool nt_mode = staged_bytes >= kGatherNtMinBytes && gather_streaming_enabled();
        const uint64_t range = gather_range(staged_bytes);
        const size_t n_ranges = (size_t)((staged_bytes + range - 1) / range);
        std::atomic<int> first_err{(int)hipSuccess};
        auto gather = [&](size_t r) {
          DeviceGuard tg(device);
          const uint64_t lo_b = r * range, hi_b = lo_b + range < staged_bytes ? lo_b + range : staged_bytes;
          // first staged chunk that reaches into the range
          size_t q = (size_t)(std::upper_bound(staged.begin(), staged.end(), lo_b,
                                               [](uint64_t v, const Item* it) { return v < it->data_off; }) -
                              staged.begin());
          if (q) q--;
          for (; q < staged.size() && staged[q]->data_off < hi_b; q++) {
            const ozcs_chunk& ch = chunks[staged[q]->idx];
            const uint64_t a = staged[q]->data_off > lo_b ? staged[q]->data_off : lo_b;
            const uint64_t b = staged[q]->data_off + ch.len < hi_b ? staged[q]->data_off + ch.len : hi_b;
            if (a < b)
              gather_copy(hs + a, static_cast<const uint8_t*>(ch.data) + (a - staged[q]->data_off), b - a, nt_mode);
          }
          if (nt_mode) _mm_sfence();  // streaming stores are visible to the copy engine only after a fence
          const hipError_t x = hipMemcpyAsync(dd + lo_b, hs + lo_b, hi_b - lo_b, hipMemcpyHostToDevice,
                                              (side && (r & 1)) ? side : c->stream);
          int expect = (int)hipSuccess;
          if (x != hipSuccess) first_err.compare_exchange_strong(expect, (int)x);
        };
        GatherPool::get().run(n_ranges, gather);
        if (first_err.load() != (int)hipSuccess) e = (hipError_t)first_err.load();
        if (side) {  // the kernels follow every copy: the main stream takes the side stream's copies in
          hipError_t x = hipEventRecord(c->side_done, side);
          if (x == hipSuccess) x = hipStreamWaitEvent(c->stream, c->side_done, 0);
          if (x != hipSuccess) {  // not ordered by the event: order by waiting
            (void)hipStreamSynchronize(side);
            if (e == hipSuccess) e = x;
          }
        }
      }
      // the other chunks' bytes: a copy each (they are too large to be worth staging)
      for (const Item& it : items)
        if (!it.staged && rc == OZCS_OK && e == hipSuccess)
          e = hipMemcpyAsync(dd + it.data_off, chunks[it.idx].data, chunks[it.idx].len, hipMemcpyHostToDevice, c->stream);
      timer.mark(1);
      trace.lap("gather");
      if (rc == OZCS_OK && n_desc) {
        // ---- descriptors of the small chunks' windows, grouped by checksum type ----
        WindowDesc* hd = reinterpret_cast<WindowDesc*>(static_cast<uint8_t*>(c->h_stage) + staged_bytes);  // 256-aligned
        WindowDesc* ddsc = static_cast<WindowDesc*>(c->d_aux);
        uint64_t first[5] = {0, 0, 0, 0, 0}, bytes[4] = {0, 0, 0, 0};
        uint64_t k = 0;
        for (int t = 0; t < 4; t++) {
          first[t] = k;
          for (const Item& it : items) {
            const ozcs_chunk& ch = chunks[it.idx];
            if (!it.small || ch.type != kTypes[t]) continue;
            for (uint64_t w = 0; w < it.n; w++) {
              const uint64_t off = w * ch.bpc;
              hd[k].data = dd + it.data_off + off;
              hd[k].out = dsums + it.sums_off + w * it.ds;
              hd[k].len = (uint32_t)(ch.len - off < ch.bpc ? ch.len - off : ch.bpc);
              hd[k].reserved = 0;
              k++;
            }
            bytes[t] += ch.len;
          }
        }
        first[4] = k;
        if (e == hipSuccess)
          e = hipMemcpyAsync(ddsc, hd, desc_bytes, hipMemcpyHostToDevice, c->stream);
        for (int t = 0; t < 4 && e == hipSuccess; t++) {
          const uint64_t nw = first[t + 1] - first[t];
          if (!nw) continue;
          if (t < 2) e = crc_launch_desc(ddsc + first[t], (uint32_t)nw, bytes[t], t, dev->num_cus, c->stream);
          else e = digest_launch_desc(ddsc + first[t], (uint32_t)nw, t == 2 ? kAlgoSha256 : kAlgoMd5, c->stream);
          s_launches.fetch_add(1, std::memory_order_relaxed);
          s_windows.fetch_add(nw, std::memory_order_relaxed);
          s_bytes.fetch_add(bytes[t], std::memory_order_relaxed);
        }
      }
      // ---- the other chunks: digests through the job list, CRC one launch per run of equal chunks ----
      if (rc == OZCS_OK && e == hipSuccess && n_jobs) {
        // one job per chunk, grouped by algorithm; each group is one launch (more only where the windows of a
        // group pass what a launch can number)
        DigestJob* hj = reinterpret_cast<DigestJob*>(static_cast<uint8_t*>(c->h_stage) + staged_bytes + desc_bytes);
        const DigestJob* dj = reinterpret_cast<const DigestJob*>(static_cast<const uint8_t*>(c->d_aux) + desc_bytes);
        struct Run { uint32_t first, n_jobs, n_windows; int algo; };
        std::vector<Run> runs;
        uint32_t k = 0;
        for (int algo = 0; algo < 2; algo++) {
          const int type = algo == kAlgoSha256 ? OZCS_SHA256 : OZCS_MD5;
          Run run{k, 0, 0, algo};
          for (const Item& it : items) {
            const ozcs_chunk& ch = chunks[it.idx];
            if (it.small || ch.type != type) continue;
            if ((uint64_t)run.n_windows + it.n > kMaxLaunchWindows) {
              if (run.n_jobs) runs.push_back(run);
              run = Run{k, 0, 0, algo};
            }
            hj[k] = DigestJob{dd + it.data_off, ch.len, dsums + it.sums_off, ch.bpc, run.n_windows};
            k++;
            run.n_jobs++;
            run.n_windows += (uint32_t)it.n;
            s_bytes.fetch_add(ch.len, std::memory_order_relaxed);
          }
          if (run.n_jobs) runs.push_back(run);
        }
        e = hipMemcpyAsync(static_cast<uint8_t*>(c->d_aux) + desc_bytes, hj, n_jobs * sizeof(DigestJob),
                           hipMemcpyHostToDevice, c->stream);
        for (const Run& run : runs) {
          if (e != hipSuccess) break;
          e = digest_launch_jobs(dj + run.first, run.n_jobs, run.n_windows, run.algo, dev->num_cus, c->stream);
          s_launches.fetch_add(1, std::memory_order_relaxed);
          s_windows.fetch_add(run.n_windows, std::memory_order_relaxed);
        }
      }
      for (const Item* it : order) {
        if (!it->run) continue;
        const ozcs_chunk& ch = chunks[it->idx];
        if (rc == OZCS_OK && e == hipSuccess)
          rc = launch_windows(*dev, ch.type, ch.bpc, dd + it->data_off, ch.len, it->run, (ch.len + 255) & ~(uint64_t)255,
                              kModeCompute, dsums + it->sums_off, nullptr, 0, 0, 0, nullptr, c->stream);
      }
      timer.mark(2);
      // ---- digests back: one copy into pinned memory and a scatter, or a copy per chunk ----
      if (rc == OZCS_OK && e == hipSuccess) {
        if (pinned_sums) {
          e = hipMemcpyAsync(c->h_sums, dsums, sums_bytes, hipMemcpyDeviceToHost, c->stream);
        } else {
          for (const Item& it : items)
            if (e == hipSuccess)
              e = hipMemcpyAsync(chunks[it.idx].sums, dsums + it.sums_off, it.n * it.ds, hipMemcpyDeviceToHost, c->stream);
        }
      }
      timer.mark(3);
      trace.lap("enqueue");
      const hipError_t es = hipStreamSynchronize(c->stream);
      trace.lap("sync");
      if (rc == OZCS_OK && e != hipSuccess) rc = fail_hip(e, "batch enqueue");
      if (rc == OZCS_OK && es != hipSuccess) rc = fail_hip(es, "hipStreamSynchronize");
      if (rc == OZCS_OK) timer.finish();
      for (const Item& it : items) {
        ozcs_chunk& ch = chunks[it.idx];
        ch.status = rc;
        ch.n_sums = rc == OZCS_OK ? it.n : 0;
        if (rc == OZCS_OK) {
          if (pinned_sums) memcpy(ch.sums, static_cast<const uint8_t*>(c->h_sums) + it.sums_off, it.n * it.ds);
          s_h2d.fetch_add(ch.len, std::memory_order_relaxed);
          s_d2h.fetch_add(it.n * it.ds, std::memory_order_relaxed);
        }
      }
      trace.lap("scatter");
    }
    i = group_end;
  }
  release_ctx(c);
}

// Appends one host buffer to a chunk list: as one chunk, or -- from split_from bytes on -- as pieces of whole
// windows of at most kStagedChunkBytes, so that the batched path's pinned, shared gather carries it instead of one
// copy from pageable memory (over 50 GB/s against about 30).  `sums` receives the digests back to back either way.
static void append_pieces(std::vector<ozcs_chunk>& out, int type, uint32_t bpc, const void* data, uint64_t len,
                          void* sums, size_t ds, uint64_t split_from) {
  uint64_t piece = len, wpp = 0;
  if (len >= split_from && len > kStagedChunkBytes && bpc <= kStagedChunkBytes) {
    wpp = kStagedChunkBytes / bpc;
    piece = wpp * bpc;
  }
  for (uint64_t off = 0, w = 0; off < len; off += piece, w += wpp) {
    ozcs_chunk p{};
    p.data = static_cast<const uint8_t*>(data) + off;
    p.len = len - off < piece ? len - off : piece;
    p.sums = static_cast<uint8_t*>(sums) + w * ds;
    p.sums_cap = ozcs_num_checksums(p.len, bpc) * ds;
    p.bpc = bpc;
    p.type = type;
    out.push_back(p);
  }
}

}  // namespace ozcs

using namespace ozcs;

extern "C" {

int ozcs_abi_version(void) { return OZCS_ABI_VERSION; }

const char* ozcs_last_error(void) { return t_err.c_str(); }

size_t ozcs_digest_size(int type) {
  switch (type) {
    case OZCS_CRC32:
    case OZCS_CRC32C: return 4;
    case OZCS_SHA256: return 32;
    case OZCS_MD5: return 16;
    default: return 0;
  }
}

uint64_t ozcs_num_checksums(uint64_t len, uint32_t bpc) {
  if (bpc == 0) return 0;
  return len / bpc + (len % bpc ? 1 : 0);
}

void ozcs_shard_chunks(const ozcs_chunk* chunks, uint32_t n_chunks, uint32_t n_parts, uint32_t* cuts) {
  if (!cuts || n_parts == 0) return;
  uint64_t total = 0;
  for (uint32_t i = 0; chunks && i < n_chunks; i++) total += chunks[i].len;
  cuts[0] = 0;
  uint32_t i = 0;
  uint64_t before = 0;  // bytes of chunks [0, i)
  for (uint32_t d = 1; d < n_parts; d++) {
    // part d starts at the first chunk whose predecessors hold at least d/n_parts of the bytes; a list without
    // bytes (all chunks empty) is split by count
    if (total == 0) {
      i = (uint32_t)((uint64_t)n_chunks * d / n_parts);
    } else {
      const unsigned __int128 target = (unsigned __int128)total * d;
      while (i < n_chunks && (unsigned __int128)before * n_parts < target) before += chunks[i++].len;
    }
    cuts[d] = i;
  }
  cuts[n_parts] = n_chunks;
}

static int init_impl(uint32_t device_mask) {
  Global& g = global();
  std::lock_guard<std::mutex> lk(g.init_mu);
  if (g.inited.load(std::memory_order_acquire)) return OZCS_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(OZCS_ENODEV, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  std::vector<DeviceInfo> devs;
  for (int i = 0; i < n && i < 32; i++) {
    if (device_mask && !(device_mask & (1u << i))) continue;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
    // The kernels are built for gfx950 only and need 135 KiB of LDS per workgroup.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) continue;
    DeviceInfo d;
    d.id = i;
    d.num_cus = prop.multiProcessorCount;
    {
      DeviceGuard dg(i);
      if (!dg.ok()) continue;
      d.noalloc_verdicts = resolve_noalloc_verdicts(i);  // null: only ozcs_verify_device_noalloc is affected
    }
    devs.push_back(d);
  }
  if (devs.empty()) return fail(OZCS_ENODEV, "no gfx950 device among %d HIP device(s)", n);
#ifdef OZCS_TEST_HOOKS
  // Test builds only (libozcs_testhooks.so): OZCS_TEST_REPLICATE_DEVICES = k lists every usable device k times, so
  // that the multi-device code paths (sharding a batch over devices, one host thread per shard) run on a one-GPU
  // machine.  The product library does not read the variable.
  if (const char* rep = getenv("OZCS_TEST_REPLICATE_DEVICES")) {
    const int k = atoi(rep);
    const std::vector<DeviceInfo> once = devs;
    for (int r = 1; r < k && r < 8; r++) devs.insert(devs.end(), once.begin(), once.end());
  }
#endif
  g.devs = devs;
  g.inited.store(true, std::memory_order_release);
  return OZCS_OK;
}

void ozcs_shutdown(void) {
  Global& g = global();
  std::lock_guard<std::mutex> lk(g.init_mu);
  if (!g.inited.load(std::memory_order_acquire)) return;
  std::vector<HostCtx*> ctxs;
  {
    std::lock_guard<std::mutex> lk2(g.mu);
    ctxs.swap(g.free_ctx);
  }
  for (HostCtx* c : ctxs) destroy_ctx(c);
  GatherPool::get().stop();
  g.devs.clear();
  g.inited.store(false, std::memory_order_release);
}

int ozcs_available(void) { return ensure_init() == OZCS_OK ? 1 : 0; }

int ozcs_device_count(void) {
  if (ensure_init() != OZCS_OK) return 0;
  return (int)global().devs.size();
}

void ozcs_get_stats(ozcs_stats* out) {
  if (!out) return;
  out->calls = s_calls.load();
  out->bytes_hashed = s_bytes.load();
  out->windows = s_windows.load();
  out->h2d_bytes = s_h2d.load();
  out->d2h_bytes = s_d2h.load();
  out->kernel_launches = s_launches.load();
  out->errors = s_errors.load();
  out->unavailable = s_unavailable.load();
  out->timed_calls = s_timed_calls.load();
  out->timed_bytes = s_timed_bytes.load();
  out->h2d_us = s_h2d_us.load();
  out->kernel_us = s_kernel_us.load();
  out->d2h_us = s_d2h_us.load();
}

static int compute_device_impl(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                        uint64_t n_chunks, uint64_t chunk_stride, void* d_sums, uint64_t sums_cap,
                        void* hip_stream) {
  int rc = check_type_bpc(type, bpc);
  if (rc != OZCS_OK) return rc;
  if ((rc = ensure_init()) != OZCS_OK) return rc;
  if (type == OZCS_NONE || chunk_len == 0 || n_chunks == 0) return OZCS_OK;
  if (!d_data || !d_sums) return fail(OZCS_EINVAL, "null device pointer");
  if (n_chunks > 1 && chunk_stride < chunk_len) return fail(OZCS_EINVAL, "chunk_stride < chunk_len");
  const uint64_t need = ozcs_num_checksums(chunk_len, bpc) * n_chunks * ozcs_digest_size(type);
  if (sums_cap < need) return fail(OZCS_EINVAL, "sums_cap %llu < %llu", (unsigned long long)sums_cap, (unsigned long long)need);
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail(OZCS_ENODEV, "device %d is not initialised", device);
  DeviceGuard dg(device);
  if (!dg.ok()) return fail(OZCS_ENODEV, "cannot select device %d", device);
  s_calls.fetch_add(1, std::memory_order_relaxed);
  return launch_windows(*dev, type, bpc, d_data, chunk_len, n_chunks, chunk_stride, kModeCompute, d_sums, nullptr, 0,
                        0, 0, nullptr, static_cast<hipStream_t>(hip_stream));
}

static int verify_device_impl(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                       uint64_t n_chunks, uint64_t chunk_stride, const void* d_expected,
                       uint64_t expected_stride, uint64_t n_expected, uint64_t start_index,
                       int64_t* d_first_mismatch, void* hip_stream) {
  int rc = check_type_bpc(type, bpc);
  if (rc != OZCS_OK) return rc;
  if ((rc = ensure_init()) != OZCS_OK) return rc;
  if (!d_first_mismatch) return fail(OZCS_EINVAL, "null d_first_mismatch");
  if (n_chunks == 0) return OZCS_OK;
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail(OZCS_ENODEV, "device %d is not initialised", device);
  DeviceGuard dg(device);
  if (!dg.ok()) return fail(OZCS_ENODEV, "cannot select device %d", device);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  hipError_t e = preset_verdicts(d_first_mismatch, n_chunks, stream);  // -1 = no mismatch
  if (e != hipSuccess) return fail_hip(e, "verdict preset");
  if (type == OZCS_NONE || chunk_len == 0) return OZCS_OK;  // Checksum.java:241-244
  if (!d_data || !d_expected) return fail(OZCS_EINVAL, "null device pointer");
  if (n_chunks > 1 && chunk_stride < chunk_len) return fail(OZCS_EINVAL, "chunk_stride < chunk_len");
  s_calls.fetch_add(1, std::memory_order_relaxed);
  return launch_windows(*dev, type, bpc, d_data, chunk_len, n_chunks, chunk_stride, kModeVerify, nullptr, d_expected,
                        expected_stride, n_expected, start_index,
                        reinterpret_cast<unsigned long long*>(d_first_mismatch), stream);
}

static int verify_device_noalloc_impl(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                               uint64_t n_chunks, uint64_t chunk_stride, const void* d_expected,
                               uint64_t expected_stride, uint64_t n_expected, uint64_t start_index,
                               int64_t* d_summary, void* hip_stream) {
  int rc = check_type_bpc(type, bpc);
  if (rc != OZCS_OK) return rc;
  if ((rc = ensure_init()) != OZCS_OK) return rc;
  if (!d_summary) return fail(OZCS_EINVAL, "null d_summary");
  if (n_chunks > OZCS_NOALLOC_MAX_CHUNKS)
    return fail(OZCS_EINVAL, "n_chunks %llu exceeds OZCS_NOALLOC_MAX_CHUNKS (%d): split the batch or use "
                "ozcs_verify_device", (unsigned long long)n_chunks, OZCS_NOALLOC_MAX_CHUNKS);
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail(OZCS_ENODEV, "device %d is not initialised", device);
  if (!dev->noalloc_verdicts) return fail(OZCS_EHIP, "the verdict array of device %d could not be resolved at init", device);
  DeviceGuard dg(device);
  if (!dg.ok()) return fail(OZCS_ENODEV, "cannot select device %d", device);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  unsigned long long* verdicts = dev->noalloc_verdicts;  // the array keeps the previous call's words: preset what is used
  const bool hashes = type != OZCS_NONE && chunk_len != 0 && n_chunks != 0;  // NONE: Checksum.java:241-244
  if (hashes) {
    if (!d_data || !d_expected) return fail(OZCS_EINVAL, "null device pointer");
    if (n_chunks > 1 && chunk_stride < chunk_len) return fail(OZCS_EINVAL, "chunk_stride < chunk_len");
    hipError_t e = preset_verdicts(verdicts, n_chunks, stream);
    if (e != hipSuccess) return fail_hip(e, "verdict preset");
    s_calls.fetch_add(1, std::memory_order_relaxed);
    rc = launch_windows(*dev, type, bpc, d_data, chunk_len, n_chunks, chunk_stride, kModeVerify, nullptr, d_expected,
                        expected_stride, n_expected, start_index, verdicts, stream);
    if (rc != OZCS_OK) return rc;
  }
  summarize_verdicts_kernel<<<dim3(1), dim3(1024), 0, stream>>>(verdicts, hashes ? (uint32_t)n_chunks : 0u,
                                                                reinterpret_cast<long long*>(d_summary));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "verdict summary");
  return OZCS_OK;
}

static int compute_impl(int type, uint32_t bpc, const void* data, uint64_t len, void* sums_out, uint64_t sums_cap,
                 uint64_t* n_sums) try {
  if (n_sums) *n_sums = 0;
  int rc = check_type_bpc(type, bpc);
  if (rc != OZCS_OK) return rc;
  if (type == OZCS_NONE || len == 0) return OZCS_OK;  // Checksum.java:157-160; empty buffer -> no windows
  if (!data || !sums_out) return fail(OZCS_EINVAL, "null pointer");
  const size_t ds = ozcs_digest_size(type);
  const uint64_t n = ozcs_num_checksums(len, bpc);
  if (sums_cap < n * ds) return fail(OZCS_EINVAL, "sums_cap %llu < %llu", (unsigned long long)sums_cap, (unsigned long long)(n * ds));
  if ((rc = ensure_init()) != OZCS_OK) return rc;
  s_calls.fetch_add(1, std::memory_order_relaxed);
  if (len >= kSplitCallBytes && bpc <= kStagedChunkBytes) {
    std::vector<ozcs_chunk> parts;
    append_pieces(parts, type, bpc, data, len, sums_out, ds, kSplitCallBytes);
    batch_on_device(pick_device(), parts.data(), 0, (uint32_t)parts.size());
    for (const ozcs_chunk& p : parts)
      if (p.status != OZCS_OK) return p.status;
    if (n_sums) *n_sums = n;
    return OZCS_OK;
  }
  rc = compute_host_on(pick_device(), type, bpc, data, len, sums_out, n, ds);
  if (rc == OZCS_OK && n_sums) *n_sums = n;
  return rc;
} catch (const std::bad_alloc&) {
  return fail(OZCS_ENOMEM, "out of host memory in %s", "ozcs_compute");
} catch (...) {
  return fail(OZCS_EHIP, "unexpected failure in %s", "ozcs_compute");
}

static int verify_impl(int type, uint32_t bpc, const void* data, uint64_t len, const void* expected,
                uint64_t n_expected, uint64_t start_index, int64_t* first_mismatch) try {
  if (first_mismatch) *first_mismatch = -1;
  int rc = check_type_bpc(type, bpc);
  if (rc != OZCS_OK) return rc;
  if (type == OZCS_NONE) return OZCS_OK;  // Checksum.java:241-244
  const size_t ds = ozcs_digest_size(type);
  const uint64_t n = ozcs_num_checksums(len, bpc);
  if (n_expected == 0) return fail(OZCS_EEMPTY, "Original checksumData has no checksums");  // ChecksumData.java:121-124
  if (n == 0) return fail(OZCS_EEMPTY, "Computed checksumData has no checksums");          // ChecksumData.java:126-129
  if (!data || !expected) return fail(OZCS_EINVAL, "null pointer");
  if ((rc = ensure_init()) != OZCS_OK) return rc;
  if (len >= kSplitCallBytes && bpc <= kStagedChunkBytes) {  // large buffer: the batched path moves it faster
    ozcs_vchunk vc{};
    vc.data = data;
    vc.len = len;
    vc.expected = expected;
    vc.n_expected = n_expected;
    vc.start_index = start_index;
    vc.bpc = bpc;
    vc.type = type;
    (void)verify_batch_impl(&vc, 1);
    if (vc.status == OZCS_OK && first_mismatch) *first_mismatch = vc.first_mismatch;
    return vc.status;
  }
  const int device = pick_device();
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail(OZCS_ENODEV, "device %d is not initialised", device);
  DeviceGuard dg(device);
  if (!dg.ok()) return fail(OZCS_ENODEV, "cannot select device %d", device);
  HostCtx* c = acquire_ctx(device);
  if (!c) return fail(OZCS_EHIP, "cannot create a stream on device %d", device);
  s_calls.fetch_add(1, std::memory_order_relaxed);
  // only the stored digests the computed list can reach are shipped
  const uint64_t first = start_index < n_expected ? start_index : n_expected;
  const uint64_t avail = n_expected - first;
  const uint64_t ship = avail < n ? avail : n;
  rc = ctx_reserve(&c->d_data, &c->data_cap, len);
  if (rc == OZCS_OK) rc = ctx_reserve(&c->d_sums, &c->sums_cap, (ship ? ship : 1) * ds);
  if (rc == OZCS_OK) rc = ctx_reserve(&c->d_aux, &c->aux_cap, sizeof(int64_t));
  if (rc == OZCS_OK) rc = host_reserve(&c->h_sums, &c->h_sums_cap, sizeof(int64_t));
  hipError_t e = hipSuccess;
  int64_t fm = -1;
  PhaseTimer timer(c, len);
  timer.mark(0);
  if (rc == OZCS_OK) {
    e = hipMemcpyAsync(c->d_data, data, len, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && ship)
      e = hipMemcpyAsync(c->d_sums, static_cast<const uint8_t*>(expected) + first * ds, ship * ds,
                         hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = preset_verdicts(c->d_aux, 1, c->stream);
    if (e != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync H2D");
  }
  timer.mark(1);
  if (rc == OZCS_OK)
    rc = launch_windows(*dev, type, bpc, c->d_data, len, 1, len, kModeVerify, nullptr, c->d_sums, 0, ship, 0,
                        static_cast<unsigned long long*>(c->d_aux), c->stream);
  timer.mark(2);
  if (rc == OZCS_OK) {  // the verdict word returns through pinned memory, as a lone compute call's digests do
    e = hipMemcpyAsync(c->h_sums, c->d_aux, sizeof fm, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync D2H");
  }
  timer.mark(3);
  e = hipStreamSynchronize(c->stream);
  if (rc == OZCS_OK && e != hipSuccess) rc = fail_hip(e, "hipStreamSynchronize");
  if (rc == OZCS_OK) {
    memcpy(&fm, c->h_sums, sizeof fm);
    timer.finish();
  }
  release_ctx(c);
  if (rc != OZCS_OK) return rc;
  s_h2d.fetch_add(len + ship * ds, std::memory_order_relaxed);
  s_d2h.fetch_add(sizeof fm, std::memory_order_relaxed);
  if (fm >= 0 && (uint64_t)fm >= ship) {
    // every reachable stored digest matched; the computed list runs past the stored one
    return fail(OZCS_ERANGE, "Computed checksum has %llu number of checksums. Original checksum has %llu number of "
                "checksums starting from index %llu", (unsigned long long)n, (unsigned long long)avail,
                (unsigned long long)start_index);  // ChecksumData.java:142-147
  }
  if (first_mismatch) *first_mismatch = fm;
  return OZCS_OK;
} catch (const std::bad_alloc&) {
  return fail(OZCS_ENOMEM, "out of host memory in %s", "ozcs_verify");
} catch (...) {
  return fail(OZCS_EHIP, "unexpected failure in %s", "ozcs_verify");
}

static int compute_batch_impl(ozcs_chunk* chunks, uint32_t n_chunks);

// A chunk list with chunks too large to be staged whole (a copy from pageable memory carries 30 GB/s, the staged path
// over 50): those are hashed as pieces of whole windows, each piece's digests going straight to its place in the
// chunk's digest list, and the pieces' outcomes are folded back into the chunk.  Chunks the batched path will reject
// (null pointers, a digest list too short) are handed on whole, so that it is the one place that judges them.
static bool splittable(const ozcs_chunk& ch) {
  if (ch.len <= kStagedChunkBytes || ch.bpc == 0 || ch.bpc > kStagedChunkBytes || !ch.data || !ch.sums) return false;
  const size_t ds = ozcs_digest_size(ch.type);
  return ds != 0 && ch.sums_cap >= ozcs_num_checksums(ch.len, ch.bpc) * ds;
}

static int compute_batch_split(ozcs_chunk* chunks, uint32_t n_chunks) {
  std::vector<ozcs_chunk> parts;
  std::vector<size_t> first(n_chunks + 1);
  for (uint32_t i = 0; i < n_chunks; i++) {
    first[i] = parts.size();
    if (splittable(chunks[i]))
      append_pieces(parts, chunks[i].type, chunks[i].bpc, chunks[i].data, chunks[i].len, chunks[i].sums,
                    ozcs_digest_size(chunks[i].type), 0);
    else
      parts.push_back(chunks[i]);
  }
  first[n_chunks] = parts.size();
  if (parts.size() > 0xffffffffull) return fail(OZCS_EINVAL, "chunk list too long once large chunks are cut into pieces");
  const int inner = compute_batch_impl(parts.data(), (uint32_t)parts.size());
  int rc = OZCS_OK;
  for (uint32_t i = 0; i < n_chunks; i++) {
    chunks[i].status = OZCS_OK;
    chunks[i].n_sums = 0;
    for (size_t k = first[i]; k < first[i + 1]; k++) {
      if (parts[k].status != OZCS_OK) {
        chunks[i].status = parts[k].status;
        chunks[i].n_sums = 0;
        break;
      }
      chunks[i].n_sums += parts[k].n_sums;
    }
    if (chunks[i].status != OZCS_OK && rc == OZCS_OK) rc = chunks[i].status;
  }
  if (inner != OZCS_OK && rc == OZCS_OK) {  // the call failed as a whole before any piece was judged
    for (uint32_t i = 0; i < n_chunks; i++) { chunks[i].status = inner; chunks[i].n_sums = 0; }
    rc = inner;
  }
  return rc;
}

static int compute_batch_impl(ozcs_chunk* chunks, uint32_t n_chunks) try {
  if (n_chunks == 0) return OZCS_OK;
  if (!chunks) return fail(OZCS_EINVAL, "null chunk list");
  int rc = ensure_init();
  if (rc != OZCS_OK) {
    for (uint32_t i = 0; i < n_chunks; i++) { chunks[i].status = rc; chunks[i].n_sums = 0; }
    return rc;
  }
  for (uint32_t i = 0; i < n_chunks; i++)
    if (splittable(chunks[i])) return compute_batch_split(chunks, n_chunks);
  Global& g = global();
  const size_t ndev = g.devs.size();
  s_calls.fetch_add(1, std::memory_order_relaxed);
  if (ndev == 1 || n_chunks == 1) {
    batch_on_device(g.devs[0].id, chunks, 0, n_chunks);
  } else {
    // contiguous shards of the chunk list with about equal bytes, one host thread per device, no inter-device
    // traffic.  A worker keeps its failure text (the error text is per thread) and lets no exception escape.
    std::vector<uint32_t> cuts(ndev + 1);
    ozcs_shard_chunks(chunks, n_chunks, (uint32_t)ndev, cuts.data());
    std::vector<std::string> msgs(ndev);
    auto work = [&](size_t d) {
      const uint32_t lo = cuts[d], hi = cuts[d + 1];
      try {
        t_err.clear();
        batch_on_device(g.devs[d].id, chunks, lo, hi);
        msgs[d] = t_err;
      } catch (...) {  // std::bad_alloc from the staging lists
        for (uint32_t i = lo; i < hi; i++) { chunks[i].status = OZCS_ENOMEM; chunks[i].n_sums = 0; }
        msgs[d] = "out of host memory in ozcs_compute_batch";
      }
    };
    std::vector<std::thread> th;
    std::vector<size_t> inline_shards;
    for (size_t d = 1; d < ndev; d++) {
      if (cuts[d] >= cuts[d + 1]) continue;
      try {
        th.emplace_back(work, d);
      } catch (...) {  // no thread to be had: the caller's thread serves that device's share
        inline_shards.push_back(d);
      }
    }
    const std::string callers = t_err;
    if (cuts[0] < cuts[1]) work(0);
    for (size_t d : inline_shards) work(d);
    for (auto& t : th) t.join();
    t_err = callers;
    for (size_t d = 0; d < ndev; d++) {
      bool failed = false;
      for (uint32_t i = cuts[d]; i < cuts[d + 1] && !failed; i++) failed = chunks[i].status != OZCS_OK;
      if (failed) { t_err = msgs[d]; break; }
    }
  }
  for (uint32_t i = 0; i < n_chunks; i++)
    if (chunks[i].status != OZCS_OK) return chunks[i].status;
  return OZCS_OK;
} catch (const std::bad_alloc&) {
  return fail(OZCS_ENOMEM, "out of host memory in %s", "ozcs_compute_batch");
} catch (...) {
  return fail(OZCS_EHIP, "unexpected failure in %s", "ozcs_compute_batch");
}

static int verify_batch_impl(ozcs_vchunk* v, uint32_t n_chunks) try {
  if (n_chunks == 0) return OZCS_OK;
  if (!v) return fail(OZCS_EINVAL, "null chunk list");
  // Hash everything that needs hashing through the batched compute path into scratch digests, then compare
  // on the host: digests are a few bytes per window, the hashing is the work.
  std::vector<ozcs_chunk> jobs;
  struct Owner { uint32_t idx; size_t first_job, end_job; uint64_t n; };
  std::vector<Owner> owners;
  std::vector<std::vector<uint8_t>> sums;
  for (uint32_t i = 0; i < n_chunks; i++) {
    ozcs_vchunk& c = v[i];
    c.first_mismatch = -1;
    c.status = check_type_bpc(c.type, c.bpc);
    if (c.status != OZCS_OK || c.type == OZCS_NONE) continue;  // NONE: Checksum.java:241-244
    const uint64_t n = ozcs_num_checksums(c.len, c.bpc);
    if (c.n_expected == 0) { c.status = fail(OZCS_EEMPTY, "Original checksumData has no checksums"); continue; }
    if (n == 0) { c.status = fail(OZCS_EEMPTY, "Computed checksumData has no checksums"); continue; }
    if (!c.data || !c.expected) { c.status = fail(OZCS_EINVAL, "null pointer in chunk %u", i); continue; }
    const size_t ds = ozcs_digest_size(c.type);
    sums.emplace_back((size_t)(n * ds));
    const size_t first_job = jobs.size();
    append_pieces(jobs, c.type, c.bpc, c.data, c.len, sums.back().data(), ds, 0);
    owners.push_back(Owner{i, first_job, jobs.size(), n});
  }
  if (!jobs.empty()) (void)compute_batch_impl(jobs.data(), (uint32_t)jobs.size());
  for (size_t k = 0; k < owners.size(); k++) {
    ozcs_vchunk& c = v[owners[k].idx];
    for (size_t j = owners[k].first_job; j < owners[k].end_job && c.status == OZCS_OK; j++) c.status = jobs[j].status;
    if (c.status != OZCS_OK) continue;
    const size_t ds = ozcs_digest_size(c.type);
    const uint8_t* ex = static_cast<const uint8_t*>(c.expected);
    const uint8_t* got = sums[k].data();
    for (uint64_t i = 0; i < owners[k].n; i++) {  // ChecksumData.java:135-141
      if (c.start_index + i >= c.n_expected) {
        c.status = fail(OZCS_ERANGE, "Computed checksum has %llu number of checksums. Original checksum has %llu number "
                        "of checksums starting from index %llu", (unsigned long long)owners[k].n,
                        (unsigned long long)(c.n_expected > c.start_index ? c.n_expected - c.start_index : 0),
                        (unsigned long long)c.start_index);
        break;
      }
      if (memcmp(ex + (c.start_index + i) * ds, got + i * ds, ds) != 0) {
        c.first_mismatch = (int64_t)i;
        break;
      }
    }
  }
  for (uint32_t i = 0; i < n_chunks; i++)
    if (v[i].status != OZCS_OK) return v[i].status;
  return OZCS_OK;
} catch (const std::bad_alloc&) {
  return fail(OZCS_ENOMEM, "out of host memory in %s", "ozcs_verify_batch");
} catch (...) {
  return fail(OZCS_EHIP, "unexpected failure in %s", "ozcs_verify_batch");
}

// ---- the exported names: each counts its outcome (ozcs_get_stats) and hands it on ----
int ozcs_init(uint32_t device_mask) { return tally(init_impl(device_mask)); }
int ozcs_compute_device(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len, uint64_t n_chunks,
                        uint64_t chunk_stride, void* d_sums, uint64_t sums_cap, void* hip_stream) {
  return tally(compute_device_impl(device, type, bpc, d_data, chunk_len, n_chunks, chunk_stride, d_sums, sums_cap, hip_stream));
}
int ozcs_verify_device(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len, uint64_t n_chunks,
                       uint64_t chunk_stride, const void* d_expected, uint64_t expected_stride, uint64_t n_expected,
                       uint64_t start_index, int64_t* d_first_mismatch, void* hip_stream) {
  return tally(verify_device_impl(device, type, bpc, d_data, chunk_len, n_chunks, chunk_stride, d_expected, expected_stride,
                                  n_expected, start_index, d_first_mismatch, hip_stream));
}
int ozcs_verify_device_noalloc(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                               uint64_t n_chunks, uint64_t chunk_stride, const void* d_expected, uint64_t expected_stride,
                               uint64_t n_expected, uint64_t start_index, int64_t* d_summary, void* hip_stream) {
  return tally(verify_device_noalloc_impl(device, type, bpc, d_data, chunk_len, n_chunks, chunk_stride, d_expected,
                                          expected_stride, n_expected, start_index, d_summary, hip_stream));
}
int ozcs_compute(int type, uint32_t bpc, const void* data, uint64_t len, void* sums_out, uint64_t sums_cap, uint64_t* n_sums) {
  return tally(compute_impl(type, bpc, data, len, sums_out, sums_cap, n_sums));
}
int ozcs_verify(int type, uint32_t bpc, const void* data, uint64_t len, const void* expected, uint64_t n_expected,
                uint64_t start_index, int64_t* first_mismatch) {
  return tally(verify_impl(type, bpc, data, len, expected, n_expected, start_index, first_mismatch));
}
int ozcs_compute_batch(ozcs_chunk* chunks, uint32_t n_chunks) { return tally(compute_batch_impl(chunks, n_chunks)); }
int ozcs_verify_batch(ozcs_vchunk* chunks, uint32_t n_chunks) { return tally(verify_batch_impl(chunks, n_chunks)); }

}  // extern "C"
