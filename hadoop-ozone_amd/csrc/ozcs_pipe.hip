// Asynchronous pinned pipeline (ozcs_pipe_*): pinned chunk slots, three streams chained by events so
// that the host-to-device copy of one chunk, the kernel of the previous one and the digest copy-back
// of the one before overlap.  See include/ozone_checksum.h for the contract.
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/ozone_checksum.h"
#include "crc_kernels.h"
#include "digest_kernels.h"
#include "ozcs_internal.h"

using namespace ozcs;

namespace {

enum SlotState { kFree = 0, kAcquired = 1, kInFlight = 2 };

struct Slot {
  void* h_data = nullptr;   // pinned, slot_bytes
  void* d_data = nullptr;   // device, slot_bytes
  void* h_sums = nullptr;   // pinned, sums_bytes
  void* d_sums = nullptr;   // device, sums_bytes
  hipEvent_t copied = nullptr, hashed = nullptr, done = nullptr;
  int state = kFree;
  bool deferred = false;   // SHA-256 / MD5 slot whose kernel waits for a batch (see flush_deferred)
  int type = 0;
  uint32_t bpc = 0;
  uint64_t len = 0;
  void* user_sums = nullptr;
  uint64_t n_sums = 0, sums_len = 0;
  int status = OZCS_OK;
};

}  // namespace

struct ozcs_pipe {
  int device = 0;
  DeviceInfo dev;
  uint64_t slot_bytes = 0, sums_bytes = 0;
  hipStream_t s_h2d = nullptr, s_kernel = nullptr, s_d2h = nullptr;
  hipStream_t s_digest = nullptr, s_digest_d2h = nullptr;  // SHA-256 / MD5 batches: milliseconds each, kept off the CRC streams
  hipEvent_t digest_done = nullptr;                        // end of the digest batch in flight
  bool digest_running = false;
  std::vector<Slot> slots;
  std::vector<int> deferred;  // tickets of digest slots copied to the device but not yet hashed
  std::mutex mu;
};

namespace {

// digests of a slot never exceed this: one 32-byte digest per 16 bytes would be absurd, so the pinned
// digest buffer is sized for windows of at least 512 bytes and submit rejects what does not fit
uint64_t sums_capacity(uint64_t slot_bytes) { return (slot_bytes / 512 + 1) * 32; }

void free_slot(Slot& s) {
  if (s.h_data) (void)hipHostFree(s.h_data);
  if (s.d_data) (void)hipFree(s.d_data);
  if (s.h_sums) (void)hipHostFree(s.h_sums);
  if (s.d_sums) (void)hipFree(s.d_sums);
  if (s.copied) (void)hipEventDestroy(s.copied);
  if (s.hashed) (void)hipEventDestroy(s.hashed);
  if (s.done) (void)hipEventDestroy(s.done);
  s = Slot();
}

// hipEventQuery that does not leave hipErrorNotReady behind as the thread's last error (the caller's own
// HIP code would otherwise find it there).
hipError_t query_event(hipEvent_t ev) {
  const hipError_t e = hipEventQuery(ev);
  if (e == hipErrorNotReady) (void)hipGetLastError();
  return e;
}

// Completes a slot whose `done` event has fired: hands the digests over and frees the slot.
void retire(Slot& s) {
  if (s.status == OZCS_OK && s.sums_len) memcpy(s.user_sums, s.h_sums, s.sums_len);
  s.state = kFree;
}

// SHA-256 and MD5 run one window per lane, so the four windows of one chunk would occupy four lanes of
// one wavefront for milliseconds.  Digest slots are therefore parked after their host-to-device copy and
// hashed together, one launch per algorithm over up to kMaxDigestJobs chunks, when enough have gathered
// or when somebody needs a result or a slot.  Caller holds p->mu and has selected the device.
int flush_deferred(ozcs_pipe* p) {
  if (p->deferred.empty()) return OZCS_OK;
  int rc = OZCS_OK;
  for (int algo = 0; algo < 2 && rc == OZCS_OK; algo++) {
    const int type = algo == kAlgoSha256 ? OZCS_SHA256 : OZCS_MD5;
    size_t i = 0;
    while (i < p->deferred.size() && rc == OZCS_OK) {
      DigestJobList jl{};
      jl.algo = algo;
      int batch[kMaxDigestJobs];
      int nbatch = 0;
      for (; i < p->deferred.size() && jl.n_jobs < (uint32_t)kMaxDigestJobs; i++) {
        Slot& s = p->slots[p->deferred[i]];
        if (s.type != type) continue;
        DigestJob& j = jl.jobs[jl.n_jobs++];
        j.data = static_cast<const uint8_t*>(s.d_data);
        j.len = s.len;
        j.sums = static_cast<uint8_t*>(s.d_sums);
        j.bpc = s.bpc;
        j.first_window = jl.n_windows;
        jl.n_windows += (uint32_t)s.n_sums;
        batch[nbatch++] = p->deferred[i];
      }
      if (nbatch == 0) break;
      hipError_t e = hipSuccess;
      for (int bi = 0; bi < nbatch; bi++)
        if (e == hipSuccess) e = hipStreamWaitEvent(p->s_digest, p->slots[batch[bi]].copied, 0);
      if (e == hipSuccess) e = digest_launch_jobs(jl, p->dev.num_cus, p->s_digest);
      Slot& lead = p->slots[batch[0]];
      if (e == hipSuccess) e = hipEventRecord(lead.hashed, p->s_digest);
      if (e == hipSuccess) e = hipStreamWaitEvent(p->s_digest_d2h, lead.hashed, 0);
      for (int bi = 0; bi < nbatch; bi++) {
        Slot& s = p->slots[batch[bi]];
        if (e == hipSuccess) e = hipMemcpyAsync(s.h_sums, s.d_sums, s.sums_len, hipMemcpyDeviceToHost, p->s_digest_d2h);
        if (e == hipSuccess) e = hipEventRecord(s.done, p->s_digest_d2h);
        s.deferred = false;
      }
      if (e == hipSuccess) e = hipEventRecord(p->digest_done, p->s_digest_d2h);
      if (e == hipSuccess) p->digest_running = true;
      if (e != hipSuccess) {
        rc = fail_hip(e, "pipe digest batch");
        (void)hipStreamSynchronize(p->s_digest);
        (void)hipStreamSynchronize(p->s_digest_d2h);
        for (int bi = 0; bi < nbatch; bi++) {
          p->slots[batch[bi]].status = rc;
          (void)hipEventRecord(p->slots[batch[bi]].done, p->s_digest_d2h);
        }
      } else {
        s_launches.fetch_add(1, std::memory_order_relaxed);
        s_windows.fetch_add(jl.n_windows, std::memory_order_relaxed);
      }
    }
  }
  p->deferred.clear();
  return rc;
}

// Launches the parked digest slots when the previous digest batch has finished (so exactly one batch runs
// while the next gathers) or when `limit` slots have gathered.  Caller holds p->mu.
int maybe_flush(ozcs_pipe* p, size_t limit) {
  if (p->deferred.empty()) return OZCS_OK;
  if (p->digest_running && query_event(p->digest_done) == hipSuccess) p->digest_running = false;
  if (!p->digest_running || p->deferred.size() >= limit) return flush_deferred(p);
  return OZCS_OK;
}

}  // namespace

extern "C" {

uint64_t ozcs_pipe_slot_bytes(const ozcs_pipe* pipe) { return pipe ? pipe->slot_bytes : 0; }

int ozcs_pipe_create(int device, uint32_t n_slots, uint64_t slot_bytes, ozcs_pipe** out) {
  if (!out) return fail(OZCS_EINVAL, "null out");
  *out = nullptr;
  if (n_slots == 0 || n_slots > 4096 || slot_bytes == 0) return fail(OZCS_EINVAL, "bad pipe geometry");
  int rc = ensure_init();
  if (rc != OZCS_OK) return rc;
  const DeviceInfo* dev = find_device(device);
  if (!dev) return fail(OZCS_ENODEV, "device %d is not initialised", device);
  DeviceGuard dg(device);
  if (!dg.ok()) return fail(OZCS_ENODEV, "cannot select device %d", device);
  ozcs_pipe* p = new (std::nothrow) ozcs_pipe();
  if (!p) return fail(OZCS_ENOMEM, "out of host memory");
  p->device = device;
  p->dev = *dev;
  p->slot_bytes = slot_bytes;
  p->sums_bytes = sums_capacity(slot_bytes);
  try {
    p->slots.resize(n_slots);
    p->deferred.reserve(n_slots);  // so that parking a digest slot later never allocates
  } catch (...) {
    delete p;
    return fail(OZCS_ENOMEM, "out of host memory");
  }
  hipError_t e = hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_kernel, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_digest, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_digest_d2h, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->digest_done, hipEventDisableTiming);
  for (Slot& s : p->slots) {
    if (e == hipSuccess) e = hipHostMalloc(&s.h_data, slot_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&s.d_data, slot_bytes);
    if (e == hipSuccess) e = hipHostMalloc(&s.h_sums, p->sums_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&s.d_sums, p->sums_bytes);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.hashed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    const int code = (e == hipErrorOutOfMemory) ? OZCS_ENOMEM : OZCS_EHIP;
    fail(code, "pipe allocation: %s", hipGetErrorString(e));
    ozcs_pipe_destroy(p);
    return code;
  }
  *out = p;
  return OZCS_OK;
}

void ozcs_pipe_destroy(ozcs_pipe* p) {
  if (!p) return;
  DeviceGuard dg(p->device);
  if (p->s_h2d) (void)hipStreamSynchronize(p->s_h2d);
  if (p->s_kernel) (void)hipStreamSynchronize(p->s_kernel);
  if (p->s_d2h) (void)hipStreamSynchronize(p->s_d2h);
  if (p->s_digest) (void)hipStreamSynchronize(p->s_digest);
  if (p->s_digest_d2h) (void)hipStreamSynchronize(p->s_digest_d2h);
  for (Slot& s : p->slots) free_slot(s);
  if (p->s_digest) (void)hipStreamDestroy(p->s_digest);
  if (p->s_digest_d2h) (void)hipStreamDestroy(p->s_digest_d2h);
  if (p->digest_done) (void)hipEventDestroy(p->digest_done);
  if (p->s_h2d) (void)hipStreamDestroy(p->s_h2d);
  if (p->s_kernel) (void)hipStreamDestroy(p->s_kernel);
  if (p->s_d2h) (void)hipStreamDestroy(p->s_d2h);
  delete p;
}

int ozcs_pipe_acquire(ozcs_pipe* p, void** buf, int* ticket) {
  if (!p || !buf || !ticket) return fail(OZCS_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  DeviceGuard dg(p->device);
  // A slot stays its ticket holder's until that ticket has been polled to completion, waited for or
  // drained: a finished slot is never recycled behind its owner's back.
  for (size_t i = 0; i < p->slots.size(); i++) {
    Slot& s = p->slots[i];
    if (s.state == kFree) {
      s.state = kAcquired;
      *buf = s.h_data;
      *ticket = (int)i;
      return OZCS_OK;
    }
  }
  // parked digest slots finish only once launched
  (void)maybe_flush(p, p->slots.size() / 2 ? p->slots.size() / 2 : 1);
  return fail(OZCS_EBUSY, "all %zu slots are in flight", p->slots.size());
}

int ozcs_pipe_submit_slot(ozcs_pipe* p, int ticket, int type, uint32_t bpc, uint64_t len, void* sums_out,
                          uint64_t sums_cap) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::lock_guard<std::mutex> lk(p->mu);
  if (ticket < 0 || (size_t)ticket >= p->slots.size() || p->slots[ticket].state != kAcquired)
    return fail(OZCS_EINVAL, "ticket %d is not an acquired slot", ticket);
  Slot& s = p->slots[ticket];
  auto reject = [&](int code) { s.state = kFree; return code; };
  if (type < OZCS_NONE || type > OZCS_MD5) return reject(fail(OZCS_EINVAL, "Unrecognized ChecksumType: %d", type));
  if (type != OZCS_NONE && bpc == 0) return reject(fail(OZCS_EINVAL, "bytesPerChecksum must be positive"));
  if (len > p->slot_bytes) return reject(fail(OZCS_EINVAL, "len exceeds the slot size"));
  const size_t ds = ozcs_digest_size(type);
  const uint64_t n = (type == OZCS_NONE) ? 0 : ozcs_num_checksums(len, bpc);
  if (n * ds > p->sums_bytes) return reject(fail(OZCS_EINVAL, "windows this small do not fit a pipe slot's digest buffer"));
  if (n && (!sums_out || sums_cap < n * ds)) return reject(fail(OZCS_EINVAL, "sums_cap too small"));
  s.user_sums = sums_out;
  s.n_sums = n;
  s.sums_len = n * ds;
  s.status = OZCS_OK;
  s.type = type;
  s.bpc = bpc;
  s.len = len;
  s.deferred = false;
  DeviceGuard dg(p->device);
  hipError_t e = hipSuccess;
  int rc = OZCS_OK;
  const bool digest = n && (type == OZCS_SHA256 || type == OZCS_MD5);
  if (n) {
    e = hipMemcpyAsync(s.d_data, s.h_data, len, hipMemcpyHostToDevice, p->s_h2d);
    if (e == hipSuccess) e = hipEventRecord(s.copied, p->s_h2d);
    if (e == hipSuccess) {
      s_h2d.fetch_add(len, std::memory_order_relaxed);
      s_d2h.fetch_add(n * ds, std::memory_order_relaxed);
      s_bytes.fetch_add(digest ? len : 0, std::memory_order_relaxed);
    }
  }
  if (digest && e == hipSuccess) {
    s.deferred = true;
    s.state = kInFlight;
    p->deferred.push_back(ticket);
    const size_t half = p->slots.size() / 2 ? p->slots.size() / 2 : 1;
    return maybe_flush(p, half);
  }
  if (n && e == hipSuccess) {
    e = hipStreamWaitEvent(p->s_kernel, s.copied, 0);
    if (e == hipSuccess)
      rc = launch_windows(p->dev, type, bpc, s.d_data, len, 1, len, kModeCompute, s.d_sums, nullptr, 0, 0, 0, nullptr,
                          p->s_kernel);
    if (e == hipSuccess && rc == OZCS_OK) e = hipEventRecord(s.hashed, p->s_kernel);
    if (e == hipSuccess && rc == OZCS_OK) e = hipStreamWaitEvent(p->s_d2h, s.hashed, 0);
    if (e == hipSuccess && rc == OZCS_OK) e = hipMemcpyAsync(s.h_sums, s.d_sums, n * ds, hipMemcpyDeviceToHost, p->s_d2h);
  }
  if (e == hipSuccess && rc == OZCS_OK) e = hipEventRecord(s.done, p->s_d2h);
  if (e != hipSuccess) rc = fail_hip(e, "pipe submit");
  if (rc != OZCS_OK) {
    (void)hipStreamSynchronize(p->s_h2d);
    (void)hipStreamSynchronize(p->s_kernel);
    (void)hipStreamSynchronize(p->s_d2h);
    return reject(rc);
  }
  s.state = kInFlight;
  return OZCS_OK;
}

int ozcs_pipe_submit(ozcs_pipe* p, int type, uint32_t bpc, const void* data, uint64_t len, void* sums_out,
                     uint64_t sums_cap, int* ticket) {
  if (!p || !ticket || (!data && len)) return fail(OZCS_EINVAL, "null argument");
  if (len > p->slot_bytes) return fail(OZCS_EINVAL, "len exceeds the slot size");
  void* buf = nullptr;
  int rc = ozcs_pipe_acquire(p, &buf, ticket);
  if (rc != OZCS_OK) return rc;
  if (len) memcpy(buf, data, len);
  return ozcs_pipe_submit_slot(p, *ticket, type, bpc, len, sums_out, sums_cap);
}

static int finish(ozcs_pipe* p, int ticket, uint64_t* n_sums, bool block) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::unique_lock<std::mutex> lk(p->mu);
  if (ticket < 0 || (size_t)ticket >= p->slots.size()) return fail(OZCS_EINVAL, "bad ticket %d", ticket);
  Slot& s = p->slots[ticket];
  if (s.state == kAcquired) return fail(OZCS_EINVAL, "ticket %d was never submitted", ticket);
  if (s.state == kInFlight) {
    DeviceGuard dg(p->device);
    if (s.deferred) (void)flush_deferred(p);
    hipError_t e = query_event(s.done);
    if (e == hipErrorNotReady && block) {
      // wait without the pipe's lock, so that other threads keep submitting meanwhile; only the ticket's
      // holder or ozcs_pipe_drain can complete this slot, and drain leaves the same result behind
      hipEvent_t done = s.done;
      lk.unlock();
      e = hipEventSynchronize(done);
      lk.lock();
      if (s.state != kInFlight) {  // drained by another thread while this one waited
        if (n_sums) *n_sums = s.n_sums;
        return s.status;
      }
    }
    if (e == hipErrorNotReady) return OZCS_EBUSY;
    if (e != hipSuccess) {
      s.status = OZCS_EHIP;
      s.state = kFree;
      return fail_hip(e, "pipe completion");
    }
    retire(s);
  }
  if (n_sums) *n_sums = s.n_sums;
  return s.status;
}

int ozcs_pipe_poll(ozcs_pipe* p, int ticket, uint64_t* n_sums) { return finish(p, ticket, n_sums, false); }
int ozcs_pipe_wait(ozcs_pipe* p, int ticket, uint64_t* n_sums) { return finish(p, ticket, n_sums, true); }

int ozcs_pipe_drain(ozcs_pipe* p) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::lock_guard<std::mutex> lk(p->mu);
  DeviceGuard dg(p->device);
  int rc = flush_deferred(p);
  for (Slot& s : p->slots) {
    if (s.state != kInFlight) continue;
    hipError_t e = hipEventSynchronize(s.done);
    if (e != hipSuccess) {
      s.status = OZCS_EHIP;
      s.state = kFree;
      rc = fail_hip(e, "pipe drain");
    } else {
      retire(s);
    }
  }
  return rc;
}

}  // extern "C"
