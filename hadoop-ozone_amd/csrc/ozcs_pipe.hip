// Asynchronous pinned pipeline (ozcs_pipe_*): pinned chunk slots whose host-to-device copy, kernel and digest
// hand-over overlap across chunks.  See include/ozone_checksum.h for the contract.
//
// Streams.  Copies alternate over four streams, CRC kernels run on a fifth (a kernel waits for its slot's copy
// through an event), so the copy of chunk n+1 overlaps the kernel of chunk n.  Several copy streams because one
// stream carries chunk-sized copies one after the other with a gap between them: measured on MI355X (PCIe Gen5
// x16, pinned memory, 4 MiB pieces) bare copies move 50.3 GB/s on one stream and 56.7-57.1 GB/s alternating over
// two or four, which is what the link gives to copies of any size (57.2-57.6 GB/s for 256 MiB pieces).  The pipe
// itself (copy + event + kernel per chunk, CRC32 at 1 MiB windows) carries 46.6 GB/s with one copy stream, 50-55
// with two and 53.4-56.8 with four or eight; three and six are worse for mixed CRC / MD5 traffic than one (the
// runtime deals streams out over four hardware queues: GPU_MAX_HW_QUEUES), so the default is four.  Kernels write their digests straight into
// the slot's pinned, device-visible digest buffer: there is no copy-back and no third stream, and a slot is
// complete when the event behind its kernel fires.
//
// SHA-256 / MD5.  One lane hashes one window, so the windows of one chunk fill a few lanes of one wavefront
// for as long as one window takes (about 11 ms for a 1 MiB MD5 window), and the chip has room for thousands
// of such wavefronts side by side.  Digest slots are therefore parked after their copy and launched in
// batches (one launch per batch, its job list read from pinned memory) on a pool of low-priority digest streams: a batch
// starts as soon as a digest stream is idle, takes everything of its algorithm that has gathered by then, and
// several batches run concurrently.  The digest streams are low-priority so that they get hardware queues of
// their own (the runtime keeps a queue pool per priority) and a long digest kernel never sits in front of a
// CRC kernel.
//
// Tickets.  ticket = slot number | tenancy epoch << 12.  The epoch changes with every acquisition of the slot,
// so a ticket names one tenancy: a thread that waits without the lock cannot complete a later tenancy of the
// same slot, and a ticket whose tenancy has ended is recognised as such.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/ozone_checksum.h"
#include "crc_kernels.h"
#include "digest_kernels.h"
#include "ozcs_internal.h"

using namespace ozcs;

namespace {

enum SlotState {
  kFree = 0,
  kAcquired = 1,  // handed to a producer, not yet submitted
  kInFlight = 2,  // submitted; copy / kernel pending, or parked for a digest batch
  kDone = 3       // digests are in h_sums; waiting for the ticket's holder to pick them up
};

constexpr int kSlotBits = 12;  // ozcs_pipe_create caps n_slots at 4096
constexpr int kSlotMask = (1 << kSlotBits) - 1;
constexpr uint32_t kEpochMask = (1u << 18) - 1;
constexpr int kMaxDigestStreams = 8;
constexpr int kMaxCopyStreams = 8;

constexpr uint32_t kPastOutcomes = 16;
struct Past {
  uint32_t epoch = 0;
  int status = OZCS_OK;
  uint64_t n_sums = 0;
};

struct Slot {
  uint8_t* h_data = nullptr;  // pinned, slot_bytes
  uint8_t* d_data = nullptr;  // device, slot_bytes
  uint8_t* h_sums = nullptr;  // pinned, sums_bytes; kernels write it through d_sums_alias
  uint8_t* d_sums_alias = nullptr;  // device-side address of h_sums
  hipEvent_t copied = nullptr, done = nullptr;
  int copy_stream = 0;  // which copy stream carried (or carries) this tenancy's bytes
  int state = kFree;
  bool deferred = false;  // digest slot parked for a batch
  int batch = -1;         // digest stream whose running batch hashes this slot (launched digest slots)
  uint32_t epoch = 0;
  int type = 0;
  uint32_t bpc = 0;
  uint64_t len = 0;
  void* user_sums = nullptr;
  uint64_t n_sums = 0, sums_len = 0;
  int status = OZCS_OK;
  // outcomes of the last tenancies that ended, for a holder that lost a race against ozcs_pipe_drain /
  // ozcs_pipe_next_completed on another thread and asks for its ticket only later: the slot may have been let
  // and retired again by then (kPastOutcomes - 1 times at most, or the ticket is unknown)
  Past past[kPastOutcomes];
  uint32_t n_past = 0;  // tenancies retired so far; entry i lives at past[i % kPastOutcomes]
  // threads inside ozcs_pipe_wait for a tenancy of this slot, waiting without the lock: the slot is not let
  // again before they have looked at the outcome, so the record above is always the one they need
  int waiters = 0;
  uint64_t seq = 0;  // in-order mode: sequence number of this tenancy's entry in the outcome queue
};

// A batch gets a completion event of its own from a small ring: a thread that took a batch's event under the
// lock and sleeps on it outside the lock wakes when THAT batch ends, even if the stream has started its next
// batch meanwhile (one event per stream, re-recorded for every batch, made such a thread sleep through the
// next batch as well).  The ring comes round after kBatchEvents further batches of the stream, each of which
// lasts milliseconds and starts only after the previous one has ended.
constexpr int kBatchEvents = 4;

struct DigestStream {
  hipStream_t stream = nullptr;
  hipEvent_t done[kBatchEvents] = {nullptr, nullptr, nullptr, nullptr};
  int cur = 0;                // done[cur]: end of the batch running on this stream
  bool running = false;
  std::vector<int> slots;     // slots of that batch
  DigestJob* h_jobs = nullptr;        // pinned, n_slots entries: the job list of that batch, read by the kernel in place
  DigestJob* d_jobs_alias = nullptr;  // its device-side address
  double launched_s = 0;              // when that batch was launched (host clock)
  int algo = 0;
};

int make_ticket(int slot, uint32_t epoch) { return slot | (int)((epoch & kEpochMask) << kSlotBits); }

// Fault injection, test builds only (libozcs_testhooks.so; the product library does not read the variable):
// OZCS_FAULT_PIPE_DIGEST_LAUNCH = k makes the k-th digest batch launch of the process fail as if the runtime had
// refused it.
#ifdef OZCS_TEST_HOOKS
const int g_fault_digest_launch = getenv("OZCS_FAULT_PIPE_DIGEST_LAUNCH") ? atoi(getenv("OZCS_FAULT_PIPE_DIGEST_LAUNCH")) : 0;
#else
constexpr int g_fault_digest_launch = 0;
#endif
std::atomic<int> g_digest_launches{0};
// Tuning override for experiments (read once): number of digest streams per pipe.  Measured on MI355X, CRC32 : MD5
// = 3 : 1 at 1 MiB windows, 64 slots: 1 / 2 / 4 / 8 low-priority streams give 43.7 / 45.0 / 37.2 / 30.3 GB/s (the
// link gives 45): two batches side by side keep the hashing ahead of the link, more streams than hardware
// queues get in the way of the copy and CRC streams.
const int g_digest_streams = getenv("OZCS_PIPE_DIGEST_STREAMS") ? atoi(getenv("OZCS_PIPE_DIGEST_STREAMS")) : 2;
// ... and whether they are created with the lowest stream priority (default) or the normal one (0).
const int g_digest_low_prio = getenv("OZCS_PIPE_DIGEST_LOW_PRIO") ? atoi(getenv("OZCS_PIPE_DIGEST_LOW_PRIO")) : 1;
// ... and the number of copy streams the slots' host-to-device copies alternate over (default 4, see above).
const int g_copy_streams = getenv("OZCS_PIPE_COPY_STREAMS") ? atoi(getenv("OZCS_PIPE_COPY_STREAMS")) : 4;

}  // namespace

struct ozcs_pipe {
  int device = 0;
  DeviceInfo dev;
  uint64_t slot_bytes = 0, sums_bytes = 0;
  hipStream_t s_h2d[kMaxCopyStreams] = {};
  int n_copy = 1;          // copy streams in use
  uint32_t next_copy = 0;  // round-robin position
  hipStream_t s_kernel = nullptr;
  std::vector<DigestStream> digest;
  uint8_t* h_data_all = nullptr;
  uint8_t* d_data_all = nullptr;
  uint8_t* h_sums_all = nullptr;
  std::vector<Slot> slots;
  std::vector<int> deferred;   // digest slots copied (or being copied) to the device, not yet launched; FIFO
  std::vector<int> crc_fifo;   // launched CRC / NONE slots in submission order (they complete in that order)
  size_t crc_head = 0;         // first live entry of crc_fifo
  uint32_t in_flight = 0;      // slots in kInFlight
  uint32_t done_unclaimed = 0; // slots in kDone
  // pacing of digest batches (see launch_deferred): when the last batch was launched, and how long the last
  // batch of each algorithm was seen to take
  double last_launch_s = -1e9;
  double batch_s[2] = {0, 0};
  // in-order mode (ozcs_pipe_set_in_order): outcomes in submission order; a slot is released when its GPU work
  // ends and its entry here waits for ozcs_pipe_next_in_order
  struct Outcome { int ticket; int status; uint64_t n_sums; bool done; };
  bool in_order = false;
  std::deque<Outcome> order;
  uint64_t order_base = 0;  // sequence number of order.front()
  std::mutex mu;
};

namespace {

// digests of a slot never exceed this: one 32-byte digest per 16 bytes would be absurd, so the pinned
// digest buffer is sized for windows of at least 512 bytes and submit rejects what does not fit
uint64_t sums_capacity(uint64_t slot_bytes) { return (slot_bytes / 512 + 1) * 32; }

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// hipEventQuery that does not leave hipErrorNotReady behind as the thread's last error (the caller's own
// HIP code would otherwise find it there).
hipError_t query_event(hipEvent_t ev) {
  const hipError_t e = hipEventQuery(ev);
  if (e == hipErrorNotReady) (void)hipGetLastError();
  return e;
}

// Ends a tenancy whose slot is kDone: hands the digests over and frees the slot.
void retire(ozcs_pipe* p, Slot& s) {
  if (s.status == OZCS_OK && s.sums_len) memcpy(s.user_sums, s.h_sums, s.sums_len);
  s.past[s.n_past++ % kPastOutcomes] = Past{s.epoch, s.status, s.n_sums};
  s.state = kFree;
  p->done_unclaimed--;
}

void mark_done(ozcs_pipe* p, Slot& s, int status) {
  if (s.state != kInFlight) return;
  if (status != OZCS_OK) s.status = status;
  s.state = kDone;
  s.deferred = false;
  s.batch = -1;
  p->in_flight--;
  p->done_unclaimed++;
  if (p->in_order) {  // the slot goes back at once; the outcome waits in the queue for its turn
    ozcs_pipe::Outcome& o = p->order[(size_t)(s.seq - p->order_base)];
    o.status = s.status;
    o.n_sums = s.n_sums;
    o.done = true;
    retire(p, s);
  }
}

// Moves every slot whose GPU work has finished from kInFlight to kDone.  CRC slots complete in submission
// order (one kernel stream), so only the head of their FIFO is queried; digest slots complete batch by batch.
// Caller holds p->mu and has selected the device.
void reap(ozcs_pipe* p) {
  while (p->crc_head < p->crc_fifo.size()) {
    Slot& s = p->slots[p->crc_fifo[p->crc_head]];
    const hipError_t e = query_event(s.done);
    if (e == hipErrorNotReady) break;
    mark_done(p, s, e == hipSuccess ? OZCS_OK : fail_hip(e, "pipe completion"));
    p->crc_head++;
  }
  if (p->crc_head == p->crc_fifo.size()) {
    p->crc_fifo.clear();
    p->crc_head = 0;
  } else if (p->crc_head >= 1024) {
    p->crc_fifo.erase(p->crc_fifo.begin(), p->crc_fifo.begin() + (long)p->crc_head);
    p->crc_head = 0;
  }
  for (DigestStream& d : p->digest) {
    if (!d.running) continue;
    const hipError_t e = query_event(d.done[d.cur]);
    if (e == hipErrorNotReady) continue;
    const int status = e == hipSuccess ? OZCS_OK : fail_hip(e, "pipe digest batch");
    for (int t : d.slots) mark_done(p, p->slots[t], status);
    d.slots.clear();
    d.running = false;
    p->batch_s[d.algo] = now_s() - d.launched_s;
  }
}

// Launches parked digest slots on idle digest streams: per idle stream one batch of one algorithm (the one of
// the oldest parked slot), however many chunks that is (the job list is read by the kernel from pinned memory).
// Slots that find no idle stream stay parked.
//
// Pacing.  A batch lasts as long as its longest window whatever its size, so the slots a stream of chunks holds
// are fewest when the K digest streams start their batches evenly spaced, each taking what gathered meanwhile.
// Left to itself the pool does the opposite: one stream takes everything that has gathered, the next idle stream
// starts a batch of one or two chunks right behind it and is then busy for a whole batch time.  So unless `force`
// is set, a batch is started only when at least 1/K of the last observed batch time of that algorithm has
// passed since the previous start, or when half of the slots are parked.  Callers that need a particular result
// (ozcs_pipe_wait on a parked ticket) or find nothing else in flight to wait for pass force.
//
// If the runtime refuses a launch, every slot of that batch completes with the error (nothing was queued for it).
// Caller holds p->mu, has selected the device and has called reap().
// The pacing interval for the oldest parked slot's algorithm (see launch_deferred).
double pacing_gap(const ozcs_pipe* p) {
  const int type = p->slots[p->deferred[0]].type;
  const int algo = type == OZCS_SHA256 ? kAlgoSha256 : kAlgoMd5;
  double gap = p->batch_s[algo] / (double)p->digest.size();
  return gap > 0.02 ? 0.02 : gap;
}

int launch_deferred(ozcs_pipe* p, bool force = false) {
  int rc = OZCS_OK;
  for (DigestStream& d : p->digest) {
    if (p->deferred.empty()) break;
    if (d.running) continue;
    const int type = p->slots[p->deferred[0]].type;
    const int algo = type == OZCS_SHA256 ? kAlgoSha256 : kAlgoMd5;
    const double now = now_s();
    if (!force && p->deferred.size() * 2 < p->slots.size()) {
      if (now - p->last_launch_s < pacing_gap(p)) break;
    }
    uint32_t n_jobs = 0, n_windows = 0;
    d.slots.clear();
    size_t keep = 0;
    // the batch's most recently submitted slot on each copy stream: a stream carries its copies in submission order,
    // so the batch waits for these and has all its bytes
    int newest[kMaxCopyStreams];
    for (int& x : newest) x = -1;
    for (size_t i = 0; i < p->deferred.size(); i++) {
      const int t = p->deferred[i];
      Slot& s = p->slots[t];
      if (s.type != type || (uint64_t)n_windows + s.n_sums > 0x7fffffffull) {
        p->deferred[keep++] = t;
        continue;
      }
      DigestJob& j = d.h_jobs[n_jobs++];
      j.data = s.d_data;
      j.len = s.len;
      j.sums = s.d_sums_alias;
      j.bpc = s.bpc;
      j.first_window = n_windows;
      n_windows += (uint32_t)s.n_sums;
      d.slots.push_back(t);
      newest[s.copy_stream] = t;
    }
    p->deferred.resize(keep);
    hipError_t e = hipSuccess;
    for (int k = 0; k < p->n_copy; k++)
      if (newest[k] >= 0 && e == hipSuccess) e = hipStreamWaitEvent(d.stream, p->slots[newest[k]].copied, 0);
    if (e == hipSuccess) {
      const int k = g_digest_launches.fetch_add(1, std::memory_order_relaxed) + 1;
      e = (g_fault_digest_launch && k == g_fault_digest_launch) ? hipErrorLaunchFailure
                                                                : digest_launch_jobs(d.d_jobs_alias, n_jobs, n_windows, algo, p->dev.num_cus, d.stream);
    }
    const int ring = (d.cur + 1) % kBatchEvents;
    if (e == hipSuccess) e = hipEventRecord(d.done[ring], d.stream);
    if (e == hipSuccess) d.cur = ring;
    if (e != hipSuccess) {
      rc = fail_hip(e, "pipe digest batch");
      (void)hipStreamSynchronize(d.stream);  // whatever part of the batch was queued has ended before its slots are reused
      for (int t : d.slots) mark_done(p, p->slots[t], rc);
      d.slots.clear();
      continue;
    }
    for (int t : d.slots) {
      Slot& s = p->slots[t];
      s.deferred = false;
      s.batch = (int)(&d - p->digest.data());
    }
    d.running = true;
    d.launched_s = now;
    d.algo = algo;
    p->last_launch_s = now;
    s_launches.fetch_add(1, std::memory_order_relaxed);
    s_windows.fetch_add(n_windows, std::memory_order_relaxed);
  }
  return rc;
}

// An event whose completion brings the pipe forward when nothing is done yet: the oldest CRC slot's, else the
// oldest running digest batch's (the one of `prefer` if it runs in one).  Null when no GPU work is queued at all.
hipEvent_t progress_event(ozcs_pipe* p, const Slot* prefer) {
  if (prefer && prefer->state == kInFlight && !prefer->deferred) {
    if (prefer->batch >= 0) return p->digest[prefer->batch].done[p->digest[prefer->batch].cur];
    return prefer->done;
  }
  if (p->crc_head < p->crc_fifo.size()) return p->slots[p->crc_fifo[p->crc_head]].done;
  const DigestStream* first = nullptr;  // the running batch that was launched first ends first
  for (const DigestStream& d : p->digest)
    if (d.running && (!first || d.launched_s < first->launched_s)) first = &d;
  return first ? first->done[first->cur] : nullptr;
}

// Blocks the calling thread (without the pipe's lock) until `ev` has fired.  One exception: while digest slots are
// parked and a digest stream is idle, nothing but the pacing interval keeps the next batch from starting, and a
// sleeping caller is often the only thread that would notice that it has passed -- then the thread sleeps until
// that moment at the latest, looking at the event every 100 us without taking the lock.  The caller re-evaluates
// afterwards.
hipError_t wait_progress(ozcs_pipe* p, std::unique_lock<std::mutex>& lk, hipEvent_t ev) {
  double deadline = -1;
  if (!p->deferred.empty()) {
    bool idle = false;
    for (const DigestStream& d : p->digest) idle = idle || !d.running;
    if (idle) deadline = p->last_launch_s + pacing_gap(p);
  }
  lk.unlock();
  hipError_t e = hipSuccess;
  if (deadline < 0) {
    e = hipEventSynchronize(ev);
  } else {
    for (;;) {
      e = query_event(ev);
      if (e != hipErrorNotReady) break;
      const double left = deadline - now_s();
      if (left <= 0) {
        e = hipSuccess;
        break;
      }
      std::this_thread::sleep_for(std::chrono::duration<double>(left < 100e-6 ? left : 100e-6));
    }
  }
  lk.lock();
  return e;
}

void destroy_pipe(ozcs_pipe* p) {
  DeviceGuard dg(p->device);
  for (hipStream_t st : p->s_h2d)
    if (st) (void)hipStreamSynchronize(st);
  if (p->s_kernel) (void)hipStreamSynchronize(p->s_kernel);
  for (DigestStream& d : p->digest)
    if (d.stream) (void)hipStreamSynchronize(d.stream);
  for (Slot& s : p->slots) {
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (DigestStream& d : p->digest) {
    for (hipEvent_t ev : d.done)
      if (ev) (void)hipEventDestroy(ev);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    if (d.h_jobs) (void)hipHostFree(d.h_jobs);
  }
  if (p->h_data_all) (void)hipHostFree(p->h_data_all);
  if (p->d_data_all) (void)hipFree(p->d_data_all);
  if (p->h_sums_all) (void)hipHostFree(p->h_sums_all);
  for (hipStream_t st : p->s_h2d)
    if (st) (void)hipStreamDestroy(st);
  if (p->s_kernel) (void)hipStreamDestroy(p->s_kernel);
  delete p;
}

// Resolves a ticket to its slot.  Returns OZCS_OK with *stale = null for the slot's current tenancy,
// OZCS_OK with *stale = the recorded outcome when that tenancy has ended and is still on record, else OZCS_EINVAL.
int resolve(ozcs_pipe* p, int ticket, Slot** out, const Past** stale) {
  const int idx = ticket & kSlotMask;
  if (ticket < 0 || (size_t)idx >= p->slots.size()) return fail(OZCS_EINVAL, "bad ticket %d", ticket);
  Slot& s = p->slots[idx];
  const uint32_t epoch = ((uint32_t)ticket >> kSlotBits) & kEpochMask;
  *out = &s;
  *stale = nullptr;
  if (s.state != kFree && s.epoch == epoch) return OZCS_OK;
  const uint32_t kept = s.n_past < kPastOutcomes ? s.n_past : kPastOutcomes;
  for (uint32_t k = 1; k <= kept; k++) {  // newest first
    const Past& q = s.past[(s.n_past - k) % kPastOutcomes];
    if (q.epoch == epoch) {
      *stale = &q;
      return OZCS_OK;
    }
  }
  return fail(OZCS_EINVAL, "ticket %d does not name a live slot tenancy", ticket);
}

// Takes the first free slot for a new tenancy.  A slot stays its ticket holder's until that ticket has been polled
// to completion, waited for, handed out by ozcs_pipe_next_completed or drained (or, on an in-order pipe, until its
// GPU work has ended): a finished slot is never recycled behind its owner's back.  Caller holds p->mu.
bool take_free_slot(ozcs_pipe* p, void** buf, int* ticket) {
  for (size_t i = 0; i < p->slots.size(); i++) {
    Slot& s = p->slots[i];
    if (s.state == kFree && s.waiters == 0) {
      s.state = kAcquired;
      s.epoch = (s.epoch + 1) & kEpochMask;
      *buf = s.h_data;
      *ticket = make_ticket((int)i, s.epoch);
      return true;
    }
  }
  return false;
}

bool backlog_full(const ozcs_pipe* p) {
  return p->in_order && p->order.size() >= (size_t)OZCS_PIPE_ORDER_BACKLOG * p->slots.size();
}

// In-order mode: the submitted slot's place in the delivery queue.  Caller holds p->mu.
void enqueue_outcome(ozcs_pipe* p, Slot& s, int ticket) {
  if (!p->in_order) return;
  s.seq = p->order_base + p->order.size();
  p->order.push_back(ozcs_pipe::Outcome{ticket, OZCS_OK, 0, false});
}

int finish(ozcs_pipe* p, int ticket, uint64_t* n_sums, bool block) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::unique_lock<std::mutex> lk(p->mu);
  if (p->in_order) return fail(OZCS_EINVAL, "the pipe delivers in order: use ozcs_pipe_next_in_order");
  DeviceGuard dg(p->device);
  for (;;) {
    Slot* sp = nullptr;
    const Past* stale = nullptr;
    const int rc = resolve(p, ticket, &sp, &stale);
    if (rc != OZCS_OK) return rc;
    Slot& s = *sp;
    if (stale) {  // completed by ozcs_pipe_drain / ozcs_pipe_next_completed on another thread
      if (n_sums) *n_sums = stale->n_sums;
      return stale->status;
    }
    if (s.state == kAcquired) return fail(OZCS_EINVAL, "ticket %d was never submitted", ticket);
    reap(p);
    // a caller blocked on a parked ticket wants it now; otherwise parked slots start when the pacing allows
    // (a failure is recorded in the slots it concerns)
    if (!p->deferred.empty()) (void)launch_deferred(p, block && s.state == kInFlight && s.deferred);
    if (s.state == kDone) {
      const int status = s.status;
      if (n_sums) *n_sums = s.n_sums;
      retire(p, s);
      return status;
    }
    if (!block) return OZCS_EBUSY;
    // Wait without the pipe's lock, so that other threads keep submitting meanwhile.  The slot may be completed
    // and even re-let by other threads during the wait: the ticket is resolved afresh afterwards.
    hipEvent_t ev = progress_event(p, &s);
    if (!ev) return fail(OZCS_EHIP, "ticket %d is in flight but no GPU work is queued", ticket);
    s.waiters++;
    const hipError_t e = wait_progress(p, lk, ev);
    s.waiters--;
    if (e != hipSuccess) {
      Slot* again = nullptr;
      const Past* st = nullptr;
      if (resolve(p, ticket, &again, &st) == OZCS_OK && !st && again->state == kInFlight) {
        mark_done(p, *again, fail_hip(e, "pipe completion"));
      }
    }
  }
}

}  // namespace

extern "C" {

uint64_t ozcs_pipe_slot_bytes(const ozcs_pipe* pipe) { return pipe ? pipe->slot_bytes : 0; }
uint64_t ozcs_pipe_sums_bytes(const ozcs_pipe* pipe) { return pipe ? pipe->sums_bytes : 0; }

int ozcs_pipe_ticket_slot(int ticket) { return ticket < 0 ? -1 : (ticket & kSlotMask); }

static int pipe_create_impl(int device, uint32_t n_slots, uint64_t slot_bytes, ozcs_pipe** out) {
  if (!out) return fail(OZCS_EINVAL, "null out");
  *out = nullptr;
  if (n_slots == 0 || n_slots > (1u << kSlotBits) || slot_bytes == 0) return fail(OZCS_EINVAL, "bad pipe geometry");
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
  p->sums_bytes = (sums_capacity(slot_bytes) + 255) & ~(uint64_t)255;
  const uint64_t stride = (slot_bytes + 255) & ~(uint64_t)255;
  int n_digest = g_digest_streams < 1 ? 1 : g_digest_streams > kMaxDigestStreams ? kMaxDigestStreams : g_digest_streams;
  try {
    p->slots.resize(n_slots);
    p->digest.resize((size_t)n_digest);
    // so that parking, launching and queueing a slot later never allocate
    p->deferred.reserve(n_slots);
    p->crc_fifo.reserve((size_t)n_slots + 1024);
    for (DigestStream& d : p->digest) d.slots.reserve(n_slots);
  } catch (...) {
    delete p;
    return fail(OZCS_ENOMEM, "out of host memory");
  }
  int prio_low = 0, prio_high = 0;  // numerically: low priority is the larger value
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  p->n_copy = g_copy_streams < 1 ? 1 : g_copy_streams > kMaxCopyStreams ? kMaxCopyStreams : g_copy_streams;
  hipError_t e = hipSuccess;
  for (int k = 0; k < p->n_copy; k++)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_h2d[k], hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_kernel, hipStreamNonBlocking);
  for (DigestStream& d : p->digest) {
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&d.stream, hipStreamNonBlocking, g_digest_low_prio ? prio_low : 0);
    for (hipEvent_t& ev : d.done)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&d.h_jobs), sizeof(DigestJob) * n_slots, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d.d_jobs_alias), d.h_jobs, 0);
  }
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->h_data_all), stride * n_slots, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_data_all), stride * n_slots);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->h_sums_all), p->sums_bytes * n_slots, hipHostMallocDefault);
  uint8_t* d_sums_alias = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d_sums_alias), p->h_sums_all, 0);
  for (size_t i = 0; i < p->slots.size(); i++) {
    Slot& s = p->slots[i];
    if (e != hipSuccess) break;
    s.h_data = p->h_data_all + i * stride;
    s.d_data = p->d_data_all + i * stride;
    s.h_sums = p->h_sums_all + i * p->sums_bytes;
    s.d_sums_alias = d_sums_alias + i * p->sums_bytes;
    e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    const int code = (e == hipErrorOutOfMemory) ? OZCS_ENOMEM : OZCS_EHIP;
    (void)hipGetLastError();
    destroy_pipe(p);
    return fail(code, "pipe allocation: %s", hipGetErrorString(e));
  }
  *out = p;
  return OZCS_OK;
}

void ozcs_pipe_destroy(ozcs_pipe* p) {
  if (!p) return;
  destroy_pipe(p);
}

static int pipe_acquire_impl(ozcs_pipe* p, void** buf, int* ticket) {
  if (!p || !buf || !ticket) return fail(OZCS_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  if (backlog_full(p)) return fail(OZCS_EBUSY, "%zu outcomes wait for ozcs_pipe_next_in_order", p->order.size());
  if (take_free_slot(p, buf, ticket)) return OZCS_OK;
  {  // completed slots of an in-order pipe go back by themselves; parked digest slots finish only once launched
    DeviceGuard dg(p->device);
    reap(p);
    if (!p->deferred.empty()) (void)launch_deferred(p);
  }
  if (p->in_order && take_free_slot(p, buf, ticket)) return OZCS_OK;
  return fail(OZCS_EBUSY, "all %zu slots are held", p->slots.size());
}

static int pipe_acquire_wait_impl(ozcs_pipe* p, void** buf, int* ticket) {
  if (!p || !buf || !ticket) return fail(OZCS_EINVAL, "null argument");
  std::unique_lock<std::mutex> lk(p->mu);
  DeviceGuard dg(p->device);
  for (;;) {
    if (backlog_full(p)) return fail(OZCS_EBUSY, "%zu outcomes wait for ozcs_pipe_next_in_order", p->order.size());
    if (take_free_slot(p, buf, ticket)) return OZCS_OK;
    reap(p);
    if (!p->deferred.empty()) (void)launch_deferred(p);
    if (p->in_order && take_free_slot(p, buf, ticket)) return OZCS_OK;
    // waiting helps only where the library itself hands slots back, and only while GPU work is pending
    if (!p->in_order || p->in_flight == 0) return fail(OZCS_EBUSY, "all %zu slots are held", p->slots.size());
    hipEvent_t ev = progress_event(p, nullptr);
    if (!ev) {  // only parked slots are in flight: start them
      (void)launch_deferred(p, true);
      continue;
    }
    const hipError_t e = wait_progress(p, lk, ev);
    if (e != hipSuccess) return fail_hip(e, "pipe completion");
  }
}

static int pipe_release_impl(ozcs_pipe* p, int ticket) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::lock_guard<std::mutex> lk(p->mu);
  Slot* sp = nullptr;
  const Past* stale = nullptr;
  if (resolve(p, ticket, &sp, &stale) != OZCS_OK || stale || sp->state != kAcquired)
    return fail(OZCS_EINVAL, "ticket %d is not an acquired slot", ticket);
  sp->state = kFree;
  return OZCS_OK;
}

static int pipe_set_in_order_impl(ozcs_pipe* p, int on) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::lock_guard<std::mutex> lk(p->mu);
  if (p->in_flight || p->done_unclaimed || !p->order.empty())
    return fail(OZCS_EBUSY, "the delivery mode changes only while no ticket is outstanding");
  p->in_order = on != 0;
  return OZCS_OK;
}

static int pipe_next_in_order_impl(ozcs_pipe* p, int block, int* ticket, int* status, uint64_t* n_sums) {
  if (!p || !ticket) return fail(OZCS_EINVAL, "null argument");
  std::unique_lock<std::mutex> lk(p->mu);
  if (!p->in_order) return fail(OZCS_EINVAL, "the pipe is not in in-order mode (ozcs_pipe_set_in_order)");
  DeviceGuard dg(p->device);
  for (;;) {
    if (p->order.empty()) return fail(OZCS_EEMPTY, "no submitted ticket is outstanding");
    reap(p);
    const ozcs_pipe::Outcome head = p->order.front();
    if (head.done) {
      p->order.pop_front();
      p->order_base++;
      *ticket = head.ticket;
      if (status) *status = head.status;
      if (n_sums) *n_sums = head.n_sums;
      if (!p->deferred.empty()) (void)launch_deferred(p);
      return OZCS_OK;
    }
    // the head is in flight, so its slot still holds that tenancy
    Slot& s = p->slots[(size_t)(head.ticket & kSlotMask)];
    if (!p->deferred.empty()) (void)launch_deferred(p, block && s.deferred);
    if (p->order.front().done) continue;  // a refused launch completes its tickets with the error
    if (!block) return OZCS_EBUSY;
    hipEvent_t ev = progress_event(p, &s);
    if (!ev) return fail(OZCS_EHIP, "ticket %d is in flight but no GPU work is queued", head.ticket);
    const hipError_t e = wait_progress(p, lk, ev);
    if (e != hipSuccess) {
      if (!p->order.empty() && p->order.front().ticket == head.ticket && !p->order.front().done && s.state == kInFlight)
        mark_done(p, s, fail_hip(e, "pipe completion"));
    }
  }
}

static int pipe_submit_slot_impl(ozcs_pipe* p, int ticket, int type, uint32_t bpc, uint64_t len, void* sums_out,
                          uint64_t sums_cap) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::lock_guard<std::mutex> lk(p->mu);
  Slot* sp = nullptr;
  const Past* stale = nullptr;
  if (resolve(p, ticket, &sp, &stale) != OZCS_OK || stale || sp->state != kAcquired)
    return fail(OZCS_EINVAL, "ticket %d is not an acquired slot", ticket);
  Slot& s = *sp;
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
  s.batch = -1;
  DeviceGuard dg(p->device);
  reap(p);
  hipError_t e = hipSuccess;
  int rc = OZCS_OK;
  const bool digest = n && (type == OZCS_SHA256 || type == OZCS_MD5);
  if (n) {
    s.copy_stream = (int)(p->next_copy++ % (uint32_t)p->n_copy);
    e = hipMemcpyAsync(s.d_data, s.h_data, len, hipMemcpyHostToDevice, p->s_h2d[s.copy_stream]);
    if (e == hipSuccess) e = hipEventRecord(s.copied, p->s_h2d[s.copy_stream]);
  }
  if (digest && e == hipSuccess) {
    s.deferred = true;
    s.state = kInFlight;
    p->in_flight++;
    enqueue_outcome(p, s, ticket);
    p->deferred.push_back(ticket & kSlotMask);
    s_h2d.fetch_add(len, std::memory_order_relaxed);
    s_d2h.fetch_add(n * ds, std::memory_order_relaxed);
    s_bytes.fetch_add(len, std::memory_order_relaxed);
    (void)launch_deferred(p);  // a refused launch is reported through the tickets of its batch
    return OZCS_OK;
  }
  if (n && e == hipSuccess) {
    e = hipStreamWaitEvent(p->s_kernel, s.copied, 0);
    if (e == hipSuccess)
      rc = launch_windows(p->dev, type, bpc, s.d_data, len, 1, len, kModeCompute, s.d_sums_alias, nullptr, 0, 0, 0, nullptr,
                          p->s_kernel);
  }
  if (e == hipSuccess && rc == OZCS_OK) e = hipEventRecord(s.done, p->s_kernel);
  if (e != hipSuccess) rc = fail_hip(e, "pipe submit");
  if (rc != OZCS_OK) {
    for (int k = 0; k < p->n_copy; k++) (void)hipStreamSynchronize(p->s_h2d[k]);
    (void)hipStreamSynchronize(p->s_kernel);
    return reject(rc);
  }
  if (n) {
    s_h2d.fetch_add(len, std::memory_order_relaxed);
    s_d2h.fetch_add(n * ds, std::memory_order_relaxed);
  }
  s.state = kInFlight;
  p->in_flight++;
  enqueue_outcome(p, s, ticket);
  p->crc_fifo.push_back(ticket & kSlotMask);
  if (!p->deferred.empty()) (void)launch_deferred(p);  // a digest stream may have become idle meanwhile
  return OZCS_OK;
}

static int pipe_submit_impl(ozcs_pipe* p, int type, uint32_t bpc, const void* data, uint64_t len, void* sums_out,
                     uint64_t sums_cap, int* ticket) {
  if (!p || !ticket || (!data && len)) return fail(OZCS_EINVAL, "null argument");
  if (len > p->slot_bytes) return fail(OZCS_EINVAL, "len exceeds the slot size");
  void* buf = nullptr;
  int rc = pipe_acquire_impl(p, &buf, ticket);
  if (rc != OZCS_OK) return rc;
  if (len) pooled_copy(buf, data, len);
  return pipe_submit_slot_impl(p, *ticket, type, bpc, len, sums_out, sums_cap);
}

int ozcs_pipe_poll(ozcs_pipe* p, int ticket, uint64_t* n_sums) { return tally(finish(p, ticket, n_sums, false)); }
int ozcs_pipe_wait(ozcs_pipe* p, int ticket, uint64_t* n_sums) { return tally(finish(p, ticket, n_sums, true)); }

static int pipe_next_completed_impl(ozcs_pipe* p, int block, int* ticket, int* status, uint64_t* n_sums) {
  if (!p || !ticket) return fail(OZCS_EINVAL, "null argument");
  std::unique_lock<std::mutex> lk(p->mu);
  if (p->in_order) return fail(OZCS_EINVAL, "the pipe delivers in order: use ozcs_pipe_next_in_order");
  DeviceGuard dg(p->device);
  for (;;) {
    reap(p);
    if (!p->deferred.empty()) (void)launch_deferred(p);
    if (p->done_unclaimed) {
      for (size_t i = 0; i < p->slots.size(); i++) {
        Slot& s = p->slots[i];
        if (s.state != kDone) continue;
        *ticket = make_ticket((int)i, s.epoch);
        if (status) *status = s.status;
        if (n_sums) *n_sums = s.n_sums;
        retire(p, s);
        return OZCS_OK;
      }
    }
    if (p->in_flight == 0) return fail(OZCS_EEMPTY, "no submitted ticket is outstanding");
    if (!block) return OZCS_EBUSY;
    hipEvent_t ev = progress_event(p, nullptr);
    if (!ev && !p->deferred.empty()) {  // only parked slots are left: nothing to wait for until they are launched
      (void)launch_deferred(p, true);
      continue;
    }
    if (!ev) return fail(OZCS_EHIP, "tickets are in flight but no GPU work is queued");
    const hipError_t e = wait_progress(p, lk, ev);
    if (e != hipSuccess) return fail_hip(e, "pipe completion");
  }
}

static int pipe_drain_impl(ozcs_pipe* p) {
  if (!p) return fail(OZCS_EINVAL, "null pipe");
  std::unique_lock<std::mutex> lk(p->mu);
  DeviceGuard dg(p->device);
  int rc = OZCS_OK;
  for (;;) {
    reap(p);
    if (!p->deferred.empty()) (void)launch_deferred(p);
    for (Slot& s : p->slots) {
      if (s.state != kDone) continue;
      if (s.status != OZCS_OK && rc == OZCS_OK) rc = s.status;
      retire(p, s);
    }
    if (p->in_flight == 0) {
      for (const ozcs_pipe::Outcome& o : p->order)
        if (o.status != OZCS_OK && rc == OZCS_OK) rc = o.status;
      return rc;
    }
    hipEvent_t ev = progress_event(p, nullptr);
    if (!ev && !p->deferred.empty()) {  // only parked slots are left: nothing to wait for until they are launched
      (void)launch_deferred(p, true);
      continue;
    }
    if (!ev) return fail(OZCS_EHIP, "tickets are in flight but no GPU work is queued");
    const hipError_t e = wait_progress(p, lk, ev);
    if (e != hipSuccess) return fail_hip(e, "pipe drain");
  }
}

// ---- the exported names: each counts its outcome (ozcs_get_stats) and hands it on ----
int ozcs_pipe_create(int device, uint32_t n_slots, uint64_t slot_bytes, ozcs_pipe** out) {
  return tally(pipe_create_impl(device, n_slots, slot_bytes, out));
}
int ozcs_pipe_acquire(ozcs_pipe* p, void** buf, int* ticket) { return tally(pipe_acquire_impl(p, buf, ticket)); }
int ozcs_pipe_acquire_wait(ozcs_pipe* p, void** buf, int* ticket) { return tally(pipe_acquire_wait_impl(p, buf, ticket)); }
int ozcs_pipe_release(ozcs_pipe* p, int ticket) { return tally(pipe_release_impl(p, ticket)); }
int ozcs_pipe_submit_slot(ozcs_pipe* p, int ticket, int type, uint32_t bpc, uint64_t len, void* sums_out,
                          uint64_t sums_cap) {
  return tally(pipe_submit_slot_impl(p, ticket, type, bpc, len, sums_out, sums_cap));
}
int ozcs_pipe_submit(ozcs_pipe* p, int type, uint32_t bpc, const void* data, uint64_t len, void* sums_out,
                     uint64_t sums_cap, int* ticket) {
  return tally(pipe_submit_impl(p, type, bpc, data, len, sums_out, sums_cap, ticket));
}
int ozcs_pipe_next_completed(ozcs_pipe* p, int block, int* ticket, int* status, uint64_t* n_sums) {
  return tally(pipe_next_completed_impl(p, block, ticket, status, n_sums), /*empty_is_flow=*/true);
}
int ozcs_pipe_set_in_order(ozcs_pipe* p, int on) { return tally(pipe_set_in_order_impl(p, on)); }
int ozcs_pipe_next_in_order(ozcs_pipe* p, int block, int* ticket, int* status, uint64_t* n_sums) {
  return tally(pipe_next_in_order_impl(p, block, ticket, status, n_sums), /*empty_is_flow=*/true);
}
int ozcs_pipe_drain(ozcs_pipe* p) { return tally(pipe_drain_impl(p)); }

}  // extern "C"
