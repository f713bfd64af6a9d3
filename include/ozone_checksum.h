/*
 * ozone_checksum.h -- C ABI of the MI355X (gfx950) implementation of Ozone's per-chunk data
 * checksumming: chunk buffers are cut into bytes-per-checksum windows and one CRC32 / CRC32C /
 * SHA-256 / MD5 value is produced (or checked) per window.
 *
 * This is the native boundary a JNI shim binds, in the style of the reference's existing native
 * checksum precedent NativeCRC32Wrapper (hadoop-hdds/common/src/main/java/org/apache/hadoop/util/
 * NativeCRC32Wrapper.java:31-72): static entry points, caller-allocated outputs, an availability
 * probe, no callbacks into the caller.  Plain pointers and sizes only.
 *
 * Common rules
 *  - `type` uses the numbering of ContainerProtos.ChecksumType
 *    (hadoop-hdds/interface-client/src/main/proto/DatanodeClientProtocol.proto:384-390).
 *  - A buffer of `len` bytes yields ceil(len / bpc) digests, in window order, the last window
 *    possibly shorter; 0 digests for len == 0 or type NONE (Checksum.java:155-177).
 *  - Digest bytes: CRC32/CRC32C = 4 bytes big-endian (Checksum.java:62-64); SHA-256 = 32 bytes;
 *    MD5 = 16 bytes.  Digests are written back to back.
 *  - Any bpc >= 1 and any pointer alignment are accepted.
 *  - Return value: 0 on success, a negative OZCS_E* code otherwise; ozcs_last_error() gives the
 *    calling thread's last message.  A checksum mismatch is NOT an error code: it is reported
 *    through first_mismatch, like the index OzoneChecksumException(int) carries
 *    (OzoneChecksumException.java:35-37).
 *  - Every entry point may be called from many threads at once.  Callers own their buffers; the
 *    library owns only its device scratch, staging buffers and streams.
 */
#ifndef OZONE_CHECKSUM_H
#define OZONE_CHECKSUM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ozcs_stats grew (timings, error counters); ozcs_pipe_release, ozcs_pipe_sums_bytes and the in-order
 * delivery calls of the pipe were added.  Everything of version 1 is unchanged. */
#define OZCS_ABI_VERSION 2

#if defined(__GNUC__)
#define OZCS_API __attribute__((visibility("default")))
#else
#define OZCS_API
#endif

/* ContainerProtos.ChecksumType, DatanodeClientProtocol.proto:384-390 */
enum ozcs_type {
  OZCS_NONE = 1,
  OZCS_CRC32 = 2,
  OZCS_CRC32C = 3,
  OZCS_SHA256 = 4,
  OZCS_MD5 = 5
};

enum ozcs_status {
  OZCS_OK = 0,
  OZCS_EINVAL = -1,   /* unknown type (-> OzoneChecksumException(ChecksumType), Checksum.java:163-167),
                         bpc == 0, null pointer, output capacity too small */
  OZCS_ENODEV = -2,   /* no usable GPU: the caller keeps its CPU path (cf. NativeCRC32Wrapper.isAvailable) */
  OZCS_EHIP = -3,     /* HIP runtime failure; message in ozcs_last_error() */
  OZCS_ENOMEM = -4,   /* host or device allocation failed */
  OZCS_ERANGE = -5,   /* verify: start_index + computed count exceeds the stored list
                         (ChecksumData.java:142-147) */
  OZCS_EEMPTY = -6,   /* verify: stored or computed list is empty (ChecksumData.java:121-129);
                         ozcs_pipe_next_completed: nothing is outstanding */
  OZCS_EBUSY = -7     /* pipeline: no free slot */
};

/* ---- lifetime (replaces the static initialiser + isAvailable() pair of NativeCRC32Wrapper.java:40-42) ---- */

/* Initialises the library on the devices selected by device_mask (bit i = HIP device i; 0 = all).
 * Idempotent and thread-safe.  Returns OZCS_ENODEV when no device is usable. */
OZCS_API int ozcs_init(uint32_t device_mask);
/* Releases streams and device buffers.  Not to be called while another thread is inside a library call or
 * while a pipe exists; ozcs_init may be called again afterwards. */
OZCS_API void ozcs_shutdown(void);
/* 1 when ozcs_init has succeeded (calling it first if needed), else 0. */
OZCS_API int ozcs_available(void);
OZCS_API int ozcs_device_count(void);
OZCS_API int ozcs_abi_version(void);
OZCS_API const char* ozcs_last_error(void);

/* 0 / 4 / 4 / 32 / 16 for NONE / CRC32 / CRC32C / SHA256 / MD5; 0 for unknown types. */
OZCS_API size_t ozcs_digest_size(int type);
/* ceil(len / bpc); 0 when bpc == 0. */
OZCS_API uint64_t ozcs_num_checksums(uint64_t len, uint32_t bpc);

/* ---- host-pointer, synchronous: Checksum.computeChecksum / Checksum.verifyChecksum ---- */

/* Checksum.computeChecksum(ByteBuffer) (Checksum.java:140-177), shaped like
 * NativeCRC32Wrapper.calculateChunkedSumsByteArray (NativeCRC32Wrapper.java:65-70).
 * data/sums_out are host memory.  *n_sums receives the digest count. */
OZCS_API int ozcs_compute(int type, uint32_t bpc, const void* data, uint64_t len,
                 void* sums_out, uint64_t sums_cap, uint64_t* n_sums);

/* Checksum.verifyChecksum(data, checksumData, startIndex) (Checksum.java:237-250) followed by
 * ChecksumData.verifyChecksumDataMatches (ChecksumData.java:117-149), shaped like
 * NativeCRC32Wrapper.verifyChunkedSumsByteArray (NativeCRC32Wrapper.java:52-58).
 * `expected` holds n_expected stored digests; computed[i] is compared with expected[start_index + i].
 * *first_mismatch = the first failing i (relative to the computed list), or -1.
 * type NONE returns OZCS_OK without looking at the data (Checksum.java:241-244). */
OZCS_API int ozcs_verify(int type, uint32_t bpc, const void* data, uint64_t len,
                const void* expected, uint64_t n_expected, uint64_t start_index,
                int64_t* first_mismatch);

/* ---- host-pointer, batched: many independent chunks, mixed types allowed ---- */

typedef struct ozcs_chunk {
  const void* data;     /* in : host pointer                                   */
  uint64_t len;         /* in                                                  */
  void* sums;           /* in : host output buffer                             */
  uint64_t sums_cap;    /* in : its capacity in bytes                          */
  uint32_t bpc;         /* in                                                  */
  int32_t type;         /* in : enum ozcs_type                                 */
  uint64_t n_sums;      /* out: digests written                                */
  int32_t status;       /* out: per-chunk OZCS_* code                          */
  int32_t reserved;
} ozcs_chunk;

/* The loop of BlockOutputStream.writeChunkToContainer (BlockOutputStream.java:645-656) over many
 * chunks at once.  Chunks are sharded contiguously over the initialised devices (chunk list, not
 * windows; about equal bytes per device, see ozcs_shard_chunks); there is no inter-device traffic.  Chunks of any sizes and types may be mixed: small ones (small
 * keys, tail chunks) are gathered and served by one kernel launch per checksum type, so the cost per small
 * chunk is well under a microsecond rather than a copy and a launch each.  A call in flight holds up to
 * 512 MiB of pinned host memory and as much device memory (kept for the next call of the same thread pool
 * slot, released by ozcs_shutdown).  Returns the first non-OK per-chunk status, or OZCS_OK. */
OZCS_API int ozcs_compute_batch(ozcs_chunk* chunks, uint32_t n_chunks);

/* The partition ozcs_compute_batch uses to spread a chunk list over n_parts devices (pure function, no GPU
 * needed): contiguous parts of about equal bytes, so that a list of mixed chunk sizes loads the devices evenly.
 * cuts must hold n_parts + 1 entries; part d is chunks [cuts[d], cuts[d+1]), cuts[0] = 0, cuts[n_parts] =
 * n_chunks; parts may be empty.  No counterpart in the reference, whose callers checksum one chunk at a time. */
OZCS_API void ozcs_shard_chunks(const ozcs_chunk* chunks, uint32_t n_chunks, uint32_t n_parts, uint32_t* cuts);

typedef struct ozcs_vchunk {
  const void* data;        /* in : host pointer of the bytes read                                 */
  uint64_t len;            /* in                                                                   */
  const void* expected;    /* in : host pointer of the chunk's stored digests, back to back        */
  uint64_t n_expected;     /* in : how many stored digests                                         */
  uint64_t start_index;    /* in : stored[start_index + i] is compared with computed[i]            */
  uint32_t bpc;            /* in                                                                   */
  int32_t type;            /* in : enum ozcs_type                                                  */
  int64_t first_mismatch;  /* out: first failing i, or -1                                          */
  int32_t status;          /* out: OZCS_OK / OZCS_ERANGE / OZCS_EEMPTY / ... exactly as ozcs_verify */
  int32_t reserved;
} ozcs_vchunk;

/* ozcs_verify for many chunks in one call: the loop of KeyValueContainerCheck.verifyChecksum
 * (hadoop-hdds/container-service/src/main/java/org/apache/hadoop/ozone/container/keyvalue/
 * KeyValueContainerCheck.java:262-322) over all chunks of a block or container, hashed in bulk (sharded over
 * the devices like ozcs_compute_batch).  Per chunk the outcome is the one ozcs_verify would give.  Returns
 * the first per-chunk status that is not OZCS_OK, or OZCS_OK (a mismatch is not an error). */
OZCS_API int ozcs_verify_batch(ozcs_vchunk* chunks, uint32_t n_chunks);

/* ---- device-resident: pointers are device memory on `device`, work is queued on hip_stream ----
 * (hip_stream is a hipStream_t passed as void*; NULL = the default stream.)  No implicit
 * synchronisation: results are valid once the stream reaches the end of the call's work.  These calls
 * allocate nothing and enqueue kernels only, so they may be issued on a stream under HIP graph capture. */

/* n_chunks chunks of chunk_len bytes, chunk c at d_data + c*chunk_stride, each cut into windows from
 * its own offset 0.  Digests of chunk c start at d_sums + c * ceil(chunk_len/bpc) * digest_size. */
OZCS_API int ozcs_compute_device(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                        uint64_t n_chunks, uint64_t chunk_stride, void* d_sums, uint64_t sums_cap,
                        void* hip_stream);

/* Same layout; chunk c's stored list starts at d_expected + c*expected_stride and holds n_expected
 * digests.  d_first_mismatch[c] (int64, device memory) receives the first failing window index of
 * chunk c relative to its computed list, or -1.  An index i with start_index + i >= n_expected
 * counts as failing (the host wrapper maps it to OZCS_ERANGE). */
OZCS_API int ozcs_verify_device(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                       uint64_t n_chunks, uint64_t chunk_stride, const void* d_expected,
                       uint64_t expected_stride, uint64_t n_expected, uint64_t start_index,
                       int64_t* d_first_mismatch, void* hip_stream);

/* ---- device-resident verify that allocates nothing ----
 * No counterpart in the reference.  It is for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  ozcs_verify_device wants one caller-allocated
 * verdict word per chunk; this variant keeps those words in a file-scope device array inside the library
 * (OZCS_NOALLOC_MAX_CHUNKS x 8 bytes = 512 KiB, which every GPU the library is initialised on holds until the
 * process ends) and hands back a two-word summary in caller memory, which is what the scrubber's stop-at-the-
 * first-failure loop needs (KeyValueContainerCheck.java:294-311):
 *   d_summary[0] = index of the first chunk with a failing window, or -1;
 *   d_summary[1] = that chunk's first failing window index relative to its computed list (the value
 *                  ozcs_verify_device would leave in d_first_mismatch[d_summary[0]]), or -1.
 * Same layout arguments and the same failing-index rules as ozcs_verify_device.  The call enqueues kernels on
 * hip_stream and calls no allocation routine; n_chunks above OZCS_NOALLOC_MAX_CHUNKS is rejected with
 * OZCS_EINVAL.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams
 * or threads without ordering between them) race; calls ordered on one stream do not. */
#define OZCS_NOALLOC_MAX_CHUNKS 65536
OZCS_API int ozcs_verify_device_noalloc(int device, int type, uint32_t bpc, const void* d_data, uint64_t chunk_len,
                               uint64_t n_chunks, uint64_t chunk_stride, const void* d_expected,
                               uint64_t expected_stride, uint64_t n_expected, uint64_t start_index,
                               int64_t* d_summary, void* hip_stream);

/* ---- asynchronous pinned pipeline: host buffer -> GPU -> digests back on the host ----
 * The write path of BlockOutputStream (BlockOutputStream.java:495-507, 645-690) fills one chunk buffer while
 * the previous one is checksummed and shipped.  A pipe gives that overlap across the PCIe link: n_slots
 * pinned chunk buffers (taking the place of BufferPool's heap buffers, BufferPool.java:74-85); the
 * host-to-device copy of chunk n+1 runs while the kernel of chunk n does, and kernels write their digests
 * straight into pinned host memory.  Mixed checksum types per chunk are allowed: CRC chunks complete in
 * submission order; SHA-256 / MD5 chunks are hashed in batches on streams of their own (several batches at a
 * time), never in front of a CRC chunk, and complete when their batch does.
 * A pipe may be shared by threads (calls are serialised inside).  A ticket is a non-negative int that names one
 * tenancy of one slot (slot number in its low 12 bits, see ozcs_pipe_ticket_slot); it ends when the ticket has
 * been polled to completion, waited for, handed out by ozcs_pipe_next_completed or drained -- whichever comes
 * first; the others then report the same outcome. */

typedef struct ozcs_pipe ozcs_pipe;

/* n_slots <= 4096. */
OZCS_API int ozcs_pipe_create(int device, uint32_t n_slots, uint64_t slot_bytes, ozcs_pipe** out);
OZCS_API void ozcs_pipe_destroy(ozcs_pipe* pipe);
/* Capacity of every slot buffer of the pipe in bytes (the slot_bytes it was created with); 0 for NULL.
 * A binding that wraps a slot in a bounded buffer object (a direct ByteBuffer) sizes it with this. */
OZCS_API uint64_t ozcs_pipe_slot_bytes(const ozcs_pipe* pipe);
/* The slot number (0 .. n_slots-1) a ticket belongs to, for callers that keep per-slot state; -1 for a negative
 * ticket. */
OZCS_API int ozcs_pipe_ticket_slot(int ticket);

/* Capacity in bytes of the digest buffer behind every slot: a submission whose digests need more is rejected
 * (slot_bytes / 512 + 1 digests of 32 bytes, rounded up to 256: windows of 512 bytes and more always fit).  A
 * binding that packs several windows into one slot bounds their number with this; 0 for NULL. */
OZCS_API uint64_t ozcs_pipe_sums_bytes(const ozcs_pipe* pipe);

/* Borrow a free slot: *buf is its pinned host buffer of slot_bytes bytes for the producer to fill in place
 * (zero staging copies).  OZCS_EBUSY when no slot is free: a submitted slot stays its ticket's until that
 * ticket's tenancy has ended. */
OZCS_API int ozcs_pipe_acquire(ozcs_pipe* pipe, void** buf, int* ticket);
/* Hands back an acquired slot that will not be submitted (the producer failed, or found nothing to write).
 * OZCS_EINVAL unless the ticket names an acquired, unsubmitted slot. */
OZCS_API int ozcs_pipe_release(ozcs_pipe* pipe, int ticket);
/* Queue copy + checksum for the first `len` bytes of an acquired slot.  sums_out (host, any memory) receives
 * the digests when the ticket completes.  A rejected submission (OZCS_EINVAL ...) releases the slot. */
OZCS_API int ozcs_pipe_submit_slot(ozcs_pipe* pipe, int ticket, int type, uint32_t bpc, uint64_t len,
                                   void* sums_out, uint64_t sums_cap);
/* Convenience for callers that already hold the chunk elsewhere (e.g. a heap byte[] pinned through
 * GetPrimitiveArrayCritical): acquire + memcpy into the slot + submit. */
OZCS_API int ozcs_pipe_submit(ozcs_pipe* pipe, int type, uint32_t bpc, const void* data, uint64_t len,
                              void* sums_out, uint64_t sums_cap, int* ticket);
/* OZCS_OK once the ticket's digests are in sums_out (the slot is free again), OZCS_EBUSY while in flight. */
OZCS_API int ozcs_pipe_poll(ozcs_pipe* pipe, int ticket, uint64_t* n_sums);
/* Blocks until the ticket completes. */
OZCS_API int ozcs_pipe_wait(ozcs_pipe* pipe, int ticket, uint64_t* n_sums);
/* Hands out any one completed ticket (its digests are in its sums_out, its slot is free again): the way many
 * streams that share one pipe -- or one stream whose chunks may complete out of order -- keep every slot
 * busy without waiting for a particular ticket.  Returns OZCS_OK with *ticket, *status (that ticket's own
 * OZCS_* outcome) and *n_sums set; OZCS_EBUSY when tickets are in flight but none has completed yet (only
 * with block == 0; with block != 0 the call waits for one); OZCS_EEMPTY when no submitted ticket is
 * outstanding.  status and n_sums may be NULL. */
OZCS_API int ozcs_pipe_next_completed(ozcs_pipe* pipe, int block, int* ticket, int* status, uint64_t* n_sums);
/* Blocks until every submitted ticket has completed (their slots are freed).  Returns the first failure among
 * them, or OZCS_OK.  A thread that holds one of those tickets may still poll or wait for it afterwards (its digests
 * are already in its sums_out): the pipe remembers the outcomes of a slot's last 16 tenancies, older tickets are
 * answered with OZCS_EINVAL. */
OZCS_API int ozcs_pipe_drain(ozcs_pipe* pipe);

/* ---- in-order delivery: the write path of BlockOutputStream ships chunk n before chunk n+1 ----
 * BlockOutputStream.writeChunkToContainer (BlockOutputStream.java:645-690) copies the chunk's bytes into the
 * request (chunk.toByteString, :648-649) and sends requests in the order the chunks were written.  A caller of
 * that shape needs digests in submission order but does not need the pinned slot once the GPU has read it.
 * After ozcs_pipe_set_in_order(pipe, 1) the pipe works that way: a submitted slot is released by the library
 * the moment its digests are in sums_out -- a CRC chunk (0.1 ms) does not keep its slot while an older SHA-256 /
 * MD5 chunk (milliseconds per 1 MiB window) is still being hashed -- and outcomes are queued and handed out
 * strictly in submission order by ozcs_pipe_next_in_order.  In this mode ozcs_pipe_poll, ozcs_pipe_wait and
 * ozcs_pipe_next_completed return OZCS_EINVAL (a ticket's slot may already belong to a later chunk); acquire,
 * submit, release and drain work as before (drain waits for the GPU work; the outcomes stay queued).  At most
 * OZCS_PIPE_ORDER_BACKLOG x n_slots outcomes may be waiting for delivery: beyond that ozcs_pipe_acquire
 * reports OZCS_EBUSY until some have been taken.  The mode can be changed only while no ticket is outstanding
 * and no outcome is queued (OZCS_EBUSY otherwise). */
#define OZCS_PIPE_ORDER_BACKLOG 16
OZCS_API int ozcs_pipe_set_in_order(ozcs_pipe* pipe, int on);
/* The outcome of the oldest submitted ticket not yet handed out: OZCS_OK with *ticket, *status, *n_sums set once
 * it has completed (its digests are in its sums_out); OZCS_EBUSY while it is in flight (block == 0 only; with
 * block != 0 the call waits for it); OZCS_EEMPTY when every submitted ticket has been handed out. */
OZCS_API int ozcs_pipe_next_in_order(ozcs_pipe* pipe, int block, int* ticket, int* status, uint64_t* n_sums);
/* ozcs_pipe_acquire that waits for a slot instead of reporting OZCS_EBUSY, for producers that have nothing else
 * to do meanwhile.  Still OZCS_EBUSY when waiting cannot help: every slot is held by a ticket that only its
 * holder can end (not in-order mode: acquired, or completed and unclaimed), or the in-order backlog is full. */
OZCS_API int ozcs_pipe_acquire_wait(ozcs_pipe* pipe, void** buf, int* ticket);

/* ---- counters (no counterpart in the reference, which has no checksum-throughput metric) ---- */

typedef struct ozcs_stats {
  uint64_t calls;            /* entry-point calls that launched work */
  uint64_t bytes_hashed;     /* input bytes handed to kernels        */
  uint64_t windows;          /* digests produced or checked          */
  uint64_t h2d_bytes;
  uint64_t d2h_bytes;
  uint64_t kernel_launches;
  /* ---- from OZCS_ABI_VERSION 2 on ---- */
  uint64_t errors;           /* entry-point calls that returned OZCS_EINVAL / ERANGE / EEMPTY-from-verify / EHIP /
                                ENOMEM / ENODEV (OZCS_EBUSY and the pipe's OZCS_EEMPTY are flow control, not errors) */
  uint64_t unavailable;      /* of those, OZCS_ENODEV / OZCS_EHIP / OZCS_ENOMEM: the calls a binding answers by
                                falling back to its own CPU path (the library itself has none) */
  /* Phase times of the synchronous host-pointer calls (ozcs_compute, ozcs_verify and each staged group of the
   * batched calls), measured with HIP events on the call's own stream for one call in OZCS_STATS_SAMPLE_EVERY
   * (counted over all threads, the first call of the process included), so that the events cost nothing on the
   * others: */
  uint64_t timed_calls;      /* calls (or groups) that were timed */
  uint64_t timed_bytes;      /* their input bytes */
  uint64_t h2d_us;           /* host-to-device copies, microseconds, summed over the timed calls */
  uint64_t kernel_us;        /* kernels */
  uint64_t d2h_us;           /* digests / verdicts back */
} ozcs_stats;
#define OZCS_STATS_SAMPLE_EVERY 16
OZCS_API void ozcs_get_stats(ozcs_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* OZONE_CHECKSUM_H */
