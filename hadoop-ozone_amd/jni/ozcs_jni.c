/*
 * JNI shim between org.apache.hadoop.ozone.common.GpuChecksum and the C ABI (include/ozone_checksum.h).
 * The build image has no JDK (no jni.h), so no JVM-loadable shim has been built here; the file is kept
 * syntax- and signature-checked against a declaration stub (tests/jni_stub/jni.h) and its functions are
 * run on the GPU over a stand-in function table (tests/jni_stub/fake_jvm.c, tests/test_jni_shim.py).
 * Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       ozcs_jni.c -L../csrc -lozcs -lpthread -o libozcs_jni.so
 *
 * No critical regions.  The library calls block on the GPU, may allocate and may start helper threads, none of
 * which is allowed between GetPrimitiveArrayCritical and its release (the collector is locked out for that
 * time).  Heap arrays are therefore copied out with Get<Type>ArrayRegion: a chunk that fits a slot of the
 * shim's pinned pipe goes straight from the Java heap into pinned memory (one copy, which the GPU path needs
 * anyway), larger ones and the verify inputs into a malloc'd buffer.  Direct buffers are used in place; their
 * position and limit come from the Java side as arguments, so nothing here calls back into the JVM.
 */
#include <jni.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

#include "ozone_checksum.h"

#define JFN(name) Java_org_apache_hadoop_ozone_common_GpuChecksum_##name

/* The shim's own pipe for heap-array chunks: 8 pinned slots of 8 MiB (a chunk is 4 MiB by default,
 * ScmConfigKeys.java:134) on device 0, created on first use. */
#define SHIM_SLOTS 8
#define SHIM_SLOT_BYTES (8u << 20)
static pthread_once_t g_pipe_once = PTHREAD_ONCE_INIT;
static ozcs_pipe* g_pipe = NULL;
static void make_pipe(void) {
  if (ozcs_pipe_create(0, SHIM_SLOTS, SHIM_SLOT_BYTES, &g_pipe) != OZCS_OK) g_pipe = NULL;
}

/* deviceMask: bit i = HIP device i, 0 = every gfx950 device of the node (ozcs_init's argument).  A JVM that is one of
 * several per node passes its own card's bit, so that it opens no context on the others. */
JNIEXPORT jboolean JNICALL JFN(nativeAvailable)(JNIEnv* env, jclass cls, jint deviceMask) {
  (void)env; (void)cls;
  if (deviceMask != 0 && ozcs_init((uint32_t)deviceMask) != OZCS_OK) return JNI_FALSE;
  return ozcs_available() ? JNI_TRUE : JNI_FALSE;
}

/* Digests of `len` bytes at `d` into the Java array `sums`; returns the count or a negative code. */
static jlong sums_to_array(JNIEnv* env, int rc, uint64_t n, int type, const uint8_t* out, jbyteArray sums, jint sumsOff) {
  if (rc != OZCS_OK) return (jlong)rc;
  (*env)->SetByteArrayRegion(env, sums, sumsOff, (jsize)(n * ozcs_digest_size(type)), (const jbyte*)out);
  return (*env)->ExceptionCheck(env) ? (jlong)OZCS_EINVAL : (jlong)n;
}

JNIEXPORT jlong JNICALL JFN(calculateChunkedSumsByteArray)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jbyteArray sums, jint sumsOff, jbyteArray data, jint dataOff, jint dataLen) {
  (void)cls;
  if (bpc <= 0 || dataLen < 0 || dataOff < 0 || sumsOff < 0) return OZCS_EINVAL;
  const size_t ds = ozcs_digest_size(type);
  const uint64_t n = type == OZCS_NONE ? 0 : ozcs_num_checksums((uint64_t)dataLen, (uint32_t)bpc);
  if ((jlong)dataOff + dataLen > (*env)->GetArrayLength(env, data)) return OZCS_EINVAL;
  if ((jlong)sumsOff + (jlong)(n * ds) > (*env)->GetArrayLength(env, sums)) return OZCS_EINVAL;
  if (n == 0) return ds || type == OZCS_NONE ? 0 : OZCS_EINVAL;
  uint8_t* out = (uint8_t*)malloc(n * ds);
  if (!out) return OZCS_ENOMEM;
  jlong result;
  uint64_t got = 0;
  void* slot = NULL;
  int ticket = -1;
  pthread_once(&g_pipe_once, make_pipe);
  if (g_pipe && (uint64_t)dataLen <= SHIM_SLOT_BYTES && n * ds <= ozcs_pipe_sums_bytes(g_pipe) &&
      ozcs_pipe_acquire(g_pipe, &slot, &ticket) == OZCS_OK) {
    /* Java heap -> pinned slot in one copy, then copy + kernel through the pipe */
    (*env)->GetByteArrayRegion(env, data, dataOff, dataLen, (jbyte*)slot);
    if ((*env)->ExceptionCheck(env)) {  /* the range was checked above, so none is expected */
      (void)ozcs_pipe_release(g_pipe, ticket);
      result = OZCS_EINVAL;
    } else {
      int rc = ozcs_pipe_submit_slot(g_pipe, ticket, type, (uint32_t)bpc, (uint64_t)dataLen, out, n * ds);
      if (rc == OZCS_OK) rc = ozcs_pipe_wait(g_pipe, ticket, &got);
      result = sums_to_array(env, rc, got, type, out, sums, sumsOff);
    }
  } else {
    uint8_t* d = (uint8_t*)malloc((size_t)dataLen);
    if (!d) {
      free(out);
      return OZCS_ENOMEM;
    }
    (*env)->GetByteArrayRegion(env, data, dataOff, dataLen, (jbyte*)d);
    if ((*env)->ExceptionCheck(env)) {
      result = OZCS_EINVAL;
    } else {
      const int rc = ozcs_compute(type, (uint32_t)bpc, d, (uint64_t)dataLen, out, n * ds, &got);
      result = sums_to_array(env, rc, got, type, out, sums, sumsOff);
    }
    free(d);
  }
  free(out);
  return result;
}

JNIEXPORT jlong JNICALL JFN(calculateChunkedSumsDirect)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jobject sums, jint sumsPos, jint sumsLimit, jobject data, jint dataPos, jint dataLimit) {
  (void)cls;
  uint8_t* d = (uint8_t*)(*env)->GetDirectBufferAddress(env, data);
  uint8_t* s = (uint8_t*)(*env)->GetDirectBufferAddress(env, sums);
  if (!d || !s || bpc <= 0 || dataPos < 0 || dataLimit < dataPos || sumsPos < 0 || sumsLimit < sumsPos) return OZCS_EINVAL;
  if (dataLimit > (*env)->GetDirectBufferCapacity(env, data) || sumsLimit > (*env)->GetDirectBufferCapacity(env, sums))
    return OZCS_EINVAL;
  uint64_t n = 0;
  const int rc = ozcs_compute(type, (uint32_t)bpc, d + dataPos, (uint64_t)(dataLimit - dataPos), s + sumsPos,
                              (uint64_t)(sumsLimit - sumsPos), &n);
  return rc == OZCS_OK ? (jlong)n : (jlong)rc;
}

/* One window per direct buffer, each cut to bytesPerSum bytes (Checksum.java:186-198) and empty ones skipped, as
 * ChunkBufferImplWithByteBufferList.iterate hands them out: the List<ByteString> / list-backed ChunkBuffer case in
 * one library call.  Digests go to sums[sumsPos ...] back to back; returns their count.
 *
 * Local references: every GetObjectArrayElement creates one and a native frame is only guaranteed 16, so each is
 * deleted as soon as the buffer's address has been taken (the address stays valid: the Java array keeps the
 * buffer alive for the duration of the call).
 *
 * Windows that turn out to lie back to back in memory, all bytesPerSum long except possibly the last -- the
 * slices ChunkBufferImplWithByteBuffer.iterate cuts from one direct buffer -- are one chunk: they go to
 * ozcs_compute as such (one copy, one launch) instead of one descriptor each. */
JNIEXPORT jlong JNICALL JFN(calculateWindowSumsDirect)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jobject sums, jint sumsPos, jint sumsLimit, jobjectArray windows, jintArray positions, jintArray limits) {
  (void)cls;
  uint8_t* s = (uint8_t*)(*env)->GetDirectBufferAddress(env, sums);
  const jsize n = (*env)->GetArrayLength(env, windows);
  if (!s || bpc <= 0 || sumsPos < 0 || sumsLimit < sumsPos || sumsLimit > (*env)->GetDirectBufferCapacity(env, sums)) return OZCS_EINVAL;
  if ((*env)->GetArrayLength(env, positions) < n || (*env)->GetArrayLength(env, limits) < n) return OZCS_EINVAL;
  const size_t ds = ozcs_digest_size(type);
  if (type == OZCS_NONE || n == 0) return 0;
  if (!ds) return OZCS_EINVAL;
  ozcs_chunk* jobs = (ozcs_chunk*)calloc((size_t)n, sizeof(ozcs_chunk));
  jint* pl = (jint*)malloc(2 * (size_t)n * sizeof(jint));
  if (!jobs || !pl) {
    free(jobs);
    free(pl);
    return OZCS_ENOMEM;
  }
  jlong result = 0;
  (*env)->GetIntArrayRegion(env, positions, 0, n, pl);
  if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, limits, 0, n, pl + n);
  if ((*env)->ExceptionCheck(env)) result = OZCS_EINVAL;
  uint32_t k = 0;
  int one_chunk = 1;  /* so far the windows lie back to back and all but the last are bpc long */
  for (jsize i = 0; i < n && result == 0; i++) {
    jobject w = (*env)->GetObjectArrayElement(env, windows, i);
    if ((*env)->ExceptionCheck(env)) {
      result = OZCS_EINVAL;
      break;
    }
    uint8_t* d = w ? (uint8_t*)(*env)->GetDirectBufferAddress(env, w) : NULL;
    const jlong cap = w ? (*env)->GetDirectBufferCapacity(env, w) : -1;
    if (w) (*env)->DeleteLocalRef(env, w);
    const jint pos = pl[i], lim = pl[n + i];
    if (!d || pos < 0 || lim < pos || lim > cap) {
      result = OZCS_EINVAL;
    } else if (lim > pos) {
      if ((jlong)sumsPos + (jlong)((k + 1) * ds) > sumsLimit) {
        result = OZCS_EINVAL;
        break;
      }
      jobs[k].data = d + pos;
      jobs[k].len = (uint64_t)(lim - pos < bpc ? lim - pos : bpc);
      jobs[k].sums = s + sumsPos + (size_t)k * ds;
      jobs[k].sums_cap = ds;
      jobs[k].bpc = (uint32_t)bpc;
      jobs[k].type = type;
      if (k && (jobs[k - 1].len != (uint64_t)bpc || (const uint8_t*)jobs[k - 1].data + bpc != d + pos)) one_chunk = 0;
      k++;
    }
  }
  if (result == 0 && k) {
    int rc;
    if (one_chunk && k > 1) {
      uint64_t got = 0;
      rc = ozcs_compute(type, (uint32_t)bpc, jobs[0].data, (uint64_t)(k - 1) * (uint64_t)bpc + jobs[k - 1].len, s + sumsPos,
                        (uint64_t)k * ds, &got);
      if (rc == OZCS_OK && got != k) rc = OZCS_EINVAL;
    } else {
      rc = ozcs_compute_batch(jobs, k);
    }
    result = rc == OZCS_OK ? (jlong)k : (jlong)rc;
  }
  free(jobs);
  free(pl);
  return result;
}

JNIEXPORT jlong JNICALL JFN(verifyChunkedSumsByteArray)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jbyteArray sums, jint sumsOff, jint numSums, jint startIndex, jbyteArray data, jint dataOff, jint dataLen) {
  (void)cls;
  if (bpc <= 0 || dataLen < 0 || dataOff < 0 || sumsOff < 0 || numSums < 0 || startIndex < 0) return -1 + OZCS_EINVAL;
  if ((jlong)dataOff + dataLen > (*env)->GetArrayLength(env, data)) return -1 + OZCS_EINVAL;
  const jlong sumBytes = numSums * (jlong)ozcs_digest_size(type);
  if ((jlong)sumsOff + sumBytes > (*env)->GetArrayLength(env, sums)) return -1 + OZCS_EINVAL;
  uint8_t* d = (uint8_t*)malloc((size_t)dataLen + 1);
  uint8_t* s = (uint8_t*)malloc((size_t)sumBytes + 1);
  int64_t first = -1;
  int rc = OZCS_ENOMEM;
  if (d && s) {
    (*env)->GetByteArrayRegion(env, data, dataOff, dataLen, (jbyte*)d);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetByteArrayRegion(env, sums, sumsOff, (jsize)sumBytes, (jbyte*)s);
    rc = (*env)->ExceptionCheck(env) ? OZCS_EINVAL : ozcs_verify(type, (uint32_t)bpc, d, (uint64_t)dataLen, s, (uint64_t)numSums, (uint64_t)startIndex, &first);
  }
  free(d);
  free(s);
  return rc == OZCS_OK ? (jlong)first : (jlong)(-1 + rc);
}

JNIEXPORT jlong JNICALL JFN(verifyChunkedSumsDirect)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jobject sums, jint sumsPos, jint sumsLimit, jint numSums, jint startIndex, jobject data, jint dataPos, jint dataLimit) {
  (void)cls;
  const uint8_t* d = (const uint8_t*)(*env)->GetDirectBufferAddress(env, data);
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, sums);
  if (!d || !s || bpc <= 0 || dataPos < 0 || dataLimit < dataPos || sumsPos < 0 || sumsLimit < sumsPos || numSums < 0 ||
      startIndex < 0)
    return -1 + OZCS_EINVAL;
  if (dataLimit > (*env)->GetDirectBufferCapacity(env, data) || sumsLimit > (*env)->GetDirectBufferCapacity(env, sums))
    return -1 + OZCS_EINVAL;
  if ((jlong)numSums * (jlong)ozcs_digest_size(type) > (jlong)(sumsLimit - sumsPos)) return -1 + OZCS_EINVAL;
  int64_t first = -1;
  const int rc = ozcs_verify(type, (uint32_t)bpc, d + dataPos, (uint64_t)(dataLimit - dataPos), s + sumsPos, (uint64_t)numSums,
                             (uint64_t)startIndex, &first);
  return rc == OZCS_OK ? (jlong)first : (jlong)(-1 + rc);
}

JNIEXPORT jlong JNICALL JFN(pipeCreate)(JNIEnv* env, jclass cls, jint device, jint slots, jlong slotBytes) {
  (void)env; (void)cls;
  ozcs_pipe* p = NULL;
  return ozcs_pipe_create(device, (uint32_t)slots, (uint64_t)slotBytes, &p) == OZCS_OK ? (jlong)(intptr_t)p : 0;
}

JNIEXPORT void JNICALL JFN(pipeDestroy)(JNIEnv* env, jclass cls, jlong pipe) {
  (void)env; (void)cls;
  ozcs_pipe_destroy((ozcs_pipe*)(intptr_t)pipe);
}

JNIEXPORT jobject JNICALL JFN(pipeAcquire)(JNIEnv* env, jclass cls, jlong pipe, jintArray ticketOut) {
  (void)cls;
  void* buf = NULL;
  int ticket = -1;
  if (ozcs_pipe_acquire((ozcs_pipe*)(intptr_t)pipe, &buf, &ticket) != OZCS_OK) return NULL;
  const jint t = ticket;
  (*env)->SetIntArrayRegion(env, ticketOut, 0, 1, &t);
  if ((*env)->ExceptionCheck(env)) {  /* ticketOut too short: the caller never learns the ticket */
    (void)ozcs_pipe_release((ozcs_pipe*)(intptr_t)pipe, ticket);
    return NULL;
  }
  return (*env)->NewDirectByteBuffer(env, buf, (jlong)ozcs_pipe_slot_bytes((ozcs_pipe*)(intptr_t)pipe));
}

JNIEXPORT jobject JNICALL JFN(pipeAcquireWait)(JNIEnv* env, jclass cls, jlong pipe, jintArray ticketOut) {
  (void)cls;
  void* buf = NULL;
  int ticket = -1;
  if (ozcs_pipe_acquire_wait((ozcs_pipe*)(intptr_t)pipe, &buf, &ticket) != OZCS_OK) return NULL;
  const jint t = ticket;
  (*env)->SetIntArrayRegion(env, ticketOut, 0, 1, &t);
  if ((*env)->ExceptionCheck(env)) {  /* ticketOut too short: the caller never learns the ticket */
    (void)ozcs_pipe_release((ozcs_pipe*)(intptr_t)pipe, ticket);
    return NULL;
  }
  return (*env)->NewDirectByteBuffer(env, buf, (jlong)ozcs_pipe_slot_bytes((ozcs_pipe*)(intptr_t)pipe));
}

JNIEXPORT jint JNICALL JFN(pipeRelease)(JNIEnv* env, jclass cls, jlong pipe, jint ticket) {
  (void)env; (void)cls;
  return ozcs_pipe_release((ozcs_pipe*)(intptr_t)pipe, ticket);
}

JNIEXPORT jlong JNICALL JFN(pipeSlotBytes)(JNIEnv* env, jclass cls, jlong pipe) {
  (void)env; (void)cls;
  return (jlong)ozcs_pipe_slot_bytes((ozcs_pipe*)(intptr_t)pipe);
}

JNIEXPORT jlong JNICALL JFN(pipeSumsBytes)(JNIEnv* env, jclass cls, jlong pipe) {
  (void)env; (void)cls;
  return (jlong)ozcs_pipe_sums_bytes((ozcs_pipe*)(intptr_t)pipe);
}

JNIEXPORT jint JNICALL JFN(pipeSetInOrder)(JNIEnv* env, jclass cls, jlong pipe, jboolean on) {
  (void)env; (void)cls;
  return ozcs_pipe_set_in_order((ozcs_pipe*)(intptr_t)pipe, on ? 1 : 0);
}

/* As pipeNextCompleted, but the oldest submitted ticket only (OZCS_EBUSY while it is in flight and block is false). */
JNIEXPORT jlong JNICALL JFN(pipeNextInOrder)(JNIEnv* env, jclass cls, jlong pipe, jboolean block, jintArray ticketStatusOut) {
  (void)cls;
  if ((*env)->GetArrayLength(env, ticketStatusOut) < 2) return OZCS_EINVAL;  /* before an outcome is consumed */
  int ticket = -1, status = OZCS_OK;
  uint64_t n = 0;
  const int rc = ozcs_pipe_next_in_order((ozcs_pipe*)(intptr_t)pipe, block ? 1 : 0, &ticket, &status, &n);
  if (rc != OZCS_OK) return (jlong)rc;
  const jint out[2] = {ticket, status};
  (*env)->SetIntArrayRegion(env, ticketStatusOut, 0, 2, out);
  return (jlong)n;
}

JNIEXPORT jint JNICALL JFN(pipeSubmit)(JNIEnv* env, jclass cls, jlong pipe, jint ticket, jint type, jint bpc,
    jlong length, jobject sums) {
  (void)cls;
  uint8_t* s = (uint8_t*)(*env)->GetDirectBufferAddress(env, sums);
  const jlong cap = (*env)->GetDirectBufferCapacity(env, sums);
  if (!s || bpc <= 0 || length < 0) return OZCS_EINVAL;
  return ozcs_pipe_submit_slot((ozcs_pipe*)(intptr_t)pipe, ticket, type, (uint32_t)bpc, (uint64_t)length, s, (uint64_t)cap);
}

JNIEXPORT jlong JNICALL JFN(pipeWait)(JNIEnv* env, jclass cls, jlong pipe, jint ticket) {
  (void)env; (void)cls;
  uint64_t n = 0;
  const int rc = ozcs_pipe_wait((ozcs_pipe*)(intptr_t)pipe, ticket, &n);
  return rc == OZCS_OK ? (jlong)n : (jlong)rc;
}

/* ticketStatusOut[0] = the completed ticket, [1] = its own OZCS_* outcome.  Returns its digest count, or a
 * negative code when no ticket was handed out (OZCS_EBUSY: none complete yet, OZCS_EEMPTY: none outstanding). */
JNIEXPORT jlong JNICALL JFN(pipeNextCompleted)(JNIEnv* env, jclass cls, jlong pipe, jboolean block, jintArray ticketStatusOut) {
  (void)cls;
  int ticket = -1, status = OZCS_OK;
  uint64_t n = 0;
  if ((*env)->GetArrayLength(env, ticketStatusOut) < 2) return OZCS_EINVAL;  /* before a ticket is consumed */
  const int rc = ozcs_pipe_next_completed((ozcs_pipe*)(intptr_t)pipe, block ? 1 : 0, &ticket, &status, &n);
  if (rc != OZCS_OK) return (jlong)rc;
  const jint out[2] = {ticket, status};
  (*env)->SetIntArrayRegion(env, ticketStatusOut, 0, 2, out);
  return (jlong)n;
}

JNIEXPORT jint JNICALL JFN(pipeTicketSlot)(JNIEnv* env, jclass cls, jint ticket) {
  (void)env; (void)cls;
  return ozcs_pipe_ticket_slot(ticket);
}
