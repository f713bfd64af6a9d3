/*
 * JNI shim between org.apache.hadoop.ozone.common.GpuChecksum and the C ABI (include/ozone_checksum.h).
 * Source only: the build image has no JDK (no jni.h), so this file has never been compiled here.
 * Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       ozcs_jni.c -L../csrc -lozcs -o libozcs_jni.so
 *
 * Heap arrays are reached with GetPrimitiveArrayCritical (no copy on HotSpot; the library then stages
 * the bytes to the device), direct buffers with GetDirectBufferAddress.  Nothing calls back into the JVM.
 * The critical section spans the whole synchronous call, as in Hadoop's NativeCrc32 (about 0.2 ms for a
 * 4 MiB chunk); callers with very large arrays should use the pinned pipe (direct buffers) instead.
 */
#include <jni.h>
#include <stdint.h>

#include "ozone_checksum.h"

#define JFN(name) Java_org_apache_hadoop_ozone_common_GpuChecksum_##name

JNIEXPORT jboolean JNICALL JFN(nativeAvailable)(JNIEnv* env, jclass cls) {
  (void)env; (void)cls;
  return ozcs_available() ? JNI_TRUE : JNI_FALSE;
}

JNIEXPORT jlong JNICALL JFN(calculateChunkedSumsByteArray)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jbyteArray sums, jint sumsOff, jbyteArray data, jint dataOff, jint dataLen) {
  (void)cls;
  if (bpc <= 0 || dataLen < 0 || dataOff < 0 || sumsOff < 0) return OZCS_EINVAL;
  const jsize sumsLen = (*env)->GetArrayLength(env, sums);
  if ((jlong)dataOff + dataLen > (*env)->GetArrayLength(env, data) || sumsOff > sumsLen) return OZCS_EINVAL;
  uint8_t* d = (uint8_t*)(*env)->GetPrimitiveArrayCritical(env, data, NULL);
  if (!d) return OZCS_ENOMEM;
  uint8_t* s = (uint8_t*)(*env)->GetPrimitiveArrayCritical(env, sums, NULL);
  uint64_t n = 0;
  int rc = OZCS_ENOMEM;
  if (s) {
    rc = ozcs_compute(type, (uint32_t)bpc, d + dataOff, (uint64_t)dataLen, s + sumsOff,
                      (uint64_t)(sumsLen - sumsOff), &n);
    (*env)->ReleasePrimitiveArrayCritical(env, sums, s, 0);
  }
  (*env)->ReleasePrimitiveArrayCritical(env, data, d, JNI_ABORT);
  return rc == OZCS_OK ? (jlong)n : (jlong)rc;
}

JNIEXPORT jlong JNICALL JFN(calculateChunkedSums)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jobject sums, jobject data) {
  (void)cls;
  uint8_t* d = (uint8_t*)(*env)->GetDirectBufferAddress(env, data);
  uint8_t* s = (uint8_t*)(*env)->GetDirectBufferAddress(env, sums);
  if (!d || !s || bpc <= 0) return OZCS_EINVAL;
  /* position/limit are read through the Buffer accessors */
  jclass bufCls = (*env)->FindClass(env, "java/nio/Buffer");
  jmethodID pos = (*env)->GetMethodID(env, bufCls, "position", "()I");
  jmethodID lim = (*env)->GetMethodID(env, bufCls, "limit", "()I");
  const jint dp = (*env)->CallIntMethod(env, data, pos), dl = (*env)->CallIntMethod(env, data, lim);
  const jint sp = (*env)->CallIntMethod(env, sums, pos), sl = (*env)->CallIntMethod(env, sums, lim);
  uint64_t n = 0;
  const int rc = ozcs_compute(type, (uint32_t)bpc, d + dp, (uint64_t)(dl - dp), s + sp, (uint64_t)(sl - sp), &n);
  return rc == OZCS_OK ? (jlong)n : (jlong)rc;
}

JNIEXPORT jlong JNICALL JFN(verifyChunkedSumsByteArray)(JNIEnv* env, jclass cls, jint bpc, jint type,
    jbyteArray sums, jint sumsOff, jint numSums, jint startIndex, jbyteArray data, jint dataOff, jint dataLen) {
  (void)cls;
  if (bpc <= 0 || dataLen < 0 || dataOff < 0 || sumsOff < 0 || numSums < 0 || startIndex < 0) return -1 + OZCS_EINVAL;
  if ((jlong)dataOff + dataLen > (*env)->GetArrayLength(env, data)) return -1 + OZCS_EINVAL;
  {
    const jlong ds = (jlong)ozcs_digest_size(type);
    if ((jlong)sumsOff + numSums * ds > (*env)->GetArrayLength(env, sums)) return -1 + OZCS_EINVAL;
  }
  uint8_t* d = (uint8_t*)(*env)->GetPrimitiveArrayCritical(env, data, NULL);
  if (!d) return -1 + OZCS_ENOMEM;
  uint8_t* s = (uint8_t*)(*env)->GetPrimitiveArrayCritical(env, sums, NULL);
  int64_t first = -1;
  int rc = OZCS_ENOMEM;
  if (s) {
    rc = ozcs_verify(type, (uint32_t)bpc, d + dataOff, (uint64_t)dataLen, s + sumsOff, (uint64_t)numSums,
                     (uint64_t)startIndex, &first);
    (*env)->ReleasePrimitiveArrayCritical(env, sums, s, JNI_ABORT);
  }
  (*env)->ReleasePrimitiveArrayCritical(env, data, d, JNI_ABORT);
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
  return (*env)->NewDirectByteBuffer(env, buf, (jlong)ozcs_pipe_slot_bytes((ozcs_pipe*)(intptr_t)pipe));
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
