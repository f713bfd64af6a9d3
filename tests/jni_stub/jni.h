/*
 * Declaration stub of the few JNI names hadoop-ozone_amd/jni/ozcs_jni.c uses, written by hand for tests in an image
 * that has no JDK: `gcc -fsyntax-only` of the shim, so that drift between the shim, the C ABI header and
 * GpuChecksum.java is caught, and a test build of the shim that is driven through the stand-in function table of
 * fake_jvm.c.  It is not the JDK's jni.h, declares nothing else, and must never be used to build a shim that is
 * loaded into a JVM (the function table's layout here is not the real one).
 */
#ifndef OZCS_TEST_JNI_STUB_H
#define OZCS_TEST_JNI_STUB_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jobjectArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jsize (*GetArrayLength)(JNIEnv* env, jarray array);
  void (*GetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, jbyte* buf);
  void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
  void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
  jboolean (*ExceptionCheck)(JNIEnv* env);
  jobject (*NewDirectByteBuffer)(JNIEnv* env, void* address, jlong capacity);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
  jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
  jobject (*GetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index);
  void (*GetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, jint* buf);
  void (*DeleteLocalRef)(JNIEnv* env, jobject ref);
};
#endif
