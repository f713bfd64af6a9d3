/*
 * Java side of the JNI binding for libozcs (include/ozone_checksum.h).  Source only: the build image has no
 * JDK (tests/test_jni_shim.py keeps the native method list in step with ozcs_jni.c).  Modelled on org.apache.hadoop.util.NativeCRC32Wrapper (static methods, caller-allocated sums,
 * isAvailable() probe) so that Checksum can use it the way the reference's benchmarks use the Hadoop
 * native CRC.  Type constants are ContainerProtos.ChecksumType numbers, not Hadoop's 1/2.
 */
package org.apache.hadoop.ozone.common;

import java.nio.ByteBuffer;

public final class GpuChecksum {
  private static final boolean AVAILABLE;

  static {
    boolean ok = false;
    try {
      System.loadLibrary("ozcs_jni");   // links libozcs.so
      ok = nativeAvailable();
    } catch (UnsatisfiedLinkError e) {
      ok = false;                         // no library on java.library.path: keep the Java engines
    }
    AVAILABLE = ok;
  }

  private GpuChecksum() { }

  /** True when a gfx950 device is usable; callers keep the pure-Java path otherwise. */
  public static boolean isAvailable() {
    return AVAILABLE;
  }

  /** ozcs_digest_size. */
  public static int digestSize(int checksumType) {
    switch (checksumType) {
    case 2: case 3: return 4;
    case 4: return 32;
    case 5: return 16;
    default: return 0;
    }
  }

  /**
   * ozcs_compute over a heap array: ceil(dataLength / bytesPerSum) digests are written to
   * sums[sumsOffset ...].  Returns the digest count, or a negative OZCS_E* code (ENODEV / EHIP: fall back).
   * Shape of NativeCRC32Wrapper.calculateChunkedSumsByteArray.
   */
  public static native long calculateChunkedSumsByteArray(int bytesPerSum, int checksumType,
      byte[] sums, int sumsOffset, byte[] data, int dataOffset, int dataLength);

  /** Same for direct buffers (position..limit of data; digests at sums.position()).  Neither buffer's
   *  position moves, as with ChecksumByteBufferImpl.update on array-backed buffers. */
  public static long calculateChunkedSums(int bytesPerSum, int checksumType, ByteBuffer sums, ByteBuffer data) {
    return calculateChunkedSumsDirect(bytesPerSum, checksumType, sums, sums.position(), sums.limit(),
        data, data.position(), data.limit());
  }

  private static native long calculateChunkedSumsDirect(int bytesPerSum, int checksumType,
      ByteBuffer sums, int sumsPosition, int sumsLimit, ByteBuffer data, int dataPosition, int dataLimit);

  /**
   * One digest per buffer of the list, each over at most bytesPerSum bytes from its position, empty buffers skipped
   * (what ChunkBufferImplWithByteBufferList.iterate is meant to hand to the per-window function): the
   * computeChecksum(List&lt;ByteString&gt;) case in one native call.  Direct buffers; no position moves.  Returns the
   * digest count or a negative OZCS_E* code.
   */
  public static long calculateWindowSums(int bytesPerSum, int checksumType, ByteBuffer sums, ByteBuffer[] windows) {
    final int[] positions = new int[windows.length];
    final int[] limits = new int[windows.length];
    for (int i = 0; i < windows.length; i++) {
      positions[i] = windows[i].position();
      limits[i] = windows[i].limit();
    }
    return calculateWindowSumsDirect(bytesPerSum, checksumType, sums, sums.position(), sums.limit(), windows,
        positions, limits);
  }

  private static native long calculateWindowSumsDirect(int bytesPerSum, int checksumType,
      ByteBuffer sums, int sumsPosition, int sumsLimit, ByteBuffer[] windows, int[] positions, int[] limits);

  /**
   * ozcs_verify: returns -1 when every window matches, the first mismatching index (relative to the
   * computed list) otherwise, or a code below -1 ( -1 + OZCS_E* ) when the library could not run.
   * Shape of NativeCRC32Wrapper.verifyChunkedSumsByteArray.
   */
  public static native long verifyChunkedSumsByteArray(int bytesPerSum, int checksumType,
      byte[] sums, int sumsOffset, int numSums, int startIndex, byte[] data, int dataOffset, int dataLength);

  /** Same for direct buffers (position..limit of data; numSums stored digests from sums.position()).  Shape of
   *  NativeCRC32Wrapper.verifyChunkedSums; no position moves. */
  public static long verifyChunkedSums(int bytesPerSum, int checksumType, ByteBuffer sums, int numSums, int startIndex,
      ByteBuffer data) {
    return verifyChunkedSumsDirect(bytesPerSum, checksumType, sums, sums.position(), sums.limit(), numSums, startIndex,
        data, data.position(), data.limit());
  }

  private static native long verifyChunkedSumsDirect(int bytesPerSum, int checksumType,
      ByteBuffer sums, int sumsPosition, int sumsLimit, int numSums, int startIndex,
      ByteBuffer data, int dataPosition, int dataLimit);

  // ---- pinned pipeline: replaces BufferPool's heap buffers by pinned direct buffers ----
  public static native long pipeCreate(int device, int slots, long slotBytes);
  public static native void pipeDestroy(long pipe);
  /** Returns a direct ByteBuffer over a free pinned slot (its ticket is returned in ticketOut[0]), or null. */
  public static native ByteBuffer pipeAcquire(long pipe, int[] ticketOut);
  public static native int pipeSubmit(long pipe, int ticket, int checksumType, int bytesPerSum, long length,
      ByteBuffer sums);
  public static native long pipeWait(long pipe, int ticket);
  /** Any completed ticket: ticketStatusOut[0] = ticket, [1] = its status; returns its digest count or a negative
   *  code (OZCS_EBUSY = -7: none complete yet; OZCS_EEMPTY = -6: nothing outstanding). */
  public static native long pipeNextCompleted(long pipe, boolean block, int[] ticketStatusOut);
  /** The slot number (0 .. slots-1) behind a ticket, for per-slot state on the Java side. */
  public static native int pipeTicketSlot(int ticket);

  private static native boolean nativeAvailable();
}
