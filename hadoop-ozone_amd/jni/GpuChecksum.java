/*
 * Java side of the JNI binding for libozcs (include/ozone_checksum.h).  Source only: the build image has no
 * JDK (tests/test_jni_shim.py keeps the native method list in step with ozcs_jni.c).  Modelled on org.apache.hadoop.util.NativeCRC32Wrapper (static methods, caller-allocated sums,
 * isAvailable() probe) so that Checksum can use it the way the reference's benchmarks use the Hadoop
 * native CRC.  Type constants are ContainerProtos.ChecksumType numbers, not Hadoop's 1/2.
 */
package org.apache.hadoop.ozone.common;

import java.nio.ByteBuffer;
import java.util.List;

import org.apache.hadoop.hdds.protocol.datanode.proto.ContainerProtos.ChecksumType;
import org.apache.ratis.thirdparty.com.google.protobuf.ByteString;

public final class GpuChecksum {
  private static final boolean AVAILABLE;

  /** Bit i = HIP device i; 0 = every gfx950 device of the node.  One JVM per GPU sets its own card's bit. */
  private static final int DEVICE_MASK = Integer.getInteger("ozone.gpu.checksum.device.mask", 0);
  /** Calls with fewer bytes than this stay on the Java engines (one PCIe round trip costs more than a small CRC). */
  private static final long MIN_BYTES = Long.getLong("ozone.gpu.checksum.min.bytes", 0L);
  /** Geometry of the process-wide pinned pipe that carries non-direct buffers to the GPU. */
  private static final int PIPE_SLOTS = Integer.getInteger("ozone.gpu.checksum.pipe.slots", 16);
  private static final long PIPE_SLOT_BYTES = Long.getLong("ozone.gpu.checksum.pipe.slot.bytes", 8L << 20);

  private static final Object PIPE_LOCK = new Object();
  private static long sharedPipe;          // 0 until created
  private static boolean sharedPipeFailed; // creation was tried and refused: do not try again
  private static long sharedSlotBytes;
  private static long sharedSumsBytes;

  /** Per-thread direct buffer the digests of one computeWindows call come back in. */
  private static final ThreadLocal<ByteBuffer> SUMS = new ThreadLocal<>();

  static {
    boolean ok = false;
    try {
      System.loadLibrary("ozcs_jni");   // links libozcs.so
      ok = nativeAvailable(DEVICE_MASK);
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

  /**
   * The per-window loop of Checksum.computeChecksum(ChunkBuffer) in one pass over the GPU.  {@code windows} are the
   * buffers {@code data.iterate(bytesPerChecksum)} handed out, in order; each contributes the digest of at most
   * {@code bytesPerSum} bytes from its position (the clamp of Checksum.computeChecksum(ByteBuffer, Function, int)).
   * Whatever kind the buffers are, their bytes reach the library:
   * <ul>
   * <li>direct buffers (read-only ones included) are used in place, all windows in one native call;</li>
   * <li>everything else -- heap buffers and the read-only heap views that computeChecksum(byte[]),
   *     computeChecksum(ByteBuffer), computeChecksum(List&lt;ByteString&gt;) and every verifyChecksum overload produce,
   *     which offer neither array() nor an address -- is copied once, with ByteBuffer.put, into pinned slots of a
   *     process-wide pipe (the one copy ChecksumByteBufferImpl.update pays for such buffers today) and checksummed
   *     from there, several slots in flight.</li>
   * </ul>
   * On success the digests are appended to {@code out}, one ByteString per window, each window buffer is left where
   * the Java function would have left it, and true is returned.  On false nothing was appended and no buffer was
   * touched: the caller runs its Java loop over the same windows (library unavailable, a window larger than a pipe
   * slot, every slot held by other threads, an empty window, fewer than ozone.gpu.checksum.min.bytes bytes).
   */
  public static boolean computeWindows(ChecksumType checksumType, int bytesPerSum, List<ByteBuffer> windows,
      List<ByteString> out) {
    final int type = checksumType.getNumber();
    final int ds = digestSize(type);
    final int n = windows.size();
    if (!AVAILABLE || ds == 0 || bytesPerSum <= 0 || n == 0 || (long) n * ds > Integer.MAX_VALUE) {
      return false;
    }
    final int[] lens = new int[n];
    long total = 0;
    boolean allDirect = true;
    for (int i = 0; i < n; i++) {
      final ByteBuffer w = windows.get(i);
      lens[i] = Math.min(w.remaining(), bytesPerSum);
      if (lens[i] == 0) {
        return false;   // the digest of no bytes: rare (a consumed buffer of an IncrementalChunkBuffer); Java does it
      }
      total += lens[i];
      allDirect &= w.isDirect();
    }
    if (total < MIN_BYTES) {
      return false;
    }
    final ByteBuffer sums = sumsBuffer(n * ds);
    final boolean ok = allDirect ? sumsOfDirectWindows(type, bytesPerSum, windows, lens, sums)
        : sumsThroughPipe(type, bytesPerSum, windows, lens, sums, ds);
    if (!ok) {
      return false;
    }
    final byte[] raw = new byte[n * ds];
    sums.position(0).limit(n * ds);
    sums.get(raw);
    for (int i = 0; i < n; i++) {
      out.add(ByteString.copyFrom(raw, i * ds, ds));
    }
    // MessageDigest.update(ByteBuffer) consumes what it hashes; ChecksumByteBufferImpl.update consumes only buffers
    // without an accessible array (ChecksumByteBufferImpl.java:38-47)
    final boolean digest = type == 4 || type == 5;
    for (int i = 0; i < n; i++) {
      final ByteBuffer w = windows.get(i);
      if (digest || !w.hasArray()) {
        w.position(w.position() + lens[i]);
      }
    }
    return true;
  }

  private static ByteBuffer sumsBuffer(int bytes) {
    ByteBuffer b = SUMS.get();
    if (b == null || b.capacity() < bytes) {
      b = ByteBuffer.allocateDirect(Math.max(bytes, 4096));
      SUMS.set(b);
    }
    b.clear();
    return b;
  }

  private static boolean sumsOfDirectWindows(int type, int bytesPerSum, List<ByteBuffer> windows, int[] lens,
      ByteBuffer sums) {
    final int n = lens.length;
    final ByteBuffer[] w = new ByteBuffer[n];
    final int[] positions = new int[n];
    final int[] limits = new int[n];
    for (int i = 0; i < n; i++) {
      w[i] = windows.get(i);
      positions[i] = w[i].position();
      limits[i] = positions[i] + lens[i];
    }
    return calculateWindowSumsDirect(bytesPerSum, type, sums, 0, sums.capacity(), w, positions, limits) == n;
  }

  /** The process-wide pipe for non-direct buffers, created on first use; 0 when it cannot be had. */
  private static long sharedPipe() {
    synchronized (PIPE_LOCK) {
      if (sharedPipe == 0 && !sharedPipeFailed) {
        sharedPipe = pipeCreate(0, PIPE_SLOTS, PIPE_SLOT_BYTES);
        if (sharedPipe == 0) {
          sharedPipeFailed = true;
        } else {
          sharedSlotBytes = pipeSlotBytes(sharedPipe);
          sharedSumsBytes = pipeSumsBytes(sharedPipe);
        }
      }
      return sharedPipe;
    }
  }

  /**
   * Windows are packed back to back into pinned slots; a slot is submitted as one chunk of bytesPerSum windows, so
   * a window shorter than bytesPerSum ends its slot's group (only the last window of a chunk may be short).  Slots
   * are submitted as they fill and all waited for at the end, so copies, transfers and kernels of one call overlap.
   */
  private static boolean sumsThroughPipe(int type, int bytesPerSum, List<ByteBuffer> windows, int[] lens,
      ByteBuffer sums, int ds) {
    final long pipe = sharedPipe();
    if (pipe == 0) {
      return false;
    }
    final int n = lens.length;
    for (int i = 0; i < n; i++) {
      if (lens[i] > sharedSlotBytes) {
        return false;
      }
    }
    final int maxWindowsPerSlot = (int) Math.min(Integer.MAX_VALUE, sharedSumsBytes / ds);
    final int[] tickets = new int[n];
    final int[] counts = new int[n];
    final int[] ticketOut = new int[1];
    int groups = 0;
    int harvested = 0;
    boolean ok = true;
    int i = 0;
    while (ok && i < n) {
      ByteBuffer slot = pipeAcquire(pipe, ticketOut);
      while (slot == null && harvested < groups) {   // every slot is busy: our own oldest group gives one back
        ok &= pipeWait(pipe, tickets[harvested]) == counts[harvested];
        harvested++;
        slot = pipeAcquire(pipe, ticketOut);
      }
      if (slot == null) {
        ok = false;                                  // other threads hold every slot: this call stays on the CPU
        break;
      }
      final int first = i;
      int filled = 0;
      try {
        while (i < n && i - first < maxWindowsPerSlot && filled + lens[i] <= slot.capacity()) {
          final ByteBuffer w = windows.get(i).duplicate();
          w.limit(w.position() + lens[i]);
          slot.put(w);
          filled += lens[i];
          i++;
          if (lens[i - 1] < bytesPerSum) {
            break;
          }
        }
      } catch (RuntimeException e) {
        pipeRelease(pipe, ticketOut[0]);
        throw e;
      }
      final ByteBuffer s = sums.duplicate();
      s.position(first * ds).limit(i * ds);
      if (pipeSubmit(pipe, ticketOut[0], type, bytesPerSum, filled, s.slice()) == 0) {
        tickets[groups] = ticketOut[0];
        counts[groups] = i - first;
        groups++;
      } else {
        ok = false;                                  // a rejected submission has released its slot
      }
    }
    while (harvested < groups) {                     // also after a failure: the digests land in this thread's buffer
      ok &= pipeWait(pipe, tickets[harvested]) == counts[harvested];
      harvested++;
    }
    return ok;
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
  /** Hands back an acquired slot that will not be submitted. */
  public static native int pipeRelease(long pipe, int ticket);
  /** Capacity of every slot's chunk buffer / of the digest buffer behind it (bounds the windows one slot can take). */
  public static native long pipeSlotBytes(long pipe);
  public static native long pipeSumsBytes(long pipe);
  /**
   * In-order delivery (ozcs_pipe_set_in_order): slots go back to the pool when the GPU is done with them and outcomes
   * are handed out in submission order by pipeNextInOrder -- for a BlockOutputStream that has copied the chunk into
   * its request (chunk.toByteString) and sends requests in order.  pipeAcquireWait blocks for a slot.
   */
  public static native int pipeSetInOrder(long pipe, boolean on);
  public static native long pipeNextInOrder(long pipe, boolean block, int[] ticketStatusOut);
  public static native ByteBuffer pipeAcquireWait(long pipe, int[] ticketOut);

  private static native boolean nativeAvailable(int deviceMask);
}
