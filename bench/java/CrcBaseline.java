/*
 * CPU baseline on a JVM, for boxes that have a JDK (the build image and the GPU boxes of this project do not).
 * Times the engines the reference uses in production -- java.util.zip.CRC32 and, on Java 9+, java.util.zip.CRC32C
 * (ChecksumByteBufferFactory.java:74-89) -- in the two shapes the reference itself uses:
 *
 *   streaming  the loop of BenchMarkCRCStreaming.runCRC (BenchMarkCRCStreaming.java:155-168): one 64 MiB heap
 *              buffer of random bytes, per window update(slice); getValue(); reset()
 *   chunk      what Checksum.computeChecksum does per chunk on the write path (Checksum.java:155-198): a 4 MiB
 *              chunk, a fresh engine per call, one 4-byte big-endian digest per window
 *
 * on one thread and on T threads over independent chunks (many open BlockOutputStreams), plus
 * MessageDigest SHA-256 / MD5 per window (Checksum.java:46-60) on T threads.  Prints one JSON line.
 *
 *   javac -d /tmp/ozcs_java bench/java/CrcBaseline.java && java -cp /tmp/ozcs_java CrcBaseline [bpc] [threads] [seconds]
 */
import java.nio.ByteBuffer;
import java.security.MessageDigest;
import java.util.Random;
import java.util.concurrent.atomic.AtomicLong;
import java.util.function.Supplier;
import java.util.zip.CRC32;
import java.util.zip.Checksum;

public final class CrcBaseline {
  private static final int BUFFER = 64 << 20;
  private static final int CHUNK = 4 << 20;

  private CrcBaseline() { }

  private static Supplier<Checksum> crc32c() {
    try {
      final Class<?> c = Class.forName("java.util.zip.CRC32C");
      c.getDeclaredConstructor().newInstance();
      return () -> {
        try {
          return (Checksum) c.getDeclaredConstructor().newInstance();
        } catch (ReflectiveOperationException e) {
          throw new IllegalStateException(e);
        }
      };
    } catch (ReflectiveOperationException e) {
      return null;   // Java 8: the reference falls back to Hadoop's PureJavaCrc32C, which is not in the JDK
    }
  }

  /** GB/s of the streaming shape on one thread. */
  private static double streaming(Supplier<Checksum> make, byte[] data, int bpc, double seconds) {
    final Checksum engine = make.get();
    long sink = 0;
    long bytes = 0;
    final long t0 = System.nanoTime();
    final long end = t0 + (long) (seconds * 1e9);
    do {
      for (int off = 0; off < data.length; off += bpc) {
        engine.update(data, off, Math.min(bpc, data.length - off));
        sink += engine.getValue();
        engine.reset();
      }
      bytes += data.length;
    } while (System.nanoTime() < end);
    final double dt = (System.nanoTime() - t0) / 1e9;
    if (sink == 42) {
      System.err.println();
    }
    return bytes / dt / 1e9;
  }

  /** GB/s of the per-chunk shape on `threads` threads, each over its own chunks of the buffer. */
  private static double chunks(Supplier<Checksum> make, byte[] data, int bpc, int threads, double seconds)
      throws InterruptedException {
    final AtomicLong total = new AtomicLong();
    final long t0 = System.nanoTime();
    final long end = t0 + (long) (seconds * 1e9);
    final Thread[] ts = new Thread[threads];
    final int perBuffer = data.length / CHUNK;
    for (int t = 0; t < threads; t++) {
      final int first = t;
      ts[t] = new Thread(() -> {
        final ByteBuffer sums = ByteBuffer.allocate(4 * ((CHUNK + bpc - 1) / bpc));
        long bytes = 0;
        int chunk = first % perBuffer;
        do {
          final Checksum engine = make.get();            // a fresh engine per computeChecksum call
          sums.clear();
          final int base = chunk * CHUNK;
          for (int off = 0; off < CHUNK; off += bpc) {
            engine.reset();
            engine.update(data, base + off, Math.min(bpc, CHUNK - off));
            sums.putInt((int) engine.getValue());         // Ints.toByteArray: big-endian
          }
          bytes += CHUNK;
          chunk = (chunk + threads) % perBuffer;
        } while (System.nanoTime() < end);
        total.addAndGet(bytes);
      });
      ts[t].start();
    }
    for (Thread t : ts) {
      t.join();
    }
    return total.get() / ((System.nanoTime() - t0) / 1e9) / 1e9;
  }

  /** GB/s of MessageDigest per window (Checksum.java:46-60: reset, update, digest) on `threads` threads. */
  private static double digests(String algorithm, byte[] data, int bpc, int threads, double seconds)
      throws InterruptedException {
    final AtomicLong total = new AtomicLong();
    final long t0 = System.nanoTime();
    final long end = t0 + (long) (seconds * 1e9);
    final Thread[] ts = new Thread[threads];
    final int perBuffer = data.length / CHUNK;
    for (int t = 0; t < threads; t++) {
      final int first = t;
      ts[t] = new Thread(() -> {
        try {
          final MessageDigest md = MessageDigest.getInstance(algorithm);
          long bytes = 0;
          int chunk = first % perBuffer;
          do {
            final int base = chunk * CHUNK;
            for (int off = 0; off < CHUNK; off += bpc) {
              md.reset();
              md.update(data, base + off, Math.min(bpc, CHUNK - off));
              md.digest();
            }
            bytes += CHUNK;
            chunk = (chunk + threads) % perBuffer;
          } while (System.nanoTime() < end);
          total.addAndGet(bytes);
        } catch (java.security.NoSuchAlgorithmException e) {
          throw new IllegalStateException(e);
        }
      });
      ts[t].start();
    }
    for (Thread t : ts) {
      t.join();
    }
    return total.get() / ((System.nanoTime() - t0) / 1e9) / 1e9;
  }

  public static void main(String[] args) throws Exception {
    final int bpc = args.length > 0 ? Integer.parseInt(args[0]) : 1 << 20;
    final int threads = args.length > 1 ? Integer.parseInt(args[1]) : Runtime.getRuntime().availableProcessors();
    final double seconds = args.length > 2 ? Double.parseDouble(args[2]) : 2.0;
    final byte[] data = new byte[BUFFER];
    new Random(1234).nextBytes(data);
    final Supplier<Checksum> crc32 = CRC32::new;
    final Supplier<Checksum> c32c = crc32c();
    streaming(crc32, data, bpc, 1.0);                      // warm-up: let the intrinsic compile
    final StringBuilder sb = new StringBuilder("{");
    sb.append("\"java\": \"").append(System.getProperty("java.version")).append("\", ");
    sb.append("\"bytes_per_checksum\": ").append(bpc).append(", \"threads\": ").append(threads).append(", ");
    sb.append(String.format("\"crc32_streaming_one_thread_GBps\": %.3f, ", streaming(crc32, data, bpc, seconds)));
    sb.append(String.format("\"crc32_chunk_one_thread_GBps\": %.3f, ", chunks(crc32, data, bpc, 1, seconds)));
    sb.append(String.format("\"crc32_chunk_all_threads_GBps\": %.3f", chunks(crc32, data, bpc, threads, seconds)));
    if (c32c != null) {
      streaming(c32c, data, bpc, 1.0);
      sb.append(String.format(", \"crc32c_streaming_one_thread_GBps\": %.3f", streaming(c32c, data, bpc, seconds)));
      sb.append(String.format(", \"crc32c_chunk_all_threads_GBps\": %.3f", chunks(c32c, data, bpc, threads, seconds)));
    }
    digests("MD5", data, bpc, 1, 0.5);
    sb.append(String.format(", \"sha256_all_threads_GBps\": %.3f", digests("SHA-256", data, bpc, threads, seconds)));
    sb.append(String.format(", \"md5_all_threads_GBps\": %.3f", digests("MD5", data, bpc, threads, seconds)));
    System.out.println(sb.append("}"));
  }
}
