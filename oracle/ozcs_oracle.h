/*
 * CPU oracle for Ozone's per-chunk data checksumming.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing shipped by the library may include, link
 * or call this code: it is the checker the HIP path is compared with (tests,
 * the smoke check and the CPU-baseline leg of the benchmark), never the thing
 * being measured or delivered.
 *
 * Each function restates, in plain C, one piece of the reference Java path:
 *   - CRC tables   : PureJavaCrc32ByteBuffer.java:35-556 (poly 0xEDB88320),
 *                    PureJavaCrc32CByteBuffer.java:38-559 (poly 0x82F63B78);
 *                    8 x 256 ints, T8_k at offset k*0x100.
 *   - CRC update   : ChecksumByteBuffer.java:82-120 (slice-by-8 + byte tail),
 *                    reset/getValue :63-70.
 *   - windowing    : Checksum.java:155-198 with the segmentation of
 *                    ChunkBufferImplWithByteBuffer.java:62-83.
 *   - CRC encoding : Checksum.java:62-64 (Guava Ints.toByteArray, big-endian).
 *   - verify       : ChecksumData.java:117-149, Checksum.java:237-250.
 *   - SHA-256, MD5 : the JDK MessageDigest algorithms the reference calls at
 *                    Checksum.java:46-60,83-84, i.e. FIPS 180-4 and RFC 1321
 *                    (JDK source is not part of the reference tree; the
 *                    published algorithms are restated here).
 *
 * Pinned by: tests/test_oracle.py against tests/golden/ (table digests of the
 * reference's own Java tables, CRC/SHA/MD5 known-answer vectors, vectors
 * produced by zlib / hashlib, and for SHA-256 the one literal digest of known
 * content that a test of the reference holds for the same MessageDigest,
 * hadoop-ozone/s3gateway/src/test/java/org/apache/hadoop/ozone/s3/signature/
 * TestStringToSignProducer.java:86).  MD5: parity unpinned by the reference --
 * no test of it holds an MD5 digest of known content; RFC 1321 vectors and
 * hashlib only.
 */
#ifndef OZCS_ORACLE_H
#define OZCS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ContainerProtos.ChecksumType numbering, DatanodeClientProtocol.proto:384-390 */
enum {
  ORACLE_NONE = 1,
  ORACLE_CRC32 = 2,
  ORACLE_CRC32C = 3,
  ORACLE_SHA256 = 4,
  ORACLE_MD5 = 5
};

#define ORACLE_POLY_CRC32 0xEDB88320u
#define ORACLE_POLY_CRC32C 0x82F63B78u

/* Return codes of oracle_verify, mirroring ChecksumData.verifyChecksumDataMatches. */
enum {
  ORACLE_VERIFY_OK = 0,            /* returns true                                  */
  ORACLE_VERIFY_MISMATCH = 1,      /* OzoneChecksumException(index)                 */
  ORACLE_VERIFY_ORIG_EMPTY = 2,    /* "Original checksumData has no checksums"      */
  ORACLE_VERIFY_COMPUTED_EMPTY = 3,/* "Computed checksumData has no checksums"      */
  ORACLE_VERIFY_OUT_OF_RANGE = 4,  /* startIndex + index beyond the stored list     */
  ORACLE_VERIFY_BAD_TYPE = -1
};

/* 8 x 256 table in the layout of PureJavaCrc32[C]ByteBuffer.T. */
void oracle_crc_make_table(uint32_t reflected_poly, uint32_t table[2048]);

/* CrcIntTable.update(int crc, ByteBuffer b, int[] table): raw register in, raw register out. */
uint32_t oracle_crc_update(uint32_t crc, const uint8_t *data, size_t len,
                           const uint32_t table[2048]);

/* reset(); update(data); getValue() for one buffer. */
uint32_t oracle_crc32(const uint8_t *data, size_t len);
uint32_t oracle_crc32c(const uint8_t *data, size_t len);

void oracle_sha256(const uint8_t *data, size_t len, uint8_t out[32]);
void oracle_md5(const uint8_t *data, size_t len, uint8_t out[16]);

size_t oracle_digest_size(int type);

/*
 * Checksum.computeChecksum(ChunkBuffer) for a contiguous buffer.
 * Writes ceil(len/bpc) digests back to back (0 for len == 0 or type NONE).
 * Returns the number of digests, or -1 for an unknown type / bpc == 0 /
 * too small an output buffer.
 */
int64_t oracle_compute(int type, uint32_t bpc, const uint8_t *data, uint64_t len,
                       uint8_t *sums_out, uint64_t sums_cap);

/*
 * Checksum.verifyChecksum(data, stored, startIndex): recompute with the stored
 * type and bpc, then ChecksumData.verifyChecksumDataMatches.  *first_mismatch
 * is the index relative to the computed list (as the exception reports it),
 * or -1.
 */
int oracle_verify(int type, uint32_t bpc, const uint8_t *data, uint64_t len,
                  const uint8_t *stored, uint64_t n_stored, uint64_t start_index,
                  int64_t *first_mismatch);

/* SplitMix64 byte stream used by every test and benchmark of this repository:
 * 8 little-endian bytes per 64-bit output, state starts at `seed`. */
void oracle_fill_splitmix64(uint64_t seed, uint8_t *out, uint64_t len);

/* Times `reps` passes of oracle_compute over `len` bytes on `threads` threads
 * (each thread owns an independent copy-free slice of whole windows) and
 * returns the best wall time of one pass in seconds. */
double oracle_bench(int type, uint32_t bpc, const uint8_t *data, uint64_t len,
                    int threads, int reps);

/* crc_clmul.c: the same CRCs by PCLMULQDQ folding (what the JDK's CRC32 / CRC32C intrinsics are built on), for
 * the benchmark's CPU baseline; falls back to the table method where the instruction is missing. */
int oracle_has_clmul(void);
uint32_t oracle_crc_clmul(int type, const uint8_t *data, size_t len);
int64_t oracle_compute_clmul(int type, uint32_t bpc, const uint8_t *data, uint64_t len, uint8_t *out, uint64_t cap);
/* oracle_bench with oracle_compute_clmul as the worker (CRC types only). */
double oracle_bench_clmul(int type, uint32_t bpc, const uint8_t *data, uint64_t len, int threads, int reps);

#ifdef __cplusplus
}
#endif
#endif
