/*
 * CPU oracle for Ozone's per-chunk data checksumming -- see ozcs_oracle.h.
 * TEST INFRASTRUCTURE ONLY: never linked into, or called by, the shipped library.
 */
#include "ozcs_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ CRC -- */

/* Table layout of PureJavaCrc32ByteBuffer.java:35-556 / PureJavaCrc32CByteBuffer.java:38-559:
 * T8_0 is the classic reflected byte table, T8_k[i] = (T8_{k-1}[i] >>> 8) ^ T8_0[T8_{k-1}[i] & 0xff]. */
void oracle_crc_make_table(uint32_t poly, uint32_t table[2048]) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ poly : (c >> 1);
    table[i] = c;
  }
  for (int k = 1; k < 8; k++)
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t p = table[(k - 1) * 256 + i];
      table[k * 256 + i] = (p >> 8) ^ table[p & 0xffu];
    }
}

/* ChecksumByteBuffer.java:82-120. */
uint32_t oracle_crc_update(uint32_t crc, const uint8_t *b, size_t len, const uint32_t table[2048]) {
  size_t pos = 0;
  while (len - pos > 7) {
    const uint32_t c0 = (b[pos + 0] ^ crc) & 0xff;
    const uint32_t c1 = (b[pos + 1] ^ (crc >>= 8)) & 0xff;
    const uint32_t c2 = (b[pos + 2] ^ (crc >>= 8)) & 0xff;
    const uint32_t c3 = (b[pos + 3] ^ (crc >> 8)) & 0xff;
    crc = (table[0x700 + c0] ^ table[0x600 + c1]) ^ (table[0x500 + c2] ^ table[0x400 + c3]);
    const uint32_t c4 = b[pos + 4], c5 = b[pos + 5], c6 = b[pos + 6], c7 = b[pos + 7];
    crc ^= (table[0x300 + c4] ^ table[0x200 + c5]) ^ (table[0x100 + c6] ^ table[c7]);
    pos += 8;
  }
  while (pos < len) { /* the fall-through switch at :101-117 */
    crc = (crc >> 8) ^ table[(crc ^ b[pos]) & 0xff];
    pos++;
  }
  return crc;
}

static uint32_t g_tab32[2048], g_tab32c[2048];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_tables(void) {
  oracle_crc_make_table(ORACLE_POLY_CRC32, g_tab32);
  oracle_crc_make_table(ORACLE_POLY_CRC32C, g_tab32c);
}

/* reset() -> 0xffffffff (:68-70); getValue() -> ~crc (:63-65). */
uint32_t oracle_crc32(const uint8_t *d, size_t n) {
  pthread_once(&g_once, init_tables);
  return ~oracle_crc_update(0xffffffffu, d, n, g_tab32);
}
uint32_t oracle_crc32c(const uint8_t *d, size_t n) {
  pthread_once(&g_once, init_tables);
  return ~oracle_crc_update(0xffffffffu, d, n, g_tab32c);
}

/* -------------------------------------------------------------- SHA-256 -- */
/* FIPS 180-4 section 6.2. */

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline uint32_t rol32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

static void sha256_block(uint32_t h[8], const uint8_t *p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++)
    w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void oracle_sha256(const uint8_t *data, size_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t full = len / 64;
  for (size_t i = 0; i < full; i++) sha256_block(h, data + 64 * i);
  uint8_t tail[128];
  size_t rem = len - 64 * full;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, data + 64 * full, rem);
  tail[rem] = 0x80;
  size_t tl = (rem < 56) ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha256_block(h, tail);
  if (tl == 128) sha256_block(h, tail + 64);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
  }
}

/* ------------------------------------------------------------------ MD5 -- */
/* RFC 1321 section 3.4. */

static const uint32_t KMD5[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t SMD5[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

static void md5_block(uint32_t s[4], const uint8_t *p) {
  uint32_t m[16];
  for (int i = 0; i < 16; i++)
    m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  uint32_t a = s[0], b = s[1], c = s[2], d = s[3];
  for (int i = 0; i < 64; i++) {
    uint32_t f;
    int g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    uint32_t t = d;
    d = c; c = b;
    b = b + rol32(a + f + KMD5[i] + m[g], SMD5[i]);
    a = t;
  }
  s[0] += a; s[1] += b; s[2] += c; s[3] += d;
}

void oracle_md5(const uint8_t *data, size_t len, uint8_t out[16]) {
  uint32_t s[4] = {0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476};
  size_t full = len / 64;
  for (size_t i = 0; i < full; i++) md5_block(s, data + 64 * i);
  uint8_t tail[128];
  size_t rem = len - 64 * full;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, data + 64 * full, rem);
  tail[rem] = 0x80;
  size_t tl = (rem < 56) ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 8 + i] = (uint8_t)(bits >> (8 * i));
  md5_block(s, tail);
  if (tl == 128) md5_block(s, tail + 64);
  for (int i = 0; i < 4; i++) {
    out[4 * i] = (uint8_t)s[i]; out[4 * i + 1] = (uint8_t)(s[i] >> 8);
    out[4 * i + 2] = (uint8_t)(s[i] >> 16); out[4 * i + 3] = (uint8_t)(s[i] >> 24);
  }
}

/* ------------------------------------------------------------ windowing -- */

size_t oracle_digest_size(int type) {
  switch (type) {
    case ORACLE_NONE: return 0;
    case ORACLE_CRC32: case ORACLE_CRC32C: return 4;
    case ORACLE_SHA256: return 32;
    case ORACLE_MD5: return 16;
    default: return 0;
  }
}

/* One window -> one digest: the per-window function of Checksum.java:55-59 (digests)
 * and :69-73 (CRC: reset, update, int2ByteString((int) getValue()) = 4 bytes big-endian). */
static void window_digest(int type, const uint8_t *p, size_t n, uint8_t *out) {
  if (type == ORACLE_CRC32 || type == ORACLE_CRC32C) {
    uint32_t v = (type == ORACLE_CRC32) ? oracle_crc32(p, n) : oracle_crc32c(p, n);
    out[0] = (uint8_t)(v >> 24); out[1] = (uint8_t)(v >> 16); out[2] = (uint8_t)(v >> 8); out[3] = (uint8_t)v;
  } else if (type == ORACLE_SHA256) {
    oracle_sha256(p, n, out);
  } else {
    oracle_md5(p, n, out);
  }
}

/* Checksum.java:155-177: NONE -> no checksums; otherwise one checksum per bpc-sized
 * slice from offset 0, the last slice possibly shorter
 * (ChunkBufferImplWithByteBuffer.java:62-83). */
int64_t oracle_compute(int type, uint32_t bpc, const uint8_t *data, uint64_t len,
                       uint8_t *sums_out, uint64_t sums_cap) {
  if (type < ORACLE_NONE || type > ORACLE_MD5) return -1;
  if (type == ORACLE_NONE) return 0;
  if (bpc == 0) return -1;
  const size_t ds = oracle_digest_size(type);
  const uint64_t n = (len + bpc - 1) / bpc;
  if (n * ds > sums_cap) return -1;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t off = i * (uint64_t)bpc;
    uint64_t wl = (len - off < bpc) ? len - off : bpc;
    window_digest(type, data + off, (size_t)wl, sums_out + i * ds);
  }
  return (int64_t)n;
}

/* Checksum.java:237-250 then ChecksumData.java:117-149. */
int oracle_verify(int type, uint32_t bpc, const uint8_t *data, uint64_t len,
                  const uint8_t *stored, uint64_t n_stored, uint64_t start_index,
                  int64_t *first_mismatch) {
  if (first_mismatch) *first_mismatch = -1;
  if (type < ORACLE_NONE || type > ORACLE_MD5) return ORACLE_VERIFY_BAD_TYPE;
  if (type == ORACLE_NONE) return ORACLE_VERIFY_OK; /* Checksum.java:241-244 */
  if (bpc == 0) return ORACLE_VERIFY_BAD_TYPE;
  const size_t ds = oracle_digest_size(type);
  const uint64_t n = (len + bpc - 1) / bpc;
  if (n_stored == 0) return ORACLE_VERIFY_ORIG_EMPTY; /* :121-124 */
  if (n == 0) return ORACLE_VERIFY_COMPUTED_EMPTY;    /* :126-129 */
  uint8_t *computed = (uint8_t *)malloc((size_t)(n * ds));
  if (!computed) return ORACLE_VERIFY_BAD_TYPE;
  oracle_compute(type, bpc, data, len, computed, n * ds);
  int rc = ORACLE_VERIFY_OK;
  for (uint64_t i = 0; i < n; i++) { /* :135-141 */
    if (start_index + i >= n_stored) { rc = ORACLE_VERIFY_OUT_OF_RANGE; break; }
    if (memcmp(stored + (start_index + i) * ds, computed + i * ds, ds) != 0) {
      if (first_mismatch) *first_mismatch = (int64_t)i;
      rc = ORACLE_VERIFY_MISMATCH;
      break;
    }
  }
  free(computed);
  return rc;
}

/* ------------------------------------------------------------ test data -- */

void oracle_fill_splitmix64(uint64_t seed, uint8_t *out, uint64_t len) {
  uint64_t x = seed;
  uint64_t i = 0;
  while (i < len) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    for (int k = 0; k < 8 && i < len; k++, i++) out[i] = (uint8_t)(z >> (8 * k));
  }
}

/* ------------------------------------------------------------ benchmark -- */

typedef struct {
  int type; uint32_t bpc; const uint8_t *data; uint64_t len; uint8_t *sums; uint64_t cap; int clmul;
} bench_job;

static void *bench_worker(void *arg) {
  bench_job *j = (bench_job *)arg;
  if (j->clmul) oracle_compute_clmul(j->type, j->bpc, j->data, j->len, j->sums, j->cap);
  else oracle_compute(j->type, j->bpc, j->data, j->len, j->sums, j->cap);
  return NULL;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static double bench_run(int type, uint32_t bpc, const uint8_t *data, uint64_t len, int threads, int reps, int clmul);

double oracle_bench(int type, uint32_t bpc, const uint8_t *data, uint64_t len, int threads, int reps) {
  return bench_run(type, bpc, data, len, threads, reps, 0);
}

double oracle_bench_clmul(int type, uint32_t bpc, const uint8_t *data, uint64_t len, int threads, int reps) {
  if (type != ORACLE_CRC32 && type != ORACLE_CRC32C) return -1.0;
  return bench_run(type, bpc, data, len, threads, reps, 1);
}

static double bench_run(int type, uint32_t bpc, const uint8_t *data, uint64_t len, int threads, int reps, int clmul) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  const size_t ds = oracle_digest_size(type);
  if (ds == 0 || bpc == 0) return -1.0;
  const uint64_t nwin = (len + bpc - 1) / bpc;
  uint8_t *sums = (uint8_t *)malloc((size_t)(nwin * ds + 64));
  bench_job jobs[256];
  pthread_t th[256];
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    double t0 = now_s();
    uint64_t w0 = 0;
    for (int t = 0; t < threads; t++) {
      uint64_t w1 = nwin * (uint64_t)(t + 1) / (uint64_t)threads;
      uint64_t b0 = w0 * bpc, b1 = w1 * bpc;
      if (b1 > len) b1 = len;
      jobs[t] = (bench_job){type, bpc, data + b0, b1 - b0, sums + w0 * ds, (w1 - w0) * ds, clmul};
      w0 = w1;
      if (threads > 1) pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
      else bench_worker(&jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    double dt = now_s() - t0;
    if (dt < best) best = dt;
  }
  free(sums);
  return best;
}
