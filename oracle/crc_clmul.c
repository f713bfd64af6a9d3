/*
 * CRC-32 / CRC-32C by carry-less-multiply folding on x86 (PCLMULQDQ), for the CPU-baseline leg of the
 * benchmark only.
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/.  Why it exists: the reference's production CRC
 * engines are java.util.zip.CRC32 / CRC32C (ChecksumByteBufferFactory.java:74-89), which HotSpot runs as
 * intrinsics built on this very instruction; no JVM exists on the build or GPU machines, so the closest
 * honest stand-in for "the reference's CPU path on these cores" is the same folding scheme in C.  The
 * slice-by-8 restatement in ozcs_oracle.c stays the parity oracle; this file is checked against it.
 *
 * Method (the folding of Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ"):
 * a 128-bit register X stands for "the message so far"; taking in the next 128 bits D is
 *     X' = X * x^128 + D  (mod P),   X * x^128 = lo(X) * (x^192 mod P) + hi(X) * (x^128 mod P)
 * and both products fit 128 bits, so one step is two PCLMULQDQ and two XORs.  Four registers 64 bytes apart
 * keep the multiplier busy.  In the reflected bit order CRC-32 uses, a 64x64 carry-less product lands one
 * bit low in the 128-bit frame, so the constants are x^(D-1+64) and x^(D-1) for a fold distance of D bits.
 * They are computed at start-up from the polynomial, not copied from anywhere.  The last 16 bytes (the
 * register) and the tail go through the table method from state 0.
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "ozcs_oracle.h"

#if defined(__x86_64__)
#include <immintrin.h>

typedef struct {
  uint64_t k512_lo, k512_hi; /* fold distance 512 bits: low qword * x^575, high qword * x^511 */
  uint64_t k128_lo, k128_hi; /* fold distance 128 bits: x^191, x^127 */
  uint32_t table[2048];
} clmul_ctx;

static clmul_ctx g_ctx[2];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* x^n mod P in the reflected register form (bit 31 = coefficient of x^0). */
static uint32_t xpow(unsigned n, uint32_t poly) {
  uint32_t v = 0x80000000u;
  for (unsigned i = 0; i < n; i++) v = (v & 1u) ? (v >> 1) ^ poly : (v >> 1);
  return v;
}

static void init_ctx(void) {
  const uint32_t polys[2] = {ORACLE_POLY_CRC32, ORACLE_POLY_CRC32C};
  for (int i = 0; i < 2; i++) {
    /* a degree-<32 constant sits in the top half of a 64-bit reflected operand */
    g_ctx[i].k512_lo = (uint64_t)xpow(575, polys[i]) << 32;
    g_ctx[i].k512_hi = (uint64_t)xpow(511, polys[i]) << 32;
    g_ctx[i].k128_lo = (uint64_t)xpow(191, polys[i]) << 32;
    g_ctx[i].k128_hi = (uint64_t)xpow(127, polys[i]) << 32;
    oracle_crc_make_table(polys[i], g_ctx[i].table);
  }
}

__attribute__((target("pclmul,sse4.1"))) static inline __m128i fold(__m128i x, __m128i k, __m128i d) {
  /* k = {lo constant, hi constant}: lo(x)*k.lo ^ hi(x)*k.hi ^ d */
  return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), d);
}

__attribute__((target("pclmul,sse4.1"))) static uint32_t update_clmul(const clmul_ctx *c, uint32_t crc,
                                                                       const uint8_t *p, size_t len) {
  if (len < 64) return oracle_crc_update(crc, p, len, c->table);
  const __m128i k512 = _mm_set_epi64x((long long)c->k512_hi, (long long)c->k512_lo);
  const __m128i k128 = _mm_set_epi64x((long long)c->k128_hi, (long long)c->k128_lo);
  __m128i x0 = _mm_loadu_si128((const __m128i *)(p + 0));
  __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 16));
  __m128i x2 = _mm_loadu_si128((const __m128i *)(p + 32));
  __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 48));
  x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)crc)); /* the running register goes onto the first four bytes */
  p += 64;
  len -= 64;
  while (len >= 64) {
    x0 = fold(x0, k512, _mm_loadu_si128((const __m128i *)(p + 0)));
    x1 = fold(x1, k512, _mm_loadu_si128((const __m128i *)(p + 16)));
    x2 = fold(x2, k512, _mm_loadu_si128((const __m128i *)(p + 32)));
    x3 = fold(x3, k512, _mm_loadu_si128((const __m128i *)(p + 48)));
    p += 64;
    len -= 64;
  }
  __m128i x = fold(x0, k128, x1);
  x = fold(x, k128, x2);
  x = fold(x, k128, x3);
  while (len >= 16) {
    x = fold(x, k128, _mm_loadu_si128((const __m128i *)p));
    p += 16;
    len -= 16;
  }
  uint8_t last[16];
  _mm_storeu_si128((__m128i *)last, x);
  uint32_t s = oracle_crc_update(0, last, 16, c->table);
  return oracle_crc_update(s, p, len, c->table);
}

int oracle_has_clmul(void) { return __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); }

uint32_t oracle_crc_clmul(int type, const uint8_t *data, size_t len) {
  pthread_once(&g_once, init_ctx);
  const clmul_ctx *c = &g_ctx[type == ORACLE_CRC32C ? 1 : 0];
  if (!oracle_has_clmul()) return ~oracle_crc_update(0xffffffffu, data, len, c->table);
  return ~update_clmul(c, 0xffffffffu, data, len);
}

#else /* not x86-64: the table method */

int oracle_has_clmul(void) { return 0; }

uint32_t oracle_crc_clmul(int type, const uint8_t *data, size_t len) {
  return type == ORACLE_CRC32C ? oracle_crc32c(data, len) : oracle_crc32(data, len);
}

#endif

/* One digest per bpc window, 4 big-endian bytes each (Checksum.java:62-64, 169-175); returns the count. */
int64_t oracle_compute_clmul(int type, uint32_t bpc, const uint8_t *data, uint64_t len, uint8_t *out, uint64_t cap) {
  if ((type != ORACLE_CRC32 && type != ORACLE_CRC32C) || bpc == 0) return -1;
  const uint64_t n = (len + bpc - 1) / bpc;
  if (cap < n * 4) return -1;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t off = i * bpc, l = len - off < bpc ? len - off : bpc;
    const uint32_t v = oracle_crc_clmul(type, data + off, (size_t)l);
    out[4 * i] = (uint8_t)(v >> 24);
    out[4 * i + 1] = (uint8_t)(v >> 16);
    out[4 * i + 2] = (uint8_t)(v >> 8);
    out[4 * i + 3] = (uint8_t)v;
  }
  return (int64_t)n;
}
