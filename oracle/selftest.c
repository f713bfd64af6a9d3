/*
 * Self-test of the CPU oracle, written to be run under AddressSanitizer and UndefinedBehaviorSanitizer
 * (`make -C oracle asan`, run by tests/test_sanitizers.py): every buffer handed to the oracle is an exact-size heap
 * allocation, so that a read or write one byte outside it is caught, and the cases walk all the internal
 * boundaries -- the 8-byte groups and the byte tail of the slice-by-8 loop (ChecksumByteBuffer.java:82-120), the
 * 64-byte blocks and padding of SHA-256 / MD5, the 16/64-byte folding steps of the carry-less-multiply engine, ragged
 * last windows, unaligned starts.  Checked: published known answers, agreement between the two CRC engines, the
 * verify outcomes, and the multi-threaded timing entry points.  Test infrastructure only.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ozcs_oracle.h"

static int g_fail;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)

/* An allocation of exactly n bytes (n == 0: one byte, never touched). */
static uint8_t* exact(size_t n) {
  uint8_t* p = (uint8_t*)malloc(n ? n : 1);
  if (!p) abort();
  return p;
}

static void hex(const uint8_t* d, size_t n, char* out) {
  for (size_t i = 0; i < n; i++) sprintf(out + 2 * i, "%02x", d[i]);
}

int main(void) {
  /* ---- known answers ---- */
  const uint8_t* check = (const uint8_t*)"123456789";
  CHECK(oracle_crc32(check, 9) == 0xCBF43926u);
  CHECK(oracle_crc32c(check, 9) == 0xE3069283u);
  CHECK(oracle_crc_clmul(2, check, 9) == 0xCBF43926u);
  CHECK(oracle_crc_clmul(3, check, 9) == 0xE3069283u);
  {
    uint8_t z[32];
    memset(z, 0, 32);
    CHECK(oracle_crc32c(z, 32) == 0x8A9136AAu); /* RFC 3720 B.4 */
    memset(z, 0xFF, 32);
    CHECK(oracle_crc32c(z, 32) == 0x62A8AB43u);
    uint8_t d[32];
    char h[65];
    oracle_sha256((const uint8_t*)"abc", 3, d);
    hex(d, 32, h);
    CHECK(!strcmp(h, "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"));
    oracle_sha256((const uint8_t*)"", 0, d);
    hex(d, 32, h);
    CHECK(!strcmp(h, "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"));
    oracle_md5((const uint8_t*)"abc", 3, d);
    hex(d, 16, h);
    CHECK(!strcmp(h, "900150983cd24fb0d6963f7d28e17f72"));
    oracle_md5((const uint8_t*)"", 0, d);
    hex(d, 16, h);
    CHECK(!strcmp(h, "d41d8cd98f00b204e9800998ecf8427e"));
  }
  /* ---- tables: 2048 entries, T8_k from T8_(k-1) ---- */
  {
    uint32_t* t = (uint32_t*)exact(2048 * sizeof(uint32_t));
    oracle_crc_make_table(0xEDB88320u, t);
    CHECK(t[1] == 0x77073096u && t[255] == 0x2D02EF8Du);
    for (int k = 1; k < 8; k++)
      for (int i = 0; i < 256; i += 17) CHECK(t[k * 256 + i] == ((t[(k - 1) * 256 + i] >> 8) ^ t[t[(k - 1) * 256 + i] & 0xff]));
    free(t);
  }
  /* ---- every length 0..300 and a few large ones, at offsets 0..3 of an exact-size buffer: both CRC engines agree,
   *      incremental update equals one-shot, digests of split buffers are stable ---- */
  {
    uint32_t* t32 = (uint32_t*)exact(2048 * sizeof(uint32_t));
    oracle_crc_make_table(0xEDB88320u, t32);
    const size_t lens[] = {511, 512, 513, 1023, 4096, 4097, 65535, 65536 + 7, 1 << 20};
    for (size_t li = 0; li < 301 + sizeof lens / sizeof *lens; li++) {
      const size_t n = li < 301 ? li : lens[li - 301];
      for (size_t off = 0; off < 4; off++) {
        uint8_t* b = exact(n + off);
        oracle_fill_splitmix64(1000 + n, b, n + off);
        const uint32_t a = oracle_crc32(b + off, n), c = oracle_crc32c(b + off, n);
        CHECK(oracle_crc_clmul(2, b + off, n) == a);
        CHECK(oracle_crc_clmul(3, b + off, n) == c);
        if (n) { /* update(bytes, 0, i) then the rest: TestChecksumByteBuffer.java:87-91 */
          const size_t cut = n / 3;
          uint32_t r = oracle_crc_update(0xFFFFFFFFu, b + off, cut, t32);
          r = oracle_crc_update(r, b + off + cut, n - cut, t32);
          CHECK((~r) == a);
        }
        uint8_t d1[32], d2[32];
        oracle_sha256(b + off, n, d1);
        uint8_t* copy = exact(n);
        memcpy(copy, b + off, n);
        oracle_sha256(copy, n, d2);
        CHECK(!memcmp(d1, d2, 32));
        oracle_md5(b + off, n, d1);
        oracle_md5(copy, n, d2);
        CHECK(!memcmp(d1, d2, 16));
        free(copy);
        free(b);
      }
    }
    free(t32);
  }
  /* ---- windowing: exact-size outputs, ragged last windows, the reference tests' shapes; verify outcomes ---- */
  {
    const struct { uint64_t len; uint32_t bpc; } shapes[] = {{55, 10}, {100, 20}, {3072, 2048}, {4u << 20, 1u << 20}, {1, 1},
                                                             {17, 16}, {65, 64}, {70000, 16384}, {0, 10}, {300, 7}};
    for (size_t s = 0; s < sizeof shapes / sizeof *shapes; s++) {
      const uint64_t len = shapes[s].len;
      const uint32_t bpc = shapes[s].bpc;
      const uint64_t nwin = (len + bpc - 1) / bpc;
      uint8_t* data = exact(len);
      oracle_fill_splitmix64(7 + s, data, len);
      for (int type = 2; type <= 5; type++) {
        const size_t ds = oracle_digest_size(type);
        uint8_t* sums = exact(nwin * ds);
        CHECK(oracle_compute(type, bpc, data, len, sums, nwin * ds) == (int64_t)nwin);
        if (nwin) CHECK(oracle_compute(type, bpc, data, len, sums, nwin * ds - 1) == -1); /* short output: refused, not overrun */
        if (type <= 3) {
          uint8_t* s2 = exact(nwin * ds);
          CHECK(oracle_compute_clmul(type, bpc, data, len, s2, nwin * ds) == (int64_t)nwin);
          CHECK(!memcmp(sums, s2, nwin * ds));
          free(s2);
        }
        int64_t first = 99;
        if (nwin) {
          CHECK(oracle_verify(type, bpc, data, len, sums, nwin, 0, &first) == ORACLE_VERIFY_OK && first == -1);
          data[len - 1] ^= 1; /* last window */
          CHECK(oracle_verify(type, bpc, data, len, sums, nwin, 0, &first) == ORACLE_VERIFY_MISMATCH && first == (int64_t)nwin - 1);
          data[len - 1] ^= 1;
          CHECK(oracle_verify(type, bpc, data, len, sums, nwin - 1, 0, &first) ==
                (nwin > 1 ? ORACLE_VERIFY_OUT_OF_RANGE : ORACLE_VERIFY_ORIG_EMPTY));
          if (nwin > 1) /* windows 1.. against the stored list from index 1 */
            CHECK(oracle_verify(type, bpc, data + bpc, len - bpc, sums, nwin, 1, &first) == ORACLE_VERIFY_OK);
        } else {
          CHECK(oracle_verify(type, bpc, data, len, sums, 1, 0, &first) == ORACLE_VERIFY_COMPUTED_EMPTY);
        }
        free(sums);
      }
      CHECK(oracle_compute(1, bpc, data, len, NULL, 0) == 0);  /* NONE */
      CHECK(oracle_compute(9, bpc, data, len, NULL, 0) == -1); /* unknown type */
      free(data);
    }
  }
  /* ---- the timing entry points on threads (bounded: 8 MiB, 3 threads) ---- */
  {
    const uint64_t len = 8u << 20;
    uint8_t* data = exact(len);
    oracle_fill_splitmix64(99, data, len);
    CHECK(oracle_bench(2, 1u << 20, data, len, 3, 1) > 0);
    CHECK(oracle_bench_clmul(3, 16384, data, len, 3, 1) > 0);
    CHECK(oracle_bench(4, 1u << 20, data, len, 3, 1) > 0);
    free(data);
  }
  printf(g_fail ? "oracle selftest: %d failure(s)\n" : "oracle selftest ok\n", g_fail);
  return g_fail ? 1 : 0;
}
