// The GF(2) arithmetic the CRC kernels are built on (hadoop-ozone_amd/csrc/gf2_crc.h), checked on the CPU against the
// bit-serial definition of the two CRCs: check values, operator algebra, the operator bases the LDS tables are
// generated from, and the two merge identities the kernels use (pieces of a window summed independently; a window
// entered in the middle and wrapped round).  No GPU, no library.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "../../hadoop-ozone_amd/csrc/gf2_crc.h"

using namespace ozcs;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

// register after the bytes, starting from `state` (no initial or final inversion)
static uint32_t run(uint32_t state, const uint8_t* p, size_t n, uint32_t poly) {
  for (size_t i = 0; i < n; i++) state = crc_byte_bitwise(state, p[i], poly);
  return state;
}
static uint32_t apply(const OpBasis& b, uint32_t x) {
  uint32_t r = 0;
  for (int i = 0; i < 32; i++)
    if (x & (1u << i)) r ^= b.v[i];
  return r;
}

int main() {
  const uint8_t check[] = "123456789";
  EXPECT((~run(0xFFFFFFFFu, check, 9, kPolyCrc32)) == 0xCBF43926u);    // CRC-32 check value
  EXPECT((~run(0xFFFFFFFFu, check, 9, kPolyCrc32c)) == 0xE3069283u);   // CRC-32C check value
  std::mt19937 rng(7);
  for (uint32_t poly : {kPolyCrc32, kPolyCrc32c}) {
    // x^(8a) * x^(8b) = x^(8(a+b)); multiplication commutes
    for (int t = 0; t < 50; t++) {
      const uint64_t a = rng() % 100000, b = rng() % 100000;
      EXPECT(gf2_mulmod(gf2_xpow8n(a, poly), gf2_xpow8n(b, poly), poly) == gf2_xpow8n(a + b, poly));
      const uint32_t x = rng(), y = rng();
      EXPECT(gf2_mulmod(x, y, poly) == gf2_mulmod(y, x, poly));
    }
    // an operator basis advances the register through n zero bytes, for every operator the kernels tabulate
    const PolyBasis pb = make_poly_basis(poly);
    for (int k = 0; k < kNumOps; k++) {
      std::vector<uint8_t> zeros(kOpBytes[k], 0);
      for (int t = 0; t < 4; t++) {
        const uint32_t x = rng();
        EXPECT(apply(pb.op[k], x) == run(x, zeros.data(), zeros.size(), poly));
        EXPECT(apply(pb.op[k], x) == gf2_mulmod(x, gf2_xpow8n(kOpBytes[k], poly), poly));
      }
    }
    // pieces summed independently: state(A | B) = state(A) * x^(8|B|) ^ state0(B)
    std::vector<uint8_t> w(16384);
    for (auto& c : w) c = (uint8_t)rng();
    const uint32_t whole = run(0xFFFFFFFFu, w.data(), w.size(), poly);
    for (size_t cut : {1ul, 4ul, 15ul, 16ul, 1024ul, 4096ul, 8192ul, 12288ul, 16383ul}) {
      const uint32_t head = run(0xFFFFFFFFu, w.data(), cut, poly);
      const uint32_t tail = run(0u, w.data() + cut, w.size() - cut, poly);
      EXPECT((gf2_mulmod(head, gf2_xpow8n(w.size() - cut, poly), poly) ^ tail) == whole);
    }
    // the initial register folded into the first four bytes: state(init, W) = state(0, W with init XORed onto bytes 0..3)
    {
      std::vector<uint8_t> v = w;
      for (int i = 0; i < 4; i++) v[i] ^= 0xFF;
      EXPECT(run(0u, v.data(), v.size(), poly) == whole);
    }
    // a window entered at 1/4, 1/2, 3/4 and wrapped round: tail first, then the head on a register of its own, the
    // head advanced through the tail's length by the power-of-two operators of bpc/2 and bpc/4 bytes
    const int half = __builtin_ctz((unsigned)w.size()) - 2, quarter = half - 1;   // operator indices, as in the kernel
    EXPECT(kOpBytes[half] == w.size() / 2 && kOpBytes[quarter] == w.size() / 4);
    for (int p = 1; p < 4; p++) {
      const size_t phase = p * w.size() / 4, tlen = w.size() - phase;
      const uint32_t tail = run(0u, w.data() + phase, tlen, poly);
      uint32_t head = run(0xFFFFFFFFu, w.data(), phase, poly);
      if (tlen & (w.size() >> 1)) head = apply(pb.op[half], head);
      if (tlen & (w.size() >> 2)) head = apply(pb.op[quarter], head);
      EXPECT((head ^ tail) == whole);
    }
  }
  std::printf(failures ? "%d FAILURES\n" : "gf2: all passed\n", failures);
  return failures ? 1 : 0;
}
