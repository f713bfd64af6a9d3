// GF(2) arithmetic for reflected 32-bit CRCs (CRC-32 / IEEE 802.3 and CRC-32C / Castagnoli),
// usable from host code, device code and constant expressions.
//
// Representation: a 32-bit register holds a polynomial over GF(2) in the reflected bit order the
// reference uses (ChecksumByteBuffer.java:73-75: crc = (crc >>> 8) ^ T[(crc ^ b) & 0xff]), i.e.
// bit 31 is the coefficient of x^0 and bit 0 the coefficient of x^31.  "Advancing the register
// through n zero bytes" is multiplication by x^(8n) modulo the CRC polynomial; it is linear, which
// is what lets a window be cut into pieces that are summed independently and merged afterwards.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define OZCS_HD __host__ __device__
#else
#define OZCS_HD
#endif

namespace ozcs {

// Reflected polynomials: PureJavaCrc32ByteBuffer.java:32 and PureJavaCrc32CByteBuffer.java:35
// (the table generators named in their class comments).
constexpr uint32_t kPolyCrc32 = 0xEDB88320u;
constexpr uint32_t kPolyCrc32c = 0x82F63B78u;

// a(x) * b(x) mod P(x)
OZCS_HD constexpr uint32_t gf2_mulmod(uint32_t a, uint32_t b, uint32_t poly) {
  uint32_t p = 0;
  for (int i = 0; i < 32; i++) {
    if (a & (0x80000000u >> i)) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ poly : (b >> 1);
  }
  return p;
}

// x^(8n) mod P: the operator "advance through n zero bytes" as a multiplier.
OZCS_HD constexpr uint32_t gf2_xpow8n(uint64_t nbytes, uint32_t poly) {
  uint32_t result = 0x80000000u;          // x^0
  uint32_t sq = 0x00800000u;              // x^8
  while (nbytes) {
    if (nbytes & 1) result = gf2_mulmod(result, sq, poly);
    sq = gf2_mulmod(sq, sq, poly);
    nbytes >>= 1;
  }
  return result;
}

// One byte of input, bit by bit (the definition the tables of the reference are generated from).
OZCS_HD constexpr uint32_t crc_byte_bitwise(uint32_t crc, uint32_t byte, uint32_t poly) {
  crc ^= byte;
  for (int k = 0; k < 8; k++) crc = (crc & 1u) ? (crc >> 1) ^ poly : (crc >> 1);
  return crc;
}

// The image of each register bit under "advance n bytes": basis[i] = (1u << i) * x^(8n) mod P.
// Every lookup table the kernels use is an XOR-closure of such a basis.
struct OpBasis {
  uint32_t v[32];
};

constexpr OpBasis make_op_basis(uint64_t nbytes, uint32_t poly) {
  OpBasis b{};
  const uint32_t m = gf2_xpow8n(nbytes, poly);
  for (int i = 0; i < 32; i++) b.v[i] = gf2_mulmod(1u << i, m, poly);
  return b;
}

// Operators the kernels need, as byte counts.  Index k <-> advance by kOpBytes[k].
//   0..3   : 4, 8, 12, 16          merge the four 32-bit accumulators of a 16-byte unit
//   3..8   : 16, 32, ... 512       merge the 64 lanes of a wavefront (one 1 KiB row)
//   9      : 1024                  merge the two rows a wavefront holds
//   10..13 : 2048 ... 16384        merge up to 16 wavefronts (one stripe)
//   10..14 : 2048 ... 32768        the streaming operator: advance by one whole stripe
constexpr int kNumOps = 15;
constexpr uint32_t kOpBytes[kNumOps] = {4, 8, 12, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
constexpr int kOpZ4 = 0, kOpZ8 = 1, kOpZ12 = 2, kOpZ16 = 3, kOpZ1024 = 9, kOpZ2048 = 10;

struct PolyBasis {
  OpBasis op[kNumOps];
};

constexpr PolyBasis make_poly_basis(uint32_t poly) {
  PolyBasis pb{};
  for (int k = 0; k < kNumOps; k++) pb.op[k] = make_op_basis(kOpBytes[k], poly);
  return pb;
}

}  // namespace ozcs
