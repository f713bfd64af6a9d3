// CRC32 / CRC32C per-window kernels for gfx950 (MI355X).
//
// What is computed: for every bytes-per-checksum window of every chunk, the value the reference
// produces with reset(); update(window); getValue() (Checksum.java:66-74 over the engines of
// ChecksumByteBufferFactory.java:74-89), stored as 4 big-endian bytes (Checksum.java:62-64), in
// window order (Checksum.java:172-176).
//
// How: CDNA4 has no carry-less multiply, so the byte-serial table walk of
// ChecksumByteBuffer.java:82-120 is replaced by its linear-algebra equivalent.
//   * A window is cut into stripes of 2*G KiB: G wavefronts, each covering two 1 KiB rows
//     (64 lanes x 16 bytes).  Stripes are aligned to the END of the window, so that every stripe
//     but the first is full and every 128-bit load is naturally aligned; bytes in front of the
//     window are masked to zero (leading zeros do not change a CRC whose register starts at 0),
//     and the 0xFFFFFFFF initial register is folded into the first four bytes of the window.
//   * Each lane keeps eight 32-bit accumulators, one per dword of its two 16-byte units, and per
//     stripe does  acc = (acc * x^(8*stripe) mod P) ^ dword.  The multiplication is four
//     byte-indexed lookups.  The four tables are held in LDS in 32 interleaved copies (copy r of
//     every entry sits in bank r), so lane l always reads bank l%32: the lookups are conflict-free
//     whatever the data is, at one LDS lookup per input byte.  (The kernels with several lanes per
//     window get the same from 16 copies and keep a second table set in the room: kSmallSetBytes.)
//     The two rows give every wavefront two independent dependency chains, which keeps the LDS
//     pipeline fed.
//   * At the end of a window the accumulators are one stripe-sized message with the same CRC
//     as the window.  It is folded with fixed shift operators: 4 slots -> unit, 64 lanes -> row
//     (reduction tree over DPP row shifts), 2 rows -> wavefront, G wavefronts -> window (through LDS).
//     These operators use 16-entry nibble tables (conflict-free without copies: 16 consecutive
//     banks).
//   * The at most 15 bytes behind the last aligned 16-byte boundary are finished bit-serially.
//
// Three kernels share that inner loop:
//   crc_windows_kernel  G wavefronts per window (1..16), for windows above 256 KiB or scarce windows;
//   crc_subwave_kernel  4..32 lanes per window, any alignment and ragged last windows;
//   crc_flat_kernel     the same for the gapless layout (full, aligned windows back to back), where the
//                       per-window geometry disappears; optionally with the wavefronts entering their windows
//                       at four different offsets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "crc_kernels.h"
#include "gf2_crc.h"

namespace ozcs {

__constant__ PolyBasis c_basis[2] = {make_poly_basis(kPolyCrc32), make_poly_basis(kPolyCrc32c)};

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kWaveBytes = 2048;  // bytes of a stripe one wavefront covers: two 1 KiB rows

// LDS layout in bytes, crc_windows_kernel
constexpr uint32_t kMainBytes = 4u * 256u * 32u * 4u;          // 128 KiB: 4 byte tables x 32 copies
constexpr int kNibOps = 14;                                    // operators 0..13 of kOpBytes
constexpr uint32_t kNibOpBytes = 8u * 16u * 4u;                // 512 B per operator
constexpr uint32_t kNibOff = kMainBytes;
static_assert((kNibOff & 0xffu) == 0, "nib_apply takes bytes 1..2 of the table area's address");
constexpr uint32_t kPartOff = kNibOff + kNibOps * kNibOpBytes;  // 2 x 16 partial registers
constexpr uint32_t kLdsBytes = kPartOff + 2u * kWaves * 4u;
static_assert(kLdsBytes <= 160u * 1024u, "one workgroup's tables must fit the CU's 160 KiB of LDS");

// LDS layout of the kernels with several lanes per window (crc_subwave_kernel, crc_desc_kernel, crc_flat_kernel).  Two sets
// of four byte tables, each in 16 copies: the stripe operator and the "advance 4 bytes" operator that merges the four
// accumulators of a unit at the end of every window.  Entry b of table t, copy r, stands at b * 256 + t * 64 + r * 4, so
// table t occupies banks 16 * (t & 1) + r.  Sixteen copies are enough for conflict-free lookups because the four lookups
// of an accumulator are XORed and their order is therefore free: lanes 0-15 of each half of the wavefront look up
// (table 0, byte 0) while lanes 16-31 look up (table 1, byte 1), and the next instruction has them the other way round;
// likewise tables 2 and 3 (SmallLanes).  With one copy of the second set, as it was at first, random bytes met about
// 3.5 deep in a bank and a third of the LDS cycles at 512-byte windows were conflict cycles; with 16 copies there are
// none (MI355X, 256-byte windows: 5.63 -> 6.15 TB/s; 512 B: 6.48 -> 6.61).
constexpr uint32_t kSmallSetBytes = 4u * 256u * 16u * 4u;  // 64 KiB
constexpr uint32_t kSmallZ4Off = kSmallSetBytes;
static_assert((kSmallZ4Off & 0xffffu) == 0, "the second set's offset is byte 2 of a lane's address constant");
constexpr uint32_t kSmallNibOff = kSmallZ4Off + kSmallSetBytes;
static_assert((kSmallNibOff & 0xffu) == 0, "nib_apply takes bytes 1..2 of the table area's address");
constexpr uint32_t kSmallLdsBytes = kSmallNibOff + kNibOps * kNibOpBytes;
static_assert(kSmallLdsBytes <= 160u * 1024u, "one workgroup's tables must fit the CU's 160 KiB of LDS");

__device__ __forceinline__ uint32_t lds_u32(const char* base, uint32_t off) {
  return *reinterpret_cast<const uint32_t*>(base + off);
}

// a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// x * x^(8*n) mod P through the 16-entry nibble tables of operator `op` (an index into kOpBytes).
// Index extraction: the eight nibbles, times four, are masked into the bytes of two registers and every
// LDS address is then ONE v_perm_b32 (byte 0 <- that byte, bytes 1..2 <- the table area's base); the
// operator's and the nibble's offsets ride in the instruction's immediate offset field.
template <uint32_t kArea = kNibOff>
__device__ __forceinline__ uint32_t nib_apply(const char* lds, int op, uint32_t x) {
  const uint32_t m0 = (x << 2) & 0x3c3c3c3cu;  // nibbles 0, 2, 4, 6
  const uint32_t m1 = (x >> 2) & 0x3c3c3c3cu;  // nibbles 1, 3, 5, 7
  const uint32_t off = (uint32_t)op * kNibOpBytes;
  uint32_t t[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t[2 * k] = lds_u32(lds, __builtin_amdgcn_perm(m0, kArea, 0x0c020104u + k) + (off + (2 * k) * 64u));
    t[2 * k + 1] = lds_u32(lds, __builtin_amdgcn_perm(m1, kArea, 0x0c020104u + k) + (off + (2 * k + 1) * 64u));
  }
  return xor3(xor3(t[0], t[1], t[2]), xor3(t[3], t[4], t[5]), t[6] ^ t[7]);
}

// The same with the operator chosen at run time: `area` = kNibOff + op * kNibOpBytes (a multiple of 256, uniform
// over the wavefront) takes the table area's place in the address.
__device__ __forceinline__ uint32_t nib_apply_at(const char* lds, uint32_t area, uint32_t x) {
  const uint32_t m0 = (x << 2) & 0x3c3c3c3cu;
  const uint32_t m1 = (x >> 2) & 0x3c3c3c3cu;
  uint32_t t[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t[2 * k] = lds_u32(lds, __builtin_amdgcn_perm(m0, area, 0x0c020104u + k) + (2 * k) * 64u);
    t[2 * k + 1] = lds_u32(lds, __builtin_amdgcn_perm(m1, area, 0x0c020104u + k) + (2 * k + 1) * 64u);
  }
  return xor3(xor3(t[0], t[1], t[2]), xor3(t[3], t[4], t[5]), t[6] ^ t[7]);
}

// Value of the lane N places lower in the same row of 16 lanes (0 where there is none): one DPP move.
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + N, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t lanes_up(uint32_t x, int level) {
  switch (level) {
    case 0: return row_shr<1>(x);
    case 1: return row_shr<2>(x);
    case 2: return row_shr<4>(x);
    case 3: return row_shr<8>(x);
    default: return __shfl_up(x, 1u << level, 64);
  }
}

struct Acc4 {
  uint32_t a0, a1, a2, a3;
};


// LDS addresses of the four table entries an accumulator selects.  Tables 0/1 share rows of 256 B
// (entry b of table t, copy r at b*256 + t*128 + r*4) and so do tables 2/3, 64 KiB further on; with
// byte-aligned fields the whole address is ONE v_perm_b32: byte 1 <- the selected byte of x,
// bytes 0 and 2 <- the lane's constant (copy offset; 0 or 1 for the table pair).
struct Rows4 {
  uint32_t o0, o1, o2, o3;
};
__device__ __forceinline__ Rows4 rows_of(uint32_t x, uint32_t lb01, uint32_t lb23) {
  Rows4 r;
  r.o0 = __builtin_amdgcn_perm(x, lb01, 0x0c020400u);
  r.o1 = __builtin_amdgcn_perm(x, lb01, 0x0c020500u);
  r.o2 = __builtin_amdgcn_perm(x, lb23, 0x0c020600u);
  r.o3 = __builtin_amdgcn_perm(x, lb23, 0x0c020700u);
  return r;
}
__device__ __forceinline__ uint32_t gather4(const char* lds, const Rows4& r, uint32_t d) {
  const uint32_t t0 = lds_u32(lds, r.o0);
  const uint32_t t1 = lds_u32(lds, r.o1 + 128u);
  const uint32_t t2 = lds_u32(lds, r.o2);
  const uint32_t t3 = lds_u32(lds, r.o3 + 128u);
  return xor3(xor3(t0, t1, t2), t3, d);
}

// One stripe step for the four accumulators of a 16-byte unit: acc = acc * x^(8*stripe) ^ dword.
// The sixteen lookups are independent of each other.
__device__ __forceinline__ void absorb(const char* lds, uint32_t lb01, uint32_t lb23, Acc4& a, const uint4 d) {
  const Rows4 r0 = rows_of(a.a0, lb01, lb23), r1 = rows_of(a.a1, lb01, lb23), r2 = rows_of(a.a2, lb01, lb23),
              r3 = rows_of(a.a3, lb01, lb23);
  a.a0 = gather4(lds, r0, d.x);
  a.a1 = gather4(lds, r1, d.y);
  a.a2 = gather4(lds, r2, d.z);
  a.a3 = gather4(lds, r3, d.w);
}

// Lane constants and lookups of the 16-copy layout (kSmallSetBytes).  An address is still ONE v_perm_b32 -- byte 0 <- the
// lane's constant (table and copy), byte 1 <- the selected byte of x, byte 2 <- the set -- but which byte of x a lane
// selects in which instruction is now the lane's own, so the selector is a register too.
struct SmallLanes {
  uint32_t ca, cb;                  // table and copy, stripe set: first and second instruction of a pair
  uint32_t caz, cbz;                // the same in the "advance 4 bytes" set
  uint32_t sa01, sb01, sa23, sb23;  // v_perm selectors: bytes 0/1 and 2/3 of x, in the lane's order
  __device__ __forceinline__ explicit SmallLanes(int lane) {
    const uint32_t r4 = (uint32_t)(lane & 15) * 4u;
    const bool hi = (lane & 16) != 0;
    ca = r4 + (hi ? 64u : 0u);  // first instruction: lanes 0-15 the even table (banks 0-15), lanes 16-31 the odd one (16-31)
    cb = r4 + (hi ? 0u : 64u);  // second instruction: the other way round
    caz = ca + kSmallZ4Off;
    cbz = cb + kSmallZ4Off;
    sa01 = hi ? 0x0c020500u : 0x0c020400u;
    sb01 = hi ? 0x0c020400u : 0x0c020500u;
    sa23 = hi ? 0x0c020700u : 0x0c020600u;
    sb23 = hi ? 0x0c020600u : 0x0c020700u;
  }
  // x * (the set's operator) ^ d
  __device__ __forceinline__ uint32_t apply(const char* lds, uint32_t a_, uint32_t b_, uint32_t x, uint32_t d) const {
    const uint32_t t0 = lds_u32(lds, __builtin_amdgcn_perm(x, a_, sa01));
    const uint32_t t1 = lds_u32(lds, __builtin_amdgcn_perm(x, b_, sb01));
    const uint32_t t2 = lds_u32(lds, __builtin_amdgcn_perm(x, a_, sa23) + 128u);
    const uint32_t t3 = lds_u32(lds, __builtin_amdgcn_perm(x, b_, sb23) + 128u);
    return xor3(xor3(t0, t1, t2), t3, d);
  }
  // one stripe step for the four accumulators of a 16-byte unit (sixteen independent lookups)
  __device__ __forceinline__ void absorb(const char* lds, Acc4& a, const uint4 d) const {
    const uint32_t n0 = apply(lds, ca, cb, a.a0, d.x), n1 = apply(lds, ca, cb, a.a1, d.y),
                   n2 = apply(lds, ca, cb, a.a2, d.z), n3 = apply(lds, ca, cb, a.a3, d.w);
    a = {n0, n1, n2, n3};
  }
  // The four accumulators of a unit -> the register after the unit, as a Horner chain in z = x^32:
  // (((a0 z + a1) z + a2) z + a3) z; the chains of the two rows a wavefront holds interleave.
  __device__ __forceinline__ void merge_units(const char* lds, const Acc4& a, const Acc4& b, uint32_t& x, uint32_t& y) const {
    x = apply(lds, caz, cbz, a.a0, a.a1);
    y = apply(lds, caz, cbz, b.a0, b.a1);
    x = apply(lds, caz, cbz, x, a.a2);
    y = apply(lds, caz, cbz, y, b.a2);
    x = apply(lds, caz, cbz, x, a.a3);
    y = apply(lds, caz, cbz, y, b.a3);
    x = apply(lds, caz, cbz, x, 0u);
    y = apply(lds, caz, cbz, y, 0u);
  }
};

// 128-bit load through the global (not flat) address space: flat loads would tie the LDS and the
// memory wait counters together.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;
// The data is read exactly once, so it is loaded non-temporally: measured on MI355X, a pure streaming
// read in this kernel's access pattern reaches 7.0 TB/s with nt loads against 6.2 TB/s without.
__device__ __forceinline__ uint4 ld16(uintptr_t p) {
#ifdef OZCS_TUNE_NOLOAD  // timing-only build: no memory traffic, results are wrong on purpose
  return make_uint4((uint32_t)p, (uint32_t)(p >> 7), (uint32_t)(p >> 13), (uint32_t)(p >> 3));
#endif
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<gptr_u4>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Mask of the bytes of the little-endian dword at address p whose address is >= s.
__device__ __forceinline__ uint32_t keep_from(uintptr_t p, uintptr_t s) {
  const int64_t off = (int64_t)(s - p);
  if (off <= 0) return 0xFFFFFFFFu;
  if (off >= 4) return 0u;
  return 0xFFFFFFFFu << (8 * (int)off);
}

// The 16-byte unit at u of a window that starts at s (u + 16 > s is not required): bytes in front
// of s are dropped, and the 0xFFFFFFFF initial register is XORed onto bytes s .. s+3.
__device__ __forceinline__ Acc4 first_unit(uintptr_t u, uintptr_t s) {
  Acc4 a = {0, 0, 0, 0};
  if (u + 16 > s) {
    const uint4 d = ld16(u);
    const uint32_t k0 = keep_from(u, s), k1 = keep_from(u + 4, s), k2 = keep_from(u + 8, s), k3 = keep_from(u + 12, s);
    a.a0 = (d.x & k0) ^ (k0 & ~keep_from(u, s + 4));
    a.a1 = (d.y & k1) ^ (k1 & ~keep_from(u + 4, s + 4));
    a.a2 = (d.z & k2) ^ (k2 & ~keep_from(u + 8, s + 4));
    a.a3 = (d.w & k3) ^ (k3 & ~keep_from(u + 12, s + 4));
  }
  return a;
}

// What a unit at u >= s still owes the folded initial register (non-zero only when s + 4 > u).
__device__ __forceinline__ uint4 init_spill(uintptr_t u, uintptr_t s) {
  return make_uint4(~keep_from(u, s + 4), ~keep_from(u + 4, s + 4), ~keep_from(u + 8, s + 4), ~keep_from(u + 12, s + 4));
}
__device__ __forceinline__ uint4 xor4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

// The 16-byte unit at u of a window whose start s is 16-byte aligned: a unit lies wholly in front of the
// window or wholly inside it, and the initial register lands on the first dword of the unit at s.
__device__ __forceinline__ Acc4 first_unit_aligned(uintptr_t u, uintptr_t s) {
  Acc4 a = {0, 0, 0, 0};
  if (u >= s) {
    const uint4 d = ld16(u);
    a.a0 = d.x ^ (u == s ? 0xFFFFFFFFu : 0u);
    a.a1 = d.y;
    a.a2 = d.z;
    a.a3 = d.w;
  }
  return a;
}

// Window g -> (chunk, index inside the chunk).  64-bit division is some 150 instructions on this
// hardware, so the common cases avoid it: a power-of-two window count per chunk, or a batch whose
// window count fits 32 bits.
__device__ __forceinline__ void split_window(const CrcLaunch& p, uint64_t g, uint64_t& chunk, uint64_t& wi) {
  if (p.wpc_shift >= 0) chunk = g >> p.wpc_shift;
  else if ((p.n_windows >> 32) == 0) chunk = (uint32_t)g / (uint32_t)p.wpc;
  else chunk = g / p.wpc;
  wi = g - chunk * p.wpc;
}

// Builds the LDS tables from the 32-entry operator bases: the four byte tables of operator `main_op` in
// 32 copies, and the nibble tables of operators 0..kNibOps-1.  All kThreads threads take part.
__device__ __forceinline__ void build_tables(char* lds, const PolyBasis& pb, int main_op, int tid) {
  const int j = tid >> 8, b = tid & 255;  // byte table j, entry b
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) v ^= ((b >> i) & 1) ? pb.op[main_op].v[8 * j + i] : 0u;
  uint4* dst = reinterpret_cast<uint4*>(lds + (j >> 1) * 65536 + b * 256 + (j & 1) * 128);
  const uint4 v4 = make_uint4(v, v, v, v);
#pragma unroll
  for (int r = 0; r < 8; r++) dst[r] = v4;
  for (int e = tid; e < kNibOps * 128; e += kThreads) {
    const int op = e >> 7, q = (e >> 4) & 7, nv = e & 15;
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) w ^= ((nv >> i) & 1) ? pb.op[op].v[4 * q + i] : 0u;
    *reinterpret_cast<uint32_t*>(lds + kNibOff + e * 4) = w;
  }
}

// The tables of the 16-copy layout: the byte tables of operator `main_op` and of "advance 4 bytes", and the nibble
// tables of operators 0..kNibOps-1.  All kThreads threads take part.
// A thread owns entry b of table j in the stripe set and entry b of table j ^ 1 in the other set: the two lie in
// different halves of the banks.  A 16-byte store is served eight lanes at a time, and the eight have the same j and
// consecutive b, i.e. addresses 256 bytes apart: written in table order they would all meet in one group of four banks
// (eight deep; with a batch cut into 16-32 launches the build was 2-3 % of a launch at 128- and 256-byte windows).  So
// even lanes store into the first set while odd lanes store into the second, each starting at a different quarter of
// its 64 bytes: the eight lanes of a group write eight different bank groups.
__device__ __forceinline__ void build_small_tables(char* lds, const PolyBasis& pb, int main_op, int tid) {
  const int j = tid >> 8, b = tid & 255;  // byte table j (uniform over a wavefront), entry b
  uint32_t v = 0, z = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    v ^= ((b >> i) & 1) ? pb.op[main_op].v[8 * j + i] : 0u;
    z ^= ((b >> i) & 1) ? pb.op[kOpZ4].v[8 * (j ^ 1) + i] : 0u;
  }
  char* dv = lds + b * 256 + j * 64;
  char* dz = lds + kSmallZ4Off + b * 256 + (j ^ 1) * 64;
  const bool odd = (tid & 1) != 0;
  const int rot = (tid >> 1) & 3;
  char* first = odd ? dz : dv;
  char* second = odd ? dv : dz;
  const uint32_t fv = odd ? z : v, sv = odd ? v : z;
  const uint4 f4 = make_uint4(fv, fv, fv, fv), s4 = make_uint4(sv, sv, sv, sv);
#pragma unroll
  for (int r = 0; r < 4; r++) *reinterpret_cast<uint4*>(first + ((r + rot) & 3) * 16) = f4;
#pragma unroll
  for (int r = 0; r < 4; r++) *reinterpret_cast<uint4*>(second + ((r + rot) & 3) * 16) = s4;
  for (int e = tid; e < kNibOps * 128; e += kThreads) {
    const int op = e >> 7, q = (e >> 4) & 7, nv = e & 15;
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) w ^= ((nv >> i) & 1) ? pb.op[op].v[4 * q + i] : 0u;
    *reinterpret_cast<uint32_t*>(lds + kSmallNibOff + e * 4) = w;
  }
}

// Stores (compute) or checks (verify) the digest `be` (already big-endian in memory order) of window g.
template <int MODE>
__device__ __forceinline__ void emit(const CrcLaunch& p, uint64_t g, uint64_t chunk, uint64_t wi, uint32_t be) {
  if (MODE == kModeCompute) {
    uint8_t* out = p.sums + g * 4;
    if ((reinterpret_cast<uintptr_t>(out) & 3) == 0) {
      *reinterpret_cast<uint32_t*>(out) = be;
    } else {
      out[0] = (uint8_t)be; out[1] = (uint8_t)(be >> 8); out[2] = (uint8_t)(be >> 16); out[3] = (uint8_t)(be >> 24);
    }
  } else {
    // ChecksumData.verifyChecksumDataMatches: stored[startIndex + i] against computed[i]
    const uint64_t si = p.start_index + wi;
    bool bad;
    if (si >= p.n_expected) {
      bad = true;
    } else {
      const uint8_t* ex = p.expected + chunk * p.expected_stride + si * 4;
      uint32_t ev;
      if ((reinterpret_cast<uintptr_t>(ex) & 3) == 0) ev = *reinterpret_cast<const uint32_t*>(ex);
      else ev = (uint32_t)ex[0] | ((uint32_t)ex[1] << 8) | ((uint32_t)ex[2] << 16) | ((uint32_t)ex[3] << 24);
      bad = ev != be;
    }
    if (bad) atomicMin(p.first_mismatch + chunk, (unsigned long long)wi);
  }
}

// Verify, in two halves, for kernels that know a window's place long before its digest: the stored digest is requested
// first (stored[startIndex + i], ChecksumData.verifyChecksumDataMatches) and compared when the computed one exists, so
// that the request's latency passes behind the window's own loads instead of in front of the next window's.
struct Stored {
  uint32_t value;
  bool missing;  // startIndex + i lies beyond the stored list
};
__device__ __forceinline__ Stored stored_digest(const CrcLaunch& p, uint64_t chunk, uint64_t wi) {
  const uint64_t si = p.start_index + wi;
  if (si >= p.n_expected) return Stored{0u, true};
  const uint8_t* ex = p.expected + chunk * p.expected_stride + si * 4;
  if ((reinterpret_cast<uintptr_t>(ex) & 3) == 0) return Stored{*reinterpret_cast<const uint32_t*>(ex), false};
  return Stored{(uint32_t)ex[0] | ((uint32_t)ex[1] << 8) | ((uint32_t)ex[2] << 16) | ((uint32_t)ex[3] << 24), false};
}
__device__ __forceinline__ void judge(const CrcLaunch& p, uint64_t chunk, uint64_t wi, const Stored& st, uint32_t be) {
  if (st.missing || st.value != be) atomicMin(p.first_mismatch + chunk, (unsigned long long)wi);
}

// Finishes a window: the bytes behind its last 16-byte boundary (or all of a tiny window) bit-serially,
// then getValue() and Ints.toByteArray.
__device__ __forceinline__ uint32_t finish_window(uint32_t state, bool vec, uintptr_t s, uintptr_t e16, uintptr_t e, uint32_t poly) {
  uintptr_t tb = e16;
  if (!vec) {
    state = 0xFFFFFFFFu;
    tb = s;
  }
  for (; tb < e; tb++)
    state = crc_byte_bitwise(state, *reinterpret_cast<const __attribute__((address_space(1))) uint8_t*>(tb), poly);
  return __builtin_bswap32(~state);
}

__device__ __forceinline__ uint32_t slot_fold(const char* lds, const Acc4& a) {
  return nib_apply(lds, kOpZ16, a.a0) ^ nib_apply(lds, kOpZ12, a.a1) ^ nib_apply(lds, kOpZ8, a.a2) ^
         nib_apply(lds, kOpZ4, a.a3);
}

// Folds the accumulators of a wavefront's two rows.  The result, valid in lane 63, is the register a
// zero-initialised CRC would hold after the 2 KiB the 512 accumulators spell out.  Row A lies 1 KiB in
// front of row B, so lane by lane A is advanced by 1 KiB and merged into B before the lanes are folded.
__device__ __forceinline__ uint32_t wave_fold(const char* lds, const Acc4& a, const Acc4& b, int lane) {
  const uint32_t x = slot_fold(lds, a);
  uint32_t y = slot_fold(lds, b) ^ nib_apply(lds, kOpZ1024, x);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const uint32_t t = nib_apply(lds, kOpZ16 + i, lanes_up(y, i));
    y ^= (lane >= (1 << i)) ? t : 0u;
  }
  return y;
}

template <int G, int MODE, int PF>
__global__ __launch_bounds__(kThreads) void crc_windows_kernel(const CrcLaunch p) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  constexpr uint32_t kStripe = kWaveBytes * G;
  constexpr int kWinPerBlock = kWaves / G;  // windows a block works on at once
  constexpr int kMainOp = kOpZ2048 + (G == 1 ? 0 : G == 2 ? 1 : G == 4 ? 2 : G == 8 ? 3 : 4);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const PolyBasis& pb = c_basis[p.poly_idx];

  build_tables(lds, pb, kMainOp, tid);
  __syncthreads();

  uint32_t* part = reinterpret_cast<uint32_t*>(lds + kPartOff);
  const uint32_t lb01 = (uint32_t)(lane & 31) * 4u;
  const uint32_t lb23 = lb01 + 65536u;
  const int sub = wave / G;   // which of the block's concurrent windows this wavefront works on
  const int wsub = wave % G;  // its place inside each stripe

  uint32_t it = 0;
  for (uint64_t g0 = (uint64_t)blockIdx.x * kWinPerBlock; g0 < p.n_windows;
       g0 += (uint64_t)gridDim.x * kWinPerBlock, it++) {
    const uint64_t g = g0 + sub;
    const bool valid = g < p.n_windows;
    uintptr_t s = 0, e = 0, e16 = 0;
    uint64_t chunk = 0, wi = 0;
    bool vec = false;
    uint32_t r = 0;
    if (valid) {
      split_window(p, g, chunk, wi);
      const uint64_t woff = wi * (uint64_t)p.bpc;
      const uint64_t left = p.chunk_len - woff;
      const uint64_t len = left < p.bpc ? left : p.bpc;
      s = reinterpret_cast<uintptr_t>(p.data) + chunk * p.chunk_stride + woff;
      e = s + len;
      e16 = e & ~(uintptr_t)15;
      vec = e16 >= s + 4;  // the aligned part can hold the folded initial register
    }
    if (vec) {
      const uintptr_t s16 = s & ~(uintptr_t)15;
      const uint64_t nstripes = (e16 - s16 + kStripe - 1) / kStripe;
      uintptr_t u = e16 - nstripes * kStripe + kWaveBytes * wsub + 16u * lane;  // row A; row B is 1 KiB further
      // first stripe: may start in front of the window; also takes the initial register
      Acc4 a = first_unit(u, s), b = first_unit(u + 1024, s);
      u += kStripe;
      uint64_t n = nstripes - 1;  // full stripes left
      // When the window starts in the last bytes of the first stripe, up to three bytes of the folded
      // initial register belong to the first 16-byte unit of the next stripe.
      const uint4 spill = init_spill(u, s);
      bool first = true;
      if (n >= PF) {  // software pipeline: PF stripes (2*PF KiB per wavefront) stay in flight
        uint4 qa[PF], qb[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) {
          qa[k] = ld16(u + k * (uintptr_t)kStripe);
          qb[k] = ld16(u + k * (uintptr_t)kStripe + 1024);
        }
        qa[0] = xor4(qa[0], spill);
        first = false;
        u += PF * (uintptr_t)kStripe;
        n -= PF;
        while (n >= PF) {
#pragma unroll
          for (int k = 0; k < PF; k++) {
            absorb(lds, lb01, lb23, a, qa[k]);
            absorb(lds, lb01, lb23, b, qb[k]);
            qa[k] = ld16(u + k * (uintptr_t)kStripe);
            qb[k] = ld16(u + k * (uintptr_t)kStripe + 1024);
          }
          u += PF * (uintptr_t)kStripe;
          n -= PF;
        }
#pragma unroll
        for (int k = 0; k < PF; k++) {
          absorb(lds, lb01, lb23, a, qa[k]);
          absorb(lds, lb01, lb23, b, qb[k]);
        }
      }
      for (; n > 0; n--, u += kStripe) {
        uint4 da = ld16(u);
        const uint4 db = ld16(u + 1024);
        if (first) {
          da = xor4(da, spill);
          first = false;
        }
        absorb(lds, lb01, lb23, a, da);
        absorb(lds, lb01, lb23, b, db);
      }
      r = wave_fold(lds, a, b, lane);
    }

    // ---- merge the G wavefronts of a window ----
    uint32_t state;
    if (G > 1) {
      uint32_t* slot = part + (it & 1) * kWaves;
      if (lane == 63) slot[wave] = r;
      __syncthreads();
      if (wsub != 0) continue;  // one wavefront per window finishes it (loop bounds are block-uniform)
      uint32_t x = (lane < G) ? slot[sub * G + lane] : 0u;
#pragma unroll
      for (int i = 0; (1 << i) < G; i++) {
        const uint32_t t = nib_apply(lds, kOpZ2048 + i, lanes_up(x, i));
        x ^= (lane >= (1 << i)) ? t : 0u;
      }
      state = __shfl(x, G - 1, 64);
    } else {
      state = __shfl(r, 63, 64);
    }
    if (!valid) continue;

    const uint32_t be = finish_window(state, vec, s, e16, e, p.poly_idx ? kPolyCrc32c : kPolyCrc32);
    if (lane == 0) emit<MODE>(p, g, chunk, wi, be);
  }
}

// Small windows: LPW lanes per window, so one wavefront row holds 64/LPW windows and a wavefront works on
// 2*64/LPW windows at once.  Same inner loop (stripe = 16*LPW bytes of each window), but the fold is shared:
// log2(LPW) tree levels serve all windows of a row.  Window geometry is per lane; lanes whose window has
// fewer stripes simply stop absorbing earlier.
struct Win {
  uintptr_t s, e, e16;
  uint64_t chunk, wi;
  uint32_t nst;  // stripes of the aligned part (0: nothing to vectorise)
  bool valid, vec;
};

template <uint32_t kStripe>
__device__ __forceinline__ Win window_of(const CrcLaunch& p, uint64_t g) {
  Win w{};
  w.valid = g < p.n_windows;
  if (w.valid) {
    split_window(p, g, w.chunk, w.wi);
    const uint64_t woff = w.wi * (uint64_t)p.bpc;
    const uint64_t left = p.chunk_len - woff;
    const uint64_t len = left < p.bpc ? left : p.bpc;
    w.s = reinterpret_cast<uintptr_t>(p.data) + w.chunk * p.chunk_stride + woff;
    w.e = w.s + len;
    w.e16 = w.e & ~(uintptr_t)15;
    w.vec = w.e16 >= w.s + 4;
    if (w.vec) w.nst = (uint32_t)((w.e16 - (w.s & ~(uintptr_t)15) + kStripe - 1) / kStripe);
  }
  return w;
}

// The registers of the two windows a lane group works on (rows A and B of the wavefront), valid in the last
// lane of each group.  All 64 lanes call together.
template <int LPW, bool AL>
__device__ __forceinline__ void subwave_pair(const char* lds, const SmallLanes& ln, int sl, const Win& wa,
                                             const Win& wb, uint32_t& x, uint32_t& y) {
  constexpr uint32_t kStripe = 16u * LPW;
  constexpr int kLevels = LPW == 4 ? 2 : LPW == 8 ? 3 : LPW == 16 ? 4 : 5;
  uintptr_t ua = wa.e16 - (uintptr_t)wa.nst * kStripe + 16u * sl;
  uintptr_t ub = wb.e16 - (uintptr_t)wb.nst * kStripe + 16u * sl;
  Acc4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (wa.nst) a = AL ? first_unit_aligned(ua, wa.s) : first_unit(ua, wa.s);
  if (wb.nst) b = AL ? first_unit_aligned(ub, wb.s) : first_unit(ub, wb.s);
  ua += kStripe;
  ub += kStripe;
  uint4 na = make_uint4(0, 0, 0, 0), nb = make_uint4(0, 0, 0, 0);
  if (1 < wa.nst) na = AL ? ld16(ua) : xor4(ld16(ua), init_spill(ua, wa.s));
  if (1 < wb.nst) nb = AL ? ld16(ub) : xor4(ld16(ub), init_spill(ub, wb.s));
  for (uint32_t j = 1; __ballot(j < wa.nst || j < wb.nst) != 0ull; j++) {
    const uint4 ca = na, cb = nb;
    ua += kStripe;
    ub += kStripe;
    if (j + 1 < wa.nst) na = ld16(ua);
    if (j + 1 < wb.nst) nb = ld16(ub);
    if (j < wa.nst) ln.absorb(lds, a, ca);
    if (j < wb.nst) ln.absorb(lds, b, cb);
  }
  // Fold.  Only the last lane of a group needs its window's register, so level i has work for the lanes
  // whose low i+1 bits are all ones (they take in the lane 2^i below).  That leaves the lanes below idle:
  // they advance row B's values in the same instruction stream, and the result moves up by one DPP shift.
  // Lanes outside that pattern end up with values nobody reads.
  ln.merge_units(lds, a, b, x, y);
#pragma unroll
  for (int i = 0; i < kLevels; i++) {
    const bool taker = (sl & ((2 << i) - 1)) == (2 << i) - 1;
    const uint32_t below = lanes_up(x, i);  // outside the select: every lane must execute the shift
    const uint32_t t = nib_apply<kSmallNibOff>(lds, kOpZ16 + i, taker ? below : y);
    x ^= t;
    y ^= lanes_up(t, i);
  }
}

template <int LPW, int MODE, bool AL>
__global__ __launch_bounds__(kThreads) void crc_subwave_kernel(const CrcLaunch p) {
  __shared__ __attribute__((aligned(16))) char lds[kSmallLdsBytes];
  constexpr uint32_t kStripe = 16u * LPW;
  constexpr int kWinPerRow = 64 / LPW;
  constexpr int kWinPerWave = 2 * kWinPerRow;
  constexpr int kLevels = LPW == 4 ? 2 : LPW == 8 ? 3 : LPW == 16 ? 4 : 5;
  constexpr int kMainOp = kOpZ16 + kLevels;  // advance by 16*LPW bytes

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  build_small_tables(lds, c_basis[p.poly_idx], kMainOp, tid);
  __syncthreads();

  const SmallLanes ln(lane);
  const int sl = lane & (LPW - 1);  // lane inside its window's group
  const int grp = lane / LPW;
  const uint32_t poly = p.poly_idx ? kPolyCrc32c : kPolyCrc32;

  for (uint64_t w0 = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave; w0 < p.n_windows;
       w0 += (uint64_t)gridDim.x * kWaves * kWinPerWave) {
    const uint64_t ga = w0 + grp, gb = w0 + kWinPerRow + grp;
    const Win wa = window_of<kStripe>(p, ga), wb = window_of<kStripe>(p, gb);
    uint32_t x, y;
    subwave_pair<LPW, AL>(lds, ln, sl, wa, wb, x, y);
    if (sl == LPW - 1) {  // the last lane of a group holds its window's register
      if (wa.valid) emit<MODE>(p, ga, wa.chunk, wa.wi, finish_window(x, wa.vec, wa.s, wa.e16, wa.e, poly));
      if (wb.valid) emit<MODE>(p, gb, wb.chunk, wb.wi, finish_window(y, wb.vec, wb.s, wb.e16, wb.e, poly));
    }
  }
}

// Windows given one by one (start, length, where the digest goes): a batch of unrelated small chunks in one
// launch.  Same work per window as crc_subwave_kernel with the any-alignment geometry.
template <uint32_t kStripe>
__device__ __forceinline__ Win window_of_desc(const WindowDesc* desc, uint32_t n, uint64_t g) {
  Win w{};
  w.valid = g < n;
  if (w.valid) {
    const WindowDesc d = desc[g];
    w.s = reinterpret_cast<uintptr_t>(d.data);
    w.e = w.s + d.len;
    w.e16 = w.e & ~(uintptr_t)15;
    w.vec = w.e16 >= w.s + 4;
    if (w.vec) w.nst = (uint32_t)((w.e16 - (w.s & ~(uintptr_t)15) + kStripe - 1) / kStripe);
  }
  return w;
}

__device__ __forceinline__ void store_be(uint8_t* out, uint32_t be) {
  if ((reinterpret_cast<uintptr_t>(out) & 3) == 0) {
    *reinterpret_cast<uint32_t*>(out) = be;
  } else {
    out[0] = (uint8_t)be; out[1] = (uint8_t)(be >> 8); out[2] = (uint8_t)(be >> 16); out[3] = (uint8_t)(be >> 24);
  }
}

template <int LPW>
__global__ __launch_bounds__(kThreads) void crc_desc_kernel(const WindowDesc* desc, uint32_t n, int poly_idx) {
  __shared__ __attribute__((aligned(16))) char lds[kSmallLdsBytes];
  constexpr uint32_t kStripe = 16u * LPW;
  constexpr int kWinPerRow = 64 / LPW;
  constexpr int kWinPerWave = 2 * kWinPerRow;
  constexpr int kLevels = LPW == 4 ? 2 : LPW == 8 ? 3 : LPW == 16 ? 4 : 5;
  constexpr int kMainOp = kOpZ16 + kLevels;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  build_small_tables(lds, c_basis[poly_idx], kMainOp, tid);
  __syncthreads();

  const SmallLanes ln(lane);
  const int sl = lane & (LPW - 1);
  const int grp = lane / LPW;
  const uint32_t poly = poly_idx ? kPolyCrc32c : kPolyCrc32;

  for (uint64_t w0 = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave; w0 < n;
       w0 += (uint64_t)gridDim.x * kWaves * kWinPerWave) {
    const uint64_t ga = w0 + grp, gb = w0 + kWinPerRow + grp;
    const Win wa = window_of_desc<kStripe>(desc, n, ga), wb = window_of_desc<kStripe>(desc, n, gb);
    uint32_t x, y;
    subwave_pair<LPW, false>(lds, ln, sl, wa, wb, x, y);
    if (sl == LPW - 1) {
      if (wa.valid) store_be(desc[ga].out, finish_window(x, wa.vec, wa.s, wa.e16, wa.e, poly));
      if (wb.valid) store_be(desc[gb].out, finish_window(y, wb.vec, wb.s, wb.e16, wb.e, poly));
    }
  }
}

// The small-window kernel for the plainest layout: every window is a full bpc bytes, windows follow one
// another without gaps (chunks back to back, chunk length a multiple of bpc), everything 16-byte aligned and
// bpc a multiple of UNR stripes.  Then window g is simply [data + g*bpc, +bpc), every lane runs the same
// number of stripes, nothing is masked and nothing is left for the bit-serial tail: the per-window
// bookkeeping, a quarter of the instructions at 512-byte windows, shrinks to one multiply-add.
// Lanes past the last window redo the last window and store nothing.
//
// Stripes are loaded UNR at a time.  The inner loop costs the same per byte whatever the stripe size, but
// the fold costs one operator application per accumulator, so few lanes per window (a short stripe) would
// make small windows cheaper -- if the memory path kept up: 4 lanes per window read 64-byte pieces and reach
// 4.1 TB/s at 512-byte windows, 5.3 with the two halves of a line requested by adjacent instructions
// (UNR = 2), against 5.8 for 8 lanes (profiles/r01_crc_flat_sweep.log).  UNR = 2 is used where it measured
// faster: windows of 256 bytes and less.
//
// Phases (PH = 4).  Left alone, every wavefront of the chip walks through its windows from offset 0, so at any
// moment tens of thousands of reads go to addresses that agree in the bits below the window size; a bare read in
// that pattern runs at 6.2-6.9 TB/s for windows of 4 to 32 KiB, and at 7.0-7.2 when the wavefronts of a workgroup
// start 0, 1/4, 1/2 and 3/4 of the way into their windows and wrap round (tools/hbm_read_bench.hip, "flat").  A
// CRC does not care where its window is entered: with window = head | tail, the wavefront sums the tail first,
// parks those accumulators at the wrap, sums the head on fresh ones (the initial register belongs to the head),
// and at the end advances the head's accumulators through the tail's length -- 3/4, 1/2 or 1/4 of a power-of-two
// window: one or two of the power-of-two nibble operators -- and XORs the tail's in.  The fold is unchanged.
//
// Measured and not kept: issuing the loads of a wavefront's NEXT windows before the arithmetic of the current ones
// (whole windows of 128 B to 2 KiB as one load group, two register buffers taking turns, 114 VGPRs, no spills) was
// no faster at 128 / 256 B and 5-12 % slower from 512 B to 2 KiB: what these windows lack is not bytes in flight.
//
// HOLD > 0 (compute): a wavefront keeps its digests in registers and stores them when all its reads are done.  A digest
// line written while the chip streams the data is pushed out of the L2 by that stream and reaches DRAM as a lone write
// inside a saturated read stream, which costs its channel a bus turnaround and a row change: 128 MiB of digests (512-byte
// windows, 16 GiB of data) cost 15 % of the launch, where the same stores issued after the reads cost 2 %
// (tools/hbm_read_bench.hip, "rolling").  The digests of one pass (valid in the last lane of each lane group) are pulled
// into one slot group of a collector register by a ds_bpermute; a collector holds LPW passes, a wavefront HOLD collectors
// per row, i.e. HOLD * LPW passes -- the launcher cuts a larger batch into launches of that many passes (crc_launch), and a
// wavefront that is nevertheless given more stores the excess as it comes.
template <int LPW, int MODE, int UNR, int PH, int HOLD = 0>
__global__ __launch_bounds__(kThreads) void crc_flat_kernel(const CrcLaunch p) {
  __shared__ __attribute__((aligned(16))) char lds[kSmallLdsBytes];
  static_assert(HOLD == 0 || MODE == kModeCompute, "digests are held by the compute forms only");
  constexpr uint32_t kStripe = 16u * LPW;
  constexpr uint32_t kGroup = UNR * kStripe;  // bytes of a window one load group covers
  constexpr int kWinPerRow = 64 / LPW;
  constexpr int kWinPerWave = 2 * kWinPerRow;
  constexpr int kLevels = LPW == 4 ? 2 : LPW == 8 ? 3 : LPW == 16 ? 4 : 5;
  constexpr int kMainOp = kOpZ16 + kLevels;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  build_small_tables(lds, c_basis[p.poly_idx], kMainOp, tid);
  __syncthreads();

  const SmallLanes ln(lane);
  const int sl = lane & (LPW - 1);
  const int grp = lane / LPW;
  const uint32_t init = sl == 0 ? 0xFFFFFFFFu : 0u;  // the window's first four bytes are lane 0's first dword
  const uintptr_t base = reinterpret_cast<uintptr_t>(p.data) + 16u * sl;
  const uint64_t last = p.n_windows - 1;
  // PH > 1: this wavefront starts `phase` bytes into each of its windows and wraps round (bpc is then a power of
  // two and a multiple of PH load groups); the part behind the phase is the window's tail, read first
  // (the wavefront's number is the same in all its lanes: taken from the first lane it is a scalar, and with it the
  // whole loop control below)
  const uint32_t phase = PH > 1 ? (uint32_t)(__builtin_amdgcn_readfirstlane(wave) & (PH - 1)) * (p.bpc / PH) : 0u;
  const uint32_t init0 = phase == 0 ? init : 0u;
  // HOLD: collectors of rows A and B; slot group s (lanes s * kWinPerRow ...) of collector q holds pass q * LPW + s
  constexpr int kHeld = HOLD ? HOLD : 1;
  uint32_t keep_a[kHeld], keep_b[kHeld];
#pragma unroll
  for (int q = 0; q < kHeld; q++) keep_a[q] = keep_b[q] = 0u;
  uint32_t cur_a = 0, cur_b = 0;
  uint32_t pass = 0;  // passes of this wavefront so far (the same in all lanes)
  const uint32_t pull = (uint32_t)((lane % kWinPerRow) * LPW + (LPW - 1)) * 4u;  // ds_bpermute address: the lane that holds window (lane % kWinPerRow) of a row
  const uint32_t slot = (uint32_t)(lane / kWinPerRow);                            // the pass (mod LPW) this lane keeps
  // The fold of a window group -- seven LDS round trips that wait for one another -- runs one pass late in the holding
  // form, between the requests for the next group's first stripes and their first use: it then fills the memory latency
  // every pass begins with instead of standing behind the pass.  Measured on MI355X with 8 lanes per window: 5.46 -> 5.73
  // TB/s at 256-byte windows, 6.44 -> 6.59 at 512 B, 6.89 -> 6.96 at 1 KiB; with 4 lanes (128 B) it loses (5.42 -> 5.06)
  // and with 16 (4 KiB) it changes nothing, so those keep the fold inside the pass.  Only the collectors depend on the
  // order of the passes, and they see it unchanged.
  // Verify does the same: the late pass carries its windows' places and stored digests along.
  constexpr bool kLate = (HOLD > 0 || MODE == kModeVerify) && PH == 1 && LPW == 8;
  Acc4 late_a = {0, 0, 0, 0}, late_b = {0, 0, 0, 0};
  bool late = false;
  struct Place {  // verify: where a pass's two windows (of this lane group) stand and what is stored for them
    uint64_t chunk_a, wi_a, chunk_b, wi_b;
    Stored st_a, st_b;
    bool has_a, has_b;
  };
  Place late_at = {0, 0, 0, 0, {0u, false}, {0u, false}, false, false};
  auto judge_pass = [&](const Place& at, uint32_t x, uint32_t y) {
    if (sl == LPW - 1) {
      if (at.has_a) judge(p, at.chunk_a, at.wi_a, at.st_a, __builtin_bswap32(~x));
      if (at.has_b) judge(p, at.chunk_b, at.wi_b, at.st_b, __builtin_bswap32(~y));
    }
  };
  auto fold = [&](const Acc4& fa, const Acc4& fb, uint32_t& x, uint32_t& y) {
#ifdef OZCS_TUNE_NOFOLD  // timing-only build: what the loop and the loads cost without the per-window fold; results are wrong on purpose
    x = fa.a0 ^ fa.a1 ^ fa.a2 ^ fa.a3;
    y = fb.a0 ^ fb.a1 ^ fb.a2 ^ fb.a3;
#else
    ln.merge_units(lds, fa, fb, x, y);
#pragma unroll
    for (int i = 0; i < kLevels; i++) {  // as in crc_subwave_kernel
      const bool taker = (sl & ((2 << i) - 1)) == (2 << i) - 1;
      const uint32_t below = lanes_up(x, i);
      const uint32_t t = nib_apply<kSmallNibOff>(lds, kOpZ16 + i, taker ? below : y);
      x ^= t;
      y ^= lanes_up(t, i);
    }
#endif
  };
  auto collect = [&](uint32_t x, uint32_t y) {  // files one pass's digests into the collectors (pass < HOLD * LPW)
    const uint32_t ta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pull, (int)x);
    const uint32_t tb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pull, (int)y);
    const bool mine = slot == (pass & (LPW - 1));
    cur_a = mine ? ta : cur_a;
    cur_b = mine ? tb : cur_b;
    pass++;
    if ((pass & (LPW - 1)) == 0) {  // the collector is full (a scalar condition)
      const uint32_t full = pass / LPW - 1;
#pragma unroll
      for (int q = 0; q < kHeld; q++) {
        keep_a[q] = full == (uint32_t)q ? cur_a : keep_a[q];
        keep_b[q] = full == (uint32_t)q ? cur_b : keep_b[q];
      }
    }
  };

  for (uint64_t w0 = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave; w0 < p.n_windows;
       w0 += (uint64_t)gridDim.x * kWaves * kWinPerWave) {
    const uint64_t ga = w0 + grp, gb = w0 + kWinPerRow + grp;
    const uintptr_t wa = base + (ga < last ? ga : last) * p.bpc, wb = base + (gb < last ? gb : last) * p.bpc;
    uint64_t chunk_a = 0, wi_a = 0, chunk_b = 0, wi_b = 0;
    Stored st_a = {0u, false}, st_b = {0u, false};
    if (MODE == kModeVerify && sl == LPW - 1) {  // the lanes that will hold the windows' registers
      if (ga <= last) {
        split_window(p, ga, chunk_a, wi_a);
        st_a = stored_digest(p, chunk_a, wi_a);
      }
      if (gb <= last) {
        split_window(p, gb, chunk_b, wi_b);
        st_b = stored_digest(p, chunk_b, wi_b);
      }
    }
    uint4 ca[UNR], cb[UNR], na[UNR], nb[UNR];
    Acc4 a, b;
    Acc4 ta = {0, 0, 0, 0}, tb = {0, 0, 0, 0};  // PH > 1: the accumulators of the tail, parked at the wrap
    if (PH == 1) {
      uintptr_t ua = wa, ub = wb;
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        ca[k] = ld16(ua + k * kStripe);
        cb[k] = ld16(ub + k * kStripe);
      }
      if (p.bpc > kGroup) {
        ua += kGroup;
        ub += kGroup;
#pragma unroll
        for (int k = 0; k < UNR; k++) {
          na[k] = ld16(ua + k * kStripe);
          nb[k] = ld16(ub + k * kStripe);
        }
      }
      if (kLate && late) {  // the previous pass's fold, while this pass's first stripes are on their way
        uint32_t x, y;
        fold(late_a, late_b, x, y);
        if (MODE == kModeVerify) judge_pass(late_at, x, y);
        else collect(x, y);
        late = false;
      }
      a = {ca[0].x ^ init, ca[0].y, ca[0].z, ca[0].w};
      b = {cb[0].x ^ init, cb[0].y, cb[0].z, cb[0].w};
#pragma unroll
      for (int k = 1; k < UNR; k++) {
        ln.absorb(lds, a, ca[k]);
        ln.absorb(lds, b, cb[k]);
      }
      for (uint32_t done = kGroup; done < p.bpc; done += kGroup) {
#pragma unroll
        for (int k = 0; k < UNR; k++) {
          ca[k] = na[k];
          cb[k] = nb[k];
        }
        if (done + kGroup < p.bpc) {
          ua += kGroup;
          ub += kGroup;
#pragma unroll
          for (int k = 0; k < UNR; k++) {
            na[k] = ld16(ua + k * kStripe);
            nb[k] = ld16(ub + k * kStripe);
          }
        }
#pragma unroll
        for (int k = 0; k < UNR; k++) {
          ln.absorb(lds, a, ca[k]);
          ln.absorb(lds, b, cb[k]);
        }
      }
    } else {
      // Two buffers of load groups take turns (the window holds an even number of groups), so that nothing is
      // copied from "next" to "current"; all offsets and conditions are scalars.
      auto load = [&](uint4 (&da)[UNR], uint4 (&db)[UNR], uint32_t off) {
#pragma unroll
        for (int k = 0; k < UNR; k++) {
          da[k] = ld16(wa + off + k * kStripe);
          db[k] = ld16(wb + off + k * kStripe);
        }
      };
      auto consume = [&](const uint4 (&da)[UNR], const uint4 (&db)[UNR], uint32_t off) {
        if (off == 0) {  // wrapped round: the window's head starts here, on accumulators of its own
          ta = a;
          tb = b;
          a = {da[0].x ^ init, da[0].y, da[0].z, da[0].w};
          b = {db[0].x ^ init, db[0].y, db[0].z, db[0].w};
#pragma unroll
          for (int k = 1; k < UNR; k++) {
            ln.absorb(lds, a, da[k]);
            ln.absorb(lds, b, db[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < UNR; k++) {
            ln.absorb(lds, a, da[k]);
            ln.absorb(lds, b, db[k]);
          }
        }
      };
      auto next = [&](uint32_t off) { return off + kGroup == p.bpc ? 0u : off + kGroup; };
      uint32_t oc = phase, on = next(phase);  // offsets of the groups in ca/cb and in na/nb
      load(ca, cb, oc);
      load(na, nb, on);
      a = {ca[0].x ^ init0, ca[0].y, ca[0].z, ca[0].w};
      b = {cb[0].x ^ init0, cb[0].y, cb[0].z, cb[0].w};
#pragma unroll
      for (int k = 1; k < UNR; k++) {
        ln.absorb(lds, a, ca[k]);
        ln.absorb(lds, b, cb[k]);
      }
      for (uint32_t done = kGroup; done < p.bpc; done += 2 * kGroup) {  // an odd number of groups is left
        const bool more = done + kGroup < p.bpc;
        if (more) {
          oc = next(on);
          load(ca, cb, oc);
        }
        consume(na, nb, on);
        if (more) {
          on = next(oc);
          if (done + 2 * kGroup < p.bpc) load(na, nb, on);
          consume(ca, cb, oc);
        }
      }
    }
    if (PH > 1 && phase != 0) {
      // window = head | tail: the head's accumulators advance through the tail's bpc - phase bytes (one or two
      // power-of-two operators: PH = 4 gives tails of 3/4, 1/2 and 1/4 of a window) and meet the tail's
      const uint32_t half = kSmallNibOff + (uint32_t)(__builtin_ctz(p.bpc) - 2) * kNibOpBytes;  // operator of bpc / 2 bytes
      const uint32_t quarter = half - kNibOpBytes;
      const uint32_t tail = p.bpc - phase;
      uint32_t* va[8] = {&a.a0, &a.a1, &a.a2, &a.a3, &b.a0, &b.a1, &b.a2, &b.a3};
      const uint32_t vt[8] = {ta.a0, ta.a1, ta.a2, ta.a3, tb.a0, tb.a1, tb.a2, tb.a3};
      if (tail & (p.bpc >> 1)) {
#pragma unroll
        for (int k = 0; k < 8; k++) *va[k] = nib_apply_at(lds, half, *va[k]);
      }
      if (tail & (p.bpc >> 2)) {
#pragma unroll
        for (int k = 0; k < 8; k++) *va[k] = nib_apply_at(lds, quarter, *va[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; k++) *va[k] ^= vt[k];
    }
    if (kLate && (MODE == kModeVerify || pass < (uint32_t)(HOLD * LPW))) {  // folded during the next pass (or after the last)
      late_a = a;
      late_b = b;
      late = true;
      if (MODE == kModeVerify) late_at = Place{chunk_a, wi_a, chunk_b, wi_b, st_a, st_b, ga <= last, gb <= last};
      continue;
    }
    uint32_t x, y;
    fold(a, b, x, y);
    if (HOLD > 0 && !kLate && pass < (uint32_t)(HOLD * LPW)) {
      collect(x, y);
      continue;
    }
    if (sl == LPW - 1) {
      if (MODE == kModeCompute) {  // sums is 4-byte aligned in this layout (checked by the launcher)
        uint32_t* out = reinterpret_cast<uint32_t*>(p.sums);
        if (ga <= last) out[ga] = __builtin_bswap32(~x);
        if (gb <= last) out[gb] = __builtin_bswap32(~y);
      } else {
        if (ga <= last) judge(p, chunk_a, wi_a, st_a, __builtin_bswap32(~x));
        if (gb <= last) judge(p, chunk_b, wi_b, st_b, __builtin_bswap32(~y));
      }
    }
  }
  if (MODE == kModeVerify && kLate && late) {
    uint32_t x, y;
    fold(late_a, late_b, x, y);
    judge_pass(late_at, x, y);
  }
  if (HOLD > 0) {  // all reads of this wavefront are done: its digests leave now
    if (kLate && late) {
      uint32_t x, y;
      fold(late_a, late_b, x, y);
      collect(x, y);
    }
    if ((pass & (LPW - 1)) != 0) {  // a collector that is not full yet
      const uint32_t part = pass / LPW;
#pragma unroll
      for (int q = 0; q < kHeld; q++) {
        keep_a[q] = part == (uint32_t)q ? cur_a : keep_a[q];
        keep_b[q] = part == (uint32_t)q ? cur_b : keep_b[q];
      }
    }
    uint32_t* out = reinterpret_cast<uint32_t*>(p.sums);
    const uint64_t first = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave + (uint32_t)(lane % kWinPerRow);
    const uint64_t stride = (uint64_t)gridDim.x * kWaves * kWinPerWave;
#pragma unroll
    for (int q = 0; q < kHeld; q++) {
      const uint32_t k = (uint32_t)q * LPW + slot;  // the pass this lane's slot of collector q belongs to
      if (k < pass) {
        const uint64_t g = first + (uint64_t)k * stride;
        if (g <= last) out[g] = __builtin_bswap32(~keep_a[q]);
        if (g + kWinPerRow <= last) out[g + kWinPerRow] = __builtin_bswap32(~keep_b[q]);
      }
    }
  }
}

// (lanes per window, stripes per load group, phases) triples that are built
template <int MODE>
hipError_t launch_flat(const CrcLaunch& p, int lpw, int unr, int phases, int blocks, hipStream_t stream) {
  switch (phases * 1000 + lpw * 8 + unr) {
    case 1000 + 4 * 8 + 1: crc_flat_kernel<4, MODE, 1, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 4 * 8 + 2: crc_flat_kernel<4, MODE, 2, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 8 * 8 + 1: crc_flat_kernel<8, MODE, 1, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 8 * 8 + 2: crc_flat_kernel<8, MODE, 2, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 16 * 8 + 1: crc_flat_kernel<16, MODE, 1, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 16 * 8 + 2: crc_flat_kernel<16, MODE, 2, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 32 * 8 + 1: crc_flat_kernel<32, MODE, 1, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 1000 + 32 * 8 + 2: crc_flat_kernel<32, MODE, 2, 1><<<blocks, kThreads, 0, stream>>>(p); break;
    case 4000 + 8 * 8 + 1: crc_flat_kernel<8, MODE, 1, 4><<<blocks, kThreads, 0, stream>>>(p); break;
    case 4000 + 16 * 8 + 1: crc_flat_kernel<16, MODE, 1, 4><<<blocks, kThreads, 0, stream>>>(p); break;
    case 4000 + 32 * 8 + 1: crc_flat_kernel<32, MODE, 1, 4><<<blocks, kThreads, 0, stream>>>(p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// The compute forms that hold their digests until a wavefront's reads are done; flat_hold(lanes) collectors per row.
// Sixteen with 4 lanes per window, where a collector is full after 4 passes: 128-byte windows then leave in launches of
// 1 GiB instead of 512 MiB, and a launch's fixed cost (table build, first and last pass) weighs half as much
// (MI355X: +2 % at 128 B; with 8 lanes and more it changes nothing).
constexpr int flat_hold(int lanes) { return lanes == 4 ? 16 : 8; }
hipError_t launch_flat_hold(const CrcLaunch& p, int lpw, int unr, int phases, int blocks, hipStream_t stream) {
  if (phases == 4) {
    if (lpw == 8 && unr == 1) crc_flat_kernel<8, kModeCompute, 1, 4, flat_hold(8)><<<blocks, kThreads, 0, stream>>>(p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  switch (lpw * 8 + unr) {
    case 4 * 8 + 1: crc_flat_kernel<4, kModeCompute, 1, 1, flat_hold(4)><<<blocks, kThreads, 0, stream>>>(p); break;
    case 4 * 8 + 2: crc_flat_kernel<4, kModeCompute, 2, 1, flat_hold(4)><<<blocks, kThreads, 0, stream>>>(p); break;
    case 8 * 8 + 1: crc_flat_kernel<8, kModeCompute, 1, 1, flat_hold(8)><<<blocks, kThreads, 0, stream>>>(p); break;
    case 8 * 8 + 2: crc_flat_kernel<8, kModeCompute, 2, 1, flat_hold(8)><<<blocks, kThreads, 0, stream>>>(p); break;
    case 16 * 8 + 1: crc_flat_kernel<16, kModeCompute, 1, 1, flat_hold(16)><<<blocks, kThreads, 0, stream>>>(p); break;
    case 16 * 8 + 2: crc_flat_kernel<16, kModeCompute, 2, 1, flat_hold(16)><<<blocks, kThreads, 0, stream>>>(p); break;
    case 32 * 8 + 1: crc_flat_kernel<32, kModeCompute, 1, 1, flat_hold(32)><<<blocks, kThreads, 0, stream>>>(p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int MODE, bool AL>
hipError_t launch_subwave(const CrcLaunch& p, int lpw, int blocks, hipStream_t stream) {
  switch (lpw) {
    case 4: crc_subwave_kernel<4, MODE, AL><<<blocks, kThreads, 0, stream>>>(p); break;
    case 8: crc_subwave_kernel<8, MODE, AL><<<blocks, kThreads, 0, stream>>>(p); break;
    case 16: crc_subwave_kernel<16, MODE, AL><<<blocks, kThreads, 0, stream>>>(p); break;
    case 32: crc_subwave_kernel<32, MODE, AL><<<blocks, kThreads, 0, stream>>>(p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int MODE, int PF>
hipError_t launch_mode(const CrcLaunch& p, int group, int blocks, hipStream_t stream) {
  switch (group) {
    case 1: crc_windows_kernel<1, MODE, PF><<<blocks, kThreads, 0, stream>>>(p); break;
    case 2: crc_windows_kernel<2, MODE, PF><<<blocks, kThreads, 0, stream>>>(p); break;
    case 4: crc_windows_kernel<4, MODE, PF><<<blocks, kThreads, 0, stream>>>(p); break;
    case 8: crc_windows_kernel<8, MODE, PF><<<blocks, kThreads, 0, stream>>>(p); break;
    case 16: crc_windows_kernel<16, MODE, PF><<<blocks, kThreads, 0, stream>>>(p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Tuning overrides for experiments (read once): OZCS_CRC_GROUP = 1|2|4|8|16, OZCS_CRC_PF = 1|2|3|4.
int env_int(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
const int g_force_group = env_int("OZCS_CRC_GROUP");
const int g_force_pf = env_int("OZCS_CRC_PF");
const int g_force_lpw = env_int("OZCS_CRC_LPW");  // 4|8|16|32 lanes per window, 64 = never use the small-window kernel
const int g_no_aligned = env_int("OZCS_CRC_NO_ALIGNED");  // 1 = always take the any-alignment code path
const int g_no_flat = env_int("OZCS_CRC_NO_FLAT");        // 1 = never take the gapless-full-windows kernel
const int g_force_flat = env_int("OZCS_CRC_FLAT");        // lanes * 8 + stripes per load group, e.g. 34 = 4 lanes, 2 stripes
const int g_force_phases = env_int("OZCS_CRC_PHASES");    // 1 = no wavefront starts inside its windows, 4 = four phases
const int g_force_hold = env_int("OZCS_CRC_HOLD");        // 1 = digests always held to the end of a wavefront's reads, -1 = never

}  // namespace

int crc_choose_group(uint64_t n_windows, uint32_t bpc, int num_cus) {
  if (g_force_group == 1 || g_force_group == 2 || g_force_group == 4 || g_force_group == 8 || g_force_group == 16)
    return g_force_group;
  // Measured on MI355X at 1 MiB windows (profiles/): the best G gives every wavefront about two
  // window-shares of work (n_windows * G ~ 2 x the chip's 4096 wavefronts) and is never 1 -- with one
  // wavefront per window the slowest wavefronts finish late (6.68 against 6.90 TB/s at 8192 windows).
  // A wavefront should also get at least ~8 KiB of a window.
  // From four windows per wavefront of the chip on (16,384 windows of 1 MiB: the benchmark's batch) eight
  // wavefronts per window are ahead again, for the kernel (7.147 against 7.105 TB/s, three alternating rounds;
  // level at 8,192 windows, behind at 4,096) as for a bare read in the same pattern (7.14 against 7.06 TB/s).
  int g = 16;
  const uint64_t want_waves = (uint64_t)num_cus * kWaves;
  while (g > 2 && n_windows * (uint64_t)(g / 2) >= 2 * want_waves) g >>= 1;
  if (n_windows >= 4 * want_waves) g = 8;
  while (g > 1 && (uint64_t)bpc < (uint64_t)g * 8192u) g >>= 1;
  return g;
}

// Lanes per window for the small-window kernel, or 0 to use the wavefront-per-window kernels.  Measured on
// MI355X (16 GiB batches, profiles/): 8 lanes per window is fastest from 256 B to 256 KiB windows (6.7 TB/s
// at 16 KiB against 5.3 for a wavefront per window; 5.1 against 0.55 at 512 B; 6.96 against 6.75 at 128 KiB);
// 4 lanes read 64-byte pieces and stop at 3.7 TB/s.  Fewer windows ask for more lanes each so that every
// wavefront still has work; tiny windows cap the lanes so that a window still spans a few stripes.
int crc_choose_lpw(uint64_t n_windows, uint32_t bpc, int num_cus) {
  if (g_force_lpw == 4 || g_force_lpw == 8 || g_force_lpw == 16 || g_force_lpw == 32) return g_force_lpw;
  if (g_force_lpw == 64) return 0;
  if (bpc > 262144) return 0;
  const uint64_t waves = (uint64_t)num_cus * kWaves;
  // 256 KiB windows in plenty: the wavefront-per-window kernel (eight wavefronts per window) reads them at 7.10-7.14 TB/s
  // where 8 lanes per window reach 6.92 (65,536 windows, twenty launches after ten); at 32,768 windows it is level with the
  // 16 lanes taken before (6.96 against 6.91), at 16,384 behind the 32 lanes taken there (6.63 against 6.75)
  if (bpc == 262144 && n_windows >= 8 * waves) return 0;
  int want = n_windows >= waves * 16 ? 8 : n_windows >= waves * 8 ? 16 : n_windows >= waves * 4 ? 32 : 64;
  // measured twice over: 16 lanes beat 8 at 4 KiB (6.53 against 6.41 TB/s) and 8 KiB windows (6.67 against
  // 6.33), and nowhere else
  if (want == 8 && bpc >= 4096 && bpc < 16384) want = 16;
  // between 16 and 64 KiB (exclusive) 8 lanes lose to 32: 6.53-6.58 against 6.74-6.76 TB/s at 32 KiB, 6.58 against
  // 6.80 at 24 KiB, 6.54 against 6.82 at 40 KiB, 6.66 against 6.79 at 56 KiB (16 lanes: 5.8 at 32 KiB); at
  // 16 KiB and from 64 KiB on 8 lanes are best again (6.76 / 7.02).  The ranking follows no simple rule of the
  // stride between the pieces one instruction reads, so it is taken from the measurements.
  if (want == 8 && bpc > 16384 && bpc < 65536) want = 32;
  const int cap = bpc >= 4096 ? 64 : bpc >= 2048 ? 32 : bpc >= 1024 ? 16 : bpc >= 256 ? 8 : 4;
  const int lpw = want < cap ? want : cap;
  return lpw == 64 ? 0 : lpw;
}

hipError_t crc_launch_desc(const WindowDesc* d_desc, uint32_t n, uint64_t total_bytes, int poly_idx, int num_cus,
                           hipStream_t stream) {
  if (n == 0) return hipSuccess;
  // lanes per window as for a uniform batch of the same average window size; never the wavefront-per-window
  // kernels (0), which have no descriptor form
  const uint64_t avg = total_bytes / n;
  int lpw = crc_choose_lpw(n, (uint32_t)(avg > 262144 ? 262144 : (avg ? avg : 1)), num_cus);
  if (lpw == 0) lpw = 32;
  const uint64_t per_block = (uint64_t)kWaves * 2 * (64 / lpw);
  uint64_t nb = (n + per_block - 1) / per_block;
  if (nb > (uint64_t)num_cus) nb = num_cus;
  switch (lpw) {
    case 4: crc_desc_kernel<4><<<(int)nb, kThreads, 0, stream>>>(d_desc, n, poly_idx); break;
    case 8: crc_desc_kernel<8><<<(int)nb, kThreads, 0, stream>>>(d_desc, n, poly_idx); break;
    case 16: crc_desc_kernel<16><<<(int)nb, kThreads, 0, stream>>>(d_desc, n, poly_idx); break;
    default: crc_desc_kernel<32><<<(int)nb, kThreads, 0, stream>>>(d_desc, n, poly_idx); break;
  }
  return hipGetLastError();
}

hipError_t crc_launch(const CrcLaunch& launch, int mode, int group, int num_cus, hipStream_t stream) {
  if (launch.n_windows == 0) return hipSuccess;
  CrcLaunch p = launch;
  p.wpc_shift = (p.wpc & (p.wpc - 1)) == 0 ? __builtin_ctzll(p.wpc) : -1;
  if (const int lpw = crc_choose_lpw(p.n_windows, p.bpc, num_cus)) {
    const uint64_t per_block = (uint64_t)kWaves * 2 * (64 / lpw);
    uint64_t nb = (p.n_windows + per_block - 1) / per_block;
    if (nb > (uint64_t)num_cus) nb = num_cus;
    // every window starts on a 16-byte boundary: the masking of partial leading units is not needed
    const bool aligned = !g_no_aligned && ((reinterpret_cast<uintptr_t>(p.data) | p.chunk_stride | p.bpc) & 15u) == 0;
    // full windows back to back, a whole number of load groups each, digests storable as aligned words
    const bool gapless = aligned && !g_no_flat && p.chunk_len % p.bpc == 0 &&
                         (p.chunk_stride == p.chunk_len || p.n_windows == p.wpc) &&
                         (mode != kModeCompute || (reinterpret_cast<uintptr_t>(p.sums) & 3u) == 0);
    if (gapless) {
      int fl = lpw, fu = (p.bpc <= 256 && p.bpc % (32u * lpw) == 0) ? 2 : 1;
      if (fl == 16 && p.bpc == 8192) fu = 2;  // measured: 6.82 against 6.73 TB/s with one stripe per load group
      int fp = 1;
      if (g_force_flat > 0) {
        fl = g_force_flat / 8;
        fu = g_force_flat % 8;
      }
      // Measured on MI355X (16 GiB batches, profiles/r02_crc_phases_sweep.log): four phases win where the windows
      // of the chip's wavefronts would otherwise all be read at the same offset at the same time and that offset's
      // address bits matter -- 6.94 against 6.84 TB/s at 16 KiB windows (8 lanes), 6.91 against 6.82 at 32 KiB
      // (32 lanes); at 4 and 8 KiB the merge of head and tail costs what the reads gain, at 64 KiB nothing is gained.
      if ((fl == 8 && p.bpc == 16384) || (fl == 32 && p.bpc == 32768)) fp = fu == 1 ? 4 : 1;
      if (g_force_phases == 4) fp = 4;
      if (g_force_phases == 1) fp = 1;
      // four phases need the operators of bpc / 2 and bpc / 4 bytes (nibble tables reach 16 KiB) and whole load
      // groups per quarter window
      if (fp == 4 && ((p.bpc & (p.bpc - 1)) != 0 || p.bpc > 32768 || p.bpc < 64u * fl * fu)) fp = 1;
      // Compute at windows of 4 KiB and less holds its digests in registers until a wavefront's reads are done (measured
      // on MI355X: verify, which writes none, runs at 6.5 / 7.0 / 7.1 / 7.0 TB/s at 512 B / 1 / 2 / 4 KiB where compute with
      // its stores in the stream ran at 6.0 / 6.3 / 6.65 / 6.7; from 8 KiB on the two are level, but for 16 KiB, where
      // the phased form reads faster than its neighbours and the stores showed again: 6.92 against 7.01).  A wavefront holds
      // flat_hold(lanes) * lanes passes; a larger batch is cut into launches of that many passes over the whole grid.
      if (mode == kModeCompute && g_force_hold >= 0 && (p.bpc <= 4096 || p.bpc == 16384 || g_force_hold > 0) &&
          p.bpc % (16u * fl * fu) == 0) {
        const uint64_t per_pass = (uint64_t)num_cus * kWaves * 2 * (64 / fl);  // windows of one pass of a full grid
        const uint64_t cap = per_pass * (uint64_t)(flat_hold(fl) * fl);
        const uint64_t launches = (p.n_windows + cap - 1) / cap;
        uint64_t share = (p.n_windows + launches - 1) / launches;  // equal shares, whole passes
        share = (share + per_pass - 1) / per_pass * per_pass;
        hipError_t e = hipSuccess;
        for (uint64_t off = 0; off < p.n_windows && e == hipSuccess; off += share) {
          CrcLaunch q = p;  // window g of the gapless layout is data + g * bpc, its digest sums + 4 g
          q.data = p.data + off * p.bpc;
          q.sums = p.sums + off * 4;
          q.n_windows = p.n_windows - off < share ? p.n_windows - off : share;
          const uint64_t per = (uint64_t)kWaves * 2 * (64 / fl);
          uint64_t fb = (q.n_windows + per - 1) / per;
          if (fb > (uint64_t)num_cus) fb = num_cus;
          e = launch_flat_hold(q, fl, fu, fp, (int)fb, stream);
          if (e == hipErrorInvalidValue && off == 0) break;  // that (lanes, group) pair is not built in this form
        }
        if (e != hipErrorInvalidValue) return e;
        (void)hipGetLastError();
      }
      if (p.bpc % (16u * fl * fu) == 0) {
        const uint64_t per = (uint64_t)kWaves * 2 * (64 / fl);
        uint64_t fb = (p.n_windows + per - 1) / per;
        if (fb > (uint64_t)num_cus) fb = num_cus;
        const hipError_t e = mode == kModeCompute ? launch_flat<kModeCompute>(p, fl, fu, fp, (int)fb, stream)
                                                  : launch_flat<kModeVerify>(p, fl, fu, fp, (int)fb, stream);
        if (e != hipErrorInvalidValue) return e;
        (void)hipGetLastError();  // that (lanes, group) pair is not built: use the general kernel
      }
    }
    if (aligned)
      return mode == kModeCompute ? launch_subwave<kModeCompute, true>(p, lpw, (int)nb, stream)
                                  : launch_subwave<kModeVerify, true>(p, lpw, (int)nb, stream);
    return mode == kModeCompute ? launch_subwave<kModeCompute, false>(p, lpw, (int)nb, stream)
                                : launch_subwave<kModeVerify, false>(p, lpw, (int)nb, stream);
  }
  const uint64_t per_block = kWaves / group;
  uint64_t blocks = (p.n_windows + per_block - 1) / per_block;
  if (blocks > (uint64_t)num_cus) blocks = num_cus;  // 135 KiB of LDS: one block per CU, persistent
  // Default: one stripe ahead.  Deeper prefetch measured slower on MI355X (the kernel is limited by
  // the clock the chip holds under this LDS + HBM load, and more registers in flight cost more).
  switch (g_force_pf) {
    case 2:
      return mode == kModeCompute ? launch_mode<kModeCompute, 2>(p, group, (int)blocks, stream)
                                  : launch_mode<kModeVerify, 2>(p, group, (int)blocks, stream);
    case 3:
      return mode == kModeCompute ? launch_mode<kModeCompute, 3>(p, group, (int)blocks, stream)
                                  : launch_mode<kModeVerify, 3>(p, group, (int)blocks, stream);
    case 4:
      return mode == kModeCompute ? launch_mode<kModeCompute, 4>(p, group, (int)blocks, stream)
                                  : launch_mode<kModeVerify, 4>(p, group, (int)blocks, stream);
    default:
      return mode == kModeCompute ? launch_mode<kModeCompute, 1>(p, group, (int)blocks, stream)
                                  : launch_mode<kModeVerify, 1>(p, group, (int)blocks, stream);
  }
}

}  // namespace ozcs
