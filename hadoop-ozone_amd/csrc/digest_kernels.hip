// SHA-256 / MD5 per-window kernels for gfx950 (MI355X).
//
// What is computed: for every bytes-per-checksum window, MessageDigest("SHA-256") or
// MessageDigest("MD5") over the window's bytes (Checksum.java:46-60, 83-84), i.e. FIPS 180-4
// section 6.2 / RFC 1321 section 3, digest bytes in the standard order.
//
// How: both hashes are Merkle-Damgard chains, so one window cannot be split; the parallelism is
// across windows.  One lane owns one window (64 windows per wavefront, one wavefront per
// workgroup).  Per step the wavefront fetches the next 64-byte message block of each of its 64
// windows with coalesced 16-byte loads (four lanes per window), parks them in LDS in rows of
// 80 bytes (64 + 16 pad: both the 128-bit stores of the loader role and the 128-bit loads of the
// owner role are then conflict-free), and every lane runs the compression function on its own
// block while the loads of the following block are in flight.  Blocks that are plain data in all
// 64 windows run in a loop without bounds checks; padding blocks are built in registers.  The
// kernels are integer-ALU bound, not HBM bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "crc_kernels.h"  // kModeCompute / kModeVerify
#include "digest_kernels.h"

namespace ozcs {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u32x4_unaligned* gptr_u4u;
typedef const __attribute__((address_space(1))) uint8_t* gptr_u8;

// One v_alignbit_b32 each (the shift-shift-or form is not always folded into it).
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }
// Three-input bitwise functions in one instruction (v_bitop3_b32; the table is the function of 0xF0, 0xCC, 0xAA).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
// Ch(e,f,g) = (e & f) ^ (~e & g) and Maj(a,b,c): written as source expressions they compile to a
// v_bitop3 plus an and and an xor each round; spelled out they are one instruction each.
__device__ __forceinline__ uint32_t choose(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t majority(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// ------------------------------------------------------------------ SHA-256 (FIPS 180-4, 6.2.2)
__constant__ const uint32_t kK256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

struct Sha256 {
  static constexpr int kWords = 8;
  static constexpr bool kBigEndian = true;
  static constexpr int kStepBlocks = 1;  // 64-byte blocks fetched per loader step (Sha256Lines: 2)
  static __device__ __forceinline__ void init(uint32_t* h) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
    h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
  }
  static __device__ __forceinline__ void compress(uint32_t* h, uint32_t* w) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      uint32_t wi;
      if (i < 16) {
        wi = w[i];
      } else {
        const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
        const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
        const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
        wi = w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
      }
      const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
      const uint32_t ch = choose(e, f, g);
      const uint32_t t1 = hh + S1 + ch + kK256[i] + wi;
      const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
      const uint32_t mj = majority(a, b, c);
      const uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
};

// ------------------------------------------------------------------ MD5 (RFC 1321, 3.4)
__constant__ const uint32_t kKmd5[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};

struct Md5 {
  static constexpr int kWords = 4;
  static constexpr bool kBigEndian = false;
  // 64-byte blocks fetched per loader step.  Measured on MI355X: 2 (whole 128-byte lines) gives 4.05 / 4.35 TB/s
  // at 16 / 64 KiB windows against 3.95 / 4.18 with 1, but 2.45 against 3.04 TB/s at 1 MiB windows (32 GiB),
  // where one wavefront per SIMD runs a long chain; HBM traffic equals the input size either way.
  static constexpr int kStepBlocks = 1;
  static __device__ __forceinline__ void init(uint32_t* h) {
    h[0] = 0x67452301; h[1] = 0xefcdab89; h[2] = 0x98badcfe; h[3] = 0x10325476;
  }
  static __device__ __forceinline__ void compress(uint32_t* h, uint32_t* m) {
    constexpr int S[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      uint32_t f;
      int g;
      if (i < 16) { f = (b & c) | (~b & d); g = i; }
      else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = xor3(b, c, d); g = (3 * i + 5) & 15; }
      else { f = c ^ (b | ~d); g = (7 * i) & 15; }
      const uint32_t t = d;
      d = c;
      c = b;
      b = b + rotl(a + f + kKmd5[i] + m[g], S[i >> 4][i & 3]);
      a = t;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
  }
};

// MD5 with whole 128-byte lines per loader step.  With 64-byte steps every line is touched by two consecutive
// steps; when the chip is full of wavefronts the second touch often misses the L2 and the line comes in again
// (rocprofv3: 1.24 x the input size fetched at 16 KiB windows).  128-byte steps fetch every line once and are
// 2.5-4 % faster there, but slower when wavefronts are scarce and each runs a long chain alone on its SIMD
// (2.45 against 3.04 TB/s at 1 MiB windows, 32 GiB), so the launcher picks by window count.
struct Md5Lines : Md5 {
  static constexpr int kStepBlocks = 2;
};

// SHA-256 likewise: 1.43 x the input size fetched at 16 KiB windows with 64-byte steps, the input size with
// 128-byte steps, which are 1 % faster at 16 KiB and 5 % at 64 KiB windows (8 GiB batches) although the kernel
// sits at the vector-issue ceiling either way.
struct Sha256Lines : Sha256 {
  static constexpr int kStepBlocks = 2;
};

constexpr int kLanes = 64;
// LDS row of one window slot: PB 64-byte blocks + 16 bytes of pad (80 or 144 bytes: the 128-bit stores of
// the loader role and the 128-bit loads of the owner role are then both conflict-free).
template <int PB>
constexpr int row_bytes() { return PB * 64 + 16; }
constexpr int kRowBytes = row_bytes<1>();

// Loader role of a wavefront.  A step covers PB consecutive 64-byte blocks of each of the 64 windows:
// 4*PB lanes fetch the 16-byte pieces of one window slot, so one instruction reads 16/PB windows x 64*PB
// contiguous bytes and 4*PB instructions read the step.  PB = 2 reads whole 128-byte lines; with PB = 1
// each line is touched by two separate steps and throughput stops at 3.8 TB/s, which only MD5 can exceed.
template <int PB>
struct BlockLoader {
  static constexpr int kInstr = 4 * PB;        // load instructions per step
  static constexpr int kLanesPerSlot = 4 * PB;
  static constexpr int kSlotsPerInstr = kLanes / kLanesPerSlot;
  static constexpr uint32_t kStepBytes = 64u * PB;
  uint64_t lstart[kInstr];
  uint32_t llen[kInstr];
  uint32_t piece;
  int lane;

  __device__ __forceinline__ void init(int lane_, bool valid, uint64_t start, uint32_t len) {
    lane = lane_;
#pragma unroll
    for (int i = 0; i < kInstr; i++) {
      const int slot = kSlotsPerInstr * i + lane / kLanesPerSlot;
      const uint32_t lo = __shfl((uint32_t)start, slot, 64), hi = __shfl((uint32_t)(start >> 32), slot, 64);
      lstart[i] = ((uint64_t)hi << 32) | lo;
      llen[i] = __shfl(valid ? len : 0u, slot, 64);
    }
    piece = (lane % kLanesPerSlot) * 16;
  }

  __device__ __forceinline__ void fetch(uint32_t step, u32x4* r) const {
#pragma unroll
    for (int i = 0; i < kInstr; i++) {
      const uint64_t o = (uint64_t)step * kStepBytes + piece;
      u32x4 v = {0, 0, 0, 0};
      if (o + 16 <= llen[i]) {
        v = *reinterpret_cast<gptr_u4u>(lstart[i] + o);
      } else if (o < llen[i]) {  // last, partial piece of a window: byte loads, never past its end
        uint32_t t[4] = {0, 0, 0, 0};
        const uint32_t nb = llen[i] - (uint32_t)o;
#pragma unroll
        for (uint32_t k = 0; k < 15; k++)
          if (k < nb) t[k >> 2] |= (uint32_t)(*reinterpret_cast<gptr_u8>(lstart[i] + o + k)) << (8 * (k & 3));
        v = (u32x4){t[0], t[1], t[2], t[3]};
      }
      r[i] = v;
    }
  }

  // fetch() for a step that lies wholly inside every one of the wavefront's 64 windows: no bounds to check.
  // Plain (cached) loads: with PB = 1 a 128-byte line serves two consecutive steps, and non-temporal loads
  // made the second one fetch it from HBM again.
  __device__ __forceinline__ void fetch_inside(uint32_t step, u32x4* r) const {
    const uint64_t o = (uint64_t)step * kStepBytes + piece;
#pragma unroll
    for (int i = 0; i < kInstr; i++) r[i] = *reinterpret_cast<gptr_u4u>(lstart[i] + o);
  }

  // Parks the fetched pieces in the rows of `stage` (row = window slot).
  __device__ __forceinline__ void park(char* stage, const u32x4* r) const {
#pragma unroll
    for (int i = 0; i < kInstr; i++)
      *reinterpret_cast<u32x4*>(stage + (kSlotsPerInstr * i + lane / kLanesPerSlot) * row_bytes<PB>() + piece) = r[i];
  }
};

// Owner role: reads a 64-byte block at `row` (this lane's row of `stage`, plus the block's offset in the
// step) as 16 words in the hash's word order.
template <class H>
__device__ __forceinline__ void data_words(const char* row, uint32_t* w) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(row + 16 * j);
    w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
  if (H::kBigEndian) {
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = __builtin_bswap32(w[j]);
  }
}

// data_words plus the padding (0x80, zeros, bit length) where the message ends.
template <class H>
__device__ __forceinline__ void block_words(const char* row, uint32_t blk, uint32_t len, uint32_t nblk, uint32_t* w) {
  data_words<H>(row, w);
  const int64_t rem = (int64_t)len - (int64_t)blk * 64;  // message bytes from this block on
  if (rem < 64) {  // this block holds the end of the message and/or padding
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int64_t rj = rem - 4 * j;  // message bytes in word j
      if (rj >= 4) continue;
      if (rj < 0) { w[j] = 0; continue; }
      const int sh = 8 * (int)rj;
      if (H::kBigEndian) w[j] = (rj == 0 ? 0u : (w[j] & (0xFFFFFFFFu << (32 - sh)))) | (0x80000000u >> sh);
      else w[j] = (w[j] & ((1u << sh) - 1u)) | (0x80u << sh);
    }
    if (blk + 1 == nblk) {
      const uint64_t bits = (uint64_t)len * 8;
      if (H::kBigEndian) { w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; }
      else { w[14] = (uint32_t)bits; w[15] = (uint32_t)(bits >> 32); }
    }
  }
}

__device__ __forceinline__ uint32_t num_blocks(bool valid, uint32_t len) {
  return valid ? (uint32_t)(((uint64_t)len + 8) / 64 + 1) : 0u;  // data + padding blocks
}

// Hashes this lane's window [start, start+len) (nothing when !valid); h receives the digest words in
// output byte order.  Must be called by all 64 lanes of the one-wavefront workgroup together.  `stage` holds
// 64 rows of row_bytes<H::kStepBlocks>().
template <class H>
__device__ __forceinline__ void digest_core(char* stage, int lane, bool valid, uint64_t start, uint32_t len, uint32_t* h) {
  constexpr int PB = H::kStepBlocks;
  // A lane without a window hashes a copy of the first valid lane's window (its result is never stored):
  // the wavefront runs in lockstep anyway, and with every lane on a real window a partly filled wavefront
  // keeps the check-free loop below -- it took 23 ms instead of 11 ms per 1 MiB MD5 window without this.
  {
    const uint64_t vm = __ballot(valid);
    const int src = vm ? __builtin_ctzll(vm) : 0;
    const uint32_t slo = __shfl((uint32_t)start, src, 64), shi = __shfl((uint32_t)(start >> 32), src, 64);
    const uint32_t sl = __shfl(len, src, 64);
    if (!valid && vm) {
      start = ((uint64_t)shi << 32) | slo;
      len = sl;
      valid = true;
    }
  }
  const uint32_t nblk = num_blocks(valid, len);
  BlockLoader<PB> ld;
  ld.init(lane, valid, start, len);
  H::init(h);
  // Steps that are plain data in every window of the wavefront (none if a lane has no window) run in a
  // loop of their own without bounds checks or padding logic: that bookkeeping is a third of the
  // instructions of an MD5 block.
  uint32_t inside = valid ? len / (64u * PB) : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = __shfl_xor(inside, o, 64);
    inside = other < inside ? other : inside;
  }
  const char* row = stage + lane * row_bytes<PB>();
  u32x4 r[BlockLoader<PB>::kInstr];
  uint32_t step = 0;
  if (inside > 0) ld.fetch_inside(0, r);
  else ld.fetch(0, r);
  for (; step + 1 < inside; step++) {  // steps `step` and `step + 1` are inside everywhere
    __syncthreads();  // the previous step has been read by every owner
    ld.park(stage, r);
    __syncthreads();
    ld.fetch_inside(step + 1, r);  // in flight during the compression below
#pragma unroll
    for (int k = 0; k < PB; k++) {
      uint32_t w[16];
      data_words<H>(row + 64 * k, w);
      H::compress(h, w);
    }
  }
  for (; __ballot(step * PB < nblk) != 0ull; step++) {
    __syncthreads();
    ld.park(stage, r);
    __syncthreads();
    ld.fetch(step + 1, r);
#pragma unroll
    for (int k = 0; k < PB; k++) {
      const uint32_t blk = step * PB + k;
      uint32_t w[16];
      block_words<H>(row + 64 * k, blk, len, nblk, w);
      if (blk < nblk) H::compress(h, w);
    }
  }
  if (H::kBigEndian) {
#pragma unroll
    for (int j = 0; j < H::kWords; j++) h[j] = __builtin_bswap32(h[j]);
  }
}

// SHA-256 with the message schedule and the compression function on two wavefronts of one workgroup.
// A lone wavefront issues one VALU instruction per ~4.6 cycles however much instruction-level parallelism
// it has, so when windows are too few to put two wavefronts on every SIMD the only way to use more of the
// chip is more wavefronts per window set.  The schedule W[16..63] of a block depends on the block alone,
// not on the chain: wavefront 0 loads blocks, builds the padding and expands W (about 40 % of the
// instructions), wavefront 1 runs the 64 rounds.  W travels through LDS, 16 KiB per block, double-buffered,
// one workgroup barrier per block: while wavefront 1 compresses block b-1, wavefront 0 expands block b.
// A workgroup is four wavefronts = two such pairs (window sets 0 and 1): the hardware spreads the four
// wavefronts of one workgroup over the CU's four SIMDs, which two separate two-wavefront workgroups on one
// CU do not get (measured: 53 ms instead of 34 ms per 1 MiB window).  Wavefronts 0/1 expand sets 0/1,
// wavefronts 2/3 compress them; lane l of wavefronts s and s+2 owns the same window.  h is valid in
// wavefronts 2 and 3 only.  `stage`, `wbuf` and `sync` are this workgroup's; all 256 threads call.
constexpr int kWBufBytes = 64 * kLanes * 4;
constexpr int kSplitThreads = 4 * kLanes;
constexpr int kSplitWindows = 2 * kLanes;  // windows per workgroup

__device__ __forceinline__ void sha256_split_core(char* stage_all, char* wbuf_all, uint32_t* sync, int tid, bool valid,
                                                  uint64_t start, uint32_t len, uint32_t* h) {
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int set = wave & 1;
  const bool compressor = wave >= 2;
  char* stage = stage_all + set * (kLanes * kRowBytes);
  char* wbuf = wbuf_all + set * (2 * kWBufBytes);
  const uint32_t nblk = num_blocks(valid, len);
  uint32_t nmax = nblk;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = __shfl_xor(nmax, o, 64);
    nmax = other > nmax ? other : nmax;
  }
  if (lane == 0) sync[wave] = nmax;  // every wavefront must pass the same number of barriers
  __syncthreads();
  nmax = sync[0] > sync[1] ? sync[0] : sync[1];
  BlockLoader<1> ld;
  u32x4 r[4];
  if (!compressor) {
    ld.init(lane, valid, start, len);
    ld.fetch(0, r);
  }
  Sha256::init(h);
  for (uint32_t blk = 0; blk <= nmax; blk++) {
    __syncthreads();  // W of block blk-1 is complete; the buffer of block blk-2 has been consumed
    if (!compressor) {
      if (blk < nmax) {
        ld.park(stage, r);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // stage is private to this wavefront; its DS
        __builtin_amdgcn_wave_barrier();                        // operations execute in issue order
        ld.fetch(blk + 1, r);
        uint32_t w[16];
        block_words<Sha256>(stage + lane * kRowBytes, blk, len, nblk, w);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        char* dst = wbuf + (blk & 1) * kWBufBytes + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<u32x4*>(dst + q * (kLanes * 16)) = (u32x4){w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
#pragma unroll
        for (int i = 16; i < 64; i++) {
          const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
          const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
          const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
          w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
          if ((i & 3) == 3) {
            const int b = i & 12;  // ring position of the four words just completed
            *reinterpret_cast<u32x4*>(dst + (i >> 2) * (kLanes * 16)) = (u32x4){w[b], w[b + 1], w[b + 2], w[b + 3]};
          }
        }
      }
    } else if (blk >= 1 && blk - 1 < nblk) {
      const char* src = wbuf + ((blk - 1) & 1) * kWBufBytes + lane * 16;
      uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
      // W is read eight rounds ahead of its use: the loads of rounds 8(p+1)..8(p+1)+7 are issued before
      // the rounds 8p..8p+7 run (the scheduling barrier keeps them there), so the LDS latency is covered.
      u32x4 v0 = *reinterpret_cast<const u32x4*>(src), v1 = *reinterpret_cast<const u32x4*>(src + kLanes * 16);
#pragma unroll
      for (int p = 0; p < 8; p++) {
        u32x4 n0 = v0, n1 = v1;
        if (p < 7) {
          n0 = *reinterpret_cast<const u32x4*>(src + (2 * p + 2) * (kLanes * 16));
          n1 = *reinterpret_cast<const u32x4*>(src + (2 * p + 3) * (kLanes * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t wq[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
          const uint32_t ch = choose(e, f, g);
          const uint32_t t1 = hh + S1 + ch + kK256[8 * p + k] + wq[k];
          const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
          const uint32_t mj = majority(a, b, c);
          const uint32_t t2 = S0 + mj;
          hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        v0 = n0;
        v1 = n1;
      }
      h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) h[j] = __builtin_bswap32(h[j]);
}

template <class H>
__device__ __forceinline__ void store_digest(uint8_t* out, const uint32_t* h) {
  constexpr int kDigest = H::kWords * 4;
  if ((reinterpret_cast<uintptr_t>(out) & 3) == 0) {
#pragma unroll
    for (int j = 0; j < H::kWords; j++) reinterpret_cast<uint32_t*>(out)[j] = h[j];
  } else {
#pragma unroll
    for (int j = 0; j < kDigest; j++) out[j] = (uint8_t)(h[j >> 2] >> (8 * (j & 3)));
  }
}

// Uniform batch: chunk c at data + c*chunk_stride, wpc windows per chunk.
template <class H, int MODE>
__global__ __launch_bounds__(kLanes) void digest_windows_kernel(const DigestLaunch p) {
  __shared__ __attribute__((aligned(16))) char stage[kLanes * row_bytes<H::kStepBlocks>()];
  const int lane = threadIdx.x;
  const uint64_t g = (uint64_t)blockIdx.x * kLanes + lane;
  const bool valid = g < p.n_windows;
  uint64_t chunk = 0, wi = 0, start = 0;
  uint32_t len = 0;
  if (valid) {
    chunk = g / p.wpc;
    wi = g - chunk * p.wpc;
    const uint64_t woff = wi * (uint64_t)p.bpc;
    const uint64_t left = p.chunk_len - woff;
    len = (uint32_t)(left < p.bpc ? left : p.bpc);
    start = reinterpret_cast<uintptr_t>(p.data) + chunk * p.chunk_stride + woff;
  }
  uint32_t h[H::kWords];
  digest_core<H>(stage, lane, valid, start, len, h);
  if (!valid) return;
  constexpr int kDigest = H::kWords * 4;
  if (MODE == kModeCompute) {
    store_digest<H>(p.sums + g * kDigest, h);
  } else {
    const uint64_t si = p.start_index + wi;
    bool bad = si >= p.n_expected;
    if (!bad) {
      const uint8_t* ex = p.expected + chunk * p.expected_stride + si * kDigest;
#pragma unroll
      for (int j = 0; j < kDigest; j++) bad |= ex[j] != (uint8_t)(h[j >> 2] >> (8 * (j & 3)));
    }
    if (bad) atomicMin(p.first_mismatch + chunk, (unsigned long long)wi);
  }
}

// Windows given one by one (start, length, where the digest goes): any number of unrelated small chunks in one
// launch, the list in device memory.
template <class H>
__global__ __launch_bounds__(kLanes) void digest_desc_kernel(const WindowDesc* desc, uint32_t n) {
  __shared__ __attribute__((aligned(16))) char stage[kLanes * row_bytes<H::kStepBlocks>()];
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x * kLanes + lane;
  const bool valid = g < n;
  uint64_t start = 0;
  uint32_t len = 0;
  uint8_t* out = nullptr;
  if (valid) {
    const WindowDesc d = desc[g];
    start = reinterpret_cast<uintptr_t>(d.data);
    len = d.len;
    out = d.out;
  }
  uint32_t h[H::kWords];
  digest_core<H>(stage, lane, valid, start, len, h);
  if (valid) store_digest<H>(out, h);
}

// The split SHA-256 forms of the two kernels above (workgroup = 256 threads = 128 windows).
template <int MODE>
__global__ __launch_bounds__(kSplitThreads) void sha256_split_windows_kernel(const DigestLaunch p) {
  __shared__ __attribute__((aligned(16))) char stage[2 * kLanes * kRowBytes];
  __shared__ __attribute__((aligned(16))) char wbuf[4 * kWBufBytes];
  __shared__ uint32_t sync[4];
  const int tid = threadIdx.x;
  const uint64_t g = (uint64_t)blockIdx.x * kSplitWindows + (tid & (kSplitWindows - 1));
  const bool valid = g < p.n_windows;
  uint64_t chunk = 0, wi = 0, start = 0;
  uint32_t len = 0;
  if (valid) {
    chunk = g / p.wpc;
    wi = g - chunk * p.wpc;
    const uint64_t woff = wi * (uint64_t)p.bpc;
    const uint64_t left = p.chunk_len - woff;
    len = (uint32_t)(left < p.bpc ? left : p.bpc);
    start = reinterpret_cast<uintptr_t>(p.data) + chunk * p.chunk_stride + woff;
  }
  uint32_t h[8];
  sha256_split_core(stage, wbuf, sync, tid, valid, start, len, h);
  if (!valid || tid < kSplitWindows) return;
  if (MODE == kModeCompute) {
    store_digest<Sha256>(p.sums + g * 32, h);
  } else {
    const uint64_t si = p.start_index + wi;
    bool bad = si >= p.n_expected;
    if (!bad) {
      const uint8_t* ex = p.expected + chunk * p.expected_stride + si * 32;
#pragma unroll
      for (int j = 0; j < 32; j++) bad |= ex[j] != (uint8_t)(h[j >> 2] >> (8 * (j & 3)));
    }
    if (bad) atomicMin(p.first_mismatch + chunk, (unsigned long long)wi);
  }
}

// Job list: any number of unrelated chunks (own pointer, length, window size, output) in one launch, so that a
// handful of windows per chunk still adds up to full wavefronts.  The list lies in device-visible memory,
// sorted by first_window; the job of window g is found by bisection.
__device__ __forceinline__ void job_window(const DigestJob* jobs, uint32_t n_jobs, uint32_t g, uint32_t digest_bytes,
                                           uint64_t& start, uint32_t& len, uint8_t*& out) {
  uint32_t lo = 0, hi = n_jobs;  // jobs[lo].first_window <= g < jobs[hi].first_window (hi == n_jobs: no bound)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobs[mid].first_window <= g) lo = mid;
    else hi = mid;
  }
  const DigestJob job = jobs[lo];
  const uint64_t wi = g - job.first_window;
  const uint64_t woff = wi * (uint64_t)job.bpc;
  const uint64_t left = job.len - woff;
  len = (uint32_t)(left < job.bpc ? left : job.bpc);
  start = reinterpret_cast<uintptr_t>(job.data) + woff;
  out = job.sums + wi * digest_bytes;
}

template <class H>
__global__ __launch_bounds__(kLanes) void digest_jobs_kernel(const DigestJob* jobs, uint32_t n_jobs, uint32_t n_windows) {
  __shared__ __attribute__((aligned(16))) char stage[kLanes * row_bytes<H::kStepBlocks>()];
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x * kLanes + lane;
  const bool valid = g < n_windows;
  uint64_t start = 0;
  uint32_t len = 0;
  uint8_t* out = nullptr;
  if (valid) job_window(jobs, n_jobs, g, H::kWords * 4, start, len, out);
  uint32_t h[H::kWords];
  digest_core<H>(stage, lane, valid, start, len, h);
  if (valid) store_digest<H>(out, h);
}

__global__ __launch_bounds__(kSplitThreads) void sha256_split_jobs_kernel(const DigestJob* jobs, uint32_t n_jobs,
                                                                              uint32_t n_windows) {
  __shared__ __attribute__((aligned(16))) char stage[2 * kLanes * kRowBytes];
  __shared__ __attribute__((aligned(16))) char wbuf[4 * kWBufBytes];
  __shared__ uint32_t sync[4];
  const int tid = threadIdx.x;
  const uint32_t g = blockIdx.x * kSplitWindows + (tid & (kSplitWindows - 1));
  const bool valid = g < n_windows;
  uint64_t start = 0;
  uint32_t len = 0;
  uint8_t* out = nullptr;
  if (valid) job_window(jobs, n_jobs, g, 32, start, len, out);
  uint32_t h[8];
  sha256_split_core(stage, wbuf, sync, tid, valid, start, len, h);
  if (valid && tid >= kSplitWindows) store_digest<Sha256>(out, h);
}

// Tuning override for experiments (read once): OZCS_MD5_LINES = 0 never, 1 always whole-line MD5 loader steps;
// unset = by window count.
const int g_md5_lines = getenv("OZCS_MD5_LINES") ? atoi(getenv("OZCS_MD5_LINES")) : -1;

// Tuning override for experiments (read once): OZCS_SHA_LINES = 0 never, 1 always whole-line SHA-256 loader steps.
const int g_sha_lines = getenv("OZCS_SHA_LINES") ? atoi(getenv("OZCS_SHA_LINES")) : -1;

// Tuning override for experiments (read once): OZCS_SHA_SPLIT = 0 never, 1 always; unset = by window count.
const int g_sha_split = getenv("OZCS_SHA_SPLIT") ? atoi(getenv("OZCS_SHA_SPLIT")) : -1;

// The split form pays while SIMDs are idle.  These integer instructions issue at one wave64 instruction
// per 4 cycles per SIMD, so a single wavefront already saturates its SIMD: with up to one split workgroup
// per CU (128 windows) every window set gets two SIMDs instead of one (34 ms instead of 50 ms per 1 MiB
// window); beyond that the one-wavefront form executes fewer instructions in total.
bool use_sha_split(uint64_t n_windows, int num_cus) {
  if (g_sha_split == 0) return false;
  if (g_sha_split == 1) return true;
  return n_windows <= (uint64_t)num_cus * kSplitWindows;
}

}  // namespace

hipError_t digest_launch(const DigestLaunch& p, int mode, int num_cus, hipStream_t stream) {
  if (p.n_windows == 0) return hipSuccess;
  if (p.algo == kAlgoSha256 && use_sha_split(p.n_windows, num_cus)) {
    const uint64_t nb = (p.n_windows + kSplitWindows - 1) / kSplitWindows;
    if (mode == kModeCompute) sha256_split_windows_kernel<kModeCompute><<<dim3((unsigned)nb), dim3(kSplitThreads), 0, stream>>>(p);
    else sha256_split_windows_kernel<kModeVerify><<<dim3((unsigned)nb), dim3(kSplitThreads), 0, stream>>>(p);
    return hipGetLastError();
  }
  const uint64_t blocks = (p.n_windows + kLanes - 1) / kLanes;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(kLanes);
  if (p.algo == kAlgoSha256) {
    // as for MD5 below: whole-line loader steps once every SIMD has at least two wavefronts
    const bool lines = g_sha_lines < 0 ? blocks >= (uint64_t)num_cus * 8 : g_sha_lines != 0;
    if (lines) {
      if (mode == kModeCompute) digest_windows_kernel<Sha256Lines, kModeCompute><<<grid, block, 0, stream>>>(p);
      else digest_windows_kernel<Sha256Lines, kModeVerify><<<grid, block, 0, stream>>>(p);
    } else {
      if (mode == kModeCompute) digest_windows_kernel<Sha256, kModeCompute><<<grid, block, 0, stream>>>(p);
      else digest_windows_kernel<Sha256, kModeVerify><<<grid, block, 0, stream>>>(p);
    }
  } else {
    // at least two wavefronts for every SIMD (4 per CU): whole-line loader steps
    const bool lines = g_md5_lines < 0 ? blocks >= (uint64_t)num_cus * 8 : g_md5_lines != 0;
    if (lines) {
      if (mode == kModeCompute) digest_windows_kernel<Md5Lines, kModeCompute><<<grid, block, 0, stream>>>(p);
      else digest_windows_kernel<Md5Lines, kModeVerify><<<grid, block, 0, stream>>>(p);
    } else {
      if (mode == kModeCompute) digest_windows_kernel<Md5, kModeCompute><<<grid, block, 0, stream>>>(p);
      else digest_windows_kernel<Md5, kModeVerify><<<grid, block, 0, stream>>>(p);
    }
  }
  return hipGetLastError();
}

hipError_t digest_launch_desc(const WindowDesc* d_desc, uint32_t n, int algo, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + kLanes - 1) / kLanes), block(kLanes);
  if (algo == kAlgoSha256) digest_desc_kernel<Sha256><<<grid, block, 0, stream>>>(d_desc, n);
  else digest_desc_kernel<Md5><<<grid, block, 0, stream>>>(d_desc, n);
  return hipGetLastError();
}

hipError_t digest_launch_jobs(const DigestJob* d_jobs, uint32_t n_jobs, uint32_t n_windows, int algo, int num_cus,
                                  hipStream_t stream) {
  if (n_windows == 0 || n_jobs == 0) return hipSuccess;
  if (algo == kAlgoSha256 && use_sha_split(n_windows, num_cus)) {
    sha256_split_jobs_kernel<<<dim3((n_windows + kSplitWindows - 1) / kSplitWindows), dim3(kSplitThreads), 0, stream>>>(
        d_jobs, n_jobs, n_windows);
    return hipGetLastError();
  }
  const dim3 grid((n_windows + kLanes - 1) / kLanes), block(kLanes);
  if (algo == kAlgoSha256) digest_jobs_kernel<Sha256><<<grid, block, 0, stream>>>(d_jobs, n_jobs, n_windows);
  else digest_jobs_kernel<Md5><<<grid, block, 0, stream>>>(d_jobs, n_jobs, n_windows);
  return hipGetLastError();
}

}  // namespace ozcs
