// Small CRC windows: would a second workgroup per CU pay?  (gfx950 only; an experiment beside the library, not part of it)
//
// When this was written the library's CRC kernels all kept their four byte tables in LDS in 32 copies (128 KiB), so that
// lane l always reads bank l % 32; with the fold's tables that was 139 KiB and one workgroup of 16 wavefronts per CU, four
// per SIMD.  Windows of a few hundred bytes spend most of a pass in the per-window fold (LDS round trips that wait for one
// another) and in the memory latency a pass begins with; more wavefronts per SIMD would cover both.
//
// 16 copies need not cost bank conflicts: a lookup instruction serves byte k of an accumulator through table k in ALL
// lanes only by convention -- the four results are XORed, so their order is free.  Here lanes 0-15 of each half of the
// wavefront look up (table 0, byte 0) while lanes 16-31 look up (table 1, byte 1), and the next instruction swaps them;
// table t lives in banks 16 * (t & 1) + copy.  Every instruction still reads 32 different banks per half, the tables
// take 64 KiB, the whole set 75 KiB, and two workgroups fit a CU.  The selector of the address v_perm becomes a lane
// constant in a register instead of an immediate.
//
// This program runs the library's gapless small-window kernel (compute form: digests held in registers until the
// wavefront's reads are done, optionally the fold one pass late) in both table layouts, with 1024-, 768- and 512-thread
// workgroups, one or two per CU, checks sampled digests against a bitwise CRC on the host, and prints the rates.
// Outcome on MI355X: the 16-copy layout is free, the second workgroup buys nothing -- and the room spent on 16 copies of
// the unit merge's byte tables instead (one copy before: bank conflicts about 3.5 deep) gives 256-byte windows 9 %.  The
// library's lanes-per-window kernels took that last form.
//
// usage: crc_occupancy_bench <bytes per window: 128..4096> [GiB per step, default 8]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../hadoop-ozone_amd/csrc/gf2_crc.h"

using namespace ozcs;

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

__constant__ PolyBasis c_basis = make_poly_basis(kPolyCrc32);

constexpr int kNibOps = 14;
constexpr uint32_t kNibOpBytes = 8u * 16u * 4u;

template <int COPIES, int ZC>
struct Lay {
  static constexpr uint32_t kMain = 4u * 256u * COPIES * 4u;  // 128 KiB or 64 KiB: a multiple of 64 KiB either way
  static constexpr uint32_t kZ4 = kMain;
  static constexpr uint32_t kNib = kZ4 + 4096u * ZC;  // the "advance 4 bytes" byte tables: one copy, or 16 laid out like the main tables
  static constexpr uint32_t kBytes = kNib + kNibOps * kNibOpBytes;
};

__device__ __forceinline__ uint32_t lds_u32(const char* base, uint32_t off) {
  return *reinterpret_cast<const uint32_t*>(base + off);
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

template <uint32_t kNibOff>
__device__ __forceinline__ uint32_t nib_apply(const char* lds, int op, uint32_t x) {
  const uint32_t m0 = (x << 2) & 0x3c3c3c3cu;
  const uint32_t m1 = (x >> 2) & 0x3c3c3c3cu;
  const uint32_t off = (uint32_t)op * kNibOpBytes;
  uint32_t t[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t[2 * k] = lds_u32(lds, __builtin_amdgcn_perm(m0, kNibOff, 0x0c020104u + k) + (off + (2 * k) * 64u));
    t[2 * k + 1] = lds_u32(lds, __builtin_amdgcn_perm(m1, kNibOff, 0x0c020104u + k) + (off + (2 * k + 1) * 64u));
  }
  return xor3(xor3(t[0], t[1], t[2]), xor3(t[3], t[4], t[5]), t[6] ^ t[7]);
}

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

// Lane constants of the main lookups.
template <int COPIES>
struct LaneC;
template <>
struct LaneC<32> {  // entry b of table t, copy r at (t >> 1) * 64 KiB + b * 256 + (t & 1) * 128 + r * 4
  uint32_t lb01, lb23;
  __device__ __forceinline__ explicit LaneC(int lane) : lb01((uint32_t)(lane & 31) * 4u), lb23(lb01 + 65536u) {}
  __device__ __forceinline__ uint32_t gather(const char* lds, uint32_t x, uint32_t d) const {
    const uint32_t t0 = lds_u32(lds, __builtin_amdgcn_perm(x, lb01, 0x0c020400u));
    const uint32_t t1 = lds_u32(lds, __builtin_amdgcn_perm(x, lb01, 0x0c020500u) + 128u);
    const uint32_t t2 = lds_u32(lds, __builtin_amdgcn_perm(x, lb23, 0x0c020600u));
    const uint32_t t3 = lds_u32(lds, __builtin_amdgcn_perm(x, lb23, 0x0c020700u) + 128u);
    return xor3(xor3(t0, t1, t2), t3, d);
  }
};
template <>
struct LaneC<16> {  // entry b of table t, copy r at b * 256 + t * 64 + r * 4: table t in banks 16 * (t & 1) + r
  uint32_t ca, cb, sa01, sb01, sa23, sb23;
  __device__ __forceinline__ explicit LaneC(int lane) {
    const uint32_t r4 = (uint32_t)(lane & 15) * 4u;
    const bool hi = (lane & 16) != 0;
    ca = r4 + (hi ? 64u : 0u);  // first instruction of a pair: the even table below, the odd table above
    cb = r4 + (hi ? 0u : 64u);  // second instruction: the other way round
    sa01 = hi ? 0x0c020500u : 0x0c020400u;
    sb01 = hi ? 0x0c020400u : 0x0c020500u;
    sa23 = hi ? 0x0c020700u : 0x0c020600u;
    sb23 = hi ? 0x0c020600u : 0x0c020700u;
  }
  __device__ __forceinline__ uint32_t gather(const char* lds, uint32_t x, uint32_t d) const {
    const uint32_t t0 = lds_u32(lds, __builtin_amdgcn_perm(x, ca, sa01));
    const uint32_t t1 = lds_u32(lds, __builtin_amdgcn_perm(x, cb, sb01));
    const uint32_t t2 = lds_u32(lds, __builtin_amdgcn_perm(x, ca, sa23) + 128u);
    const uint32_t t3 = lds_u32(lds, __builtin_amdgcn_perm(x, cb, sb23) + 128u);
    return xor3(xor3(t0, t1, t2), t3, d);
  }
  // the same lookups in a second table set 64 KiB further on (caz = ca + 65536, cbz = cb + 65536: byte 2 of the constant)
  __device__ __forceinline__ uint32_t gather_at(const char* lds, uint32_t caz, uint32_t cbz, uint32_t x, uint32_t d) const {
    const uint32_t t0 = lds_u32(lds, __builtin_amdgcn_perm(x, caz, sa01));
    const uint32_t t1 = lds_u32(lds, __builtin_amdgcn_perm(x, cbz, sb01));
    const uint32_t t2 = lds_u32(lds, __builtin_amdgcn_perm(x, caz, sa23) + 128u);
    const uint32_t t3 = lds_u32(lds, __builtin_amdgcn_perm(x, cbz, sb23) + 128u);
    return xor3(xor3(t0, t1, t2), t3, d);
  }
};

template <int COPIES>
__device__ __forceinline__ void absorb(const char* lds, const LaneC<COPIES>& lc, Acc4& a, const uint4 d) {
  const uint32_t n0 = lc.gather(lds, a.a0, d.x), n1 = lc.gather(lds, a.a1, d.y), n2 = lc.gather(lds, a.a2, d.z),
                 n3 = lc.gather(lds, a.a3, d.w);
  a = {n0, n1, n2, n3};
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;
__device__ __forceinline__ uint4 ld16(uintptr_t p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<gptr_u4>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <int COPIES, int ZC, int THREADS>
__device__ __forceinline__ void build_tables(char* lds, int main_op, int tid) {
  using L = Lay<COPIES, ZC>;
  for (int e = tid; e < 1024; e += THREADS) {
    const int j = e >> 8, b = e & 255;  // byte table j, entry b
    uint32_t v = 0, z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      v ^= ((b >> i) & 1) ? c_basis.op[main_op].v[8 * j + i] : 0u;
      z ^= ((b >> i) & 1) ? c_basis.op[kOpZ4].v[8 * j + i] : 0u;
    }
    const uint4 v4 = make_uint4(v, v, v, v);
    uint4* dst = reinterpret_cast<uint4*>(COPIES == 32 ? lds + (j >> 1) * 65536 + b * 256 + (j & 1) * 128
                                                       : lds + b * 256 + j * 64);
#pragma unroll
    for (int r = 0; r < COPIES / 4; r++) dst[r] = v4;
    if (ZC == 1) {
      *reinterpret_cast<uint32_t*>(lds + L::kZ4 + (j * 256 + b) * 4) = z;
    } else {
      const uint4 z4 = make_uint4(z, z, z, z);
      uint4* dz = reinterpret_cast<uint4*>(lds + L::kZ4 + b * 256 + j * 64);
#pragma unroll
      for (int r = 0; r < 4; r++) dz[r] = z4;
    }
  }
  for (int e = tid; e < kNibOps * 128; e += THREADS) {
    const int op = e >> 7, q = (e >> 4) & 7, nv = e & 15;
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) w ^= ((nv >> i) & 1) ? c_basis.op[op].v[4 * q + i] : 0u;
    *reinterpret_cast<uint32_t*>(lds + L::kNib + e * 4) = w;
  }
}

struct Z4Addr {
  uint32_t o0, o1, o2, o3;
};
__device__ __forceinline__ uint32_t z4_step(const char* lds, Z4Addr& ad, uint32_t x, uint32_t nxt) {
  const uint32_t two = 2u;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0"
      : "+v"(ad.o0) : "v"(two), "v"(x));
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1"
      : "+v"(ad.o1) : "v"(two), "v"(x));
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2"
      : "+v"(ad.o2) : "v"(two), "v"(x));
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3"
      : "+v"(ad.o3) : "v"(two), "v"(x));
  const uint32_t t0 = lds_u32(lds, ad.o0), t1 = lds_u32(lds, ad.o1 + 1024u), t2 = lds_u32(lds, ad.o2 + 2048u),
                 t3 = lds_u32(lds, ad.o3 + 3072u);
  return xor3(xor3(t0, t1, t2), t3, nxt);
}
__device__ __forceinline__ uint32_t slot_fold_chain(const char* lds, Z4Addr& ad, const Acc4& a) {
  uint32_t x = z4_step(lds, ad, a.a0, a.a1);
  x = z4_step(lds, ad, x, a.a2);
  x = z4_step(lds, ad, x, a.a3);
  return z4_step(lds, ad, x, 0u);
}

// The library's crc_flat_kernel, compute form without phases: LPW lanes per window, two rows of 64 / LPW windows per
// wavefront and pass, digests held in HOLD collectors per row until the wavefront's reads are done.
template <int LPW, int UNR, int COPIES, int THREADS, int WPE, int HOLD, bool LATE, int ZC = 1, bool PRE = false, int CUT = 0>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void flat_kernel(
    const uint8_t* data, uint32_t* sums, uint32_t bpc, uint64_t n_windows) {
  static_assert(ZC == 1 || (ZC == 16 && COPIES == 16), "16 copies of the 4-byte operator go with 16 copies of the main tables");
  using L = Lay<COPIES, ZC>;
  __shared__ __attribute__((aligned(16))) char lds[L::kBytes];
  constexpr int kWaves = THREADS / 64;
  constexpr uint32_t kStripe = 16u * LPW;
  constexpr uint32_t kGroup = UNR * kStripe;
  constexpr int kWinPerRow = 64 / LPW;
  constexpr int kWinPerWave = 2 * kWinPerRow;
  constexpr int kLevels = LPW == 4 ? 2 : LPW == 8 ? 3 : LPW == 16 ? 4 : 5;
  constexpr int kMainOp = kOpZ16 + kLevels;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  build_tables<COPIES, ZC, THREADS>(lds, kMainOp, tid);
  __syncthreads();

  const LaneC<COPIES> lc(lane);
  const int sl = lane & (LPW - 1);
  const int grp = lane / LPW;
  const uint32_t init = sl == 0 ? 0xFFFFFFFFu : 0u;
  const uintptr_t base = reinterpret_cast<uintptr_t>(data) + 16u * sl;
  const uint64_t last = n_windows - 1;
  Z4Addr za = {L::kZ4, L::kZ4, L::kZ4, L::kZ4}, zb = za;
  uint32_t keep_a[HOLD], keep_b[HOLD];
#pragma unroll
  for (int q = 0; q < HOLD; q++) keep_a[q] = keep_b[q] = 0u;
  uint32_t cur_a = 0, cur_b = 0;
  uint32_t pass = 0;
  const uint32_t pull = (uint32_t)((lane % kWinPerRow) * LPW + (LPW - 1)) * 4u;
  const uint32_t slot = (uint32_t)(lane / kWinPerRow);
  Acc4 late_a = {0, 0, 0, 0}, late_b = {0, 0, 0, 0};
  bool late = false;

  uint32_t caz = 0, cbz = 0;
  if constexpr (ZC == 16) {
    caz = lc.ca + L::kZ4;
    cbz = lc.cb + L::kZ4;
  }
  // CUT (timing only, digests wrong on purpose): 1 no unit merge, 2 no lane tree, 3 neither
  auto fold = [&](const Acc4& fa, const Acc4& fb, uint32_t& x, uint32_t& y) {
    if constexpr ((CUT & 1) != 0) {
      x = fa.a0 ^ fa.a1 ^ fa.a2 ^ fa.a3;
      y = fb.a0 ^ fb.a1 ^ fb.a2 ^ fb.a3;
    } else if constexpr (ZC == 16) {  // the Horner chain through conflict-free copies; the two rows' chains interleave
      x = lc.gather_at(lds, caz, cbz, fa.a0, fa.a1);
      y = lc.gather_at(lds, caz, cbz, fb.a0, fb.a1);
      x = lc.gather_at(lds, caz, cbz, x, fa.a2);
      y = lc.gather_at(lds, caz, cbz, y, fb.a2);
      x = lc.gather_at(lds, caz, cbz, x, fa.a3);
      y = lc.gather_at(lds, caz, cbz, y, fb.a3);
      x = lc.gather_at(lds, caz, cbz, x, 0u);
      y = lc.gather_at(lds, caz, cbz, y, 0u);
    } else {
      x = slot_fold_chain(lds, za, fa);
      y = slot_fold_chain(lds, zb, fb);
    }
#pragma unroll
    for (int i = 0; i < ((CUT & 2) ? 0 : kLevels); i++) {
      const bool taker = (sl & ((2 << i) - 1)) == (2 << i) - 1;
      const uint32_t below = lanes_up(x, i);
      const uint32_t t = nib_apply<L::kNib>(lds, kOpZ16 + i, taker ? below : y);
      x ^= t;
      y ^= lanes_up(t, i);
    }
  };
  auto collect = [&](uint32_t x, uint32_t y) {
    const uint32_t ta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pull, (int)x);
    const uint32_t tb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pull, (int)y);
    const bool mine = slot == (pass & (LPW - 1));
    cur_a = mine ? ta : cur_a;
    cur_b = mine ? tb : cur_b;
    pass++;
    if ((pass & (LPW - 1)) == 0) {
      const uint32_t full = pass / LPW - 1;
#pragma unroll
      for (int q = 0; q < HOLD; q++) {
        keep_a[q] = full == (uint32_t)q ? cur_a : keep_a[q];
        keep_b[q] = full == (uint32_t)q ? cur_b : keep_b[q];
      }
    }
  };

  if constexpr (PRE) {
    // A whole window is one load group (the host checks bpc == kGroup): the next pass's windows are requested before the
    // current pass's arithmetic, two register buffers taking turns.
    const uint64_t stride = (uint64_t)gridDim.x * kWaves * kWinPerWave;
    uint64_t w0 = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave;
    auto load = [&](uint4 (&da)[UNR], uint4 (&db)[UNR], uint64_t w) {
      const uint64_t ga = w + grp, gb = w + kWinPerRow + grp;
      const uintptr_t wa = base + (ga < last ? ga : last) * bpc, wb = base + (gb < last ? gb : last) * bpc;
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        da[k] = ld16(wa + k * kStripe);
        db[k] = ld16(wb + k * kStripe);
      }
    };
    auto work = [&](const uint4 (&da)[UNR], const uint4 (&db)[UNR]) {
      Acc4 a = {da[0].x ^ init, da[0].y, da[0].z, da[0].w};
      Acc4 b = {db[0].x ^ init, db[0].y, db[0].z, db[0].w};
#pragma unroll
      for (int k = 1; k < UNR; k++) {
        absorb<COPIES>(lds, lc, a, da[k]);
        absorb<COPIES>(lds, lc, b, db[k]);
      }
      uint32_t x, y;
      fold(a, b, x, y);
      collect(x, y);
    };
    // One buffer in use, one on its way; the copy at the end of a pass is what the next pass reads, so the request cannot
    // be moved behind the arithmetic it is meant to overlap (requests that are used in the same loop iteration were:
    // the compiler sank them to just in front of their first use).  Windows beyond the end are clamped to the last one,
    // so the request is unconditional.
    uint4 pa[UNR], pb[UNR];
    if (w0 < n_windows) load(pa, pb, w0);
    for (; w0 < n_windows; w0 += stride) {
      uint4 qa[UNR], qb[UNR];
      load(qa, qb, w0 + stride);
      work(pa, pb);
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        pa[k] = qa[k];
        pb[k] = qb[k];
      }
    }
  } else {
#ifdef INCR_ADDR  // A/B build (-DINCR_ADDR): window addresses by increment, the clamp only in a wavefront's last, partial pass.
                  // Measured: no difference at any window size (the pass's floor is not its address arithmetic).
  const uint64_t stride_w = (uint64_t)gridDim.x * kWaves * kWinPerWave;
  const uint64_t step_bytes = stride_w * bpc;
  uint64_t w0 = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave;
  uintptr_t pa = base + (w0 + grp) * bpc, pb = pa + (uint64_t)kWinPerRow * bpc;
  for (; w0 < n_windows; w0 += stride_w, pa += step_bytes, pb += step_bytes) {
    uintptr_t wa = pa, wb = pb;
    if (w0 + kWinPerWave > n_windows) {
      const uint64_t ga = w0 + grp, gb = w0 + kWinPerRow + grp;
      wa = base + (ga < last ? ga : last) * bpc;
      wb = base + (gb < last ? gb : last) * bpc;
    }
#else
  for (uint64_t w0 = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave; w0 < n_windows;
       w0 += (uint64_t)gridDim.x * kWaves * kWinPerWave) {
    const uint64_t ga = w0 + grp, gb = w0 + kWinPerRow + grp;
    const uintptr_t wa = base + (ga < last ? ga : last) * bpc, wb = base + (gb < last ? gb : last) * bpc;
#endif
    uint4 ca[UNR], cb[UNR], na[UNR], nb[UNR];
    uintptr_t ua = wa, ub = wb;
#pragma unroll
    for (int k = 0; k < UNR; k++) {
      ca[k] = ld16(ua + k * kStripe);
      cb[k] = ld16(ub + k * kStripe);
    }
    if (bpc > kGroup) {
      ua += kGroup;
      ub += kGroup;
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        na[k] = ld16(ua + k * kStripe);
        nb[k] = ld16(ub + k * kStripe);
      }
    }
    if (LATE && late) {
      uint32_t x, y;
      fold(late_a, late_b, x, y);
      collect(x, y);
      late = false;
    }
    Acc4 a = {ca[0].x ^ init, ca[0].y, ca[0].z, ca[0].w};
    Acc4 b = {cb[0].x ^ init, cb[0].y, cb[0].z, cb[0].w};
#pragma unroll
    for (int k = 1; k < UNR; k++) {
      absorb<COPIES>(lds, lc, a, ca[k]);
      absorb<COPIES>(lds, lc, b, cb[k]);
    }
    for (uint32_t done = kGroup; done < bpc; done += kGroup) {
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        ca[k] = na[k];
        cb[k] = nb[k];
      }
      if (done + kGroup < bpc) {
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
        absorb<COPIES>(lds, lc, a, ca[k]);
        absorb<COPIES>(lds, lc, b, cb[k]);
      }
    }
    if (LATE) {
      late_a = a;
      late_b = b;
      late = true;
      continue;
    }
    uint32_t x, y;
    fold(a, b, x, y);
    collect(x, y);
  }
  }
  if (LATE && late) {
    uint32_t x, y;
    fold(late_a, late_b, x, y);
    collect(x, y);
  }
  if ((pass & (LPW - 1)) != 0) {
    const uint32_t part = pass / LPW;
#pragma unroll
    for (int q = 0; q < HOLD; q++) {
      keep_a[q] = part == (uint32_t)q ? cur_a : keep_a[q];
      keep_b[q] = part == (uint32_t)q ? cur_b : keep_b[q];
    }
  }
  const uint64_t first = ((uint64_t)blockIdx.x * kWaves + wave) * kWinPerWave + (uint32_t)(lane % kWinPerRow);
  const uint64_t stride = (uint64_t)gridDim.x * kWaves * kWinPerWave;
#pragma unroll
  for (int q = 0; q < HOLD; q++) {
    const uint32_t k = (uint32_t)q * LPW + slot;
    if (k < pass) {
      const uint64_t g = first + (uint64_t)k * stride;
      if (g <= last) sums[g] = __builtin_bswap32(~keep_a[q]);
      if (g + kWinPerRow <= last) sums[g + kWinPerRow] = __builtin_bswap32(~keep_b[q]);
    }
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------

__host__ __device__ inline uint32_t mix32(uint64_t i) {  // dword i of the synthetic input
  uint64_t z = i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)(z ^ (z >> 31));
}
__global__ void fill_kernel(uint32_t* p, uint64_t n_dwords) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = mix32(i);
}
static uint32_t host_digest(uint64_t window, uint32_t bpc) {  // as stored: 4 big-endian bytes read as a dword
  uint32_t crc = 0xFFFFFFFFu;
  for (uint32_t k = 0; k < bpc / 4; k++) {
    const uint32_t d = mix32(window * (bpc / 4) + k);
    for (int b = 0; b < 4; b++) crc = crc_byte_bitwise(crc, (d >> (8 * b)) & 0xffu, kPolyCrc32);
  }
  return __builtin_bswap32(~crc);
}

struct Config {
  const char* name;
  const void* fn;
  int lpw, threads, hold, wgs_per_cu, table_bytes;
  void (*launch)(int blocks, size_t pad, hipStream_t s, const uint8_t* data, uint32_t* sums, uint32_t bpc, uint64_t n);
  uint32_t only_bpc = 0;  // a form built for one window size only
  bool timing_only = false;  // a part of the fold is left out: the digests are wrong on purpose
};

template <int LPW, int UNR, int COPIES, int THREADS, int WPE, int HOLD, bool LATE, int ZC, bool PRE = false, int CUT = 0>
static void launch_one(int blocks, size_t pad, hipStream_t s, const uint8_t* data, uint32_t* sums, uint32_t bpc, uint64_t n) {
  flat_kernel<LPW, UNR, COPIES, THREADS, WPE, HOLD, LATE, ZC, PRE, CUT><<<blocks, THREADS, pad, s>>>(data, sums, bpc, n);
}
#define CFGZ(name, LPW, UNR, COPIES, THREADS, WPE, HOLD, LATE, WGS, ZC)                                         \
  Config {                                                                                                      \
    name, (const void*)flat_kernel<LPW, UNR, COPIES, THREADS, WPE, HOLD, LATE, ZC>, LPW, THREADS, HOLD, WGS,    \
        (int)Lay<COPIES, ZC>::kBytes, launch_one<LPW, UNR, COPIES, THREADS, WPE, HOLD, LATE, ZC>                \
  }
#define CFGP(name, LPW, UNR, HOLD)                                                                              \
  Config {                                                                                                      \
    name, (const void*)flat_kernel<LPW, UNR, 16, 1024, 4, HOLD, false, 16, true>, LPW, 1024, HOLD, 1,           \
        (int)Lay<16, 16>::kBytes, launch_one<LPW, UNR, 16, 1024, 4, HOLD, false, 16, true>, 16u * LPW * UNR       \
  }
#define CFGC(name, LPW, UNR, LATE, CUT)                                                                         \
  Config {                                                                                                      \
    name, (const void*)flat_kernel<LPW, UNR, 16, 1024, 4, 8, LATE, 16, false, CUT>, LPW, 1024, 8, 1,            \
        (int)Lay<16, 16>::kBytes, launch_one<LPW, UNR, 16, 1024, 4, 8, LATE, 16, false, CUT>, 0, true            \
  }
#define CFG(name, LPW, UNR, COPIES, THREADS, WPE, HOLD, LATE, WGS) CFGZ(name, LPW, UNR, COPIES, THREADS, WPE, HOLD, LATE, WGS, 1)

template <int LPW, int UNR>
static std::vector<Config> configs_for() {
  return {
      CFG("32 copies, 1024 threads, 1 workgroup/CU (the library before this experiment)", LPW, UNR, 32, 1024, 4, 8, false, 1),
      CFG("32 copies, 1024 threads, 1 workgroup/CU, fold one pass late", LPW, UNR, 32, 1024, 4, 8, true, 1),
      CFG("16 copies, 1024 threads, 1 workgroup/CU (LDS padded)", LPW, UNR, 16, 1024, 4, 8, false, 1),
      CFG("16 copies, 1024 threads, 1 workgroup/CU (LDS padded), late fold", LPW, UNR, 16, 1024, 4, 8, true, 1),
      CFGZ("16 copies + 16 of the unit merge's tables (135 KiB), 1/CU (the library since)", LPW, UNR, 16, 1024, 4, 8, false, 1, 16),
      CFGZ("16 copies + 16 of the unit merge's tables (135 KiB), 1/CU, late fold", LPW, UNR, 16, 1024, 4, 8, true, 1, 16),
      CFGZ("16 copies + 16 of the unit merge's tables, 1/CU, 16 collectors per row", LPW, UNR, 16, 1024, 4, 16, false, 1, 16),
      CFGP("16 + 16 copies, 1/CU, the next pass's windows requested a pass ahead (whole window = one load group)", LPW, UNR, 8),
      CFGC("  timing only: the library's form without the unit merge", LPW, UNR, false, 1),
      CFGC("  timing only: without the lane tree", LPW, UNR, false, 2),
      CFGC("  timing only: without either (the loop, the loads, the stripe lookups, the collectors)", LPW, UNR, false, 3),
      CFGC("  timing only: late fold, without either", LPW, UNR, true, 3),
      CFG("16 copies,  512 threads, 2 workgroups/CU (4 wavefronts/SIMD)", LPW, UNR, 16, 512, 4, 8, false, 2),
      CFG("16 copies,  512 threads, 2 workgroups/CU, late fold", LPW, UNR, 16, 512, 4, 8, true, 2),
      CFG("16 copies,  768 threads, 2 workgroups/CU (6 wavefronts/SIMD)", LPW, UNR, 16, 768, 6, 8, false, 2),
      CFG("16 copies,  768 threads, 2 workgroups/CU, late fold", LPW, UNR, 16, 768, 6, 8, true, 2),
      CFG("16 copies,  768 threads, 2 workgroups/CU, 4 collectors", LPW, UNR, 16, 768, 6, 4, false, 2),
      CFG("16 copies, 1024 threads, 2 workgroups/CU (8 wavefronts/SIMD), 4 collectors", LPW, UNR, 16, 1024, 8, 4, false, 2),
  };
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <bytes per window: 128, 256, 512, 1024, 2048, 4096 or 8192> [GiB per step]\n", argv[0]);
    return 1;
  }
  const uint32_t bpc = (uint32_t)atoi(argv[1]);
  const uint64_t step_bytes = (uint64_t)(argc > 2 ? atof(argv[2]) : 8.0) * (1ull << 30);
  std::vector<Config> cfgs;
  switch (bpc) {  // lanes per window and stripes per load group as the library chooses them
    case 128: cfgs = configs_for<4, 2>(); break;
    case 256: cfgs = configs_for<8, 2>(); break;
    case 512: cfgs = configs_for<8, 1>(); break;
    case 1024: cfgs = configs_for<8, 1>(); break;
    case 2048: cfgs = configs_for<8, 1>(); break;
    case 4096: cfgs = configs_for<16, 1>(); break;
    case 8192: cfgs = configs_for<16, 2>(); break;
    default: fprintf(stderr, "unsupported window size %u\n", bpc); return 1;
  }
  if (getenv("SHAPE")) {  // another (lanes per window, stripes per load group) than the library's, e.g. SHAPE=4x4
    const char* sh = getenv("SHAPE");
    if (!strcmp(sh, "4x4")) cfgs = configs_for<4, 4>();
    else if (!strcmp(sh, "4x2")) cfgs = configs_for<4, 2>();
    else if (!strcmp(sh, "4x1")) cfgs = configs_for<4, 1>();
    else if (!strcmp(sh, "8x1")) cfgs = configs_for<8, 1>();
    else if (!strcmp(sh, "8x2")) cfgs = configs_for<8, 2>();
    else if (!strcmp(sh, "16x1")) cfgs = configs_for<16, 1>();
    else if (!strcmp(sh, "16x2")) cfgs = configs_for<16, 2>();
    else if (!strcmp(sh, "32x1")) cfgs = configs_for<32, 1>();
    else { fprintf(stderr, "unknown shape %s\n", sh); return 1; }
    if (bpc % (16u * (sh[0] == '1' ? 16 : sh[0] == '3' ? 32 : sh[0] - '0') * (sh[strlen(sh) - 1] - '0')) != 0) { fprintf(stderr, "shape does not divide the window\n"); return 1; }
    printf("shape %s\n", sh);
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t lds_cu = 160 * 1024;
  const uint64_t n_all = step_bytes / bpc;
  uint8_t* data;
  uint32_t* sums;
  CK(hipMalloc(&data, step_bytes));
  CK(hipMalloc(&sums, n_all * 4));
  fill_kernel<<<cus * 8, 256>>>(reinterpret_cast<uint32_t*>(data), step_bytes / 4);
  CK(hipDeviceSynchronize());
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%u-byte windows, %.0f GiB a step, %d CUs\n", bpc, step_bytes / 1073741824.0, cus);

  if (getenv("REVERSE")) std::reverse(cfgs.begin(), cfgs.end());  // the same table from the other end: is the order part of the result?
  // Whatever runs in a process's first 100-200 ms of GPU work is measured 3-5 % low (the same form reads 6.55 TB/s as the
  // first of the table and 6.82 as the last), so the first form runs twice: 200 untimed steps, result not printed, first.
  cfgs.insert(cfgs.begin(), cfgs.front());
  bool discard = true;
  for (const Config& c : cfgs) {
    if (c.only_bpc && c.only_bpc != bpc) continue;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, c.fn));
    // dynamic LDS on top of the tables, so that exactly wgs_per_cu workgroups fit a CU
    size_t pad = 0;
    if ((size_t)c.table_bytes * (c.wgs_per_cu + 1) <= lds_cu) pad = lds_cu / c.wgs_per_cu - c.table_bytes - 1024;
    CK(hipFuncSetAttribute(c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    int resident = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, c.fn, c.threads, pad));
    const int blocks = cus * c.wgs_per_cu;
    // windows one launch takes: every wavefront fills its collectors exactly
    const uint64_t cap = (uint64_t)blocks * (c.threads / 64) * (2 * 64 / c.lpw) * c.hold * c.lpw;
    const uint64_t launches = (n_all + cap - 1) / cap;
    const uint64_t per_pass = (uint64_t)blocks * (c.threads / 64) * (2 * 64 / c.lpw);
    uint64_t share = (n_all + launches - 1) / launches;
    share = (share + per_pass - 1) / per_pass * per_pass;
    auto step = [&]() {
      for (uint64_t off = 0; off < n_all; off += share) {
        const uint64_t n = n_all - off < share ? n_all - off : share;
        c.launch(blocks, pad, s, data + off * bpc, sums + off, bpc, n);
      }
    };
    CK(hipMemsetAsync(sums, 0, n_all * 4, s));
    step();
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    // sampled digests against the host
    std::vector<uint32_t> got(n_all);
    CK(hipMemcpy(got.data(), sums, n_all * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0, checked = 0;
    for (uint64_t k = 0; k < 4099; k++) {
      uint64_t w = k == 0 ? 0 : k == 1 ? n_all - 1 : k == 2 ? share - 1 : k == 3 ? share : mix32(k);  // ends of launches, then anywhere
      w %= n_all;
      checked++;
      if (!c.timing_only && got[w] != host_digest(w, bpc)) {
        if (bad++ < 3) fprintf(stderr, "  window %llu: %08x, host %08x\n", (unsigned long long)w, got[w], host_digest(w, bpc));
      }
    }
    for (int i = 0; i < (discard ? 200 : 10); i++) step();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; i++) step();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double tbs = (double)step_bytes * 20 / (ms * 1e-3) / 1e12;
    if (discard) {
      discard = false;
      continue;
    }
    printf("%-82s %3d VGPRs, %4d B spilled, %d resident/CU, %llu launches a step: %5.2f TB/s  digests %s (%llu sampled)\n",
           c.name, fa.numRegs, (int)fa.localSizeBytes, resident, (unsigned long long)launches, tbs,
           c.timing_only ? "not checked" : bad ? "WRONG" : "ok", (unsigned long long)checked);
    fflush(stdout);
    if (bad) return 3;
  }
  return 0;
}
