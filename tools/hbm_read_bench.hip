// Measures what a pure streaming read reaches on this GPU, in the access patterns the CRC kernel could use.
// Build: make -C tools   (hipcc --offload-arch=gfx950 -O3 -o tools/hbm_read_bench tools/hbm_read_bench.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// grid-stride, every lane 16 B per iteration, UNROLL loads in flight
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_gridstride(const u32x4* p, uint64_t n16, uint32_t* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) v[k] = NT ? __builtin_nontemporal_load((gptr)(p + i + k * stride)) : *(gptr)(p + i + k * stride);
#pragma unroll
    for (int k = 0; k < UNROLL; k++) acc ^= v[k];
  }
  for (; i < n16; i += stride) acc ^= *(gptr)(p + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// the CRC kernel's pattern: 1024-thread blocks, one per CU, wave w streams its own windows of `win` bytes, 1 KiB rows
template <int UNROLL, bool NT>
__global__ __launch_bounds__(1024) void read_wave_windows(const uint8_t* p, uint64_t nwin, uint64_t win, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t g = (uint64_t)blockIdx.x * 16 + wave; g < nwin; g += (uint64_t)gridDim.x * 16) {
    const uint8_t* q = p + g * win + 16 * lane;
    for (uint64_t o = 0; o < win; o += 1024 * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; k++) v[k] = NT ? __builtin_nontemporal_load((gptr)(q + o + 1024 * k)) : *(gptr)(q + o + 1024 * k);
#pragma unroll
      for (int k = 0; k < UNROLL; k++) acc ^= v[k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// wave-windows with the cache policy spelled in the instruction (POLICY: 0 plain, 1 nt, 2 sc1 nt, 3 sc0 sc1 nt, 4 sc1)
// and an optional staggered start: wave g begins its window at offset (g * 8 KiB) mod win and wraps, so that the
// 4096 wavefronts do not all sit at the same offset of their windows.
template <int POLICY>
__device__ __forceinline__ u32x4 ld_policy(const uint8_t* a) {
  u32x4 v;
  if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(a) : "memory");
  if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(a) : "memory");
  if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(a) : "memory");
  if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(a) : "memory");
  if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
template <int UNROLL, int POLICY, bool STAGGER>
__global__ __launch_bounds__(1024) void read_wave_windows_policy(const uint8_t* p, uint64_t nwin, uint64_t win, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t g = (uint64_t)blockIdx.x * 16 + wave; g < nwin; g += (uint64_t)gridDim.x * 16) {
    const uint8_t* q = p + g * win + 16 * lane;
    const uint64_t start = STAGGER ? (g * 8192) & (win - 1) : 0;
    for (uint64_t o = 0; o < win; o += 1024 * UNROLL) {
      const uint64_t oo = (o + start) & (win - 1);
      u32x4 v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; k++) v[k] = ld_policy<POLICY>(q + oo + 1024 * k);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < UNROLL; k++) acc ^= v[k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// exactly the CRC kernel's pattern: G wavefronts share a window, stripe = 2*G KiB, each wavefront reads two 1 KiB
// rows of every stripe, one stripe prefetched ahead (non-temporal loads)
// XCDMAP: workgroups go to the chip's eight XCDs round robin (workgroup b runs on XCD b % 8).  0: workgroup b takes the b-th
// share of every pass over the windows, so the eight XCDs' shares interleave; 1: XCD x takes the x-th eighth of every
// pass (its workgroups' shares lie side by side); 2: XCD x takes the x-th eighth of the whole batch.
template <int G, int XCDMAP = 0>
__global__ __launch_bounds__(1024) void read_crc_pattern(const uint8_t* p, uint64_t nwin, uint64_t win, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = wave / G, wsub = wave % G;
  constexpr uint64_t kStripe = 2048ull * G;
  u32x4 acc = {0, 0, 0, 0};
  const uint32_t per_xcd = gridDim.x / 8;
  const uint32_t blk = XCDMAP ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
  if (XCDMAP == 2) {
    const uint64_t share = nwin / gridDim.x;  // windows of one workgroup, contiguous
    for (uint64_t k = sub; k < share; k += 16 / G) {
      const uint8_t* q = p + ((uint64_t)blk * share + k) * win + 2048 * wsub + 16 * lane;
      u32x4 a = __builtin_nontemporal_load((gptr)q), b = __builtin_nontemporal_load((gptr)(q + 1024));
      for (uint64_t o = kStripe; o < win; o += kStripe) {
        const u32x4 na = __builtin_nontemporal_load((gptr)(q + o)), nb = __builtin_nontemporal_load((gptr)(q + o + 1024));
        acc ^= a ^ b;
        a = na;
        b = nb;
      }
      acc ^= a ^ b;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
    return;
  }
  for (uint64_t g = (uint64_t)blk * (16 / G) + sub; g < nwin; g += (uint64_t)gridDim.x * (16 / G)) {
    const uint8_t* q = p + g * win + 2048 * wsub + 16 * lane;
    u32x4 a = __builtin_nontemporal_load((gptr)q), b = __builtin_nontemporal_load((gptr)(q + 1024));
    for (uint64_t o = kStripe; o < win; o += kStripe) {
      const u32x4 na = __builtin_nontemporal_load((gptr)(q + o)), nb = __builtin_nontemporal_load((gptr)(q + o + 1024));
      acc ^= a ^ b;
      a = na;
      b = nb;
    }
    acc ^= a ^ b;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// the small-window CRC kernels' pattern: LPW lanes share a window (a load instruction gathers 64/LPW pieces of
// 16*LPW bytes, one per window, `win` bytes apart), two rows of windows per wavefront, one stripe group ahead;
// with LDS != 0 the workgroup also holds 154 KiB of LDS (more than half a CU, as the CRC kernels do) (one workgroup per CU)
// SPREAD (odd): logical window g is physical window (g * SPREAD) mod nwin, nwin a power of two -- the windows one
// wavefront reads together are then SPREAD windows apart instead of adjacent
// PHASE: every wavefront starts its windows at an offset of its own and wraps round (1: by wavefront number, 2: by
// wavefront and workgroup number), so that the wavefronts of the chip do not all stand at the same offset inside
// their windows at the same time
template <int LPW, int LDS, int SPREAD = 1, int PHASE = 0>
__global__ __launch_bounds__(1024) void read_flat_pattern(const uint8_t* p, uint64_t nwin, uint32_t win, uint32_t* out) {
  __shared__ char lds[LDS ? 157696 : 16];
  if (LDS && threadIdx.x == 0) lds[win & 15] = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kPerRow = 64 / LPW;
  constexpr uint32_t kStripe = 16u * LPW;
  const int sl = lane & (LPW - 1), grp = lane / LPW;
  const uint32_t nst = win / kStripe;
  const uint32_t slot = PHASE == 2 ? (uint32_t)(wave * 16 + (blockIdx.x & 15)) : (uint32_t)wave * 16;
  uint32_t ph = PHASE ? (uint32_t)(((uint64_t)slot * nst) >> 8) * kStripe : 0;
  if (PHASE >= 4) ph = (uint32_t)(((uint64_t)(wave % PHASE) * nst) / PHASE) * kStripe;  // PHASE uniform phases
  if (PHASE == 3) {  // the tail behind the phase is a power of two: win, win/2, win/4, ... (at least one stripe)
    uint32_t tail = win >> (wave & 7);
    if (tail < kStripe) tail = kStripe;
    ph = win - tail;
  }
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t w0 = ((uint64_t)blockIdx.x * 16 + wave) * 2 * kPerRow; w0 < nwin; w0 += (uint64_t)gridDim.x * 16 * 2 * kPerRow) {
    const uint64_t ga = w0 + grp, gb = w0 + kPerRow + grp;
    const uint8_t* qa = p + (((ga < nwin ? ga : nwin - 1) * SPREAD) & (nwin - 1)) * win + 16 * sl;
    const uint8_t* qb = p + (((gb < nwin ? gb : nwin - 1) * SPREAD) & (nwin - 1)) * win + 16 * sl;
    u32x4 a = __builtin_nontemporal_load((gptr)(qa + ph)), b = __builtin_nontemporal_load((gptr)(qb + ph));
    uint32_t o = ph;
    for (uint32_t k = 1; k < nst; k++) {
      o += kStripe;
      if (o >= win) o -= win;
      const u32x4 na = __builtin_nontemporal_load((gptr)(qa + o)), nb = __builtin_nontemporal_load((gptr)(qb + o));
      acc ^= a ^ b;
      a = na;
      b = nb;
    }
    acc ^= a ^ b;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = lds[0];
}

// The flat pattern with a rolling queue of D stripe groups per wavefront (requests cross from one window group into the
// next, explicit loads and vmcnt waits) and, optionally, the CRC kernels' kind of work between the requests:
// WORK & 1: sixteen conflict-free LDS lookups per 16-byte unit (lane l reads bank l % 32 of a random row),
// WORK & 2: twenty-four dependent vector instructions per unit.  Shows what the memory system delivers to this request
// stream when the CUs are idle, busy in the LDS, busy in the vector ALUs, or both.
#ifndef ROLLING_LOAD_POLICY
#define ROLLING_LOAD_POLICY "nt"
#endif
__device__ __forceinline__ void ld_async(u32x4& v, const uint8_t* a) {
  asm volatile("global_load_dwordx4 %0, %1, off " ROLLING_LOAD_POLICY : "=v"(v) : "v"(a) : "memory");
}
template <int K>
__device__ __forceinline__ void wait_vm(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(K) : "memory");
}
template <int WORK>
__device__ __forceinline__ void unit_work(const char* lds, uint32_t lane4, u32x4& acc, const u32x4 d) {
  if (WORK & 1) {
    uint32_t r[4] = {acc.x, acc.y, acc.z, acc.w};
    const uint32_t dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint32_t t = dd[q];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t row = (r[q] >> (8 * k)) & 0xffu;
        t ^= *reinterpret_cast<const uint32_t*>(lds + row * 128u + lane4 + k * 32768u);
      }
      r[q] = t;
    }
    acc.x = r[0]; acc.y = r[1]; acc.z = r[2]; acc.w = r[3];
  }
  if (WORK & 2) {
    uint32_t r[4] = {acc.x, acc.y, acc.z, acc.w};
    const uint32_t dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int k = 0; k < 3; k++) r[q] = __builtin_amdgcn_perm(r[q], dd[q], 0x05020704u + k) ^ (r[q] >> (k + 1));
    }
    acc.x = r[0]; acc.y = r[1]; acc.z = r[2]; acc.w = r[3];
  }
  if (WORK == 0) acc ^= d;
}
template <int LPW, int NST, int D, int WORK>
__global__ __launch_bounds__(1024) void read_flat_rolling(const uint8_t* p, uint64_t nwin, uint32_t* out, uint32_t* digests) {
  __shared__ char lds[157696];
  for (int i = threadIdx.x; i < 131072 / 4; i += 1024) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int kPerRow = 64 / LPW;
  constexpr uint32_t kStripe = 16u * LPW, kWin = kStripe * NST;
  const int sl = lane & (LPW - 1), grp = lane / LPW;
  const uint32_t lane4 = (lane & 31) * 4u;
  const uint64_t step = (uint64_t)gridDim.x * 16 * 2 * kPerRow;
  uint64_t w0 = ((uint64_t)blockIdx.x * 16 + wave) * 2 * kPerRow;
  auto row = [&](uint64_t w, int r) { uint64_t g = w + r * kPerRow + grp; if (g >= nwin) g = nwin - 1; return p + g * kWin + 16 * sl; };
  const uint8_t *wa = row(w0, 0), *wb = row(w0, 1);
  u32x4 qa[NST], qb[NST];
#pragma unroll
  for (int k = 0; k < D; k++) { ld_async(qa[k], wa + k * kStripe); ld_async(qb[k], wb + k * kStripe); }
  u32x4 acc = {0, 0, 0, 0}, acc2 = {1, 2, 3, 4};
  for (; w0 < nwin; w0 += step) {
    const uint8_t *na = row(w0 + step, 0), *nb = row(w0 + step, 1);  // past the end: the last window again
#pragma unroll
    for (int s = 0; s < NST; s++) {
      const int t = s + D;
      if (D < NST) {
        if (t < NST) { ld_async(qa[t], wa + t * kStripe); ld_async(qb[t], wb + t * kStripe); }
        else { ld_async(qa[t - NST], na + (t - NST) * kStripe); ld_async(qb[t - NST], nb + (t - NST) * kStripe); }
        wait_vm<2 * D>(qa[s], qb[s]);
      } else {
        wait_vm<2 * (NST - 1)>(qa[s], qb[s]);
      }
      unit_work<WORK>(lds, lane4, acc, qa[s]);
      unit_work<WORK>(lds, lane4, acc2, qb[s]);
      if (D == NST) { ld_async(qa[s], na + s * kStripe); ld_async(qb[s], nb + s * kStripe); }
    }
    if (WORK & 8) {  // a per-window fold's shape: seven LDS round trips in a row, each waiting for the one before
#pragma unroll
      for (int k = 0; k < 7; k++) {
        uint32_t t0 = 0, t1 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          t0 ^= *reinterpret_cast<const uint32_t*>(lds + ((acc.x >> (8 * q)) & 0xffu) * 128u + lane4 + q * 32768u);
          t1 ^= *reinterpret_cast<const uint32_t*>(lds + ((acc2.x >> (8 * q)) & 0xffu) * 128u + lane4 + q * 32768u);
        }
        acc.x = t0 ^ acc.y;
        acc2.x = t1 ^ acc2.y;
      }
    }
    if ((WORK & 4) && sl == LPW - 1) {  // the digests of the window group: two 32-byte pieces per wavefront
      if (WORK & 16) {  // into a region small enough to stay in the L2: the store instructions without the DRAM writes
        digests[(w0 + grp) & 0xffff] = acc.x;
        digests[(w0 + kPerRow + grp) & 0xffff] = acc2.x;
      } else if (WORK & 32) {  // non-temporal stores
        __builtin_nontemporal_store(acc.x, digests + w0 + grp);
        __builtin_nontemporal_store(acc2.x, digests + w0 + kPerRow + grp);
      } else if (WORK & 128) {  // write-through stores (system scope)
        asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(digests + w0 + grp), "v"(acc.x) : "memory");
        asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(digests + w0 + kPerRow + grp), "v"(acc2.x) : "memory");
      } else if (WORK & 256) {  // write-through, streaming
        asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" : : "v"(digests + w0 + grp), "v"(acc.x) : "memory");
        asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" : : "v"(digests + w0 + kPerRow + grp), "v"(acc2.x) : "memory");
      } else if (WORK & 1024) {  // the same volume, but every wavefront writes one contiguous stream of its own (32 KiB
                                 // per wavefront, 512 KiB per workgroup: one page per CU as far as address translation goes)
        const uint64_t it = w0 / step;
        uint32_t* mine = digests + ((uint64_t)blockIdx.x * 16 + wave) * (nwin / ((uint64_t)gridDim.x * 16)) + it * 2 * kPerRow;
        mine[grp] = acc.x;
        mine[kPerRow + grp] = acc2.x;
      } else if (WORK & 512) {  // device scope
        asm volatile("global_store_dword %0, %1, off sc1" : : "v"(digests + w0 + grp), "v"(acc.x) : "memory");
        asm volatile("global_store_dword %0, %1, off sc1" : : "v"(digests + w0 + kPerRow + grp), "v"(acc2.x) : "memory");
      } else if (WORK & 64) {  // every fourth window group, four times the bytes: fewer store instructions, whole lines
        if (((w0 / step) & 3) == 3) {
          const u32x4 va = {acc.x, acc.y, acc.z, acc.w}, vb = {acc2.x, acc2.y, acc2.z, acc2.w};
          *reinterpret_cast<u32x4*>(digests + ((w0 / step) >> 2) * 4 * 16 * ((uint64_t)gridDim.x * 16) + ((uint64_t)blockIdx.x * 16 + wave) * 64 + grp * 4) = va;
          *reinterpret_cast<u32x4*>(digests + ((w0 / step) >> 2) * 4 * 16 * ((uint64_t)gridDim.x * 16) + ((uint64_t)blockIdx.x * 16 + wave) * 64 + 32 + grp * 4) = vb;
        }
      } else {
        digests[w0 + grp] = acc.x;
        digests[w0 + kPerRow + grp] = acc2.x;
      }
    }
    wa = na; wb = nb;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  acc ^= acc2;
  if (WORK & 2048) {  // every digest of the wavefront written when its reads are done (as if they had been kept in registers)
    const uint64_t first = ((uint64_t)blockIdx.x * 16 + wave) * 2 * kPerRow;
    for (uint64_t j = lane; first + (j / 16) * step < nwin; j += 64) digests[first + (j / 16) * step + (j % 16)] = acc.x + (uint32_t)j;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = lds[5];
}

// The rolling pattern (one stripe group in flight, LDS lookups) with the digests of KB consecutive window groups kept in
// registers and stored together: every workgroup does the same work from the same start, so without any synchronisation
// the workgroups' store bursts fall close together in time.  SC1: write-through stores (the burst reaches DRAM when it is
// issued, not when the L2 happens to evict the lines).
template <int KB, bool SC1>
__global__ __launch_bounds__(1024) void read_flat_store_bursts(const uint8_t* p, uint64_t nwin, uint32_t* out, uint32_t* digests) {
  __shared__ char lds[157696];
  for (int i = threadIdx.x; i < 131072 / 4; i += 1024) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  constexpr int LPW = 8, NST = 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int kPerRow = 64 / LPW;
  constexpr uint32_t kStripe = 16u * LPW, kWin = kStripe * NST;
  const int sl = lane & (LPW - 1), grp = lane / LPW;
  const uint32_t lane4 = (lane & 31) * 4u;
  const uint64_t step = (uint64_t)gridDim.x * 16 * 2 * kPerRow;
  uint64_t w0 = ((uint64_t)blockIdx.x * 16 + wave) * 2 * kPerRow;
  auto row = [&](uint64_t w, int r) { uint64_t g = w + r * kPerRow + grp; if (g >= nwin) g = nwin - 1; return p + g * kWin + 16 * sl; };
  const uint8_t *wa = row(w0, 0), *wb = row(w0, 1);
  u32x4 qa[NST], qb[NST];
  ld_async(qa[0], wa); ld_async(qb[0], wb);
  u32x4 acc = {0, 0, 0, 0}, acc2 = {1, 2, 3, 4};
  while (w0 < nwin) {   // nwin is a multiple of KB * step in this tool's runs
    uint32_t da[KB], db[KB];
    const uint64_t first = w0;
#pragma unroll
    for (int k = 0; k < KB; k++, w0 += step) {
      const uint8_t *na = row(w0 + step, 0), *nb = row(w0 + step, 1);
#pragma unroll
      for (int s = 0; s < NST; s++) {
        const int t = s + 1;
        if (t < NST) { ld_async(qa[t], wa + t * kStripe); ld_async(qb[t], wb + t * kStripe); }
        else { ld_async(qa[0], na); ld_async(qb[0], nb); }
        wait_vm<2>(qa[s], qb[s]);
        unit_work<1>(lds, lane4, acc, qa[s]);
        unit_work<1>(lds, lane4, acc2, qb[s]);
      }
      da[k] = acc.x; db[k] = acc2.x;
      wa = na; wb = nb;
    }
    if (sl == LPW - 1) {
#pragma unroll
      for (int k = 0; k < KB; k++) {
        uint32_t* d0 = digests + first + k * step + grp;
        if (SC1) {
          asm volatile("global_store_dword %0, %1, off sc1" : : "v"(d0), "v"(da[k]) : "memory");
          asm volatile("global_store_dword %0, %1, off sc1" : : "v"(d0 + kPerRow), "v"(db[k]) : "memory");
        } else {
          d0[0] = da[k];
          d0[kPerRow] = db[k];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  acc ^= acc2;
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = lds[5];
}

// the same with ROWS 1 KiB rows per wavefront and stripe, and PF stripes ahead
template <int G, int ROWS, int PF>
__global__ __launch_bounds__(1024) void read_crc_rows(const uint8_t* p, uint64_t nwin, uint64_t win, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = wave / G, wsub = wave % G;
  constexpr uint64_t kStripe = 1024ull * ROWS * G;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t g = (uint64_t)blockIdx.x * (16 / G) + sub; g < nwin; g += (uint64_t)gridDim.x * (16 / G)) {
    const uint8_t* q = p + g * win + 1024 * ROWS * wsub + 16 * lane;
    u32x4 buf[PF][ROWS];
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
      for (int r = 0; r < ROWS; r++) buf[k][r] = __builtin_nontemporal_load((gptr)(q + k * kStripe + 1024 * r));
    for (uint64_t o = PF * kStripe; o < win; o += PF * kStripe) {
#pragma unroll
      for (int k = 0; k < PF; k++)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          acc ^= buf[k][r];
          buf[k][r] = __builtin_nontemporal_load((gptr)(q + o + k * kStripe + 1024 * r));
        }
    }
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
      for (int r = 0; r < ROWS; r++) acc ^= buf[k][r];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// 1024-thread blocks, whole block streams one window in 16 KiB stripes
template <int UNROLL, bool NT>
__global__ __launch_bounds__(1024) void read_block_windows(const uint8_t* p, uint64_t nwin, uint64_t win, uint32_t* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t g = blockIdx.x; g < nwin; g += gridDim.x) {
    const uint8_t* q = p + g * win + 16 * threadIdx.x;
    for (uint64_t o = 0; o < win; o += 16384 * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; k++) v[k] = NT ? __builtin_nontemporal_load((gptr)(q + o + 16384 * k)) : *(gptr)(q + o + 16384 * k);
#pragma unroll
      for (int k = 0; k < UNROLL; k++) acc ^= v[k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

static int g_warmups = 3, g_reps = 10;  // launches in front of the timed ones, timed launches
template <class F>
static void timeit(const char* name, uint64_t bytes, F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < g_warmups; i++) f();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  const int reps = g_reps;
  for (int i = 0; i < reps; i++) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
    sum += ms;
  }
  printf("%-44s avg %8.1f GB/s   best %8.1f GB/s\n", name, bytes / (sum / reps) / 1e6, bytes / best / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 8ull) << 30;
  uint8_t* d; uint32_t* out;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(d, 0x5a, bytes));
  // random-ish content so that nothing can be compressed or deduplicated on the way
  std::vector<uint32_t> h(1 << 22);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
  for (uint64_t o = 0; o < bytes; o += h.size() * 4) CK(hipMemcpy(d + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const uint64_t n16 = bytes / 16, win = 1 << 20, nwin = bytes / win;
  const u32x4* p = (const u32x4*)d;
  printf("buffer %llu GiB\n", (unsigned long long)(bytes >> 30));
  if (argc > 2 && !strcmp(argv[2], "crc")) {   // only the 1 MiB-window CRC kernel's own pattern, for bench.py's ceiling line
    g_warmups = 10;                              // as bench.py times the kernel: twenty launches after ten
    g_reps = 20;
    timeit("crc pattern G=1 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<1><<<256, 1024>>>(d, nwin, win, out); });
    timeit("crc pattern G=2 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<2><<<256, 1024>>>(d, nwin, win, out); });
    timeit("crc pattern G=4 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<4><<<256, 1024>>>(d, nwin, win, out); });
    timeit("crc pattern G=8 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<8><<<256, 1024>>>(d, nwin, win, out); });
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "xcd")) {   // does it matter which XCD reads which part of the batch?
    for (int rep = 0; rep < 2; rep++) {
      timeit("crc pattern G=8, workgroup b takes share b of every pass", bytes, [&] { read_crc_pattern<8, 0><<<256, 1024>>>(d, nwin, win, out); });
      timeit("crc pattern G=8, XCD x takes eighth x of every pass", bytes, [&] { read_crc_pattern<8, 1><<<256, 1024>>>(d, nwin, win, out); });
      timeit("crc pattern G=8, XCD x takes eighth x of the batch", bytes, [&] { read_crc_pattern<8, 2><<<256, 1024>>>(d, nwin, win, out); });
      timeit("crc pattern G=2, workgroup b takes share b of every pass", bytes, [&] { read_crc_pattern<2, 0><<<256, 1024>>>(d, nwin, win, out); });
      timeit("crc pattern G=2, XCD x takes eighth x of every pass", bytes, [&] { read_crc_pattern<2, 1><<<256, 1024>>>(d, nwin, win, out); });
      timeit("crc pattern G=2, XCD x takes eighth x of the batch", bytes, [&] { read_crc_pattern<2, 2><<<256, 1024>>>(d, nwin, win, out); });
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "rolling")) {   // request depth x kind of work between the requests, 512-byte windows, 8 lanes
    const uint64_t nw = bytes / 512;
    uint32_t* dig;
    CK(hipMalloc(&dig, nw * 4));
#define ROLL(D, W, label) timeit("512 B windows, 8 lanes, " #D " stripe groups in flight, " label, bytes, [&] { read_flat_rolling<8, 4, D, W><<<256, 1024>>>(d, nw, out, dig); });
    ROLL(1, 0, "no work") ROLL(2, 0, "no work") ROLL(4, 0, "no work")
    ROLL(1, 1, "LDS lookups") ROLL(2, 1, "LDS lookups") ROLL(4, 1, "LDS lookups")
    ROLL(1, 2, "vector ALU") ROLL(2, 2, "vector ALU") ROLL(4, 2, "vector ALU")
    ROLL(1, 3, "LDS lookups + vector ALU") ROLL(2, 3, "LDS lookups + vector ALU") ROLL(4, 3, "LDS lookups + vector ALU")
    ROLL(1, 5, "LDS lookups + digest stores") ROLL(2, 5, "LDS lookups + digest stores") ROLL(4, 5, "LDS lookups + digest stores")
    ROLL(1, 21, "LDS lookups + digest stores that stay in the L2") ROLL(4, 21, "LDS lookups + digest stores that stay in the L2")
    ROLL(1, 37, "LDS lookups + non-temporal digest stores") ROLL(4, 37, "LDS lookups + non-temporal digest stores")
    ROLL(1, 133, "LDS lookups + digest stores sc0 sc1") ROLL(1, 261, "LDS lookups + digest stores sc0 sc1 nt") ROLL(1, 517, "LDS lookups + digest stores sc1")
    ROLL(1, 1029, "LDS lookups + digest stores, one contiguous stream per wavefront") ROLL(4, 1029, "LDS lookups + digest stores, one contiguous stream per wavefront")
    timeit("512 B windows, 8 lanes, LDS lookups + digest stores in bursts of 8 groups", bytes, [&] { read_flat_store_bursts<8, false><<<256, 1024>>>(d, nw, out, dig); });
    timeit("512 B windows, 8 lanes, LDS lookups + digest stores in bursts of 8 groups, sc1", bytes, [&] { read_flat_store_bursts<8, true><<<256, 1024>>>(d, nw, out, dig); });
    timeit("512 B windows, 8 lanes, LDS lookups + digest stores in bursts of 16 groups", bytes, [&] { read_flat_store_bursts<16, false><<<256, 1024>>>(d, nw, out, dig); });
    timeit("512 B windows, 8 lanes, LDS lookups + digest stores in bursts of 16 groups, sc1", bytes, [&] { read_flat_store_bursts<16, true><<<256, 1024>>>(d, nw, out, dig); });
    timeit("512 B windows, 8 lanes, LDS lookups + digest stores in bursts of 32 groups", bytes, [&] { read_flat_store_bursts<32, false><<<256, 1024>>>(d, nw, out, dig); });
    timeit("512 B windows, 8 lanes, LDS lookups + digest stores in bursts of 32 groups, sc1", bytes, [&] { read_flat_store_bursts<32, true><<<256, 1024>>>(d, nw, out, dig); });
    for (int parts : {1, 2, 4, 8, 16}) {
      char name[128];
      snprintf(name, sizeof name, "512 B windows, 8 lanes, LDS lookups + all digests stored at the end, %d launches", parts);
      timeit(name, bytes, [&] { for (int q = 0; q < parts; q++) read_flat_rolling<8, 4, 1, 2049><<<256, 1024>>>(d + (bytes / parts) * q, nw / parts, out, dig + (nw / parts) * q); });
    }
    ROLL(1, 69, "LDS lookups + digest stores every fourth group, whole lines") ROLL(4, 69, "LDS lookups + digest stores every fourth group, whole lines")
    ROLL(1, 9, "LDS lookups + serial fold") ROLL(2, 9, "LDS lookups + serial fold") ROLL(4, 9, "LDS lookups + serial fold")
    ROLL(1, 13, "LDS lookups + serial fold + digest stores") ROLL(2, 13, "LDS lookups + serial fold + digest stores")
    ROLL(4, 13, "LDS lookups + serial fold + digest stores")
#undef ROLL
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "flat")) {   // bare reads in the small-window kernels' pattern, window size x lanes per window
    std::vector<uint32_t> sizes = {512u, 2048u, 4096u, 8192u, 16384u, 32768u, 65536u, 131072u};
    if (argc > 3 && !strcmp(argv[3], "large")) sizes = {262144u, 524288u, 1048576u};
    for (uint32_t w : sizes) {
      char name[96];
      const uint64_t nw = bytes / w;
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, LDS held", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 16 lanes, LDS held", w);
      timeit(name, bytes, [&] { read_flat_pattern<16, 1><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 32 lanes, LDS held", w);
      timeit(name, bytes, [&] { read_flat_pattern<32, 1><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, 2 workgroups per CU", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 0><<<512, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, phase by wavefront", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 1, 1><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, phase by wavefront and workgroup", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 1, 2><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 16 lanes, phase by wavefront", w);
      timeit(name, bytes, [&] { read_flat_pattern<16, 1, 1, 1><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 32 lanes, phase by wavefront", w);
      timeit(name, bytes, [&] { read_flat_pattern<32, 1, 1, 1><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, power-of-two tails", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 1, 3><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 16 lanes, power-of-two tails", w);
      timeit(name, bytes, [&] { read_flat_pattern<16, 1, 1, 3><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 32 lanes, power-of-two tails", w);
      timeit(name, bytes, [&] { read_flat_pattern<32, 1, 1, 3><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, 4 phases", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 1, 4><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 32 lanes, 4 phases", w);
      timeit(name, bytes, [&] { read_flat_pattern<32, 1, 1, 4><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, 8 phases", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 1, 8><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 32 lanes, 8 phases", w);
      timeit(name, bytes, [&] { read_flat_pattern<32, 1, 1, 8><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, windows 3 apart", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 3><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, windows 17 apart", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 17><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows,  8 lanes, windows 257 apart", w);
      timeit(name, bytes, [&] { read_flat_pattern<8, 1, 257><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 16 lanes, windows 17 apart", w);
      timeit(name, bytes, [&] { read_flat_pattern<16, 1, 17><<<256, 1024>>>(d, nw, w, out); });
      snprintf(name, sizeof name, "flat pattern %6u B windows, 32 lanes, windows 17 apart", w);
      timeit(name, bytes, [&] { read_flat_pattern<32, 1, 17><<<256, 1024>>>(d, nw, w, out); });
    }
    return 0;
  }
  timeit("gridstride 2048x256 unroll4", bytes, [&] { read_gridstride<4, false><<<2048, 256>>>(p, n16, out); });
  timeit("gridstride 2048x256 unroll8", bytes, [&] { read_gridstride<8, false><<<2048, 256>>>(p, n16, out); });
  timeit("gridstride 2048x256 unroll8 nt", bytes, [&] { read_gridstride<8, true><<<2048, 256>>>(p, n16, out); });
  timeit("gridstride 4096x256 unroll4", bytes, [&] { read_gridstride<4, false><<<4096, 256>>>(p, n16, out); });
  timeit("gridstride 1024x256 unroll8", bytes, [&] { read_gridstride<8, false><<<1024, 256>>>(p, n16, out); });
  timeit("wave-windows 256x1024 unroll4", bytes, [&] { read_wave_windows<4, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows 256x1024 unroll8", bytes, [&] { read_wave_windows<8, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows 256x1024 unroll8 nt", bytes, [&] { read_wave_windows<8, true><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows 256x1024 unroll16", bytes, [&] { read_wave_windows<16, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("block-windows 256x1024 unroll4", bytes, [&] { read_block_windows<4, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("block-windows 256x1024 unroll8", bytes, [&] { read_block_windows<8, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("block-windows 256x1024 unroll8 nt", bytes, [&] { read_block_windows<8, true><<<256, 1024>>>(d, nwin, win, out); });
  timeit("block-windows 512x1024 unroll8", bytes, [&] { read_block_windows<8, false><<<512, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm plain", bytes, [&] { read_wave_windows_policy<8, 0, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm nt", bytes, [&] { read_wave_windows_policy<8, 1, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm sc1 nt", bytes, [&] { read_wave_windows_policy<8, 2, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm sc0 sc1 nt", bytes, [&] { read_wave_windows_policy<8, 3, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm sc1", bytes, [&] { read_wave_windows_policy<8, 4, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm nt staggered", bytes, [&] { read_wave_windows_policy<8, 1, true><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm nt unroll4", bytes, [&] { read_wave_windows_policy<4, 1, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("wave-windows asm nt unroll2", bytes, [&] { read_wave_windows_policy<2, 1, false><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc pattern G=1 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<1><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc pattern G=2 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<2><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc pattern G=4 (prefetch 1, nt)", bytes, [&] { read_crc_pattern<4><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=2 ROWS=1 PF=1", bytes, [&] { read_crc_rows<2, 1, 1><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=2 ROWS=1 PF=2", bytes, [&] { read_crc_rows<2, 1, 2><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=2 ROWS=2 PF=2", bytes, [&] { read_crc_rows<2, 2, 2><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=2 ROWS=4 PF=1", bytes, [&] { read_crc_rows<2, 4, 1><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=1 ROWS=4 PF=1", bytes, [&] { read_crc_rows<1, 4, 1><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=4 ROWS=1 PF=2", bytes, [&] { read_crc_rows<4, 1, 2><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=4 ROWS=4 PF=1", bytes, [&] { read_crc_rows<4, 4, 1><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=8 ROWS=2 PF=1", bytes, [&] { read_crc_rows<8, 2, 1><<<256, 1024>>>(d, nwin, win, out); });
  timeit("crc rows G=2 ROWS=2 PF=1 512 WGs", bytes, [&] { read_crc_rows<2, 2, 1><<<512, 1024>>>(d, nwin, win, out); });
  timeit("hipMemsetAsync (write only)", bytes, [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
  return 0;
}
