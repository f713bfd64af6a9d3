// Microbenchmarks of the CRC inner loop's resources on MI355X: LDS gathers from the 32-copy table layout,
// the VALU work around them, and both together.  Prints wave-cycles (s_memtime) and wall time per row-step.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_bench tools/lds_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 2048;

__device__ __forceinline__ uint32_t lds_u32(const char* base, uint32_t off) { return *reinterpret_cast<const uint32_t*>(base + off); }

// MODE 0: full step (addresses from accumulators, 16 lookups, xors) = the real loop without global loads
// MODE 1: lookups only: addresses advance by a cheap add, results xored (16 ds_read + ~20 valu)
// MODE 2: VALU only: address math + xors on register data, no LDS
// MODE 3: like 0 but ds_read_b64 on a 64-copy... (8-byte entries, 256 B rows): 8 lookups of 64 bits
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768; i += 1024) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  const uint32_t lb01 = (lane & 31) * 4u, lb23 = lb01 + 65536u;
  uint32_t a0 = tid * 17 + 1, a1 = tid * 31 + 2, a2 = tid * 101 + 3, a3 = tid * 7 + 4;
  uint32_t d = tid;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; it++) {
    if (MODE == 0) {
#define STEP(a)                                                                                 \
  {                                                                                             \
    const uint32_t o0 = lb01 | ((a << 7) & 0x7f80u), o1 = lb01 | ((a >> 1) & 0x7f80u),          \
                   o2 = lb23 | ((a >> 9) & 0x7f80u), o3 = lb23 | ((a >> 17) & 0x7f80u);         \
    a = lds_u32(lds, o0) ^ lds_u32(lds, o1 + 32768u) ^ lds_u32(lds, o2) ^ lds_u32(lds, o3 + 32768u) ^ d; \
  }
      STEP(a0) STEP(a1) STEP(a2) STEP(a3)
    } else if (MODE == 1) {
      uint32_t o = lb01 | ((a0 << 7) & 0x7f80u);
      uint32_t x = 0;
#pragma unroll
      for (int q = 0; q < 16; q++) x ^= lds_u32(lds, (o + q * 4096u) & 0x1ffffu);
      a0 = x + d;
    } else if (MODE == 2) {
#define VSTEP(a)                                                                               \
  {                                                                                             \
    const uint32_t o0 = lb01 | ((a << 7) & 0x7f80u), o1 = lb01 | ((a >> 1) & 0x7f80u),          \
                   o2 = lb23 | ((a >> 9) & 0x7f80u), o3 = lb23 | ((a >> 17) & 0x7f80u);         \
    a = (o0 * 3) ^ (o1 + 32768u) ^ (o2 * 5) ^ (o3 + 77u) ^ d;                                   \
  }
      VSTEP(a0) VSTEP(a1) VSTEP(a2) VSTEP(a3)
    } else {
      // 64-bit entries: row = 256 B (32 copies x 8 B); two byte tables of 64 KiB
      const uint32_t l8 = (lane & 31) * 8u;
#define STEP64(a, b)                                                                            \
  {                                                                                             \
    const uint32_t o0 = l8 | ((a << 8) & 0xff00u), o1 = l8 | (a & 0xff00u);                      \
    const uint32_t o2 = l8 | ((b << 8) & 0xff00u), o3 = l8 | (b & 0xff00u);                      \
    const uint2 r0 = *reinterpret_cast<const uint2*>(lds + o0), r1 = *reinterpret_cast<const uint2*>(lds + o1 + 65536u); \
    const uint2 r2 = *reinterpret_cast<const uint2*>(lds + o2), r3 = *reinterpret_cast<const uint2*>(lds + o3 + 65536u); \
    a = r0.x ^ r1.x ^ r2.x ^ r3.x ^ d;                                                          \
    b = r0.y ^ r1.y ^ r2.y ^ r3.y ^ d;                                                          \
  }
      STEP64(a0, a1) STEP64(a2, a3) STEP64(a0, a1) STEP64(a2, a3)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x1234567u) out[0] = 1;
  if (lane == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name, uint32_t* out, unsigned long long* dcyc) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) k<MODE><<<256, 1024, 131072>>>(out, dcyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; i++) k<MODE><<<256, 1024, 131072>>>(out, dcyc);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  static unsigned long long h[4096];
  CK(hipMemcpy(h, dcyc, sizeof h, hipMemcpyDeviceToHost));
  double s = 0; for (auto v : h) s += v;
  const double wave_cyc_per_it = s / 4096 / kIters;
  // one iteration of one wave = one 1 KiB row-step; 16 waves per CU
  printf("%-34s %.3f ms  wave-cycles/step %.1f  => CU cycles per 1 KiB step %.1f, clock %.2f GHz, equiv %.1f TB/s\n", name, ms,
         wave_cyc_per_it, wave_cyc_per_it / 16, s / 4096 / (ms * 1e-3) / 1e9, 256.0 * 16 * kIters * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  uint32_t* out; unsigned long long* dcyc;
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&dcyc, 4096 * 8));
  run<0>("full step (16 ds_read_b32 + valu)", out, dcyc);
  run<1>("16 ds_read_b32, little valu", out, dcyc);
  run<2>("valu only", out, dcyc);
  run<3>("8 ds_read_b64 x2 (same lookups/byte)", out, dcyc);
  return 0;
}
