// How fast one lone wavefront issues integer vector instructions on gfx950: dependent chains against independent
// ones, and the instruction kinds SHA-256 / MD5 are made of.  One wavefront per launch; cycles from s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_issue_bench.hip -o tools/valu_issue_bench && tools/valu_issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#include <stdint.h>
#include <stdio.h>

#define UNROLL 64

// HALF: only lanes 0-31 of every wavefront execute the loop (does an instruction whose upper half is masked off issue faster?)
template <int KIND, bool HALF = false>
__global__ void bench(uint32_t* out, uint64_t* cycles, uint32_t seed, int REP) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 7, d = seed * 11 + threadIdx.x;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  uint64_t t0;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  if (HALF && (threadIdx.x & 32)) REP = 0;
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (KIND == 0) {          // one dependent chain of adds
        asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(c), "v"(d));
      } else if (KIND == 1) {   // four independent chains of adds
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if (KIND == 2) {   // dependent rotates
        asm volatile("v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %0, %0, %0, 11\n v_alignbit_b32 %0, %0, %0, 13\n v_alignbit_b32 %0, %0, %0, 5" : "+v"(a));
      } else if (KIND == 3) {   // dependent three-input logic
        a = __builtin_amdgcn_bitop3_b32(a, b, c, 0xCA); a = __builtin_amdgcn_bitop3_b32(a, c, d, 0xE8);
        a = __builtin_amdgcn_bitop3_b32(a, d, b, 0x96); a = __builtin_amdgcn_bitop3_b32(a, b, d, 0xCA);
      } else if (KIND == 4) {   // dependent add3 with a scalar-register constant (VOP3 takes no literal on gfx950)
        asm volatile("v_add3_u32 %0, %0, %1, %4\n v_add3_u32 %0, %0, %2, %4\n v_add3_u32 %0, %0, %3, %4\n v_add3_u32 %0, %0, %1, %4" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(seed));
      } else if (KIND == 5) {   // an MD5 step shape: logic, add3 (literal), add, rotate, add -- all dependent
        uint32_t f, t;
        asm volatile("v_bitop3_b32 %0, %3, %4, %5 bitop3:0xca\n v_add3_u32 %1, %2, %0, %6\n v_add_u32 %1, %1, %6\n"
                     "v_alignbit_b32 %1, %1, %1, 25\n v_add_u32 %1, %3, %1" : "=&v"(f), "=&v"(t) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed));
        a = d; d = c; c = b; b = t;
      } else if (KIND == 6) {   // independent rotates (four chains)
        asm volatile("v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 11\n v_alignbit_b32 %2, %2, %2, 13\n v_alignbit_b32 %3, %3, %3, 5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      }
    }
  }
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  uint64_t t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  out[threadIdx.x] = a ^ b ^ c ^ d;
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND, bool HALF = false>
void run(const char* name, int per_iter, uint32_t* out, uint64_t* cyc, int waves) {
  const int REP = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((bench<KIND, HALF>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, 12345u + rep, REP);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  uint64_t c = 0;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double n = (double)REP * UNROLL * per_iter;
  printf("%-44s waves/WG=%d  %.3f s_memtime ticks, %.3f ns per instruction\n", name, waves, (double)c / n, ms * 1e6 / n);
}

int main() {
  uint32_t* out;
  uint64_t* cyc;
  CK(hipMalloc(&out, 4096 * 4));
  CK(hipMalloc(&cyc, 8));
  for (int waves : {1, 4, 8, 16}) {
    run<0>("dependent v_add_u32", 4, out, cyc, waves);
    run<1>("4 independent v_add_u32 (literal)", 4, out, cyc, waves);
    run<2>("dependent v_alignbit_b32", 4, out, cyc, waves);
    run<6>("4 independent v_alignbit_b32", 4, out, cyc, waves);
    run<3>("dependent v_bitop3_b32", 4, out, cyc, waves);
    run<4>("dependent v_add3_u32 (sgpr operand)", 4, out, cyc, waves);
    run<5>("MD5 step shape (5 dependent)", 5, out, cyc, waves);
    run<0, true>("dependent v_add_u32, lanes 32-63 masked off", 4, out, cyc, waves);
    run<1, true>("4 independent v_add_u32, lanes 32-63 masked off", 4, out, cyc, waves);
    run<2, true>("dependent v_alignbit_b32, lanes 32-63 masked off", 4, out, cyc, waves);
    run<4, true>("dependent v_add3_u32, lanes 32-63 masked off", 4, out, cyc, waves);
  }
  return 0;
}
