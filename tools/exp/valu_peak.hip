// Chip-wide VALU issue rate (is a wave64 vector instruction 2 or 4 cycles on a gfx950 SIMD?).
// Every op class is run at 1, 2, 4 and 8 waves per SIMD on every CU; each wave issues a long stream of INDEPENDENT instructions of
// one class (eight accumulators, so a lone wave is bound by issue and not by the dependent latency).  Two figures per run:
//   wall   G wave-instructions/s over the whole chip (HIP events)          -> the ceiling roofline_valu prices against
//   cyc    shader cycles per wave-instruction PER SIMD from s_memtime inside the kernel (lane 0 of every wave: the wave's own
//          instruction count x waves per SIMD / its elapsed ticks), median over the waves -> independent of DVFS
// MI355X_MICROARCH.md:54, 473 lists v_fma_f32 at 2 cycles with several waves per SIMD; profiles/r2_valu_peak.txt measured 4 for the
// integer mixes at 8 waves.  Build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define S4(x) x x x x
#define S8(x) S4(x) S4(x)
enum { OP_INT_MIX, OP_DOT_MIX, OP_FMA_F32, OP_ADD_F32, OP_PK_FMA_F32, OP_ADD_U32, OP_MIN3_I32, OP_MAD_I24, OP_MAX_I32, OP_CMP_CNDMASK, OP_LSHL_OR, OP_COUNT };
static const char* kNames[OP_COUNT] = {"int mix (min3, max3, mad_u24, add)", "dot mix (dot4_u8, dot2_u16, alignbyte, perm)", "v_fma_f32", "v_add_f32",
                                       "v_pk_fma_f32 (2 fp32 FMAs per lane)", "v_add_u32", "v_min3_i32", "v_mad_i32_i24", "v_max_i32",
                                       "v_cmp_gt_i32 + v_cndmask_b32 (pairs)", "v_lshl_or_b32"};
// instructions per loop trip (all variants: 8 accumulators x 4)
#define PER_TRIP 32
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* ticks, int iters) {
  uint32_t a[8];
  float f[8];
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * (2 * i + 3) + i, f[i] = (float)a[i] * 1e-3f, p[i] = float2v{f[i], f[i] + 1.f};
  const uint32_t c1 = threadIdx.x | 1u, c2 = 0x01020304u;
  const float g1 = 1.0001f, g2 = 1e-7f;
  const float2v q1 = {1.0001f, 0.9999f}, q2 = {1e-7f, 2e-7f};
  extern __shared__ uint32_t lds_dummy[];
  if (iters < 0) lds_dummy[threadIdx.x] = 1;  // (keeps the allocation)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == OP_INT_MIX) {
          if (r == 0) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
          if (r == 1) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
          if (r == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
          if (r == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        } else if (OP == OP_DOT_MIX) {
          if (r == 0) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
          if (r == 1) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
          if (r == 2) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(c1));
          if (r == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        } else if (OP == OP_FMA_F32) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g1), "v"(g2));
        } else if (OP == OP_ADD_F32) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g2));
        } else if (OP == OP_PK_FMA_F32) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q1), "v"(q2));
        } else if (OP == OP_ADD_U32) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        } else if (OP == OP_MIN3_I32) {
          asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        } else if (OP == OP_MAD_I24) {
          asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        } else if (OP == OP_MAX_I32) {
          asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        } else if (OP == OP_CMP_CNDMASK) {
          if (r & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1) : "vcc");
          else asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(c2) : "vcc");
        } else if (OP == OP_LSHL_OR) {
          asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c1));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  float fs = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i], fs += f[i] + p[i].x + p[i].y;
  if (s == 0x12345678u && fs == 3.25f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(uint32_t* d, unsigned long long* d_ticks, int n_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int iters = 8000 / wps, blocks = n_cu * wps;  // wps workgroups of 4 waves per CU = wps waves per SIMD
    // ... enforced by the LDS request: exactly wps workgroups fit a CU's 160 KB, so the dispatcher cannot stack them unevenly
    const int lds = (160 * 1024) / wps - (wps == 1 ? 0 : 512);
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    double best_g = 0, med_cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, d_ticks, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double winstr = (double)blocks * 4 * iters * PER_TRIP;
      std::vector<unsigned long long> t(blocks * 4);
      hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost);
      std::sort(t.begin(), t.end());
      const double cyc = (double)t[t.size() / 2] / ((double)iters * PER_TRIP * wps);
      if (winstr / ms / 1e6 > best_g) best_g = winstr / ms / 1e6, med_cyc = cyc;
    }
    printf("  %-46s %d waves/SIMD: %7.1f G wave-instr/s (wall, best of 3)   %.2f cycles per wave-instruction per SIMD (s_memtime, median wave)\n",
           kNames[OP], wps, best_g, med_cyc);
  }
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int n_cu = pr.multiProcessorCount;
  printf("%s, %d CUs, %d MHz\n", pr.name, n_cu, pr.clockRate / 1000);
  uint32_t* d;
  unsigned long long* dt;
  hipMalloc(&d, 64);
  hipMalloc(&dt, (size_t)n_cu * 8 * 4 * 8);
  run<OP_INT_MIX>(d, dt, n_cu);
  run<OP_DOT_MIX>(d, dt, n_cu);
  run<OP_FMA_F32>(d, dt, n_cu);
  run<OP_ADD_F32>(d, dt, n_cu);
  run<OP_PK_FMA_F32>(d, dt, n_cu);
  run<OP_ADD_U32>(d, dt, n_cu);
  run<OP_MIN3_I32>(d, dt, n_cu);
  run<OP_MAD_I24>(d, dt, n_cu);
  run<OP_MAX_I32>(d, dt, n_cu);
  run<OP_CMP_CNDMASK>(d, dt, n_cu);
  run<OP_LSHL_OR>(d, dt, n_cu);
  printf("ceiling at 4 cycles: %d CUs x 4 SIMDs x clock / 4 = %.1f G wave-instr/s at 2.4 GHz; at 2 cycles twice that\n", n_cu, n_cu * 4 * 2.4 / 4);
  return 0;
}
