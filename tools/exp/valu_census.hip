// Issue rate per vector-instruction CLASS on gfx950 (follow-up of valu_peak.hip, r5): which opcodes go through a SIMD at the fast rate
// (~920 G wave-instr/s chip-wide: v_add_f32, v_add_u32, v_fma_f32 in valu_peak) and which at the slow one (~575 G: v_min3_i32,
// v_mad_i32_i24, v_max_i32, the dot / perm family).  Eight independent accumulators per lane, 2 and 8 waves per SIMD (LDS request pins the
// occupancy), wall time by HIP events, best of 3.  Build: hipcc -O3 --offload-arch=gfx950 valu_census.hip -o valu_census
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#define OPS(X)                                                                   \
  X(0, "v_add_u32 %0, %0, %1", a, c1, c2)                                        \
  X(1, "v_sub_u32 %0, %0, %1", a, c1, c2)                                        \
  X(2, "v_and_b32 %0, %0, %1", a, c1, c2)                                        \
  X(3, "v_or_b32 %0, %0, %1", a, c1, c2)                                         \
  X(4, "v_xor_b32 %0, %0, %1", a, c1, c2)                                        \
  X(5, "v_lshlrev_b32 %0, 1, %0", a, c1, c2)                                     \
  X(6, "v_lshrrev_b32 %0, 1, %0", a, c1, c2)                                     \
  X(7, "v_mov_b32 %0, %1", a, c1, c2)                                            \
  X(8, "v_max_i32 %0, %0, %1", a, c1, c2)                                        \
  X(9, "v_min_u32 %0, %0, %1", a, c1, c2)                                        \
  X(10, "v_max_f32 %0, %0, %1", f, g1, g2)                                       \
  X(11, "v_min_f32 %0, %0, %1", f, g1, g2)                                       \
  X(12, "v_min3_f32 %0, %0, %1, %2", f, g1, g2)                                  \
  X(13, "v_max3_f32 %0, %0, %1, %2", f, g1, g2)                                  \
  X(14, "v_med3_f32 %0, %0, %1, %2", f, g1, g2)                                  \
  X(15, "v_sub_f32 %0, %0, %1", f, g1, g2)                                       \
  X(16, "v_mul_f32 %0, %0, %1", f, g1, g2)                                       \
  X(17, "v_fma_f32 %0, %0, %1, %2", f, g1, g2)                                   \
  X(18, "v_fmac_f32 %0, %1, %2", f, g1, g2)                                       \
  X(19, "v_cvt_f32_ubyte0 %0, %0", a, c1, c2)                                    \
  X(20, "v_cvt_f32_ubyte3 %0, %0", a, c1, c2)                                    \
  X(21, "v_cvt_f32_i32 %0, %0", a, c1, c2)                                       \
  X(22, "v_cvt_i32_f32 %0, %0", a, c1, c2)                                       \
  X(23, "v_add3_u32 %0, %0, %1, %2", a, c1, c2)                                  \
  X(24, "v_bfe_u32 %0, %0, 8, 8", a, c1, c2)                                     \
  X(25, "v_mul_u32_u24 %0, %0, %1", a, c1, c2)                                   \
  X(26, "v_mul_i32_i24 %0, %0, %1", a, c1, c2)                                   \
  X(27, "v_sad_u8 %0, %0, %1, %2", a, c1, c2)                                    \
  X(28, "v_sad_u16 %0, %0, %1, %2", a, c1, c2)                                   \
  X(29, "v_and_or_b32 %0, %0, %1, %2", a, c1, c2)                                \
  X(30, "v_pk_min_i16 %0, %0, %1", a, c1, c2)                                    \
  X(31, "v_pk_max_i16 %0, %0, %1", a, c1, c2)                                    \
  X(32, "v_pk_sub_i16 %0, %0, %1", a, c1, c2)                                    \
  X(33, "v_pk_add_u16 %0, %0, %1", a, c1, c2)                                    \
  X(34, "v_pk_min_f16 %0, %0, %1", a, c1, c2)                                    \
  X(35, "v_pk_max_f16 %0, %0, %1", a, c1, c2)                                    \
  X(36, "v_pk_add_f16 %0, %0, %1", a, c1, c2)                                    \
  X(37, "v_pk_fma_f16 %0, %0, %1, %2", a, c1, c2)                                \
  X(38, "v_min3_f16 %0, %0, %1, %2", a, c1, c2)                                  \
  X(39, "v_min3_i16 %0, %0, %1, %2", a, c1, c2)                                  \
  X(40, "v_min_i16 %0, %0, %1", a, c1, c2)                                       \
  X(41, "v_max_f16 %0, %0, %1", a, c1, c2)                                       \
  X(42, "v_cmp_gt_f32 vcc, %0, %1", f, g1, g2)                                   \
  X(43, "v_cmp_gt_i32 vcc, %0, %1", a, c1, c2)                                   \
  X(44, "v_cmp_gt_u32 vcc, %0, %1", a, c1, c2)                                   \
  X(45, "v_cndmask_b32 %0, %0, %1, vcc", a, c1, c2)                              \
  X(46, "v_add_co_u32 %0, vcc, %0, %1", a, c1, c2)                               \
  X(47, "v_lshl_add_u32 %0, %0, 1, %1", a, c1, c2)                               \
  X(48, "v_add_lshl_u32 %0, %0, %1, 1", a, c1, c2)                               \
  X(49, "v_xad_u32 %0, %0, %1, %2", a, c1, c2)                                   \
  X(50, "v_mbcnt_lo_u32_b32 %0, %1, %0", a, c1, c2)                              \
  X(51, "v_min3_i32 %0, %0, %1, %2", a, c1, c2)                                  \
  X(52, "v_mad_i32_i24 %0, %0, %1, %2", a, c1, c2)                               \
  X(53, "v_dot4_u32_u8 %0, %0, %1, %2", a, c1, c2)                               \
  X(54, "v_add_f16 %0, %0, %1", a, c1, c2)                                       \
  X(55, "v_pk_mul_f32 %0, %0, %1", p, q1, q2)                                    \
  X(56, "v_pk_add_f32 %0, %0, %1", p, q1, q2)                                    \
  X(57, "v_sub_co_u32 %0, vcc, %0, %1", a, c1, c2)                               \
  X(58, "v_ashrrev_i32 %0, 1, %0", a, c1, c2)                                    \
  X(59, "v_max_u16 %0, %0, %1", a, c1, c2)                                       \
  X(60, "v_pk_lshlrev_b16 %0, 1, %0", a, c1, c2)                                 \
  X(61, "v_cvt_pk_u8_f32 %0, %1, 0, %0", a, g1, c2)                              \
  X(62, "v_alignbit_b32 %0, %0, %1, 8", a, c1, c2)                               \
  X(63, "v_bfi_b32 %0, %0, %1, %2", a, c1, c2)                                  \
  X(64, "v_sub_u16 %0, %0, %1", a, c1, c2)                                       \
  X(65, "v_min_u16 %0, %0, %1", a, c1, c2)                                       \
  X(66, "v_add_u16 %0, %0, %1", a, c1, c2)                                       \
  X(67, "v_max_i16 %0, %0, %1", a, c1, c2)                                       \
  X(68, "v_cmp_gt_i16 vcc, %0, %1", a, c1, c2)                                   \
  X(69, "v_subrev_u16 %0, %0, %1", a, c1, c2)                                    \
  X(70, "v_lshrrev_b16 %0, 1, %0", a, c1, c2)                                    \
  X(71, "v_mul_lo_u16 %0, %0, %1", a, c1, c2)                                    \
  X(72, "v_max_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", a, c1, c2) \
  X(73, "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3", a, c1, c2) \
  X(74, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", a, c1, c2)   \
  X(75, "v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", a, c1, c2) \
  X(76, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", a, c1, c2)  \
  X(77, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", a, c1, c2)
#define N_OPS 78

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
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
  if (iters < 0) lds_dummy[threadIdx.x] = 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(id, txt, acc, s1, s2) \
  if (OP == id) asm volatile(txt : "+v"(acc[i]) : "v"(s1), "v"(s2) : "vcc");
        OPS(X)
#undef X
      }
    }
  }
  uint32_t s = 0;
  float fs = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i], fs += f[i] + p[i].x + p[i].y;
  if (s == 0x12345678u && fs == 3.25f) out[0] = 1;
}

template <int OP>
static void run(uint32_t* d, int n_cu, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double g[2] = {0, 0};
  int w = 0;
  for (int wps = 2; wps <= 8; wps *= 4, ++w) {
    const int iters = 8000 / wps, blocks = n_cu * wps;
    const int lds = (160 * 1024) / wps - 512;
    (void)hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double winstr = (double)blocks * 4 * iters * 32;
      if (winstr / ms / 1e6 > g[w]) g[w] = winstr / ms / 1e6;
    }
  }
  printf("  %-36s 2 waves/SIMD %7.1f   8 waves/SIMD %7.1f G wave-instr/s   %s\n", name, g[0], g[1], g[1] > 760 ? "FAST" : "slow");
}

template <int OP>
struct Runner {
  static void go(uint32_t* d, int n_cu, const char* const* names) {
    run<OP>(d, n_cu, names[OP]);
    Runner<OP + 1>::go(d, n_cu, names);
  }
};
template <>
struct Runner<N_OPS> {
  static void go(uint32_t*, int, const char* const*) {}
};

int main() {
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  static const char* names[N_OPS] = {
#define X(id, txt, acc, s1, s2) txt,
      OPS(X)
#undef X
  };
  printf("%s, %d CUs; G wave-instructions per second chip-wide (4-cycle issue at 2.4 GHz = 614.4, 2-cycle = 1228.8)\n", pr.name, pr.multiProcessorCount);
  uint32_t* d;
  (void)hipMalloc(&d, 64);
  Runner<0>::go(d, pr.multiProcessorCount, names);
  return 0;
}
