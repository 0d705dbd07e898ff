// Issue cost (cycles per wave-instruction, one wave alone on its SIMD) of the vector instructions the front end leans on.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/exp/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITER 256
#define S1(x) x
#define S4(x) x x x x
#define S16(x) S4(x) S4(x) S4(x) S4(x)
#define S64(x) S16(x) S16(x) S16(x) S16(x)
template <int OP>
__global__ void k(uint64_t* out, double seed) {
  double d0 = seed + threadIdx.x, d1 = seed * 2, d2 = seed * 3, d3 = seed * 5;
  float f0 = (float)seed, f1 = f0 * 2, f2 = f0 * 3, f3 = f0 * 7;
  uint32_t u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 * 7 + 5, u3 = 77;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) { S16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(f0) : "v"(f1), "v"(f2), "v"(f3), "v"(f1));) }
    if (OP == 1) { S16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %2, %2, %1\n v_add_f64 %3, %3, %1\n v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d1), "v"(d2), "v"(d3));) }
    if (OP == 2) { S16(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %2, %2, %1\n v_mul_f64 %3, %3, %1\n v_mul_f64 %0, %0, %1" : "+v"(d0) : "v"(d1), "v"(d2), "v"(d3));) }
    if (OP == 3) { S16(asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f32_f64 %2, %3\n v_cvt_f32_f64 %4, %5\n v_cvt_f32_f64 %6, %7" : "+v"(f0) : "v"(d0), "v"(f1), "v"(d1), "v"(f2), "v"(d2), "v"(f3), "v"(d3));) }
    if (OP == 4) { S16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 5) { S16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %1, %3\n v_mad_u32_u24 %3, %3, %1, %0\n v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 6) { S16(asm volatile("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %2, %2, %1, %3\n v_dot4_u32_u8 %3, %3, %1, %0\n v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 7) { S16(asm volatile("v_dot2_u32_u16 %0, %0, %1, %2\n v_dot2_u32_u16 %2, %2, %1, %3\n v_dot2_u32_u16 %3, %3, %1, %0\n v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 8) { S16(asm volatile("v_min3_i32 %0, %0, %1, %2\n v_min3_i32 %2, %2, %1, %3\n v_min3_i32 %3, %3, %1, %0\n v_min3_i32 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 9) { S16(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %2, %2, %1, %3\n v_perm_b32 %3, %3, %1, %0\n v_perm_b32 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 10) { S16(asm volatile("v_mul_hi_u32_u24 %0, %0, %1\n v_mul_hi_u32_u24 %2, %2, %1\n v_mul_hi_u32_u24 %3, %3, %1\n v_mul_hi_u32_u24 %0, %0, %1" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 11) { S16(asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_min_u16 %2, %2, %1\n v_pk_min_u16 %3, %3, %1\n v_pk_min_u16 %0, %0, %1" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 12) { S16(asm volatile("v_sad_u32 %0, %0, %1, %2\n v_sad_u32 %2, %2, %1, %3\n v_sad_u32 %3, %3, %1, %0\n v_sad_u32 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 13) { S16(asm volatile("v_cvt_f64_i32 %0, %1\n v_cvt_f64_i32 %2, %3\n v_cvt_f64_i32 %4, %5\n v_cvt_f64_i32 %0, %3" : "+v"(d0) : "v"(u0), "v"(d1), "v"(u1), "v"(d2), "v"(u2));) }
    if (OP == 14) { S16(asm volatile("v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %2, %2, %1, 2\n v_alignbyte_b32 %3, %3, %1, 3\n v_alignbyte_b32 %0, %0, %1, 1" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 15) { S16(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %3, %3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 16) { S16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 17) { S16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %2, %2, %1, %3\n v_fma_f64 %3, %3, %1, %0\n v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d1), "v"(d2), "v"(d3));) }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (f0 == 1.2345f || d0 == 1.2345 || u0 == 0x12345678u || d2 == 77.5 || d3 == 3.25 || u2 == 99 || u3 == 98 || f2 == 5.f || f3 == 6.f) out[1000] = 1;
}
template <int OP> void run(const char* name, uint64_t* d) {
  hipLaunchKernelGGL(k<OP>, dim3(8), dim3(64), 0, 0, d, 1.0);
  hipDeviceSynchronize();
  uint64_t h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // s_memtime counts at a constant 100 MHz; report against wall-clock only as a ratio to v_add_f32
  printf("%-18s %8.3f ticks/1k-instr\n", name, (double)h[0] / (ITER * 64 / 1000.0));
}
int main() {
  uint64_t* d; hipMalloc(&d, 8192);
  for (int w = 0; w < 2; ++w) {
  run<0>("v_add_f32", d); run<1>("v_add_f64", d); run<2>("v_mul_f64", d); run<17>("v_fma_f64", d); run<3>("v_cvt_f32_f64", d); run<13>("v_cvt_f64_i32", d);
  run<4>("v_mul_lo_u32", d); run<5>("v_mad_u32_u24", d); run<10>("v_mul_hi_u32_u24", d); run<6>("v_dot4_u32_u8", d); run<7>("v_dot2_u32_u16", d);
  run<8>("v_min3_i32", d); run<9>("v_perm_b32", d); run<11>("v_pk_min_u16", d); run<12>("v_sad_u32", d); run<14>("v_alignbyte_b32", d);
  run<15>("v_add_u32_sdwa", d); run<16>("v_mov_b32_dpp", d);
  }
  return 0;
}
