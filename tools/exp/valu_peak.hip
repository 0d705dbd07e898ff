// Chip-wide VALU issue rate under a full-occupancy integer load (what clock does the chip hold?): 8 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define S4(x) x x x x
#define S16(x) S4(x) S4(x) S4(x) S4(x)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  uint32_t u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 * 7 + 5, u3 = 77;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { S16(asm volatile("v_min3_i32 %0, %0, %1, %2\n v_max3_i32 %2, %2, %1, %3\n v_mad_u32_u24 %3, %3, %1, %0\n v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
    if (OP == 1) { S16(asm volatile("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot2_u32_u16 %2, %2, %1, %3\n v_alignbyte_b32 %3, %3, %1, 1\n v_perm_b32 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
  }
  if (u0 == 0x12345678u && u2 == 99 && u3 == 98) out[0] = 1;
}
int main() {
  uint32_t* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int op = 0; op < 2; ++op)
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 4000, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
      hipEventRecord(e0);
      if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double winstr = (double)blocks * 4 * iters * 64;
      printf("op %d: %.3f ms, %.1f G wave-instr/s  => %.3f GHz at 256 CUs x 4 SIMDs x 1/4 per cycle\n", op, ms, winstr / ms / 1e6, winstr / ms / 1e6 / 256.0);
    }
  return 0;
}
