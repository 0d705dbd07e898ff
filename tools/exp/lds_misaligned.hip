// LDS read cost by alignment: ds_read_b32 at (4 * lane + mis), mis = 0..3, beside ds_read_u8 / ds_read_u16 -- ticks per 1k wave-instructions
// with 1, 4 and 8 waves per SIMD resident (throughput, not latency).  hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_mis tools/exp/lds_misaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 256
template <int OP>
__global__ void k(uint64_t* out, int mis) {
  __shared__ uint32_t buf[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)buf + 4 * lane + mis + (threadIdx.x >> 6) * 1024;
  uint32_t acc = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; ++i) {
    uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
    if (OP == 0)
      asm volatile("ds_read_b32 %0, %8 offset:0\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                   "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a));
    if (OP == 1)
      asm volatile("ds_read_u8 %0, %8 offset:0\n ds_read_u8 %1, %8 offset:256\n ds_read_u8 %2, %8 offset:512\n ds_read_u8 %3, %8 offset:768\n"
                   "ds_read_u8 %4, %8 offset:1024\n ds_read_u8 %5, %8 offset:1280\n ds_read_u8 %6, %8 offset:1536\n ds_read_u8 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a));
    if (OP == 2)
      asm volatile("ds_read_u16 %0, %8 offset:0\n ds_read_u16 %1, %8 offset:256\n ds_read_u16 %2, %8 offset:512\n ds_read_u16 %3, %8 offset:768\n"
                   "ds_read_u16 %4, %8 offset:1024\n ds_read_u16 %5, %8 offset:1280\n ds_read_u16 %6, %8 offset:1536\n ds_read_u16 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a));
    acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345678u) out[4000] = acc;
}
template <int OP> void run(const char* name, uint64_t* d, int mis, int waves) {
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, d, mis);
  hipDeviceSynchronize();
  uint64_t h[16];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("%-12s mis %d waves/CU %2d: %8.1f ticks per 1k wave-reads (all waves together)\n", name, mis, waves, (double)mx / (ITER * 8 * waves / 1000.0));
}
int main() {
  uint64_t* d; hipMalloc(&d, 65536);
  for (int waves : {1, 4, 16}) {
    for (int mis = 0; mis < 4; ++mis) run<0>("ds_read_b32", d, mis, waves);
    for (int mis = 0; mis < 2; ++mis) run<1>("ds_read_u8", d, mis, waves);
    for (int mis = 0; mis < 2; ++mis) run<2>("ds_read_u16", d, mis, waves);
  }
  return 0;
}
