// Does ds_read_u8_d16_hi preserve the low half of its destination on gfx950 (SRAM-ECC on)?  Prints the packed registers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t* out) {
  __shared__ uint8_t lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  uint32_t r = 0xAAAAAAAAu;
  uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + threadIdx.x;
  asm volatile("ds_read_u8_d16 %0, %1 offset:0\n\tds_read_u8_d16_hi %0, %1 offset:64\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(a) : "memory");
  uint32_t r2 = 0xAAAAAAAAu;
  asm volatile("ds_read_u8_d16_hi %0, %1 offset:64\n\ts_waitcnt lgkmcnt(0)\n\tds_read_u8_d16 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r2) : "v"(a) : "memory");
  out[threadIdx.x] = r;
  out[64 + threadIdx.x] = r2;
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  uint32_t h[128];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; ++i) {
    uint32_t want = (uint32_t)(uint8_t)(i * 7 + 3) | ((uint32_t)(uint8_t)((i + 64) * 7 + 3) << 16);
    if (h[i] != want || h[64 + i] != want) ok = 0;
  }
  printf("lane0: %08x %08x  lane5: %08x %08x  -> %s\n", h[0], h[64], h[5], h[69], ok ? "PRESERVED (packing by the LDS unit works)" : "NOT preserved");
  return 0;
}
