// How fast can the machine START one-wave workgroups?  k_fast launches 1.26 M of them per 512-pair step (one cell each); this times empty
// kernels of the same grid shape by LDS request and workgroup width.  hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/dispatch_rate tools/exp/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_empty(uint32_t* out, int spin) {
  extern __shared__ uint32_t lds[];
  uint32_t v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;   // ~ 3 vector instructions per trip
  if (v == 0x12345678u) { lds[threadIdx.x] = v; out[blockIdx.x] = lds[(threadIdx.x + 1) & 63]; }
}
int main() {
  uint32_t* d; hipMalloc(&d, 1 << 22);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int cells = 450, imgs = 1024;   // level 0 of 1241x376: ~ 440 cells per image
  for (int spin : {0, 100, 270}) {      // 270 trips ~ 810 vector instructions: k_fast's count
    for (int lds : {0, 4736, 5120, 5632, 6144}) {
      for (int width : {64, 256}) {
        const int n_wg = cells * imgs * 64 / width;
        hipLaunchKernelGGL(k_empty, dim3(n_wg), dim3(width), lds * width / 64, 0, d, spin);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_empty, dim3(n_wg), dim3(width), lds * width / 64, 0, d, spin);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("spin %3d lds/wave %5d B  wg width %3d: %7.3f ms for %d waves = %6.1f M waves/s\n", spin, lds, width, ms, cells * imgs, cells * imgs / ms / 1e3);
      }
    }
  }
  return 0;
}
