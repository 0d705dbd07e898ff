// FETCH_SIZE / WRITE_SIZE calibration for the load and store SHAPES the front-end kernels use (the bench line's
// `traffic` applied the guide's x2 to every kernel's FETCH_SIZE although the guide calibrates it only for 16-byte-per-lane coalesced
// streams).  Every kernel below moves a KNOWN number of bytes exactly once with one access shape; run it under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- tools/exp/bin/fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -- tools/exp/bin/fetch_calib
// and tools/exp/fetch_calib_summary.py turns the two counter files into profiles/r3_fetch_calibration.json: counter bytes / true bytes
// per shape.  Build: hipcc -O3 --offload-arch=gfx950 tools/exp/fetch_calib.hip -o tools/exp/bin/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));                   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// every kernel: grid-stride over `n_units` units, one unit per lane per trip, checksum to keep the loads alive
// (1) aligned dword per lane: 256 contiguous bytes per wave instruction (k_blur rows, k_quadtree records)
__global__ void k_cal_dword_aligned(const uint8_t* __restrict__ p, size_t n_units, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) acc += *(const uint32_t*)(p + 4 * i);
  if (acc == 0x12345678u) *sink = acc;
}
// (2) byte-aligned dword per lane (address = 4 i + 1): k_ic_moments' rows that start at x - 15
__global__ void k_cal_dword_byte_aligned(const uint8_t* __restrict__ p, size_t n_units, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t v;
    __builtin_memcpy(&v, p + 4 * i + 1, 4);
    acc += v;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (3) 16 bytes per lane, 16-byte aligned, all 64 lanes: the shape the guide's x2 is calibrated on
__global__ void k_cal_dwordx4_aligned(const uint8_t* __restrict__ p, size_t n_units, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = *(const uint4*)(p + 16 * i);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (4) 16 bytes per lane at 4-byte-aligned (not 16-byte-aligned) addresses, 48 of 64 lanes active: k_fast's patch rows
//     (a row of 48 lanes x 16 B = 768 contiguous bytes starting 4 bytes into a line; rows `pitch` bytes apart)
__global__ void k_cal_dwordx4_4aligned_48(const uint8_t* __restrict__ p, size_t n_rows, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n_rows; r += n_waves)
    if (lane < 48) {
      uint4 v;
      __builtin_memcpy(&v, p + r * 768 + 4 + 16 * lane, 16);  // rows are contiguous here: every byte of the buffer is read once
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  if (acc == 0x12345678u) *sink = acc;
}
// (5) 16 bytes per lane at BYTE-aligned addresses, all lanes: k_resize_regions' reads of the caller's images, k_load_level0
__global__ void k_cal_dwordx4_byte_aligned(const uint8_t* __restrict__ p, size_t n_units, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v;
    __builtin_memcpy(&v, p + 16 * i + 3, 16);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (6) gather windows: groups of 8 lanes read 32 contiguous bytes, consecutive groups `pitch` bytes apart (rows of a 31 x 31 /
//     37 x 37 keypoint window: k_ic_moments, k_brief, k_stereo's SAD windows) -- true bytes = the 32-byte segments
__global__ void k_cal_window_rows_32B(const uint8_t* __restrict__ p, size_t n_seg, size_t pitch, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_seg * 8; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t v;
    __builtin_memcpy(&v, p + (i >> 3) * pitch + 5 + 4 * (i & 7), 4);
    acc += v;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// stores: (7) aligned dword, (8) 16 bytes aligned, (9) single bytes (keypoint / descriptor tails)
__global__ void k_cal_store_dword(uint8_t* __restrict__ p, size_t n_units) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) *(uint32_t*)(p + 4 * i) = (uint32_t)i;
}
__global__ void k_cal_store_dwordx4(uint8_t* __restrict__ p, size_t n_units) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x)
    *(uint4*)(p + 16 * i) = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void k_cal_store_byte(uint8_t* __restrict__ p, size_t n_units) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint8_t)i;
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: four times the 256 MiB memory-side cache
  uint8_t* buf = nullptr;
  uint32_t* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes + 4096));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes + 4096));
  const dim3 g(256 * 8), b(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_cal_dword_aligned, g, b, 0, 0, buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_cal_dword_byte_aligned, g, b, 0, 0, buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_cal_dwordx4_aligned, g, b, 0, 0, buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_cal_dwordx4_4aligned_48, g, b, 0, 0, buf, bytes / 768, sink);
    hipLaunchKernelGGL(k_cal_dwordx4_byte_aligned, g, b, 0, 0, buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_cal_window_rows_32B, g, b, 0, 0, buf, bytes / 1280, (size_t)1280, sink);  // 1241-px rows padded to 1280: 32 of every 1280 bytes
    hipLaunchKernelGGL(k_cal_store_dword, g, b, 0, 0, buf, bytes / 4);
    hipLaunchKernelGGL(k_cal_store_dwordx4, g, b, 0, 0, buf, bytes / 16);
    hipLaunchKernelGGL(k_cal_store_byte, g, b, 0, 0, buf, bytes / 4);  // a quarter of the buffer, one byte per lane
    CHECK(hipDeviceSynchronize());
  }
  // true bytes per launch, for the summary script
  printf("TRUE k_cal_dword_aligned %zu\nTRUE k_cal_dword_byte_aligned %zu\nTRUE k_cal_dwordx4_aligned %zu\nTRUE k_cal_dwordx4_4aligned_48 %zu\n"
         "TRUE k_cal_dwordx4_byte_aligned %zu\nTRUE k_cal_window_rows_32B %zu\nTRUE k_cal_store_dword %zu\nTRUE k_cal_store_dwordx4 %zu\nTRUE k_cal_store_byte %zu\n",
         bytes, bytes, bytes, bytes / 768 * 768, bytes, bytes / 1280 * 32, bytes, bytes, bytes / 4);
  return 0;
}
