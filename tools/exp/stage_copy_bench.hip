// Diagnostic: what the staging of two 1241x376 images costs on the host (row-wise memcpy into page-locked memory, plain and with streaming
// stores), what hipPointerGetAttributes costs, and the upload itself from staging / straight from page-locked rows.
//   hipcc -O3 --offload-arch=gfx950 -mavx2 tools/exp/stage_copy_bench.hip -o tools/exp/bin/stage_copy_bench
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_stream(uint8_t* dst, const uint8_t* src, size_t n) {  // dst 32-byte aligned
  size_t i = 0;
  for (; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i*)(dst + i), _mm256_loadu_si256((const __m256i*)(src + i)));
  if (i < n) std::memcpy(dst + i, src + i, n - i);
}

int main() {
  const int W = 1241, H = 376, S = 1280;
  std::vector<uint8_t> a((size_t)W * H), b((size_t)W * H);
  for (size_t i = 0; i < a.size(); ++i) a[i] = (uint8_t)(i * 7), b[i] = (uint8_t)(i * 13);
  uint8_t *stage, *pin_a, *pin_b, *dev;
  hipHostMalloc(&stage, (size_t)2 * S * H), hipHostMalloc(&pin_a, (size_t)W * H), hipHostMalloc(&pin_b, (size_t)W * H);
  hipMalloc(&dev, (size_t)4 * S * H);
  std::memcpy(pin_a, a.data(), a.size()), std::memcpy(pin_b, b.data(), b.size());
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int reps = 300;
  auto med = [&](auto f) {
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
      for (size_t i = 0; i < a.size(); i += 64) a[i] ^= 1, b[i] ^= 1;  // the producer touched the images last
      const double t0 = now_us();
      f();
      t.push_back(now_us() - t0);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  const uint8_t* srcs[2] = {a.data(), b.data()};
  printf("rows memcpy into staging        %7.1f us\n", med([&] {
           for (int i = 0; i < 2; ++i)
             for (int y = 0; y < H; ++y) std::memcpy(stage + ((size_t)i * H + y) * S, srcs[i] + (size_t)y * W, W);
         }));
  printf("rows streaming stores           %7.1f us\n", med([&] {
           for (int i = 0; i < 2; ++i)
             for (int y = 0; y < H; ++y) copy_stream(stage + ((size_t)i * H + y) * S, srcs[i] + (size_t)y * W, W);
           _mm_sfence();
         }));
  printf("one memcpy per image (dense)    %7.1f us\n", med([&] {
           for (int i = 0; i < 2; ++i) std::memcpy(stage + (size_t)i * W * H, srcs[i], (size_t)W * H);
         }));
  hipPointerAttribute_t at;
  printf("hipPointerGetAttributes pinned  %7.2f us\n", med([&] { (void)hipPointerGetAttributes(&at, pin_a); (void)hipPointerGetAttributes(&at, pin_b); }) / 2);
  printf("  -> type %d\n", (int)at.type);
  printf("hipPointerGetAttributes pageable%7.2f us", med([&] {
           hipError_t e = hipPointerGetAttributes(&at, a.data());
           if (e != hipSuccess) (void)hipGetLastError();
         }));
  printf("  -> err %d type %d\n", (int)hipPointerGetAttributes(&at, a.data()), (int)at.type);
  (void)hipGetLastError();
  printf("upload staging, one 2D copy     %7.1f us\n", med([&] {
           hipMemcpy2DAsync(dev, (size_t)2 * S * H, stage, (size_t)S * H, (size_t)S * H, 2, hipMemcpyHostToDevice, st);
           hipStreamSynchronize(st);
         }));
  printf("upload page-locked rows, 2 x 2D %7.1f us\n", med([&] {
           hipMemcpy2DAsync(dev, S, pin_a, W, W, H, hipMemcpyHostToDevice, st);
           hipMemcpy2DAsync(dev + (size_t)2 * S * H, S, pin_b, W, W, H, hipMemcpyHostToDevice, st);
           hipStreamSynchronize(st);
         }));
  printf("upload page-locked dense, 2 x 1D%7.1f us\n", med([&] {
           hipMemcpyAsync(dev, pin_a, (size_t)W * H, hipMemcpyHostToDevice, st);
           hipMemcpyAsync(dev + (size_t)2 * S * H, pin_b, (size_t)W * H, hipMemcpyHostToDevice, st);
           hipStreamSynchronize(st);
         }));
  return 0;
}
