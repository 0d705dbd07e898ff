#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
// D = A(16xK) * B(Kx16): find the lane maps with one-hot operands.
__global__ void k32(const long* a, const long* b, int* d) {
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = acc[i];
}
__global__ void k64(const v4i* a, const v4i* b, int* d) {
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = acc[i];
}
template <int K, typename F>
void run(F launch) {
  const int BPL = K / 4;  // bytes per lane
  // hypothesis: lane l holds A[row l&15][k = BPL*(l>>4) + j], B[k = BPL*(l>>4)+j][col l&15]; D reg i of lane l = (row 4*(l>>4)+i, col l&15)
  std::vector<int8_t> A(16 * K), B(K * 16);
  srand(7);
  for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
  for (auto& v : B) v = (int8_t)(rand() % 255 - 127);
  std::vector<int8_t> la(64 * BPL), lb(64 * BPL);
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < BPL; ++j) {
      const int k = BPL * (l >> 4) + j;
      la[l * BPL + j] = A[(l & 15) * K + k];
      lb[l * BPL + j] = B[k * 16 + (l & 15)];
    }
  void *da, *db; int* dd;
  hipMalloc(&da, la.size()); hipMalloc(&db, lb.size()); hipMalloc(&dd, 256 * 4);
  hipMemcpy(da, la.data(), la.size(), hipMemcpyHostToDevice);
  hipMemcpy(db, lb.data(), lb.size(), hipMemcpyHostToDevice);
  launch(da, db, dd);
  std::vector<int> d(256);
  hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * (l >> 4) + i, c = l & 15;
      int ref = 0;
      for (int k = 0; k < K; ++k) ref += (int)A[r * K + k] * (int)B[k * 16 + c];
      if (ref != d[l * 4 + i]) ++bad;
    }
  printf("K=%d: %d of 256 outputs differ from the hypothesis (%s)\n", K, bad, hipGetErrorString(hipGetLastError()));
}
int main() {
  run<32>([](void* a, void* b, int* d) { hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, (const long*)a, (const long*)b, d); hipDeviceSynchronize(); });
  run<64>([](void* a, void* b, int* d) { hipLaunchKernelGGL(k64, dim3(1), dim3(64), 0, 0, (const v4i*)a, (const v4i*)b, d); hipDeviceSynchronize(); });
  return 0;
}
