// Standalone check + timing of the blocked reduced-system solver (k_lmbig.hip): random SPD system, L L^T x = b against a CPU Cholesky,
// and the diagonal-tile factorisation / panel solve timed alone.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include
// tools/exp/lmb_bench.hip -o tools/exp/bin/lmb_bench ; run on the GPU box: tools/exp/bin/lmb_bench [nf]
#include "../../orb_slam2_ros2_amd/csrc/k_lmbig.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace orbfe;

__global__ __launch_bounds__(192) void k_time_potrf(const double* __restrict__ A, double* __restrict__ out, long long* cycles, int reps, int what) {
  __shared__ LmbShared sh;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  long long tot = 0;
  for (int it = 0; it < reps; ++it) {
    for (int e = t; e < LMB_T * LMB_T; e += 192) {
      const int r = e / LMB_T, c = e - r * LMB_T;
      sh.C[r * LMB_LS + c] = A[r * LMB_T + c];
      sh.Li[r * LMB_LS + c] = c <= r ? A[r * LMB_T + c] * 0.01 + (r == c ? 1.0 : 0.0) : 0.0;
      if (r < 16 && c < 16) sh.Inv[0][r * LMB_IS + c] = sh.Inv[1][r * LMB_IS + c] = sh.Inv[2][r * LMB_IS + c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (what == 0) lmb_potrf_tile(sh, wv, lane);
    else lmb_trsm_strip(sh.C + 16 * wv * LMB_LS, sh.Li, sh, lane);
    __syncthreads();
    tot += __builtin_amdgcn_s_memtime() - t0;
  }
  for (int e = t; e < LMB_T * LMB_T; e += 192) out[e] = sh.C[(e / LMB_T) * LMB_LS + e % LMB_T];
  if (t == 0) *cycles = tot / reps;
}

int main(int argc, char** argv) {
  const int nf = argc > 1 ? atoi(argv[1]) : 64;
  const int n = 6 * nf, ld = lm_big_ld(nf), KT = ld / LMB_T;
  std::vector<double> G((size_t)n * n), S((size_t)n * n, 0.0), b(n), M((size_t)(ld + LMB_T) * ld, 0.0);
  srand(7);
  for (auto& v : G) v = (rand() % 2001 - 1000) / 1000.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += G[(size_t)i * n + k] * G[(size_t)j * n + k];
      S[(size_t)i * n + j] = s + (i == j ? 1.0 : 0.0);
    }
  for (int i = 0; i < n; ++i) b[i] = (rand() % 2001 - 1000) / 1000.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) M[(size_t)i * ld + j] = S[(size_t)i * n + j];
  for (int i = n; i < ld; ++i) M[(size_t)i * ld + i] = 1.0;
  for (int i = 0; i < n; ++i) M[(size_t)ld * ld + i] = b[i];
  // CPU reference
  std::vector<double> Lc = S, x(n);
  for (int j = 0; j < n; ++j) {
    double d = Lc[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= Lc[(size_t)j * n + k] * Lc[(size_t)j * n + k];
    d = std::sqrt(d);
    Lc[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = Lc[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= Lc[(size_t)i * n + k] * Lc[(size_t)j * n + k];
      Lc[(size_t)i * n + j] = s / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= Lc[(size_t)i * n + k] * y[k];
    y[i] = s / Lc[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= Lc[(size_t)k * n + i] * x[k];
    x[i] = s / Lc[(size_t)i * n + i];
  }
  double *dM, *dM0, *dx;
  LmState* dst;
  int32_t* dfl;
  hipMalloc(&dM, M.size() * 8), hipMalloc(&dM0, M.size() * 8), hipMalloc(&dx, (n + 48) * 8), hipMalloc(&dst, sizeof(LmState)), hipMalloc(&dfl, (2 * KT + 4) * 4);
  hipMemcpy(dM0, M.data(), M.size() * 8, hipMemcpyHostToDevice);
  LmState st{};
  st.run_step = 1, st.ok = 1;
  hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice);
  hipMemset(dfl, 0, (2 * KT + 4) * 4);
  LmLaunch L{};
  double* dinv;
  hipMalloc(&dinv, lm_big_inv_bytes(nf));
  L.nf = nf, L.M = dM, L.ld = ld, L.state = dst, L.lmb_flags = dfl, L.x = dx, L.lmb_inv = dinv;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpy(dM, dM0, M.size() * 8, hipMemcpyDeviceToDevice);
    hipEventRecord(e0, 0);
    launch_lm_chol_big(0, L);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  {  // the multi-workgroup back substitution alone
    float bb = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(dfl + KT + 1, 0, KT * 4);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_lmb_back_mw, dim3(KT), dim3(192), 0, 0, n, KT, ld, dM, dinv, dst, dfl, dfl + KT + 1, dx);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      bb = std::min(bb, ms);
    }
    printf("  back substitution, one workgroup per tile column %.3f ms\n", bb);
  }
  {  // the back substitution alone (L and y are in place; it only reads them)
    float bb = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_lmb_back, dim3(1), dim3(1024), (size_t)(ld + LMB_T + LMB_T * LMB_LS + 3 * 16 * LMB_IS) * sizeof(double), 0, n, KT, ld, dM, dinv, dst, dfl, dx);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      bb = std::min(bb, ms);
    }
    printf("  back substitution alone %.3f ms\n", bb);
  }
  std::vector<double> xg(n);
  hipMemcpy(xg.data(), dx, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost);
  double err = 0, nx = 0;
  for (int i = 0; i < n; ++i) err = std::max(err, std::fabs(xg[i] - x[i])), nx = std::max(nx, std::fabs(x[i]));
  printf("nf %d n %d KT %d: factor + solve %.3f ms, ok %d, max |x - x_cpu| %.3e (max |x| %.3e)\n", nf, n, KT, best, st.ok, err, nx);
  // tile kernels alone
  std::vector<double> T(LMB_T * LMB_T);
  for (int i = 0; i < LMB_T; ++i)
    for (int j = 0; j < LMB_T; ++j) T[i * LMB_T + j] = S[(size_t)std::max(i, j) * n + std::min(i, j)];
  double *dA, *dO;
  long long* dcy;
  hipMalloc(&dA, T.size() * 8), hipMalloc(&dO, T.size() * 8), hipMalloc(&dcy, 8);
  hipMemcpy(dA, T.data(), T.size() * 8, hipMemcpyHostToDevice);
  for (int what = 0; what < 2; ++what) {
    hipLaunchKernelGGL(k_time_potrf, dim3(1), dim3(192), 0, 0, dA, dO, dcy, 20, what);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_time_potrf, dim3(1), dim3(192), 0, 0, dA, dO, dcy, 200, what);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long cy;
    hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost);
    printf("%s of one 48 x 48 tile: %lld memtime ticks, %.2f us per call wall (incl. tile reload)\n", what ? "panel solve" : "potrf", cy, ms * 1e3 / 200);
  }
  return 0;
}
