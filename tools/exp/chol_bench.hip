// Diagnostic: k_lm_chol alone on a random SPD reduced system of nb block rows, with per-wave cycle stamps of its phases.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I include -DLM_CHOL_STAMPS tools/exp/chol_bench.hip -o /tmp/chol_bench && /tmp/chol_bench 40
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../orb_slam2_ros2_amd/csrc/k_lm.hip"
#include "../../orb_slam2_ros2_amd/csrc/k_lmbig.hip"

int main(int argc, char** argv) {
  using namespace orbfe;
  const int nb = argc > 1 ? atoi(argv[1]) : 40, n = 6 * nb;
  std::vector<double> M((size_t)n * n), S((size_t)n * n, 0.0), rhs(n);
  srand(1);
  for (auto& v : M) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double a = i == j ? (double)n : 0.0;
      for (int k = 0; k < n; ++k) a += M[(size_t)i * n + k] * M[(size_t)j * n + k];
      S[(size_t)i * n + j] = a;
    }
  for (int i = 0; i < n; ++i) rhs[i] = (double)rand() / RAND_MAX;
  std::vector<double> blk((size_t)nb * (nb + 1) / 2 * 36);
  for (int I = 0; I < nb; ++I)
    for (int J = 0; J <= I; ++J)
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 6; ++c) blk[((size_t)I * (I + 1) / 2 + J) * 36 + 6 * a + c] = S[(size_t)(6 * I + a) * n + 6 * J + c];
  double *d_blk, *d_rhs, *d_x;
  LmState* d_st;
  hipMalloc(&d_blk, blk.size() * 8), hipMalloc(&d_rhs, n * 8), hipMalloc(&d_x, n * 8), hipMalloc(&d_st, sizeof(LmState));
  hipMemcpy(d_blk, blk.data(), blk.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_rhs, rhs.data(), n * 8, hipMemcpyHostToDevice);
  LmState st{};
  st.run_step = 1, st.ok = 1;
  hipMemcpy(d_st, &st, sizeof st, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int reps = 50;
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_lm_chol, dim3(1), dim3(LM_CHOL_THREADS), 0, 0, nb, d_st, d_blk, d_rhs, d_x);
  long long zero[64] = {0};
#ifdef LM_CHOL_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(g_lm_stamps), zero, sizeof zero);
#endif
  hipEventRecord(a);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k_lm_chol, dim3(1), dim3(LM_CHOL_THREADS), 0, 0, nb, d_st, d_blk, d_rhs, d_x);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<double> x(n);
  hipMemcpy(x.data(), d_x, n * 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < n; ++i) {
    double r = -rhs[i];
    for (int j = 0; j < n; ++j) r += S[(size_t)i * n + j] * x[j];
    err = std::fmax(err, std::fabs(r));
  }
  printf("nb %d: %.1f us per factorisation + solve, max residual %.2e\n", nb, ms / reps * 1e3, err);
#ifdef LM_CHOL_STAMPS
  {  // launch by launch: duration against the SIMD each wave landed on (HW_ID bits 5:4)
    for (int rep = 0; rep < 24; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k_lm_chol, dim3(1), dim3(LM_CHOL_THREADS), 0, 0, nb, d_st, d_blk, d_rhs, d_x);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float one = 0;
      hipEventElapsedTime(&one, a, b);
      unsigned int hw[8];
      hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_lm_hwid), sizeof hw);
      printf("launch %2d: %6.1f us  simd of waves 0..7:", rep, one * 1e3);
      for (int w = 0; w < 8; ++w) printf(" %u", (hw[w] >> 4) & 3);
      printf("  cu %u se %u\n", (hw[0] >> 8) & 15, (hw[0] >> 13) & 7);
    }
  }
  long long stp[8][8];
  hipMemcpyFromSymbol(stp, HIP_SYMBOL(g_lm_stamps), sizeof stp);
  const char* names[8] = {"load+diag0", "panel work", "panel wait", "update work", "update wait", "diag updates", "back-subst", "-"};
  for (int w = 0; w < 8; ++w) {
    printf("wave %d:", w);
    for (int k = 0; k < 7; ++k) printf("  %s %.1f", names[k], (double)stp[w][k] / reps / 100.0);  // s_memtime ticks at 100 MHz -> us
    printf("\n");
  }
#endif
  return 0;
}
