// Diagnostic: k_pose_only_reg alone on a problem dumped by tools/exp/pose_dump.py, with cycle stamps of the state machine's sections.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I include -DPOSE_STAMPS tools/exp/pose_bench.hip -o tools/exp/bin/pose_bench && tools/exp/bin/pose_bench /tmp/pose.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../orb_slam2_ros2_amd/csrc/k_pose.hip"

int main(int argc, char** argv) {
  using namespace orbfe;
  FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/pose.bin", "rb");
  if (!f) return 1;
  int n = 0;
  double cam[5], pose[7];
  fread(&n, 4, 1, f);
  const int n_file = n;
  fread(cam, 8, 5, f);
  fread(pose, 8, 7, f);
  std::vector<double> xw(3 * n), meas(3 * n), info(n);
  std::vector<float> s2(n);
  fread(xw.data(), 8, 3 * n, f), fread(meas.data(), 8, 3 * n, f), fread(info.data(), 8, n, f), fread(s2.data(), 4, n, f);
  fclose(f);
  if (argc > 2) n = std::min(n, atoi(argv[2]));  // the first n edges only
  (void)n_file;
  double *d_x, *d_m, *d_i, *d_p, *d_po, *d_e;
  float* d_s;
  uint8_t *d_l, *d_r, *d_in;
  int32_t* d_ng;
  hipMalloc(&d_x, 24 * n), hipMalloc(&d_m, 24 * n), hipMalloc(&d_i, 8 * n), hipMalloc(&d_s, 4 * n), hipMalloc(&d_p, 56), hipMalloc(&d_po, 56);
  hipMalloc(&d_e, 24 * n), hipMalloc(&d_l, n), hipMalloc(&d_r, n), hipMalloc(&d_in, n), hipMalloc(&d_ng, 8);
  hipMemcpy(d_x, xw.data(), 24 * n, hipMemcpyHostToDevice), hipMemcpy(d_m, meas.data(), 24 * n, hipMemcpyHostToDevice);
  hipMemcpy(d_i, info.data(), 8 * n, hipMemcpyHostToDevice), hipMemcpy(d_s, s2.data(), 4 * n, hipMemcpyHostToDevice);
  hipMemcpy(d_p, pose, 56, hipMemcpyHostToDevice);
  BaParamsDev prm = {cam[0], cam[1], cam[2], cam[3], cam[4]};
  auto run = [&]() {
    launch_pose_only(0, n, d_x, d_m, d_i, d_s, d_p, prm, (double)(float)std::sqrt(5.991), (double)(float)std::sqrt(7.815), d_e, d_l, d_r, d_in, d_po,
                     d_ng, nullptr);
  };
  for (int w = 0; w < 3; ++w) run();
  hipDeviceSynchronize();
#ifdef POSE_STAMPS
  long long zero[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_pose_stamps), zero, sizeof zero);
#endif
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int reps = 20;
  hipEventRecord(a);
  for (int w = 0; w < reps; ++w) run();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  int ng = 0;
  double po[7];
  hipMemcpy(&ng, d_ng, 4, hipMemcpyDeviceToHost), hipMemcpy(po, d_po, 56, hipMemcpyDeviceToHost);
  printf("n %d: %.1f us per optimisation, n_good %d, t = (%.6f %.6f %.6f)\n", n, ms / reps * 1e3, ng, po[4], po[5], po[6]);
#ifdef POSE_STAMPS
  long long st[8];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pose_stamps), sizeof st);
  const char* names[6] = {"loop top", "solve + oplus", "evaluation", "decision", "build: sums", "build: edges"};
  for (int k = 0; k < 6; ++k) printf("  %-14s %8.1f k cycles per call\n", names[k], (double)st[k] / reps / 1e3);
  printf("  passes %.1f, builds %.1f per call\n", (double)st[6] / reps, (double)st[7] / reps);
#endif
  return 0;
}
