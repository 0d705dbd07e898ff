// Compile-and-run check of the C++ host mirror (orb_slam2_ros2_amd/host/orbfe_shim.hpp).
// Usage: test_shim <image.raw> <w> <h>   prints "n_left n_right n_matches fnv1a(keypoints) fnv1a(descriptors)" or
//        "NO_DEVICE" (exit 3) when no HIP device is usable -- there is no CPU fallback to run instead.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../orb_slam2_ros2_amd/host/orbfe_shim.hpp"

static uint64_t fnv1a(const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= b[i];
    h *= 1099511628211ull;
  }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int w = atoi(argv[3]), h = atoi(argv[4]);
  std::vector<uint8_t> L((size_t)w * h), R((size_t)w * h);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(L.data(), 1, L.size(), f) != L.size()) return 2;
  fclose(f);
  f = fopen(argv[2], "rb");
  if (!f || fread(R.data(), 1, R.size(), f) != R.size()) return 2;
  fclose(f);
  try {
    orbfe::ORBExtractor el({L.data(), w, h, (size_t)w}, 2000, 8, 1.2f, "", 20, 7, 0);
    orbfe::ORBExtractor er({R.data(), w, h, (size_t)w}, 2000, 8, 1.2f, "", 20, 7, 1);
    std::vector<orbfe_keypoint> kl, kr;
    std::vector<orbfe::Descriptor> dl, dr;
    // slot 1 first: extracting slot 1 also refills slot 0 with the right image, the left extract then restores slot 0
    er.extract(kr, dr);
    el.extract(kl, dl);
    // put the right image back into slot 1 for the stereo match (slot 0 keeps the left features)
    const uint8_t* imgs[2] = {L.data(), R.data()};
    std::vector<orbfe_keypoint> kk(4000);
    std::vector<uint8_t> dd(4000 * 32);
    int32_t cnt[2];
    orbfe::check(el.context(), orbfe_extract_batch(el.context(), 2, imgs, w, kk.data(), dd.data(), cnt));
    std::vector<double> ru, dp;
    const int nm = orbfe::ORBMatcher().searchByStereo(el.context(), 2000, (int)kl.size(), 718.856f, 718.856f * 0.537166f, ru, dp);
    printf("%zu %zu %d %016llx %016llx %d\n", kl.size(), kr.size(), nm, (unsigned long long)fnv1a(kl.data(), kl.size() * sizeof(orbfe_keypoint)),
           (unsigned long long)fnv1a(dl.data(), dl.size() * 32), orbfe::ORBMatcher::descDistance(dl[0], dl[1]));
  } catch (const std::exception& e) {
    if (std::string(e.what()).find("no HIP device") != std::string::npos) {
      printf("NO_DEVICE\n");
      return 3;
    }
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
