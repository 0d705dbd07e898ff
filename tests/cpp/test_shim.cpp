// Compile-and-run check of the C++ host mirror (orb_slam2_ros2_amd/host/orbfe_shim.hpp).
// Usage: test_shim <image.raw> <w> <h>   prints "n_left n_right n_matches fnv1a(keypoints) fnv1a(descriptors)" or
//        "NO_DEVICE" (exit 3) when no HIP device is usable -- there is no CPU fallback to run instead.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
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
    orbfe::ORBExtractor el({L.data(), w, h, (size_t)w}, 2000, 8, 1.2f, "", 20, 7);
    orbfe::ORBExtractor er({R.data(), w, h, (size_t)w}, 2000, 8, 1.2f, "", 20, 7);
    std::vector<orbfe_keypoint> kl, kr;
    std::vector<orbfe::Descriptor> dl, dr;
    // Frame::Frame (src/Frame.cc:100-105): the two extractions on two threads, each object in its own slot of the shared context
    {
      std::thread tl([&] { el.extract(kl, dl); }), tr([&] { er.extract(kr, dr); });
      tl.join();
      tr.join();
    }
    std::vector<double> ru, dp;
    const int nm = orbfe::ORBMatcher().searchByStereo(el, er, 718.856f, 718.856f * 0.537166f, ru, dp);
    printf("%zu %zu %d %016llx %016llx %d", kl.size(), kr.size(), nm, (unsigned long long)fnv1a(kl.data(), kl.size() * sizeof(orbfe_keypoint)),
           (unsigned long long)fnv1a(dl.data(), dl.size() * 32), orbfe::ORBMatcher::descDistance(dl[0], dl[1]));
    // guided search: every left keypoint looks for itself in slot 0 (radius 3 px, its own octave): best = itself at distance 0
    {
      const size_t n = std::min<size_t>(kl.size(), 200);
      std::vector<float> uv(2 * n), rad(n, 3.0f);
      std::vector<int8_t> lo(n), hi(n);
      std::vector<orbfe::Descriptor> qd(dl.begin(), dl.begin() + n);
      for (size_t i = 0; i < n; ++i) {
        uv[2 * i] = kl[i].x, uv[2 * i + 1] = kl[i].y;
        lo[i] = hi[i] = (int8_t)kl[i].octave;
      }
      const auto m = orbfe::ORBMatcher().searchInArea(el.context(), el.slot(), uv, rad, lo, hi, qd);
      int self = 0;
      for (size_t i = 0; i < n; ++i) self += (m.bestDist[i] == 0 && m.nCand[i] >= 1);
      printf(" %d/%zu", self, n);
      // ORBMatcher::searchByProjection(frame, frame): with ratio 1 every keypoint whose own cell neighbourhood holds no closer
      // descriptor matches itself at distance 0; verifyAngle keeps them all (angle difference 0 -> one bin)
      std::vector<float> sf(8);
      orbfe::check(el.context(), orbfe_get_scale_factors(el.context(), sf.data(), 8));
      std::vector<uint8_t> valid(kl.size(), 1), none(kl.size(), 0);
      auto mm = orbfe::ORBMatcher(1.0f).searchByProjection(el.context(), el.slot(), sf, kl, dl, valid, none, 3.0f, 0.f, 0.5f, false);
      int selfm = 0;
      for (const auto& d : mm) selfm += (d.queryIdx == d.trainIdx && d.distance == 0);
      std::vector<float> ang(kl.size());
      for (size_t i = 0; i < kl.size(); ++i) ang[i] = kl[i].angle;
      const size_t before = mm.size();
      orbfe::ORBMatcher::verifyAngle(mm, ang, ang);
      printf(" %d/%zu/%zu", selfm, before, mm.size());
    }
    // ORBMatcher::searchBySim3 with the identity similarity between a keyframe and a copy of itself (uploaded feature sets): every
    // map point projects onto its own feature and must match it
    {
      orbfe::ORBMatcher::KeyFrameView kf;
      kf.kps = kl, kf.desc = dl;
      const size_t n = kl.size();
      kf.pos.resize(3 * n), kf.good.assign(n, 1), kf.inMap.assign(n, 1), kf.maxDist.assign(n, 0.f), kf.minDist.assign(n, 0.f);
      const float fx = 718.856f, cx = 607.1928f, cy = 185.2157f;
      std::vector<float> sf(8);
      orbfe::check(el.context(), orbfe_get_scale_factors(el.context(), sf.data(), 8));
      for (size_t i = 0; i < n; ++i) {
        const float z = 10.f;
        kf.pos[3 * i] = (kl[i].x - cx) / fx * z, kf.pos[3 * i + 1] = (kl[i].y - cy) / fx * z, kf.pos[3 * i + 2] = z;
        const float d = std::sqrt(kf.pos[3 * i] * kf.pos[3 * i] + kf.pos[3 * i + 1] * kf.pos[3 * i + 1] + z * z);
        kf.maxDist[i] = d * sf[kl[i].octave] * 1.01f, kf.minDist[i] = 0.5f * d;   // predictLevel(d) == the feature's own octave
      }
      const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
      std::vector<std::pair<int, int>> mm = {{0, 0}};
      const orbfe::ORBMatcher::Intrinsics K = {fx, fx, cx, cy, 0.f, (float)w, 0.f, (float)h};
      orbfe::ORBMatcher(1.0f).searchBySim3(el.context(), kf, kf, mm, orbfe::ORBMatcher::Sim3(), I, zero, I, zero, 7.5f, K, sf);
      size_t self = 0;
      for (const auto& m : mm) self += m.first == m.second;
      printf(" %zu/%zu", self, mm.size());
    }
    // local BA through the Optimizer mirror: 4 keyframes (2 fixed) looking at a 5x4x2 grid of points, exact stereo
    // measurements, perturbed free poses and points -> the optimum is the truth
    {
      const int NK = 4, NP = 40;
      std::vector<double> poses(NK * 7, 0.0), truth, points(NP * 3), pts_true, meas;
      std::vector<int32_t> ek, ep;
      for (int k = 0; k < NK; ++k) poses[7 * k + 3] = 1.0, poses[7 * k + 4] = -0.3 * k;  // identity rotation, camera at x = 0.3 k
      for (int p = 0; p < NP; ++p) {
        points[3 * p] = -1.0 + 0.5 * (p % 5), points[3 * p + 1] = -0.6 + 0.4 * ((p / 5) % 4), points[3 * p + 2] = 4.0 + 1.5 * (p / 20);
      }
      truth = poses, pts_true = points;
      const double fx = 520.9, fy = 521.0, cx = 325.1, cy = 249.7, bf = 40.0;
      for (int p = 0; p < NP; ++p)
        for (int k = 0; k < NK; ++k) {
          const double x = points[3 * p] + poses[7 * k + 4], y = points[3 * p + 1], z = points[3 * p + 2];
          const double u = fx * x / z + cx;
          meas.insert(meas.end(), {u, fy * y / z + cy, u - bf / z});
          ek.push_back(k), ep.push_back(p);
        }
      const int E = (int)ek.size();
      std::vector<uint8_t> st(E, 1), fixed = {1, 1, 0, 0};
      std::vector<double> info(E, 1.0), delta(E, (double)orbfe::Optimizer::deltaStereo);
      poses[7 * 2 + 4] += 0.03, poses[7 * 3 + 5] -= 0.02, poses[7 * 3 + 6] += 0.025;
      for (int p = 0; p < NP; ++p) points[3 * p + (p % 3)] += 0.04 * ((p % 2) ? 1 : -1);
      orbfe_ba_problem prob = {NK, NP, E, poses.data(), points.data(), ek.data(), ep.data(), meas.data(), st.data(), info.data(),
                               delta.data(), fx, fy, cx, cy, bf};
      const auto r = orbfe::Optimizer::OptimizeLocalMap(el.context(), prob, fixed);
      double err = 0;
      for (size_t i = 0; i < truth.size(); ++i) err = std::max(err, std::fabs(r.poses[i] - truth[i]));
      for (size_t i = 0; i < pts_true.size(); ++i) err = std::max(err, std::fabs(r.points[i] - pts_true[i]));
      int nbad = 0;
      for (uint8_t b : r.bad) nbad += b;
      printf(" %.3e %d %d", err, nbad, r.iterations[0] + r.iterations[1]);
    }
    if (argc >= 8) {  // <map.pb in> <keyframe id> <map.pb out>: Optimizer::OptimizeLocalMap on a map file through host/map_pb.hpp
      std::vector<uint8_t> bytes;
      FILE* mf = fopen(argv[5], "rb");
      if (!mf) return 2;
      for (int c; (c = fgetc(mf)) != EOF;) bytes.push_back((uint8_t)c);
      fclose(mf);
      orbfe::mappb::MapRec map;
      if (!orbfe::mappb::parse(bytes.data(), bytes.size(), map)) return 2;
      const orbfe_camera cam = {520.908620f, 521.007327f, 325.141442f, 249.701764f, 0, 0, 0, 0, 0, (float)(520.908620 * 0.0767889)};
      const auto rep = orbfe::Optimizer::OptimizeLocalMap(el.context(), map, (uint64_t)atoll(argv[6]), cam);
      const std::string out = orbfe::mappb::serialize(map);
      mf = fopen(argv[7], "wb");
      if (!mf || fwrite(out.data(), 1, out.size(), mf) != out.size()) return 2;
      fclose(mf);
      printf(" %d/%d/%d/%d/%d", rep.n_group, rep.n_points, rep.n_edges, rep.n_outlier_edges, rep.written);
    }
    printf("\n");
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
