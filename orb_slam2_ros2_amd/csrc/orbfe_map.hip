// map.pb entry points of the C-ABI (include/orbfe.h): host code only -- the wire format and the graph construction live in
// host/map_pb.hpp, the solve is orbfe_ba_local_optimize (k_lba.hip).
#include <cstring>
#include <string>
#include <vector>

#include "../host/map_pb.hpp"
#include "../host/map_txt.hpp"
#include "orbfe.h"

using namespace orbfe::mappb;

static orbfe_status emit(const std::string& bytes, uint8_t* out, size_t cap, size_t* out_len) {
  if (!out_len) return ORBFE_EBADARG;
  *out_len = bytes.size();
  if (!out) return ORBFE_OK;
  if (cap < bytes.size()) return ORBFE_ECAPACITY;
  std::memcpy(out, bytes.data(), bytes.size());
  return ORBFE_OK;
}

extern "C" {

orbfe_status orbfe_map_pb_summary(const uint8_t* pb, size_t len, orbfe_map_summary* out) {
  if ((!pb && len) || !out) return ORBFE_EBADARG;
  MapRec map;
  if (!parse(pb, len, map)) return ORBFE_EBADARG;
  out->next_id = map.next_id;
  out->n_scale_factors = (int32_t)map.scale_factors.size();
  out->n_keyframes = (int32_t)map.keyframes.size(), out->n_mappoints = (int32_t)map.mappoints.size();
  out->n_keypoints = 0, out->n_observations = 0;
  for (const KeyFrameRec& k : map.keyframes) {
    out->n_keypoints += (int64_t)k.keypoints.size();
    for (int64_t mp : k.map_points) out->n_observations += mp >= 0;
  }
  return ORBFE_OK;
}

orbfe_status orbfe_map_pb_reencode(const uint8_t* pb, size_t len, uint8_t* out, size_t cap, size_t* out_len) {
  if (!pb && len) return ORBFE_EBADARG;
  MapRec map;
  if (!parse(pb, len, map)) return ORBFE_EBADARG;
  return emit(serialize(map), out, cap, out_len);
}

// Map::saveToTxtFile / loadFromTxtFile (src/Map.cc:82-165) as conversions to and from map.pb: host only
orbfe_status orbfe_map_pb_to_txt(const uint8_t* pb, size_t len, char* kf_out, size_t kf_cap, size_t* kf_len, char* mp_out, size_t mp_cap,
                                 size_t* mp_len) {
  if ((!pb && len) || !kf_len || !mp_len) return ORBFE_EBADARG;
  MapRec map;
  if (!parse(pb, len, map)) return ORBFE_EBADARG;
  std::string kf, mp;
  serialize_txt(map, kf, mp);
  const orbfe_status a = emit(kf, (uint8_t*)kf_out, kf_cap, kf_len), b = emit(mp, (uint8_t*)mp_out, mp_cap, mp_len);
  return a != ORBFE_OK ? a : b;
}

orbfe_status orbfe_map_txt_to_pb(const char* kf_txt, size_t kf_len, const char* mp_txt, size_t mp_len, uint8_t* out, size_t cap,
                                 size_t* out_len) {
  if ((!kf_txt && kf_len) || (!mp_txt && mp_len)) return ORBFE_EBADARG;
  MapRec map;
  if (!parse_txt(std::string(kf_txt ? kf_txt : "", kf_len), std::string(mp_txt ? mp_txt : "", mp_len), map)) return ORBFE_EBADARG;
  return emit(serialize(map), out, cap, out_len);
}

orbfe_status orbfe_map_local_graph(const uint8_t* pb, size_t len, uint64_t kf_id, int32_t sizes[4], const orbfe_map_graph* out) {
  if ((!pb && len) || !sizes) return ORBFE_EBADARG;
  MapRec map;
  if (!parse(pb, len, map)) return ORBFE_EBADARG;
  LocalGraph g;
  if (!build_local_graph(map, kf_id, g)) return ORBFE_EBADARG;
  sizes[0] = (int32_t)g.pose_kf_id.size(), sizes[1] = g.n_group, sizes[2] = (int32_t)g.point_id.size(), sizes[3] = (int32_t)g.edge_pose.size();
  if (!out) return ORBFE_OK;
  auto put = [](auto* dst, const auto& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  put(out->pose_kf_id, g.pose_kf_id), put(out->pose_fixed, g.pose_fixed), put(out->poses, g.poses);
  put(out->point_id, g.point_id), put(out->points, g.points);
  put(out->edge_pose, g.edge_pose), put(out->edge_point, g.edge_point), put(out->edge_feat, g.edge_feat);
  put(out->meas, g.meas), put(out->is_stereo, g.is_stereo), put(out->info, g.info), put(out->huber_delta, g.huber);
  return ORBFE_OK;
}

orbfe_status orbfe_map_local_ba(orbfe_ctx* ctx, const uint8_t* pb, size_t len, uint64_t kf_id, const orbfe_camera* cam,
                                const volatile uint8_t* stop_flag, uint8_t* out, size_t cap, size_t* out_len,
                                orbfe_map_ba_report* report) {
  if (!ctx || (!pb && len) || !cam || !out_len) return ORBFE_EBADARG;
  MapRec map;
  if (!parse(pb, len, map)) return ORBFE_EBADARG;
  LocalGraph g;
  if (!build_local_graph(map, kf_id, g)) return ORBFE_EBADARG;
  const int32_t np = (int32_t)g.pose_kf_id.size(), npt = (int32_t)g.point_id.size(), ne = (int32_t)g.edge_pose.size();
  orbfe_map_ba_report rep{};
  rep.n_poses = np, rep.n_group = g.n_group, rep.n_points = npt, rep.n_edges = ne;
  if (ne == 0 || npt == 0) {  // nothing to optimise: g2o's optimize() returns at once on an empty graph; the map is unchanged
    if (report) *report = rep;
    return emit(serialize(map), out, cap, out_len);
  }
  orbfe_ba_problem prob{};
  prob.n_poses = np, prob.n_points = npt, prob.n_edges = ne;
  prob.poses = g.poses.data(), prob.points = g.points.data(), prob.edge_pose = g.edge_pose.data(), prob.edge_point = g.edge_point.data();
  prob.meas = g.meas.data(), prob.is_stereo = g.is_stereo.data(), prob.info = g.info.data(), prob.huber_delta = g.huber.data();
  prob.fx = (double)cam->fx, prob.fy = (double)cam->fy, prob.cx = (double)cam->cx, prob.cy = (double)cam->cy, prob.bf = (double)cam->bf;

  std::vector<double> chi2_0((size_t)ne), err((size_t)ne * 3), rho((size_t)ne * 2);
  orbfe_ba_edge_out eo{};
  eo.error = err.data(), eo.chi2 = chi2_0.data(), eo.rho = rho.data();
  orbfe_status st = orbfe_ba_eval_edges(ctx, &prob, &eo);
  if (st != ORBFE_OK) return st;
  for (double c : chi2_0) rep.chi2_before += c;

  std::vector<double> poses((size_t)np * 7), points((size_t)npt * 3), chi2((size_t)ne);
  std::vector<uint8_t> level((size_t)ne), bad((size_t)ne);
  orbfe_ba_optimize_out oo = {poses.data(), points.data(), level.data(), chi2.data(), bad.data(), rep.iterations};
  st = orbfe_ba_local_optimize(ctx, &prob, g.pose_fixed.data(), 5, 10, stop_flag, &oo);  // src/Optimizer.cc:336, 361
  if (st != ORBFE_OK) return st;
  for (double c : chi2) rep.chi2_after += c;

  const LocalBaReport r = apply_local_ba(map, g, poses.data(), points.data(), bad.data());
  rep.n_outlier_edges = r.n_outlier_edges, rep.n_keyframes_hit = r.n_keyframes_hit, rep.n_bad_keyframes = r.n_bad_keyframes,
  rep.written = r.written;
  if (report) *report = rep;
  return emit(serialize(map), out, cap, out_len);
}

}  // extern "C"
