// map.pb -- the reference's on-disk map (proto/Map.proto, proto/Keyframe.proto, proto/MapPoint.proto) without libprotobuf.
//
// The reference writes one `orbslam2.MapData` message with Map::saveToProtobuf (src/Map.cc:200-250) and reads it back with
// Map::loadFromProtobuf (src/Map.cc:252-313).  This header holds
//   * plain records of the three schemas (field numbers cited per member),
//   * a proto3 wire-format reader that accepts anything libprotobuf may emit for them (fields in any order, packed or
//     unpacked repeated scalars, unknown fields skipped) and a writer that emits what libprotobuf's C++ serialiser emits for
//     a message filled the way KeyFrame::serializeToProtobuf / MapPoint::serializeToProtobuf fill it (src/KeyFrame.cc:553-645,
//     src/MapPoint.cc:609-646): fields in number order, zero scalars omitted, repeated scalars packed, the sub-messages the
//     reference touches with mutable_*() always present,
//   * the graph construction of Optimizer::OptimizeLocalMap (src/Optimizer.cc:232-330) over such a map, and its write-back
//     policy (src/Optimizer.cc:363-441), so that a local bundle adjustment can be run on a map file: the solve itself is
//     orbfe_ba_local_optimize (device).
// Everything here is host code (C++17, no dependencies).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace orbfe {
namespace mappb {

// ---- records ------------------------------------------------------------------------------------------------------------
struct KeyPointRec {  // Keyframe.proto:7-12
  float x = 0.f, y = 0.f;
  int32_t octave = 0;
  float angle = 0.f;
};

struct FeatureNodeRec {  // Keyframe.proto:25-30
  uint32_t node_id = 0;
  std::vector<uint32_t> feature_ids;
};

struct KeyFrameRec {  // Keyframe.proto:45-67
  uint64_t id = 0;                                          // 1
  float max_u = 0.f, max_v = 0.f, min_u = 0.f, min_v = 0.f;  // 2..5
  std::vector<KeyPointRec> keypoints;                       // 6
  std::vector<float> right_u, depths;                       // 7, 8
  std::vector<std::array<uint8_t, 32>> descriptors;         // 9  (Descriptor.data, 32 raw bytes)
  std::vector<uint8_t> descriptor_len;                      //    bytes actually present per descriptor (32 in files the reference writes)
  std::map<uint32_t, double> bow;                           // 10 (BowVector.words)
  std::vector<FeatureNodeRec> feature_nodes;                // 11
  std::vector<float> rotation, translation;                 // 12 (Pose: 9 + 3 floats, row-major Rcw, tcw)
  std::vector<std::pair<uint64_t, int32_t>> connected;      // 13 (id, weight)
  std::vector<uint64_t> children, loop_edges;               // 14, 15
  std::vector<int64_t> map_points;                          // 16 (-1: no map point at this keypoint)
};

struct MapPointRec {  // MapPoint.proto:14-32
  uint64_t id = 0;                                  // 1
  float max_distance = 0.f, min_distance = 0.f;     // 2, 3
  uint64_t ref_kf_id = 0, ref_feat_id = 0;          // 4, 5
  int32_t matches_in_track = 0, inliers_in_track = 0;  // 6, 7
  float position[3] = {0.f, 0.f, 0.f};              // 8
  float view_direction[3] = {0.f, 0.f, 0.f};        // 9
  std::array<uint8_t, 32> desc{};                   // 10
  uint8_t desc_len = 0;
};

struct MapRec {  // Map.proto:9-12, Keyframe.proto:69-73, MapPoint.proto:34-36
  uint64_t next_id = 0;              // KeyFrameList.next_id  (KeyFrame::mnNextId)
  std::vector<float> scale_factors;  // KeyFrameList.scale_factors (KeyFrame::mvfScaledFactors)
  std::vector<KeyFrameRec> keyframes;
  std::vector<MapPointRec> mappoints;
};

// ---- wire format: reader ------------------------------------------------------------------------------------------------
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  bool done() const { return !ok || p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      if (p >= end) return ok = false, 0;
      const uint8_t b = *p++;
      if (shift < 64) v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    return ok = false, 0;
  }
  uint32_t fixed32() {
    if (end - p < 4) return ok = false, 0;
    uint32_t v;
    std::memcpy(&v, p, 4), p += 4;
    return v;
  }
  uint64_t fixed64() {
    if (end - p < 8) return ok = false, 0;
    uint64_t v;
    std::memcpy(&v, p, 8), p += 8;
    return v;
  }
  float f32() {
    const uint32_t u = fixed32();
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  }
  double f64() {
    const uint64_t u = fixed64();
    double d;
    std::memcpy(&d, &u, 8);
    return d;
  }
  Reader sub() {  // length-delimited payload
    const uint64_t n = varint();
    if (!ok || n > (uint64_t)(end - p)) {
      ok = false;
      return Reader(p, 0);
    }
    Reader r(p, (size_t)n);
    p += n;
    return r;
  }
  void skip(uint32_t wt) {
    switch (wt) {
      case 0: (void)varint(); break;
      case 1: (void)fixed64(); break;
      case 2: (void)sub(); break;
      case 5: (void)fixed32(); break;
      default: ok = false;  // groups (3/4) do not occur in proto3 files
    }
  }
  // one tag; false at the end of the payload or on error
  bool tag(uint32_t& field, uint32_t& wt) {
    if (done()) return false;
    const uint64_t t = varint();
    field = (uint32_t)(t >> 3), wt = (uint32_t)(t & 7);
    if (ok && field == 0) ok = false;
    return ok;
  }
};

// repeated scalar, packed (wire type 2) or one element per tag
template <class T, class F>
inline void rep_scalar(Reader& r, uint32_t wt, uint32_t elem_wt, std::vector<T>& out, F&& read_one) {
  if (wt == 2) {
    Reader s = r.sub();
    while (!s.done()) out.push_back(read_one(s));
    if (!s.ok) r.ok = false;
  } else if (wt == elem_wt) {
    out.push_back(read_one(r));
  } else
    r.skip(wt);
}

inline void parse_vec3(Reader r, float v[3], bool& ok) {
  uint32_t f, wt;
  while (r.tag(f, wt)) {
    if (f >= 1 && f <= 3 && wt == 5)
      v[f - 1] = r.f32();
    else
      r.skip(wt);
  }
  ok = ok && r.ok;
}

inline void parse_desc(Reader r, uint8_t out[32], uint8_t& len, bool& ok) {
  uint32_t f, wt;
  while (r.tag(f, wt)) {
    if (f == 1 && wt == 2) {
      Reader s = r.sub();
      const size_t n = (size_t)(s.end - s.p);
      std::memset(out, 0, 32);
      std::memcpy(out, s.p, std::min<size_t>(n, 32));  // the reference copies 32 bytes when at least 32 are there (src/KeyFrame.cc:688-689)
      len = (uint8_t)std::min<size_t>(n, 255);
    } else
      r.skip(wt);
  }
  ok = ok && r.ok;
}

inline bool parse_keyframe(Reader r, KeyFrameRec& k) {
  uint32_t f, wt;
  bool ok = true;
  auto rf32 = [](Reader& s) { return s.f32(); };
  auto ru64 = [](Reader& s) { return s.varint(); };
  auto ri64 = [](Reader& s) { return (int64_t)s.varint(); };
  while (r.tag(f, wt)) {
    if (f == 1 && wt == 0)
      k.id = r.varint();
    else if (f >= 2 && f <= 5 && wt == 5)
      (f == 2 ? k.max_u : f == 3 ? k.max_v : f == 4 ? k.min_u : k.min_v) = r.f32();
    else if (f == 6 && wt == 2) {
      Reader s = r.sub();
      KeyPointRec kp;
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1 && w == 5)
          kp.x = s.f32();
        else if (g == 2 && w == 5)
          kp.y = s.f32();
        else if (g == 3 && w == 0)
          kp.octave = (int32_t)s.varint();
        else if (g == 4 && w == 5)
          kp.angle = s.f32();
        else
          s.skip(w);
      }
      ok = ok && s.ok;
      k.keypoints.push_back(kp);
    } else if (f == 7)
      rep_scalar(r, wt, 5, k.right_u, rf32);
    else if (f == 8)
      rep_scalar(r, wt, 5, k.depths, rf32);
    else if (f == 9 && wt == 2) {
      k.descriptors.emplace_back();
      k.descriptor_len.push_back(0);
      k.descriptors.back().fill(0);
      parse_desc(r.sub(), k.descriptors.back().data(), k.descriptor_len.back(), ok);
    } else if (f == 10 && wt == 2) {
      Reader s = r.sub();
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1 && w == 2) {  // map entry {key = 1, value = 2}
          Reader e = s.sub();
          uint32_t key = 0;
          double val = 0.0;
          uint32_t h, x;
          while (e.tag(h, x)) {
            if (h == 1 && x == 0)
              key = (uint32_t)e.varint();
            else if (h == 2 && x == 1)
              val = e.f64();
            else
              e.skip(x);
          }
          ok = ok && e.ok;
          k.bow[key] = val;
        } else
          s.skip(w);
      }
      ok = ok && s.ok;
    } else if (f == 11 && wt == 2) {
      Reader s = r.sub();
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1 && w == 2) {
          Reader e = s.sub();
          FeatureNodeRec n;
          uint32_t h, x;
          while (e.tag(h, x)) {
            if (h == 1 && x == 0)
              n.node_id = (uint32_t)e.varint();
            else if (h == 2)
              rep_scalar(e, x, 0, n.feature_ids, [](Reader& q) { return (uint32_t)q.varint(); });
            else
              e.skip(x);
          }
          ok = ok && e.ok;
          k.feature_nodes.push_back(std::move(n));
        } else
          s.skip(w);
      }
      ok = ok && s.ok;
    } else if (f == 12 && wt == 2) {
      Reader s = r.sub();
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1)
          rep_scalar(s, w, 5, k.rotation, rf32);
        else if (g == 2)
          rep_scalar(s, w, 5, k.translation, rf32);
        else
          s.skip(w);
      }
      ok = ok && s.ok;
    } else if (f == 13 && wt == 2) {
      Reader s = r.sub();
      std::pair<uint64_t, int32_t> c{0, 0};
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1 && w == 0)
          c.first = s.varint();
        else if (g == 2 && w == 0)
          c.second = (int32_t)s.varint();
        else
          s.skip(w);
      }
      ok = ok && s.ok;
      k.connected.push_back(c);
    } else if (f == 14)
      rep_scalar(r, wt, 0, k.children, ru64);
    else if (f == 15)
      rep_scalar(r, wt, 0, k.loop_edges, ru64);
    else if (f == 16)
      rep_scalar(r, wt, 0, k.map_points, ri64);
    else
      r.skip(wt);
  }
  return ok && r.ok;
}

inline bool parse_mappoint(Reader r, MapPointRec& m) {
  uint32_t f, wt;
  bool ok = true;
  while (r.tag(f, wt)) {
    if (f == 1 && wt == 0)
      m.id = r.varint();
    else if (f == 2 && wt == 5)
      m.max_distance = r.f32();
    else if (f == 3 && wt == 5)
      m.min_distance = r.f32();
    else if (f == 4 && wt == 0)
      m.ref_kf_id = r.varint();
    else if (f == 5 && wt == 0)
      m.ref_feat_id = r.varint();
    else if (f == 6 && wt == 0)
      m.matches_in_track = (int32_t)r.varint();
    else if (f == 7 && wt == 0)
      m.inliers_in_track = (int32_t)r.varint();
    else if (f == 8 && wt == 2)
      parse_vec3(r.sub(), m.position, ok);
    else if (f == 9 && wt == 2)
      parse_vec3(r.sub(), m.view_direction, ok);
    else if (f == 10 && wt == 2)
      parse_desc(r.sub(), m.desc.data(), m.desc_len, ok);
    else
      r.skip(wt);
  }
  return ok && r.ok;
}

// MapData (src/Map.cc:263-267: ParseFromIstream); false = malformed input, `map` then holds what was read so far
inline bool parse(const uint8_t* bytes, size_t len, MapRec& map) {
  Reader r(bytes, len);
  uint32_t f, wt;
  bool ok = true;
  while (r.tag(f, wt)) {
    if (f == 1 && wt == 2) {  // KeyFrameList
      Reader s = r.sub();
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1 && w == 0)
          map.next_id = s.varint();
        else if (g == 2)
          rep_scalar(s, w, 5, map.scale_factors, [](Reader& q) { return q.f32(); });
        else if (g == 3 && w == 2) {
          map.keyframes.emplace_back();
          ok = parse_keyframe(s.sub(), map.keyframes.back()) && ok;
        } else
          s.skip(w);
      }
      ok = ok && s.ok;
    } else if (f == 2 && wt == 2) {  // MapPointList
      Reader s = r.sub();
      uint32_t g, w;
      while (s.tag(g, w)) {
        if (g == 1 && w == 2) {
          map.mappoints.emplace_back();
          ok = parse_mappoint(s.sub(), map.mappoints.back()) && ok;
        } else
          s.skip(w);
      }
      ok = ok && s.ok;
    } else
      r.skip(wt);
  }
  return ok && r.ok;
}

// ---- wire format: writer ------------------------------------------------------------------------------------------------
struct Writer {
  std::string b;
  void varint(uint64_t v) {
    while (v >= 0x80) b.push_back((char)(v | 0x80)), v >>= 7;
    b.push_back((char)v);
  }
  void tag(uint32_t field, uint32_t wt) { varint(((uint64_t)field << 3) | wt); }
  void raw(const void* p, size_t n) { b.append((const char*)p, n); }
  static bool nonzero(float f) {  // proto3 omits a float only when its bit pattern is +0.0
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u != 0;
  }
  void f32(uint32_t field, float v) {
    if (!nonzero(v)) return;
    tag(field, 5), raw(&v, 4);
  }
  void u64(uint32_t field, uint64_t v) {
    if (v) tag(field, 0), varint(v);
  }
  void i32(uint32_t field, int32_t v) {  // int32: negative values are sign-extended to ten bytes
    if (v) tag(field, 0), varint((uint64_t)(int64_t)v);
  }
  void bytes(uint32_t field, const void* p, size_t n) {
    if (n) tag(field, 2), varint(n), raw(p, n);
  }
  void msg(uint32_t field, const Writer& w) { tag(field, 2), varint(w.b.size()), b += w.b; }  // present even when empty
  void packed_f32(uint32_t field, const std::vector<float>& v) {
    if (v.empty()) return;
    tag(field, 2), varint(v.size() * 4), raw(v.data(), v.size() * 4);
  }
  template <class T>
  void packed_varint(uint32_t field, const std::vector<T>& v) {
    if (v.empty()) return;
    Writer w;
    for (const T& x : v) w.varint((uint64_t)(int64_t)x);
    tag(field, 2), varint(w.b.size()), b += w.b;
  }
};
inline Writer write_vec3(const float v[3]) {
  Writer w;
  w.f32(1, v[0]), w.f32(2, v[1]), w.f32(3, v[2]);
  return w;
}

inline Writer write_keyframe(const KeyFrameRec& k) {  // what src/KeyFrame.cc:553-645 fills, in field order
  Writer w;
  w.u64(1, k.id);
  w.f32(2, k.max_u), w.f32(3, k.max_v), w.f32(4, k.min_u), w.f32(5, k.min_v);
  for (const KeyPointRec& kp : k.keypoints) {
    Writer s;
    s.f32(1, kp.x), s.f32(2, kp.y), s.i32(3, kp.octave), s.f32(4, kp.angle);
    w.msg(6, s);
  }
  w.packed_f32(7, k.right_u), w.packed_f32(8, k.depths);
  for (size_t i = 0; i < k.descriptors.size(); ++i) {
    Writer s;
    s.bytes(1, k.descriptors[i].data(), i < k.descriptor_len.size() ? std::min<size_t>(k.descriptor_len[i], 32) : 32);
    w.msg(9, s);
  }
  {
    Writer s;  // map<uint32,double>: libprotobuf's order is unspecified; ascending keys here (the order of DBoW3's std::map)
    for (const auto& kv : k.bow) {
      Writer e;
      e.tag(1, 0), e.varint(kv.first);  // both entry fields are always written
      e.tag(2, 1), e.raw(&kv.second, 8);
      s.msg(1, e);
    }
    w.msg(10, s);
  }
  {
    Writer s;
    for (const FeatureNodeRec& n : k.feature_nodes) {
      Writer e;
      e.u64(1, n.node_id), e.packed_varint(2, n.feature_ids);
      s.msg(1, e);
    }
    w.msg(11, s);
  }
  {
    Writer s;
    s.packed_f32(1, k.rotation), s.packed_f32(2, k.translation);
    w.msg(12, s);
  }
  for (const auto& c : k.connected) {
    Writer s;
    s.u64(1, c.first), s.i32(2, c.second);
    w.msg(13, s);
  }
  w.packed_varint(14, k.children), w.packed_varint(15, k.loop_edges), w.packed_varint(16, k.map_points);
  return w;
}

inline Writer write_mappoint(const MapPointRec& m) {  // src/MapPoint.cc:609-646
  Writer w;
  w.u64(1, m.id), w.f32(2, m.max_distance), w.f32(3, m.min_distance), w.u64(4, m.ref_kf_id), w.u64(5, m.ref_feat_id);
  w.i32(6, m.matches_in_track), w.i32(7, m.inliers_in_track);
  w.msg(8, write_vec3(m.position)), w.msg(9, write_vec3(m.view_direction));
  Writer d;
  d.bytes(1, m.desc.data(), std::min<size_t>(m.desc_len, 32));
  w.msg(10, d);
  return w;
}

inline std::string serialize(const MapRec& map) {  // src/Map.cc:200-248
  Writer kfl, mpl, out;
  kfl.u64(1, map.next_id), kfl.packed_f32(2, map.scale_factors);
  for (const KeyFrameRec& k : map.keyframes) kfl.msg(3, write_keyframe(k));
  for (const MapPointRec& m : map.mappoints) mpl.msg(1, write_mappoint(m));
  out.msg(1, kfl), out.msg(2, mpl);
  return std::move(out.b);
}

// ---- Optimizer::OptimizeLocalMap over a map file ----------------------------------------------------------------------------
// Converter::ConvertTcw2SE3 (src/Optimizer.cc:630-645): float matrix -> double, Eigen::Quaterniond(Matrix3d) (Eigen's
// trace / largest-diagonal branches), normalize(); g2o::SE3Quat's constructor then normalises again and makes w >= 0.
inline void tcw_to_se3(const float* R /*row-major 3x3*/, const float* t, double out[7]) {
  double m[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = (double)R[i * 3 + j];
  double q[4];  // x y z w
  double tr = m[0][0] + m[1][1] + m[2][2];
  if (tr > 0.0) {
    tr = std::sqrt(tr + 1.0);
    q[3] = 0.5 * tr;
    tr = 0.5 / tr;
    q[0] = (m[2][1] - m[1][2]) * tr, q[1] = (m[0][2] - m[2][0]) * tr, q[2] = (m[1][0] - m[0][1]) * tr;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    tr = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * tr;
    tr = 0.5 / tr;
    q[3] = (m[k][j] - m[j][k]) * tr;
    q[j] = (m[j][i] + m[i][j]) * tr;
    q[k] = (m[k][i] + m[i][k]) * tr;
  }
  for (int pass = 0; pass < 2; ++pass) {  // Quaterniond::normalize(), then SE3Quat::normalizeRotation()
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0.0)
      for (double& c : q) c /= n;
  }
  if (q[3] < 0.0)
    for (double& c : q) c = -c;
  out[0] = q[0], out[1] = q[1], out[2] = q[2], out[3] = q[3];
  out[4] = (double)t[0], out[5] = (double)t[1], out[6] = (double)t[2];
}

// Converter::ConvertSE32Tcw (src/Optimizer.cc:653-675): Eigen's Quaternion::toRotationMatrix, cast to float
inline void se3_to_tcw(const double p[7], float R[9], float t[3]) {
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
               tzz = tz * z;
  R[0] = (float)(1.0 - (tyy + tzz)), R[1] = (float)(txy - twz), R[2] = (float)(txz + twy);
  R[3] = (float)(txy + twz), R[4] = (float)(1.0 - (txx + tzz)), R[5] = (float)(tyz - twx);
  R[6] = (float)(txz - twy), R[7] = (float)(tyz + twx), R[8] = (float)(1.0 - (txx + tyy));
  t[0] = (float)p[4], t[1] = (float)p[5], t[2] = (float)p[6];
}

struct LocalGraph {
  // vertices: free keyframes first (covisible keyframes in the reference's order, then the keyframe itself), then the fixed observers
  std::vector<uint64_t> pose_kf_id;
  std::vector<int32_t> pose_kf_index;  // index into MapRec::keyframes
  std::vector<uint8_t> pose_fixed;     // setFixed: keyframe 0 inside the free group (src/Optimizer.cc:248), every observer outside it (:288)
  int32_t n_group = 0;                 // size of the free group (its keyframes receive the optimised pose, :422-429)
  std::vector<double> poses;           // [n][7]
  std::vector<uint64_t> point_id;
  std::vector<int32_t> point_index;    // index into MapRec::mappoints
  std::vector<double> points;          // [n][3]
  std::vector<int32_t> edge_pose, edge_point, edge_feat;  // edge_feat: keypoint index of the observation in its keyframe
  std::vector<double> meas, info, huber;
  std::vector<uint8_t> is_stereo;
};

// Map::processConnection (src/Map.cc:318-374) + KeyFrame::getConnectedKfs(0) (src/KeyFrame.cc:15-45) + the graph of
// Optimizer::OptimizeLocalMap (src/Optimizer.cc:232-330).  Map points are visited in ascending id (the reference iterates a
// std::set of shared_ptr, i.e. in address order -- the order only permutes the edge list), observations in ascending keyframe
// id (KeyFrame::weakCompare, src/KeyFrame.cc:207-225).  false: kf_id is not in the map.
inline bool build_local_graph(const MapRec& map, uint64_t kf_id, LocalGraph& g, float delta_mono = std::sqrt(5.991f),
                              float delta_stereo = std::sqrt(7.815f)) {
  std::unordered_map<uint64_t, int32_t> kf_index, mp_index;
  for (size_t i = 0; i < map.keyframes.size(); ++i) kf_index[map.keyframes[i].id] = (int32_t)i;  // later duplicates win, like mKeyFramesInfo[id] = ...
  for (size_t i = 0; i < map.mappoints.size(); ++i) mp_index[map.mappoints[i].id] = (int32_t)i;
  auto kit = kf_index.find(kf_id);
  if (kit == kf_index.end()) return false;
  const KeyFrameRec& cur = map.keyframes[kit->second];

  // mlpConnectedKfs: weights > 15, descending, equal weights in the order the (id-keyed) map yields them (src/Map.cc:332-345)
  std::map<uint64_t, int32_t> all_connected;
  for (const auto& c : cur.connected) all_connected.insert({c.first, c.second});  // std::map::insert keeps the first
  std::multimap<int32_t, uint64_t, std::greater<int32_t>> ordered;
  for (const auto& c : all_connected) ordered.insert({c.second, c.first});
  std::vector<int32_t> group;
  for (const auto& o : ordered)
    if (o.first > 15) {
      auto it = kf_index.find(o.second);
      if (it != kf_index.end()) group.push_back(it->second);
    }
  group.push_back(kit->second);

  auto add_pose = [&](int32_t ki, bool fixed) {
    const KeyFrameRec& k = map.keyframes[ki];
    static const float eyeR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
    double p[7];
    tcw_to_se3(k.rotation.size() >= 9 ? k.rotation.data() : eyeR, k.translation.size() >= 3 ? k.translation.data() : zero, p);
    g.pose_kf_id.push_back(k.id), g.pose_kf_index.push_back(ki), g.pose_fixed.push_back(fixed ? 1 : 0);
    g.poses.insert(g.poses.end(), p, p + 7);
    return (int32_t)g.pose_kf_id.size() - 1;
  };
  std::unordered_map<int32_t, int32_t> vertex_of;  // keyframe index -> vertex
  std::set<uint64_t> group_points;
  for (int32_t ki : group) {
    if (vertex_of.count(ki)) continue;
    vertex_of[ki] = add_pose(ki, map.keyframes[ki].id == 0);
    for (int64_t mp : map.keyframes[ki].map_points)
      if (mp >= 0 && mp_index.count((uint64_t)mp)) group_points.insert((uint64_t)mp);
  }
  g.n_group = (int32_t)g.pose_kf_id.size();

  // observations: mObs is rebuilt from the keyframes' map_points lists (src/Map.cc:357-369); a later keypoint of the same
  // keyframe does not replace an earlier one (std::map::insert)
  std::unordered_map<uint64_t, std::map<uint64_t, std::pair<int32_t, int32_t>>> obs;  // mp id -> kf id -> (kf index, feature)
  for (const auto& kv : kf_index) {
    const KeyFrameRec& k = map.keyframes[kv.second];
    for (size_t idx = 0; idx < k.map_points.size(); ++idx) {
      const int64_t mp = k.map_points[idx];
      if (mp >= 0 && group_points.count((uint64_t)mp)) obs[(uint64_t)mp].insert({k.id, {kv.second, (int32_t)idx}});
    }
  }

  auto sf = [&](int32_t oct) { return (oct >= 0 && (size_t)oct < map.scale_factors.size()) ? map.scale_factors[oct] : 1.0f; };
  for (uint64_t mp : group_points) {
    const int32_t mi = mp_index[mp];
    const MapPointRec& m = map.mappoints[mi];
    const int32_t pv = (int32_t)g.point_id.size();
    g.point_id.push_back(mp), g.point_index.push_back(mi);
    for (int a = 0; a < 3; ++a) g.points.push_back((double)m.position[a]);  // Converter::ConvertPw2Vector3 (:683-689)
    for (const auto& o : obs[mp]) {
      const int32_t ki = o.second.first, feat = o.second.second;
      const KeyFrameRec& k = map.keyframes[ki];
      if ((size_t)feat >= k.keypoints.size()) continue;
      auto vit = vertex_of.find(ki);
      const int32_t v = vit != vertex_of.end() ? vit->second : (vertex_of[ki] = add_pose(ki, true));
      const KeyPointRec& kp = k.keypoints[feat];
      const double right_u = (size_t)feat < k.right_u.size() ? (double)k.right_u[feat] : -1.0;  // getRightU returns the stored value as double
      const float inv = 1.0f / sf(kp.octave);  // Frame.h:208-213
      g.edge_pose.push_back(v), g.edge_point.push_back(pv), g.edge_feat.push_back(feat);
      g.meas.push_back((double)kp.x), g.meas.push_back((double)kp.y);
      if (right_u > 0) {  // :296-312  Identity * getScaledFactorInv2 = (float)pow(inv, 2)
        g.meas.push_back(right_u), g.is_stereo.push_back(1);
        g.info.push_back((double)(float)((double)inv * (double)inv)), g.huber.push_back((double)delta_stereo);
      } else {            // :314-329  Identity * getScaledFactorInv (not squared)
        g.meas.push_back(0.0), g.is_stereo.push_back(0);
        g.info.push_back((double)inv), g.huber.push_back((double)delta_mono);
      }
    }
  }
  return true;
}

struct LocalBaReport {
  int32_t n_poses = 0, n_group = 0, n_points = 0, n_edges = 0;
  int32_t n_outlier_edges = 0;   // edges failing the final chi2 / depth test (:363-388)
  int32_t n_keyframes_hit = 0;   // keyframes owning at least one such edge (vToProcess.size())
  int32_t n_bad_keyframes = 0;   // ... with more than 30 % of their map points affected (:391-402)
  int32_t written = 0;           // bSetAndErase (:403-404)
};

// The tail of Optimizer::OptimizeLocalMap (src/Optimizer.cc:363-441) given the solver's result: `bad[e]` = final-state
// chi2 > 7.815 (stereo) / 5.991 (mono) or non-positive depth.  Erases the outlier observations and stores poses / points as
// float, unless more than 20 % of the affected keyframes would lose more than 30 % of their map points.  The map points'
// updateDescriptor / updateNormalAndDepth and KeyFrame::updateConnections (:436-440) are map bookkeeping and stay with the caller.
inline LocalBaReport apply_local_ba(MapRec& map, const LocalGraph& g, const double* poses, const double* points, const uint8_t* bad) {
  LocalBaReport r;
  r.n_poses = (int32_t)g.pose_kf_id.size(), r.n_group = g.n_group, r.n_points = (int32_t)g.point_id.size(),
  r.n_edges = (int32_t)g.edge_pose.size();
  std::map<int32_t, std::vector<std::pair<int32_t, int32_t>>> to_process;  // kf index -> (map point index, feature)
  for (size_t e = 0; e < g.edge_pose.size(); ++e)
    if (bad[e]) {
      ++r.n_outlier_edges;
      to_process[g.pose_kf_index[g.edge_pose[e]]].push_back({g.point_index[g.edge_point[e]], g.edge_feat[e]});
    }
  r.n_keyframes_hit = (int32_t)to_process.size();
  for (const auto& item : to_process) {
    int n_good = 0;
    for (int64_t mp : map.keyframes[item.first].map_points) n_good += mp >= 0;
    // `item.second.size() / (float)nGoodMp > 0.3` (:400): a FLOAT quotient compared with the DOUBLE literal 0.3 -- at exactly 30 % (3 of 10)
    // the float 0.3f, widened, exceeds 0.3 and the keyframe counts as bad
    if ((double)((float)item.second.size() / (float)n_good) > 0.3) ++r.n_bad_keyframes;
  }
  r.written = !((double)r.n_bad_keyframes / ((double)to_process.size() + 1e-5) > 0.2);
  if (!r.written) return r;
  for (const auto& item : to_process)
    for (const auto& era : item.second) map.keyframes[item.first].map_points[era.second] = -1;  // setMapPoint(idx, nullptr) + eraseObservetion
  for (int32_t v = 0; v < g.n_group; ++v) {
    KeyFrameRec& k = map.keyframes[g.pose_kf_index[v]];
    k.rotation.resize(9), k.translation.resize(3);
    se3_to_tcw(poses + (size_t)v * 7, k.rotation.data(), k.translation.data());
  }
  for (size_t p = 0; p < g.point_id.size(); ++p)
    for (int a = 0; a < 3; ++a) map.mappoints[g.point_index[p]].position[a] = (float)points[p * 3 + a];  // ConvertVector32Pw (:697-703)
  return r;
}

}  // namespace mappb
}  // namespace orbfe
