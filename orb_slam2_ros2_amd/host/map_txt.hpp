// map_txt.hpp -- the reference's TEXT map format: `KeyFrames.txt` + `MapPoints.txt` as Map::saveToTxtFile / loadFromTxtFile write
// and read them (src/ORB_SLAM2/src/Map.cc:82-165), on the same records as map_pb.hpp (MapRec).  Header only, iostreams like the
// reference, so numbers are formatted and parsed by the very same library calls (`os << float` = 6 significant digits).
//
//   KeyFrames.txt   first line: next_id scale_0 scale_1 ...  (written once, KeyFrame.cc:458-467; read by the first keyframe, :236-246)
//                   then 10 lines per keyframe (operator<<, KeyFrame.cc:400-530 / readFromStream, :231-391):
//                     id maxU maxV minU minV | x y octave angle rightU depth per keypoint | 32 ints per descriptor | word value per BoW
//                     entry | node n id_1..id_n per feature-vector node | Rcw (9) tcw (3) | id weight per connected keyframe (ascending id) |
//                     children ids | loop-edge ids | map point id per keypoint (-1: none)
//   MapPoints.txt   3 lines per map point (MapPoint.cc:538-596): id maxDist minDist refKF refFeat matchesInTrack inliersInTrack |
//                   position (3) viewDirection (3) | 32 descriptor ints
// Every value is followed by one blank, every line ends with std::endl, as the reference writes them.
#pragma once
#include <sstream>
#include <string>

#include "map_pb.hpp"

namespace orbfe {
namespace mappb {

inline void serialize_txt(const MapRec& map, std::string& keyframes_txt, std::string& mappoints_txt) {
  std::ostringstream os;
  os << map.next_id << " ";
  for (float s : map.scale_factors) os << s << " ";
  os << std::endl;
  for (const KeyFrameRec& k : map.keyframes) {
    os << k.id << " " << k.max_u << " " << k.max_v << " " << k.min_u << " " << k.min_v << std::endl;
    for (size_t i = 0; i < k.keypoints.size(); ++i) {
      const KeyPointRec& kp = k.keypoints[i];
      // mvFeatsRightU / mvDepths are doubles in the reference; a float widened to double prints the same 6 digits
      os << kp.x << " " << kp.y << " " << kp.octave << " " << kp.angle << " " << (double)(i < k.right_u.size() ? k.right_u[i] : -1.f) << " "
         << (double)(i < k.depths.size() ? k.depths[i] : -1.f) << " ";
    }
    os << std::endl;
    for (size_t i = 0; i < k.keypoints.size(); ++i)
      for (int j = 0; j < 32; ++j) os << (i < k.descriptors.size() ? (int)k.descriptors[i][j] : 0) << " ";
    os << std::endl;
    for (const auto& w : k.bow) os << w.first << " " << w.second << " ";
    os << std::endl;
    for (const FeatureNodeRec& n : k.feature_nodes) {
      os << n.node_id << " " << n.feature_ids.size() << " ";
      for (uint32_t id : n.feature_ids) os << id << " ";
    }
    os << std::endl;
    for (int i = 0; i < 9; ++i) os << (k.rotation.size() >= 9 ? k.rotation[i] : (i % 4 == 0 ? 1.f : 0.f)) << " ";
    for (int i = 0; i < 3; ++i) os << (k.translation.size() >= 3 ? k.translation[i] : 0.f) << " ";
    os << std::endl;
    std::map<uint64_t, int32_t> connected;  // the reference copies mmConnectedKfs into a std::map<id, weight> first (:410-416)
    for (const auto& c : k.connected) connected.insert({c.first, c.second});
    for (const auto& c : connected) os << c.first << " " << c.second << " ";
    os << std::endl;
    for (uint64_t c : k.children) os << c << " ";
    os << std::endl;
    for (uint64_t c : k.loop_edges) os << c << " ";
    os << std::endl;
    for (int64_t m : k.map_points) os << m << " ";
    os << std::endl;
  }
  keyframes_txt = os.str();
  std::ostringstream om;
  for (const MapPointRec& m : map.mappoints) {
    om << m.id << " " << m.max_distance << " " << m.min_distance << " " << m.ref_kf_id << " " << m.ref_feat_id << " " << m.matches_in_track << " "
       << m.inliers_in_track << std::endl;
    om << m.position[0] << " " << m.position[1] << " " << m.position[2] << " ";
    om << m.view_direction[0] << " " << m.view_direction[1] << " " << m.view_direction[2] << std::endl;
    for (int i = 0; i < 32; ++i) om << (int)m.desc[i] << " ";
    om << std::endl;
  }
  mappoints_txt = om.str();
}

// Map::loadFromTxtFile's two readers.  false: the keyframe file does not even hold its header line.
inline bool parse_txt(const std::string& keyframes_txt, const std::string& mappoints_txt, MapRec& map) {
  map = MapRec();
  std::istringstream is(keyframes_txt);
  std::string line;
  if (!std::getline(is, line)) return keyframes_txt.empty() && mappoints_txt.empty();
  {
    std::stringstream ss;
    ss << line;
    ss >> map.next_id;
    float scale;
    while (ss >> scale) map.scale_factors.push_back(scale);
  }
  while (std::getline(is, line)) {
    KeyFrameRec k;
    {
      std::stringstream ss;
      ss << line;
      ss >> k.id >> k.max_u >> k.max_v >> k.min_u >> k.min_v;
    }
    auto next = [&](std::stringstream& ss) {
      line.clear();
      std::getline(is, line);
      ss << line;
    };
    {
      std::stringstream ss;
      next(ss);
      while (true) {
        KeyPointRec kp;
        float ru, dp;
        ss >> kp.x >> kp.y >> kp.octave >> kp.angle >> ru >> dp;
        if (!ss) break;
        k.keypoints.push_back(kp), k.right_u.push_back(ru), k.depths.push_back(dp);
      }
    }
    {
      std::stringstream ss;
      next(ss);
      while (true) {
        std::array<uint8_t, 32> d{};
        for (int i = 0; i < 32; ++i) {
          int v = 0;
          ss >> v;
          d[i] = (uint8_t)v;
        }
        if (!ss) break;
        k.descriptors.push_back(d), k.descriptor_len.push_back(32);
      }
    }
    {
      std::stringstream ss;
      next(ss);
      while (true) {
        unsigned int w;
        double v;
        ss >> w >> v;
        if (!ss) break;
        k.bow.insert({w, v});
      }
    }
    {
      std::stringstream ss;
      next(ss);
      while (true) {
        FeatureNodeRec n;
        size_t num = 0;
        ss >> n.node_id >> num;
        for (size_t i = 0; i < num && ss; ++i) {
          unsigned int id = 0;
          ss >> id;
          n.feature_ids.push_back(id);
        }
        if (!ss) break;
        k.feature_nodes.push_back(n);
      }
    }
    {
      std::stringstream ss;
      next(ss);
      k.rotation.assign(9, 0.f), k.translation.assign(3, 0.f);
      for (int i = 0; i < 9; ++i) ss >> k.rotation[i];
      for (int i = 0; i < 3; ++i) ss >> k.translation[i];
    }
    {
      std::stringstream ss;
      next(ss);
      while (true) {
        uint64_t id;
        int w;
        ss >> id >> w;
        if (!ss) break;
        k.connected.push_back({id, w});
      }
    }
    {
      std::stringstream ss;
      next(ss);
      uint64_t id;
      while (ss >> id) k.children.push_back(id);
    }
    {
      std::stringstream ss;
      next(ss);
      uint64_t id;
      while (ss >> id) k.loop_edges.push_back(id);
    }
    {
      std::stringstream ss;
      next(ss);
      long id;
      while (ss >> id) k.map_points.push_back(id);
    }
    map.keyframes.push_back(std::move(k));
  }
  std::istringstream im(mappoints_txt);
  while (std::getline(im, line)) {
    MapPointRec m;
    {
      std::stringstream ss;
      ss << line;
      ss >> m.id >> m.max_distance >> m.min_distance >> m.ref_kf_id >> m.ref_feat_id >> m.matches_in_track >> m.inliers_in_track;
    }
    {
      std::stringstream ss;
      line.clear();
      std::getline(im, line);
      ss << line;
      ss >> m.position[0] >> m.position[1] >> m.position[2] >> m.view_direction[0] >> m.view_direction[1] >> m.view_direction[2];
    }
    {
      std::stringstream ss;
      line.clear();
      std::getline(im, line);
      ss << line;
      for (int i = 0; i < 32; ++i) {
        int v = 0;
        ss >> v;
        m.desc[i] = (uint8_t)v;
      }
      m.desc_len = 32;
    }
    map.mappoints.push_back(m);
  }
  return true;
}

}  // namespace mappb
}  // namespace orbfe
