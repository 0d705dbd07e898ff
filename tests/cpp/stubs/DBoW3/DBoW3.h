// A STAND-IN for <DBoW3/DBoW3.h> (tests/test_reference_compile.py): the three DBoW3 types the reference's frame classes hold, with
// DBoW3's names and container shapes -- BowVector = map<WordId, WordValue>, FeatureVector = map<NodeId, vector<unsigned>> --
// declarations only.  NOT DBoW3.
#pragma once
#include <map>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

namespace DBoW3 {
typedef unsigned WordId;
typedef double WordValue;
typedef unsigned NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned>> {};
class Vocabulary {
 public:
  Vocabulary();
  explicit Vocabulary(const std::string& file);
  void load(const std::string& file);
  bool empty() const;
  void transform(const std::vector<cv::Mat>& features, BowVector& v, FeatureVector& fv, int levelsup) const;
  double score(const BowVector& a, const BowVector& b) const;
};
}  // namespace DBoW3
