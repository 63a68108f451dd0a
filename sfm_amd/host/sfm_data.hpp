// Data model of the matching stage, API-compatible with the reference's
// src/sfm_data.hpp:13-57, src/descriptor.hpp:10-39, src/keypoint.hpp:8-13,
// src/types_define.hpp:5 and src/eigen_alias_types.hpp:8, gathered in one header so that
// the matcher in this directory builds on its own.  When the matcher is compiled inside the
// reference tree the reference's own headers of the same names are found first and this file
// is not used.
//
// Observation: the reference aliases Eigen::Vector2d.  On this path only its
// two-argument constructor and x()/y() are used (src/sfm_data.hpp:18-19,
// src/sfm_data_io.cc:32-35), so a two-double value type stands in when Eigen is absent.
#ifndef SFM_AMD_HOST_SFM_DATA_HPP_
#define SFM_AMD_HOST_SFM_DATA_HPP_

#include <array>
#include <cstddef>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#if defined(SFM_AMD_USE_EIGEN) && __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
using Observation = Eigen::Vector2d;
#else
class Observation {
 public:
  Observation() : xy_{0.0, 0.0} {}
  Observation(double x, double y) : xy_{x, y} {}
  double& x() { return xy_[0]; }
  double& y() { return xy_[1]; }
  const double& x() const { return xy_[0]; }
  const double& y() const { return xy_[1]; }
  bool operator==(const Observation& o) const { return xy_[0] == o.xy_[0] && xy_[1] == o.xy_[1]; }

 private:
  double xy_[2];
};
#endif

using IndexT = int64_t;

struct KeyPoint {
  double x = 0.0, y = 0.0;
  KeyPoint() = default;
  KeyPoint(double x_, double y_) : x(x_), y(y_) {}
};

// n descriptors of 128 floats, row-major in one contiguous vector.
class Descriptors {
 public:
  static constexpr std::size_t kDim = 128;
  Descriptors() = default;
  std::size_t size() const { return raw_data_.size() / kDim; }
  const float* data() const { return raw_data_.data(); }
  float* data() { return raw_data_.data(); }
  std::vector<float>& getRaw_Data() { return raw_data_; }
  const std::vector<float>& getRaw_Data() const { return raw_data_; }
  void push_back(const std::array<float, 128>& d) { raw_data_.insert(raw_data_.end(), d.begin(), d.end()); }

 private:
  std::vector<float> raw_data_;
};

struct View {
  std::string image_path;
};

template <typename T>
Observation ToObservation(const T& p) {
  return Observation(p.x, p.y);
}

struct Match {
  Observation lhs_observation, rhs_observation;
  IndexT lhs_idx = 0, rhs_idx = 0;
};

using Pair = std::pair<IndexT, IndexT>;
using Matches = std::vector<Match>;

struct SparsePoint {
  double x, y, z;
  explicit SparsePoint(double x_ = 0.0, double y_ = 0.0, double z_ = 0.0) : x(x_), y(y_), z(z_) {}
  std::map<IndexT, Observation> obs;
};

struct SfMData {
  std::size_t image_width = 0, image_height = 0;
  std::map<IndexT, View> views;
  std::map<IndexT, std::vector<KeyPoint>> key_points;
  std::map<IndexT, Descriptors> descriptors;
  std::map<Pair, Matches> matches;
  std::vector<SparsePoint> structure_points;
};

#endif  // SFM_AMD_HOST_SFM_DATA_HPP_
