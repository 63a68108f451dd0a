// Drop-in for the reference's GeometryFilter (src/sfm_main.cc:259-282): for every image pair of
// sfm_data.matches a fundamental matrix by RANSAC (ComputeFundamentalMatrix, src/sfm_main.cc:17-50) on
// MI355X GPUs through sfm_hip_geometry_verify_pairs (include/sfm_hip.h); a pair keeps the matches
// whose epipolar-line error under that matrix is below 3.84 if there are more than 30 of them, and
// is dropped from sfm_data.matches otherwise.  Prints the reference's lines ("<n>Match ",
// "inlier : <k>").
//
// The reference's sampler is seeded from std::random_device in every round, so the reference itself
// gives a different answer each time it runs; here the random stream is a function of `seed` and of
// the pair's position in sfm_data.matches, and a given seed always gives the same answer.  A pair
// with fewer than eight matches, which the reference cannot sample (it asserts), is dropped.
// A failure of the GPU path throws std::runtime_error; there is no CPU fallback.
#ifndef SFM_AMD_HOST_HIP_GEOMETRY_FILTER_HPP_
#define SFM_AMD_HOST_HIP_GEOMETRY_FILTER_HPP_
#include <array>
#include <cstdint>
#include <map>

#include "sfm_data.hpp"

struct HipGeometryFilterOptions {
  std::uint64_t seed = 0;        // 0: take one from std::random_device, as unrepeatable as the reference
  int n_devices = 0;             // <= 0: SFM_HIP_DEVICES if set, else every visible device
  bool print_progress = true;
  // if not null: receives the fundamental matrix (row-major) of every pair that was kept
  std::map<Pair, std::array<double, 9>>* fundamental = nullptr;
};

void HipGeometryFilter(SfMData& sfm_data, const HipGeometryFilterOptions& options = HipGeometryFilterOptions());
#endif
