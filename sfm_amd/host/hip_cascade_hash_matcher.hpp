// Drop-in for the reference's CascadeHashMatcher
// (src/feature_matcher/cascade_hash_matcher.hpp:24-45, .cc:118-205), the matcher the reference's
// driver runs (src/feature_match_main.cc:26), on MI355X GPUs through the C ABI of
// include/sfm_hip.h (sfm_hip_cascade_*).  Same interface; for every pair of the set
// sfm_data.matches[pair] receives what the reference would store: with cross-check one
// value-initialised record per forward match first (the reference builds its result in
// `Matches temp(matches.size())`, .cc:187, and appends to it), then the surviving matches in
// ascending lhs index; observations of every record, default ones included, are filled from the
// key points (.cc:197-202).
//
// What is not reproduced bit for bit, because the reference's source does not fix it: the order in
// which the reference's matrix library adds floats in the mean vector and in the hash projections.
// This implementation adds in index order, unfused.  A hash bit of a descriptor whose projection
// lies within rounding distance of zero can therefore differ from a particular build of the
// reference.
//
// Differences a caller can observe otherwise:
//   * a failure of the GPU path throws std::runtime_error; there is no CPU fallback;
//   * key_points[id] shorter than descriptors[id] throws std::out_of_range where the reference
//     reads out of bounds;
//   * the pair list is spread over every visible GPU (or SFM_HIP_DEVICES=n of them), every GPU
//     holding all images; the reference walks it on one thread;
//   * the per-pair lines ("Piar : [a,b] n Matches.") are printed in pair order once the pairs are
//     done, not while they are being worked on.
#ifndef SFM_AMD_HOST_HIP_CASCADE_HASH_MATCHER_HPP_
#define SFM_AMD_HOST_HIP_CASCADE_HASH_MATCHER_HPP_
#include "feature_matcher_interface.hpp"

class HipCascadeHashMatcher : public FeatureMatcherInterface {
 public:
  // n_devices <= 0: use SFM_HIP_DEVICES if set, else every visible device.
  explicit HipCascadeHashMatcher(int n_devices = 0, bool print_progress = true)
      : n_devices_(n_devices), print_progress_(print_progress) {}
  void Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid = true) override;
  ~HipCascadeHashMatcher() override = default;

 private:
  int n_devices_;
  bool print_progress_;
};

// When this file is compiled into the reference tree in place of cascade_hash_matcher.cc,
// -DSFM_HIP_REPLACES_CASCADE_HASH makes the reference's own class name resolve to the GPU matcher.
#ifdef SFM_HIP_REPLACES_CASCADE_HASH
using CascadeHashMatcher = HipCascadeHashMatcher;
#endif
#endif
