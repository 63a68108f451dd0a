// The matcher boundary of the reference: src/feature_matcher/feature_matcher_interface.hpp:5-11.
#ifndef SFM_AMD_HOST_FEATURE_MATCHER_INTERFACE_HPP_
#define SFM_AMD_HOST_FEATURE_MATCHER_INTERFACE_HPP_
#include <set>

#include "sfm_data.hpp"

class FeatureMatcherInterface {
 public:
  // Overwrites sfm_data.matches[pair] for every pair of the set.
  virtual void Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid = true) = 0;
  virtual ~FeatureMatcherInterface() = default;
};
#endif
