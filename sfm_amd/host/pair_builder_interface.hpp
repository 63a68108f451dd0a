// src/feature_matcher/pair_builder_interface.hpp:6-11.
#ifndef SFM_AMD_HOST_PAIR_BUILDER_INTERFACE_HPP_
#define SFM_AMD_HOST_PAIR_BUILDER_INTERFACE_HPP_
#include <set>

#include "sfm_data.hpp"

class PairBuilderInterface {
 public:
  virtual void BuildPair(SfMData& sfm_data, std::set<Pair>& pair_result) = 0;
  virtual ~PairBuilderInterface() = default;
};
#endif
