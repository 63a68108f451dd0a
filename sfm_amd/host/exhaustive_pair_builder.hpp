// src/feature_matcher/exhaustive_pair_builder.hpp / .cc:3-19.
#ifndef SFM_AMD_HOST_EXHAUSTIVE_PAIR_BUILDER_HPP_
#define SFM_AMD_HOST_EXHAUSTIVE_PAIR_BUILDER_HPP_
#include "pair_builder_interface.hpp"

class ExhaustivePairBuilder : public PairBuilderInterface {
 public:
  // Every unordered pair of view ids as (smaller, larger).  Like the reference it walks
  // sfm_data.views, not sfm_data.descriptors.
  void BuildPair(SfMData& sfm_data, std::set<Pair>& pair_result) override;
};
#endif
