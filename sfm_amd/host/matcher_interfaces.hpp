// The two abstract classes the matching stage of the reference is written against, and its
// exhaustive pair builder, declared here with the reference's names and signatures so that code
// written for the reference compiles unchanged:
//   FeatureMatcherInterface   src/feature_matcher/feature_matcher_interface.hpp:5-11
//   PairBuilderInterface      src/feature_matcher/pair_builder_interface.hpp:6-11
//   ExhaustivePairBuilder     src/feature_matcher/exhaustive_pair_builder.hpp, .cc:3-19
// The headers carrying the reference's file names only include this one.
#ifndef SFM_AMD_HOST_MATCHER_INTERFACES_HPP_
#define SFM_AMD_HOST_MATCHER_INTERFACES_HPP_
#include <set>

#include "sfm_data.hpp"

// A matcher fills sfm_data.matches[pair] for every pair of the set (whatever was stored under
// those keys before is replaced); other keys are left alone.
class FeatureMatcherInterface {
 public:
  virtual ~FeatureMatcherInterface() = default;
  virtual void Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid = true) = 0;
};

// A pair builder decides which image pairs are matched.
class PairBuilderInterface {
 public:
  virtual ~PairBuilderInterface() = default;
  virtual void BuildPair(SfMData& sfm_data, std::set<Pair>& pair_result) = 0;
};

// Every unordered pair of view ids as (smaller, larger).  Like the reference it walks
// sfm_data.views, not sfm_data.descriptors.
class ExhaustivePairBuilder : public PairBuilderInterface {
 public:
  void BuildPair(SfMData& sfm_data, std::set<Pair>& pair_result) override;
};
#endif
