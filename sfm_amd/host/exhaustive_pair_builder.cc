#include "exhaustive_pair_builder.hpp"

void ExhaustivePairBuilder::BuildPair(SfMData& sfm_data, std::set<Pair>& pair_result) {
  // map keys come out ascending, so (first, second) is already (min, max)
  for (auto a = sfm_data.views.begin(); a != sfm_data.views.end(); ++a) {
    auto b = a;
    for (++b; b != sfm_data.views.end(); ++b) pair_result.emplace(a->first, b->first);
  }
}
