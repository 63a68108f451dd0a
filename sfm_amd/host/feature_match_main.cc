// Matching stage driver: src/feature_match_main.cc:7-33 with the GPU brute-force matcher.
//   feature_match_main <sfm_data.bin> [--no-cross-check]
// Load -> all pairs over the views -> Match -> Save back to the same path.  As in the
// reference a failed Load is reported and matching goes on with whatever was loaded (nothing).
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>

#include "exhaustive_pair_builder.hpp"
#include "hip_brute_force_matcher.hpp"
#include "sfm_data_io.hpp"

int main(int argc, char** argv) {
  if (argc < 2 || argc > 3) return 1;
  const bool cross = !(argc == 3 && std::strcmp(argv[2], "--no-cross-check") == 0);

  SfMData sfm_data;
  if (Load(sfm_data, argv[1])) {
    std::printf("Load Sfm Data Finish\n");
  } else {
    std::printf("Load Sfm Data From %s Fails\n", argv[1]);
  }

  auto pair_selector = std::make_shared<ExhaustivePairBuilder>();
  std::set<Pair> pairs;
  pair_selector->BuildPair(sfm_data, pairs);
  std::printf("Need To Match %lu Pairs.\n", (unsigned long)pairs.size());

  try {
    auto matcher = std::make_shared<HipBruteForceMatcher>();
    matcher->Match(sfm_data, pairs, cross);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 2;
  }
  return Save(sfm_data, argv[1]) ? 0 : 3;
}
