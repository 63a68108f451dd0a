// Whole-job timing of the matching stage through the C++ class a host program uses:
// build a synthetic image set in memory, enumerate all pairs, call HipBruteForceMatcher::Match
// (uploads, GPU matching on every visible device, Match records built on the host) and report
// pairs per second over the whole call.
//
//   match_set_bench <n_images> <n_features> [planted_fraction=0.5] [devices=0] [engine=-1] [matcher=brute]
//   (engine: -1 library default / SFM_HIP_ENGINE, 0 fp32 matrix instructions, 1 split fp16;
//    matcher: brute | cascade -- with cascade every shared scene row appears twice per image with
//    little noise: that matcher reports nothing unless a query has two near-duplicate candidates)
//
// Images share a pool of "scene" descriptors (non-negative rows of unit length, as RootSIFT
// produces: vlfeat_feature_extractor.cc:15-22); each image takes planted_fraction of its rows from
// the pool with a little noise, in shuffled order, and fills the rest with rows of its own.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <numeric>
#include <set>
#include <vector>

#include "exhaustive_pair_builder.hpp"
#include <memory>
#include <string>

#include "hip_brute_force_matcher.hpp"
#include "hip_cascade_hash_matcher.hpp"
#include "sfm_data.hpp"

namespace {
struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double noise() {   // sum of four uniforms, variance 1
    double t = 0;
    for (int k = 0; k < 4; k++) t += uniform();
    return (t - 2.0) * std::sqrt(3.0);
  }
};

void scene_row(Rng& rng, float* row) {
  double u[128], sum = 0;
  for (double& v : u) sum += (v = rng.uniform());
  for (int k = 0; k < 128; k++) row[k] = (float)std::sqrt(u[k] / sum);
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <n_images> <n_features> [planted_fraction] [devices] [engine]\n", argv[0]);
    return 2;
  }
  const int n_images = std::atoi(argv[1]);
  const int n_feat = std::atoi(argv[2]);
  const double planted = argc > 3 ? std::atof(argv[3]) : 0.5;
  const int devices = argc > 4 ? std::atoi(argv[4]) : 0;
  const int engine = argc > 5 ? std::atoi(argv[5]) : -1;
  const bool cascade = argc > 6 && std::string(argv[6]) == "cascade";
  const int copies = cascade ? 2 : 1;
  const double sigma = cascade ? 0.002 : 0.01;
  const int n_planted = (int)(planted * n_feat) / copies * copies;   // rows of an image taken from the pool

  Rng rng{12345};
  std::vector<float> pool((size_t)(n_planted / copies) * 128);
  for (int r = 0; r < n_planted / copies; r++) scene_row(rng, &pool[(size_t)r * 128]);

  SfMData data;
  for (int id = 0; id < n_images; id++) {
    data.views[id] = View{"synthetic"};
    std::vector<int> order(n_feat);
    std::iota(order.begin(), order.end(), 0);
    for (int k = n_feat - 1; k > 0; k--) std::swap(order[k], order[rng.next() % (uint64_t)(k + 1)]);
    std::vector<float> raw((size_t)n_feat * 128);
    for (int r = 0; r < n_feat; r++) {
      float* row = &raw[(size_t)order[r] * 128];
      if (r < n_planted) {
        for (int k = 0; k < 128; k++) row[k] = pool[(size_t)(r / copies) * 128 + k] + (float)(sigma * rng.noise());
      } else {
        scene_row(rng, row);
      }
    }
    Descriptors d;
    d.getRaw_Data() = raw;
    data.descriptors[id] = d;
    std::vector<KeyPoint> kp((size_t)n_feat);
    for (int r = 0; r < n_feat; r++) kp[(size_t)r] = KeyPoint{(double)r, (double)id};
    data.key_points[id] = kp;
  }

  std::set<Pair> pairs;
  ExhaustivePairBuilder().BuildPair(data, pairs);
  std::shared_ptr<FeatureMatcherInterface> matcher_ptr;
  if (cascade) {
    matcher_ptr = std::make_shared<HipCascadeHashMatcher>(devices, /*print_progress=*/false);
  } else {
    auto brute = std::make_shared<HipBruteForceMatcher>(devices, 0.8, /*print_progress=*/false);
    brute->set_engine(engine);
    matcher_ptr = brute;
  }
  FeatureMatcherInterface& matcher = *matcher_ptr;
  // SFM_MATCH_BENCH_REPEAT=n: call Match n times and report the last (the first call of a
  // process also pays for loading the GPU code and creating the work space)
  const char* rep_env = std::getenv("SFM_MATCH_BENCH_REPEAT");
  const int repeat = rep_env ? std::max(1, std::atoi(rep_env)) : 1;
  double sec = 0;
  for (int rep = 0; rep < repeat; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    matcher.Match(data, pairs, true);
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  size_t total = 0, wrong = 0;
  for (const auto& [pair, list] : data.matches) {
    total += list.size();
    for (const Match& m : list)   // keypoints encode (feature index, image id)
      if (m.lhs_observation.x() != (double)m.lhs_idx || m.rhs_observation.y() != (double)pair.second) wrong++;
  }
  std::printf("{\"matcher\": \"%s\", \"engine\": %d, \"images\": %d, \"features\": %d, \"pairs\": %zu, \"seconds\": %.4f, \"pairs_per_s\": %.1f, "
              "\"matches\": %zu, \"matches_per_pair\": %.1f, \"observation_mismatches\": %zu}\n",
              cascade ? "cascade" : "brute", engine, n_images, n_feat, pairs.size(), sec, pairs.size() / sec, total, (double)total / pairs.size(), wrong);
  return wrong ? 1 : 0;
}
