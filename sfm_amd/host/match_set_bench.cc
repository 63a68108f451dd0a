// Whole-job timing of the matching stage through the C++ class a host program uses:
// build a synthetic image set in memory, enumerate all pairs, call HipBruteForceMatcher::Match
// (uploads, GPU matching on every visible device, Match records built on the host) and report
// pairs per second over the whole call.
//
//   match_set_bench <n_images> <n_features> [planted_fraction=0.5] [devices=0] [engine=-1] [matcher=brute] [sink=store]
//   (engine: -1 library default / SFM_HIP_ENGINE, 0 fp32 matrix instructions, 1 split fp16;
//    matcher: brute | cascade -- with cascade every shared scene row appears twice per image with
//    little noise: that matcher reports nothing unless a query has two near-duplicate candidates;
//    sink, brute only: store -- Match(), every record kept in sfm_data.matches as the reference does;
//    discard -- MatchStream() with a sink that checks and counts each pair's records and drops
//    them: the job's memory stays what the descriptors need, however many pairs it has)
//
// The line it prints also says what the host side cost: peak resident memory, Match records built
// per second per helper thread, and from that the host cores eight GPUs would keep busy.
//
// Images share a pool of "scene" descriptors (non-negative rows of unit length, as RootSIFT
// produces: vlfeat_feature_extractor.cc:15-22); each image takes planted_fraction of its rows from
// the pool with a little noise, in shuffled order, and fills the rest with rows of its own.
#include <sys/resource.h>
#include <unistd.h>

#include <atomic>
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

// A pair's list as one number that depends on every record's indices and position; the job's
// checksum is the sum over pairs, so it does not depend on the order pairs finish in.
uint64_t list_checksum(const Pair& pair, const Matches& list) {
  auto mix = [](uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  uint64_t sum = mix((uint64_t)pair.first * 1000003ull + (uint64_t)pair.second + 0x9E3779B97F4A7C15ull * list.size());
  const uint64_t key = mix((uint64_t)pair.first << 32 ^ (uint64_t)pair.second);
  for (size_t m = 0; m < list.size(); m++)
    sum += mix(key + 0x9E3779B97F4A7C15ull * (m + 1) + ((uint64_t)list[m].lhs_idx << 24) + (uint64_t)list[m].rhs_idx);
  return sum;
}

// resident set size now, MiB (/proc/self/statm: pages)
double rss_mib() {
  long pages = 0, resident = 0;
  if (FILE* f = std::fopen("/proc/self/statm", "r")) {
    if (std::fscanf(f, "%ld %ld", &pages, &resident) != 2) resident = 0;
    std::fclose(f);
  }
  return (double)resident * (double)sysconf(_SC_PAGESIZE) / 1048576.0;
}
double peak_rss_mib() {
  struct rusage u {};
  getrusage(RUSAGE_SELF, &u);
  return (double)u.ru_maxrss / 1024.0;   // kilobytes on Linux
}

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
  const std::string sink_mode = argc > 7 ? argv[7] : "store";
  if (sink_mode != "store" && sink_mode != "discard") {
    std::fprintf(stderr, "sink must be store or discard\n");
    return 2;
  }
  if (cascade && sink_mode != "store") {
    std::fprintf(stderr, "the cascade matcher has no streaming form\n");
    return 2;
  }
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
  std::shared_ptr<HipBruteForceMatcher> brute;
  if (cascade) {
    matcher_ptr = std::make_shared<HipCascadeHashMatcher>(devices, /*print_progress=*/false);
  } else {
    brute = std::make_shared<HipBruteForceMatcher>(devices, 0.8, /*print_progress=*/false);
    brute->set_engine(engine);
    matcher_ptr = brute;
  }
  FeatureMatcherInterface& matcher = *matcher_ptr;
  // The discarding sink: looks at every record once (keypoints encode (feature index, image id)),
  // counts, and lets the list go.  Resident memory is sampled every 256 pairs.
  std::atomic<size_t> s_total{0}, s_wrong{0}, s_pairs{0};
  std::atomic<uint64_t> s_sum{0};
  std::atomic<long> s_rss_max_kib{0};
  auto discard = [&](const Pair& pair, Matches&& list) {
    size_t wrong = 0;
    for (const Match& m : list)
      if (m.lhs_observation.x() != (double)m.lhs_idx || m.rhs_observation.y() != (double)pair.second) wrong++;
    s_total += list.size();
    s_wrong += wrong;
    s_sum += list_checksum(pair, list);
    if ((++s_pairs & 255) == 0) {
      const long now = (long)(rss_mib() * 1024.0);
      long seen = s_rss_max_kib.load();
      while (now > seen && !s_rss_max_kib.compare_exchange_weak(seen, now)) {
      }
    }
  };
  const double rss_before = rss_mib();
  // SFM_MATCH_BENCH_REPEAT=n: call Match n times and report the last (the first call of a
  // process also pays for loading the GPU code and creating the work space)
  const char* rep_env = std::getenv("SFM_MATCH_BENCH_REPEAT");
  const int repeat = rep_env ? std::max(1, std::atoi(rep_env)) : 1;
  double sec = 0;
  for (int rep = 0; rep < repeat; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    if (sink_mode == "discard") {
      s_total = 0;
      s_wrong = 0;
      s_sum = 0;
      brute->MatchStream(data, pairs, true, discard);
    } else {
      matcher.Match(data, pairs, true);
    }
    sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double rss_after = rss_mib();

  size_t total = s_total, wrong = s_wrong;
  uint64_t checksum = s_sum;
  const size_t stored_lists = data.matches.size();
  for (const auto& [pair, list] : data.matches) {
    total += list.size();
    checksum += list_checksum(pair, list);
    for (const Match& m : list)   // keypoints encode (feature index, image id)
      if (m.lhs_observation.x() != (double)m.lhs_idx || m.rhs_observation.y() != (double)pair.second) wrong++;
  }
  std::printf("{\"matcher\": \"%s\", \"engine\": %d, \"images\": %d, \"features\": %d, \"pairs\": %zu, \"seconds\": %.4f, \"pairs_per_s\": %.1f, "
              "\"matches\": %zu, \"matches_per_pair\": %.1f, \"observation_mismatches\": %zu, \"sink\": \"%s\", "
              "\"records_checksum\": \"%016llx\", \"lists_in_sfm_data\": %zu, "
              "\"rss_mib_before_matching\": %.0f, \"rss_mib_after_matching\": %.0f, \"rss_mib_max_seen_by_sink\": %.0f, \"peak_rss_mib\": %.0f",
              cascade ? "cascade" : "brute", engine, n_images, n_feat, pairs.size(), sec, pairs.size() / sec, total,
              (double)total / pairs.size(), wrong, sink_mode.c_str(), (unsigned long long)checksum, stored_lists, rss_before, rss_after, (double)s_rss_max_kib / 1024.0, peak_rss_mib());
  if (brute) {
    // host side of the last call: what one helper thread builds per second, and what eight GPUs at the
    // single-GPU rate measured here, with this many matches per pair, would ask of the host
    const HipBruteForceMatcher::HostStats& h = brute->host_stats();
    const double per_thread = h.materialise_thread_seconds > 0 ? (double)h.records / h.materialise_thread_seconds : 0.0;
    const double per_gpu_pairs = pairs.size() / sec / std::max(1, h.gpus);
    const double need8 = per_thread > 0 ? 8.0 * per_gpu_pairs * ((double)total / pairs.size()) / per_thread : 0.0;
    std::printf(", \"gpus\": %d, \"helper_threads_per_gpu\": %d, \"match_records\": %llu, \"helper_thread_seconds\": %.3f, "
                "\"records_per_s_per_helper_thread\": %.0f, \"host_cores_to_feed_8_gpus_at_this_rate\": %.2f",
                h.gpus, h.helper_threads_per_gpu, (unsigned long long)h.records, h.materialise_thread_seconds, per_thread, need8);
  }
  std::printf("}\n");
  return wrong ? 1 : 0;
}
