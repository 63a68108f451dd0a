#include "hip_geometry_filter.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "sfm_hip.h"

namespace {

struct Shard {
  std::vector<int64_t> offsets;       // per pair of the shard, starting at 0
  std::vector<double> lhs, rhs;       // x, y per match
  std::vector<double> F;              // 9 per pair
  std::vector<uint8_t> flags;
  std::vector<int32_t> counts;
};

void run_device(int device, Shard& s, uint64_t seed, int64_t first_pair) {
  sfm_hip_ctx* ctx = nullptr;
  if (sfm_hip_ctx_create(device, &ctx) != SFM_HIP_OK)
    throw std::runtime_error(std::string("HipGeometryFilter: sfm_hip_ctx_create: ") + sfm_hip_last_error(nullptr));
  const int32_t n = (int32_t)s.offsets.size() - 1;
  s.F.assign((size_t)n * 9, 0.0);
  s.flags.assign(s.lhs.size() / 2 + 1, 0);
  s.counts.assign((size_t)n, 0);
  const int rc = sfm_hip_geometry_verify_pairs(ctx, n, s.offsets.data(), s.lhs.data(), s.rhs.data(), seed, first_pair,
                                               s.F.data(), s.flags.data(), s.counts.data(), nullptr);
  const std::string why = rc == SFM_HIP_OK ? "" : sfm_hip_last_error(ctx);
  sfm_hip_ctx_destroy(ctx);
  if (rc != SFM_HIP_OK) throw std::runtime_error("HipGeometryFilter: sfm_hip_geometry_verify_pairs: " + why);
}

}  // namespace

void HipGeometryFilter(SfMData& sfm_data, const HipGeometryFilterOptions& options) {
  if (options.print_progress) std::cout << sfm_data.matches.size() << "Match " << std::endl;   // sfm_main.cc:260
  if (options.fundamental) options.fundamental->clear();
  if (sfm_data.matches.empty()) return;
  uint64_t seed = options.seed;
  if (seed == 0) {
    std::random_device rd;
    seed = ((uint64_t)rd() << 32) | rd() | 1u;
  }
  int visible = 0;
  if (sfm_hip_device_count(&visible) != SFM_HIP_OK || visible <= 0)
    throw std::runtime_error(std::string("HipGeometryFilter: no HIP device: ") + sfm_hip_last_error(nullptr));
  int want = options.n_devices;
  if (want <= 0) {
    const char* env = std::getenv("SFM_HIP_DEVICES");
    want = env ? std::atoi(env) : visible;
    if (want <= 0) want = visible;
  }
  const char* over = std::getenv("SFM_HIP_OVERSUBSCRIBE");   // more shards than GPUs, for testing
  const bool oversubscribe = over && std::string(over) == "1";
  const size_t n_pairs = sfm_data.matches.size();
  const int n_dev = (int)std::min<size_t>((size_t)(oversubscribe ? want : std::min(want, visible)), n_pairs);

  // contiguous shards of the pairs in map order (the order sfm_main.cc:262 walks them in)
  std::vector<const Pair*> keys;
  std::vector<const Matches*> lists;
  for (const auto& kv : sfm_data.matches) {
    keys.push_back(&kv.first);
    lists.push_back(&kv.second);
  }
  std::vector<Shard> shards((size_t)n_dev);
  std::vector<size_t> first((size_t)n_dev + 1, 0);
  const size_t base = n_pairs / n_dev, extra = n_pairs % n_dev;
  for (int d = 0; d < n_dev; d++) first[(size_t)d + 1] = first[(size_t)d] + base + ((size_t)d < extra ? 1 : 0);
  for (int d = 0; d < n_dev; d++) {
    Shard& s = shards[(size_t)d];
    s.offsets.push_back(0);
    for (size_t p = first[(size_t)d]; p < first[(size_t)d + 1]; p++) {
      for (const Match& m : *lists[p]) {           // sfm_main.cc:264-267
        s.lhs.push_back(m.lhs_observation.x());
        s.lhs.push_back(m.lhs_observation.y());
        s.rhs.push_back(m.rhs_observation.x());
        s.rhs.push_back(m.rhs_observation.y());
      }
      s.offsets.push_back((int64_t)(s.lhs.size() / 2));
    }
  }
  std::vector<std::thread> workers;
  std::vector<std::exception_ptr> errors((size_t)n_dev);
  for (int d = 0; d < n_dev; d++)
    workers.emplace_back([&, d] {
      try {
        run_device(d % visible, shards[(size_t)d], seed, (int64_t)first[(size_t)d]);
      } catch (...) {
        errors[(size_t)d] = std::current_exception();
      }
    });
  for (auto& t : workers) t.join();
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);

  std::map<Pair, Matches> filter_matches;            // sfm_main.cc:261, :273-281
  for (int d = 0; d < n_dev; d++) {
    const Shard& s = shards[(size_t)d];
    for (size_t p = first[(size_t)d]; p < first[(size_t)d + 1]; p++) {
      const size_t k = p - first[(size_t)d];
      const int32_t inliers = s.counts[k];
      if (options.print_progress) std::printf("inlier : %lu\n", (unsigned long)inliers);   // sfm_main.cc:43
      if (!(inliers > 30)) continue;                 // sfm_main.cc:44, :272
      Matches kept;
      kept.reserve((size_t)inliers);
      const Matches& all = *lists[p];
      for (size_t i = 0; i < all.size(); i++)
        if (s.flags[(size_t)s.offsets[k] + i]) kept.push_back(all[i]);
      filter_matches[*keys[p]] = std::move(kept);
      if (options.fundamental) {
        std::array<double, 9> F;
        std::copy(s.F.begin() + (long)k * 9, s.F.begin() + (long)k * 9 + 9, F.begin());
        (*options.fundamental)[*keys[p]] = F;
      }
    }
  }
  sfm_data.matches = std::move(filter_matches);      // sfm_main.cc:281
}
