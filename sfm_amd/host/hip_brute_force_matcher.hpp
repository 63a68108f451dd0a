// Drop-in for the reference's BruteForceMatcher
// (src/feature_matcher/brute_force_matcher.hpp:5-10, .cc:8-88) that runs on MI355X GPUs
// through the C ABI of include/sfm_hip.h.  Same interface, same results: for every pair of the
// set, sfm_data.matches[pair] receives the matches the reference would store, in the same order.
//
// Differences a caller can observe:
//   * a failure of the GPU path throws std::runtime_error (the reference cannot fail);
//     there is no CPU fallback;
//   * key_points[id] shorter than descriptors[id] throws std::out_of_range where the reference
//     reads out of bounds (brute_force_matcher.cc:36);
//   * the work is spread over every visible GPU (or SFM_HIP_DEVICES=n of them) instead of the
//     reference's two worker threads.
// The distance engine (include/sfm_hip.h: SFM_HIP_ENGINE_F32 / SFM_HIP_ENGINE_F16X3) can be chosen
// with set_engine() or the environment variable SFM_HIP_ENGINE=f32|f16x3; the matches are the same.
// How the descriptors reach the GPUs can be chosen with set_replication() or SFM_HIP_REPLICATION:
// "host" (default) -- every GPU uploads, from host memory, the images its shard of the pair list
// names; "collective" -- every image is uploaded to one GPU only and an RCCL all-gather over xGMI
// hands everything to all GPUs (sfm_hip_replicate_images).
#ifndef SFM_AMD_HOST_HIP_BRUTE_FORCE_MATCHER_HPP_
#define SFM_AMD_HOST_HIP_BRUTE_FORCE_MATCHER_HPP_
#include <cstdint>
#include <functional>

#include "feature_matcher_interface.hpp"

class HipBruteForceMatcher : public FeatureMatcherInterface {
 public:
  // n_devices <= 0: use SFM_HIP_DEVICES if set, else every visible device.
  explicit HipBruteForceMatcher(int n_devices = 0, double ratio = 0.8, bool print_progress = true)
      : n_devices_(n_devices), ratio_(ratio), print_progress_(print_progress) {}
  void Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid = true) override;
  ~HipBruteForceMatcher() override = default;

  // The same matching with a bounded memory footprint.  Match() stores every pair's list in
  // sfm_data.matches, as the reference does (brute_force_matcher.cc:80,83): 48 bytes per Match
  // record, which is tens to hundreds of gigabytes for some hundred thousand pairs.  MatchStream
  // instead hands each finished pair's list -- the very records Match() would store, in the same
  // order -- to `sink` and keeps nothing: the sink writes them to a file, filters them, feeds the next
  // stage, or drops them.  sfm_data.matches is not touched.  The sink is called once per pair of
  // the set, from the matcher's helper threads: calls for different pairs may run at the same time
  // (never two for one pair), in no particular pair order; an exception it throws ends the call and
  // is rethrown to the caller.  No counterpart in the reference, whose only result path is the map.
  using MatchSink = std::function<void(const Pair& pair, Matches&& matches)>;
  void MatchStream(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid, const MatchSink& sink);

  // What the host side of the last Match()/MatchStream() call did: Match records built (and handed
  // to the sink), and the time the helper threads spent on that, summed over threads.  records /
  // materialise_thread_seconds is the rate of one helper thread; the GPUs' match rate times the
  // matches per pair, divided by it, is the number of host cores a job needs to keep up.
  struct HostStats {
    uint64_t pairs = 0, records = 0;
    double materialise_thread_seconds = 0;
    int gpus = 0, helper_threads_per_gpu = 0;
  };
  const HostStats& host_stats() const { return host_stats_; }
  // threads per GPU that build Match records; <= 0: SFM_HIP_HELPERS if set, else min(8, cores / GPUs)
  void set_helper_threads(int n) { helper_threads_ = n; }
  // engine < 0: use SFM_HIP_ENGINE if set, else the library's default.
  void set_engine(int engine) { engine_ = engine; }
  // 0: host uploads, 1: collective; < 0: use SFM_HIP_REPLICATION if set, else host uploads.
  void set_replication(int mode) { replication_ = mode; }

 private:
  int n_devices_;
  double ratio_;        // the reference hard-codes 0.8 (brute_force_matcher.cc:35)
  bool print_progress_;
  int engine_ = -1;
  int replication_ = -1;
  int helper_threads_ = 0;
  HostStats host_stats_;
};

// When this file is compiled into the reference tree in place of brute_force_matcher.cc,
// -DSFM_HIP_REPLACES_BRUTE_FORCE makes the reference's own class name resolve to the GPU matcher.
#ifdef SFM_HIP_REPLACES_BRUTE_FORCE
using BruteForceMatcher = HipBruteForceMatcher;
#endif
#endif
