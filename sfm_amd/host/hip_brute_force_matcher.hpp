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
#include "feature_matcher_interface.hpp"

class HipBruteForceMatcher : public FeatureMatcherInterface {
 public:
  // n_devices <= 0: use SFM_HIP_DEVICES if set, else every visible device.
  explicit HipBruteForceMatcher(int n_devices = 0, double ratio = 0.8, bool print_progress = true)
      : n_devices_(n_devices), ratio_(ratio), print_progress_(print_progress) {}
  void Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid = true) override;
  ~HipBruteForceMatcher() override = default;
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
};

// When this file is compiled into the reference tree in place of brute_force_matcher.cc,
// -DSFM_HIP_REPLACES_BRUTE_FORCE makes the reference's own class name resolve to the GPU matcher.
#ifdef SFM_HIP_REPLACES_BRUTE_FORCE
using BruteForceMatcher = HipBruteForceMatcher;
#endif
#endif
