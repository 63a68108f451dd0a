#include "hip_brute_force_matcher.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "sfm_hip.h"

namespace {

struct PairResult {
  std::vector<int32_t> lhs, rhs;
};

[[noreturn]] void fail(sfm_hip_ctx* ctx, const char* what) {
  std::string msg = std::string("HipBruteForceMatcher: ") + what + ": " + sfm_hip_last_error(ctx);
  throw std::runtime_error(msg);
}

// One GPU: upload the images its pairs touch, match its slice of the pair list in sub-batches.
void run_device(int device, const std::vector<Pair>& pairs, size_t lo, size_t hi, SfMData& sfm_data, double ratio,
                bool cross, std::vector<PairResult>& results) {
  sfm_hip_ctx* ctx = nullptr;
  if (sfm_hip_ctx_create(device, &ctx) != SFM_HIP_OK) fail(nullptr, "sfm_hip_ctx_create");
  struct Guard {
    sfm_hip_ctx* c;
    std::vector<void*> pinned;
    ~Guard() {
      for (void* p : pinned) sfm_hip_host_free(p);
      sfm_hip_ctx_destroy(c);
    }
  } guard{ctx, {}};

  std::set<IndexT> ids;
  int64_t stride = 1;
  for (size_t p = lo; p < hi; p++) {
    ids.insert(pairs[p].first);
    ids.insert(pairs[p].second);
  }
  for (IndexT id : ids) {
    const Descriptors& d = sfm_data.descriptors.at(id);
    if (sfm_hip_upload_image(ctx, id, d.data(), (int32_t)d.size()) != SFM_HIP_OK) fail(ctx, "sfm_hip_upload_image");
    stride = std::max<int64_t>(stride, (int64_t)d.size());
  }

  // sub-batches bound the pinned result buffers (stride ints per pair and array)
  const size_t batch = std::max<size_t>(1, std::min<size_t>(256, ((size_t)64 << 20) / ((size_t)stride * 8)));
  auto pin = [&](size_t bytes) {
    void* p = sfm_hip_host_alloc(bytes);
    if (!p) throw std::runtime_error("HipBruteForceMatcher: pinned host allocation failed");
    guard.pinned.push_back(p);
    return p;
  };
  int32_t* counts = (int32_t*)pin(batch * sizeof(int32_t));
  int32_t* out_l = (int32_t*)pin(batch * (size_t)stride * sizeof(int32_t));
  int32_t* out_r = (int32_t*)pin(batch * (size_t)stride * sizeof(int32_t));
  std::vector<int64_t> flat(2 * batch);
  for (size_t b0 = lo; b0 < hi; b0 += batch) {
    const size_t n = std::min(batch, hi - b0);
    for (size_t k = 0; k < n; k++) {
      flat[2 * k] = pairs[b0 + k].first;
      flat[2 * k + 1] = pairs[b0 + k].second;
    }
    if (sfm_hip_match_pairs(ctx, flat.data(), (int32_t)n, ratio, cross ? 1 : 0, counts, out_l, out_r, nullptr,
                            stride) != SFM_HIP_OK)
      fail(ctx, "sfm_hip_match_pairs");
    if (sfm_hip_sync(ctx) != SFM_HIP_OK) fail(ctx, "sfm_hip_sync");
    for (size_t k = 0; k < n; k++) {
      PairResult& r = results[b0 + k];
      r.lhs.assign(out_l + k * stride, out_l + k * stride + counts[k]);
      r.rhs.assign(out_r + k * stride, out_r + k * stride + counts[k]);
    }
  }
}

}  // namespace

void HipBruteForceMatcher::Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid) {
  // brute_force_matcher.cc:10-12: every pair starts from an empty list
  for (const Pair& pair : pairs) sfm_data.matches[pair] = Matches();
  if (pairs.empty()) return;
  // brute_force_matcher.cc:47-50 reads through map::operator[], which creates empty entries
  for (const Pair& pair : pairs) {
    for (IndexT id : {pair.first, pair.second}) {
      (void)sfm_data.descriptors[id];
      (void)sfm_data.key_points[id];
    }
  }

  int visible = 0;
  if (sfm_hip_device_count(&visible) != SFM_HIP_OK || visible <= 0) fail(nullptr, "no HIP device");
  int want = n_devices_;
  if (want <= 0) {
    const char* env = std::getenv("SFM_HIP_DEVICES");
    want = env ? std::atoi(env) : visible;
    if (want <= 0) want = visible;
  }
  const std::vector<Pair> list(pairs.begin(), pairs.end());   // std::set order: lexicographic
  const int n_dev = (int)std::min<size_t>((size_t)std::min(want, visible), list.size());

  // static contiguous shards, sizes differ by at most one
  std::vector<PairResult> results(list.size());
  std::vector<std::thread> workers;
  std::vector<std::exception_ptr> errors((size_t)n_dev);
  const size_t base = list.size() / n_dev, extra = list.size() % n_dev;
  size_t lo = 0;
  for (int d = 0; d < n_dev; d++) {
    const size_t hi = lo + base + ((size_t)d < extra ? 1 : 0);
    workers.emplace_back([&, d, lo, hi] {
      try {
        run_device(d, list, lo, hi, sfm_data, ratio_, is_cross_valid, results);
      } catch (...) {
        errors[(size_t)d] = std::current_exception();
      }
    });
    lo = hi;
  }
  for (auto& t : workers) t.join();
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);

  // brute_force_matcher.cc:36-37, 80-85: build the Match records on the host (observations never
  // leave it) and report progress per pair as the reference does
  int count = 0;
  const int total = (int)list.size();
  for (size_t p = 0; p < list.size(); p++) {
    const std::vector<KeyPoint>& kl = sfm_data.key_points.at(list[p].first);
    const std::vector<KeyPoint>& kr = sfm_data.key_points.at(list[p].second);
    Matches& out = sfm_data.matches[list[p]];
    out.reserve(results[p].lhs.size());
    for (size_t m = 0; m < results[p].lhs.size(); m++) {
      const IndexT i = results[p].lhs[m], j = results[p].rhs[m];
      // 'struct': inside this member function the bare name means the function
      struct Match rec{ToObservation(kl.at((size_t)i)), ToObservation(kr.at((size_t)j)), i, j};
      out.push_back(rec);
    }
    if (print_progress_) std::printf("Finish %f %%\n", 100.0 * ++count / total);
  }
}
