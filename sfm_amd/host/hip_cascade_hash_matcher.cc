#include "hip_cascade_hash_matcher.hpp"

#include <algorithm>
#include <cstdlib>
#include <exception>
#include <iostream>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "sfm_hip.h"

namespace {

struct PairJob {
  Pair pair;
  const std::vector<KeyPoint>* kl;
  const std::vector<KeyPoint>* kr;
  size_t n_first;
  Matches result;
};

[[noreturn]] void fail(sfm_hip_ctx* ctx, const char* what) {
  throw std::runtime_error(std::string("HipCascadeHashMatcher: ") + what + ": " + sfm_hip_last_error(ctx));
}

// cascade_hash_matcher.cc:187-202: the default records first, then the survivors; every record's
// observations come from the key points its indices name (index 0 for the default records).
void materialise(PairJob& job, int32_t n_default, const int32_t* lhs, const int32_t* rhs, int32_t n) {
  Matches& out = job.result;
  out.clear();
  out.reserve((size_t)n_default + (size_t)n);
  for (int32_t m = 0; m < n_default + n; m++) {
    const IndexT i = m < n_default ? 0 : lhs[m - n_default];
    const IndexT j = m < n_default ? 0 : rhs[m - n_default];
    struct Match rec;
    rec.lhs_idx = i;
    rec.rhs_idx = j;
    rec.lhs_observation = ToObservation(job.kl->at((size_t)i));
    rec.rhs_observation = ToObservation(job.kr->at((size_t)j));
    out.push_back(rec);
  }
}

// One GPU: all images of the job (the mean vector is taken over all of them, and any pair may
// need any image), hashes, then this GPU's slice of the pair list in batches.  Two sets of pinned
// result buffers alternate: the GPU matches batch b while this thread builds batch b-1's records.
void run_device(int device, const SfMData& scene, const std::vector<int64_t>& flat_pairs, std::vector<PairJob>& jobs,
                size_t lo, size_t hi, bool cross, int helpers) {
  sfm_hip_ctx* ctx = nullptr;
  if (sfm_hip_ctx_create(device, &ctx) != SFM_HIP_OK) fail(nullptr, "sfm_hip_ctx_create");
  struct Guard {
    sfm_hip_ctx* c;
    std::vector<void*> pinned;
    ~Guard() {
      sfm_hip_sync(c);
      for (void* p : pinned) sfm_hip_host_free(p);
      sfm_hip_ctx_destroy(c);
    }
  } guard{ctx, {}};

  std::set<IndexT> ids;
  for (size_t k = 0; k < flat_pairs.size(); k++) ids.insert(flat_pairs[k]);
  for (IndexT id : ids) {
    const Descriptors& d = scene.descriptors.at(id);
    if (sfm_hip_upload_image_async(ctx, id, d.data(), (int32_t)d.size()) != SFM_HIP_OK) fail(ctx, "sfm_hip_upload_image_async");
  }
  if (sfm_hip_cascade_prepare(ctx, flat_pairs.data(), (int32_t)(flat_pairs.size() / 2), nullptr) != SFM_HIP_OK)
    fail(ctx, "sfm_hip_cascade_prepare");

  int64_t stride = 1;
  for (size_t p = lo; p < hi; p++) stride = std::max<int64_t>(stride, (int64_t)jobs[p].n_first);
  const size_t batch = std::max<size_t>(1, std::min<size_t>(256, ((size_t)64 << 20) / ((size_t)stride * 8)));
  auto pin = [&](size_t bytes) {
    void* p = sfm_hip_host_alloc(bytes);
    if (!p) throw std::runtime_error("HipCascadeHashMatcher: pinned host allocation failed");
    guard.pinned.push_back(p);
    return p;
  };
  struct Set {
    int32_t *defaults, *counts, *lhs, *rhs;
  } sets[2];
  for (Set& s : sets) {
    s.defaults = (int32_t*)pin(batch * sizeof(int32_t));
    s.counts = (int32_t*)pin(batch * sizeof(int32_t));
    s.lhs = (int32_t*)pin(batch * (size_t)stride * sizeof(int32_t));
    s.rhs = (int32_t*)pin(batch * (size_t)stride * sizeof(int32_t));
  }
  auto submit = [&](size_t b0, const Set& s) {
    const size_t n = std::min(batch, hi - b0);
    if (sfm_hip_cascade_match_pairs(ctx, flat_pairs.data() + 2 * b0, (int32_t)n, cross ? 1 : 0, s.defaults, s.counts,
                                    s.lhs, s.rhs, nullptr, stride) != SFM_HIP_OK)
      fail(ctx, "sfm_hip_cascade_match_pairs");
  };
  // Building the records is the bulk of the host's work (48 bytes each, observations looked up per
  // index) and every pair's list is independent: a few helper threads share a batch's pairs.
  auto collect = [&](size_t b0, const Set& s) {
    const size_t n = std::min(batch, hi - b0);
    const size_t n_helpers = std::min<size_t>((size_t)helpers, n);
    auto part = [&](size_t t) {
      for (size_t k = t; k < n; k += n_helpers)
        materialise(jobs[b0 + k], s.defaults[k], s.lhs + k * stride, s.rhs + k * stride, s.counts[k]);
    };
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> failed(n_helpers);
    for (size_t t = 1; t < n_helpers; t++)
      pool.emplace_back([&, t] {
        try {
          part(t);
        } catch (...) {
          failed[t] = std::current_exception();
        }
      });
    try {
      part(0);
    } catch (...) {
      failed[0] = std::current_exception();
    }
    for (auto& th : pool) th.join();
    for (auto& f : failed)
      if (f) std::rethrow_exception(f);
  };
  size_t which = 0, prev = lo;
  submit(lo, sets[which]);
  for (size_t b0 = lo + batch; b0 < hi; b0 += batch) {
    if (sfm_hip_sync(ctx) != SFM_HIP_OK) fail(ctx, "sfm_hip_sync");
    submit(b0, sets[which ^ 1]);
    collect(prev, sets[which]);
    prev = b0;
    which ^= 1;
  }
  if (sfm_hip_sync(ctx) != SFM_HIP_OK) fail(ctx, "sfm_hip_sync");
  collect(prev, sets[which]);
}

}  // namespace

void HipCascadeHashMatcher::Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid) {
  // cascade_hash_matcher.cc:158,175,198 read through map::operator[], which creates empty entries
  for (const Pair& pair : pairs) {
    for (IndexT id : {pair.first, pair.second}) {
      (void)sfm_data.descriptors[id];
      (void)sfm_data.key_points[id];
    }
  }
  if (print_progress_) std::cout << "mean " << std::endl;                          // .cc:166
  if (pairs.empty()) {
    if (print_progress_) std::cout << "Create Descriptors Finish" << std::endl;   // .cc:171
    return;
  }
  int visible = 0;
  if (sfm_hip_device_count(&visible) != SFM_HIP_OK || visible <= 0) fail(nullptr, "no HIP device");
  int want = n_devices_;
  if (want <= 0) {
    const char* env = std::getenv("SFM_HIP_DEVICES");
    want = env ? std::atoi(env) : visible;
    if (want <= 0) want = visible;
  }
  std::vector<PairJob> jobs;
  std::vector<int64_t> flat;
  jobs.reserve(pairs.size());
  for (const Pair& pair : pairs) {      // std::set order, the order the reference walks (.cc:173)
    jobs.push_back(PairJob{pair, &sfm_data.key_points.at(pair.first), &sfm_data.key_points.at(pair.second),
                           sfm_data.descriptors.at(pair.first).size(), Matches()});
    flat.push_back(pair.first);
    flat.push_back(pair.second);
  }
  const char* over = std::getenv("SFM_HIP_OVERSUBSCRIBE");   // more shards than GPUs, for testing
  const bool oversubscribe = over && std::string(over) == "1";
  const int n_dev = (int)std::min<size_t>((size_t)(oversubscribe ? want : std::min(want, visible)), jobs.size());
  // host threads that turn index lists into Match records, per GPU
  const unsigned cores = std::thread::hardware_concurrency();
  const int helpers = (int)std::max(1u, std::min(8u, (cores ? cores : 8u) / (unsigned)n_dev));
  std::vector<std::thread> workers;
  std::vector<std::exception_ptr> errors((size_t)n_dev);
  const size_t base = jobs.size() / n_dev, extra = jobs.size() % n_dev;
  size_t lo = 0;
  for (int d = 0; d < n_dev; d++) {
    const size_t hi = lo + base + ((size_t)d < extra ? 1 : 0);
    workers.emplace_back([&, d, lo, hi] {
      try {
        run_device(d % visible, sfm_data, flat, jobs, lo, hi, is_cross_valid, helpers);
      } catch (...) {
        errors[(size_t)d] = std::current_exception();
      }
    });
    lo = hi;
  }
  for (auto& t : workers) t.join();
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);
  if (print_progress_) std::cout << "Create Descriptors Finish" << std::endl;
  for (PairJob& job : jobs) {
    if (print_progress_)                                                           // .cc:196
      std::cout << "Piar : [" << job.pair.first << "," << job.pair.second << "] " << job.result.size() << " Matches."
                << std::endl;
    sfm_data.matches[job.pair] = std::move(job.result);                            // .cc:204
  }
}
