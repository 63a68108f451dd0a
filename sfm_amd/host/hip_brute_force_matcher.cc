#include "hip_brute_force_matcher.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "sfm_hip.h"

namespace {

// Everything a worker needs for one pair, looked up once on the calling thread so that workers
// never touch the maps' structure.
struct PairJob {
  Pair pair;
  const Descriptors* first;
  const Descriptors* second;
  const std::vector<KeyPoint>* kl;
  const std::vector<KeyPoint>* kr;
};

// What the helper threads did, summed over all of them (each adds its own share once per batch).
struct HostWork {
  std::mutex mu;
  HipBruteForceMatcher::HostStats total;
  void add(uint64_t pairs, uint64_t records, double seconds) {
    std::lock_guard<std::mutex> lock(mu);
    total.pairs += pairs;
    total.records += records;
    total.materialise_thread_seconds += seconds;
  }
};

struct Progress {
  std::mutex mu;
  int done = 0;
  int total = 0;
  bool print = true;
  // brute_force_matcher.cc:85: one line per finished pair
  void pair_finished() {
    if (!print) return;
    std::lock_guard<std::mutex> lock(mu);
    std::printf("Finish %f %%\n", 100.0 * ++done / total);
  }
};

[[noreturn]] void fail(sfm_hip_ctx* ctx, const char* what) {
  std::string msg = std::string("HipBruteForceMatcher: ") + what + ": " + sfm_hip_last_error(ctx);
  throw std::runtime_error(msg);
}

// brute_force_matcher.cc:36-37: the Match records are built on the host, observations never
// leave it.
void materialise(const PairJob& job, const int32_t* lhs, const int32_t* rhs, int32_t n, Matches& out) {
  out.clear();
  out.reserve((size_t)n);
  for (int32_t m = 0; m < n; m++) {
    const IndexT i = lhs[m], j = rhs[m];
    // 'struct': inside the matcher's translation unit the bare name also names a member function
    struct Match rec{ToObservation(job.kl->at((size_t)i)), ToObservation(job.kr->at((size_t)j)), i, j};
    out.push_back(rec);
  }
}

// One GPU: upload the images its pairs touch, then match its shard of the pair list (indices into
// jobs) in batches.
// Two sets of pinned result buffers alternate, so the GPU works on batch b (with batch b+1 queued
// behind it) while this thread turns batch b-1 into Match records.
void run_device(int device, const std::vector<PairJob>& jobs, const std::vector<size_t>& shard, double ratio, bool cross,
                int engine, Progress& progress, int helpers, sfm_hip_ctx* resident_ctx,
                const HipBruteForceMatcher::MatchSink& sink, HostWork& work) {
  // resident_ctx: a context that already holds every image (replication by collective); this worker
  // takes it over.  Otherwise the worker makes its own and uploads what its shard names.
  sfm_hip_ctx* ctx = resident_ctx;
  if (!ctx) {
    if (shard.empty()) return;
    if (sfm_hip_ctx_create(device, &ctx) != SFM_HIP_OK) fail(nullptr, "sfm_hip_ctx_create");
  }
  struct Guard {
    sfm_hip_ctx* c;
    std::vector<void*> pinned;
    ~Guard() {
      sfm_hip_sync(c);   // nothing may still be writing into the pinned buffers
      for (void* p : pinned) sfm_hip_host_free(p);
      sfm_hip_ctx_destroy(c);
    }
  } guard{ctx, {}};
  if (shard.empty()) return;
  if (engine >= 0 && sfm_hip_set_engine(ctx, engine) != SFM_HIP_OK) fail(ctx, "sfm_hip_set_engine");

  // Work through the shard in the order of each pair's larger image id, not in list order: the
  // pairs of the list's first image need every other image at once, whereas in this order the k-th
  // new image brings k - 1 new pairs, so from a handful of images on the GPU has more to do than
  // the uploads take.  (Where a pair's result lands does not depend on when it is computed.)
  std::vector<size_t> order(shard);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    const IndexT ma = std::max(jobs[a].pair.first, jobs[a].pair.second);
    const IndexT mb = std::max(jobs[b].pair.first, jobs[b].pair.second);
    return ma < mb;
  });
  // An image is uploaded right before the first batch that needs it (below), while the GPU is
  // busy with the batch before.
  std::set<IndexT> uploaded;
  int64_t stride = 1;
  for (size_t p : shard) stride = std::max<int64_t>(stride, (int64_t)jobs[p].first->size());
  auto upload_for = [&](size_t b0, size_t b1) {     // positions in 'order'
    if (resident_ctx) return;
    for (size_t q = b0; q < b1; q++) {
      const size_t p = order[q];
      const IndexT ids[2] = {jobs[p].pair.first, jobs[p].pair.second};
      const Descriptors* ds[2] = {jobs[p].first, jobs[p].second};
      for (int k = 0; k < 2; k++) {
        if (!uploaded.insert(ids[k]).second) continue;
        // queued, not waited for (the descriptors live in sfm_data for the whole call): the uploads of a
        // batch's new images overlap each other, and the library waits for them when the batch is queued
        if (sfm_hip_upload_image_async(ctx, ids[k], ds[k]->data(), (int32_t)ds[k]->size()) != SFM_HIP_OK)
          fail(ctx, "sfm_hip_upload_image_async");
      }
    }
  };

  // batches bound the pinned result buffers (two arrays of stride ints per pair, two sets)
  const size_t batch = std::max<size_t>(1, std::min<size_t>(64, ((size_t)32 << 20) / ((size_t)stride * 8)));
  auto pin = [&](size_t bytes) {
    void* p = sfm_hip_host_alloc(bytes);
    if (!p) throw std::runtime_error("HipBruteForceMatcher: pinned host allocation failed");
    guard.pinned.push_back(p);
    return p;
  };
  struct Set {
    int32_t *counts, *lhs, *rhs;
  } sets[2];
  for (Set& s : sets) {
    s.counts = (int32_t*)pin(batch * sizeof(int32_t));
    s.lhs = (int32_t*)pin(batch * (size_t)stride * sizeof(int32_t));
    s.rhs = (int32_t*)pin(batch * (size_t)stride * sizeof(int32_t));
  }
  std::vector<int64_t> flat(2 * batch);

  const size_t n_jobs = order.size();
  auto submit = [&](size_t b0, const Set& s) {      // b0: position in 'order'
    const size_t n = std::min(batch, n_jobs - b0);
    for (size_t k = 0; k < n; k++) {
      flat[2 * k] = jobs[order[b0 + k]].pair.first;
      flat[2 * k + 1] = jobs[order[b0 + k]].pair.second;
    }
    if (sfm_hip_match_pairs(ctx, flat.data(), (int32_t)n, ratio, cross ? 1 : 0, s.counts, s.lhs, s.rhs, nullptr,
                            stride) != SFM_HIP_OK)
      fail(ctx, "sfm_hip_match_pairs");
  };
  // Building the records is the bulk of the host's work and every pair's list is independent: a
  // few helper threads share a batch's pairs.
  auto collect = [&](size_t b0, const Set& s) {
    const size_t n = std::min(batch, n_jobs - b0);
    const size_t n_helpers = std::min<size_t>((size_t)helpers, n);
    auto part = [&](size_t t) {
      const auto t0 = std::chrono::steady_clock::now();
      uint64_t pairs_done = 0, records = 0;
      for (size_t k = t; k < n; k += n_helpers) {
        const PairJob& job = jobs[order[b0 + k]];
        Matches list;
        materialise(job, s.lhs + k * stride, s.rhs + k * stride, s.counts[k], list);
        records += list.size();
        pairs_done++;
        sink(job.pair, std::move(list));   // the list is the sink's now: kept, forwarded or dropped
        progress.pair_finished();
      }
      work.add(pairs_done, records, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
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

  // The GPU always has the next batch queued behind the one it is working on: batch b + 1 is
  // submitted (into the set of buffers that batch b - 1 has been collected from) before this thread
  // waits for batch b alone and turns it into records.
  size_t which = 0;
  size_t prev = 0;
  upload_for(0, std::min(batch, n_jobs));
  submit(0, sets[which]);
  for (size_t b0 = batch; b0 < n_jobs; b0 += batch) {
    upload_for(b0, std::min(b0 + batch, n_jobs));                     // while batch 'prev' runs
    submit(b0, sets[which ^ 1]);
    if (sfm_hip_wait_outputs(ctx, sets[which].counts, batch * sizeof(int32_t)) != SFM_HIP_OK) fail(ctx, "sfm_hip_wait_outputs");
    collect(prev, sets[which]);
    prev = b0;
    which ^= 1;
  }
  if (sfm_hip_sync(ctx) != SFM_HIP_OK) fail(ctx, "sfm_hip_sync");
  collect(prev, sets[which]);
}

}  // namespace

void HipBruteForceMatcher::Match(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid) {
  // brute_force_matcher.cc:10-12: every pair starts from an empty list
  for (const Pair& pair : pairs) sfm_data.matches[pair] = Matches();
  // brute_force_matcher.cc:80,83: the finished list becomes sfm_data.matches[pair].  The map's
  // entries exist already (above), so helper threads never change the map's structure and each
  // writes a different entry.
  std::map<Pair, Matches*> slot;
  for (const Pair& pair : pairs) slot.emplace(pair, &sfm_data.matches.at(pair));
  MatchStream(sfm_data, pairs, is_cross_valid, [&slot](const Pair& pair, Matches&& list) { *slot.at(pair) = std::move(list); });
}

void HipBruteForceMatcher::MatchStream(SfMData& sfm_data, const std::set<Pair>& pairs, bool is_cross_valid,
                                       const MatchSink& sink) {
  host_stats_ = HostStats{};
  if (!sink) throw std::invalid_argument("HipBruteForceMatcher::MatchStream: empty sink");
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
  int engine = engine_;
  if (engine < 0) {
    if (const char* env = std::getenv("SFM_HIP_ENGINE")) {
      const std::string name(env);
      if (name == "f32") engine = SFM_HIP_ENGINE_F32;
      else if (name == "f16x3") engine = SFM_HIP_ENGINE_F16X3;
      else throw std::runtime_error("HipBruteForceMatcher: SFM_HIP_ENGINE must be f32 or f16x3, got '" + name + "'");
    }
  }
  // std::set order: lexicographic.  The maps are not modified from here on, so the pointers
  // taken now stay valid while the workers run, and each worker writes only its own pairs' lists.
  std::vector<PairJob> jobs;
  jobs.reserve(pairs.size());
  for (const Pair& pair : pairs)
    jobs.push_back(PairJob{pair, &sfm_data.descriptors.at(pair.first), &sfm_data.descriptors.at(pair.second),
                           &sfm_data.key_points.at(pair.first), &sfm_data.key_points.at(pair.second)});
  // SFM_HIP_OVERSUBSCRIBE=1 lets more shards than GPUs run (shard d on GPU d modulo the number
  // of GPUs, each with its own context and host thread): a way to exercise the multi-GPU path on
  // a machine with fewer GPUs, not a way to go faster.
  const char* over = std::getenv("SFM_HIP_OVERSUBSCRIBE");
  const bool oversubscribe = over && std::string(over) == "1";
  const int n_dev = (int)std::min<size_t>((size_t)(oversubscribe ? want : std::min(want, visible)), jobs.size());

  // host threads that turn index lists into Match records, per GPU
  const unsigned cores = std::thread::hardware_concurrency();
  int helpers = (int)std::max(1u, std::min(8u, (cores ? cores : 8u) / (unsigned)n_dev));
  if (helper_threads_ > 0) helpers = helper_threads_;
  else if (const char* env = std::getenv("SFM_HIP_HELPERS")) helpers = std::max(1, std::atoi(env));
  // Static shards: compact 2-D blocks of the (first, second) triangle of equal cost, where a pair
  // costs rows(first) x rows(second) (sfm_hip_partition_pairs); a GPU uploads only its block's images.
  std::vector<std::vector<size_t>> shards((size_t)n_dev);
  {
    std::vector<int64_t> flat(2 * jobs.size()), ids;
    std::vector<int32_t> rows;
    for (size_t k = 0; k < jobs.size(); k++) {
      flat[2 * k] = jobs[k].pair.first;
      flat[2 * k + 1] = jobs[k].pair.second;
    }
    std::set<IndexT> named;
    for (const PairJob& j : jobs) {
      if (named.insert(j.pair.first).second) {
        ids.push_back(j.pair.first);
        rows.push_back((int32_t)j.first->size());
      }
      if (named.insert(j.pair.second).second) {
        ids.push_back(j.pair.second);
        rows.push_back((int32_t)j.second->size());
      }
    }
    std::vector<int32_t> order(jobs.size()), begin((size_t)n_dev + 1);
    if (sfm_hip_partition_pairs(flat.data(), (int32_t)jobs.size(), ids.data(), rows.data(), (int32_t)ids.size(), n_dev,
                                order.data(), begin.data()) != SFM_HIP_OK)
      throw std::runtime_error("HipBruteForceMatcher: sfm_hip_partition_pairs refused the pair list");
    for (int d = 0; d < n_dev; d++)
      shards[(size_t)d].assign(order.begin() + begin[(size_t)d], order.begin() + begin[(size_t)d + 1]);
  }
  // Replication by collective: contexts are made here, every image goes to one GPU over PCIe and to
  // the others in one RCCL all-gather, and the workers start with everything resident.
  int replication = replication_;
  if (replication < 0) {
    replication = 0;
    if (const char* env = std::getenv("SFM_HIP_REPLICATION")) {
      const std::string name(env);
      if (name == "collective") replication = 1;
      else if (name != "host")
        throw std::runtime_error("HipBruteForceMatcher: SFM_HIP_REPLICATION must be host or collective, got '" + name + "'");
    }
  }
  std::vector<sfm_hip_ctx*> resident((size_t)n_dev, nullptr);
  if (replication == 1) {
    if (n_dev > visible) throw std::runtime_error("HipBruteForceMatcher: replication by collective needs one GPU per shard");
    auto destroy_all = [&] {
      for (sfm_hip_ctx* c : resident) sfm_hip_ctx_destroy(c);
    };
    for (int d = 0; d < n_dev; d++)
      if (sfm_hip_ctx_create(d, &resident[(size_t)d]) != SFM_HIP_OK) {
        destroy_all();
        fail(nullptr, "sfm_hip_ctx_create");
      }
    std::vector<int64_t> ids;
    std::vector<const float*> data;
    std::vector<int32_t> rows;
    std::set<IndexT> named;
    for (const PairJob& j : jobs) {
      const IndexT two[2] = {j.pair.first, j.pair.second};
      const Descriptors* ds[2] = {j.first, j.second};
      for (int k = 0; k < 2; k++)
        if (named.insert(two[k]).second) {
          ids.push_back(two[k]);
          data.push_back(ds[k]->data());
          rows.push_back((int32_t)ds[k]->size());
        }
    }
    if (sfm_hip_replicate_images(resident.data(), n_dev, ids.data(), data.data(), rows.data(), (int32_t)ids.size(), nullptr) !=
        SFM_HIP_OK) {
      const std::string msg = std::string("HipBruteForceMatcher: sfm_hip_replicate_images: ") + sfm_hip_last_error(resident[0]);
      destroy_all();
      throw std::runtime_error(msg);
    }
  }
  Progress progress;
  progress.total = (int)jobs.size();
  progress.print = print_progress_;
  HostWork work;
  std::vector<std::thread> workers;
  std::vector<std::exception_ptr> errors((size_t)n_dev);
  for (int d = 0; d < n_dev; d++) {
    workers.emplace_back([&, d] {
      try {
        run_device(d % visible, jobs, shards[(size_t)d], ratio_, is_cross_valid, engine, progress, helpers,
                   resident[(size_t)d], sink, work);
      } catch (...) {
        errors[(size_t)d] = std::current_exception();
      }
    });
  }
  for (auto& t : workers) t.join();
  host_stats_ = work.total;
  host_stats_.gpus = n_dev;
  host_stats_.helper_threads_per_gpu = helpers;
  for (auto& e : errors)
    if (e) std::rethrow_exception(e);
}
