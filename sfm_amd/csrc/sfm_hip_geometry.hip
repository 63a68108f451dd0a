// Host side of sfm_hip_geometry_verify_pairs (include/sfm_hip.h): uploads the matches' pixel
// positions, runs the RANSAC rounds in launches of 256 until every pair has stopped, classifies.
#include "ctx.hpp"

#include <algorithm>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "geometry_kernels.hpp"

using namespace sfm_geometry;
using namespace sfm_host;

namespace {

// Device memory a call may spend on the models of one launch (27,648 bytes per pair and run of 256 rounds)
constexpr size_t HYP_BYTES_BUDGET = (size_t)1 << 30;

// One of the context's grow-only buffers, at least `bytes` large.  The call is blocking and runs on
// one stream, so nothing queued earlier can still be using the old allocation.
int grown(sfm_hip_ctx* ctx, int which, size_t bytes, void** out) {
  DevBuf& b = ctx->geometry_buffers[which];
  if (b.bytes < bytes) {
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
  }
  *out = b.p;
  return SFM_HIP_OK;
}

// ransac.hpp:128-131 for every possible inlier count of a pair with n matches: how many rounds
// are enough once a model with that many inliers has been seen, never more than the round limit.
// Computed here, with the host's math library, because it is the same library the reference's
// std::log / std::pow resolve to: the stopping round is then the reference's to the unit.
void fill_rounds_needed(size_t n, uint32_t* table) {
  for (size_t c = 0; c <= n; c++) {
    const double epsilon = (n - c) * 1.0 / n;
    const double t = std::ceil(std::log(1 - 0.99) / std::log(1 - std::pow(1 - epsilon, SAMPLE)));
    // few inliers: w^8 vanishes against 1, the quotient is -infinity, and the reference's
    // conversion of that to size_t leaves its N as it is
    table[c] = (t >= 0.0 && t < (double)MAX_ROUNDS) ? (uint32_t)t : MAX_ROUNDS;
  }
}

}  // namespace

extern "C" {

int sfm_hip_geometry_verify_pairs(sfm_hip_ctx* ctx, int32_t n_pairs, const int64_t* match_offsets, const double* lhs_xy,
                                  const double* rhs_xy, uint64_t seed, int64_t first_pair_index, double* out_F,
                                  uint8_t* out_inlier_flags, int32_t* out_inlier_counts, int32_t* out_rounds) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n_pairs < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative pair count");
  if (n_pairs == 0) return SFM_HIP_OK;
  if (!match_offsets || !out_F || !out_inlier_counts) return fail(ctx, SFM_HIP_ERR_INVALID, "null offsets or output buffer");
  if (match_offsets[0] != 0) return fail(ctx, SFM_HIP_ERR_INVALID, "match_offsets must start at 0");
  for (int32_t p = 0; p < n_pairs; p++)
    if (match_offsets[p + 1] < match_offsets[p] || match_offsets[p + 1] - match_offsets[p] > 0x7FFFFFFF)
      return fail(ctx, SFM_HIP_ERR_INVALID, "match_offsets must ascend by less than 2^31 per pair");
  const int64_t total = match_offsets[n_pairs];
  if (total > 0 && (!lhs_xy || !rhs_xy || !out_inlier_flags))
    return fail(ctx, SFM_HIP_ERR_INVALID, "null positions or flag buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->upload_stream;

  // pairs with fewer than eight matches cannot be sampled (ransac.hpp:38-42): nothing is computed
  std::vector<int32_t> live;
  for (int32_t p = 0; p < n_pairs; p++) {
    out_inlier_counts[p] = 0;
    if (out_rounds) out_rounds[p] = 0;
    for (int k = 0; k < 9; k++) out_F[(size_t)p * 9 + k] = 0.0;
    if (match_offsets[p + 1] - match_offsets[p] >= SAMPLE) live.push_back(p);
  }
  if (total > 0) std::fill(out_inlier_flags, out_inlier_flags + total, (uint8_t)0);
  if (live.empty()) return SFM_HIP_OK;
  const int n_live = (int)live.size();

  // one table of rounds needed per distinct match count
  std::map<int64_t, size_t> table_at;
  size_t table_words = 0;
  for (int32_t p : live) {
    const int64_t n = match_offsets[p + 1] - match_offsets[p];
    if (table_at.emplace(n, table_words).second) table_words += (size_t)n + 1;
  }
  std::vector<uint32_t> tables(table_words);
  for (const auto& kv : table_at) fill_rounds_needed((size_t)kv.first, tables.data() + kv.second);

  struct {
    void* p;
  } d_lhs, d_rhs, d_tables, d_state, d_hyp_F, d_hyp_idx, d_hyp_count, d_flags, d_counts, d_pairs, d_unfinished;
  int rc;
  if ((rc = grown(ctx, 0, (size_t)total * 16, &d_lhs.p))) return rc;
  if ((rc = grown(ctx, 1, (size_t)total * 16, &d_rhs.p))) return rc;
  if ((rc = grown(ctx, 2, tables.size() * 4, &d_tables.p))) return rc;
  if ((rc = grown(ctx, 3, (size_t)n_live * sizeof(GeoState), &d_state.p))) return rc;
  // A launch evaluates 256 rounds of every unfinished pair, or several runs of 256 ("chunks") once most
  // pairs have finished -- a launch takes a round's serial time whatever the number of rounds in it,
  // so the pairs that need many rounds get them in few launches: 1, 2, 4, 8 runs, then one at a time.
  // What a launch may hold is bounded by the memory of its models: 27,648 bytes per pair and run.
  const size_t run_bytes = (size_t)ROUNDS_PER_LAUNCH * (9 * sizeof(double) + SAMPLE * 4 + 4);
  const int max_chunks = (int)std::max<size_t>(1, std::min<size_t>(8, HYP_BYTES_BUDGET / ((size_t)n_live * run_bytes)));
  const size_t slots = (size_t)max_chunks * ROUNDS_PER_LAUNCH;   // per pair
  if ((rc = grown(ctx, 4, (size_t)n_live * slots * 9 * sizeof(double), &d_hyp_F.p))) return rc;
  if ((rc = grown(ctx, 5, (size_t)n_live * slots * SAMPLE * 4, &d_hyp_idx.p))) return rc;
  if ((rc = grown(ctx, 6, (size_t)n_live * slots * 4, &d_hyp_count.p))) return rc;
  if ((rc = grown(ctx, 7, (size_t)total, &d_flags.p))) return rc;
  if ((rc = grown(ctx, 8, (size_t)n_live * 4, &d_counts.p))) return rc;
  if ((rc = grown(ctx, 9, (size_t)n_live * sizeof(GeoPair), &d_pairs.p))) return rc;
  // one "unfinished" counter per launch, then the two work counters of the statistics
  if ((rc = grown(ctx, 10, 4 * 16 + 2 * sizeof(unsigned long long), &d_unfinished.p))) return rc;
  unsigned long long* d_work = ctx->profiling ? (unsigned long long*)((char*)d_unfinished.p + 4 * 16) : nullptr;
  HIP_TRY(ctx, hipMemcpyAsync(d_lhs.p, lhs_xy, (size_t)total * 16, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_rhs.p, rhs_xy, (size_t)total * 16, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_tables.p, tables.data(), tables.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(d_flags.p, 0, (size_t)total, st));
  HIP_TRY(ctx, hipMemsetAsync(d_counts.p, 0, (size_t)n_live * 4, st));

  std::vector<GeoState> states((size_t)n_live);
  std::vector<GeoPair> pairs((size_t)n_live);
  int max_n = 0;
  for (int k = 0; k < n_live; k++) {
    const int32_t p = live[(size_t)k];
    const int64_t at = match_offsets[p], n = match_offsets[p + 1] - at;
    GeoState& s = states[(size_t)k];
    for (double& v : s.best_F) v = 0.0;
    s.rounds_limit = MAX_ROUNDS;
    s.rounds_done = 0;
    s.best_count = 0;
    s.done = 0;
    GeoPair& g = pairs[(size_t)k];
    g.lhs = (const double*)d_lhs.p + 2 * at;
    g.rhs = (const double*)d_rhs.p + 2 * at;
    g.rounds_needed = (const uint32_t*)d_tables.p + table_at.at(n);
    g.state = (GeoState*)d_state.p + k;
    g.hyp_F = (double*)d_hyp_F.p + (size_t)k * slots * 9;
    g.hyp_idx = (int32_t*)d_hyp_idx.p + (size_t)k * slots * SAMPLE;
    g.hyp_count = (uint32_t*)d_hyp_count.p + (size_t)k * slots;
    g.flags = (uint8_t*)d_flags.p + at;
    g.n_inliers = (uint32_t*)d_counts.p + k;
    g.n = (int32_t)n;
    g.pad_ = 0;
    g.stream = pair_stream(seed, (uint64_t)(first_pair_index + p));
    max_n = std::max(max_n, (int)n);
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_state.p, states.data(), states.size() * sizeof(GeoState), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_pairs.p, pairs.data(), pairs.size() * sizeof(GeoPair), hipMemcpyHostToDevice, st));

  const GeoPair* dp = (const GeoPair*)d_pairs.p;
  // Launch k + 1 is queued before launch k's "unfinished" counter has been read: when pairs are
  // left it starts without waiting for the host, and when none are its kernels return at once.
  constexpr int MAX_LAUNCHES = MAX_ROUNDS / ROUNDS_PER_LAUNCH;
  static_assert(MAX_LAUNCHES <= 16, "one event and one counter per launch");
  if (!ctx->geometry_readback) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->geometry_readback, hipStreamNonBlocking));
  for (int k = 0; k < MAX_LAUNCHES; k++)
    if (!ctx->geometry_events[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->geometry_events[k], hipEventDisableTiming));
  HIP_TRY(ctx, hipMemsetAsync(d_unfinished.p, 0, 4 * MAX_LAUNCHES + 2 * sizeof(unsigned long long), st));
  // profiling: three events per launch (in front of the fit, between fit and count, behind the count)
  struct Marks {   // back into the context's pool however the call ends
    sfm_hip_ctx* ctx;
    std::vector<hipEvent_t> v;
    ~Marks() { for (hipEvent_t e : v) ctx->event_pool.push_back(e); }
  } held{ctx, {}};
  std::vector<hipEvent_t>& marks = held.v;
  auto mark = [&]() -> int {
    if (!ctx->profiling) return SFM_HIP_OK;
    hipEvent_t e = nullptr;
    if (!ctx->event_pool.empty()) {
      e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
    } else {
      HIP_TRY(ctx, hipEventCreate(&e));
    }
    marks.push_back(e);
    HIP_TRY(ctx, hipEventRecord(e, st));
    return SFM_HIP_OK;
  };
  constexpr int TOTAL_CHUNKS = MAX_ROUNDS / ROUNDS_PER_LAUNCH;
  int chunks_queued = 0;   // runs of 256 rounds in the launches queued so far
  auto queue_launch = [&](int k) -> int {
    static const int schedule[4] = {1, 2, 4, 8};
    const int chunks = std::min(std::min(k < 4 ? schedule[k] : 1, max_chunks), TOTAL_CHUNKS - chunks_queued);
    const uint32_t first = (uint32_t)chunks_queued * ROUNDS_PER_LAUNCH;
    chunks_queued += chunks;
    int mrc;
    if ((mrc = mark())) return mrc;
    const dim3 per_round((unsigned)(chunks * (ROUNDS_PER_LAUNCH / HYP_THREADS)), (unsigned)n_live);
    hipLaunchKernelGGL(k_geo_sample, per_round, dim3(HYP_THREADS), 0, st, dp, first);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_geo_fit, per_round, dim3(HYP_THREADS), 0, st, dp, first);
    HIP_TRY(ctx, hipGetLastError());
    if ((mrc = mark())) return mrc;
    hipLaunchKernelGGL(k_geo_count, dim3((unsigned)n_live, (unsigned)chunks), dim3(ROUNDS_PER_LAUNCH), 0, st, dp, first);
    HIP_TRY(ctx, hipGetLastError());
    if ((mrc = mark())) return mrc;
    hipLaunchKernelGGL(k_geo_bookkeeping, dim3((unsigned)n_live), dim3(64), 0, st, dp, n_live, chunks, (uint32_t*)d_unfinished.p + k,
                       d_work);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->geometry_events[k], st));
    return SFM_HIP_OK;
  };
  if ((rc = queue_launch(0))) return rc;
  for (int k = 0; k < MAX_LAUNCHES; k++) {
    const bool more = chunks_queued < TOTAL_CHUNKS;
    if (more && (rc = queue_launch(k + 1))) return rc;
    uint32_t unfinished = 0;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->geometry_readback, ctx->geometry_events[k], 0));
    HIP_TRY(ctx, hipMemcpyAsync(&unfinished, (uint32_t*)d_unfinished.p + k, 4, hipMemcpyDeviceToHost, ctx->geometry_readback));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->geometry_readback));
    if (unfinished == 0 || !more) break;
  }
  hipLaunchKernelGGL(k_geo_classify, dim3((unsigned)((max_n + 255) / 256), (unsigned)n_live), dim3(256), 0, st, dp);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<uint32_t> counts((size_t)n_live);
  HIP_TRY(ctx, hipMemcpyAsync(states.data(), d_state.p, states.size() * sizeof(GeoState), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(counts.data(), d_counts.p, counts.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(out_inlier_flags, d_flags.p, (size_t)total, hipMemcpyDeviceToHost, st));
  unsigned long long work[2] = {0, 0};
  if (d_work) HIP_TRY(ctx, hipMemcpyAsync(work, d_work, sizeof(work), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->geometry_stats.calls += 1;
  ctx->geometry_stats.pairs += (uint64_t)n_live;
  ctx->geometry_stats.rounds += work[0];
  ctx->geometry_stats.match_rounds += work[1];
  for (size_t k = 0; k + 2 < marks.size(); k += 3) {
    float fit = 0.f, count = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&fit, marks[k], marks[k + 1]));
    HIP_TRY(ctx, hipEventElapsedTime(&count, marks[k + 1], marks[k + 2]));
    ctx->geometry_stats.fit_ms += fit;
    ctx->geometry_stats.count_ms += count;
    ctx->geometry_stats.timed_launches += 1;
  }
  for (int k = 0; k < n_live; k++) {
    const int32_t p = live[(size_t)k];
    for (int c = 0; c < 9; c++) out_F[(size_t)p * 9 + c] = states[(size_t)k].best_F[c];
    out_inlier_counts[p] = (int32_t)counts[(size_t)k];
    if (out_rounds) out_rounds[p] = (int32_t)states[(size_t)k].rounds_done;
  }
  return SFM_HIP_OK;
}

int sfm_hip_geometry_get_stats(sfm_hip_ctx* ctx, sfm_hip_geometry_stats* out) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!out) return fail(ctx, SFM_HIP_ERR_INVALID, "out is null");
  *out = ctx->geometry_stats;   // the verification is a blocking call: nothing of it is ever pending
  return SFM_HIP_OK;
}

int sfm_hip_geometry_reset_stats(sfm_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  ctx->geometry_stats = sfm_hip_geometry_stats{};
  return SFM_HIP_OK;
}

}  // extern "C"
