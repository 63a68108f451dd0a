// Host side of the cascade-hash entry points of include/sfm_hip.h: projection upload, per-image
// hashing and bucket order, batching of pairs into launches, result copies.
#include "ctx.hpp"

#include <algorithm>
#include <cstring>
#include <random>
#include <string>
#include <unordered_set>
#include <vector>

#include "cascade_kernels.hpp"

using namespace sfm_cascade;
using namespace sfm_host;

namespace {

constexpr size_t PROJ_FLOATS = (size_t)(NPROJ + 1) * DIM;   // projection rows, then the mean
constexpr int MAX_PAIRS_PER_BATCH = 1024;
constexpr size_t SLOT_BYTES_PER_BATCH = (size_t)2 << 30;

struct Scratch {   // a device allocation that lives for one call
  void* p = nullptr;
  ~Scratch() {
    if (p) (void)hipFree(p);
  }
};

int find_images(sfm_hip_ctx* ctx, const int64_t* ids, int32_t n, std::vector<Image*>& out) {
  out.clear();
  for (int32_t k = 0; k < n; k++) {
    auto it = ctx->images.find(ids[k]);
    if (it == ctx->images.end())
      return fail(ctx, SFM_HIP_ERR_INVALID, "image " + std::to_string(ids[k]) + " was not uploaded");
    out.push_back(&it->second);
  }
  return SFM_HIP_OK;
}

int fold_cascade_timing(sfm_hip_ctx* ctx) {
  for (size_t k = 0; k + 1 < ctx->cascade_timing.size(); k += 2) {
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->cascade_timing[k], ctx->cascade_timing[k + 1]));
    ctx->cascade_stats.match_ms += ms;
    ctx->cascade_stats.match_launches += 1;
    ctx->event_pool.push_back(ctx->cascade_timing[k]);
    ctx->event_pool.push_back(ctx->cascade_timing[k + 1]);
  }
  ctx->cascade_timing.clear();
  return SFM_HIP_OK;
}

hipEvent_t take_event(sfm_hip_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

}  // namespace

extern "C" {

int sfm_hip_cascade_set_projection(sfm_hip_ctx* ctx, const float* primary, const float* secondary) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!primary || !secondary) return fail(ctx, SFM_HIP_ERR_INVALID, "null projection matrix");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->cascade_projection) HIP_TRY(ctx, hipMalloc((void**)&ctx->cascade_projection, PROJ_FLOATS * sizeof(float)));
  // device layout: blocks of four rows stored [j][4] (see k_ch_hash)
  std::vector<float> t((size_t)NPROJ * DIM);
  for (int r = 0; r < NPROJ; r++) {
    const float* row = r < CODE_BITS ? primary + (size_t)r * DIM : secondary + (size_t)(r - CODE_BITS) * DIM;
    for (int j = 0; j < DIM; j++) t[((size_t)(r / PROJ_BLOCK) * DIM + j) * PROJ_BLOCK + r % PROJ_BLOCK] = row[j];
  }
  if (int rc = sync_all(ctx)) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->cascade_projection, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  ctx->cascade_has_projection = true;
  return SFM_HIP_OK;
}

int sfm_hip_cascade_column_sums(sfm_hip_ctx* ctx, const int64_t* image_ids, int32_t n_images, float* out_sums) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n_images < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative image count");
  if (n_images == 0) return SFM_HIP_OK;
  if (!image_ids || !out_sums) return fail(ctx, SFM_HIP_ERR_INVALID, "null image list or output buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<Image*> ims;
  if (int rc = find_images(ctx, image_ids, n_images, ims)) return rc;
  std::vector<ImageRef> refs((size_t)n_images);
  for (int32_t k = 0; k < n_images; k++) refs[(size_t)k] = ImageRef{ims[(size_t)k]->d, ims[(size_t)k]->n, 0};
  Scratch table, sums;
  HIP_TRY(ctx, hipMalloc(&table.p, refs.size() * sizeof(ImageRef)));
  HIP_TRY(ctx, hipMalloc(&sums.p, (size_t)n_images * DIM * sizeof(float)));
  hipStream_t st = ctx->upload_stream;
  HIP_TRY(ctx, hipMemcpyAsync(table.p, refs.data(), refs.size() * sizeof(ImageRef), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_ch_colsum, dim3((unsigned)n_images), dim3(DIM), 0, st, (const ImageRef*)table.p, (float*)sums.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out_sums, sums.p, (size_t)n_images * DIM * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return SFM_HIP_OK;
}

int sfm_hip_cascade_hash_images(sfm_hip_ctx* ctx, const int64_t* image_ids, int32_t n_images, const float* mean128) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n_images < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative image count");
  if (n_images == 0) return SFM_HIP_OK;
  if (!image_ids || !mean128) return fail(ctx, SFM_HIP_ERR_INVALID, "null image list or mean vector");
  if (!ctx->cascade_has_projection)
    return fail(ctx, SFM_HIP_ERR_INVALID, "sfm_hip_cascade_set_projection has not been called on this context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<Image*> ims;
  if (int rc = find_images(ctx, image_ids, n_images, ims)) return rc;
  // the same image may be named twice: hash it once
  std::sort(ims.begin(), ims.end());
  ims.erase(std::unique(ims.begin(), ims.end()), ims.end());
  // an image that is hashed again may still be read by queued matching work
  bool rehash = false;
  for (Image* im : ims) rehash = rehash || im->hash_block;
  if (rehash)
    if (int rc = sync_all(ctx)) return rc;
  hipStream_t st = ctx->upload_stream;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->cascade_projection + (size_t)NPROJ * DIM, mean128, DIM * sizeof(float),
                              hipMemcpyHostToDevice, st));
  std::vector<HashJob> jobs;
  int blocks = 0;
  for (Image* im : ims) {
    free_image_hashes(*im);
    const size_t n = (size_t)im->n;
    const size_t recs_bytes = align_up(std::max<size_t>(n, 1) * sizeof(HashRec), 256);
    const size_t grouped_bytes = align_up(std::max<size_t>(n, 1) * GROUPS * sizeof(HashRec), 256);
    const size_t off_bytes = align_up((size_t)GROUPS * (BUCKETS + 1) * sizeof(int32_t), 256);
    hipError_t e = hipMalloc(&im->hash_block, recs_bytes + grouped_bytes + off_bytes);
    if (e != hipSuccess) {
      im->hash_block = nullptr;
      return fail(ctx, SFM_HIP_ERR_NOMEM, std::string("hipMalloc hash records: ") + hipGetErrorString(e));
    }
    im->hash_recs = im->hash_block;
    im->hash_grouped = (char*)im->hash_block + recs_bytes;
    im->hash_offsets = (int32_t*)((char*)im->hash_block + recs_bytes + grouped_bytes);
    HashJob j;
    j.d = im->d;
    j.recs = (HashRec*)im->hash_recs;
    j.grouped = (HashRec*)im->hash_grouped;
    j.offsets = im->hash_offsets;
    j.n = im->n;
    j.first_block = blocks;
    blocks += (im->n + 63) / 64;
    jobs.push_back(j);
  }
  Scratch table;
  HIP_TRY(ctx, hipMalloc(&table.p, jobs.size() * sizeof(HashJob)));
  HIP_TRY(ctx, hipMemcpyAsync(table.p, jobs.data(), jobs.size() * sizeof(HashJob), hipMemcpyHostToDevice, st));
  if (blocks > 0) {
    hipLaunchKernelGGL(k_ch_hash, dim3((unsigned)blocks), dim3(64), 0, st, (const HashJob*)table.p, (int)jobs.size(),
                       (const float*)ctx->cascade_projection);
    HIP_TRY(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(k_ch_group, dim3(GROUPS, (unsigned)jobs.size()), dim3(64), 0, st, (const HashJob*)table.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->cascade_stats.images_hashed += (uint64_t)jobs.size();
  return SFM_HIP_OK;
}

int sfm_hip_cascade_get_hashes(sfm_hip_ctx* ctx, int64_t image_id, uint32_t* codes, uint16_t* bucket_ids) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  auto it = ctx->images.find(image_id);
  if (it == ctx->images.end()) return fail(ctx, SFM_HIP_ERR_INVALID, "unknown image id " + std::to_string(image_id));
  const Image& im = it->second;
  if (!im.hash_block) return fail(ctx, SFM_HIP_ERR_INVALID, "image " + std::to_string(image_id) + " has not been hashed");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<HashRec> recs((size_t)im.n);
  if (im.n > 0) HIP_TRY(ctx, hipMemcpy(recs.data(), im.hash_recs, recs.size() * sizeof(HashRec), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < recs.size(); i++) {
    if (codes) std::memcpy(codes + 4 * i, recs[i].code, 16);
    if (bucket_ids)
      for (int g = 0; g < GROUPS; g++) bucket_ids[GROUPS * i + g] = (uint16_t)((recs[i].bids[g >> 1] >> ((g & 1) * 16)) & 0xFFFFu);
  }
  return SFM_HIP_OK;
}

int sfm_hip_cascade_match_pairs(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, int cross_check,
                                int32_t* out_defaults, int32_t* out_counts, int32_t* out_lhs_idx,
                                int32_t* out_rhs_idx, float* out_dist, int64_t out_stride) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n_pairs < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative pair count");
  if (n_pairs == 0) return SFM_HIP_OK;
  if (!pair_ids || !out_defaults || !out_counts || !out_lhs_idx || !out_rhs_idx)
    return fail(ctx, SFM_HIP_ERR_INVALID, "null pair list or output buffer");
  if (out_stride < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative out_stride");
  HIP_TRY(ctx, hipSetDevice(ctx->device));

  // validate everything before queuing anything
  std::vector<std::pair<const Image*, const Image*>> plan((size_t)n_pairs);
  for (int p = 0; p < n_pairs; p++) {
    auto ix = ctx->images.find(pair_ids[2 * p]);
    auto iy = ctx->images.find(pair_ids[2 * p + 1]);
    if (ix == ctx->images.end() || iy == ctx->images.end())
      return fail(ctx, SFM_HIP_ERR_INVALID,
                  "pair " + std::to_string(p) + " names an image that was not uploaded (" +
                      std::to_string(pair_ids[2 * p]) + ", " + std::to_string(pair_ids[2 * p + 1]) + ")");
    if (!ix->second.hash_block || !iy->second.hash_block)
      return fail(ctx, SFM_HIP_ERR_INVALID, "pair " + std::to_string(p) + " names an image that has not been hashed");
    if (ix->second.n > out_stride)
      return fail(ctx, SFM_HIP_ERR_INVALID, "out_stride smaller than the first image of pair " + std::to_string(p));
    plan[(size_t)p] = {&ix->second, &iy->second};
  }

  int done = 0;
  while (done < n_pairs) {
    int count = 0;
    size_t slot_bytes = 0;
    while (done + count < n_pairs && count < MAX_PAIRS_PER_BATCH) {
      const auto& pl = plan[(size_t)(done + count)];
      const size_t b = ((size_t)pl.first->n + (size_t)pl.second->n) * SLOT_WORDS * 4;
      if (count > 0 && slot_bytes + b > SLOT_BYTES_PER_BATCH) break;
      slot_bytes += b;
      count++;
    }
    Lane& L = ctx->lanes[ctx->next_lane];
    ctx->next_lane = (ctx->next_lane + 1) % N_LANES;
    hipStream_t st = L.stream;
    int rc;
    const size_t table_bytes = sizeof(CascadePair) * (size_t)count;
    if ((rc = ensure(ctx, L, L.cascade_tables, table_bytes))) return rc;
    if ((rc = ensure(ctx, L, L.cascade_slots, std::max<size_t>(slot_bytes, 256)))) return rc;

    // Outputs: straight into the caller's buffers when those are pinned host memory, otherwise
    // through a device staging area that is copied back.
    const size_t out_words = (size_t)count * (size_t)out_stride;
    const size_t o = (size_t)done * (size_t)out_stride;
    int32_t* k_defaults = (int32_t*)device_view_of_host(out_defaults + done);
    int32_t* k_counts = (int32_t*)device_view_of_host(out_counts + done);
    int32_t* k_lhs = (int32_t*)device_view_of_host(out_lhs_idx + o);
    int32_t* k_rhs = (int32_t*)device_view_of_host(out_rhs_idx + o);
    float* k_dist = out_dist ? (float*)device_view_of_host(out_dist + o) : nullptr;
    const bool direct = k_defaults && k_counts && k_lhs && k_rhs && (k_dist || !out_dist);
    if (!direct || !out_dist) {
      // defaults | counts | lhs | rhs | dist
      const size_t cb = align_up((size_t)count * 4, 256), wb = align_up(out_words * 4, 256);
      if ((rc = ensure(ctx, L, L.out_dev, 2 * cb + 3 * wb + 256))) return rc;
      char* ob = (char*)L.out_dev.p;
      if (!direct) {
        k_defaults = (int32_t*)ob;
        k_counts = (int32_t*)(ob + cb);
        k_lhs = (int32_t*)(ob + 2 * cb);
        k_rhs = (int32_t*)(ob + 2 * cb + wb);
      }
      k_dist = (float*)(ob + 2 * cb + 2 * wb);   // staging, or scratch when the caller wants no distances
    }

    const int slot = L.stage_slot;
    L.stage_slot ^= 1;
    HIP_TRY(ctx, hipEventSynchronize(L.stage_free[slot]));
    if ((rc = ensure_stage(ctx, L, slot, table_bytes))) return rc;
    CascadePair* hp = reinterpret_cast<CascadePair*>(L.stage_host[slot]);
    uint32_t* sp = (uint32_t*)L.cascade_slots.p;
    int max_na = 0, max_n = 0;
    double bytes = 0.0;
    for (int k = 0; k < count; k++) {
      const Image& X = *plan[(size_t)(done + k)].first;
      const Image& Y = *plan[(size_t)(done + k)].second;
      CascadePair& d = hp[k];
      d.ga = (const HashRec*)X.hash_grouped;
      d.gb = (const HashRec*)Y.hash_grouped;
      d.offa = X.hash_offsets;
      d.offb = Y.hash_offsets;
      d.da = X.d;
      d.db = Y.d;
      d.slots_a = sp;
      sp += (size_t)X.n * SLOT_WORDS;
      d.slots_b = sp;
      sp += (size_t)Y.n * SLOT_WORDS;
      d.na = X.n;
      d.nb = Y.n;
      d.out_slot = k;
      d.pad_ = 0;
      max_na = std::max(max_na, X.n);
      max_n = std::max(max_n, std::max(X.n, cross_check ? Y.n : 0));
      const double q = (double)X.n + (double)Y.n;
      // two walks over the bucket-ordered records of both images; the per-query slots are zeroed,
      // read and rewritten by the threshold step and read by the compaction
      bytes += 2.0 * GROUPS * q * sizeof(HashRec) + 3.0 * q * SLOT_WORDS * 4;
    }
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (ctx->profiling) {
      t0 = take_event(ctx);
      t1 = take_event(ctx);
    }
    HIP_TRY(ctx, hipMemcpyAsync(L.cascade_tables.p, hp, table_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(L.stage_free[slot], st));
    if (t0) HIP_TRY(ctx, hipEventRecord(t0, st));
    if (slot_bytes) HIP_TRY(ctx, hipMemsetAsync(L.cascade_slots.p, 0, slot_bytes, st));
    const CascadePair* dp = (const CascadePair*)L.cascade_tables.p;
    const int both = cross_check ? 1 : 0;
    if (max_na > 0) {
      const dim3 walk((unsigned)((max_na + 255) / 256), GROUPS, (unsigned)count);
      const dim3 per_query((unsigned)((max_n + 255) / 256), cross_check ? 2 : 1, (unsigned)count);
      hipLaunchKernelGGL(k_ch_pairs<PASS_COUNT>, walk, dim3(256), 0, st, dp, both);
      HIP_TRY(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_ch_threshold, per_query, dim3(256), 0, st, dp);
      HIP_TRY(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_ch_pairs<PASS_NEAREST>, walk, dim3(256), 0, st, dp, both);
      HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_ch_compact, dim3((unsigned)count), dim3(1024), 0, st, dp, both, k_defaults, k_counts, k_lhs, k_rhs,
                       k_dist, out_stride);
    HIP_TRY(ctx, hipGetLastError());
    if (t1) {
      HIP_TRY(ctx, hipEventRecord(t1, st));
      ctx->cascade_timing.push_back(t0);
      ctx->cascade_timing.push_back(t1);
      ctx->cascade_stats.match_bytes += bytes;
    }
    if (!direct) {
      HIP_TRY(ctx, hipMemcpyAsync(out_defaults + done, k_defaults, (size_t)count * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipMemcpyAsync(out_counts + done, k_counts, (size_t)count * 4, hipMemcpyDeviceToHost, st));
      if (out_words) {
        HIP_TRY(ctx, hipMemcpyAsync(out_lhs_idx + o, k_lhs, out_words * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_rhs_idx + o, k_rhs, out_words * 4, hipMemcpyDeviceToHost, st));
        if (out_dist) HIP_TRY(ctx, hipMemcpyAsync(out_dist + o, k_dist, out_words * 4, hipMemcpyDeviceToHost, st));
      }
    }
    ctx->cascade_stats.pairs += (uint64_t)count;
    done += count;
  }
  return SFM_HIP_OK;
}

int sfm_hip_cascade_reference_projection(float* primary, float* secondary) {
  if (!primary || !secondary) return fail(nullptr, SFM_HIP_ERR_INVALID, "null projection matrix");
  // cascade_hash_matcher.cc:127-146: one generator and one distribution object for all draws
  std::mt19937 generator(std::mt19937::default_seed);
  std::normal_distribution<> normal(0, 1);
  for (int k = 0; k < CODE_BITS * DIM; k++) primary[k] = (float)normal(generator);
  for (int k = 0; k < GROUPS * GROUP_BITS * DIM; k++) secondary[k] = (float)normal(generator);
  return SFM_HIP_OK;
}

int sfm_hip_cascade_image_order(const int64_t* pair_ids, int32_t n_pairs, int64_t* out_ids, int32_t* out_n) {
  if (n_pairs < 0 || (n_pairs > 0 && !pair_ids) || !out_ids || !out_n)
    return fail(nullptr, SFM_HIP_ERR_INVALID, "bad pair list or output buffer");
  std::unordered_set<int64_t> ids;    // cascade_hash_matcher.cc:148-152
  for (int32_t p = 0; p < n_pairs; p++) {
    ids.insert(pair_ids[2 * p]);
    ids.insert(pair_ids[2 * p + 1]);
  }
  int32_t n = 0;
  for (int64_t id : ids) out_ids[n++] = id;
  *out_n = n;
  return SFM_HIP_OK;
}

int sfm_hip_cascade_prepare(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, float* out_mean128) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n_pairs < 0 || (n_pairs > 0 && !pair_ids)) return fail(ctx, SFM_HIP_ERR_INVALID, "bad pair list");
  std::vector<float> primary((size_t)CODE_BITS * DIM), secondary((size_t)GROUPS * GROUP_BITS * DIM);
  int rc = sfm_hip_cascade_reference_projection(primary.data(), secondary.data());
  if (rc == SFM_HIP_OK) rc = sfm_hip_cascade_set_projection(ctx, primary.data(), secondary.data());
  if (rc != SFM_HIP_OK) return rc;
  std::vector<int64_t> ids((size_t)2 * (size_t)std::max(n_pairs, 1));
  int32_t n_ids = 0;
  if ((rc = sfm_hip_cascade_image_order(pair_ids, n_pairs, ids.data(), &n_ids))) return fail(ctx, rc, g_last_error);
  std::vector<float> sums((size_t)std::max(n_ids, 1) * DIM);
  if ((rc = sfm_hip_cascade_column_sums(ctx, ids.data(), n_ids, sums.data()))) return rc;
  // cascade_hash_matcher.cc:154-165: float accumulation image after image, int total, one division
  float mean[DIM];
  for (int c = 0; c < DIM; c++) mean[c] = 0.0f;
  int total_size = 0;
  for (int32_t k = 0; k < n_ids; k++) {
    for (int c = 0; c < DIM; c++) mean[c] = mean[c] + sums[(size_t)k * DIM + c];
    total_size += ctx->images.at(ids[(size_t)k]).n;
  }
  for (int c = 0; c < DIM; c++) mean[c] = mean[c] / (float)total_size;
  if (out_mean128) std::memcpy(out_mean128, mean, sizeof(mean));
  return sfm_hip_cascade_hash_images(ctx, ids.data(), n_ids, mean);
}

int sfm_hip_cascade_get_stats(sfm_hip_ctx* ctx, sfm_hip_cascade_stats* out) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!out) return fail(ctx, SFM_HIP_ERR_INVALID, "out is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = sync_all(ctx)) return rc;
  if (int rc = fold_cascade_timing(ctx)) return rc;
  *out = ctx->cascade_stats;
  return SFM_HIP_OK;
}

int sfm_hip_cascade_reset_stats(sfm_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = sync_all(ctx)) return rc;
  if (int rc = fold_cascade_timing(ctx)) return rc;
  ctx->cascade_stats = sfm_hip_cascade_stats{};
  return SFM_HIP_OK;
}

}  // extern "C"
