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

constexpr size_t PROJ_FLOATS = (size_t)PROJ_MEAN_AT + DIM;   // projection rows (transposed, padded), then the mean
#ifndef SFM_CASCADE_BATCH_PAIRS   // (overridden in timing experiments: tools/build_variant.sh)
#define SFM_CASCADE_BATCH_PAIRS 1024
#endif
constexpr int MAX_PAIRS_PER_BATCH = SFM_CASCADE_BATCH_PAIRS;
constexpr size_t SLOT_BYTES_PER_BATCH = (size_t)2 << 30;
constexpr size_t LIST_BYTES_PER_BATCH = (size_t)1 << 30;

#ifndef SFM_NEAREST_BLOCKS   // (overridden in timing experiments: tools/build_variant.sh)
#define SFM_NEAREST_BLOCKS 8
#endif
constexpr int NEAREST_BLOCKS = SFM_NEAREST_BLOCKS;    // workgroups striding one pair's list
constexpr int FALLBACK_BLOCKS = 8;   // workgroups striding one pair's queries / records on the list-free path

// Workgroups per (pair, group) of the bucket walk: one per 256 records of the largest first image
// of the batch (k_ch_walk).
int walk_blocks(int max_na) { return std::max(1, (max_na + WALK_THREADS - 1) / WALK_THREADS); }

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

// Events of a timed batch: in front of the walk, behind it, behind the nearest kernel, behind the
// two fallback kernels, behind the compaction.
constexpr size_t BATCH_MARKS = 5;
// cascade_counters: list overflows, scratch for checks, close pairs, -, then WORK_SLOTS counters of the walk's
// candidates and WORK_SLOTS of the exact distances
constexpr int N_COUNTERS = 4 + 2 * (int)WORK_SLOTS;

int fold_cascade_timing(sfm_hip_ctx* ctx) {
  for (size_t k = 0; k + BATCH_MARKS <= ctx->cascade_timing.size(); k += BATCH_MARKS) {
    const hipEvent_t* e = &ctx->cascade_timing[k];
    float whole = 0.f, part[BATCH_MARKS - 1] = {};
    HIP_TRY(ctx, hipEventElapsedTime(&whole, e[0], e[BATCH_MARKS - 1]));
    if (ctx->cascade_timing_per_kernel[k / BATCH_MARKS])
      for (size_t q = 0; q + 1 < BATCH_MARKS; q++) HIP_TRY(ctx, hipEventElapsedTime(&part[q], e[q], e[q + 1]));
    ctx->cascade_stats.match_ms += whole;
    ctx->cascade_stats.walk_ms += part[0];
    ctx->cascade_stats.nearest_ms += part[1];
    ctx->cascade_stats.fallback_ms += part[2];
    ctx->cascade_stats.compact_ms += part[3];
    ctx->cascade_stats.match_launches += 1;
  }
  for (hipEvent_t e : ctx->cascade_timing) ctx->event_pool.push_back(e);
  ctx->cascade_timing.clear();
  ctx->cascade_timing_per_kernel.clear();
  return SFM_HIP_OK;
}

hipEvent_t take_event(sfm_hip_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return e;
}

}  // namespace

extern "C" {

int sfm_hip_cascade_set_projection(sfm_hip_ctx* ctx, const float* primary, const float* secondary) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!primary || !secondary) return fail(ctx, SFM_HIP_ERR_INVALID, "null projection matrix");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->cascade_projection) HIP_TRY(ctx, hipMalloc((void**)&ctx->cascade_projection, PROJ_FLOATS * sizeof(float)));
  // device layout: transposed and padded to 192 rows, [j][192] (see k_ch_hash)
  std::vector<float> t((size_t)PROJ_MEAN_AT, 0.0f);
  for (int r = 0; r < NPROJ; r++) {
    const float* row = r < CODE_BITS ? primary + (size_t)r * DIM : secondary + (size_t)(r - CODE_BITS) * DIM;
    for (int j = 0; j < DIM; j++) t[(size_t)j * PROJ_PADDED + r] = row[j];
  }
  if (int rc = sync_all(ctx)) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->cascade_projection, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  ctx->cascade_has_projection = true;
  ctx->cascade_projection_is_reference = false;   // (sfm_hip_cascade_prepare sets it after its own call)
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
  // table | sums, in a buffer the context keeps (allocating and freeing per call costs as much as the kernel;
  // every use of the buffer is on the upload stream and ends with the host waiting for it)
  const size_t table_bytes = align_up(refs.size() * sizeof(ImageRef), 256);
  if (int rc = ensure(ctx, ctx->lanes[0], ctx->cascade_scratch, table_bytes + (size_t)n_images * DIM * sizeof(float))) return rc;
  void* table = ctx->cascade_scratch.p;
  float* sums = (float*)((char*)ctx->cascade_scratch.p + table_bytes);
  hipStream_t st = ctx->upload_stream;
  HIP_TRY(ctx, hipMemcpyAsync(table, refs.data(), refs.size() * sizeof(ImageRef), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_ch_colsum, dim3((unsigned)n_images), dim3(DIM), 0, st, (const ImageRef*)table, sums);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out_sums, sums, (size_t)n_images * DIM * sizeof(float), hipMemcpyDeviceToHost, st));
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
  // Nothing is touched before every image has passed: a refused call leaves all hashes as they were.
  for (Image* im : ims)
    if ((size_t)im->n > (size_t)MAX_IMAGE_ROWS)
      return fail(ctx, SFM_HIP_ERR_INVALID, "the cascade-hash path takes images of at most 262,144 descriptors");
  // an image that is hashed again may still be read by queued matching work
  bool rehash = false;
  for (Image* im : ims) rehash = rehash || im->hash_block;
  if (rehash)
    if (int rc = sync_all(ctx)) return rc;
  hipStream_t st = ctx->upload_stream;
  // New hashes are built in an allocation of their own -- one for all images of the call, with the job
  // table behind them -- and become the images' hashes only after the kernels that fill them have
  // finished; on any failure it is freed and the images keep what they had (valid older hashes, or none).
  struct Part {
    size_t at = 0, recs_bytes = 0, half_bytes = 0, bkt_bytes = 0, meta_bytes = 0;
  };
  std::vector<Part> parts(ims.size());
  const size_t off_bytes = align_up((size_t)GROUPS * (BUCKETS + 1) * sizeof(int32_t), 256);
  size_t total = 0;
  for (size_t k = 0; k < ims.size(); k++) {
    Part& f = parts[k];
    const size_t n = std::max<size_t>((size_t)ims[k]->n, 1);
    f.at = total;
    f.recs_bytes = align_up(n * sizeof(HashRec), 256);
    f.half_bytes = align_up(n * GROUPS * sizeof(uint2), 256);
    f.bkt_bytes = align_up(n * GROUPS * sizeof(uint16_t), 256);
    f.meta_bytes = align_up(n * GROUPS * sizeof(HashMeta), 256);
    total += f.recs_bytes + 2 * f.half_bytes + f.bkt_bytes + f.meta_bytes + off_bytes;
  }
  const size_t table_at = total;
  total += align_up(ims.size() * sizeof(HashJob), 256);
  void* raw = nullptr;
  {
    const hipError_t e = hipMalloc(&raw, total);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, SFM_HIP_ERR_NOMEM, std::string("hipMalloc hash records: ") + hipGetErrorString(e));
    }
  }
  std::shared_ptr<void> owner(raw, [](void* p) { (void)hipFree(p); });
  auto part_of = [&](size_t k, HashJob& j) {
    const Part& f = parts[k];
    char* at = (char*)raw + f.at;
    j.recs = (HashRec*)at;
    j.glo = (uint2*)(at + f.recs_bytes);
    j.ghi = (uint2*)(at + f.recs_bytes + f.half_bytes);
    j.gbkt = (uint16_t*)(at + f.recs_bytes + 2 * f.half_bytes);
    j.gmeta = (HashMeta*)(at + f.recs_bytes + 2 * f.half_bytes + f.bkt_bytes);
    j.offsets = (int32_t*)(at + f.recs_bytes + 2 * f.half_bytes + f.bkt_bytes + f.meta_bytes);
  };
  std::vector<HashJob> jobs(ims.size());
  auto build = [&]() -> int {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->cascade_projection + (size_t)PROJ_MEAN_AT, mean128, DIM * sizeof(float),
                                hipMemcpyHostToDevice, st));
    int blocks = 0;
    for (size_t k = 0; k < ims.size(); k++) {
      HashJob& j = jobs[k];
      part_of(k, j);
      j.d = ims[k]->d;
      j.n = ims[k]->n;
      j.first_block = blocks;
      blocks += (ims[k]->n + 63) / 64;
    }
    const HashJob* table = (const HashJob*)((char*)raw + table_at);
    HIP_TRY(ctx, hipMemcpyAsync((void*)table, jobs.data(), jobs.size() * sizeof(HashJob), hipMemcpyHostToDevice, st));
    if (blocks > 0) {
      hipLaunchKernelGGL(k_ch_hash, dim3((unsigned)blocks), dim3(256), 0, st, table, (int)jobs.size(),
                         (const float*)ctx->cascade_projection);
      HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_ch_group, dim3(GROUPS, (unsigned)jobs.size()), dim3(64), 0, st, table);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return SFM_HIP_OK;
  };
  if (const int rc = build()) {
    (void)hipStreamSynchronize(st);   // nothing may still write into the allocation, which `owner` frees on return
    return rc;
  }
  for (size_t k = 0; k < ims.size(); k++) {
    Image* im = ims[k];
    const HashJob& j = jobs[k];
    free_image_hashes(*im);
    im->hash_owner = owner;
    im->hash_block = j.recs;
    im->hash_recs = j.recs;
    im->hash_glo = j.glo;
    im->hash_ghi = j.ghi;
    im->hash_gbkt = j.gbkt;
    im->hash_gmeta = j.gmeta;
    im->hash_offsets = j.offsets;
  }
  ctx->cascade_stats.images_hashed += (uint64_t)ims.size();
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
  if (int rc = finish_uploads(ctx)) return rc;

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

  if (!ctx->cascade_counters) {
    HIP_TRY(ctx, hipMalloc((void**)&ctx->cascade_counters, N_COUNTERS * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemset(ctx->cascade_counters, 0, N_COUNTERS * sizeof(unsigned long long)));
  }

  int done = 0;
  while (done < n_pairs) {
    // a batch: its per-query slots (64 bytes per descriptor of either image), one overflow mark per
    // pair, and the close-pair lists
    int count = 0, batch_max_na = 0;
    size_t slot_bytes = 0;
    while (done + count < n_pairs && count < MAX_PAIRS_PER_BATCH) {
      const auto& pl = plan[(size_t)(done + count)];
      const size_t b = ((size_t)pl.first->n + (size_t)pl.second->n) * SLOT_WORDS * 4;
      const int mx = std::max(batch_max_na, pl.first->n);
      const size_t lists = (size_t)(count + 1) * GROUPS * walk_blocks(mx) * LIST_SEGMENT * sizeof(uint2);
      if (count > 0 && (slot_bytes + b > SLOT_BYTES_PER_BATCH || lists > LIST_BYTES_PER_BATCH)) break;
      slot_bytes += b;
      batch_max_na = mx;
      count++;
    }
    // every workgroup of the walk owns one segment of its pair's list
    const int n_walk = walk_blocks(batch_max_na);
    const size_t segments = (size_t)GROUPS * n_walk;
    const size_t counters_at = align_up(slot_bytes, 256);
    const size_t zero_bytes = counters_at + align_up((size_t)count * 4, 256);
    Lane& L = ctx->lanes[ctx->next_cascade_lane];
    ctx->next_cascade_lane = (ctx->next_cascade_lane + 1) % N_CASCADE_LANES;
    hipStream_t st = L.stream;
    int rc;
    const size_t table_bytes = sizeof(CascadePair) * (size_t)count;
    if ((rc = ensure(ctx, L, L.cascade_tables, table_bytes))) return rc;
    const size_t list_bytes = (size_t)count * segments * LIST_SEGMENT * sizeof(uint2);
    if ((rc = ensure(ctx, L, L.cascade_lists, list_bytes + (size_t)count * segments * sizeof(uint32_t)))) return rc;
    // The slots and counters are all zeros between batches (k_ch_compact puts back what a batch
    // dirtied), so only a new allocation, or one a failed call left half used, is cleared.
    const size_t slots_before = L.cascade_slots.bytes;
    if ((rc = ensure(ctx, L, L.cascade_slots, zero_bytes))) return rc;
    if (L.cascade_slots.bytes != slots_before || L.cascade_slots_dirty)
      HIP_TRY(ctx, hipMemsetAsync(L.cascade_slots.p, 0, L.cascade_slots.bytes, st));
    L.cascade_slots_dirty = true;   // until everything below is queued

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
    uint32_t* overflow_marks = (uint32_t*)((char*)L.cascade_slots.p + counters_at);
    uint2* lp = (uint2*)L.cascade_lists.p;
    uint32_t* seg_counts = (uint32_t*)((char*)L.cascade_lists.p + list_bytes);
    int max_na = 0, max_n = 0;
    double bytes = 0.0;
    for (int k = 0; k < count; k++) {
      const Image& X = *plan[(size_t)(done + k)].first;
      const Image& Y = *plan[(size_t)(done + k)].second;
      CascadePair& d = hp[k];
      d.a_lo = (const uint2*)X.hash_glo;
      d.a_hi = (const uint2*)X.hash_ghi;
      d.a_bkt = (const uint16_t*)X.hash_gbkt;
      d.a_meta = (const HashMeta*)X.hash_gmeta;
      d.b_lo = (const uint2*)Y.hash_glo;
      d.b_hi = (const uint2*)Y.hash_ghi;
      d.b_meta = (const HashMeta*)Y.hash_gmeta;
      d.offb = Y.hash_offsets;
      d.da = X.d;
      d.db = Y.d;
      d.hist_a = sp;
      sp += (size_t)X.n * HIST_WORDS;
      d.hist_b = sp;
      sp += (size_t)Y.n * HIST_WORDS;
      d.res_a = sp;
      sp += (size_t)X.n * RES_WORDS;
      d.res_b = sp;
      sp += (size_t)Y.n * RES_WORDS;
      d.list = lp + (size_t)k * segments * LIST_SEGMENT;
      d.seg_counts = seg_counts + (size_t)k * segments;
      d.overflow = overflow_marks + k;
      d.na = X.n;
      d.nb = Y.n;
      d.out_slot = k;
      d.pad_ = 0;
      max_na = std::max(max_na, X.n);
      max_n = std::max(max_n, std::max(X.n, cross_check ? Y.n : 0));
      // what one pair has to read at the least: in all six groups the low halves of both images'
      // bucket-ordered codes (8 bytes) and the first image's bucket ids (2 bytes), and one 16-byte
      // result word per descriptor
      bytes += GROUPS * ((double)X.n * (sizeof(uint2) + sizeof(uint16_t)) + (double)Y.n * sizeof(uint2)) +
               16.0 * ((double)X.n + (double)Y.n);
    }
    hipEvent_t marks[BATCH_MARKS] = {};
    bool timed = ctx->profiling;
    if (timed) {
      for (hipEvent_t& e : marks) timed = (e = take_event(ctx)) && timed;
      if (!timed)   // no events to be had: this batch runs untimed
        for (hipEvent_t e : marks)
          if (e) ctx->event_pool.push_back(e);
    }
    // per_kernel: an event at every mark below; otherwise only in front of the first kernel and behind the last
    const bool per_kernel = timed && ctx->profiling_per_kernel;
    unsigned long long* work = per_kernel ? ctx->cascade_counters : nullptr;
    size_t mark_no = 0;
    auto mark = [&]() -> int {
      const size_t k = mark_no++;
      if (timed && (per_kernel || k == 0 || k == BATCH_MARKS - 1)) HIP_TRY(ctx, hipEventRecord(marks[k], st));
      return SFM_HIP_OK;
    };
    // Results land in call order even when calls share output memory (order_outputs, ctx.hpp).  This
    // lane's earlier batches run on the same stream and need no ordering; they may still be
    // unfinished, so what they write stays on the lane's record beside this batch's ranges.
    {
      const char* const lo[5] = {(const char*)(out_counts + done), (const char*)(out_lhs_idx + o), (const char*)(out_rhs_idx + o),
                                 out_dist ? (const char*)(out_dist + o) : nullptr, (const char*)(out_defaults + done)};
      const char* const hi[5] = {lo[0] + (size_t)count * 4, lo[1] + out_words * 4, lo[2] + out_words * 4,
                                 lo[3] ? lo[3] + out_words * 4 : nullptr, lo[4] + (size_t)count * 4};
      if ((rc = order_outputs(ctx, L, st, lo, hi))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(L.cascade_tables.p, hp, table_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(L.stage_free[slot], st));
    if ((rc = mark())) return rc;
    const CascadePair* dp = (const CascadePair*)L.cascade_tables.p;
    const int both = cross_check ? 1 : 0;
    if (max_na > 0) {
      hipLaunchKernelGGL(k_ch_walk_count, dim3((unsigned)n_walk, GROUPS, (unsigned)count), dim3(WALK_THREADS), 0, st, dp,
                         both, work ? work + 4 : nullptr);
      HIP_TRY(ctx, hipGetLastError());
      if ((rc = mark())) return rc;
      hipLaunchKernelGGL(k_ch_nearest, dim3(NEAREST_BLOCKS, 1, (unsigned)count), dim3(256), (segments + 1) * sizeof(uint32_t),
                         st, dp, both, (int)segments, work ? work + 2 : nullptr);
      HIP_TRY(ctx, hipGetLastError());
      if ((rc = mark())) return rc;
      // the two kernels below return at once for every pair whose list was complete
      hipLaunchKernelGGL(k_ch_take_all, dim3(FALLBACK_BLOCKS, cross_check ? 2 : 1, (unsigned)count), dim3(256), 0, st, dp,
                         ctx->cascade_counters);
      HIP_TRY(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_ch_walk_nearest, dim3(FALLBACK_BLOCKS, 1, (unsigned)count), dim3(WALK_THREADS), 0, st, dp,
                         both);
      HIP_TRY(ctx, hipGetLastError());
    } else {
      if ((rc = mark())) return rc;
      if ((rc = mark())) return rc;
    }
    if ((rc = mark())) return rc;
    hipLaunchKernelGGL(k_ch_compact, dim3((unsigned)count), dim3(CH_COMPACT_THREADS), 0, st, dp, both, k_defaults, k_counts, k_lhs, k_rhs,
                       k_dist, out_stride);
    HIP_TRY(ctx, hipGetLastError());
    L.cascade_slots_dirty = false;
    if ((rc = mark())) return rc;
    if (timed) {
      for (hipEvent_t e : marks) ctx->cascade_timing.push_back(e);
      ctx->cascade_timing_per_kernel.push_back(per_kernel);
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
    HIP_TRY(ctx, hipEventRecord(L.tail_done, st));
    L.tail_pending = true;
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
  int rc = SFM_HIP_OK;
  if (!ctx->cascade_projection_is_reference) {
    // (24,064 normal deviates and an upload: half a millisecond, as much as hashing thirty images -- once per context)
    std::vector<float> primary((size_t)CODE_BITS * DIM), secondary((size_t)GROUPS * GROUP_BITS * DIM);
    rc = sfm_hip_cascade_reference_projection(primary.data(), secondary.data());
    if (rc == SFM_HIP_OK) rc = sfm_hip_cascade_set_projection(ctx, primary.data(), secondary.data());
    if (rc != SFM_HIP_OK) return rc;
    ctx->cascade_projection_is_reference = true;
  }
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
  ctx->cascade_stats.workspace_dirty_words = 0;
  if (ctx->cascade_counters) {
    // the work space must be all zeros now that nothing is queued
    HIP_TRY(ctx, hipMemset(ctx->cascade_counters + 1, 0, sizeof(unsigned long long)));
    for (Lane& L : ctx->lanes) {
      if (!L.cascade_slots.p || L.cascade_slots_dirty) continue;
      hipLaunchKernelGGL(k_ch_count_nonzero, dim3(1024), dim3(256), 0, L.stream, (const uint32_t*)L.cascade_slots.p,
                         L.cascade_slots.bytes / 4, ctx->cascade_counters + 1);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipStreamSynchronize(L.stream));
    }
    unsigned long long c[N_COUNTERS] = {};
    HIP_TRY(ctx, hipMemcpy(c, ctx->cascade_counters, sizeof(c), hipMemcpyDeviceToHost));
    ctx->cascade_stats.list_overflows = c[0];
    ctx->cascade_stats.workspace_dirty_words = c[1];
    ctx->cascade_stats.close_pairs = c[2];
    ctx->cascade_stats.walk_candidates = ctx->cascade_stats.exact_distances = 0;
    for (int k = 0; k < (int)WORK_SLOTS; k++) {
      ctx->cascade_stats.walk_candidates += c[4 + k];
      ctx->cascade_stats.exact_distances += c[4 + WORK_SLOTS + k];
    }
  }
  *out = ctx->cascade_stats;
  return SFM_HIP_OK;
}

int sfm_hip_cascade_reset_stats(sfm_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = sync_all(ctx)) return rc;
  if (int rc = fold_cascade_timing(ctx)) return rc;
  if (ctx->cascade_counters) HIP_TRY(ctx, hipMemset(ctx->cascade_counters, 0, N_COUNTERS * sizeof(unsigned long long)));
  ctx->cascade_stats = sfm_hip_cascade_stats{};
  return SFM_HIP_OK;
}

}  // extern "C"
