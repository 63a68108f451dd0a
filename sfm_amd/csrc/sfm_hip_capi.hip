// Host side of the C ABI declared in include/sfm_hip.h: context, image store,
// work-space management, batching of pairs into launches, result copies.
#include "ctx.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "bf_kernels.hpp"

using namespace sfm;
using namespace sfm_host;

thread_local std::string sfm_host::g_last_error;

namespace {

struct PairPlan {
  const Image* X;
  const Image* Y;
  int nchunks, chunk_cols;
  size_t rowpart_words, colpart_words, best_words;
  size_t bytes() const { return (rowpart_words + colpart_words + best_words) * 4; }
};

// Split the columns of a pair into chunks so that a batch has enough workgroups to fill the
// chip (the distance kernel runs one workgroup per CU); fewer chunks mean fewer partial lists
// and less per-workgroup set-up.
void plan_chunks(PairPlan& pl, int row_blocks_in_batch_estimate, int target_blocks) {
  const int nb_pad = pl.Y->n_pad;
  int want = (target_blocks + row_blocks_in_batch_estimate - 1) / std::max(1, row_blocks_in_batch_estimate);
  want = std::max(1, std::min(want, nb_pad / CHUNK_ALIGN));
  int cols = (int)align_up((size_t)(nb_pad + want - 1) / want, CHUNK_ALIGN);
  cols = std::min(cols, MAX_CHUNK_COLS);
  pl.chunk_cols = cols;
  pl.nchunks = (nb_pad + cols - 1) / cols;
  pl.rowpart_words = (size_t)pl.nchunks * ROWPART_FIELDS * pl.X->n_pad;
  pl.colpart_words = (size_t)(pl.X->n_pad / BM) * 3 * nb_pad;
  pl.best_words = 2 * (size_t)pl.X->n_pad + 2 * (size_t)nb_pad;
}

constexpr int N_XCD = 8;
constexpr unsigned PROFILE_EVERY = 8;   // profiling: one distance launch in this many is timed

// Order of a pair's nrb x nch blocks (row block, chunk) for a launch whose first block of this pair
// sits at a position congruent to first_pos modulo 8.  The grid is cut gr x gc = 8 ways, choosing the
// cut that needs the fewest bytes of rows plus columns per rectangle; position p then takes the next
// block of rectangle (first_pos + p) % 8, or of the next rectangle that still has blocks.  Inside a
// rectangle the blocks go chunk by chunk, row blocks ascending.  sink(row block, chunk) receives the
// blocks in launch order; nothing is allocated.
template <class Sink>
void xcd_order(int nrb, int nch, int chunk_cols, int first_pos, Sink&& sink) {
  int best_gr = 1;
  long best_cost = -1;
  for (int gr = 1; gr <= N_XCD; gr *= 2) {
    const int gc = N_XCD / gr;
    const long cost = (long)((nrb + gr - 1) / gr) * BM + (long)((nch + gc - 1) / gc) * chunk_cols;
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best_gr = gr;
    }
  }
#ifdef SFM_XCD_ROW_CUTS_FROM_ENV   // diagnostic build: the cut is taken from the environment
  if (const char* forced = getenv("SFM_XCD_ROW_CUTS")) best_gr = atoi(forced);
#endif
  const int gr = best_gr, gc = N_XCD / gr;
  int r0[N_XCD], nr[N_XCD], c0[N_XCD];
  size_t size[N_XCD], cursor[N_XCD] = {0};
  for (int i = 0; i < gr; i++)
    for (int j = 0; j < gc; j++) {
      const int x = i * gc + j;
      r0[x] = (int)((long)nrb * i / gr);
      nr[x] = (int)((long)nrb * (i + 1) / gr) - r0[x];
      c0[x] = (int)((long)nch * j / gc);
      size[x] = (size_t)nr[x] * (size_t)((int)((long)nch * (j + 1) / gc) - c0[x]);
    }
  const size_t total = (size_t)nrb * nch;
  for (size_t p = 0; p < total; p++) {
    int x = (int)((first_pos + p) % N_XCD);
    for (int tries = 0; tries < N_XCD && cursor[x] >= size[x]; tries++) x = (x + 1) % N_XCD;
    const size_t k = cursor[x]++;
    sink(r0[x] + (int)(k % (size_t)nr[x]), c0[x] + (int)(k / (size_t)nr[x]));
  }
}

hipEvent_t get_event(sfm_hip_ctx* ctx) {
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

int fold_timed(sfm_hip_ctx* ctx) {
  for (auto& t : ctx->timed) {
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, t.start, t.stop));
    ctx->stats.distance_ms += ms;
    ctx->stats.distance_flop += t.flop;
    ctx->stats.distance_launches += 1;
    ctx->event_pool.push_back(t.start);
    ctx->event_pool.push_back(t.stop);
  }
  ctx->timed.clear();
  ctx->launches_since_timed = 0;   // the next launch is timed again
  if (ctx->previous_stop) ctx->event_pool.push_back(ctx->previous_stop);
  ctx->previous_stop = nullptr;
  return SFM_HIP_OK;
}

// ---- work space of sfm_hip_match_pair_noalloc -------------------------------------------------
// One array per GPU, part of the code object: it exists from the moment the library is loaded on a
// device until the process ends.  Host code never uses this name as an address; sfm_hip_ctx_create
// asks the runtime for the array's address on the context's device and everything goes through that.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)8 << 20;
constexpr int NOALLOC_MAX_BLOCKS = 1024;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

// Launch order of the distance workgroups as (row block, chunk) bytes: travels as a kernel argument.
struct NoallocOrder {
  uint8_t row_block[NOALLOC_MAX_BLOCKS];
  uint8_t chunk[NOALLOC_MAX_BLOCKS];
};

// Writes the pair record and the table of distance workgroups into the work space (what
// sfm_hip_match_pairs builds in pinned memory and copies over).
__global__ __launch_bounds__(256) void k_noalloc_tables(PairDev pair, NoallocOrder order, int n_blocks, int engine,
                                                        PairDev* __restrict__ out_pair, BlockDev* __restrict__ out_blocks) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t == 0) *out_pair = pair;
  if (t >= n_blocks) return;
  const int rb = order.row_block[t], c = order.chunk[t];
  const int col0 = c * pair.chunk_cols;
  BlockDev b;
  b.pair = 0;
  b.row0 = rb * BM;
  b.col0 = col0;
  b.ncols = min(pair.chunk_cols, pair.nb_pad - col0);
  b.chunk = c;
  b.pad_ = 0;
  b.xrows = engine == ENGINE_F16X3 ? (const void*)(pair.xh + (size_t)rb * BM * 256)
                                   : (const void*)(pair.x + (size_t)rb * BM * DIM);
  out_blocks[t] = b;
}

// Address through which a kernel on the current device can write caller memory that is pinned host
// memory or device memory; nullptr for anything else.
void* device_view_of_output(void* p) {
  if (!p) return nullptr;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) return p;
  if (a.type != hipMemoryTypeHost) return nullptr;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return d;
}

}  // namespace

extern "C" {

int sfm_hip_device_count(int* count) {
  if (!count) return fail(nullptr, SFM_HIP_ERR_INVALID, "count is null");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(nullptr, SFM_HIP_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return SFM_HIP_OK;
}

int sfm_hip_ctx_create(int device, sfm_hip_ctx** out_ctx) {
  if (!out_ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "out_ctx is null");
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, SFM_HIP_ERR_HIP,
                std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
  if (device < 0 || device >= n) return fail(nullptr, SFM_HIP_ERR_INVALID, "device ordinal out of range");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(nullptr, SFM_HIP_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, SFM_HIP_ERR_HIP,
                std::string("device is ") + prop.gcnArchName + "; this library holds gfx950 code only");
  sfm_hip_ctx* ctx = new sfm_hip_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  e = hipSetDevice(device);
  for (hipStream_t& t : ctx->tail_streams)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&t, hipStreamNonBlocking);
  for (int i = 0; i < N_LANES; i++) ctx->lanes[i].stream = ctx->tail_streams[i % N_TAIL_STREAMS];
  for (Lane& l : ctx->lanes) {
    if (e == hipSuccess && !ctx->upload_stream) e = hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking);
    for (int s = 0; s < 2 && e == hipSuccess; s++) e = hipEventCreateWithFlags(&l.stage_free[s], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&l.distance_done);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l.tail_done, hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->distance_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->rescanned_dev, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(ctx->rescanned_dev, 0, sizeof(unsigned long long));
  // the address of this device's copy of the work-space array (every GPU holds its own)
  if (e == hipSuccess) e = hipGetSymbolAddress((void**)&ctx->noalloc_scratch, HIP_SYMBOL(scratch));
  if (e == hipSuccess && !ctx->noalloc_scratch) e = hipErrorInvalidSymbol;
  if (e != hipSuccess) {
    std::string msg = std::string("context set-up: ") + hipGetErrorString(e);
    sfm_hip_ctx_destroy(ctx);
    return fail(nullptr, SFM_HIP_ERR_HIP, msg);
  }
  *out_ctx = ctx;
  return SFM_HIP_OK;
}

void sfm_hip_ctx_destroy(sfm_hip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->distance_stream) (void)hipStreamSynchronize(ctx->distance_stream);
  for (hipStream_t t : ctx->tail_streams)
    if (t) (void)hipStreamSynchronize(t);
  for (auto& kv : ctx->images) free_image(kv.second);
  for (void* p : ctx->retired) (void)hipFree(p);
  for (Lane& l : ctx->lanes) {
    for (DevBuf* b : {&l.tables_dev, &l.rowpart, &l.colpart, &l.best, &l.out_dev, &l.cascade_tables, &l.cascade_slots, &l.cascade_lists})
      if (b->p) (void)hipFree(b->p);
    for (int s = 0; s < 2; s++) {
      if (l.stage_host[s]) (void)hipHostFree(l.stage_host[s]);
      if (l.stage_free[s]) (void)hipEventDestroy(l.stage_free[s]);
    }
    if (l.distance_done) (void)hipEventDestroy(l.distance_done);
    if (l.tail_done) (void)hipEventDestroy(l.tail_done);
  }
  for (hipStream_t t : ctx->tail_streams)
    if (t) (void)hipStreamDestroy(t);
  if (ctx->distance_stream) (void)hipStreamDestroy(ctx->distance_stream);
  if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
  if (ctx->rescanned_dev) (void)hipFree(ctx->rescanned_dev);
  if (ctx->cascade_projection) (void)hipFree(ctx->cascade_projection);
  if (ctx->cascade_counters) (void)hipFree(ctx->cascade_counters);
  if (ctx->cascade_scratch.p) (void)hipFree(ctx->cascade_scratch.p);
  for (DevBuf& b : ctx->geometry_buffers)
    if (b.p) (void)hipFree(b.p);
  if (ctx->geometry_readback) (void)hipStreamDestroy(ctx->geometry_readback);
  for (hipEvent_t e : ctx->geometry_events)
    if (e) (void)hipEventDestroy(e);
  for (auto e : ctx->cascade_timing) (void)hipEventDestroy(e);
  for (auto& t : ctx->timed) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  if (ctx->previous_stop) (void)hipEventDestroy(ctx->previous_stop);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  delete ctx;
}

const char* sfm_hip_last_error(const sfm_hip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

namespace {
int upload_image(sfm_hip_ctx* ctx, int64_t image_id, const float* descriptors, int32_t n, hipMemcpyKind kind,
                 bool blocking = true) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative descriptor count");
  if (n > 0 && !descriptors) return fail(ctx, SFM_HIP_ERR_INVALID, "descriptors is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // Uploads have a stream of their own: a new image is copied and prepared while matching work
  // queued earlier keeps running.  Only replacing an image has to wait for that work, which may
  // still read the old copy.
  hipStream_t up = ctx->upload_stream;
  auto it = ctx->images.find(image_id);
  if (it != ctx->images.end()) {
    if (int rc = sync_all(ctx)) return rc;
    free_image(it->second);
    ctx->images.erase(it);
  }
  Image im;
  im.n = n;
  im.n_pad = (int32_t)align_up((size_t)std::max(n, 1), IMG_PAD);
  const size_t row_bytes = DIM * sizeof(float);
  const size_t plane = (size_t)im.n_pad * row_bytes;                       // both planes have 512-byte rows
  const size_t norms_bytes = align_up(((size_t)im.n_pad + 2) * sizeof(float), 256);
  hipError_t e = hipMalloc(&im.block, 2 * plane + norms_bytes + 256);
  if (e != hipSuccess)
    return fail(ctx, SFM_HIP_ERR_NOMEM, std::string("hipMalloc image planes: ") + hipGetErrorString(e));
  im.d = (float*)im.block;
  im.split = (uint16_t*)((char*)im.block + plane);
  im.norms = (float*)((char*)im.block + 2 * plane);
  im.scale = (float*)((char*)im.block + 2 * plane + norms_bytes);
  e = hipSuccess;
  if (n > 0) e = hipMemcpyAsync(im.d, descriptors, (size_t)n * row_bytes, kind, up);
  if (e == hipSuccess && im.n_pad > n)
    e = hipMemsetAsync(im.d + (size_t)n * DIM, 0, (size_t)(im.n_pad - n) * row_bytes, up);
  if (e == hipSuccess) e = hipMemsetAsync(im.norms + im.n_pad, 0, 2 * sizeof(float), up);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_norms, dim3((im.n_pad + NORM_ROWS_PER_BLOCK - 1) / NORM_ROWS_PER_BLOCK), dim3(256), 0, up, im.d, im.norms, n, im.n_pad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_split, dim3((unsigned)(((size_t)im.n_pad * 16 + 255) / 256)), dim3(256), 0, up, im.d, im.norms,
                       im.split, im.scale, im.n_pad);
    e = hipGetLastError();
  }
  if (e == hipSuccess && blocking) e = hipStreamSynchronize(up);
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(up);   // nothing queued above may still write into the block that is freed now
    free_image(im);
    return fail(ctx, SFM_HIP_ERR_HIP, std::string("image upload: ") + hipGetErrorString(e));
  }
  if (!blocking) ctx->uploads_pending++;
  ctx->images[image_id] = im;
  return SFM_HIP_OK;
}
}  // namespace

int sfm_hip_upload_image(sfm_hip_ctx* ctx, int64_t image_id, const float* descriptors, int32_t n) {
  return upload_image(ctx, image_id, descriptors, n, hipMemcpyHostToDevice);
}

int sfm_hip_upload_image_async(sfm_hip_ctx* ctx, int64_t image_id, const float* descriptors, int32_t n) {
  return upload_image(ctx, image_id, descriptors, n, hipMemcpyHostToDevice, false);
}

int sfm_hip_upload_image_device(sfm_hip_ctx* ctx, int64_t image_id, const float* device_descriptors, int32_t n) {
  return upload_image(ctx, image_id, device_descriptors, n, hipMemcpyDeviceToDevice);
}

int sfm_hip_drop_image(sfm_hip_ctx* ctx, int64_t image_id) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  auto it = ctx->images.find(image_id);
  if (it == ctx->images.end()) return fail(ctx, SFM_HIP_ERR_INVALID, "unknown image id " + std::to_string(image_id));
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = sync_all(ctx)) return rc;
  free_image(it->second);
  ctx->images.erase(it);
  return SFM_HIP_OK;
}

int sfm_hip_drop_all_images(sfm_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = sync_all(ctx)) return rc;
  for (auto& kv : ctx->images) free_image(kv.second);
  ctx->images.clear();
  return SFM_HIP_OK;
}

int sfm_hip_image_size(sfm_hip_ctx* ctx, int64_t image_id, int32_t* n) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!n) return fail(ctx, SFM_HIP_ERR_INVALID, "n is null");
  auto it = ctx->images.find(image_id);
  if (it == ctx->images.end()) return fail(ctx, SFM_HIP_ERR_INVALID, "unknown image id " + std::to_string(image_id));
  *n = it->second.n;
  return SFM_HIP_OK;
}

int sfm_hip_match_pairs(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, double ratio, int cross_check,
                        int32_t* out_counts, int32_t* out_lhs_idx, int32_t* out_rhs_idx, float* out_dist,
                        int64_t out_stride) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (n_pairs < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative pair count");
  if (n_pairs == 0) return SFM_HIP_OK;
  if (!pair_ids || !out_counts || !out_lhs_idx || !out_rhs_idx)
    return fail(ctx, SFM_HIP_ERR_INVALID, "null pair list or output buffer");
  if (out_stride < 0) return fail(ctx, SFM_HIP_ERR_INVALID, "negative out_stride");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = finish_uploads(ctx)) return rc;

  // validate everything before queuing anything
  std::vector<PairPlan> plans((size_t)n_pairs);
  for (int p = 0; p < n_pairs; p++) {
    auto ix = ctx->images.find(pair_ids[2 * p]);
    auto iy = ctx->images.find(pair_ids[2 * p + 1]);
    if (ix == ctx->images.end() || iy == ctx->images.end())
      return fail(ctx, SFM_HIP_ERR_INVALID,
                  "pair " + std::to_string(p) + " names an image that was not uploaded (" +
                      std::to_string(pair_ids[2 * p]) + ", " + std::to_string(pair_ids[2 * p + 1]) + ")");
    if (ix->second.n > out_stride)
      return fail(ctx, SFM_HIP_ERR_INVALID, "out_stride smaller than the first image of pair " + std::to_string(p));
    plans[p].X = &ix->second;
    plans[p].Y = &iy->second;
  }

  int done = 0;
  while (done < n_pairs) {
    // ---- choose the batch: as many pairs as fit the work-space budget
    int remaining = n_pairs - done;
    long est_row_blocks = 0;
    for (int p = done; p < done + std::min(remaining, ctx->max_pairs_per_batch); p++)
      est_row_blocks += plans[p].X->n_pad / BM;
    int count = 0;
    size_t bytes = 0;
    while (count < remaining && count < ctx->max_pairs_per_batch) {
      PairPlan& pl = plans[done + count];
      plan_chunks(pl, (int)std::min<long>(est_row_blocks, 1 << 20), ctx->n_cu);
      if (count > 0 && bytes + pl.bytes() > ctx->batch_bytes_limit) break;
      bytes += pl.bytes();
      count++;
    }

    Lane& L = ctx->lanes[ctx->next_lane];
    ctx->next_lane = (ctx->next_lane + 1) % N_LANES;
    // the lane's work space is free once its previous tail has run
    if (L.tail_pending) {
      HIP_TRY(ctx, hipEventSynchronize(L.tail_done));
      lane_finished(L);
    }
    if (L.half_queued) {   // an earlier call failed between this lane's distance kernel and its tail
      if (int rc_sync = sync_all(ctx)) return rc_sync;
      L.half_queued = false;
    }

    // ---- lay out the work space and build the tables
    size_t rowpart_words = 0, colpart_words = 0, best_words = 0, n_blocks = 0, n_queries = 0;
    int max_na = 0, max_nb = 0;
    for (int k = 0; k < count; k++) {
      const PairPlan& pl = plans[done + k];
      rowpart_words += pl.rowpart_words;
      colpart_words += pl.colpart_words;
      best_words += pl.best_words;
      if (pl.X->n > 0 && pl.Y->n > 0) n_blocks += (size_t)(pl.X->n_pad / BM) * pl.nchunks;
      n_queries += (size_t)pl.X->n + pl.Y->n;
      max_na = std::max(max_na, pl.X->n);
      max_nb = std::max(max_nb, pl.Y->n);
    }
    int rc;
    const size_t pairs_bytes = align_up(sizeof(PairDev) * count, 256);
    const size_t table_bytes = pairs_bytes + sizeof(BlockDev) * n_blocks;
    if ((rc = ensure(ctx, L, L.tables_dev, table_bytes))) return rc;
    if ((rc = ensure(ctx, L, L.rowpart, rowpart_words * 4))) return rc;
    if ((rc = ensure(ctx, L, L.colpart, colpart_words * 4))) return rc;
    if ((rc = ensure(ctx, L, L.best, best_words * 4))) return rc;

    // Outputs: kernels write straight into the caller's buffers when those are
    // pinned host memory; otherwise into a device staging area that is copied back.
    const size_t out_words = (size_t)count * (size_t)out_stride;
    const size_t o = (size_t)done * (size_t)out_stride;
    int32_t* k_counts = (int32_t*)device_view_of_host(out_counts + done);
    int32_t* k_lhs = (int32_t*)device_view_of_host(out_lhs_idx + o);
    int32_t* k_rhs = (int32_t*)device_view_of_host(out_rhs_idx + o);
    float* k_dist = out_dist ? (float*)device_view_of_host(out_dist + o) : nullptr;
    const bool direct = k_counts && k_lhs && k_rhs && (k_dist || !out_dist);
    if (!direct || !out_dist) {
      // counts | lhs | rhs | dist
      if ((rc = ensure(ctx, L, L.out_dev, align_up((size_t)count * 4, 256) + 3 * align_up(out_words * 4, 256) + 256)))
        return rc;
      char* ob = (char*)L.out_dev.p;
      int32_t* d_counts = (int32_t*)ob;
      ob += align_up((size_t)count * 4, 256);
      int32_t* d_lhs = (int32_t*)ob;
      ob += align_up(out_words * 4, 256);
      int32_t* d_rhs = (int32_t*)ob;
      ob += align_up(out_words * 4, 256);
      float* d_dist = (float*)ob;
      if (!direct) {
        k_counts = d_counts;
        k_lhs = d_lhs;
        k_rhs = d_rhs;
        k_dist = d_dist;
      } else {
        k_dist = d_dist;  // caller does not want distances: let the kernel write them to scratch
      }
    }

    const int slot = L.stage_slot;
    L.stage_slot ^= 1;
    HIP_TRY(ctx, hipEventSynchronize(L.stage_free[slot]));
    if ((rc = ensure_stage(ctx, L, slot, table_bytes))) return rc;
    std::memset(L.stage_host[slot], 0, table_bytes);
    PairDev* hp = reinterpret_cast<PairDev*>(L.stage_host[slot]);
    BlockDev* hb = reinterpret_cast<BlockDev*>((char*)L.stage_host[slot] + pairs_bytes);

    uint32_t* rp = (uint32_t*)L.rowpart.p;
    uint32_t* cp = (uint32_t*)L.colpart.p;
    uint32_t* bp = (uint32_t*)L.best.p;
    size_t nb_i = 0;
    double flop = 0.0;
    for (int k = 0; k < count; k++) {
      const PairPlan& pl = plans[done + k];
      PairDev& d = hp[k];
      d.x = pl.X->d;
      d.xn = pl.X->norms;
      d.y = pl.Y->d;
      d.yn = pl.Y->norms;
      d.rowpart = rp;
      rp += pl.rowpart_words;
      d.colpart = cp;
      cp += pl.colpart_words;
      d.fwd_best = (int32_t*)bp;
      d.fwd_dist = (float*)(bp + pl.X->n_pad);
      d.rev_best = (int32_t*)(bp + 2 * (size_t)pl.X->n_pad);
      d.rev_dist = (float*)(bp + 2 * (size_t)pl.X->n_pad + pl.Y->n_pad);
      bp += pl.best_words;
      d.na = pl.X->n;
      d.nb = pl.Y->n;
      d.na_pad = pl.X->n_pad;
      d.nb_pad = pl.Y->n_pad;
      d.nchunks = pl.nchunks;
      d.chunk_cols = pl.chunk_cols;
      d.out_slot = k;
      d.pad_ = 0;
      d.xh = pl.X->split;
      d.yh = pl.Y->split;
      d.xscale = pl.X->scale;
      d.yscale = pl.Y->scale;
      if (pl.X->n > 0 && pl.Y->n > 0) {
        // Workgroup i of a launch runs on XCD i % 8, each XCD with its own L2.  The pair's grid of
        // (row block, chunk) blocks is cut into eight rectangles, one per XCD, and the blocks are
        // laid out so that position p of the launch holds a block of rectangle p % 8: an XCD then
        // pulls only its rectangle's rows and columns through its L2 instead of nearly everything.
        const int nrb = pl.X->n_pad / BM, nch = pl.nchunks;
        xcd_order(nrb, nch, pl.chunk_cols, (int)(nb_i % N_XCD), [&](int rb, int c) {
          const int col0 = c * pl.chunk_cols;
          BlockDev& b = hb[nb_i++];
          b.pair = k;
          b.row0 = rb * BM;
          b.col0 = col0;
          b.ncols = std::min(pl.chunk_cols, pl.Y->n_pad - col0);
          b.chunk = c;
          b.xrows = ctx->engine == SFM_HIP_ENGINE_F16X3
                        ? (const void*)(pl.X->split + (size_t)rb * BM * 256)
                        : (const void*)(pl.X->d + (size_t)rb * BM * DIM);
        });
        flop += 2.0 * pl.X->n * (double)pl.Y->n * DIM;
      }
    }

    // ---- queue the batch: tables and distance kernel on the distance stream, the rest on the lane's
    hipStream_t st = L.stream;
    hipStream_t ds = ctx->distance_stream;
    // The device keeps the previous batch's tables; identical tables (the same
    // pairs matched again) need no new upload.
    const bool same_tables = L.tables_shadow.size() == table_bytes &&
                             std::memcmp(L.tables_shadow.data(), L.stage_host[slot], table_bytes) == 0;
    if (!same_tables) {
      HIP_TRY(ctx, hipMemcpyAsync(L.tables_dev.p, L.stage_host[slot], table_bytes, hipMemcpyHostToDevice, ds));
      HIP_TRY(ctx, hipEventRecord(L.stage_free[slot], ds));
      L.tables_shadow.assign((char*)L.stage_host[slot], (char*)L.stage_host[slot] + table_bytes);
    }

    const PairDev* dp = (const PairDev*)L.tables_dev.p;
    const BlockDev* db = (const BlockDev*)((const char*)L.tables_dev.p + pairs_bytes);
    if (n_blocks) {
      // The event the tail waits for rides on the kernel's own dispatch packet
      // (hipExtLaunchKernelGGL): it costs nothing, where a separate event record puts a marker
      // packet, and several microseconds, between this kernel and the next.  While profiling is
      // on, one launch in PROFILE_EVERY is timed and its duration goes into the statistics; the
      // launches in between run as they always do.  A timed launch that is queued while the launch
      // before it has not finished -- launches queued back to back, the case that matters -- is
      // timed from that launch's completion to its own: both events ride on dispatch packets and
      // nothing is added to the stream (what lies between two launches, 1-2 us for a single pair and
      // 25-30 us of table copy for a batch, counts in).  Otherwise -- the first launch after a pause
      // -- it gets a start event of its own, i.e. one marker packet.  Markers in front of launches
      // in a busy stream were dropped for what they did to the run: 1.1 % of the single-pair step
      // rate at one in eight, and the second marker-timed 64-pair launch of a run of launches took
      // 13.6 ms instead of 8.3, reproducibly (kernel trace).
      void (*kernel)(const PairDev*, const BlockDev*) =
          ctx->engine == SFM_HIP_ENGINE_F16X3 ? k_distance_topk<ENGINE_F16X3> : k_distance_topk<ENGINE_F32>;
      hipEvent_t finished = L.distance_done;
      L.half_queued = true;
      TimedLaunch tl{};
      bool time_this = false, next_is_timed = false;
      if (ctx->profiling) {
        time_this = ctx->launches_since_timed % PROFILE_EVERY == 0;
        next_is_timed = (ctx->launches_since_timed + 1) % PROFILE_EVERY == 0;
        ctx->launches_since_timed++;
      }
      hipEvent_t timed_from = ctx->previous_stop;   // set by the launch before this one, if at all
      ctx->previous_stop = nullptr;
      const bool stale = ctx->previous_stop_stale;  // other kernels were queued on this stream since
      ctx->previous_stop_stale = false;
      bool own_marker = false;
      if (time_this) {
        if (timed_from && !stale && hipEventQuery(timed_from) == hipErrorNotReady) {
          tl.start = timed_from;
          timed_from = nullptr;
          tl.stop = get_event(ctx);
          if (!tl.stop) {
            ctx->event_pool.push_back(tl.start);
            tl.start = nullptr;
          }
        } else {
          own_marker = true;
          tl.start = get_event(ctx);
          tl.stop = get_event(ctx);
          if (!tl.start || !tl.stop) {   // no events to be had: this launch runs untimed, like its neighbours
            if (tl.start) ctx->event_pool.push_back(tl.start);
            if (tl.stop) ctx->event_pool.push_back(tl.stop);
            tl.start = tl.stop = nullptr;
          }
        }
      }
      (void)hipGetLastError();   // hipErrorNotReady is an answer, not a failure
      if (timed_from) ctx->event_pool.push_back(timed_from);   // was not needed after all
      if (tl.start) {
        tl.flop = flop;
        hipExtLaunchKernelGGL(kernel, dim3((unsigned)n_blocks), dim3(256), 0, ds, own_marker ? tl.start : nullptr, tl.stop, 0, dp,
                              db);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
          ctx->event_pool.push_back(tl.start);
          ctx->event_pool.push_back(tl.stop);
          HIP_TRY(ctx, e);
        }
        ctx->timed.push_back(tl);
        finished = tl.stop;
      } else {
        if (next_is_timed) {
          hipEvent_t ev = get_event(ctx);
          if (ev) finished = ctx->previous_stop = ev;
        }
        hipExtLaunchKernelGGL(kernel, dim3((unsigned)n_blocks), dim3(256), 0, ds, nullptr, finished, 0, dp, db);
        HIP_TRY(ctx, hipGetLastError());
      }
      HIP_TRY(ctx, hipStreamWaitEvent(st, finished, 0));
    } else if (!same_tables) {
      // no distance kernel to wait for, but the tail reads the tables
      HIP_TRY(ctx, hipStreamWaitEvent(st, L.stage_free[slot], 0));
    }
    // Results land in call order even when calls share output memory (order_outputs, ctx.hpp).
    {
      const char* const lo[5] = {(const char*)(out_counts + done), (const char*)(out_lhs_idx + o), (const char*)(out_rhs_idx + o),
                                 out_dist ? (const char*)(out_dist + o) : nullptr, nullptr};
      const char* const hi[5] = {lo[0] + (size_t)count * 4, lo[1] + out_words * 4, lo[2] + out_words * 4,
                                 lo[3] ? lo[3] + out_words * 4 : nullptr, nullptr};
      if ((rc = order_outputs(ctx, L, st, lo, hi))) return rc;
    }
    // Pairs with an empty image still go through the re-rank step: with no
    // candidates every query comes out unmatched, as in the reference.  Their
    // partial lists were never written, so give them empty ones.
    for (int k = 0; k < count; k++) {
      const PairPlan& pl = plans[done + k];
      if (pl.X->n > 0 && pl.Y->n > 0) continue;
      HIP_TRY(ctx, hipMemsetAsync(hp[k].rowpart, 0xFF, pl.rowpart_words * 4, st));
      HIP_TRY(ctx, hipMemsetAsync(hp[k].colpart, 0xFF, pl.colpart_words * 4, st));
    }
#ifdef SFM_ABLATE_TAIL   // diagnostic build: distance kernels only (the outputs are not written)
    const int max_q = 0;
#else
    const int max_q = cross_check ? std::max(max_na, max_nb) : max_na;
#endif
    if (max_q > 0) {
      const dim3 grid((max_q + RR_QUERIES - 1) / RR_QUERIES, count, cross_check ? 2 : 1);
      const int eng = ctx->engine == SFM_HIP_ENGINE_F16X3 ? ENGINE_F16X3 : ENGINE_F32;
      // enough workgroups for several per compute unit: the variant with coalesced row loads;
      // otherwise the one that fits beside a distance workgroup
      if ((size_t)grid.x * grid.y * grid.z >= (size_t)4 * ctx->n_cu)
        hipLaunchKernelGGL(k_rerank<64>, grid, dim3(256), 0, st, dp, ratio, eng, ctx->rescanned_dev, out_dist ? 1 : 0);
      else
        hipLaunchKernelGGL(k_rerank<0>, grid, dim3(256), 0, st, dp, ratio, eng, ctx->rescanned_dev, out_dist ? 1 : 0);
      HIP_TRY(ctx, hipGetLastError());
    }
#if !defined(SFM_ABLATE_TAIL) && !defined(SFM_ABLATE_COMPACT)
    if (count >= ctx->n_cu / 4)
      hipLaunchKernelGGL(k_cross_compact<1024>, dim3(count), dim3(1024), 0, st, dp, cross_check ? 1 : 0, k_counts, k_lhs,
                         k_rhs, k_dist, out_stride);
    else   // a few pairs: workgroups small enough to sit beside the next launch's distance workgroups, one per sweep
      hipLaunchKernelGGL(k_cross_compact<256>, dim3(count, (unsigned)std::max(1, (max_na + 2047) / 2048)), dim3(256), 0, st, dp,
                         cross_check ? 1 : 0, k_counts, k_lhs, k_rhs, k_dist, out_stride);
    HIP_TRY(ctx, hipGetLastError());
#endif

    if (!direct) {
      HIP_TRY(ctx, hipMemcpyAsync(out_counts + done, k_counts, (size_t)count * 4, hipMemcpyDeviceToHost, st));
      if (out_words) {
        HIP_TRY(ctx, hipMemcpyAsync(out_lhs_idx + o, k_lhs, out_words * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_rhs_idx + o, k_rhs, out_words * 4, hipMemcpyDeviceToHost, st));
        if (out_dist) HIP_TRY(ctx, hipMemcpyAsync(out_dist + o, k_dist, out_words * 4, hipMemcpyDeviceToHost, st));
      }
    }

    HIP_TRY(ctx, hipEventRecord(L.tail_done, st));
    L.tail_pending = true;
    L.half_queued = false;

    ctx->stats.pairs += (uint64_t)count;
    ctx->stats.queries += n_queries;
    done += count;
  }
  return SFM_HIP_OK;
}

int sfm_hip_match_pair_noalloc(sfm_hip_ctx* ctx, int64_t first_image, int64_t second_image, double ratio,
                               int cross_check, int32_t* out_count, int32_t* out_lhs_idx, int32_t* out_rhs_idx,
                               float* out_dist) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!out_count || !out_lhs_idx || !out_rhs_idx) return fail(ctx, SFM_HIP_ERR_INVALID, "null output buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = finish_uploads(ctx)) return rc;   // (a stream wait, no allocation)
  auto ix = ctx->images.find(first_image);
  auto iy = ctx->images.find(second_image);
  if (ix == ctx->images.end() || iy == ctx->images.end())
    return fail(ctx, SFM_HIP_ERR_INVALID, "the pair names an image that was not uploaded");
  if (ix->second.n > SFM_HIP_NOALLOC_MAX_ROWS || iy->second.n > SFM_HIP_NOALLOC_MAX_ROWS)
    return fail(ctx, SFM_HIP_ERR_INVALID,
                "sfm_hip_match_pair_noalloc takes images of at most " + std::to_string(SFM_HIP_NOALLOC_MAX_ROWS) +
                    " descriptors (got " + std::to_string(ix->second.n) + " and " + std::to_string(iy->second.n) +
                    "): its work space has a fixed size; use sfm_hip_match_pairs");
  int32_t* k_count = (int32_t*)device_view_of_output(out_count);
  int32_t* k_lhs = (int32_t*)device_view_of_output(out_lhs_idx);
  int32_t* k_rhs = (int32_t*)device_view_of_output(out_rhs_idx);
  float* k_dist = (float*)device_view_of_output(out_dist);
  if (!k_count || !k_lhs || !k_rhs || (out_dist && !k_dist))
    return fail(ctx, SFM_HIP_ERR_INVALID,
                "sfm_hip_match_pair_noalloc writes its outputs from the GPU: they must be pinned host memory "
                "(sfm_hip_host_alloc) or device memory");

  PairPlan pl{};
  pl.X = &ix->second;
  pl.Y = &iy->second;
  const int nrb = pl.X->n_pad / BM;
  plan_chunks(pl, nrb, ctx->n_cu);
  const bool has_blocks = pl.X->n > 0 && pl.Y->n > 0;
  const int n_blocks = has_blocks ? nrb * pl.nchunks : 0;

  // work space: pair record | block table | distances nobody asked for | results per query | lists
  unsigned char* base = ctx->noalloc_scratch;
  size_t at = 0;
  PairDev* d_pair = (PairDev*)(base + at);
  at += align_up(sizeof(PairDev), 256);
  BlockDev* d_blocks = (BlockDev*)(base + at);
  at += sizeof(BlockDev) * NOALLOC_MAX_BLOCKS;
  float* d_dist = (float*)(base + at);
  at += (size_t)SFM_HIP_NOALLOC_MAX_ROWS * sizeof(float);
  uint32_t* bp = (uint32_t*)(base + at);
  at += align_up(pl.best_words * 4, 256);
  uint32_t* rp = (uint32_t*)(base + at);
  at += align_up(pl.rowpart_words * 4, 256);
  uint32_t* cp = (uint32_t*)(base + at);
  at += align_up(pl.colpart_words * 4, 256);
  if (at > NOALLOC_SCRATCH_BYTES || n_blocks > NOALLOC_MAX_BLOCKS || nrb > 256 || pl.nchunks > 256)
    return fail(ctx, SFM_HIP_ERR_INVALID, "sfm_hip_match_pair_noalloc: the pair does not fit the fixed work space");

  PairDev d{};
  d.x = pl.X->d;
  d.xn = pl.X->norms;
  d.y = pl.Y->d;
  d.yn = pl.Y->norms;
  d.rowpart = rp;
  d.colpart = cp;
  d.fwd_best = (int32_t*)bp;
  d.fwd_dist = (float*)(bp + pl.X->n_pad);
  d.rev_best = (int32_t*)(bp + 2 * (size_t)pl.X->n_pad);
  d.rev_dist = (float*)(bp + 2 * (size_t)pl.X->n_pad + pl.Y->n_pad);
  d.na = pl.X->n;
  d.nb = pl.Y->n;
  d.na_pad = pl.X->n_pad;
  d.nb_pad = pl.Y->n_pad;
  d.nchunks = pl.nchunks;
  d.chunk_cols = pl.chunk_cols;
  d.out_slot = 0;
  d.xh = pl.X->split;
  d.yh = pl.Y->split;
  d.xscale = pl.X->scale;
  d.yscale = pl.Y->scale;
  NoallocOrder order;
  int nb_i = 0;
  if (has_blocks)
    xcd_order(nrb, pl.nchunks, pl.chunk_cols, 0, [&](int rb, int c) {
      order.row_block[nb_i] = (uint8_t)rb;
      order.chunk[nb_i] = (uint8_t)c;
      nb_i++;
    });

  // Everything runs on the distance stream, in order: calls of this entry point share one work space
  // and must not overlap.  Tails of earlier sfm_hip_match_pairs calls run on other streams; this
  // call's results must land after theirs.
  hipStream_t ds = ctx->distance_stream;
  Lane& NL = ctx->lanes[NOALLOC_LANE];
  for (Lane& other : ctx->lanes)
    if (&other != &NL && other.tail_pending) HIP_TRY(ctx, hipStreamWaitEvent(ds, other.tail_done, 0));
  {
    // what this call writes, for later calls that write the same memory from another stream and for
    // sfm_hip_wait_outputs (fixed-size bookkeeping in the context: nothing is allocated)
    const size_t n_first = (size_t)pl.X->n;
    const char* const lo[5] = {(const char*)out_count, (const char*)out_lhs_idx, (const char*)out_rhs_idx,
                               (const char*)out_dist, nullptr};
    const char* const hi[5] = {lo[0] + sizeof(int32_t), lo[1] + n_first * 4, lo[2] + n_first * 4,
                               lo[3] ? lo[3] + n_first * 4 : nullptr, nullptr};
    if (int rc = order_outputs(ctx, NL, ds, lo, hi)) return rc;
  }
  const int eng = ctx->engine == SFM_HIP_ENGINE_F16X3 ? ENGINE_F16X3 : ENGINE_F32;
  hipLaunchKernelGGL(k_noalloc_tables, dim3((unsigned)std::max(1, (n_blocks + 255) / 256)), dim3(256), 0, ds, d, order,
                     n_blocks, eng, d_pair, d_blocks);
  HIP_TRY(ctx, hipGetLastError());
  if (n_blocks) {
    void (*kernel)(const PairDev*, const BlockDev*) =
        eng == ENGINE_F16X3 ? k_distance_topk<ENGINE_F16X3> : k_distance_topk<ENGINE_F32>;
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_blocks), dim3(256), 0, ds, (const PairDev*)d_pair,
                       (const BlockDev*)d_blocks);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    // an empty image: no candidate lists were written, so give the re-rank empty ones
    HIP_TRY(ctx, hipMemsetAsync(rp, 0xFF, pl.rowpart_words * 4, ds));
    HIP_TRY(ctx, hipMemsetAsync(cp, 0xFF, pl.colpart_words * 4, ds));
  }
  const int max_q = cross_check ? std::max(pl.X->n, pl.Y->n) : pl.X->n;
  if (max_q > 0) {
    const dim3 grid((max_q + RR_QUERIES - 1) / RR_QUERIES, 1, cross_check ? 2 : 1);
    if ((size_t)grid.x * grid.z >= (size_t)4 * ctx->n_cu)
      hipLaunchKernelGGL(k_rerank<64>, grid, dim3(256), 0, ds, (const PairDev*)d_pair, ratio, eng, ctx->rescanned_dev,
                         out_dist ? 1 : 0);
    else
      hipLaunchKernelGGL(k_rerank<0>, grid, dim3(256), 0, ds, (const PairDev*)d_pair, ratio, eng, ctx->rescanned_dev,
                         out_dist ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(k_cross_compact<256>, dim3(1, (unsigned)std::max(1, (pl.X->n + 2047) / 2048)), dim3(256), 0, ds,
                     (const PairDev*)d_pair, cross_check ? 1 : 0, k_count,
                     k_lhs, k_rhs, k_dist ? k_dist : d_dist, (int64_t)SFM_HIP_NOALLOC_MAX_ROWS);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(NL.tail_done, ds));   // an event made with the context
  NL.tail_pending = true;
  // A distance launch that would be timed "from the completion of the launch before it" would now
  // count this call's kernels in: such a launch takes a start marker of its own instead.
  ctx->previous_stop_stale = true;
  ctx->stats.pairs += 1;
  ctx->stats.queries += (uint64_t)pl.X->n + (uint64_t)pl.Y->n;
  return SFM_HIP_OK;
}

int sfm_hip_sync(sfm_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return sync_all(ctx);
}

int sfm_hip_wait_outputs(sfm_hip_ctx* ctx, const void* out, size_t bytes) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!out || bytes == 0) return SFM_HIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const char* lo = (const char*)out;
  const char* hi = lo + bytes;
  // A lane remembers what its batches write until the host has seen them finish, and its latest
  // tail_done covers all of them (one stream per lane).  Waiting for a lane's latest batch can mean
  // waiting for a later call than the one asked about (a cascade-hash lane is taken again without a
  // host wait): never for less.
  for (Lane& L : ctx->lanes) {
    if (!L.tail_pending || !lane_writes(L, lo, hi)) continue;
    HIP_TRY(ctx, hipEventSynchronize(L.tail_done));
    lane_finished(L);
  }
  return SFM_HIP_OK;
}

void* sfm_hip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void sfm_hip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int sfm_hip_set_engine(sfm_hip_ctx* ctx, int engine) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (engine != SFM_HIP_ENGINE_F32 && engine != SFM_HIP_ENGINE_F16X3)
    return fail(ctx, SFM_HIP_ERR_INVALID, "unknown engine " + std::to_string(engine));
  ctx->engine = engine;
  return SFM_HIP_OK;
}

int sfm_hip_set_profiling(sfm_hip_ctx* ctx, int enabled) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  ctx->profiling = enabled != 0;
  ctx->profiling_per_kernel = enabled >= 2;
  return SFM_HIP_OK;
}

int sfm_hip_get_stats(sfm_hip_ctx* ctx, sfm_hip_stats* out) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  if (!out) return fail(ctx, SFM_HIP_ERR_INVALID, "out is null");
  int rc = sfm_hip_sync(ctx);
  if (rc) return rc;
  if ((rc = fold_timed(ctx))) return rc;
  unsigned long long r = 0;
  HIP_TRY(ctx, hipMemcpy(&r, ctx->rescanned_dev, sizeof(r), hipMemcpyDeviceToHost));
  ctx->stats.rescanned_queries = r;
  *out = ctx->stats;
  return SFM_HIP_OK;
}

int sfm_hip_reset_stats(sfm_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, SFM_HIP_ERR_INVALID, "ctx is null");
  int rc = sfm_hip_sync(ctx);
  if (rc) return rc;
  if ((rc = fold_timed(ctx))) return rc;
  HIP_TRY(ctx, hipMemset(ctx->rescanned_dev, 0, sizeof(unsigned long long)));
  ctx->stats = sfm_hip_stats{};
  return SFM_HIP_OK;
}

}  // extern "C"
