// Host-side state behind the C ABI of include/sfm_hip.h, shared by the translation units that
// implement it (sfm_hip_capi.hip: brute-force matcher; sfm_hip_cascade.hip: cascade-hash matcher).
#ifndef SFM_AMD_CSRC_CTX_HPP_
#define SFM_AMD_CSRC_CTX_HPP_

#include "../../include/sfm_hip.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace sfm_host {

struct Image {
  void* block = nullptr;    // one allocation holding everything below
  float* d = nullptr;       // [n_pad][128]
  uint16_t* split = nullptr;  // [n_pad][256] scaled fp16 (high, low) planes: see k_split
  float* norms = nullptr;   // [n_pad + 2]: squared norms, then the largest norm and the largest magnitude
  float* scale = nullptr;   // {2^e, 2^-e} of the split planes
  int32_t n = 0, n_pad = 0;
  // cascade-hash state (sfm_hip_cascade_hash_images): the hash records in image order, the six
  // copies sorted by bucket, and the bucket offsets, in this image's part (hash_block) of an
  // allocation that the images hashed by one call share (an allocation per image costs more than
  // hashing the image); it is freed when the last of them lets go of it
  std::shared_ptr<void> hash_owner;
  void* hash_block = nullptr;
  void* hash_recs = nullptr;      // [n] records of 32 bytes, image order
  // per group, sorted by that group's bucket id, then by index ([6][n] each):
  void* hash_glo = nullptr;       // low 64 bits of the codes
  void* hash_ghi = nullptr;       // high 64 bits
  void* hash_gbkt = nullptr;      // the group's bucket id (16 bits)
  void* hash_gmeta = nullptr;     // all bucket ids + index
  int32_t* hash_offsets = nullptr;  // [6][1025]
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct TimedLaunch {
  hipEvent_t start, stop;
  double flop;
};

// A lane = one work space and the stream its batch's short kernels run on; no memory is shared
// between two batches.
// Brute-force matcher: every batch's distance kernel goes to the context's distance stream, one
// after the other with nothing in between, and its short tail (re-rank, compaction, copies) to the
// tail stream of the batch's lane (two tail streams, lanes take them in turn), which waits for that
// distance kernel through the event that rides on its dispatch packet.  The tail of batch b then runs beside the distance kernel of batch b + 1
// (its workgroups fit in what a distance workgroup leaves of a compute unit).  A lane is taken
// again only after its previous tail has finished, which the host checks (a wait on the distance
// stream would put a packet, and several microseconds, between two distance kernels).
// Cascade-hash matcher: a batch's kernels all run on its lane's stream, lanes 0 and 1 in turn.
struct Lane {
  hipStream_t stream = nullptr;   // one of the context's tail streams (not owned): lane i uses stream i % N_TAIL_STREAMS
  // grow-only device work space
  DevBuf tables_dev, rowpart, colpart, best, out_dev;
  // cascade-hash matcher: pair table, per-query slots + list counters (all zeros between batches),
  // close-pair lists
  DevBuf cascade_tables, cascade_slots, cascade_lists;
  bool cascade_slots_dirty = false;              // a batch was abandoned half queued: clear before the next
  std::vector<char> tables_shadow;               // what tables_dev currently holds
  // pinned staging for the per-batch tables, two slots
  void* stage_host[2] = {nullptr, nullptr};
  size_t stage_bytes[2] = {0, 0};
  hipEvent_t stage_free[2] = {nullptr, nullptr};
  int stage_slot = 0;
  hipEvent_t distance_done = nullptr;   // rides on the lane's latest distance kernel
  hipEvent_t tail_done = nullptr;       // recorded after the lane's latest tail
  bool tail_pending = false;            // tail_done was recorded and not waited for yet
  // Host memory written by the lane's batches whose completion the host has not seen yet (counts, lhs,
  // rhs, distances; the cascade-hash path has a fifth array, the default-record counts).  All of a
  // lane's batches run on one stream, so the latest tail_done covers every one of them.  A later
  // batch on another lane that writes any of this memory again waits for that event, so that results
  // land in call order; sfm_hip_wait_outputs waits for it on the host.  A brute-force lane holds one
  // batch's ranges at most (it is taken again only after its tail has finished); a cascade-hash lane
  // is taken again without a host wait and accumulates the ranges of every batch since the last one.
  // Fixed capacity, no heap: when it is full the host waits for the lane and starts over.
  static constexpr int MAX_OUT_RANGES = 40;
  const char* out_lo[MAX_OUT_RANGES];
  const char* out_hi[MAX_OUT_RANGES];
  int n_out_ranges = 0;
  bool half_queued = false;             // a distance kernel was queued and the call failed before its tail was
};

constexpr int N_LANES = 4;
constexpr int N_CASCADE_LANES = 2;
// lanes[NOALLOC_LANE] is no work space at all: it stands for the calls of sfm_hip_match_pair_noalloc,
// which run on the distance stream, so that the ordering and waiting below see their outputs too
constexpr int NOALLOC_LANE = N_LANES;
// A process has four hardware queues (GPU_MAX_HW_QUEUES) and streams beyond that share them, in
// order: a tail stream on the distance stream's hardware queue would put its kernels between two
// distance kernels again.  Distance stream + upload stream + two tail streams = four.
constexpr int N_TAIL_STREAMS = 2;

}  // namespace sfm_host

struct sfm_hip_ctx {
  int device = 0;
  sfm_host::Lane lanes[sfm_host::N_LANES + 1];
  int next_lane = 0;
  int next_cascade_lane = 0;
  hipStream_t distance_stream = nullptr;   // all launches of the distance kernel, in order
  hipStream_t tail_streams[sfm_host::N_TAIL_STREAMS] = {};
  unsigned launches_since_timed = 0;       // profiling: distance launches since the last timed one
  hipEvent_t previous_stop = nullptr;      // profiling: completion of the launch before one that will be timed from it
  bool previous_stop_stale = false;        // ... and something other than the next launch was queued behind it since
  std::map<int64_t, sfm_host::Image> images;
  std::string last_error;
  unsigned long long* rescanned_dev = nullptr;  // queries answered by the exact full scan
  unsigned char* noalloc_scratch = nullptr;     // this device's copy of the fixed work space of
                                                // sfm_hip_match_pair_noalloc (an array in the code object)

  int engine = SFM_HIP_ENGINE_F32;
  bool profiling = false;
  bool profiling_per_kernel = false;   // cascade-hash batches: an event between every two kernels, and the work counters
  std::vector<sfm_host::TimedLaunch> timed;
  std::vector<hipEvent_t> event_pool;
  std::vector<void*> retired;                   // outgrown work-space buffers, freed at the next drain
  sfm_hip_stats stats{};

  hipStream_t upload_stream = nullptr;
  int uploads_pending = 0;                     // sfm_hip_upload_image_async calls whose work the host has not seen finish
  int n_cu = 256;                              // compute units of the device (one distance workgroup each)
  size_t batch_bytes_limit = (size_t)6 << 30;  // work space per batch
  int max_pairs_per_batch = 256;

  // geometric verification: grow-only device buffers kept between calls
  sfm_host::DevBuf geometry_buffers[11];
  hipStream_t geometry_readback = nullptr;   // reads a launch's "unfinished" counter while the next launch runs
  hipEvent_t geometry_events[16] = {};       // one per launch of a call
  sfm_hip_geometry_stats geometry_stats{};

  // cascade-hash matcher
  float* cascade_projection = nullptr;   // device: [128 + 60][128] projection rows, then the 128-float mean
  bool cascade_has_projection = false;
  bool cascade_projection_is_reference = false;   // ... and it is the reference's, put there by sfm_hip_cascade_prepare
  unsigned long long* cascade_counters = nullptr;   // device: {pairs whose list overflowed, scratch for checks}
  sfm_host::DevBuf cascade_scratch;                 // grow-only: the image table and the sums of sfm_hip_cascade_column_sums
  std::vector<hipEvent_t> cascade_timing;   // five events per timed cascade match batch (the middle three only per kernel)
  std::vector<char> cascade_timing_per_kernel;   // per timed batch: were the middle three recorded
  sfm_hip_cascade_stats cascade_stats{};
};

namespace sfm_host {

extern thread_local std::string g_last_error;  // failures without a context

inline int fail(sfm_hip_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->last_error = msg; else g_last_error = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                 \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      return sfm_host::fail(ctx, e_ == hipErrorOutOfMemory ? SFM_HIP_ERR_NOMEM : SFM_HIP_ERR_HIP, \
                            std::string(#expr) + ": " + hipGetErrorString(e_));            \
    }                                                                                      \
  } while (0)

// The host has seen the lane's latest tail finish: nothing of it is pending any more.
inline void lane_finished(Lane& l) {
  l.tail_pending = false;
  l.n_out_ranges = 0;
}

inline bool lane_writes(const Lane& l, const char* lo, const char* hi) {
  for (int b = 0; b < l.n_out_ranges; b++)
    if (lo < l.out_hi[b] && l.out_lo[b] < hi) return true;
  return false;
}

// Images handed over by sfm_hip_upload_image_async are complete: every entry point that lets a
// kernel on another stream read an image calls this first (the host waits, the GPU keeps working on
// what was queued before).
inline int finish_uploads(sfm_hip_ctx* ctx) {
  if (ctx->uploads_pending > 0) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->upload_stream));
    ctx->uploads_pending = 0;
  }
  return SFM_HIP_OK;
}

inline int sync_all(sfm_hip_ctx* ctx) {
  if (int rc = finish_uploads(ctx)) return rc;
  if (ctx->distance_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->distance_stream));
  for (hipStream_t t : ctx->tail_streams)
    if (t) HIP_TRY(ctx, hipStreamSynchronize(t));
  for (Lane& l : ctx->lanes) lane_finished(l);
  for (void* p : ctx->retired) (void)hipFree(p);   // work-space buffers replaced by larger ones since the last drain
  ctx->retired.clear();
  return SFM_HIP_OK;
}

inline int ensure(sfm_hip_ctx* ctx, Lane& L, DevBuf& b, size_t bytes) {
  (void)L;
  if (b.bytes >= bytes) return SFM_HIP_OK;
  // The old buffer may still be in use by work queued on its lane or on the distance stream.
  // Waiting for that work here would leave the GPU idle while the new buffer is allocated, and
  // hipFree waits for the whole device anyway: the old buffer is set aside and freed the next time
  // the context is drained (sync_all) or destroyed.
  if (b.p) ctx->retired.push_back(b.p);
  b.p = nullptr;
  b.bytes = 0;
  size_t want = bytes + bytes / 4;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    // memory is short: give back what was set aside (which needs the queued work finished) and ask for less
    if (ctx->distance_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->distance_stream));
    for (hipStream_t t : ctx->tail_streams)
      if (t) HIP_TRY(ctx, hipStreamSynchronize(t));
    for (void* p : ctx->retired) (void)hipFree(p);
    ctx->retired.clear();
    want = bytes;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
  }
  b.bytes = want;
  return SFM_HIP_OK;
}

inline int ensure_stage(sfm_hip_ctx* ctx, Lane& L, int slot, size_t bytes) {
  if (L.stage_bytes[slot] >= bytes) return SFM_HIP_OK;
  if (L.stage_host[slot]) HIP_TRY(ctx, hipHostFree(L.stage_host[slot]));
  L.stage_host[slot] = nullptr;
  L.stage_bytes[slot] = 0;
  HIP_TRY(ctx, hipHostMalloc(&L.stage_host[slot], bytes * 2, hipHostMallocDefault));
  L.stage_bytes[slot] = bytes * 2;
  return SFM_HIP_OK;
}

inline void free_image_hashes(Image& im) {
  im.hash_owner.reset();
  im.hash_block = nullptr;
  im.hash_recs = nullptr;
  im.hash_glo = nullptr;
  im.hash_ghi = nullptr;
  im.hash_gbkt = nullptr;
  im.hash_gmeta = nullptr;
  im.hash_offsets = nullptr;
}

inline void free_image(Image& im) {
  free_image_hashes(im);
  if (im.block) (void)hipFree(im.block);
  im = Image();
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// A caller may hand the same output memory to consecutive calls (two sets of buffers used in turn,
// say).  Batches of different lanes run on different streams and are not ordered among themselves,
// so before lane L's batch writes the ranges [lo[a], hi[a]) its stream waits for every pending
// batch of another lane that writes memory overlapping them; then the ranges are added to L's (a
// lane whose earlier batches are known to have finished starts from none: lane_finished).
inline int order_outputs(sfm_hip_ctx* ctx, Lane& L, hipStream_t st, const char* const (&lo)[5], const char* const (&hi)[5]) {
  for (Lane& other : ctx->lanes) {
    if (&other == &L || !other.tail_pending) continue;
    bool overlap = false;
    for (int a = 0; a < 5 && !overlap; a++) overlap = lo[a] && lane_writes(other, lo[a], hi[a]);
#ifndef SFM_ABLATE_OUTPUT_ORDER   // diagnostic build: shows that the test of this ordering can fail
    if (overlap) HIP_TRY(ctx, hipStreamWaitEvent(st, other.tail_done, 0));
#endif
  }
  if (!L.tail_pending) L.n_out_ranges = 0;
  for (int a = 0; a < 5; a++) {
    if (!lo[a]) continue;
    bool known = false;   // the same buffers handed over again and again take one slot
    for (int b = 0; b < L.n_out_ranges && !known; b++) known = L.out_lo[b] == lo[a] && L.out_hi[b] == hi[a];
    if (known) continue;
    if (L.n_out_ranges == Lane::MAX_OUT_RANGES) {
      // more distinct output buffers in flight on this lane than it can remember: let the lane's
      // earlier batches finish (they run on st or, for the noalloc lane, the distance stream)
      HIP_TRY(ctx, hipEventSynchronize(L.tail_done));
      L.n_out_ranges = 0;
    }
    L.out_lo[L.n_out_ranges] = lo[a];
    L.out_hi[L.n_out_ranges] = hi[a];
    L.n_out_ranges++;
  }
  return SFM_HIP_OK;
}

// Device address through which a kernel can write a host output buffer
// directly, or nullptr when the buffer is ordinary pageable memory.
inline void* device_view_of_host(void* host) {
  if (!host) return nullptr;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, host) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (a.type != hipMemoryTypeHost) return nullptr;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return d;
}

}  // namespace sfm_host

#endif  // SFM_AMD_CSRC_CTX_HPP_
