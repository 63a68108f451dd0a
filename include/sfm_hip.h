/*
 * C ABI of the MI355X (gfx950) brute-force descriptor matcher.
 *
 * This is the boundary a host program binds to.  The reference has no FFI on
 * this path: its matcher is a C++ class,
 *     BruteForceMatcher::Match(SfMData&, const std::set<Pair>&, bool)
 *     (src/feature_matcher/brute_force_matcher.hpp:5-10, .cc:8-88),
 * behind FeatureMatcherInterface
 *     (src/feature_matcher/feature_matcher_interface.hpp:5-11).
 * The C++ class sfm_amd/host/hip_brute_force_matcher.hpp implements that
 * interface by marshalling SfMData into the flat buffers below.  Each entry
 * point names the part of the reference it stands in for.
 *
 * Conventions: plain pointers and sizes only; every function returns an
 * sfm_hip_status (0 = success) and never throws; the text of the last failure
 * is available from sfm_hip_last_error().  One context drives one GPU from
 * one host thread at a time.  Feature indices are int32 on the wire (the
 * reference's IndexT is int64; the host wrapper widens them).
 */
#ifndef SFM_HIP_H_
#define SFM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_HIP_DIM 128 /* floats per descriptor: src/descriptor.hpp:19-21 */

typedef struct sfm_hip_ctx sfm_hip_ctx;

typedef enum {
  SFM_HIP_OK = 0,
  SFM_HIP_ERR_INVALID = 1,   /* bad argument (null pointer, negative size, unknown image id ...) */
  SFM_HIP_ERR_HIP = 2,       /* a HIP runtime call failed; see sfm_hip_last_error() */
  SFM_HIP_ERR_NOMEM = 3      /* device or pinned-host allocation failed */
} sfm_hip_status;

/* Counters accumulated by a context since creation or the last reset. */
typedef struct {
  uint64_t pairs;             /* image pairs matched */
  uint64_t queries;           /* descriptor queries answered (both directions) */
  uint64_t rescanned_queries; /* queries whose candidate set could not be proven complete and
                                 were answered by the exact full scan instead */
  uint64_t distance_launches; /* launches of the MFMA distance/top-k kernel that were timed */
  double distance_ms;         /* their summed duration, from HIP events on the context's stream
                                 (only while profiling is enabled) */
  double distance_flop;       /* 2*Na*Nb*128 summed over those launches */
} sfm_hip_stats;

/* Number of visible HIP devices. */
int sfm_hip_device_count(int* count);

/* Create / destroy a context bound to one device.  Replaces the reference's
 * per-call ThreadPool (src/internal/thread_pool.hpp:9-46,
 * brute_force_matcher.cc:44): the context owns a HIP stream and the device
 * work space instead of two worker threads. */
int sfm_hip_ctx_create(int device, sfm_hip_ctx** out_ctx);
void sfm_hip_ctx_destroy(sfm_hip_ctx* ctx);

/* Text of the most recent failure on this context (ctx may be NULL for
 * failures of sfm_hip_ctx_create / sfm_hip_device_count).  The pointer stays
 * valid until the next failing call on the same context. */
const char* sfm_hip_last_error(const sfm_hip_ctx* ctx);

/* Upload one image's descriptors: n rows of 128 floats, row-major, exactly the
 * memory of Descriptors::data() (src/descriptor.hpp:23-29).  Replaces the deep
 * copies the reference makes per (pair, direction) task
 * (brute_force_matcher.cc:47-52): an image is uploaded once, padded, and its
 * squared row norms are computed on the device.  Uploading an id again
 * replaces the previous data.  n may be 0.  Blocking. */
int sfm_hip_upload_image(sfm_hip_ctx* ctx, int64_t image_id, const float* descriptors, int32_t n);

/* The same, from descriptors that already lie in the memory of the context's device (for a
 * descriptor set that reached the GPU over xGMI from another GPU instead of from the host; the
 * reference has no counterpart: its images only ever live in host memory).  The source must not be
 * written while the call runs; it is not needed afterwards.  Blocking. */
int sfm_hip_upload_image_device(sfm_hip_ctx* ctx, int64_t image_id, const float* device_descriptors, int32_t n);

/* Forget one image / all images and free their device memory. */
int sfm_hip_drop_image(sfm_hip_ctx* ctx, int64_t image_id);
int sfm_hip_drop_all_images(sfm_hip_ctx* ctx);

/* Number of descriptors of an uploaded image. */
int sfm_hip_image_size(sfm_hip_ctx* ctx, int64_t image_id, int32_t* n);

/*
 * Match n_pairs image pairs.  pair_ids holds (first, second) image ids,
 * interleaved, 2*n_pairs values; both images of every pair must have been
 * uploaded.  This is the body of BruteForceMatcher::Match
 * (brute_force_matcher.cc:13-41 per direction, :64-84 cross-check) with the
 * reference's constants as parameters: ratio (0.8 at :35, compared in double)
 * and cross_check (is_cross_valid).
 *
 * For pair p the surviving matches, in ascending lhs index as the reference
 * emits them, are written to
 *     out_lhs_idx[p*out_stride + m], out_rhs_idx[p*out_stride + m],
 *     out_dist[p*out_stride + m]        for m < out_counts[p]
 * where lhs indexes the first image, rhs the second, and out_dist is the L2
 * distance of the match as L2_metric computes it (src/metric.hpp:15-22).
 * out_stride must be >= the descriptor count of every pair's first image.
 * out_dist may be NULL.  All output buffers are host memory; pinned memory
 * (sfm_hip_host_alloc) keeps the copies asynchronous.
 *
 * The call returns once the work is queued on the context's stream; outputs
 * are valid after sfm_hip_sync().
 */
int sfm_hip_match_pairs(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, double ratio,
                        int cross_check, int32_t* out_counts, int32_t* out_lhs_idx,
                        int32_t* out_rhs_idx, float* out_dist, int64_t out_stride);

/* How the distance stage forms the 128-term products a.b of ||a||^2 + ||b||^2 - 2 a.b
 * (the candidates it proposes are always re-ranked with the reference's exact fp32
 * arithmetic, src/metric.hpp:15-22, and proven complete, so the matches are the same
 * either way):
 *   SFM_HIP_ENGINE_F32    fp32 matrix instructions on the descriptors as uploaded (default);
 *   SFM_HIP_ENGINE_F16X3  each value split into two fp16 numbers (high + low part) and the
 *                         product taken as hi.hi + hi.lo + lo.hi on the fp16 matrix
 *                         instructions with fp32 accumulation: about the fp32 path's accuracy
 *                         at several times its matrix throughput.
 * Applies to calls of sfm_hip_match_pairs made after it.  No counterpart in the reference. */
#define SFM_HIP_ENGINE_F32 0
#define SFM_HIP_ENGINE_F16X3 1
int sfm_hip_set_engine(sfm_hip_ctx* ctx, int engine);

/* Wait for everything queued on the context (the reference's future.get(),
 * brute_force_matcher.cc:67-68,83). */
int sfm_hip_sync(sfm_hip_ctx* ctx);

/* Pinned host memory for the output buffers. */
void* sfm_hip_host_alloc(size_t bytes);
void sfm_hip_host_free(void* p);

/* Profiling: when enabled, every launch of the distance/top-k kernel is
 * bracketed by HIP events on the stream it runs on, and consecutive launches
 * are kept from overlapping (a context alternates between two streams so
 * that one batch's short tail kernels overlap the next batch's distance
 * kernel; with timing on, distance kernels additionally wait for each other,
 * which costs a few percent of throughput).  sfm_hip_get_stats() waits for
 * the queued work and folds the durations into distance_ms. */
int sfm_hip_set_profiling(sfm_hip_ctx* ctx, int enabled);
int sfm_hip_get_stats(sfm_hip_ctx* ctx, sfm_hip_stats* out);
int sfm_hip_reset_stats(sfm_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H_ */
