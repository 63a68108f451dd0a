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
  double distance_ms;         /* their summed duration, from HIP events on the stream they ran on
                                 (only while profiling is enabled) */
  double distance_flop;       /* 2*Na*Nb*128 summed over those launches */
} sfm_hip_stats;

/* Number of visible HIP devices. */
int sfm_hip_device_count(int* count);

/* Create / destroy a context bound to one device.  Replaces the reference's
 * per-call ThreadPool (src/internal/thread_pool.hpp:9-46,
 * brute_force_matcher.cc:44): the context owns its HIP streams and the device
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

/* The same without waiting for the copy and the preparation to finish: the call returns once they
 * are queued on the context's upload stream, so that consecutive uploads overlap (the copy of one
 * image runs beside the norm and split kernels of the one before) and the host goes on.  The first
 * later call that lets the GPU read images (sfm_hip_match_pairs, sfm_hip_match_pair_noalloc,
 * sfm_hip_cascade_*, sfm_hip_sync, a drop or replacement of an image) waits on the host until all
 * queued uploads are complete; work queued earlier keeps the GPU busy meanwhile.  `descriptors` must
 * stay valid and unchanged until then.  Stands in for the same copies of the reference as
 * sfm_hip_upload_image (brute_force_matcher.cc:47-52). */
int sfm_hip_upload_image_async(sfm_hip_ctx* ctx, int64_t image_id, const float* descriptors, int32_t n);

/* The same, from descriptors that already lie in the memory of the context's device (for a
 * descriptor set that reached the GPU over xGMI from another GPU instead of from the host; the
 * reference has no counterpart: its images only ever live in host memory).  The source must not be
 * written while the call runs; it is not needed afterwards.  Blocking. */
int sfm_hip_upload_image_device(sfm_hip_ctx* ctx, int64_t image_id, const float* device_descriptors, int32_t n);

/* Give every one of n_ctx contexts (one per GPU, all driven by this process) every one of n_images
 * images: image_ids[i] has rows[i] descriptors at descriptors[i] (host memory, the layout of
 * sfm_hip_upload_image).  The images are cut into n_ctx shares of about equal size; each context's
 * GPU receives only its share from the host (so an image crosses PCIe once, and all n_ctx links
 * work at the same time), then ONE all-gather over RCCL/xGMI hands every share to every GPU, and
 * every context registers all images as sfm_hip_upload_image_device does.  The alternative --
 * sfm_hip_upload_image of every image on every context -- moves n_ctx times the bytes over PCIe;
 * which is faster depends on the machine (out_times, if not NULL, receives where the time went).
 * Afterwards the contexts are as after n_ctx * n_images calls of sfm_hip_upload_image.
 * No counterpart in the reference, whose images only live in host memory
 * (brute_force_matcher.cc:47-52 copies them per task).  RCCL is loaded when this is first called;
 * without a loadable librccl the call fails with SFM_HIP_ERR_HIP and the contexts are unchanged, as
 * they are after any failure before the images are registered.  A failure while they are being
 * registered (a GPU out of memory) removes the call's images from every context, so that the
 * contexts still agree on what they hold; images the call did not name are never touched.
 * Contexts must sit on distinct devices.  Blocking; at most one call at a time per process. */
typedef struct {
  double comm_init_s;   /* creating the communicators (first call for a set of devices only) */
  double upload_s;      /* host -> device, each GPU its own share */
  double allgather_s;   /* the collective */
  double register_s;    /* padding, norms and split planes of every image on every GPU */
} sfm_hip_replicate_times;
int sfm_hip_replicate_images(sfm_hip_ctx* const* ctxs, int32_t n_ctx, const int64_t* image_ids,
                             const float* const* descriptors, const int32_t* rows, int32_t n_images,
                             sfm_hip_replicate_times* out_times);

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
 * The call returns once the work is queued on the context's streams (it waits
 * only when more than four batches are in flight); outputs are valid after
 * sfm_hip_sync().  A later call may be handed output memory that an earlier,
 * still unfinished call writes: the results land in call order, so the memory
 * ends up holding the later call's.
 */
int sfm_hip_match_pairs(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, double ratio,
                        int cross_check, int32_t* out_counts, int32_t* out_lhs_idx,
                        int32_t* out_rhs_idx, float* out_dist, int64_t out_stride);

/* The same match for ONE pair, for callers that cannot allocate: tight serving loops with a fixed
 * memory budget, harnesses that forbid hipMalloc, programs that must not disturb a HIP graph capture
 * going on elsewhere in the process with an allocation.  (The call itself runs on the context's own
 * streams and takes no stream argument: it is not something to record INTO a caller's graph.)  It
 * has no counterpart in the reference (whose matcher allocates per task,
 * brute_force_matcher.cc:47-52).  The call allocates and frees nothing, on the device, through the
 * HIP runtime or on the host heap: its work space (candidate lists, per-query results, the table of
 * workgroups) is a fixed array of 8 MiB that is part of the library's GPU code, one copy per GPU,
 * held until the process ends; its streams and events were made with the context.  Limits and rules:
 *   - both images hold at most SFM_HIP_NOALLOC_MAX_ROWS descriptors; larger ones are refused
 *     (SFM_HIP_ERR_INVALID), use sfm_hip_match_pairs for them;
 *   - out_lhs_idx / out_rhs_idx / out_dist have room for as many entries as the first image has
 *     descriptors, out_count for one; the kernels write them directly, so they must be pinned host
 *     memory (sfm_hip_host_alloc) or device memory; out_dist may be NULL;
 *   - the results are those of sfm_hip_match_pairs for the same pair, bit for bit;
 *   - asynchronous like sfm_hip_match_pairs (outputs are valid after sfm_hip_sync, or after
 *     sfm_hip_wait_outputs on one of them); calls on one context run one after the other, in call
 *     order, also relative to earlier and later sfm_hip_match_pairs calls that write the same outputs;
 *   - every call on a GPU uses that GPU's one array: calls through two contexts of the same GPU
 *     that overlap in time race (as do two host threads driving one context, which no entry point
 *     allows).  Calls on different GPUs, and in different processes, do not interfere. */
#define SFM_HIP_NOALLOC_MAX_ROWS 8192
int sfm_hip_match_pair_noalloc(sfm_hip_ctx* ctx, int64_t first_image, int64_t second_image, double ratio,
                               int cross_check, int32_t* out_count, int32_t* out_lhs_idx,
                               int32_t* out_rhs_idx, float* out_dist);

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

/* Static partition of a pair list over the GPUs of a node (host-only: no GPU is touched).  The
 * reference runs each pair as an independent task (brute_force_matcher.cc:46-61) over the
 * lexicographic list of ExhaustivePairBuilder (exhaustive_pair_builder.cc:3-19); a multi-GPU caller
 * gives shard s of this partition to GPU s and uploads to that GPU only the images its pairs name.
 * pair_ids: 2*n_pairs image ids as in sfm_hip_match_pairs; image_ids / image_rows: every image the
 * list names and its descriptor count (n_images entries, any order, ids distinct).  A pair costs
 * rows(first) * rows(second); shards get equal cost to within one pair, and each shard is a compact
 * 2-D block of the (first, second) triangle, so it names about 2 n / sqrt(n_shards) images, not n.
 * out_order (n_pairs entries) receives the indices of the pairs, shard after shard; shard s owns
 * out_order[out_shard_begin[s] .. out_shard_begin[s + 1]) (n_shards + 1 entries).  Inside a shard
 * the pairs of one block keep the order of the list.  The result depends on the arguments only. */
int sfm_hip_partition_pairs(const int64_t* pair_ids, int32_t n_pairs, const int64_t* image_ids,
                            const int32_t* image_rows, int32_t n_images, int32_t n_shards,
                            int32_t* out_order, int32_t* out_shard_begin);

/* Wait for everything queued on the context (the reference's future.get(),
 * brute_force_matcher.cc:67-68,83). */
int sfm_hip_sync(sfm_hip_ctx* ctx);

/* Wait until no queued call still has to write the host memory [out, out + bytes) -- typically the
 * out_counts array of one call: that call's outputs are then valid although later calls, writing
 * other memory, are still running.  This is the reference's future.get() for ONE pair's task
 * (brute_force_matcher.cc:67-68), where sfm_hip_sync is the join of all of them: a host loop can
 * keep a call queued behind the running one and still consume results call by call.  Calls of
 * sfm_hip_match_pairs, sfm_hip_match_pair_noalloc and sfm_hip_cascade_match_pairs are covered; the
 * wait may last until a later call on the same internal stream has finished too, never less. */
int sfm_hip_wait_outputs(sfm_hip_ctx* ctx, const void* out, size_t bytes);

/* Pinned host memory for the output buffers. */
void* sfm_hip_host_alloc(size_t bytes);
void sfm_hip_host_free(void* p);

/* Profiling: when enabled, one launch in eight of the distance/top-k kernel (the first after
 * sfm_hip_reset_stats or sfm_hip_get_stats among them) is timed with HIP events on the stream it
 * runs on.  Distance kernels run one after the other on a stream of their own and each batch's short
 * tail kernels on one of two other streams, so a launch never shares the chip with another launch
 * of its kind.  A timed launch that is queued while the launch before it is still unfinished is
 * timed from that launch's completion to its own (both events ride on the kernels' dispatch
 * packets; the microseconds between two launches count in); one that finds the stream idle gets a
 * start event of its own, a packet in front of the kernel that costs a few microseconds.
 * sfm_hip_get_stats() waits for the queued work and folds the durations into distance_ms.
 * enabled = 2 additionally puts an event between every two kernels of a cascade-hash batch and has
 * those kernels count their work (the per-kernel fields of sfm_hip_cascade_stats); each such event
 * is a packet of its own and costs the batch a few microseconds. */
int sfm_hip_set_profiling(sfm_hip_ctx* ctx, int enabled);
int sfm_hip_get_stats(sfm_hip_ctx* ctx, sfm_hip_stats* out);
int sfm_hip_reset_stats(sfm_hip_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Cascade-hash matcher: the GPU side of CascadeHashMatcher::Match
 * (src/feature_matcher/cascade_hash_matcher.hpp:24-45, .cc:118-205), the matcher the reference's
 * driver runs (src/feature_match_main.cc:26).  Images are the ones uploaded with
 * sfm_hip_upload_image.  The order of calls mirrors the reference's Match():
 *   1. sfm_hip_cascade_set_projection   the random projections (.cc:127-146)
 *   2. sfm_hip_cascade_column_sums      each image's share of the mean (.cc:157-163); the caller
 *                                       adds the shares in the reference's order and divides (.cc:165)
 *   3. sfm_hip_cascade_hash_images      ConstructHashDescriptors for every image (.cc:167-170)
 *   4. sfm_hip_cascade_match_pairs      DescriptorMath in both directions + cross-check (.cc:173-195)
 * Sums of floats (the column sums, the 188 projections of a descriptor) are taken in index order
 * with one fp32 accumulator and unfused multiply and add; the reference leaves that order to its
 * matrix library.
 * ------------------------------------------------------------------------------------------ */
#define SFM_HIP_CASCADE_GROUPS 6        /* bucket_group_size, cascade_hash_matcher.cc:122 */
#define SFM_HIP_CASCADE_GROUP_BITS 10   /* bits_of_group, :123 */
#define SFM_HIP_CASCADE_CODE_BITS 128   /* hash_code_size, :124 */

/* Counters of the cascade-hash path since context creation or the last reset. */
typedef struct {
  uint64_t pairs;           /* image pairs matched */
  uint64_t images_hashed;   /* images passed through sfm_hip_cascade_hash_images */
  uint64_t match_launches;  /* batches of sfm_hip_cascade_match_pairs that were timed */
  double match_ms;          /* their summed duration on the GPU, first kernel to last (HIP events on
                               the stream they run on; only while profiling is enabled) */
  double match_bytes;       /* bytes those batches had to read at the least: per pair and bucket group the
                               low halves of both images' hash codes in bucket order (8 bytes per
                               descriptor) and the first image's bucket ids (2 bytes), plus one 16-byte
                               result word per descriptor */
  uint64_t list_overflows;  /* pairs with more close candidate pairs than their list holds, matched by
                               the slower path that needs no list */
  uint64_t workspace_dirty_words; /* non-zero words found in the per-query work space, which every
                               batch leaves all zeros: anything but 0 is a defect */
  /* Of the batches timed with sfm_hip_set_profiling(ctx, 2), per kernel: */
  double walk_ms;           /* k_ch_walk_count */
  double nearest_ms;        /* k_ch_nearest */
  double fallback_ms;       /* k_ch_take_all + k_ch_walk_nearest (both return at once unless a list overflowed) */
  double compact_ms;        /* k_ch_compact */
  uint64_t walk_candidates; /* (first-image descriptor, second-image descriptor, group) triples that share a
                               bucket: what the walk tests, each by the Hamming distance of two 64-bit halves */
  uint64_t close_pairs;     /* entries of the close-pair lists (Hamming distance <= 10) */
  uint64_t exact_distances; /* ... of which either side takes the bin they fall in (.cc:95-105), so that the
                               exact distance of their two 512-byte descriptor rows is computed */
} sfm_hip_cascade_stats;

/* The projection matrices of cascade_hash_matcher.cc:130-146: primary[128][128] (row i produces hash
 * bit i) and secondary[6][10][128] (row k of group t produces bit k of that group's bucket id,
 * first row = most significant bit, .cc:33-35), both row-major host memory.  The reference draws
 * them from std::mt19937 and std::normal_distribution, which belong to the C++ standard library
 * of the caller; that is why they are an input here. */
int sfm_hip_cascade_set_projection(sfm_hip_ctx* ctx, const float* primary, const float* secondary);

/* For each of n_images uploaded images the 128 sums the reference adds into its mean vector
 * (cascade_hash_matcher.cc:158-162): the reference views the row-major n x 128 descriptor block as a
 * column-major matrix, so entry c is the sum of the c-th run of n consecutive floats of the block.
 * out_sums: n_images x 128 floats, host memory.  Blocking. */
int sfm_hip_cascade_column_sums(sfm_hip_ctx* ctx, const int64_t* image_ids, int32_t n_images, float* out_sums);

/* ConstructHashDescriptors (cascade_hash_matcher.cc:11-49) for n_images uploaded images with the
 * vector mean128 subtracted from every descriptor (.cc:21): 128-bit hash codes, six 10-bit bucket
 * ids per descriptor, and per group the descriptors sorted by bucket (ascending index inside a
 * bucket, as .cc:43-46 fills them).  Results stay on the GPU.  Hashing an image again replaces its
 * hashes; uploading an image again discards them.  Images of more than 262,144 descriptors are
 * refused.  Blocking. */
int sfm_hip_cascade_hash_images(sfm_hip_ctx* ctx, const int64_t* image_ids, int32_t n_images, const float* mean128);

/* Copy one image's hashes to the host: codes n x 4 words (hash bit j is bit j % 32 of word j / 32),
 * bucket_ids n x 6.  Either pointer may be NULL.  For tests and inspection. */
int sfm_hip_cascade_get_hashes(sfm_hip_ctx* ctx, int64_t image_id, uint32_t* codes, uint16_t* bucket_ids);

/* Match n_pairs pairs of hashed images: DescriptorMath (cascade_hash_matcher.cc:51-116) from the
 * first image to the second and, with cross_check, back, then the filter of .cc:182-193.  For pair p
 * the surviving matches, ascending in lhs index, are written to out_lhs_idx / out_rhs_idx /
 * out_dist[p*out_stride + m] for m < out_counts[p], as in sfm_hip_match_pairs (out_dist: the L2
 * distance of the match, src/metric.hpp:15-22; may be NULL).  out_defaults[p] receives the number of
 * value-initialised records {lhs_idx 0, rhs_idx 0} the reference stores IN FRONT of those matches:
 * with cross_check it builds the result in `Matches temp(matches.size())` (.cc:187), a vector that
 * already holds one default record per forward match, and appends to it; without cross_check the
 * number is 0.  Asynchronous like sfm_hip_match_pairs; outputs are valid after sfm_hip_sync(). */
int sfm_hip_cascade_match_pairs(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, int cross_check,
                                int32_t* out_defaults, int32_t* out_counts, int32_t* out_lhs_idx,
                                int32_t* out_rhs_idx, float* out_dist, int64_t out_stride);

/* Host-only helpers that reproduce what the reference leaves to the C++ standard library, for
 * callers that are not written in C++ (a C++ caller can equally use <random> and
 * <unordered_set> itself, as sfm_amd/host/hip_cascade_hash_matcher.cc does):
 *   reference_projection  the matrices of cascade_hash_matcher.cc:127-146 -- std::mt19937 with its
 *                         default seed feeding one std::normal_distribution<> (mean 0, deviation 1),
 *                         128 x 128 draws for primary, then 6 x 10 x 128 for secondary;
 *   image_order           the image ids of a pair list (2*n_pairs values, in the order of the
 *                         std::set the reference walks) in the order in which the reference's
 *                         std::unordered_set hands them out (.cc:148-152, :157); out_ids needs room
 *                         for 2*n_pairs values, *out_n receives how many were written.
 * Both use the standard library this shared library was built with. */
int sfm_hip_cascade_reference_projection(float* primary, float* secondary);
int sfm_hip_cascade_image_order(const int64_t* pair_ids, int32_t n_pairs, int64_t* out_ids, int32_t* out_n);

/* Everything CascadeHashMatcher::Match does before its pair loop (cascade_hash_matcher.cc:122-171)
 * for the uploaded images a pair list names: reference projection, column sums of every image,
 * the mean vector (shares added in image_order, divided by the total descriptor count) and the
 * hashes of every image.  out_mean128 (128 floats, may be NULL) receives the mean.  Blocking.
 * The reference projection is drawn and installed by the context's first call and kept; a call
 * that follows sfm_hip_cascade_set_projection installs it again. */
int sfm_hip_cascade_prepare(sfm_hip_ctx* ctx, const int64_t* pair_ids, int32_t n_pairs, float* out_mean128);

/* Statistics of the cascade path; timing as sfm_hip_set_profiling describes. */
int sfm_hip_cascade_get_stats(sfm_hip_ctx* ctx, sfm_hip_cascade_stats* out);
int sfm_hip_cascade_reset_stats(sfm_hip_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Geometric verification of matches: the GPU side of ComputeFundamentalMatrix
 * (src/sfm_main.cc:17-50) as GeometryFilter (src/sfm_main.cc:259-282) calls it once per image pair,
 * i.e. Ransac<EightPointFundamentalSolverImpl, EpipolarLineError>::Inference (src/ransac.hpp:94-150,
 * src/solver/fundamental_solver.cc:13-86, src/solver/fundamental_solver.hpp:59-78) followed by the
 * classification of every match under the model found (sfm_main.cc:32-42).
 *
 * Pair p owns the matches [match_offsets[p], match_offsets[p + 1]); lhs_xy / rhs_xy hold their pixel
 * positions in the first / second image, two doubles per match (the observations of the reference's
 * Match records).  Outputs, host memory: out_F 9 doubles per pair, row-major, the model RANSAC ended
 * with (zeros for a pair with fewer than eight matches, which the reference cannot sample);
 * out_inlier_flags one byte per match, 1 where the epipolar-line error under that model is below
 * 3.84; out_inlier_counts their number per pair -- ComputeFundamentalMatrix returns true when it
 * exceeds 30 (sfm_main.cc:44); out_rounds (may be NULL) the RANSAC rounds run.
 *
 * The reference seeds its sampler from std::random_device in every round, so its result differs from
 * run to run; here round r of pair p reads a counter-based random stream keyed by
 * (seed, first_pair_index + p, r), which makes a run repeatable and a pair's result independent of
 * how the pair list is cut into calls.  The singular vectors the reference takes from its matrix
 * library's SVD come from a Jacobi eigen-solver; all arithmetic is fp64.  Blocking. */
int sfm_hip_geometry_verify_pairs(sfm_hip_ctx* ctx, int32_t n_pairs, const int64_t* match_offsets, const double* lhs_xy,
                                  const double* rhs_xy, uint64_t seed, int64_t first_pair_index, double* out_F,
                                  uint8_t* out_inlier_flags, int32_t* out_inlier_counts, int32_t* out_rounds);

/* Counters of the geometric verification since context creation or the last reset.  The work
 * counters and the durations are collected only while profiling is enabled (sfm_hip_set_profiling):
 * every launch (256 rounds of every unfinished pair, later launches up to 2,048) is then bracketed with HIP events on the stream it runs on, one in
 * front of the fit, one between fit and count, one behind the count. */
typedef struct {
  uint64_t calls;          /* calls of sfm_hip_geometry_verify_pairs */
  uint64_t pairs;          /* pairs with at least eight matches among them */
  uint64_t timed_launches; /* launches (fit + count of 256 to 2,048 rounds of every unfinished pair) that were timed */
  uint64_t rounds;         /* RANSAC rounds those launches sampled, fitted and counted: more than the
                              rounds the pairs report, since a launch works on 256 or more rounds of a pair at once
                              and the rounds behind the pair's stopping point are discarded */
  uint64_t match_rounds;   /* (match, round) pairs the count measured: sum over those rounds of the pair's matches */
  double fit_ms;           /* summed duration of the timed launches' k_geo_sample + k_geo_fit (samples, then models) */
  double count_ms;         /* ... and of their k_geo_count */
} sfm_hip_geometry_stats;
int sfm_hip_geometry_get_stats(sfm_hip_ctx* ctx, sfm_hip_geometry_stats* out);
int sfm_hip_geometry_reset_stats(sfm_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H_ */
