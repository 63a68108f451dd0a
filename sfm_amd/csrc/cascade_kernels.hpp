// gfx950 kernels of the cascade-hash matcher (the GPU side of the reference's
// src/feature_matcher/cascade_hash_matcher.cc).
//
// Per image, once (ConstructHashDescriptors, .cc:11-49):
//   k_ch_colsum   the image's share of the mean vector (.cc:160-162)
//   k_ch_hash     188 projections of every centred descriptor -> 128-bit code + six 10-bit bucket ids
//   k_ch_group    per group the hash records sorted by bucket (stable: ascending index inside a
//                 bucket, the order .cc:43-46 fills them in), laid out as four arrays -- low halves
//                 of the codes, high halves, the group's own bucket id, all bucket ids + index --
//                 + bucket offsets
// Per pair (DescriptorMath both ways + cross-check, .cc:51-116, :173-195):
//   k_ch_walk_count      walk every (first-image descriptor, second-image descriptor) that share a
//                        bucket; those within Hamming distance 10 ("close pairs") are counted into
//                        both descriptors' Hamming histograms and appended to the pair's list
//   k_ch_nearest         per close pair of the list: how many Hamming bins the reference would take
//                        for either descriptor (.cc:95-103), whether that gives it the two
//                        candidates it needs (.cc:105), and if the pair is inside those bins its
//                        exact L2 distance, folded into a running minimum of (distance, index)
//   k_ch_compact         cross-check, ordered compaction, counts; puts the work space back to zero
// and, only for a pair whose close pairs did not fit its list (whole images of near-identical
// descriptors), the same result without the list:
//   k_ch_take_all        the bins decision for every query of the pair
//   k_ch_walk_nearest    the walk again, distances computed on the way
//
// Why the walk is by bucket and not by query as in the reference: "i and j share a bucket of group g"
// and the Hamming distance are symmetric, so one walk serves both directions, and with the records
// of a group stored in bucket order both sides of it are read as contiguous runs -- the pass is a
// stream over 6 x (na + nb) records, bound by memory, instead of a gather per query.
// What the reference derives from visiting order (first-seen de-duplication, the order inside a
// Hamming bin) does not reach its result: bins are taken whole and the winner is the minimum of
// (distance, index) pairs.
//
// The per-query work space is all zeros between batches: every kernel that dirties a word is
// followed by one that restores it (k_ch_compact), so a batch neither clears nor reads the slots of
// the (vast majority of) queries that have no close pair.
#ifndef SFM_AMD_CSRC_CASCADE_KERNELS_HPP_
#define SFM_AMD_CSRC_CASCADE_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sfm_cascade {

// Pointers that arrive through a table in memory are generic to the compiler, and a generic ("flat")
// access counts against the LDS counter as well as the memory counter, so that every wait for an
// LDS read also waits for the loads and atomics in flight.  The walk's go through global pointers.
#define SFM_GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const SFM_GLOBAL_AS T* as_global(const T* p) { return (const SFM_GLOBAL_AS T*)p; }
template <class T>
__device__ __forceinline__ SFM_GLOBAL_AS T* as_global(T* p) { return (SFM_GLOBAL_AS T*)p; }
// (HIP's uint2 / uint4 are classes whose copy constructors take generic references: 8- and 16-byte
// accesses go through the compiler's native vector types)
typedef uint32_t sfm_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t sfm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 global_load(const uint2* p) {
  const sfm_u32x2 v = *reinterpret_cast<const SFM_GLOBAL_AS sfm_u32x2*>(as_global(p));
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 global_load(const uint4* p) {
  const sfm_u32x4 v = *reinterpret_cast<const SFM_GLOBAL_AS sfm_u32x4*>(as_global(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void global_add_u32(uint32_t* p, uint32_t v) {   // no value comes back: nothing to wait for
  __hip_atomic_fetch_add(as_global(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int DIM = 128;
constexpr int GROUPS = 6;                 // bucket_group_size, cascade_hash_matcher.cc:122
constexpr int GROUP_BITS = 10;            // bits_of_group, :123
constexpr int CODE_BITS = 128;            // hash_code_size, :124
constexpr int BUCKETS = 1 << GROUP_BITS;  // group_size, :125
constexpr int NPROJ = CODE_BITS + GROUPS * GROUP_BITS;   // 188 projection rows
constexpr int MAX_HAMMING = GROUP_BITS;   // .cc:95: bins 0 .. bits_of_group are ever looked at
constexpr int SHORT_LIST = 10;            // kNumTopCandidates, .cc:57

// One descriptor's hashes.  bids[w] holds the bucket ids of groups 2w (low half) and 2w+1 (high half).
struct alignas(32) HashRec {
  uint32_t code[4];
  uint32_t bids[3];
  uint32_t idx;      // index of the descriptor in its image
};
static_assert(sizeof(HashRec) == 32, "hash records are read as two 16-byte words");

__device__ __forceinline__ uint32_t bucket_of(const HashRec& r, int g) { return (r.bids[g >> 1] >> ((g & 1) * 16)) & 0xFFFFu; }

// Bucket-ordered copy of a record, second half: the bucket ids and the index.
struct alignas(16) HashMeta {
  uint32_t bids[3];
  uint32_t idx;
};

// Per-query work space of one pair and direction, all zero between batches: a histogram row of 12
// words (bins 0..10 of close candidates by Hamming distance, one word unused) and, in an array of
// its own because the compaction reads nothing else, a result of 4 words.
#ifndef SFM_HIST_WORDS   // (overridden in timing experiments: tools/build_variant.sh)
#define SFM_HIST_WORDS 12
#endif
constexpr int HIST_WORDS = SFM_HIST_WORDS;
constexpr int RES_WORDS = 4;
constexpr int RES_STATE = 0;       // 1: seen, but no match; 2 + h: match, bins 0..h taken; 0: not seen, or not stored (k_ch_nearest
                                   // stores it only where no distance will mark the slot: touched = state or minimum non-zero)
constexpr int RES_BEST = 2;        // [2, 3] 64-bit maximum of ~(distance bits << 32 | index): the minimum, inverted
constexpr int SLOT_WORDS = HIST_WORDS + RES_WORDS;   // per query
constexpr uint32_t STATE_NONE = 1;
constexpr uint32_t STATE_MATCH = 2;

constexpr int LIST_H_SHIFT = 28;   // list entries: {i, j | h << 28}
constexpr int MAX_IMAGE_ROWS = 1 << 18;   // k_ch_nearest lays out one count per 256 first-image records and group in LDS

struct ImageRef {
  const float* d;
  int32_t n;
  int32_t pad_;
};

struct HashJob {
  const float* d;        // [n][128]
  HashRec* recs;         // [n]
  uint2* glo;            // [6][n] low 64 bits of the hash codes in group g's bucket order
  uint2* ghi;            // [6][n] high 64 bits, same order
  uint16_t* gbkt;        // [6][n] group g's bucket id, same order
  HashMeta* gmeta;       // [6][n] all bucket ids and the index, same order
  int32_t* offsets;      // [6][1025]
  int32_t n;
  int32_t first_block;   // k_ch_hash: the job's first workgroup in the launch
};

struct CascadePair {
  // first image, per group in that group's bucket order ([6][na] each): the low halves of the codes
  // and the group's bucket id are what every record of the walk reads; the high halves and the full
  // bucket ids + index only for the few candidates that pass the first test
  const uint2* a_lo;
  const uint2* a_hi;
  const uint16_t* a_bkt;
  const HashMeta* a_meta;
  // second image ([6][nb] each)
  const uint2* b_lo;
  const uint2* b_hi;
  const HashMeta* b_meta;
  const int32_t* offb;   // second image: bucket offsets [6][1025]
  const float* da;       // descriptor rows
  const float* db;
  uint32_t* hist_a;      // [na][12]
  uint32_t* hist_b;      // [nb][12]
  uint32_t* res_a;       // [na][4]
  uint32_t* res_b;       // [nb][4]
  uint2* list;           // close pairs {i, j | h << 28}: one segment of LIST_SEGMENT entries per walk workgroup
  uint32_t* seg_counts;  // entries in each segment; rewritten by every walk
  uint32_t* overflow;    // non-zero: some segment was too small, the list is incomplete (zero between batches)
  int32_t na, nb;
  int32_t out_slot;
  int32_t pad_;
};

// ---------------------------------------------------------------------------------------------
// The reference's share of the mean per image (cascade_hash_matcher.cc:160-162): it views the
// row-major block as a column-major n x 128 matrix, so "column" c is the c-th run of n consecutive
// floats.  One lane per run, one fp32 accumulator, index order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_ch_colsum(const ImageRef* __restrict__ images, float* __restrict__ out) {
#pragma clang fp contract(off)
  const ImageRef im = images[blockIdx.x];
  const int c = threadIdx.x;
  const float* p = im.d + (size_t)c * (size_t)im.n;
  float s = 0.0f;
  int r = 0;
  // Every lane is on a cache line of its own, so what a load instruction costs is its 64 lines whatever
  // it brings: 16 bytes per lane and instruction, once the lane's run has reached a 16-byte boundary.
  const int head = min(im.n, (int)((4u - (uint32_t)(((size_t)c * (size_t)im.n) & 3u)) & 3u));
  for (; r < head; r++) s = s + p[r];
  // and the next 32 values are on their way while these 32 are added up (the chain of additions is
  // the work; a trip to memory per block in front of it was nine tenths of the kernel)
  constexpr int B = 8;   // float4 per block
  float4 v[B], w[B];
  if (r + 4 * B <= im.n) {
#pragma unroll
    for (int k = 0; k < B; k++) v[k] = *reinterpret_cast<const float4*>(p + r + 4 * k);
  }
  for (; r + 4 * B <= im.n; r += 4 * B) {
    const bool more = r + 8 * B <= im.n;
    if (more) {
#pragma unroll
      for (int k = 0; k < B; k++) w[k] = *reinterpret_cast<const float4*>(p + r + 4 * B + 4 * k);
    }
#pragma unroll
    for (int k = 0; k < B; k++) {
      s = s + v[k].x;
      s = s + v[k].y;
      s = s + v[k].z;
      s = s + v[k].w;
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < B; k++) v[k] = w[k];
    }
  }
  for (; r < im.n; r++) s = s + p[r];
  out[(size_t)blockIdx.x * DIM + c] = s;
}

// ---------------------------------------------------------------------------------------------
// Hashing (cascade_hash_matcher.cc:18-38).  A workgroup takes 64 descriptors, one per lane; its
// four wavefronts each compute 48 of the 188 (padded to 192) projections of those descriptors, so a
// wavefront needs few registers and eight of them share a SIMD, which is what hides the latency
// of the scalar loads the projection rows arrive through (they are the same for every lane).
// Each projection is the sum over j = 0..127 of row[j] * d[j] in that order, product and sum
// rounded separately.
// proj holds the rows transposed and padded, proj[j * 192 + r] = row r, entry j (rows 0..127
// primary, 128..187 the 6 x 10 secondary, 188..191 zero), followed by the 128-float mean.
// ---------------------------------------------------------------------------------------------
constexpr int PROJ_PADDED = 192;
constexpr int PROJ_PER_WAVE = PROJ_PADDED / 4;    // 48
constexpr int PROJ_MEAN_AT = DIM * PROJ_PADDED;   // where the mean starts

__global__ __launch_bounds__(256) void k_ch_hash(const HashJob* __restrict__ jobs, int n_jobs,
                                                 const float* __restrict__ proj) {
#pragma clang fp contract(off)
  __shared__ unsigned long long signs[4][64];
  // which job this workgroup belongs to: the last one whose first_block is not beyond it
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= jobs[mid].first_block) lo = mid;
    else hi = mid - 1;
  }
  const HashJob job = jobs[lo];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = ((int)blockIdx.x - job.first_block) * 64 + lane;
  const bool live = row < job.n;
  const float* mean = proj + PROJ_MEAN_AT;
  const float4* src = reinterpret_cast<const float4*>(job.d + (size_t)(live ? row : 0) * DIM);
  const float* rows = proj + wv * PROJ_PER_WAVE;
  float acc[PROJ_PER_WAVE];
#pragma unroll
  for (int k = 0; k < PROJ_PER_WAVE; k++) acc[k] = 0.0f;
#pragma unroll 1
  for (int j0 = 0; j0 < DIM; j0 += 4) {
    const float4 v = src[j0 >> 2];
    const float d[4] = {v.x - mean[j0], v.y - mean[j0 + 1], v.z - mean[j0 + 2], v.w - mean[j0 + 3]};   // .cc:21
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float* p = rows + (size_t)(j0 + j) * PROJ_PADDED;
#pragma unroll
      for (int k = 0; k < PROJ_PER_WAVE; k++) acc[k] = acc[k] + p[k] * d[j];
    }
  }
  unsigned long long bits = 0;
#pragma unroll
  for (int k = 0; k < PROJ_PER_WAVE; k++) bits |= (unsigned long long)(acc[k] > 0.0f ? 1 : 0) << k;
  signs[wv][lane] = bits;
  __syncthreads();
  if (wv != 0 || !live) return;
  // sign bits of the 192 projections, row r at bit r % 32 of word r / 32
  const unsigned long long s0 = signs[0][lane], s1 = signs[1][lane], s2 = signs[2][lane], s3 = signs[3][lane];
  uint32_t word[6];
  word[0] = (uint32_t)s0;                                     // rows   0.. 31
  word[1] = (uint32_t)(s0 >> 32) | (uint32_t)(s1 << 16);      // rows  32.. 63 = s0[32..47], s1[0..15]
  word[2] = (uint32_t)(s1 >> 16);                             // rows  64.. 95 = s1[16..47]
  word[3] = (uint32_t)s2;                                     // rows  96..127 = s2[0..31]
  word[4] = (uint32_t)(s2 >> 32) | (uint32_t)(s3 << 16);      // rows 128..159 = s2[32..47], s3[0..15]
  word[5] = (uint32_t)(s3 >> 16);                             // rows 160..191 = s3[16..47]
  HashRec rec;
#pragma unroll
  for (int w = 0; w < 4; w++) rec.code[w] = word[w];                // .cc:24-26
  // .cc:31-37: group g is made of secondary rows 10g .. 10g+9, the first row ending up as the most
  // significant bit of the bucket id
  const unsigned long long sec = (unsigned long long)word[4] | ((unsigned long long)word[5] << 32);
  uint32_t bid[GROUPS];
#pragma unroll
  for (int g = 0; g < GROUPS; g++) bid[g] = __brev((uint32_t)(sec >> (g * GROUP_BITS)) & (BUCKETS - 1)) >> (32 - GROUP_BITS);
  rec.bids[0] = bid[0] | (bid[1] << 16);
  rec.bids[1] = bid[2] | (bid[3] << 16);
  rec.bids[2] = bid[4] | (bid[5] << 16);
  rec.idx = (uint32_t)row;
  job.recs[row] = rec;
}

// ---------------------------------------------------------------------------------------------
// Bucket order (cascade_hash_matcher.cc:40-47): a stable counting sort of the records by the bucket
// id of group g.  One wavefront per (group, image), 64 records per step; the lanes of a step that
// share a bucket find each other with ten ballots, so ranks need no atomics and the order inside
// a bucket is ascending index.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t same_bucket_mask(uint32_t b, bool valid) {
  uint64_t m = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < GROUP_BITS; bit++) {
    const bool one = (b >> bit) & 1u;
    const uint64_t bal = __ballot(valid && one);
    m &= one ? bal : ~bal;
  }
  return m;
}

__global__ __launch_bounds__(64) void k_ch_group(const HashJob* __restrict__ jobs) {
  __shared__ int cur[BUCKETS];
  const HashJob job = jobs[blockIdx.y];
  const int g = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = job.n;
  const uint64_t below = ((uint64_t)1 << lane) - 1;
  for (int b = lane; b < BUCKETS; b += 64) cur[b] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool valid = i < n;
    const uint32_t b = valid ? bucket_of(job.recs[i], g) : 0u;
    const uint64_t same = same_bucket_mask(b, valid);
    if (valid && (same & below) == 0) cur[b] += __popcll(same);   // lowest lane of its bucket
    __syncthreads();
  }
  // exclusive scan of the 1024 counts: 16 consecutive buckets per lane
  int local[16];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    local[k] = sum;
    sum += cur[lane * 16 + k];
  }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  const int before = incl - sum;
  __syncthreads();
  int32_t* off = job.offsets + (size_t)g * (BUCKETS + 1);
#pragma unroll
  for (int k = 0; k < 16; k++) {
    cur[lane * 16 + k] = before + local[k];
    off[lane * 16 + k] = before + local[k];
  }
  if (lane == 63) off[BUCKETS] = n;
  __syncthreads();
  uint2* out_lo = job.glo + (size_t)g * (size_t)n;
  uint2* out_hi = job.ghi + (size_t)g * (size_t)n;
  uint16_t* out_bkt = job.gbkt + (size_t)g * (size_t)n;
  HashMeta* out_meta = job.gmeta + (size_t)g * (size_t)n;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool valid = i < n;
    HashRec rec;
    if (valid) rec = job.recs[i];
    const uint32_t b = valid ? bucket_of(rec, g) : 0u;
    const uint64_t same = same_bucket_mask(b, valid);
    int pos = 0;
    if (valid) pos = cur[b] + __popcll(same & below);
    __syncthreads();
    if (valid) {
      out_lo[pos] = make_uint2(rec.code[0], rec.code[1]);
      out_hi[pos] = make_uint2(rec.code[2], rec.code[3]);
      out_bkt[pos] = (uint16_t)b;
      HashMeta m;
      m.bids[0] = rec.bids[0];
      m.bids[1] = rec.bids[1];
      m.bids[2] = rec.bids[2];
      m.idx = rec.idx;
      out_meta[pos] = m;
      if ((same >> lane) >> 1 == 0) cur[b] = pos + 1;   // highest lane of its bucket
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// src/metric.hpp:15-22: one fp32 accumulator, index order, subtract / multiply / add each rounded
// on its own, then the square root.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float exact_l2(const float* __restrict__ a, const float* __restrict__ b) {
#pragma clang fp contract(off)
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float sum = 0.0f;
#pragma unroll 4
  for (int k = 0; k < DIM / 4; k++) {
    const float4 u = a4[k], v = b4[k];
    float t;
    t = u.x - v.x; sum = sum + t * t;
    t = u.y - v.y; sum = sum + t * t;
    t = u.z - v.z; sum = sum + t * t;
    t = u.w - v.w; sum = sum + t * t;
  }
  return sqrtf(sum);
}

// ---------------------------------------------------------------------------------------------
// cascade_hash_matcher.cc:95-105 for one query, from its Hamming histogram: bins are taken whole,
// nearest first, no new bin once ten candidates are in, none beyond distance 10; a match needs two
// candidates.  Returns the slot state: STATE_NONE, or STATE_MATCH + the last bin taken.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t take_state(const uint32_t* __restrict__ slot) {
  const uint4 h0 = reinterpret_cast<const uint4*>(slot)[0];
  const uint4 h1 = reinterpret_cast<const uint4*>(slot)[1];
  const uint4 h2 = reinterpret_cast<const uint4*>(slot)[2];
  const uint32_t hist[MAX_HAMMING + 1] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z};
  uint32_t taken = 0, last = 0;
#pragma unroll
  for (int h = 0; h <= MAX_HAMMING; h++) {
    if (taken < (uint32_t)SHORT_LIST) {
      taken += hist[h];
      last = h;
    }
  }
  return taken >= 2 ? STATE_MATCH + last : STATE_NONE;
}

__device__ __forceinline__ bool inside_taken_bins(uint32_t state, uint32_t h) { return state >= STATE_MATCH + h; }

// One close pair inside the bins of a query: its exact distance into the running minima.
// (a - b)^2 and (b - a)^2 are the same float: one distance serves both directions.
__device__ __forceinline__ void fold_distance(const CascadePair& P, uint32_t i, uint32_t j, bool into_a, bool into_b) {
  const float dist = exact_l2(P.da + (size_t)i * DIM, P.db + (size_t)j * DIM);
  const unsigned long long hi = (unsigned long long)__float_as_uint(dist) << 32;
  if (into_a) atomicMax(reinterpret_cast<unsigned long long*>(P.res_a + (size_t)i * RES_WORDS + RES_BEST), ~(hi | j));
  if (into_b) atomicMax(reinterpret_cast<unsigned long long*>(P.res_b + (size_t)j * RES_WORDS + RES_BEST), ~(hi | i));
}

// ---------------------------------------------------------------------------------------------
// The walk over bucket-sharing descriptor pairs.  A workgroup takes 256 consecutive records of the
// first image in group g's bucket order, one per thread; the buckets they sit in are consecutive,
// so the records of the second image they have to meet form ONE contiguous run, which the
// workgroup stages in LDS, and every thread scans the part of it that is its own bucket.
// The walk is bound by the bytes it streams (its two loading phases run at 6-8 TB/s), so it streams
// as few as it can: per record of either image only the low 64 bits of the hash code, plus two
// bytes of bucket id on the first image's side.  Two codes that differ in more than ten of those
// 64 bits (practically every candidate) are done after two XORs and two population counts, four
// candidates per step; the candidates that pass are queued for the whole workgroup and finished
// after the chunk's scan, one per thread, from the high halves, bucket ids and indices in memory.
// The close pairs found go to a segment of the pair's list that belongs to this workgroup alone:
// nothing is reserved through an atomic and nothing waits for one.
// A pair that shares buckets in several groups is handled in the lowest such group only: that is
// the reference's "seen" map (cascade_hash_matcher.cc:84-86) without its visiting order.
//   COUNT    close pairs (Hamming distance <= 10; .cc:95 never reads further) go into both
//            descriptors' histograms (.cc:88-91) and into the workgroup's list segment
//   NEAREST  only for a pair whose list overflowed: distances computed on the way (states from
//            k_ch_take_all)
// ---------------------------------------------------------------------------------------------
constexpr int WALK_COUNT = 0;
constexpr int WALK_NEAREST = 1;
constexpr int WALK_THREADS = 256;
constexpr int WALK_CHUNK = 512;      // staged records of the second image per step
constexpr int LIST_SEGMENT = 128;    // close pairs a workgroup can report

constexpr uint32_t WORK_SLOTS = 1024; // counters the statistics of a walk are spread over
constexpr int WALK_STEP = 4;         // candidates a lane tests per step of the scan
constexpr int WALK_QUEUE = 2048;     // candidates of a workgroup and staged chunk that pass the first test and wait for the second

struct WalkShared {
  uint2 lo[WALK_CHUNK + WALK_STEP];   // low halves of the staged codes (the scan reads in fours, also across the end)
  uint2 own[WALK_THREADS];            // low halves of the workgroup's own records, by thread
  uint2 found[LIST_SEGMENT];
  uint32_t queue[WALK_QUEUE];         // owner thread << 16 | staged index
  int range[2];
  uint32_t n_found;
  uint32_t n_queued;
};

// Second half of the test for one candidate that passed the first: the rest of both records
// (high halves, bucket ids, indices) from memory.
// a_at: position of thread 0's record in the first image's bucket order; b_at: of staged index 0.
template <int MODE>
__device__ __forceinline__ void finish_candidate(const CascadePair& P, int g, int both_directions, WalkShared& sh,
                                                 uint32_t entry, size_t a_at, size_t b_at) {
  const int owner = (int)(entry >> 16), r = (int)(entry & 0xFFFFu);
#ifdef SFM_WALK_LOADS_AT_ONCE   // timing experiments: all four loads of a candidate in one trip to memory
  const uint4 ma4 = global_load(reinterpret_cast<const uint4*>(P.a_meta) + (a_at + owner));
  const uint4 mb4 = global_load(reinterpret_cast<const uint4*>(P.b_meta) + (b_at + r));
  uint2 ah = global_load(P.a_hi + (a_at + owner)), bh = global_load(P.b_hi + (b_at + r));
  asm volatile("" : "+v"(ah.x), "+v"(ah.y), "+v"(bh.x), "+v"(bh.y));   // keeps the last two loads in front of the test below
#else
  // the bucket ids first: a pair of near-duplicates shares most of its six buckets and is dropped
  // here in all groups but the lowest, before the high halves of its codes are fetched (fetching all
  // four at once saves a trip to memory and 3 % of the kernel's time, for a third more traffic: 0.46 against 0.34 GB per batch)
  const uint4 ma4 = global_load(reinterpret_cast<const uint4*>(P.a_meta) + (a_at + owner));
  const uint4 mb4 = global_load(reinterpret_cast<const uint4*>(P.b_meta) + (b_at + r));
#endif
  // a pair of near-duplicates shares most of its six buckets and is counted in the lowest such group only
  const uint32_t x0 = ma4.x ^ mb4.x, x1 = ma4.y ^ mb4.y, x2 = ma4.z ^ mb4.z;
  bool earlier = false;   // shares a bucket in a group before g: counted there
  if (g > 0) earlier |= (x0 & 0xFFFFu) == 0;
  if (g > 1) earlier |= (x0 >> 16) == 0;
  if (g > 2) earlier |= (x1 & 0xFFFFu) == 0;
  if (g > 3) earlier |= (x1 >> 16) == 0;
  if (g > 4) earlier |= (x2 & 0xFFFFu) == 0;
  if (earlier) return;
#ifndef SFM_WALK_LOADS_AT_ONCE
  const uint2 ah = global_load(P.a_hi + (a_at + owner)), bh = global_load(P.b_hi + (b_at + r));
#endif
  const uint2 al = sh.own[owner], bl = sh.lo[r];
  const uint32_t h = __popc(al.x ^ bl.x) + __popc(al.y ^ bl.y) + __popc(ah.x ^ bh.x) + __popc(ah.y ^ bh.y);
  if (h > (uint32_t)MAX_HAMMING) return;
  const uint32_t i = ma4.w, j = mb4.w;
#if defined(SFM_ABLATE_WALK_UPTO) && SFM_ABLATE_WALK_UPTO == 5   // timing experiments only: finished, nothing recorded
  if (ma4.x != 0x12345678u) return;
#endif
  if (MODE == WALK_COUNT) {
    global_add_u32(P.hist_a + (size_t)i * HIST_WORDS + h, 1u);
    if (both_directions) global_add_u32(P.hist_b + (size_t)j * HIST_WORDS + h, 1u);
    const uint32_t at = atomicAdd(&sh.n_found, 1u);
    if (at < (uint32_t)LIST_SEGMENT) sh.found[at] = make_uint2(i, j | (h << LIST_H_SHIFT));
  } else {
    const bool fa = inside_taken_bins(P.res_a[(size_t)i * RES_WORDS + RES_STATE], h);
    const bool fb = both_directions && inside_taken_bins(P.res_b[(size_t)j * RES_WORDS + RES_STATE], h);
    if (fa || fb) fold_distance(P, i, j, fa, fb);
  }
}

// The work of workgroup bx of nbx for (pair P, group g).
// work (may be null): WORK_SLOTS counters that receive the number of candidates tested (statistics).
template <int MODE>
__device__ __forceinline__ void walk_block(const CascadePair& P, int g, int bx, int nbx, int both_directions,
                                           WalkShared& sh, unsigned long long* __restrict__ work = nullptr) {
  uint2* stage_lo = sh.lo;
  int* range = sh.range;
  const int na = P.na, nb = P.nb;
  const int tid = threadIdx.x;
  const int p0 = bx * WALK_THREADS;
  const uint32_t seg = (uint32_t)g * (uint32_t)nbx + (uint32_t)bx;
  if (p0 >= na) {
    if (MODE == WALK_COUNT && tid == 0) P.seg_counts[seg] = 0;
    return;
  }
  const size_t a_group = (size_t)g * (size_t)na, b_group = (size_t)g * (size_t)nb;
  const uint2* b_lo = P.b_lo + b_group;
  const int32_t* off = P.offb + (size_t)g * (BUCKETS + 1);
  const int p = p0 + tid;
  const bool valid = p < na;
  uint2 lo_half = make_uint2(0, 0);
  int s = 0, e = 0;
  if (valid) {
    lo_half = P.a_lo[a_group + p];
    const uint32_t b = P.a_bkt[a_group + p];
    s = off[b];
    e = off[b + 1];
  }
  sh.own[tid] = lo_half;
  if (tid == 0) {
    range[0] = s;
    sh.n_found = 0;
    sh.n_queued = 0;
  }
  if (p == min(p0 + WALK_THREADS - 1, na - 1)) range[1] = e;
  if (work) {   // statistics only: candidates this wavefront will test, into one of WORK_SLOTS counters
    unsigned long long mine = (unsigned long long)(e - s);   // (a single counter would serialise the launch's atomics)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((tid & 63) == 0 && mine) atomicAdd(work + ((seg * 4u + (uint32_t)(tid >> 6) + (uint32_t)P.out_slot * 37u) & (WORK_SLOTS - 1)), mine);
  }
  __syncthreads();
#if defined(SFM_ABLATE_WALK_UPTO) && SFM_ABLATE_WALK_UPTO <= 1   // timing experiments only (tools/build_variant.sh)
  const int lo = 0, hi = lo_half.x == 0x12345678u ? range[1] : 0;
#else
  const int lo = range[0], hi = range[1];
#endif
  const size_t a_at = a_group + (size_t)p0;   // thread 0's record
  for (int c0 = lo; c0 < hi; c0 += WALK_CHUNK) {
    const int cnt = min(WALK_CHUNK, hi - c0);
    for (int r = tid; r < cnt; r += WALK_THREADS) stage_lo[r] = b_lo[c0 + r];
    __syncthreads();
    // First test, every candidate: the low 64 bits, four candidates per step.  The few that pass are
    // queued for the whole workgroup and finished after the scan, one per thread: a wavefront that
    // finished its own few on the spot would spend a whole pass of 64 lanes on them, four times per
    // workgroup.
#if defined(SFM_ABLATE_WALK_UPTO) && SFM_ABLATE_WALK_UPTO <= 2   // timing experiments only
    const int q1 = stage_lo[tid].x == 0x12345678u ? e : 0;
#else
    const int q1 = min(e, c0 + cnt);
#endif
    const size_t b_at = b_group + (size_t)c0;
    for (int q = max(s, c0); q < q1; q += WALK_STEP) {
      const int r = q - c0;
      // Nothing but the four distances on the way that nearly every step takes: which of the four are
      // real (a bucket seldom ends on a multiple of four; what lies behind it are the next bucket's
      // codes) is only asked when one of them passes.
      uint32_t h[WALK_STEP];
#pragma unroll
      for (int k = 0; k < WALK_STEP; k++) {
        const uint2 t = stage_lo[r + k];
        h[k] = __popc(lo_half.x ^ t.x) + __popc(lo_half.y ^ t.y);
      }
      const uint32_t nearest = min(min(h[0], h[1]), min(h[2], h[3]));
#if defined(SFM_ABLATE_WALK_UPTO) && SFM_ABLATE_WALK_UPTO <= 3   // timing experiments only
      if (nearest > (uint32_t)MAX_HAMMING || lo_half.x != 0x12345678u) continue;
#else
      if (nearest > (uint32_t)MAX_HAMMING) continue;
#endif
#pragma unroll
      for (int k = 0; k < WALK_STEP; k++) {
        if (q + k < q1 && h[k] <= (uint32_t)MAX_HAMMING) {
          const uint32_t entry = ((uint32_t)tid << 16) | (uint32_t)(r + k);
          const uint32_t at = atomicAdd(&sh.n_queued, 1u);
          if (at < (uint32_t)WALK_QUEUE) sh.queue[at] = entry;
          else finish_candidate<MODE>(P, g, both_directions, sh, entry, a_at, b_at);   // no room (images full of near-identical descriptors): on the spot
        }
      }
    }
    __syncthreads();
#if defined(SFM_ABLATE_WALK_UPTO) && SFM_ABLATE_WALK_UPTO == 4   // timing experiments only: candidates queued, none finished
    const uint32_t queued = lo_half.x == 0x12345678u ? min(sh.n_queued, (uint32_t)WALK_QUEUE) : 0u;
#else
    const uint32_t queued = min(sh.n_queued, (uint32_t)WALK_QUEUE);
#endif
    for (uint32_t k = tid; k < queued; k += WALK_THREADS) finish_candidate<MODE>(P, g, both_directions, sh, sh.queue[k], a_at, b_at);
    __syncthreads();
    if (tid == 0) sh.n_queued = 0;   // the next appends come after the next chunk's barrier
  }
  if (MODE == WALK_COUNT) {
    // no round trip here: the segment is this workgroup's own
    const uint32_t total = sh.n_found;
    const uint32_t mine = min(total, (uint32_t)LIST_SEGMENT);
    uint2* out = P.list + (size_t)seg * LIST_SEGMENT;
    for (uint32_t k = tid; k < mine; k += WALK_THREADS) out[k] = sh.found[k];
    if (tid == 0) {
      P.seg_counts[seg] = mine;
      if (total > (uint32_t)LIST_SEGMENT) *P.overflow = 1;
    }
  }
  __syncthreads();   // the shared arrays are free again
}

// COUNT: grid (ceil(largest na of the batch / 256), 6, pairs), one workgroup per 256 records.
// work (may be null): counter that receives the number of candidates tested, for the statistics.
__global__ __launch_bounds__(WALK_THREADS) void k_ch_walk_count(const CascadePair* __restrict__ pairs, int both_directions,
                                                                unsigned long long* __restrict__ work) {
  __shared__ WalkShared sh;
  walk_block<WALK_COUNT>(pairs[blockIdx.z], blockIdx.y, blockIdx.x, gridDim.x, both_directions, sh, work);
}

// NEAREST: grid (a few, 1, pairs); returns at once unless the pair's list overflowed, else the
// workgroups stride the pair's (group, 256 records) items.  (Few workgroups per pair on purpose:
// nearly every launch finds nothing to do, and what it costs then is its number of workgroups.)
__global__ __launch_bounds__(WALK_THREADS) void k_ch_walk_nearest(const CascadePair* __restrict__ pairs, int both_directions) {
  __shared__ WalkShared sh;
  const CascadePair& P = pairs[blockIdx.z];
  if (*P.overflow == 0) return;
  const int nbx = (P.na + WALK_THREADS - 1) / WALK_THREADS;
  for (int item = blockIdx.x; item < GROUPS * nbx; item += gridDim.x)
    walk_block<WALK_NEAREST>(P, item / nbx, item % nbx, nbx, both_directions, sh);
}

// ---------------------------------------------------------------------------------------------
// The close pairs of a pair's list, 64 at a time: the state of both descriptors from their
// histograms (every thread that names a descriptor stores the same value), then, for the close pairs
// that lie inside the bins taken for either of them, the exact distance.  grid: (blocks, 1, pairs);
// the workgroup lays the segments' counts end to end and its tiles of 64 entries stride the list.
// The two 512-byte rows of a close pair are read by 32 lanes with 16 bytes each (whole cache lines;
// one lane reading a row alone touches a line of its own with every load), the squared differences
// go through LDS, and one lane per close pair adds them up in index order: src/metric.hpp:15-22.
// Dynamic LDS: (segments + 1) words.
// ---------------------------------------------------------------------------------------------
constexpr int NEAREST_MAX_SEGMENTS = GROUPS * (MAX_IMAGE_ROWS / 256);
constexpr int NEAREST_TILE = 64;       // close pairs whose squares are in LDS at a time
#ifndef SFM_NEAREST_SUPER   // (overridden in timing experiments: tools/build_variant.sh)
#define SFM_NEAREST_SUPER 256
#endif
constexpr int NEAREST_SUPER = SFM_NEAREST_SUPER;     // close pairs whose bins decisions are taken together: four tiles
constexpr int NEAREST_SQ_STRIDE = 132;   // words per row of squares (128 + 4: conflict-free 16-byte reads)

// work (may be null): for the statistics; work[0] receives the number of list entries, one of the
// WORK_SLOTS counters from work[2 + WORK_SLOTS] on the number of exact distances computed.
__global__ __launch_bounds__(256, 4) void k_ch_nearest(const CascadePair* __restrict__ pairs, int both_directions,
                                                       int segments, unsigned long long* __restrict__ work) {
  extern __shared__ uint32_t start[];   // start[s] = entries in the segments before s
  __shared__ uint32_t wave_sum[4];
  __shared__ __attribute__((aligned(16))) float sq[NEAREST_TILE * NEAREST_SQ_STRIDE];
  __shared__ uint32_t ent_i[NEAREST_SUPER], ent_j[NEAREST_SUPER], ent_f[NEAREST_SUPER];   // ent_f: 1 = into a, 2 = into b
  const CascadePair& P = pairs[blockIdx.z];
  if (*P.overflow != 0) return;   // incomplete list: k_ch_take_all + k_ch_walk_nearest do this pair
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // exclusive scan of the counts: consecutive runs of `per` segments per thread
  const int per = (segments + 255) / 256;
  uint32_t mine = 0;
  for (int k = 0; k < per; k++) {
    const int sgm = tid * per + k;
    if (sgm < segments) mine += P.seg_counts[sgm];
  }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_sum[wv] = incl;
  __syncthreads();
  uint32_t before = incl - mine;
  for (int k = 0; k < wv; k++) before += wave_sum[k];
  for (int k = 0; k < per; k++) {
    const int sgm = tid * per + k;
    if (sgm < segments) {
      start[sgm] = before;
      before += P.seg_counts[sgm];
    }
  }
  if (tid == 255) start[segments] = before;   // the last thread ends up with the total
  __syncthreads();
  const uint32_t total = start[segments];
  if (work && blockIdx.x == 0 && tid == 0 && total) atomicAdd(work, (unsigned long long)total);
#if defined(SFM_ABLATE_NEAREST_UPTO) && SFM_ABLATE_NEAREST_UPTO == 1
  return;   // diagnostic build: the scan of the segment counts alone
#endif
  unsigned long long needed = 0;   // entries whose distance this wavefront's lanes asked for (the same in all of its lanes)
  for (uint32_t t0 = blockIdx.x * NEAREST_SUPER; t0 < total; t0 += gridDim.x * NEAREST_SUPER) {
    // one lane per entry, 256 entries at once: the bins decisions.  (Three dependent trips to memory --
    // list entry, histograms, rows -- are what a tile costs; taking the first two for four tiles
    // together and sending a tile's row loads off before the tile before it is summed leaves about one
    // trip per tile.)
    {
      uint32_t i = 0, j = 0, f = 0;
      const uint32_t k = t0 + (uint32_t)tid;
      if (tid < NEAREST_SUPER && k < total) {
        // the segment that holds entry k: the last one that starts at or before k
        int lo = 0, hi = segments - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (start[mid] <= k) lo = mid;
          else hi = mid - 1;
        }
        const uint2 entry = P.list[(size_t)lo * LIST_SEGMENT + (k - start[lo])];
        i = entry.x;
        j = entry.y & ((1u << LIST_H_SHIFT) - 1u);
        const uint32_t h = entry.y >> LIST_H_SHIFT;
        // A descriptor's result slot says that the descriptor was touched (the compaction puts such
        // slots back to zero) either through its state or through its running minimum: the state is
        // only stored where this entry will not leave a distance.
        const uint32_t state_a = take_state(P.hist_a + (size_t)i * HIST_WORDS);
        if (inside_taken_bins(state_a, h)) f |= 1u;
        else P.res_a[(size_t)i * RES_WORDS + RES_STATE] = state_a;
        if (both_directions) {
          const uint32_t state_b = take_state(P.hist_b + (size_t)j * HIST_WORDS);
          if (inside_taken_bins(state_b, h)) f |= 2u;
          else P.res_b[(size_t)j * RES_WORDS + RES_STATE] = state_b;
        }
        if (f == 0) i = j = 0;   // no distance needed: the loads below go to row 0 and are not used
      }
      if (tid < NEAREST_SUPER) {
        ent_i[tid] = i;
        ent_j[tid] = j;
        ent_f[tid] = f;
      }
      if (work) needed += (unsigned long long)__popcll(__ballot(f != 0));   // statistics only
    }
    __syncthreads();
#if defined(SFM_ABLATE_NEAREST_UPTO) && SFM_ABLATE_NEAREST_UPTO == 2
    continue;   // diagnostic build: + the bins decisions
#endif
    const uint32_t left = total - t0;
    const int tiles = (int)((left < (uint32_t)NEAREST_SUPER ? left : (uint32_t)NEAREST_SUPER) + NEAREST_TILE - 1) / NEAREST_TILE;
    // 32 lanes per row: eight entries per half-wavefront, all sixteen loads on their way before the
    // first is used (no branches: an entry that needs no distance reads row 0)
    const int half = lane >> 5, jj = lane & 31;
    float4 a[8], b[8];
    auto issue_loads = [&](int tile) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int e = tile * NEAREST_TILE + wv * 16 + half * 8 + it;
        a[it] = reinterpret_cast<const float4*>(P.da + (size_t)ent_i[e] * DIM)[jj];
        b[it] = reinterpret_cast<const float4*>(P.db + (size_t)ent_j[e] * DIM)[jj];
      }
    };
    issue_loads(0);
    for (int tile = 0; tile < tiles; tile++) {
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int e = wv * 16 + half * 8 + it;
          float4 d;
          d.x = a[it].x - b[it].x; d.y = a[it].y - b[it].y; d.z = a[it].z - b[it].z; d.w = a[it].w - b[it].w;
          d.x = d.x * d.x; d.y = d.y * d.y; d.z = d.z * d.z; d.w = d.w * d.w;
          *reinterpret_cast<float4*>(sq + e * NEAREST_SQ_STRIDE + 4 * jj) = d;
        }
      }
      if (tile + 1 < tiles) issue_loads(tile + 1);   // on their way while this tile is summed
      __syncthreads();
      const int e = tile * NEAREST_TILE + tid;
#if defined(SFM_ABLATE_NEAREST_UPTO) && SFM_ABLATE_NEAREST_UPTO == 3
      if (tid < NEAREST_TILE && ent_f[e] == 77u) {   // diagnostic build: + the rows and their squares, no sums
#else
      if (tid < NEAREST_TILE && ent_f[e] != 0) {
#endif
#pragma clang fp contract(off)
        const float4* s4 = reinterpret_cast<const float4*>(sq + tid * NEAREST_SQ_STRIDE);
        float sum = 0.0f;
#pragma unroll 8
        for (int k = 0; k < DIM / 4; k++) {
          const float4 v = s4[k];
          sum = sum + v.x;
          sum = sum + v.y;
          sum = sum + v.z;
          sum = sum + v.w;
        }
        // (a - b)^2 and (b - a)^2 are the same float: one distance serves both directions
        const uint32_t i = ent_i[e], j = ent_j[e], f = ent_f[e];
        const unsigned long long hi = (unsigned long long)__float_as_uint(sqrtf(sum)) << 32;
        if (f & 1u) atomicMax(reinterpret_cast<unsigned long long*>(P.res_a + (size_t)i * RES_WORDS + RES_BEST), ~(hi | j));
        if (f & 2u) atomicMax(reinterpret_cast<unsigned long long*>(P.res_b + (size_t)j * RES_WORDS + RES_BEST), ~(hi | i));
      }
      __syncthreads();   // the squares are free again (and, after the last tile, the entries)
    }
  }
  // one of WORK_SLOTS counters (a single one would serialise the launch's atomics); they lie behind the
  // walk's WORK_SLOTS counters, which start two words after work[0]
  if (work && lane == 0 && needed)
    atomicAdd(work + 2 + WORK_SLOTS + ((blockIdx.z * gridDim.x + blockIdx.x) & (WORK_SLOTS - 1)), needed);
}

// The bins decision for every query of a pair whose list overflowed.  One thread per query;
// grid: (blocks, directions, pairs), threads stride the queries.
__global__ __launch_bounds__(256) void k_ch_take_all(const CascadePair* __restrict__ pairs,
                                                     unsigned long long* __restrict__ overflowed_pairs) {
  const CascadePair& P = pairs[blockIdx.z];
  if (*P.overflow == 0) return;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(overflowed_pairs, 1ull);
  const int n = blockIdx.y == 0 ? P.na : P.nb;
  uint32_t* hist = blockIdx.y == 0 ? P.hist_a : P.hist_b;
  uint32_t* res = blockIdx.y == 0 ? P.res_a : P.res_b;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
    uint32_t* slot = hist + (size_t)q * HIST_WORDS;
    const uint4 h0 = reinterpret_cast<const uint4*>(slot)[0];
    const uint4 h1 = reinterpret_cast<const uint4*>(slot)[1];
    const uint4 h2 = reinterpret_cast<const uint4*>(slot)[2];
    const uint32_t any = h0.x | h0.y | h0.z | h0.w | h1.x | h1.y | h1.z | h1.w | h2.x | h2.y | h2.z;
    if (any) res[(size_t)q * RES_WORDS + RES_STATE] = take_state(slot);
  }
}

// Cross-check (cascade_hash_matcher.cc:182-193) and ordered compaction.  One workgroup of 1024
// threads per pair, sweeps of 1024 consecutive queries, one per thread: output order is ascending
// lhs index.  out_defaults: the records `Matches temp(matches.size())` holds before the survivors
// are appended (.cc:187) -- one per forward match, cross-check only.  Every slot that was touched
// goes back to zero, and so does the list's overflow mark.
__device__ __forceinline__ void clear_query(uint32_t* hist_row, uint32_t* res) {
  const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int w = 0; w < HIST_WORDS / 4; w++) reinterpret_cast<uint4*>(hist_row)[w] = z;
  *reinterpret_cast<uint4*>(res) = z;
}

constexpr int CH_COMPACT_THREADS = 1024;
__global__ __launch_bounds__(CH_COMPACT_THREADS) void k_ch_compact(const CascadePair* __restrict__ pairs, int cross_check,
                                                     int32_t* __restrict__ out_defaults,
                                                     int32_t* __restrict__ out_counts, int32_t* __restrict__ out_lhs,
                                                     int32_t* __restrict__ out_rhs, float* __restrict__ out_dist,
                                                     int64_t out_stride) {
  __shared__ int wkeep[2][CH_COMPACT_THREADS / 64], wfwd[2][CH_COMPACT_THREADS / 64];   // per wavefront and sweep parity: one barrier per sweep
  const CascadePair& P = pairs[blockIdx.x];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const unsigned long long below = ((unsigned long long)1 << l) - 1;
  const size_t ob = (size_t)P.out_slot * (size_t)out_stride;
  int base_keep = 0, base_fwd = 0;   // the same in every thread
  // One query per thread and sweep, 1,024 consecutive queries per sweep: a wavefront reads 1 KiB of
  // consecutive result words per instruction (eight consecutive queries per thread would have every
  // lane on a cache line of its own, eight times over).
  for (int i0 = 0, sweep = 0; i0 < P.na; i0 += CH_COMPACT_THREADS, sweep ^= 1) {
    const int i = i0 + tid;
    int j = 0;
    float d = 0.0f;
    bool keep = false, fwd = false;
    if (i < P.na) {
      uint32_t* ra = P.res_a + (size_t)i * RES_WORDS;
      // state, -, best (inverted) low, high.  Touched: a state or a minimum; matched: a minimum (distances
      // are only ever folded into a descriptor whose state says "match")
      const uint4 t = *reinterpret_cast<const uint4*>(ra);
      if ((t.x | t.z | t.w) != 0) clear_query(P.hist_a + (size_t)i * HIST_WORDS, ra);   // only this thread ever reads query i
      if ((t.z | t.w) != 0) {
        fwd = true;
        j = (int)~t.z;
        d = __uint_as_float(~t.w);
        keep = true;
        if (cross_check) {
          const uint4 u = *reinterpret_cast<const uint4*>(P.res_b + (size_t)j * RES_WORDS);
          keep = (u.z | u.w) != 0 && (int)~u.z == i;
        }
      }
    }
    const unsigned long long mk = __ballot(keep), mf = __ballot(fwd);
    if (l == 0) {
      wkeep[sweep][wv] = __popcll(mk);
      wfwd[sweep][wv] = __popcll(mf);
    }
    __syncthreads();
    int before = base_keep;
#pragma unroll
    for (int k = 0; k < CH_COMPACT_THREADS / 64; k++) {
      const int c = wkeep[sweep][k];
      before += k < wv ? c : 0;
      base_keep += c;
      base_fwd += wfwd[sweep][k];
    }
    if (keep) {
      const size_t o = ob + (size_t)(before + __popcll(mk & below));
      out_lhs[o] = i;
      out_rhs[o] = j;
      out_dist[o] = d;
    }
  }
  __syncthreads();
  // the second image's slots were read by any thread above: cleared only now
  if (cross_check) {
    for (int q = tid; q < P.nb; q += CH_COMPACT_THREADS) {
      uint32_t* rb = P.res_b + (size_t)q * RES_WORDS;
      const uint4 u = *reinterpret_cast<const uint4*>(rb);
      if ((u.x | u.z | u.w) != 0) clear_query(P.hist_b + (size_t)q * HIST_WORDS, rb);
    }
  }
  if (tid == 0) {
    out_counts[P.out_slot] = base_keep;
    out_defaults[P.out_slot] = cross_check ? base_fwd : 0;
    *P.overflow = 0;
  }
}

// Number of non-zero words in a buffer that should be all zeros (the work space between batches).
__global__ __launch_bounds__(256) void k_ch_count_nonzero(const uint32_t* __restrict__ words, size_t n,
                                                          unsigned long long* __restrict__ out) {
  unsigned long long mine = 0;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) mine += words[k] != 0;
  if (mine) atomicAdd(out, mine);
}

}  // namespace sfm_cascade

#endif  // SFM_AMD_CSRC_CASCADE_KERNELS_HPP_
