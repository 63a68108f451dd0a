// gfx950 kernels of the cascade-hash matcher (the GPU side of the reference's
// src/feature_matcher/cascade_hash_matcher.cc).
//
// Per image, once (ConstructHashDescriptors, .cc:11-49):
//   k_ch_colsum   the image's share of the mean vector (.cc:160-162)
//   k_ch_hash     188 projections of every centred descriptor -> 128-bit code + six 10-bit bucket ids
//   k_ch_group    per group a copy of the hash records sorted by bucket (stable: ascending index
//                 inside a bucket, the order .cc:43-46 fills them in) + bucket offsets
// Per pair (DescriptorMath both ways + cross-check, .cc:51-116, :173-195):
//   k_ch_pairs<COUNT>    walk every (first-image descriptor, second-image descriptor) that share a
//                        bucket; for those within Hamming distance 10 count them into both
//                        descriptors' Hamming histograms
//   k_ch_threshold       per query: how many Hamming bins the reference would take (.cc:95-103) and
//                        whether that gives it the two candidates it needs (.cc:105)
//   k_ch_pairs<NEAREST>  the same walk; exact L2 distance for the candidates inside a query's bins,
//                        minimum kept per query as (distance, index)
//   k_ch_compact         cross-check, ordered compaction, counts
//
// Why the walk is by bucket and not by query as in the reference: "i and j share a bucket of group g"
// and the Hamming distance are symmetric, so one walk serves both directions, and with the records
// of a group stored in bucket order both sides of it are read as contiguous runs -- the pass is a
// stream over 6 x (na + nb) records of 32 bytes, bound by memory, instead of a gather per query.
// What the reference derives from visiting order (first-seen de-duplication, the order inside a
// Hamming bin) does not reach its result: bins are taken whole and the winner is the minimum of
// (distance, index) pairs.
#ifndef SFM_AMD_CSRC_CASCADE_KERNELS_HPP_
#define SFM_AMD_CSRC_CASCADE_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sfm_cascade {

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

// Per-query work space of one pair and direction: 16 words.
constexpr int SLOT_WORDS = 16;
constexpr int SLOT_HIST = 0;       // [0 .. 10] candidates at Hamming distance h
constexpr int SLOT_TAKE = 11;      // largest Hamming distance the reference takes for this query, -1: no match
constexpr int SLOT_BEST = 12;      // [12, 13] 64-bit minimum of (distance bits << 32 | index)

struct ImageRef {
  const float* d;
  int32_t n;
  int32_t pad_;
};

struct HashJob {
  const float* d;        // [n][128]
  HashRec* recs;         // [n]
  HashRec* grouped;      // [6][n]
  int32_t* offsets;      // [6][1025]
  int32_t n;
  int32_t first_block;   // k_ch_hash: the job's first workgroup in the launch
};

struct CascadePair {
  const HashRec* ga;     // first image: records grouped by bucket, [6][na]
  const HashRec* gb;
  const int32_t* offa;   // [6][1025]
  const int32_t* offb;
  const float* da;       // descriptor rows
  const float* db;
  uint32_t* slots_a;     // [na][16]
  uint32_t* slots_b;     // [nb][16]
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
  for (; r + 8 <= im.n; r += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = p[r + k];
#pragma unroll
    for (int k = 0; k < 8; k++) s = s + v[k];
  }
  for (; r < im.n; r++) s = s + p[r];
  out[(size_t)blockIdx.x * DIM + c] = s;
}

// ---------------------------------------------------------------------------------------------
// Hashing (cascade_hash_matcher.cc:18-38).  One lane per descriptor: the centred descriptor stays
// in registers, the projection rows arrive through scalar loads (they are the same for every
// lane), each projection is the sum over j = 0..127 of row[j] * d[j] in that order, product and
// sum rounded separately.
// proj holds the 188 rows (128 primary, then 6 x 10 secondary) in blocks of four, a block stored
// as [j][4] so that one scalar load brings four rows' entries for four consecutive j, followed by
// the 128-float mean: proj[(b * 128 + j) * 4 + k] = row 4b + k, entry j.
// ---------------------------------------------------------------------------------------------
constexpr int PROJ_BLOCK = 4;
constexpr int PROJ_BLOCKS = NPROJ / PROJ_BLOCK;   // 47
static_assert(NPROJ % PROJ_BLOCK == 0, "projection rows come in blocks of four");

__device__ __forceinline__ uint32_t project_block(const float* __restrict__ blk, const float (&d)[DIM]) {
#pragma clang fp contract(off)
  float acc[PROJ_BLOCK] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j0 = 0; j0 < DIM; j0 += 4) {
    float4 p[4];
#pragma unroll
    for (int j = 0; j < 4; j++) p[j] = *reinterpret_cast<const float4*>(blk + (j0 + j) * PROJ_BLOCK);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[0] = acc[0] + p[j].x * d[j0 + j];
      acc[1] = acc[1] + p[j].y * d[j0 + j];
      acc[2] = acc[2] + p[j].z * d[j0 + j];
      acc[3] = acc[3] + p[j].w * d[j0 + j];
    }
    // keeps the compiler from issuing all scalar loads of a block up front, which would need
    // more scalar registers than there are
    __builtin_amdgcn_sched_barrier(0);
  }
  return (acc[0] > 0.0f ? 1u : 0u) | (acc[1] > 0.0f ? 2u : 0u) | (acc[2] > 0.0f ? 4u : 0u) | (acc[3] > 0.0f ? 8u : 0u);
}

__global__ __launch_bounds__(64) void k_ch_hash(const HashJob* __restrict__ jobs, int n_jobs,
                                                const float* __restrict__ proj) {
#pragma clang fp contract(off)
  // which job this workgroup belongs to: the last one whose first_block is not beyond it
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= jobs[mid].first_block) lo = mid;
    else hi = mid - 1;
  }
  const HashJob job = jobs[lo];
  const int row = ((int)blockIdx.x - job.first_block) * 64 + (int)threadIdx.x;
  const bool live = row < job.n;
  const float* mean = proj + NPROJ * DIM;
  float d[DIM];
  {
    const float4* src = reinterpret_cast<const float4*>(job.d + (size_t)(live ? row : 0) * DIM);
#pragma unroll
    for (int j = 0; j < DIM / 4; j++) {
      const float4 v = src[j];
      d[4 * j + 0] = v.x - mean[4 * j + 0];     // cascade_hash_matcher.cc:21
      d[4 * j + 1] = v.y - mean[4 * j + 1];
      d[4 * j + 2] = v.z - mean[4 * j + 2];
      d[4 * j + 3] = v.w - mean[4 * j + 3];
    }
  }
  // sign bits of the 188 projections, row r at bit r % 32 of word r / 32
  uint32_t word[6];
#pragma unroll
  for (int w = 0; w < 6; w++) {
    uint32_t bits = 0;
    const int nb = w < 5 ? 8 : PROJ_BLOCKS - 40;
#pragma unroll 1
    for (int kb = 0; kb < nb; kb++)
      bits |= project_block(proj + (size_t)(w * 8 + kb) * DIM * PROJ_BLOCK, d) << (kb * 4);
    word[w] = bits;
  }
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
  if (live) job.recs[row] = rec;
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
  HashRec* out = job.grouped + (size_t)g * (size_t)n;
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
      out[pos] = rec;
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
// The walk over bucket-sharing descriptor pairs.  Thread = one record of the first image in group
// g's bucket order (so neighbouring lanes mostly sit in the same bucket and read the same run of
// the second image's records); it loops over the second image's records of that bucket.  A pair
// that shares buckets in several groups is handled in the lowest such group only: that is the
// reference's "seen" map (cascade_hash_matcher.cc:84-86) without its visiting order.
//   COUNT    Hamming histograms of both descriptors (.cc:88-91, bins above 10 are never read)
//   NEAREST  exact distance for candidates inside the bins a query takes (.cc:95-103), running
//            minimum of (distance, index) per query (.cc:106-111: partial_sort of pairs)
// grid: (ceil(max na / 256), 6, pairs)
// ---------------------------------------------------------------------------------------------
constexpr int PASS_COUNT = 0;
constexpr int PASS_NEAREST = 1;

template <int PASS>
__global__ __launch_bounds__(256) void k_ch_pairs(const CascadePair* __restrict__ pairs, int both_directions) {
  const CascadePair& P = pairs[blockIdx.z];
  const int g = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int na = P.na, nb = P.nb;
  if (p >= na) return;
  const uint4* ga = reinterpret_cast<const uint4*>(P.ga + (size_t)g * (size_t)na + p);
  const uint4 ca = ga[0], ma = ga[1];          // code | bids[0..2], idx
  const uint32_t bida[3] = {ma.x, ma.y, ma.z};
  const uint32_t ba = (bida[g >> 1] >> ((g & 1) * 16)) & 0xFFFFu;
  const int32_t* off = P.offb + (size_t)g * (BUCKETS + 1) + ba;
  const int s = off[0], e = off[1];
  // which of the groups below g this record can collide in: compared as 16-bit halves
  const uint4* gb = reinterpret_cast<const uint4*>(P.gb + (size_t)g * (size_t)nb);
  for (int q = s; q < e; q++) {
    const uint4 cb = gb[2 * q], mb = gb[2 * q + 1];
    const uint32_t h = __popc(ca.x ^ cb.x) + __popc(ca.y ^ cb.y) + __popc(ca.z ^ cb.z) + __popc(ca.w ^ cb.w);
    if (h > (uint32_t)MAX_HAMMING) continue;
    const uint32_t x0 = ma.x ^ mb.x, x1 = ma.y ^ mb.y, x2 = ma.z ^ mb.z;
    bool earlier = false;   // shares a bucket in a group before g
    if (g > 0) earlier |= (x0 & 0xFFFFu) == 0;
    if (g > 1) earlier |= (x0 >> 16) == 0;
    if (g > 2) earlier |= (x1 & 0xFFFFu) == 0;
    if (g > 3) earlier |= (x1 >> 16) == 0;
    if (g > 4) earlier |= (x2 & 0xFFFFu) == 0;
    if (earlier) continue;
    const uint32_t i = ma.w, j = mb.w;
    uint32_t* sa = P.slots_a + (size_t)i * SLOT_WORDS;
    uint32_t* sb = P.slots_b + (size_t)j * SLOT_WORDS;
    if (PASS == PASS_COUNT) {
      atomicAdd(sa + SLOT_HIST + h, 1u);
      if (both_directions) atomicAdd(sb + SLOT_HIST + h, 1u);
    } else {
      const bool fa = (int)h <= (int)sa[SLOT_TAKE];
      const bool fb = both_directions && (int)h <= (int)sb[SLOT_TAKE];
      if (fa || fb) {
        // (a - b)^2 and (b - a)^2 are the same float: one distance serves both directions
        const float dist = exact_l2(P.da + (size_t)i * DIM, P.db + (size_t)j * DIM);
        const unsigned long long hi = (unsigned long long)__float_as_uint(dist) << 32;
        if (fa) atomicMin(reinterpret_cast<unsigned long long*>(sa + SLOT_BEST), hi | j);
        if (fb) atomicMin(reinterpret_cast<unsigned long long*>(sb + SLOT_BEST), hi | i);
      }
    }
  }
}

// cascade_hash_matcher.cc:95-105 per query: bins are taken whole, nearest first, no new bin once ten
// candidates are in, none beyond distance 10; a match needs two candidates.  Also resets the
// running minimum.  One thread per query; grid: (ceil(max(na, nb) / 256), directions, pairs).
__global__ __launch_bounds__(256) void k_ch_threshold(const CascadePair* __restrict__ pairs) {
  const CascadePair& P = pairs[blockIdx.z];
  const int dir = blockIdx.y;
  const int n = dir == 0 ? P.na : P.nb;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  uint32_t* slot = (dir == 0 ? P.slots_a : P.slots_b) + (size_t)q * SLOT_WORDS;
  const uint4 h0 = reinterpret_cast<const uint4*>(slot)[0];
  const uint4 h1 = reinterpret_cast<const uint4*>(slot)[1];
  const uint4 h2 = reinterpret_cast<const uint4*>(slot)[2];
  const uint32_t hist[MAX_HAMMING + 1] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z};
  uint32_t taken = 0;
  int last = -1;
#pragma unroll
  for (int h = 0; h <= MAX_HAMMING; h++) {
    if (taken < (uint32_t)SHORT_LIST) {
      taken += hist[h];
      last = h;
    }
  }
  uint4 tail;
  tail.x = 0xFFFFFFFFu;                       // SLOT_BEST, low word
  tail.y = 0xFFFFFFFFu;                       // SLOT_BEST, high word
  tail.z = 0;
  tail.w = 0;
  slot[SLOT_TAKE] = taken >= 2 ? (uint32_t)last : 0xFFFFFFFFu;
  reinterpret_cast<uint4*>(slot)[3] = tail;
}

// Cross-check (cascade_hash_matcher.cc:182-193) and ordered compaction.  One workgroup of 1024
// threads per pair, eight consecutive queries per thread and sweep: output order is ascending
// lhs index.  out_defaults: the records `Matches temp(matches.size())` holds before the survivors
// are appended (.cc:187) -- one per forward match, cross-check only.
__global__ __launch_bounds__(1024) void k_ch_compact(const CascadePair* __restrict__ pairs, int cross_check,
                                                     int32_t* __restrict__ out_defaults,
                                                     int32_t* __restrict__ out_counts, int32_t* __restrict__ out_lhs,
                                                     int32_t* __restrict__ out_rhs, float* __restrict__ out_dist,
                                                     int64_t out_stride) {
  __shared__ int wkeep[16], wfwd[16];
  __shared__ int base_keep, base_fwd;
  const CascadePair& P = pairs[blockIdx.x];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const size_t ob = (size_t)P.out_slot * (size_t)out_stride;
  if (tid == 0) {
    base_keep = 0;
    base_fwd = 0;
  }
  __syncthreads();
  for (int i0 = 0; i0 < P.na; i0 += 8192) {
    const int ib = i0 + tid * 8;
    int j[8];
    float d[8];
    bool keep[8];
    int nfwd = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = ib + k;
      keep[k] = false;
      j[k] = 0;
      d[k] = 0.0f;
      if (i < P.na) {
        const uint4 t = reinterpret_cast<const uint4*>(P.slots_a + (size_t)i * SLOT_WORDS)[3];   // best lo, best hi, -, -
        const int take = (int)P.slots_a[(size_t)i * SLOT_WORDS + SLOT_TAKE];
        if (take >= 0) {
          nfwd++;
          j[k] = (int)t.x;
          d[k] = __uint_as_float(t.y);
          keep[k] = true;
          if (cross_check) {
            const uint32_t* sb = P.slots_b + (size_t)j[k] * SLOT_WORDS;
            keep[k] = (int)sb[SLOT_TAKE] >= 0 && (int)sb[SLOT_BEST] == i;
          }
        }
      }
      cnt += keep[k] ? 1 : 0;
    }
    int incl = cnt, incl_f = nfwd;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      const int vf = __shfl_up(incl_f, o);
      if (l >= o) {
        incl += v;
        incl_f += vf;
      }
    }
    if (l == 63) {
      wkeep[wv] = incl;
      wfwd[wv] = incl_f;
    }
    __syncthreads();
    int before = base_keep;
    for (int k = 0; k < wv; k++) before += wkeep[k];
    int pos = before + incl - cnt;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (keep[k]) {
        const size_t o = ob + (size_t)pos++;
        out_lhs[o] = ib + k;
        out_rhs[o] = j[k];
        out_dist[o] = d[k];
      }
    }
    __syncthreads();
    if (tid == 0) {
      int s = base_keep, f = base_fwd;
      for (int k = 0; k < 16; k++) {
        s += wkeep[k];
        f += wfwd[k];
      }
      base_keep = s;
      base_fwd = f;
    }
    __syncthreads();
  }
  if (tid == 0) {
    out_counts[P.out_slot] = base_keep;
    out_defaults[P.out_slot] = cross_check ? base_fwd : 0;
  }
}

}  // namespace sfm_cascade

#endif  // SFM_AMD_CSRC_CASCADE_KERNELS_HPP_
