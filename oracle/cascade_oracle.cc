// CPU restatement of the reference's cascade-hash matcher
// (src/feature_matcher/cascade_hash_matcher.cc, .hpp).  TEST INFRASTRUCTURE: used by tests/,
// by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the sfm_amd package.
//
// PARITY UNPINNED.  The reference file cannot be built here (it needs Eigen, which this machine
// does not have) and the reference holds no test, fixture or recorded output for this matcher, so
// nothing below has been checked against the reference running.  What is and is not fixed by the
// reference's source text:
//   * fixed, and followed literally: the projection matrices (std::mt19937 with its default seed and
//     std::normal_distribution<>, both taken from the same standard library the reference would be
//     built with, :127-146), which images contribute to the mean and in which order (:148-165), the
//     layout quirk of that mean (:160-162), bucket construction (:40-47), candidate gathering,
//     Hamming binning, the choice of candidates for the exact distance, the winner (:66-114), the
//     cross-check and the default records it leaves in front of the result (:177-195);
//   * NOT fixed: the order in which Eigen adds floats in `m.colwise().sum()` (:162) and in the
//     matrix-vector products `primary_hash_project * d` (:23, :30).  That order depends on Eigen's
//     version, on the instruction set it was built for and on the alignment of the operands at run
//     time.  Here every such sum is taken in index order with one fp32 accumulator, each product and
//     each addition rounded on its own -- the convention of the reference's own src/metric.hpp:15-21.
//     A hash bit of a descriptor whose projection lies within rounding distance of zero may therefore
//     differ from a given build of the reference.
//
// Built with -ffp-contract=off (no fused multiply-add), like the rest of the oracle.
#include <algorithm>
#include <bitset>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace {

constexpr int kDim = 128;
constexpr int kGroups = 6;       // bucket_group_size, cascade_hash_matcher.cc:122
constexpr int kGroupBits = 10;   // bits_of_group, :123
constexpr int kCodeBits = 128;   // hash_code_size, :124
constexpr int kBuckets = 1 << kGroupBits;   // group_size, :125

using IndexT = int64_t;

// src/metric.hpp:15-22
float l2_metric(const float* lhs, const float* rhs) {
  float sum = 0.0f;
  for (int i = 0; i < kDim; i++) {
    const float t = lhs[i] - rhs[i];
    sum = sum + t * t;
  }
  return std::sqrt(sum);
}

struct HashDescriptor {      // cascade_hash_matcher.hpp:10-15
  std::bitset<128> hash_code;
  std::vector<uint16_t> bucket_ids;
};

struct HashDescriptors {     // cascade_hash_matcher.hpp:17-22
  std::vector<HashDescriptor> descriptor;
  std::vector<std::vector<std::vector<IndexT>>> bucket_ids;
};

struct Projection {
  std::vector<float> primary;     // [128][128], element (i, j) multiplies dimension j for bit i
  std::vector<float> secondary;   // [6][10][128]
};

// cascade_hash_matcher.cc:127-146: one generator, one distribution object, drawn in this order.
// The distribution works in double; storing into a float matrix rounds each draw once.
Projection make_projection() {
  Projection p;
  std::mt19937 gen(std::mt19937::default_seed);
  std::normal_distribution<> d(0, 1);
  p.primary.resize((size_t)kCodeBits * kDim);
  for (int i = 0; i < kCodeBits; i++)
    for (int j = 0; j < kDim; j++) p.primary[(size_t)i * kDim + j] = (float)d(gen);
  p.secondary.resize((size_t)kGroups * kGroupBits * kDim);
  for (int t = 0; t < kGroups; t++)
    for (int i = 0; i < kGroupBits; i++)
      for (int j = 0; j < kDim; j++) p.secondary[((size_t)t * kGroupBits + i) * kDim + j] = (float)d(gen);
  return p;
}

// cascade_hash_matcher.cc:160-162.  The reference maps the row-major descriptor block as a
// COLUMN-major n x 128 matrix, so "column c" of that view is the c-th run of n consecutive floats
// of the block, not dimension c of the descriptors.  Entry c of the result is the sum of that run.
void image_column_sums(const float* raw, int64_t n, float* out) {
  for (int c = 0; c < kDim; c++) {
    float s = 0.0f;
    for (int64_t r = 0; r < n; r++) s = s + raw[(size_t)c * (size_t)n + (size_t)r];
    out[c] = s;
  }
}

// cascade_hash_matcher.cc:11-49
HashDescriptors construct_hash_descriptors(const float* desc, int64_t n, const float* mean, const Projection& pj) {
  HashDescriptors h;
  h.descriptor.resize((size_t)n);
  float d[kDim];
  for (int64_t i = 0; i < n; i++) {
    for (int j = 0; j < kDim; j++) d[j] = desc[(size_t)i * kDim + j] - mean[j];                 // :21
    for (int j = 0; j < kCodeBits; j++) {                                                        // :23-26
      float s = 0.0f;
      const float* row = &pj.primary[(size_t)j * kDim];
      for (int k = 0; k < kDim; k++) s = s + row[k] * d[k];
      h.descriptor[(size_t)i].hash_code[(size_t)j] = s > 0;
    }
    h.descriptor[(size_t)i].bucket_ids.resize(kGroups);
    for (int g = 0; g < kGroups; g++) {                                                          // :29-38
      uint16_t bucket_id = 0;
      for (int k = 0; k < kGroupBits; k++) {
        float s = 0.0f;
        const float* row = &pj.secondary[((size_t)g * kGroupBits + k) * kDim];
        for (int q = 0; q < kDim; q++) s = s + row[q] * d[q];
        bucket_id = (uint16_t)((bucket_id << 1) + (s > 0));
      }
      h.descriptor[(size_t)i].bucket_ids[(size_t)g] = bucket_id;
    }
  }
  h.bucket_ids.resize(kGroups);                                                                  // :40-47
  for (int g = 0; g < kGroups; g++) {
    h.bucket_ids[(size_t)g].resize(kBuckets);
    for (size_t i = 0; i < h.descriptor.size(); i++)
      h.bucket_ids[(size_t)g][h.descriptor[i].bucket_ids[(size_t)g]].push_back((IndexT)i);
  }
  return h;
}

// Tests may replace the reference's random projections by matrices of their own, which lets a
// known-answer case choose hash bits and distances independently (ch_set_projection).
Projection g_override;
bool g_have_override = false;

Projection current_projection() { return g_have_override ? g_override : make_projection(); }

struct RawMatch {
  IndexT lhs_idx = 0, rhs_idx = 0;
  float dist = 0.0f;     // not part of the reference's record; kept so that tests can compare it
};

// One direction, cascade_hash_matcher.cc:51-116, kept container for container (a flat candidate
// list with repeats, a "seen" map, Hamming bins filled in first-seen order, a short list of exact
// distances) so that nothing about visiting order is assumed away here.
std::vector<RawMatch> descriptor_match(const HashDescriptors& q_hash, const float* q_rows, const HashDescriptors& t_hash,
                                       const float* t_rows) {
  const size_t short_list_goal = 10;                       // kNumTopCandidates, :57
  std::vector<RawMatch> found;
  std::vector<int> pool;                                   // :58
  std::unordered_map<int, std::vector<IndexT>> bins;       // :61, keyed by Hamming distance
  std::unordered_map<IndexT, bool> seen;                   // :62
  std::vector<std::pair<float, IndexT>> short_list;        // :64

  for (size_t q = 0; q < q_hash.descriptor.size(); q++) {
    pool.clear();
    bins.clear();
    seen.clear();
    short_list.clear();
    const HashDescriptor& me = q_hash.descriptor[q];
    for (int g = 0; g < kGroups; g++)                      // :72-78: the six buckets this query falls into
      for (IndexT t : t_hash.bucket_ids[(size_t)g][me.bucket_ids[(size_t)g]]) {
        pool.push_back((int)t);
        seen[t] = false;
      }
    if (pool.size() < 2) continue;                         // :80-82, repeats included in the count
    for (IndexT t : pool) {                                // :84-93
      if (seen[t]) continue;
      seen[t] = true;
      const int h = (int)(me.hash_code ^ t_hash.descriptor.at((size_t)t).hash_code).count();
      bins[h].push_back(t);
    }
    // :95-103: whole bins, nearest Hamming distance first, none beyond distance bits_of_group,
    // no new bin once the short list has reached its goal
    for (int h = 0; h <= kGroupBits && short_list.size() < short_list_goal; h++)
      for (IndexT t : bins[h]) short_list.emplace_back(l2_metric(q_rows + q * kDim, t_rows + (size_t)t * kDim), t);
    if (short_list.size() < 2) continue;                   // :105
    std::partial_sort(short_list.begin(), short_list.begin() + 2, short_list.end());   // :106-108
    RawMatch m;
    m.lhs_idx = (IndexT)q;                                 // :110-112: no ratio test
    m.rhs_idx = short_list[0].second;
    m.dist = short_list[0].first;
    found.push_back(m);
  }
  return found;
}

// cascade_hash_matcher.cc:174-195.  Returns the list as the reference stores it, except that the
// value-initialised records `Matches temp(matches.size())` puts in front (:187) are returned as a
// count: each of them is {lhs_idx 0, rhs_idx 0}.
std::vector<RawMatch> match_pair(const HashDescriptors& lh, const float* ld, const HashDescriptors& rh, const float* rd,
                                 bool cross, int64_t* n_default) {
  std::vector<RawMatch> matches = descriptor_match(lh, ld, rh, rd);
  *n_default = 0;
  if (cross) {
    std::vector<RawMatch> reverse = descriptor_match(rh, rd, lh, ld);
    std::unordered_map<IndexT, IndexT> reverse_map;
    for (const RawMatch& m : reverse) reverse_map[m.lhs_idx] = m.rhs_idx;
    std::vector<RawMatch> temp;
    *n_default = (int64_t)matches.size();
    for (const RawMatch& m : matches)
      if (reverse_map.find(m.rhs_idx) != reverse_map.end() && reverse_map.at(m.rhs_idx) == m.lhs_idx) temp.push_back(m);
    matches.swap(temp);
  }
  return matches;
}

}  // namespace

extern "C" {

// primary [128][128], secondary [6][10][128]; both NULL: back to the reference's matrices
void ch_set_projection(const float* primary, const float* secondary) {
  g_have_override = primary && secondary;
  if (!g_have_override) return;
  g_override.primary.assign(primary, primary + (size_t)kCodeBits * kDim);
  g_override.secondary.assign(secondary, secondary + (size_t)kGroups * kGroupBits * kDim);
}

void ch_projection(float* primary, float* secondary) {
  const Projection p = make_projection();
  std::memcpy(primary, p.primary.data(), p.primary.size() * sizeof(float));
  std::memcpy(secondary, p.secondary.data(), p.secondary.size() * sizeof(float));
}

// cascade_hash_matcher.cc:148-152,157: the images of the pair list in the order in which the
// reference's std::unordered_set hands them out.  pair_ids: 2*n_pairs values in std::set order.
int64_t ch_image_order(const int64_t* pair_ids, int64_t n_pairs, int64_t* out_ids) {
  std::unordered_set<IndexT> index_set;
  for (int64_t p = 0; p < n_pairs; p++) {
    index_set.insert(pair_ids[2 * p]);
    index_set.insert(pair_ids[2 * p + 1]);
  }
  int64_t n = 0;
  for (IndexT id : index_set) out_ids[n++] = id;
  return n;
}

// The two measures the matching uses, as this oracle computes them (tests compare them with the
// reference's own src/metric.hpp built from the reference checkout).
float ch_l2_metric(const float* lhs, const float* rhs) { return l2_metric(lhs, rhs); }
int ch_hamming(const uint32_t* lhs, const uint32_t* rhs) {
  std::bitset<128> a, b;
  for (int j = 0; j < 128; j++) {
    a[(size_t)j] = (lhs[j / 32] >> (j % 32)) & 1u;
    b[(size_t)j] = (rhs[j / 32] >> (j % 32)) & 1u;
  }
  return (int)(a ^ b).count();
}

void ch_image_column_sums(const float* raw, int64_t n, float* out128) { image_column_sums(raw, n, out128); }

// cascade_hash_matcher.cc:154-165: images in the order given (ch_image_order)
void ch_mean(const float* const* descs, const int64_t* sizes, int64_t n_images, float* mean128) {
  float acc[kDim];
  for (int c = 0; c < kDim; c++) acc[c] = 0.0f;
  int total_size = 0;
  float col[kDim];
  for (int64_t k = 0; k < n_images; k++) {
    image_column_sums(descs[k], sizes[k], col);
    for (int c = 0; c < kDim; c++) acc[c] = acc[c] + col[c];
    total_size += (int)sizes[k];
  }
  for (int c = 0; c < kDim; c++) mean128[c] = acc[c] / (float)total_size;
}

// codes: n x 4 words, bit j of the hash code in word j / 32 at position j % 32; bucket_ids: n x 6
void ch_hash(const float* desc, int64_t n, const float* mean128, uint32_t* codes, uint16_t* bucket_ids) {
  const Projection pj = current_projection();
  const HashDescriptors h = construct_hash_descriptors(desc, n, mean128, pj);
  for (int64_t i = 0; i < n; i++) {
    for (int w = 0; w < 4; w++) {
      uint32_t v = 0;
      for (int b = 0; b < 32; b++) v |= (uint32_t)h.descriptor[(size_t)i].hash_code[(size_t)(w * 32 + b)] << b;
      codes[i * 4 + w] = v;
    }
    for (int g = 0; g < kGroups; g++) bucket_ids[i * kGroups + g] = h.descriptor[(size_t)i].bucket_ids[(size_t)g];
  }
}

// One pair as CascadeHashMatcher::Match stores it (:174-204) given the mean of the whole job.
// Returns the number of real records written; *n_default receives the number of {0, 0} records the
// reference puts in front of them (0 without cross-check).  Outputs need room for n_first records.
int64_t ch_match_pair(const float* first, int64_t n_first, const float* second, int64_t n_second, const float* mean128,
                      int cross_check, int64_t* n_default, int64_t* out_lhs, int64_t* out_rhs, float* out_dist) {
  const Projection pj = current_projection();
  const HashDescriptors hf = construct_hash_descriptors(first, n_first, mean128, pj);
  const HashDescriptors hs = construct_hash_descriptors(second, n_second, mean128, pj);
  const std::vector<RawMatch> m = match_pair(hf, first, hs, second, cross_check != 0, n_default);
  for (size_t k = 0; k < m.size(); k++) {
    out_lhs[k] = m[k].lhs_idx;
    out_rhs[k] = m[k].rhs_idx;
    if (out_dist) out_dist[k] = m[k].dist;
  }
  return (int64_t)m.size();
}

// The whole of CascadeHashMatcher::Match over a pair list (:118-205): images are those of the pair
// list, ids[k] with descs[k] / sizes[k]; pair_ids in std::set order.  Results are written pair after
// pair into out_lhs / out_rhs (capacity: sum over pairs of the first image's size); out_counts[p] and
// out_defaults[p] as in ch_match_pair.  threads > 1 spreads hashing and pairs over that many threads
// (the reference itself is single-threaded here); the results do not depend on it.
int64_t ch_match_set(const int64_t* ids, const float* const* descs, const int64_t* sizes, int64_t n_images,
                     const int64_t* pair_ids, int64_t n_pairs, int cross_check, int threads, int64_t* out_counts,
                     int64_t* out_defaults, int64_t* out_lhs, int64_t* out_rhs, double* seconds_hash,
                     double* seconds_match) {
  const Projection pj = current_projection();
  std::unordered_map<IndexT, size_t> slot;
  for (int64_t k = 0; k < n_images; k++) slot[ids[k]] = (size_t)k;
  std::vector<int64_t> order((size_t)n_images);
  const int64_t n_used = ch_image_order(pair_ids, n_pairs, order.data());
  std::vector<const float*> od;
  std::vector<int64_t> os;
  for (int64_t k = 0; k < n_used; k++) {
    od.push_back(descs[slot.at(order[(size_t)k])]);
    os.push_back(sizes[slot.at(order[(size_t)k])]);
  }
  float mean[kDim];
  ch_mean(od.data(), os.data(), n_used, mean);

  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  std::vector<HashDescriptors> hashed((size_t)n_images);
  const int nt = std::max(1, threads);
  {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
      pool.emplace_back([&, t] {
        for (int64_t k = t; k < n_used; k += nt) {
          const size_t s = slot.at(order[(size_t)k]);
          hashed[s] = construct_hash_descriptors(descs[s], sizes[s], mean, pj);
        }
      });
    for (auto& th : pool) th.join();
  }
  const double t1 = now();
  std::vector<std::vector<RawMatch>> results((size_t)n_pairs);
  {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
      pool.emplace_back([&, t] {
        for (int64_t p = t; p < n_pairs; p += nt) {
          const size_t a = slot.at(pair_ids[2 * p]), b = slot.at(pair_ids[2 * p + 1]);
          results[(size_t)p] = match_pair(hashed[a], descs[a], hashed[b], descs[b], cross_check != 0, &out_defaults[p]);
        }
      });
    for (auto& th : pool) th.join();
  }
  const double t2 = now();
  int64_t w = 0;
  for (int64_t p = 0; p < n_pairs; p++) {
    out_counts[p] = (int64_t)results[(size_t)p].size();
    for (const RawMatch& m : results[(size_t)p]) {
      out_lhs[w] = m.lhs_idx;
      out_rhs[w] = m.rhs_idx;
      w++;
    }
  }
  if (seconds_hash) *seconds_hash = t1 - t0;
  if (seconds_match) *seconds_match = t2 - t1;
  return w;
}

}  // extern "C"
