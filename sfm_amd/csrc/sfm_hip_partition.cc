// Host-only part of the C ABI (include/sfm_hip.h): static partition of an image-pair list over the
// GPUs of a node.  No HIP call is made here.
//
// The reference runs every (pair, direction) of the list as an independent task
// (src/feature_matcher/brute_force_matcher.cc:46-61) over a list that ExhaustivePairBuilder emits in
// lexicographic order (src/feature_matcher/exhaustive_pair_builder.cc:3-19).  Independence is what
// makes a static split correct; what the split has to get right is (i) equal cost per GPU, where a
// pair costs rows(first) x rows(second), and (ii) few distinct images per GPU, since a GPU uploads
// only the images its pairs name.  Cutting the lexicographic list into runs gives (i) for equal
// images only and fails (ii) outright: a run of the list names nearly every image.
//
// Here the pair (a, b) is a point of the upper triangle of the images x images square.  The square
// is cut into T x T tiles (T a power of two, about 4 sqrt(shards); tile borders at equal shares of
// all descriptor rows), the tiles are walked along a Hilbert curve, and the walk -- pairs of one tile
// in list order -- is cut into `shards` consecutive pieces of equal cost, at pair granularity.  A
// piece of a Hilbert walk is a compact blob of tiles, so a shard of 1/G of the triangle names about
// 2 n / sqrt(2 G) .. 2 n / sqrt(G) images instead of n.
#include "../../include/sfm_hip.h"

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <unordered_map>
#include <vector>

namespace {

// Position of cell (x, y) on the Hilbert curve of an n x n grid, n a power of two.
uint64_t hilbert_position(uint32_t n, uint32_t x, uint32_t y) {
  uint64_t d = 0;
  for (uint32_t s = n / 2; s > 0; s /= 2) {
    const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += (uint64_t)s * s * ((3u * rx) ^ ry);
    if (ry == 0) {
      if (rx == 1) {
        x = n - 1 - x;
        y = n - 1 - y;
      }
      std::swap(x, y);
    }
  }
  return d;
}

}  // namespace

extern "C" int sfm_hip_partition_pairs(const int64_t* pair_ids, int32_t n_pairs, const int64_t* image_ids,
                                       const int32_t* image_rows, int32_t n_images, int32_t n_shards,
                                       int32_t* out_order, int32_t* out_shard_begin) {
  if (n_pairs < 0 || n_images < 0 || n_shards <= 0 || !out_shard_begin) return SFM_HIP_ERR_INVALID;
  if (n_pairs > 0 && (!pair_ids || !out_order)) return SFM_HIP_ERR_INVALID;
  if (n_images > 0 && (!image_ids || !image_rows)) return SFM_HIP_ERR_INVALID;
  try {
    // images in ascending id: the axis of the triangle
    std::vector<int32_t> by_id((size_t)n_images);
    std::iota(by_id.begin(), by_id.end(), 0);
    std::sort(by_id.begin(), by_id.end(), [&](int32_t a, int32_t b) { return image_ids[a] < image_ids[b]; });
    uint32_t tiles = 1;
    if (n_shards > 1)
      while ((uint64_t)tiles * tiles < (uint64_t)16 * (uint64_t)n_shards && tiles * 2 <= (uint32_t)std::max(n_images, 1)) tiles *= 2;
    double all_rows = 0;
    for (int32_t i = 0; i < n_images; i++) {
      if (image_rows[i] < 0) return SFM_HIP_ERR_INVALID;
      all_rows += std::max(image_rows[i], 1);
    }
    struct Info {
      uint32_t tile;
      double rows;
    };
    std::unordered_map<int64_t, Info> info;
    info.reserve((size_t)n_images * 2);
    double before = 0;
    for (int32_t k = 0; k < n_images; k++) {
      const int32_t i = by_id[(size_t)k];
      const double rows = std::max(image_rows[i], 1);
      // the tile the middle of the image's rows falls into
      const uint32_t t = std::min<uint32_t>(tiles - 1, (uint32_t)((before + rows / 2) / all_rows * tiles));
      before += rows;
      if (!info.emplace(image_ids[i], Info{t, rows}).second) return SFM_HIP_ERR_INVALID;   // an id twice
    }

    std::vector<uint64_t> key((size_t)n_pairs);
    std::vector<double> cost((size_t)n_pairs);
    double total = 0;
    for (int32_t p = 0; p < n_pairs; p++) {
      const auto a = info.find(pair_ids[2 * (size_t)p]), b = info.find(pair_ids[2 * (size_t)p + 1]);
      if (a == info.end() || b == info.end()) return SFM_HIP_ERR_INVALID;
      const uint32_t lo = std::min(a->second.tile, b->second.tile), hi = std::max(a->second.tile, b->second.tile);
      key[(size_t)p] = hilbert_position(tiles, lo, hi);
      cost[(size_t)p] = a->second.rows * b->second.rows;
      total += cost[(size_t)p];
    }
    std::iota(out_order, out_order + n_pairs, 0);
    std::stable_sort(out_order, out_order + n_pairs,
                     [&](int32_t x, int32_t y) { return key[(size_t)x] < key[(size_t)y]; });
    // shard s takes the pairs whose middle, on the axis of accumulated cost, lies in share s
    int32_t s = 0;
    double acc = 0;
    out_shard_begin[0] = 0;
    for (int32_t k = 0; k < n_pairs; k++) {
      const double mid = acc + cost[(size_t)out_order[k]] / 2;
      while (s + 1 < n_shards && mid >= total * (double)(s + 1) / n_shards) out_shard_begin[++s] = k;
      acc += cost[(size_t)out_order[k]];
    }
    while (s < n_shards) out_shard_begin[++s] = n_pairs;
    return SFM_HIP_OK;
  } catch (...) {
    return SFM_HIP_ERR_NOMEM;
  }
}
