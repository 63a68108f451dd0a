// Device code of the brute-force descriptor matcher for gfx950 (MI355X).
//
// One image pair (X = first image, Y = second image) is handled as the
// product matrix S[i][j] = |x_i - y_j|^2 that is never stored:
//
//   k_distance_topk   S in 128 x chunk blocks on the fp32 matrix cores
//                     (v_mfma_f32_32x32x2_f32), squared distances in the form
//                     |x|^2 + |y|^2 - 2 x.y, and from every accumulator tile the
//                     smallest few entries of each row AND of each column,
//                     kept as 32-bit keys (distance bits with the low mantissa
//                     bits replaced by a local index).  Output: short candidate
//                     lists per row and per column.
//   k_rerank          merges the candidate lists of one query, recomputes the
//                     best four exactly as the reference does (src/metric.hpp:
//                     15-22: sequential fp32 sum of squared differences, then
//                     sqrt), applies the top-2 / ratio rule of
//                     src/feature_matcher/brute_force_matcher.cc:18-38, and
//                     proves from the list bounds that no other column could
//                     have changed the answer; queries where that proof fails
//                     are queued.
//   k_rescan          answers queued queries by the exact full scan.
//   k_cross_compact   cross-check of brute_force_matcher.cc:64-80 and ordered
//                     compaction into the output arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace sfm {

constexpr int DIM = 128;             // floats per descriptor
constexpr int BM = 256;              // rows of S per workgroup (4 waves x 64 rows)
constexpr int SN = 128;              // columns of S per LDS stage (4 tiles of 32)
constexpr int CHUNK_ALIGN = 2 * SN;  // chunks hold an even number of stages
constexpr int IMG_PAD = 256;         // images are padded to a multiple of this many rows
constexpr uint32_t SENT = 0xFFFFFFFFu;   // "empty" key, larger than every real key
// A key is the bit pattern of a computed squared distance with the sign bit cleared and the low
// mantissa bits replaced by a local index.  Rounding can leave a distance that is really ~0 slightly
// negative; clearing the sign turns it into an equally tiny positive key, so it still sorts first
// and unsigned integer order equals numeric order for every key.
// One key per accumulator element serves the element's row list and its column list, so its low
// bits name both the column tile (for the row list: the lane is the column inside the tile) and
// the row (for the column list): one v_and_or_b32 per element whose OR operand is a scalar that
// changes per tile and register, built on the scalar unit beside the matrix instructions.  Eight
// value bits of the mantissa remain, which is plenty for lists whose only job is to keep the two
// smallest of a lane's entries and a bound on the rest (the re-rank's intervals widen with them).
//   bits 0..3   accumulator register   } the row inside the 256-row block, permuted: col_key_row();
//   bit  4      half-wave              } bits 4, 6 and 7 are the same for all of a lane's elements and
//   bit  5      row tile of the wave   } are put in when a column list is parked (the tile number,
//   bits 6..7   wave                   } of no use to a column list, goes out at the same time)
//   bits 6..14  32-column tile number inside the chunk (row lists)
constexpr int KEY_IDX_BITS = 15;
constexpr uint32_t KEY_LOW = (1u << KEY_IDX_BITS) - 1u;
constexpr uint32_t KEY_MASK = 0x7FFFFFFFu & ~KEY_LOW;    // the value bits of a key
constexpr int KEY_TILE_SHIFT = 6, KEY_TILE_BITS = 9;
constexpr uint32_t KEY_TILE_FIELD = ((1u << KEY_TILE_BITS) - 1u) << KEY_TILE_SHIFT;
constexpr int MAX_CHUNK_COLS = 32 << KEY_TILE_BITS;   // 16384
constexpr uint32_t COL_LOW = 0xFFu;  // parked column keys: the row inside the 256-row block (permuted)
constexpr float PAD_NORM = 1e30f;    // squared norm given to padding rows: never a near neighbour
// Keys (and with them every bound the re-rank derives from keys) mean something only while all real
// rows of both images have squared norms below this: from PAD_NORM on the distance kernel gives a
// row the key of a padding row.  The reference has no such limit (src/metric.hpp:15-22 is plain
// fp32 arithmetic up to overflow), so a pair with a larger norm on either side is answered by the
// exact full scan, query by query.
constexpr float NORM_LIMIT = 1e29f;
constexpr int NCAND = 4;             // candidates re-ranked exactly per query
constexpr int ROWPART_FIELDS = 2 * NCAND + 1;  // keys, column indices, bound key

constexpr int LDS_STAGE_BYTES = SN * DIM * 4;                 // 64 KiB
constexpr int LDS_COLBUF_WORDS = 4 * 4 * 2 * 64;              // per parity: wave, tile, entry, lane (both half-waves)
constexpr int LDS_OFF_COLBUF = 2 * LDS_STAGE_BYTES;
constexpr int LDS_TOTAL = LDS_OFF_COLBUF + 2 * LDS_COLBUF_WORDS * 4;
constexpr int ROWBUF_STRIDE = 65;    // 64 keys per row + 1 word so that rows fall on different banks

// Row of the 256-row block that a column-list key's low bits name.
__host__ __device__ inline uint32_t col_key_row(uint32_t lo) {
  const uint32_t reg = lo & 15u, h = (lo >> 4) & 1u, rt = (lo >> 5) & 1u, w = lo >> 6;
  return 64u * w + 32u * rt + (reg & 3u) + 8u * (reg >> 2) + 4u * h;
}

// One image pair as the kernels see it.
struct PairDev {
  const float* x;        // [na_pad][128], padding rows zero
  const float* xn;       // [na_pad] squared norms (padding: PAD_NORM), then [na_pad] = max real norm
  const float* y;
  const float* yn;
  uint32_t* rowpart;     // [nchunks][ROWPART_FIELDS][na_pad]
  uint32_t* colpart;     // [na_pad/BM][3][nb_pad]
  int32_t* fwd_best;     // [na_pad] best column of each row, -1 if the ratio test fails
  float* fwd_dist;       // [na_pad] its exact distance
  int32_t* rev_best;     // [nb_pad]
  float* rev_dist;       // [nb_pad]
  int32_t na, nb, na_pad, nb_pad;
  int32_t nchunks, chunk_cols;
  int32_t out_slot;      // index of this pair in the batch's output arrays
  int32_t pad_;
  // split-half engine only: the images as scaled fp16 (high, low) planes and their scale factors
  const uint16_t* xh;    // [na_pad][256]: see k_split for the layout of a row
  const uint16_t* yh;
  const float* xscale;   // {2^e, 2^-e}: the plane holds x * 2^e
  const float* yscale;
};

// One workgroup of k_distance_topk.
struct BlockDev {
  int32_t pair;
  int32_t row0;      // first row of the 128-row block
  int32_t col0;      // first column of the chunk
  int32_t ncols;     // columns in the chunk (multiple of SN)
  int32_t chunk;     // chunk number (selects the row-partial slab)
  int32_t pad_;
  const void* xrows; // first of the block's 256 rows in the form the engine multiplies (512 bytes each):
                     // known one dependent load earlier than anything reached through the pair table
};
static_assert(sizeof(BlockDev) == 32, "BlockDev is uploaded as a packed table");

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// How k_distance_topk forms the products x.y:
//   ENGINE_F32   fp32 matrix instructions on the descriptors as they are;
//   ENGINE_F16X3 every value v (scaled by a power of two per image) is split into two fp16 numbers
//                hi = fp16(v), lo = fp16(v - hi), and x.y is taken as hi.hi + hi.lo + lo.hi on the
//                fp16 matrix instructions with fp32 accumulation.  What is dropped is below
//                2^-21 of |x||y|, the same order as the rounding of the fp32 chain, and is
//                covered by the completeness bound of k_rerank like every other rounding.
constexpr int ENGINE_F32 = 0;
constexpr int ENGINE_F16X3 = 1;

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
// Median of three; the compiler selects v_med3_u32 for this shape.
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
  return umin32(umax32(a, b), umax32(umin32(a, b), c));
}

// Smallest of three, as one instruction (the compiler does not always form it from two minima).
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Median of three as one instruction where the compiler, seeing what follows, would take the
// expression apart into four.
__device__ __forceinline__ uint32_t med3u_one(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Insert x into the ascending triple (a <= b <= c), dropping the largest of the four.
__device__ __forceinline__ void insert3(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t x) {
  c = med3u(b, c, x);
  b = med3u(a, b, x);
  a = a < x ? a : x;
}

// Insert x into the ascending pair (a <= b), dropping the largest of the three.
__device__ __forceinline__ void insert2(uint32_t& a, uint32_t& b, uint32_t x) {
  b = med3u(a, b, x);
  a = a < x ? a : x;
}

// Squared norms of the rows of one image, PAD_NORM for padding rows, the largest real norm
// in xn[n_pad] and the largest magnitude of any element in xn[n_pad + 1].  One wave per row,
// 64 rows per workgroup, one pair of atomics per workgroup.
constexpr int NORM_ROWS_PER_BLOCK = 64;
__global__ __launch_bounds__(256) void k_norms(const float* __restrict__ x, float* __restrict__ xn, int n,
                                               int n_pad) {
  __shared__ float wmax[4], wabs[4];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  float big = 0.0f, mag = 0.0f;          // over this wave's real rows
  for (int k = 0; k < NORM_ROWS_PER_BLOCK / 4; k++) {
    const int row = blockIdx.x * NORM_ROWS_PER_BLOCK + k * 4 + wv;
    if (row >= n_pad) break;
    if (row >= n) {
      if (l == 0) xn[row] = PAD_NORM;
      continue;
    }
    const float2 v = reinterpret_cast<const float2*>(x + (size_t)row * DIM)[l];
    float s = v.x * v.x + v.y * v.y;
    float m = fmaxf(fabsf(v.x), fabsf(v.y));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      m = fmaxf(m, __shfl_xor(m, o));
    }
    if (l == 0) xn[row] = s;
    // A row whose squared norm is not an ordinary number below NORM_LIMIT (NaN, infinity, overflow, or
    // so large that it is no longer told apart from a padding row) makes the image's largest norm
    // infinite: k_rerank then proves nothing from keys for any pair of this image.
    big = s < NORM_LIMIT ? fmaxf(big, s) : __builtin_inff();
    mag = fmaxf(mag, m);
  }
  if (l == 0) {
    wmax[wv] = big;
    wabs[wv] = mag;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // non-negative floats order like their bit patterns
    atomicMax(reinterpret_cast<unsigned int*>(xn + n_pad),
              __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    atomicMax(reinterpret_cast<unsigned int*>(xn + n_pad + 1),
              __float_as_uint(fmaxf(fmaxf(wabs[0], wabs[1]), fmaxf(wabs[2], wabs[3]))));
  }
}

// Split-half planes of one image for ENGINE_F16X3.  With e chosen so that the largest magnitude
// times 2^e lies in [512, 1024) (fp16 holds the high parts without overflow and the low parts of
// all but negligible elements as normal numbers), every v = x * 2^e is split into hi = fp16(v)
// and lo = fp16(v - hi), both round-to-nearest.  A row of the plane is 256 halves = 512 bytes in
// 16-byte pieces: for k-slab s (16 consecutive k) pieces 4s, 4s+1 hold hi[16s .. 16s+7],
// hi[16s+8 .. 16s+15] and pieces 4s+2, 4s+3 the same of lo -- the order in which the matrix
// instruction's lanes (row, half) consume them.  One thread per 8 elements.
__global__ __launch_bounds__(256) void k_split(const float* __restrict__ x, const float* __restrict__ xn,
                                               uint16_t* __restrict__ plane, float* __restrict__ scale, int n_pad) {
  const float amax = xn[n_pad + 1];
  int k = 0;
  (void)frexpf(amax, &k);                   // amax = m * 2^k, m in [0.5, 1)
  int e = amax > 0.0f ? 10 - k : 0;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  const float sc = ldexpf(1.0f, e);
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t == 0) {
    scale[0] = sc;
    scale[1] = ldexpf(1.0f, -e);
  }
  const size_t row = t >> 4;
  const int g = (int)(t & 15);              // elements 8g .. 8g+7: slab g>>1, half g&1
  if (row >= (size_t)n_pad) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(x + row * DIM + 8 * g);
  const f32x4 a = src[0], b = src[1];
  const float v[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
  f16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
  f16x8* dst = reinterpret_cast<f16x8*>(plane + row * 256);
  dst[4 * (g >> 1) + (g & 1)] = hi;
  dst[4 * (g >> 1) + 2 + (g & 1)] = lo;
}

// ---------------------------------------------------------------------------
// k_distance_topk
//
// Measured on gfx950: vector instructions do not execute beside fp32 MFMAs, not
// from the same wave and not from another wave of the SIMD (about 5 cycles of
// matrix-pipe time per vector instruction, and ~10 more per switch between the
// two kinds).  The kernel therefore spends as few vector instructions per tile as
// it can, keeps them in one run per pair of tiles, and uses one wave per SIMD with
// a large register budget instead of two waves that would only share the pipe.
//
// Workgroup = 256 threads = 4 waves; wave w owns rows row0+64w .. +63 (two row
// tiles) and keeps them, scaled by -2, as MFMA A-operands in 128 registers for
// the whole kernel (lane (r, h) holds x[row r][4(2s+h)+e] for s<16, e<4; the k
// order inside the contraction is permuted the same way for both operands).
// The Y columns of the chunk stream through LDS in stages of 128 rows x 512 B,
// loaded by global_load_lds (no registers), double buffered, one barrier per
// stage.  LDS rows are linear; 16-byte chunk c of row R is stored at chunk
// position c ^ (R & 15) so that the ds_read_b128 of an operand fragment (32 rows,
// same chunk) touches every bank group once.
//
//   acc = [ |x|^2 , 1 ] . [ 1 , |y|^2 ]           (one MFMA, k = 128..129)
//       + sum_k (-2 x_k) y_k                       (64 MFMAs)
//       = |x|^2 + |y|^2 - 2 x.y
//
// In the 32x32 accumulator layout lane (c, h) holds column c and rows
// (reg&3) + 8(reg>>2) + 4h.  Column minima are therefore lane-local; row minima
// are spread over the 32 lanes of a half-wave, so every lane keeps a private
// 2-entry list per row over the columns it sees (c, c+32, ...) and the lists are
// merged once, at the end of the chunk.  Two entries, not three: a list's second entry is
// also the bound on what the list dropped, and that is all the re-rank needs from it (it
// decides nearly every query from the two smallest keys and a bound), and one key serves an
// element's row list and its column list -- five vector instructions per accumulator element.  The price: a query whose two
// nearest columns share a list AND whose ratio test is too close for the keys to call
// cannot be proven complete from the lists and takes the exact full scan.
// ---------------------------------------------------------------------------
// Wave w fills rows 32w .. 32w+31 of a stage: 16 LDS-DMA instructions of 1 KiB (2 rows) each.
// Addressing is a buffer descriptor over the chunk's columns + a per-lane byte offset (voff: the
// swizzled position inside a 1 KiB piece, which repeats every 8 pieces) + a scalar offset per
// piece, so issuing a stage costs no vector instructions.
template <int I>
__device__ __forceinline__ void stage_load_piece(__amdgpu_buffer_rsrc_t ysrc, char* ldsbuf, int w,
                                                 const uint32_t (&voff)[8], uint32_t sbase) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrc, (__attribute__((address_space(3))) void*)(ldsbuf + (w * 16 + I) * 1024),
                                           16, voff[I & 7], sbase + I * 1024, 0, 0);
}
template <int... I>
__device__ __forceinline__ void stage_load(std::integer_sequence<int, I...>, __amdgpu_buffer_rsrc_t ysrc, char* ldsbuf,
                                           int w, const uint32_t (&voff)[8], uint32_t sbase) {
  (stage_load_piece<I>(ysrc, ldsbuf, w, voff, sbase), ...);
}

// Running top-5 of (key, index) pairs, ascending by key; branch-free insertion.
struct Top5 {
  uint32_t v0, v1, v2, v3, v4;
  uint32_t i0, i1, i2, i3, i4;
};
__device__ __forceinline__ void top5_init(Top5& t) {
  t.v0 = t.v1 = t.v2 = t.v3 = t.v4 = SENT;
  t.i0 = t.i1 = t.i2 = t.i3 = t.i4 = 0;
}
__device__ __forceinline__ void cswap(uint32_t& va, uint32_t& ia, uint32_t& vb, uint32_t& ib) {
  // afterwards va <= vb
  const bool s = vb < va;
  const uint32_t tv = s ? vb : va, ti = s ? ib : ia;
  vb = s ? va : vb;
  ib = s ? ia : ib;
  va = tv;
  ia = ti;
}
__device__ __forceinline__ void top5_put(Top5& t, uint32_t key, uint32_t idx) {
  const bool s = key < t.v4;
  t.v4 = s ? key : t.v4;
  t.i4 = s ? idx : t.i4;
  cswap(t.v3, t.i3, t.v4, t.i4);
  cswap(t.v2, t.i2, t.v3, t.i3);
  cswap(t.v1, t.i1, t.v2, t.i2);
  cswap(t.v0, t.i0, t.v1, t.i1);
}

// Value of v held by lane l^32, valid in lanes 0..31 (lanes 32..63 get their own back).
__device__ __forceinline__ uint32_t other_half(uint32_t v) {
  // swap(vdst, src) exchanges the upper half of vdst with the lower half of src; with both
  // operands equal to v the second result holds v's upper half in every lane below 32.
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return r[1];
}

// One 16-byte LDS read whose completion the compiler does not track: the caller
// retires it with lds_wait<N>() before the first use.  OFF is an immediate byte offset.
template <int OFF>
__device__ __forceinline__ void lds_read16(f32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// Wait until at most N of this wave's LDS operations are still outstanding, and keep the
// compiler from moving anything across the wait (it cannot see what the wait is for).
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr uint32_t COL_PARK_MASK = ~KEY_TILE_FIELD;   // a column key as it is parked: without the tile number

// Private row lists of one lane for one 32-row tile: per accumulator register (= row)
// the two smallest keys over the columns this lane has seen.  The second one doubles as the bound
// on everything the lane saw and did not keep.
struct LaneLists {
  uint32_t r0[16], r1[16];
};

// Pin sixteen values to this program point (an empty asm that "modifies" them): the optimizer
// otherwise moves pure arithmetic on them across the hand-placed waits and barriers.
__device__ __forceinline__ void anchor16(uint32_t (&x)[16]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
               "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}

// B fragments in flight per lane: a fragment is fetched two steps ahead of its use, so three register
// sets rotate.  (A fourth set, so that a fetch never lands in the set the previous step has just read,
// changed nothing: 137.9 us per step either way.)
constexpr int NFRAG = 3;

// Step S of a tile pair: fetch B fragment S+2, retire fragment S, then the matrix instructions
// that consume it for each of the wave's two row tiles.  TOFF = byte offset of the column tile
// inside its stage buffer; ROT: fragment s lives in b[(s + ROT) % 3]; HAS_NEXT: the next column
// tile is in the same stage (16 KiB further) and its first two fragments are fetched by the last
// two steps.  Fragment S is the 16 bytes at piece 2S + h of the lane's row.
//   ENGINE_F32:   four k-steps of v_mfma_f32_32x32x2_f32 per row tile;
//   ENGINE_F16X3: even S = high halves of k-slab S/2: products with the high and the low halves of A;
//                 odd S = low halves of the slab: product with the high halves of A.
// What a step needs to send one 1 KiB piece of the NEXT column stage on its way (LDS-DMA).  The
// pieces of a stage are spread over the steps of the current stage, one every fourth step: issued
// in one burst they fill the memory pipeline's queue (LDS-DMA moves ~45 bytes per clock per
// compute unit) and every wave sits at its next memory instruction until the queue drains.
struct StageFeed {
  __amdgpu_buffer_rsrc_t src;
  char* ldsbuf;          // the buffer the current stage does not read
  uint32_t sbase;        // byte offset of this wave's share of the next stage in the chunk; past the
                         // chunk's end for the last stage, where the loads return nothing
  int w;
};

// The index bits of the keys of a tile pair's 32 accumulator registers (element E: register E & 15
// of row tile E >> 4), each a scalar register of its own.
struct KeyIndex {
  uint32_t c[32];
};

template <int ENGINE, int TOFF, int S, int ROT, bool HAS_NEXT, int PIECE>
__device__ __forceinline__ void pair_step(f32x16& acc0, f32x16& acc1, const f32x4 (&a0)[16], const f32x4 (&a1)[16],
                                          f32x4 (&b)[NFRAG], const uint32_t (&X)[8], const StageFeed& feed,
                                          const uint32_t (&yvoff)[8], uint32_t tile6, KeyIndex& kx) {
  if constexpr (S + 2 < 16) {
    lds_read16<TOFF + ((S + 2) >> 3) * 256>(b[(S + 2 + ROT) % NFRAG], X[(S + 2) & 7]);
    lds_wait<2>();
  } else if constexpr (HAS_NEXT) {
    lds_read16<TOFF + 32 * DIM * 4>(b[(S + 2 + ROT) % NFRAG], X[S + 2 - 16]);
    lds_wait<2>();
  } else {
    lds_wait<15 - S>();
  }
  const f32x4 bs = b[(S + ROT) % NFRAG];
  if constexpr (ENGINE == ENGINE_F32) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S].x, bs.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S].x, bs.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S].y, bs.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S].y, bs.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S].z, bs.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S].z, bs.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[S].w, bs.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[S].w, bs.w, acc1, 0, 0, 0);
  } else {
    const f16x8 bh = __builtin_bit_cast(f16x8, bs);
    if constexpr ((S & 1) == 0) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[S]), bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[S]), bh, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[S + 1]), bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[S + 1]), bh, acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[S - 1]), bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[S - 1]), bh, acc1, 0, 0, 0);
    }
  }
#ifndef SFM_ABLATE_GLDS
  if constexpr (PIECE >= 0) stage_load_piece<PIECE>(feed.src, feed.ldsbuf, feed.w, yvoff, feed.sbase);
#endif
  if constexpr (ENGINE == ENGINE_F32) {
    // two of the tile pair's key constants per step: scalar additions, which cost nothing beside the
    // matrix instructions (in the fold they would take a vector instruction's turn each)
#pragma unroll
    for (int e = 2 * S; e < 2 * S + 2; e++) {
      kx.c[e] = tile6 + (uint32_t)((e >> 4) * 32 + (e & 15));
      asm volatile("" : "+s"(kx.c[e]));   // a scalar register, and the sum as it is (not tile6 | low bits at the point of use)
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Which 1 KiB piece of the next stage leaves with step S of column tile CT (-1: none).
#ifdef SFM_EARLY_PIECES
// diagnostic build: all sixteen pieces in the first two column tiles, one every second step
#define SFM_PIECE_AT(CT, S) ((CT) < 2 && (S) % 2 == 1 ? 8 * (CT) + (S) / 2 : -1)
#else
#define SFM_PIECE_AT(CT, S) ((S) % 4 == 1 ? 4 * (CT) + (S) / 4 : -1)
#endif
template <int ENGINE, int TOFF, int ROT, bool HAS_NEXT, int CT, int... S>
__device__ __forceinline__ void pair_steps(std::integer_sequence<int, S...>, f32x16& acc0, f32x16& acc1,
                                           const f32x4 (&a0)[16], const f32x4 (&a1)[16], f32x4 (&b)[NFRAG],
                                           const uint32_t (&X)[8], const StageFeed& feed,
                                           const uint32_t (&yvoff)[8], uint32_t tile6, KeyIndex& kx) {
  // piece 4 CT + S/4 of the next stage leaves with step S = 1, 5, 9, 13 of column tile CT
  (pair_step<ENGINE, TOFF, S, ROT, HAS_NEXT, SFM_PIECE_AT(CT, S)>(acc0, acc1, a0, a1, b, X, feed, yvoff, tile6, kx),
   ...);
}

// Two keys into a sorted two-entry list (a <= b) at once: the smallest of {a, x, y}, and the second
// smallest of {a, b, x, y}, which is min(b, median of {a, x, y}) because b is not below a.  Three
// instructions where two single insertions take four.
__device__ __forceinline__ void insert_pair(uint32_t& a, uint32_t& b, uint32_t x, uint32_t y) {
  const uint32_t mid = med3u_one(a, x, y);
  a = min3u(a, x, y);
  b = umin32(b, mid);
}

// Fold one finished accumulator tile into the lane's row lists and column list: four vector
// instructions per element (the key, three for two keys into either list).
// kx = the index bits of the tile pair's keys (column-tile number << KEY_TILE_SHIFT, + RT * 32 + reg),
// one scalar per element, built beside the matrix instructions; RT = which of the wave's two row tiles.
// A row's keys arrive one per column tile, so the row lists take them two column tiles at a time:
// WHEN = 0 keeps this tile's keys in `held`, WHEN = 1 inserts them together with this tile's.
template <int RT, int WHEN>
__device__ __forceinline__ void tile_fold(const f32x16& acc, const KeyIndex& kx, uint32_t vmask, LaneLists& L, uint32_t& cl0,
                                          uint32_t& cl1, uint32_t (&held)[16]) {
  uint32_t key[16];
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    key[reg] = (__float_as_uint(acc[reg]) & vmask) | kx.c[RT * 16 + reg];
    if constexpr (WHEN == 0) held[reg] = key[reg];
    else insert_pair(L.r0[reg], L.r1[reg], held[reg], key[reg]);
  }
#pragma unroll
  for (int g = 0; g < 16; g += 2) insert_pair(cl0, cl1, key[g], key[g + 1]);
}

// One column tile (32 columns) against the wave's 64 rows: 2 x 65 MFMAs, then the vector
// epilogue for both accumulator tiles, then each lane's column list (over its half-wave's 32 rows)
// is parked in LDS.
#ifdef SFM_K1_STAMPS
#define SFM_FOLD_CYCLES_PARAM , unsigned long long& g_fold_cycles
#define SFM_FOLD_CYCLES_ARG , stamp_fold_cycles
#else
#define SFM_FOLD_CYCLES_PARAM
#define SFM_FOLD_CYCLES_ARG
#endif
template <int ENGINE, int CT>
__device__ __forceinline__ void tile_pair(const f32x4 (&a0)[16], const f32x4 (&a1)[16], float an0, float an1, float bn,
                                          f32x4 (&b)[NFRAG], const uint32_t (&X)[8], uint32_t tile6, uint32_t vmask,
                                          LaneLists& L0, LaneLists& L1, uint32_t (&held0)[16], uint32_t (&held1)[16], uint32_t lowc,
                                          int h, uint32_t* cb, const StageFeed& feed,
                                          const uint32_t (&yvoff)[8] SFM_FOLD_CYCLES_PARAM) {
  constexpr int TOFF = CT * 32 * DIM * 4;
#ifdef SFM_K1_STAMPS
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  // (the first two fragments of a stage's first column tile are requested at the top of the stage,
  // before the previous stage's column lists are merged: their trip through LDS then runs beside that work)
  f32x16 zero;
#pragma unroll
  for (int i = 0; i < 16; i++) zero[i] = 0.0f;
  f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(an0, bn, zero, 0, 0, 0);
  f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(an1, bn, zero, 0, 0, 0);
  KeyIndex kx;
  pair_steps<ENGINE, TOFF, (16 * CT) % NFRAG, (CT < 3), CT>(std::make_integer_sequence<int, 16>{}, acc0, acc1, a0, a1, b, X, feed,
                                                 yvoff, tile6, kx);

  uint32_t cl0 = SENT, cl1 = SENT;
#ifdef SFM_ABLATE_FOLD
  asm volatile("" ::"v"(acc0), "v"(acc1));   // diagnostic build: accumulators stay live, lists untouched
#else
#ifdef SFM_K1_STAMPS
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
  g_fold_cycles += (tf0 - tm0) << 32;   // high half: the matrix section of the tile pair
  __builtin_amdgcn_sched_barrier(0);
#endif
  tile_fold<0, CT & 1>(acc0, kx, vmask, L0, cl0, cl1, held0);
  tile_fold<1, CT & 1>(acc1, kx, vmask, L1, cl0, cl1, held1);
#ifdef SFM_K1_STAMPS
  anchor16(L0.r0); anchor16(L0.r1);
  anchor16(L1.r0); anchor16(L1.r1);
  asm volatile("" : "+v"(cl0), "+v"(cl1));
  __builtin_amdgcn_sched_barrier(0);
  g_fold_cycles += __builtin_amdgcn_s_memtime() - tf0;
  __builtin_amdgcn_sched_barrier(0);
#endif
#endif
  // every lane parks its list (over its half-wave's 32 rows of the column); the eight lists of a
  // column -- four waves, two half-waves -- are merged once per stage
  // (the tile number goes out, the lane's part of the row comes in: COL_PARK_MASK)
  cb[0] = (cl0 & COL_PARK_MASK) | lowc;
  cb[64] = (cl1 & COL_PARK_MASK) | lowc;
  // keep the epilogue where it is written: next to this tile pair, not sunk to the end of the stage
  anchor16(L0.r0); anchor16(L0.r1);
  anchor16(L1.r0); anchor16(L1.r1);
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------
// ENGINE_F16X3 only: software pipeline over the column tiles of a stage.  Beside
// v_mfma_f32_32x32x16_f16 -- unlike beside the fp32 matrix instructions -- vector instructions
// execute concurrently, at half their usual rate (tools/mfma_valu.hip, tools/split_step_bench.hip).
// So while the matrix instructions of column tile CT run into one pair of accumulators, the
// finished pair of column tile CT-1 is folded into the lists: three accumulator elements per
// even step (four matrix instructions, six vector instructions per element), one per odd step (two).
// ---------------------------------------------------------------------------
struct FoldState {
  uint32_t cl0, cl1;   // the column list being built for the tile that is folded
  uint32_t prev;       // the key of the element before this one, when that was an even one (keys go in two at a time)
};

// Element E of a finished tile pair: E < 16 is register E of the first row tile's accumulator.
// Keys go into the lists two at a time (insert_pair): the column list takes elements 2k and 2k + 1
// together; a row's keys arrive one per column tile, so WHEN = 0 (column tiles 0 and 2 of a stage)
// keeps the key in held[E] and WHEN = 1 (tiles 1 and 3) inserts it together with this one.
template <int E, int WHEN>
__device__ __forceinline__ void fold_elem(const f32x16& p0, const f32x16& p1, uint32_t tile6, uint32_t vmask, LaneLists& L0,
                                          LaneLists& L1, FoldState& f, uint32_t (&held)[32]) {
  constexpr int RT = E >> 4, REG = E & 15;
  const uint32_t gb = __float_as_uint(RT == 0 ? p0[REG] : p1[REG]);
  LaneLists& L = RT == 0 ? L0 : L1;
  const uint32_t key = (gb & vmask) | (tile6 + (uint32_t)(RT * 32 + REG));
  if constexpr (WHEN == 0) held[E] = key;
  else insert_pair(L.r0[REG], L.r1[REG], held[E], key);
  if constexpr ((E & 1) == 0) f.prev = key;
  else insert_pair(f.cl0, f.cl1, f.prev, key);
}
template <int WHEN, int... E>
__device__ __forceinline__ void fold_elems(std::integer_sequence<int, E...>, const f32x16& p0, const f32x16& p1,
                                           uint32_t tile6, uint32_t vmask, LaneLists& L0, LaneLists& L1, FoldState& f,
                                           uint32_t (&held)[32]) {
  (fold_elem<E, WHEN>(p0, p1, tile6, vmask, L0, L1, f, held), ...);
}

template <int CT, int S, bool FOLD, bool HAS_NEXT, int PIECE>
__device__ __forceinline__ void pipe_step(f32x16& acc0, f32x16& acc1, const f32x4 (&a0)[16], const f32x4 (&a1)[16],
                                          f32x4 (&b)[NFRAG], const uint32_t (&X)[8], const StageFeed& feed,
                                          const uint32_t (&yvoff)[8], const f32x16& p0, const f32x16& p1, uint32_t tvp,
                                          uint32_t vmask, LaneLists& L0, LaneLists& L1, FoldState& f, uint32_t (&held)[32]) {
  constexpr int TOFF = CT * 32 * DIM * 4;
  constexpr int WHEN = (CT + 1) & 1;   // the tile that is folded is CT - 1
  if constexpr (S + 2 < 16) {
    lds_read16<TOFF + ((S + 2) >> 3) * 256>(b[(16 * CT + S + 2) % NFRAG], X[(S + 2) & 7]);
    lds_wait<2>();
  } else if constexpr (HAS_NEXT) {
    lds_read16<TOFF + 32 * DIM * 4>(b[(16 * CT + S + 2) % NFRAG], X[S + 2 - 16]);
    lds_wait<2>();
  } else {
    lds_wait<15 - S>();
  }
  const f16x8 frag = __builtin_bit_cast(f16x8, b[(16 * CT + S) % NFRAG]);
  if constexpr ((S & 1) == 0) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[S]), frag, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[S]), frag, acc1, 0, 0, 0);
    if constexpr (FOLD) {
      fold_elem<4 * (S / 2), WHEN>(p0, p1, tvp, vmask, L0, L1, f, held);
      fold_elem<4 * (S / 2) + 1, WHEN>(p0, p1, tvp, vmask, L0, L1, f, held);
    }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[S + 1]), frag, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[S + 1]), frag, acc1, 0, 0, 0);
    if constexpr (FOLD) fold_elem<4 * (S / 2) + 2, WHEN>(p0, p1, tvp, vmask, L0, L1, f, held);
  } else {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[S - 1]), frag, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[S - 1]), frag, acc1, 0, 0, 0);
    if constexpr (FOLD) fold_elem<4 * (S / 2) + 3, WHEN>(p0, p1, tvp, vmask, L0, L1, f, held);
  }
#ifndef SFM_ABLATE_GLDS
  if constexpr (PIECE >= 0) stage_load_piece<PIECE>(feed.src, feed.ldsbuf, feed.w, yvoff, feed.sbase);
#endif
  __builtin_amdgcn_sched_barrier(0);
}

// Column tile CT of a stage into (acc0, acc1), folding the finished pair (p0, p1) of column tile
// CT - 1 meanwhile and parking its column lists at cbp.
template <int CT, bool FOLD, int... S>
__device__ __forceinline__ void pipe_tile(std::integer_sequence<int, S...>, f32x16& acc0, f32x16& acc1,
                                          const f32x4 (&a0)[16], const f32x4 (&a1)[16], float an0, float an1, float bn,
                                          f32x4 (&b)[NFRAG], const uint32_t (&X)[8], const StageFeed& feed,
                                          const uint32_t (&yvoff)[8], const f32x16& p0, const f32x16& p1, uint32_t tvp,
                                          uint32_t vmask, LaneLists& L0, LaneLists& L1, uint32_t (&held)[32], uint32_t lowc,
                                          uint32_t* cbp) {
  if constexpr (CT == 0) {
    lds_read16<0>(b[0], X[0]);
    lds_read16<0>(b[1], X[1]);
  }
  f32x16 zero;
#pragma unroll
  for (int i = 0; i < 16; i++) zero[i] = 0.0f;
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(an0, bn, zero, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(an1, bn, zero, 0, 0, 0);
  FoldState f{SENT, SENT, SENT};
  (pipe_step<CT, S, FOLD, (CT < 3), SFM_PIECE_AT(CT, S)>(acc0, acc1, a0, a1, b, X, feed, yvoff, p0, p1, tvp,
                                                                        vmask, L0, L1, f, held),
   ...);
  if constexpr (FOLD) {
    // all thirty-two elements went in beside the matrix instructions: the tile's column lists are complete
    cbp[0] = (f.cl0 & COL_PARK_MASK) | lowc;
    cbp[64] = (f.cl1 & COL_PARK_MASK) | lowc;
  }
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Merge the eight lists of one column of a finished stage (four waves x two half-waves, two entries
// each) and store the three smallest keys.  The third one is also the bound on everything the
// block left out for this column, so it is lowered to the smallest SECOND entry of the eight lists
// (what a list dropped is not below its second entry) -- which is one of the sixteen keys itself:
// when it undercuts the third smallest it is the second smallest, and the column's third plane
// then repeats the second.  Lane (c, hh) of the wave takes column c of the wave's column tile and
// the lists of waves 2hh and 2hh+1; the two halves are then merged with v_permlane32_swap.  cbaddr
// is the LDS byte address of the first of the lane's eight entries; it is made opaque to the
// optimizer so that the reads stay "one register + immediate offset".
__device__ __forceinline__ void stage_columns_out(uint32_t cbaddr, int hh, __amdgpu_buffer_rsrc_t dst, uint32_t voff,
                                                  uint32_t soff, uint32_t plane_bytes) {
  asm volatile("" : "+v"(cbaddr));
  const lds_u32* cb = (const lds_u32*)cbaddr;
  // all eight entries are requested before the first is used: one trip through LDS, not two
  uint32_t e0[4], e1[4];
#pragma unroll
  for (int li = 0; li < 4; li++) {       // (wave, half-wave) = (0,0), (0,1), (1,0), (1,1) relative to the first
    e0[li] = cb[(li >> 1) * 4 * 2 * 64 + (li & 1) * 32];
    e1[li] = cb[(li >> 1) * 4 * 2 * 64 + (li & 1) * 32 + 64];
  }
  asm volatile("" : "+v"(e0[0]), "+v"(e1[0]), "+v"(e0[1]), "+v"(e1[1]), "+v"(e0[2]), "+v"(e1[2]), "+v"(e0[3]), "+v"(e1[3]));
  uint32_t k0 = e0[0], k1 = e1[0], k2 = SENT;
  uint32_t second = k1;
#pragma unroll
  for (int li = 1; li < 4; li++) {
    insert3(k0, k1, k2, e0[li]);
    insert3(k0, k1, k2, e1[li]);
    second = second < e1[li] ? second : e1[li];
  }
  const uint32_t p0 = other_half(k0), p1 = other_half(k1), p2 = other_half(k2), ps = other_half(second);
  insert3(k0, k1, k2, p0);
  insert3(k0, k1, k2, p1);
  insert3(k0, k1, k2, p2);
  second = second < ps ? second : ps;
  k2 = k2 < second ? k2 : second;
  if (hh == 0) {
    __builtin_amdgcn_raw_buffer_store_b32(k0, dst, voff, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b32(k1, dst, voff, soff + plane_bytes, 0);
    __builtin_amdgcn_raw_buffer_store_b32(k2, dst, voff, soff + 2 * plane_bytes, 0);
  }
}

#ifdef SFM_K1_STAMPS
// Diagnostic build only (tools/k1_timing.hip): per workgroup clocks at start and end (cycle and
// real-time counters), after the prologue, before the row merge, and at three points inside the
// prologue (twelve slots per workgroup).
__device__ unsigned long long* g_stamps;
#endif

template <int ENGINE>
__global__ __launch_bounds__(256, 1) void k_distance_topk(const PairDev* __restrict__ pairs,
                                                          const BlockDev* __restrict__ blocks) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_TOTAL];
#ifdef SFM_K1_STAMPS
  if (threadIdx.x == 0) {
    g_stamps[12 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
    g_stamps[12 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  const BlockDev blk = blocks[blockIdx.x];
#ifdef SFM_K1_STAMPS
  asm volatile("" ::"s"(blk.row0), "s"(blk.xrows));
  if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 11] = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int r = l & 31;
  const int h = l >> 5;

  // byte offset of this lane inside piece i of a stage (piece = 2 rows): row parity, then the
  // 16-byte chunk position XOR-swizzled by the row number
  uint32_t yvoff[8];
#pragma unroll
  for (int i = 0; i < 8; i++) yvoff[i] = (uint32_t)((l >> 5) * (DIM * 4) + (((l & 31) ^ ((2 * i + (l >> 5)) & 15)) << 4));
  constexpr auto PIECES = std::make_integer_sequence<int, 16>{};

  // A operands of the wave's two row tiles, scaled by -2 (exact): 2 x 16 x 16 bytes per lane,
  // piece 2s + h of the lane's row for s < 16 (fp32 values, or fp16 pairs of the split planes).
  // A lane reading its own row from memory touches 64 different cache lines per instruction, so
  // the block's 256 rows take the same route as the columns instead: LDS-DMA in whole rows into
  // the two (still unused) stage buffers, then conflict-free 16-byte LDS reads.  The rows'
  // address comes with the block descriptor, so the transfer is on its way before the pair table
  // (one more dependent scalar load) has been read.
  {
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(blk.xrows), 0, BM * DIM * 4, 0x00020000);
    stage_load(PIECES, xsrc, smem, w, yvoff, (uint32_t)(w * 16 * 1024));
    stage_load(PIECES, xsrc, smem + LDS_STAGE_BYTES, w, yvoff, (uint32_t)(LDS_STAGE_BYTES + w * 16 * 1024));
  }
  __builtin_amdgcn_sched_barrier(0);

  const PairDev& P = pairs[blk.pair];
  uint32_t* colbuf = reinterpret_cast<uint32_t*>(smem + LDS_OFF_COLBUF);

  const int nstages = blk.ncols / SN;
  // the chunk's columns as a raw buffer (0x00020000: 32-bit raw data format)
  // (both engines stream 512-byte rows: 128 floats, or 128 + 128 halves)
  const char* yrows = ENGINE == ENGINE_F32 ? reinterpret_cast<const char*>(P.y) : reinterpret_cast<const char*>(P.yh);
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(yrows + (size_t)blk.col0 * DIM * 4), 0, blk.ncols * DIM * 4, 0x00020000);

  f32x4 a0[16], a1[16];
  float sigma = 1.0f;   // ENGINE_F16X3: the accumulators hold distances times 2^(ex + ey)
  {
#ifdef SFM_K1_STAMPS
    if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 6] = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SFM_K1_STAMPS
    if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 7] = __builtin_amdgcn_s_memtime();
#endif
    // row 64w + 32t + r of the block sits 32 KiB * w + 16 KiB * t + 512 r into the two buffers
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)(w * 32768 + r * (DIM * 4));
    const int c0 = h ^ (r & 15);
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const uint32_t at = base + (uint32_t)((((c0 ^ (2 * s)) & 15) << 4) + (s >> 3) * 256);
      a0[s] = *(const lds_f32x4*)at;
      a1[s] = *(const lds_f32x4*)(at + 16384);
    }
    if constexpr (ENGINE == ENGINE_F32) {
#pragma unroll
      for (int s = 0; s < 16; s++) {
        a0[s] = -2.0f * a0[s];
        a1[s] = -2.0f * a1[s];
      }
    } else {
      const f16x8 m2 = {-2, -2, -2, -2, -2, -2, -2, -2};
#pragma unroll
      for (int s = 0; s < 16; s++) {
        a0[s] = __builtin_bit_cast(f32x4, __builtin_bit_cast(f16x8, a0[s]) * m2);
        a1[s] = __builtin_bit_cast(f32x4, __builtin_bit_cast(f16x8, a1[s]) * m2);
      }
      sigma = P.xscale[0] * P.yscale[0];
    }
    __syncthreads();   // everyone has its rows: the buffers may take the first column stages
#ifdef SFM_K1_STAMPS
    if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 8] = __builtin_amdgcn_s_memtime();
#endif
  }
  // first column stage on its way
  stage_load(PIECES, ysrc, smem, w, yvoff, (uint32_t)(w * 16 * 1024));

  // Park the A operands in the accumulator half of the register file: only matrix instructions
  // read them (their sources may be AGPRs), which leaves the vector half to the lists and
  // accumulators that the epilogue's vector instructions work on.
  // (the fp32 instructions take single registers, the fp16 ones aligned groups of four)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if constexpr (ENGINE == ENGINE_F32) {
      asm volatile("" : "+a"(a0[i].x), "+a"(a1[i].x));
      asm volatile("" : "+a"(a0[i].y), "+a"(a1[i].y));
      asm volatile("" : "+a"(a0[i].z), "+a"(a1[i].z));
      asm volatile("" : "+a"(a0[i].w), "+a"(a1[i].w));
    } else {
      asm volatile("" : "+a"(a0[i]), "+a"(a1[i]));
    }
  }
  // operands of the norm MFMA: A = [|x_r|^2, 1], B = [1, |y_c|^2] over k = (h = 0, h = 1);
  // padding rows and columns get an infinite norm whatever the scale
  auto scaled_norm = [sigma](float n) { return n >= PAD_NORM ? __builtin_inff() : n * sigma; };
  const float an0 = h == 0 ? scaled_norm(P.xn[blk.row0 + 64 * w + r]) : 1.0f;
  const float an1 = h == 0 ? scaled_norm(P.xn[blk.row0 + 64 * w + 32 + r]) : 1.0f;

  LaneLists L0, L1;
#pragma unroll
  for (int i = 0; i < 16; i++) L0.r0[i] = L0.r1[i] = L1.r0[i] = L1.r1[i] = SENT;

  const uint32_t lowc = (uint32_t)((w << 6) | (h << 4));
  // LDS byte address of this lane's row inside a column tile, for each of the 8 chunk positions it
  // reads, in stage buffer 0 (X0) and stage buffer 1 (X1)
  uint32_t X0[8], X1[8];
  {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int c0 = h ^ (r & 15);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      X0[k] = base + (uint32_t)(r * (DIM * 4) + ((c0 ^ (2 * k)) << 4));
      X1[k] = X0[k] + LDS_STAGE_BYTES;
    }
  }
  // Column norms through a buffer descriptor as well: a load through a generic pointer would be a
  // FLAT instruction, which also counts in the LDS counter that the hand-counted waits below use
  // (every wait would then sit out a trip to memory).
  const __amdgpu_buffer_rsrc_t ynsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(P.yn + blk.col0), 0, blk.ncols * 4, 0x00020000);
  float nbn[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++)
    nbn[ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ynsrc, (r + 32 * ct) * 4, 0, 0));
  // this row block's column lists for the chunk's columns: three planes of nb_pad words
  const uint32_t plane_bytes = (uint32_t)P.nb_pad * 4u;
  const __amdgpu_buffer_rsrc_t coldst = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(P.colpart + (size_t)(blk.row0 / BM) * 3 * P.nb_pad + blk.col0), 0,
      2 * plane_bytes + (uint32_t)blk.ncols * 4u, 0x00020000);
  const uint32_t colvoff = (uint32_t)((32 * w + r) * 4);
  const uint32_t colbuf_addr =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + LDS_OFF_COLBUF +
      (uint32_t)((((2 * h) * 4 + w) * 2 * 64 + r) * 4);   // wave 2h, column tile w, entry 0, lane r of half-wave 0

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // stage 0 landed
#ifdef SFM_K1_STAMPS
  if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 4] = __builtin_amdgcn_s_memtime();
#endif

#ifdef SFM_K1_STAMPS
  unsigned long long stamp_dma_wait = 0, stamp_barrier_wait = 0;   // summed over the stages, this wave
  unsigned long long stamp_fold_cycles = 0;                        // summed over the tile pairs (fp32 engine)
  unsigned long long stamp_head = 0;                               // stage entry to its first tile pair
#endif
  // Chunks hold a multiple of 256 columns, so stages come in pairs and each stage body is
  // compiled for a fixed LDS buffer.
  auto stage_body = [&](auto UC, int gg) __attribute__((always_inline)) {
    constexpr int u = decltype(UC)::value;
#ifdef SFM_K1_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long th0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    f32x4 b[NFRAG];
    const uint32_t (&X)[8] = u == 0 ? X0 : X1;
    if constexpr (ENGINE == ENGINE_F32) {
      lds_read16<0>(b[0], X[0]);
      lds_read16<0>(b[1], X[1]);
    }
    float bn[4];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) bn[ct] = h == 0 ? 1.0f : scaled_norm(nbn[ct]);
    // the next stage's columns leave piece by piece during this stage (past the last stage the
    // offsets lie beyond the descriptor's range: those loads fetch nothing)
    const StageFeed feed{ysrc, smem + (u ^ 1) * LDS_STAGE_BYTES, (uint32_t)((gg + 1) * LDS_STAGE_BYTES + w * 16 * 1024), w};
    if (gg + 1 < nstages) {
#pragma unroll
      for (int ct = 0; ct < 4; ct++)
        nbn[ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ynsrc, (r + 32 * ct) * 4, (gg + 1) * SN * 4, 0));
    }
    // column lists of the previous stage: each wave merges and stores 32 of its 128 columns
    if (gg >= 1)
      stage_columns_out(colbuf_addr + (u ^ 1) * LDS_COLBUF_WORDS * 4, h, coldst, colvoff, (uint32_t)((gg - 1) * SN * 4),
                        plane_bytes);
    // tile number of the stage's first column tile, in its place inside a key; the same in every lane
    const uint32_t tv = (uint32_t)(4 * gg) << KEY_TILE_SHIFT;
    constexpr uint32_t T1 = 1u << KEY_TILE_SHIFT;
    uint32_t vmask = KEY_MASK;
    asm volatile("" : "+v"(vmask));   // a VGPR: the key's other operands are the accumulator and a scalar, and an
                                      // instruction takes one scalar at most
    uint32_t* cb = colbuf + u * LDS_COLBUF_WORDS + (w * 4 * 2) * 64 + l;
    if constexpr (ENGINE == ENGINE_F32) {
#ifdef SFM_K1_STAMPS
      __builtin_amdgcn_sched_barrier(0);
      stamp_head += __builtin_amdgcn_s_memtime() - th0;
      __builtin_amdgcn_sched_barrier(0);
#endif
      uint32_t held0[16], held1[16];   // the keys of column tiles 0 and 2 until those of 1 and 3 join them in the row lists
      tile_pair<ENGINE, 0>(a0, a1, an0, an1, bn[0], b, X, tv + 0 * T1, vmask, L0, L1, held0, held1, lowc, h, cb + 0 * 128, feed, yvoff SFM_FOLD_CYCLES_ARG);
      tile_pair<ENGINE, 1>(a0, a1, an0, an1, bn[1], b, X, tv + 1 * T1, vmask, L0, L1, held0, held1, lowc, h, cb + 1 * 128, feed, yvoff SFM_FOLD_CYCLES_ARG);
      tile_pair<ENGINE, 2>(a0, a1, an0, an1, bn[2], b, X, tv + 2 * T1, vmask, L0, L1, held0, held1, lowc, h, cb + 2 * 128, feed, yvoff SFM_FOLD_CYCLES_ARG);
      tile_pair<ENGINE, 3>(a0, a1, an0, an1, bn[3], b, X, tv + 3 * T1, vmask, L0, L1, held0, held1, lowc, h, cb + 3 * 128, feed, yvoff SFM_FOLD_CYCLES_ARG);
    } else {
      constexpr auto STEPS = std::make_integer_sequence<int, 16>{};
      f32x16 e0, e1, o0, o1;   // accumulators of the even and of the odd column tiles
      uint32_t held[32];   // the keys of column tiles 0 and 2 until those of 1 and 3 join them in the row lists
      pipe_tile<0, false>(STEPS, e0, e1, a0, a1, an0, an1, bn[0], b, X, feed, yvoff, o0, o1, tv, vmask, L0, L1, held, lowc, cb);
      pipe_tile<1, true>(STEPS, o0, o1, a0, a1, an0, an1, bn[1], b, X, feed, yvoff, e0, e1, tv + 0 * T1, vmask, L0, L1, held, lowc, cb + 0 * 128);
      pipe_tile<2, true>(STEPS, e0, e1, a0, a1, an0, an1, bn[2], b, X, feed, yvoff, o0, o1, tv + 1 * T1, vmask, L0, L1, held, lowc, cb + 1 * 128);
      pipe_tile<3, true>(STEPS, o0, o1, a0, a1, an0, an1, bn[3], b, X, feed, yvoff, e0, e1, tv + 2 * T1, vmask, L0, L1, held, lowc, cb + 2 * 128);
      // the stage's last tile has no matrix work left to hide behind
      FoldState f{SENT, SENT, SENT};
      fold_elems<1>(std::make_integer_sequence<int, 32>{}, o0, o1, tv + 3 * T1, vmask, L0, L1, f, held);
      cb[3 * 128 + 0] = (f.cl0 & COL_PARK_MASK) | lowc;
      cb[3 * 128 + 64] = (f.cl1 & COL_PARK_MASK) | lowc;
      anchor16(L0.r0); anchor16(L0.r1);
      anchor16(L1.r0); anchor16(L1.r1);
    }
    // The next stage's LDS-DMA must have landed before anyone passes the barrier; the
    // compiler does not always drain it for __syncthreads(), so wait explicitly.
#ifdef SFM_K1_STAMPS
    const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long tb2 = __builtin_amdgcn_s_memtime();
    stamp_dma_wait += tb1 - tb0;
    stamp_barrier_wait += tb2 - tb1;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next stage landed; everyone done reading this one; column lists parked
#endif
  };
  // Chunks hold a multiple of 256 columns, so stages come in pairs and each stage body is
  // compiled for a fixed LDS buffer.
  for (int g = 0; g < nstages; g += 2) {
    stage_body(std::integral_constant<int, 0>{}, g);
    stage_body(std::integral_constant<int, 1>{}, g + 1);
  }
  stage_columns_out(colbuf_addr + ((nstages - 1) & 1) * LDS_COLBUF_WORDS * 4, h, coldst, colvoff,
                    (uint32_t)((nstages - 1) * SN * 4), plane_bytes);

#ifdef SFM_K1_STAMPS
  if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 9] = stamp_fold_cycles;   // high half: matrix sections, low half: folds
  if (threadIdx.x == 0) g_stamps[12 * blockIdx.x + 10] = (stamp_head << 32) | ((stamp_dma_wait + stamp_barrier_wait) & 0xFFFFFFFFull);
#endif
  // row lists: through LDS (the stage buffers are free now), one thread per row
  uint32_t* rowbuf = reinterpret_cast<uint32_t*>(smem);
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    const int R = 64 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    uint32_t* d0 = rowbuf + R * ROWBUF_STRIDE + r * 2;
    uint32_t* d1 = d0 + 32 * ROWBUF_STRIDE;
    d0[0] = L0.r0[reg]; d0[1] = L0.r1[reg];
    d1[0] = L1.r0[reg]; d1[1] = L1.r1[reg];
  }
  __syncthreads();
  {
    // Row tid: 32 sorted pairs.  The five smallest keys of all 64 come from the lists whose head
    // is among the five smallest heads, so rank the heads first and then add the second entries of
    // those five lists only.  While ranking, a key carries (lane, entry) in its low bits in place
    // of the tile number, so that five v_med3/v_min per insertion suffice (no index to drag
    // along); the winners' full keys are read back from LDS afterwards.  What a lane left out is
    // not below its second entry: the smallest of those bounds everything outside the 64 keys.
    const uint32_t* src = rowbuf + tid * ROWBUF_STRIDE;
    uint32_t m0 = SENT, m1 = SENT, m2 = SENT, m3 = SENT, m4 = SENT;
    auto put = [&](uint32_t x) {
      m4 = med3u(m3, m4, x);
      m3 = med3u(m2, m3, x);
      m2 = med3u(m1, m2, x);
      m1 = med3u(m0, m1, x);
      m0 = m0 < x ? m0 : x;
    };
    uint32_t lanebound = SENT;
#pragma unroll 8
    for (int p = 0; p < 32; p++) {
      put((src[p * 2] & KEY_MASK) | (uint32_t)p);
      const uint32_t k1 = src[p * 2 + 1];
      lanebound = lanebound < k1 ? lanebound : k1;
    }
    const uint32_t hp[5] = {m0 & 31u, m1 & 31u, m2 & 31u, m3 & 31u, m4 & 31u};
#pragma unroll
    for (int e = 0; e < 5; e++) put((src[hp[e] * 2 + 1] & KEY_MASK) | hp[e] | 32u);
    // low bits of a ranked key: lane (5 bits), entry in that lane's pair (1 bit)
    const uint32_t mk[4] = {m0, m1, m2, m3};
    uint32_t fullkey[4], lane[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      lane[e] = mk[e] & 31u;
      fullkey[e] = src[lane[e] * 2 + ((mk[e] >> 5) & 1u)];
    }
    const uint32_t fifth = m4 == SENT ? SENT : (m4 & KEY_MASK);   // value bits only: never above the real key
    const uint32_t bound = lanebound < fifth ? lanebound : fifth;
    uint32_t* dst = P.rowpart + (size_t)blk.chunk * ROWPART_FIELDS * P.na_pad + blk.row0 + tid;
    const size_t st = (size_t)P.na_pad;
    const uint32_t c0 = (uint32_t)blk.col0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      dst[e * st] = fullkey[e];
      dst[(4 + e) * st] = c0 + (((fullkey[e] & KEY_TILE_FIELD) >> KEY_TILE_SHIFT) << 5) + lane[e];
    }
    dst[8 * st] = bound;
  }
#ifdef SFM_K1_STAMPS
  __syncthreads();
  if (threadIdx.x == 0) {
    g_stamps[12 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    g_stamps[12 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---------------------------------------------------------------------------
// Exact arithmetic of the reference.
// ---------------------------------------------------------------------------

// src/metric.hpp:15-21 without the sqrt: one fp32 accumulator, index order,
// subtract / multiply / add each rounded on its own (no fused multiply-add).
__device__ __forceinline__ float ref_sumsq(const float* __restrict__ q, const float* __restrict__ c) {
#pragma clang fp contract(off)
  const f32x4* q4 = reinterpret_cast<const f32x4*>(q);
  const f32x4* c4 = reinterpret_cast<const f32x4*>(c);
  float sum = 0.0f;
#pragma unroll 8
  for (int k = 0; k < DIM / 4; k++) {
    f32x4 u = q4[k], v = c4[k];
    float t;
    t = u.x - v.x; sum = sum + t * t;
    t = u.y - v.y; sum = sum + t * t;
    t = u.z - v.z; sum = sum + t * t;
    t = u.w - v.w; sum = sum + t * t;
  }
  return sum;
}

// Running state of the scan at brute_force_matcher.cc:18-33, in a form that
// does not depend on visiting order: the smallest distance, the lowest index
// attaining it, and the second smallest distance counted with multiplicity.
// (The reference scans j upward with strict '<', which yields exactly that.)
struct Top2 {
  float d1; int32_t i1; float d2;
  float r2;   // second smallest squared sum, for the completeness proof
  float r1;
};
__device__ __forceinline__ void top2_init(Top2& t) {
  t.d1 = 3.402823466e+38f; t.i1 = -1; t.d2 = 3.402823466e+38f;
  t.r1 = __builtin_inff(); t.r2 = __builtin_inff();
}
__device__ __forceinline__ void top2_add(Top2& t, float d, float rsq, int32_t idx) {
  if (d < t.d1 || (d == t.d1 && t.i1 >= 0 && idx < t.i1)) {
    // new minimum (or same value at a lower index): old minimum becomes the second
    t.d2 = t.d1; t.d1 = d; t.i1 = idx;
  } else if (d < t.d2) {
    t.d2 = d;
  }
  if (rsq < t.r1) { t.r2 = t.r1; t.r1 = rsq; }
  else if (rsq < t.r2) t.r2 = rsq;
}
__device__ __forceinline__ void top2_merge(Top2& t, const Top2& o) {
  // fold o's two values in; o.d2 has no index of its own and never needs one
  if (o.i1 >= 0) {
    float od2 = o.d2;
    if (o.d1 < t.d1 || (o.d1 == t.d1 && (t.i1 < 0 || o.i1 < t.i1))) {
      float s = t.d1 < od2 ? t.d1 : od2;   // second = min(old minimum, o's second)
      t.d2 = s < t.d2 ? s : t.d2;
      // when t had a real minimum equal to o.d1 the old minimum is the second: covered by s
      t.d1 = o.d1; t.i1 = o.i1;
    } else {
      float s = o.d1 < t.d2 ? o.d1 : t.d2;
      t.d2 = s;
    }
  }
  // squared sums: merge two sorted pairs
  float a1 = t.r1, a2 = t.r2, b1 = o.r1, b2 = o.r2;
  float m1 = a1 < b1 ? a1 : b1;
  float hi = a1 < b1 ? b1 : a1;
  float lo2 = a2 < b2 ? a2 : b2;
  t.r1 = m1;
  t.r2 = hi < lo2 ? hi : lo2;
}

// Lower bound on the reference's squared sum of any column that is NOT among a
// query's candidates, given the smallest key such a column can have.
//   g  = computed |x|^2+|y|^2-2x.y; key value <= |g| (truncation only lowers it; a negative g has
//        |g| below the error term that is subtracted, so its bound is <= 0 and still valid);
//   |g - t| <= 3.1*gamma128*(nx+ny) + u*g  (norm sums, the 128-term fma chain started at
//              -nx/2, the final fma), t = exact squared distance, u = 2^-24;
//   reference sum r >= t*(1 - 131u).
// The constants below are rounded up generously.
__device__ __forceinline__ double excluded_lower_bound(uint32_t boundkey, uint32_t mask, float nq, float nomax) {
  const double u = 5.9604644775390625e-8;
  double bv = (double)__uint_as_float(boundkey & mask);
  return bv * (1.0 - 160.0 * u) - 450.0 * u * ((double)nq + (double)nomax);
}

// The same for ENGINE_F16X3, whose keys are distances times sigma = sq * so (the two images'
// power-of-two scales).  In scaled units, with nx', ny' the scaled squared norms:
//   * norm sums: 128u each; the accumulation: one fp32 step for the norms and 24 fp16 steps
//     of 16 products (exact in fp32) each, at most 17 roundings per step on partial sums
//     below 2(nx'+ny')(1+2^-10): 3 + 24*17 = 411 roundings, 822u(nx'+ny');
//   * split residue of an element v: |v - hi - lo| <= 2^-22 |v|, or <= 2^-14 absolute should the
//     matrix unit flush an fp16 subnormal low part; the dropped lo.lo products are below
//     2^-22 |x_k||y_k|.  Together, times the factor 2 of -2x.y:
//       3 * 2^-22 * 2|x'||y'|  <=  12u (nx'+ny'),   2 * 2^-14 * sqrt(128) * (|x'| + |y'|).
// 1100u covers the relative part with margin; the absolute part is kept as it is.
__device__ __forceinline__ double excluded_lower_bound_split(uint32_t boundkey, uint32_t mask, float nq, float nomax,
                                                             float sq, float so) {
  const double u = 5.9604644775390625e-8;
  const double sigma = (double)sq * (double)so;
  const double bv = (double)__uint_as_float(boundkey & mask);
  const double floor_term = 2.0 * 6.103515625e-5 * 11.32 *
                            (sqrt((double)nq) * (double)sq + sqrt((double)nomax) * (double)so);
  const double scaled = bv * (1.0 - 160.0 * u) - 1100.0 * u * sigma * ((double)nq + (double)nomax) - floor_term;
  return scaled / sigma;
}

// The other side of the same analysis: an upper bound on the reference's squared sum of the column
// a key belongs to.  The key's value bits v satisfy |g| < v (1 + 2^-8) (the 15 low mantissa
// bits were replaced), t <= |g| (1 + u) + E with the same absolute error term E as above, and the
// reference's sum is at most t (1 + 160u).
constexpr double KEY_SLACK = 3.95e-3;   // a little over 2^-(23 - KEY_IDX_BITS)
static_assert(KEY_IDX_BITS == 15, "KEY_SLACK goes with the number of mantissa bits a key keeps");
__device__ __forceinline__ double key_upper_bound(uint32_t key, uint32_t mask, float nq, float nomax) {
  const double u = 5.9604644775390625e-8;
  const double v = (double)__uint_as_float(key & mask);
  return v * (1.0 + KEY_SLACK + 400.0 * u) + 460.0 * u * ((double)nq + (double)nomax);
}
__device__ __forceinline__ double key_upper_bound_split(uint32_t key, uint32_t mask, float nq, float nomax, float sq,
                                                        float so) {
  const double u = 5.9604644775390625e-8;
  const double sigma = (double)sq * (double)so;
  const double v = (double)__uint_as_float(key & mask);
  const double floor_term = 2.0 * 6.103515625e-5 * 11.32 *
                            (sqrt((double)nq) * (double)sq + sqrt((double)nomax) * (double)so);
  return (v * (1.0 + KEY_SLACK + 400.0 * u) + 1120.0 * u * sigma * ((double)nq + (double)nomax) + floor_term * 1.001) /
         sigma;
}

// Largest float <= x and smallest float >= x (x finite, or +inf).
__device__ __forceinline__ float float_below(double x) {
  float f = (float)x;
  if ((double)f > x) f = __uint_as_float(f > 0.0f ? __float_as_uint(f) - 1u : (f == 0.0f ? 0x80000001u : __float_as_uint(f) + 1u));
  return f;
}
__device__ __forceinline__ float float_above(double x) {
  float f = (float)x;
  if ((double)f < x) f = __uint_as_float(f > 0.0f ? __float_as_uint(f) + 1u : (f == 0.0f ? 1u : __float_as_uint(f) - 1u));
  return f;
}

constexpr int RR_QUERIES = 64;       // queries per workgroup of k_rerank
// How a query is answered after its lists are merged (decided from the key values alone):
constexpr int RR_FULL = 0;     // exact distances of all four candidates, then the rule and the completeness check
constexpr int RR_REJECT = 1;   // whatever the exact values are, the ratio test fails: nothing to compute
constexpr int RR_ACCEPT = 2;   // the first candidate is the nearest and passes the ratio test whatever the exact
                               // values are: only its distance is computed (it is part of the output)
constexpr int RR_SQ_STRIDE = 132;    // words per row of squares (128 + 4: conflict-free b128 reads)

// k_rerank: grid (ceil(max queries / 64), pairs, directions), 256 threads.
// Direction 0: queries are rows of X, lists come from rowpart, candidates are rows of Y.
// Direction 1: queries are rows of Y, lists come from colpart, candidates are rows of X.
//   A. four lanes per query merge the query's partial lists (a quarter each, then two
//      shuffle rounds) into the best five keys and the bound key;
//   B. the workgroup computes the 128 squared differences of every (query, candidate)
//      with coalesced row loads, through LDS; one lane per (query, candidate) then adds
//      them up in index order and takes the square root: src/metric.hpp:15-22;
//   C. one lane per query applies brute_force_matcher.cc:26-38 to its four candidates
//      and checks the completeness bound;
//   D. queries that failed the check are re-answered by an exact scan of every
//      candidate row, by the whole workgroup.
// PP = 64: step B in four passes of 64 (query, candidate) pairs through 34 KiB of LDS, with coalesced
// row loads; for batches, where the step is bound by how many memory round trips are in flight (four
// workgroups per compute unit).
// PP = 0: the variant for launches that do not fill the chip (a single pair).  Step B without the
// transposition through LDS: one lane per (query, candidate) reads both rows itself and adds up in
// index order.  Its workgroups need 3.4 KiB of LDS and 104 registers, which is what a compute unit
// has left beside a workgroup of k_distance_topk (144 KiB of LDS, one wave of 320 registers per
// SIMD), so the re-rank of one pair can be resident while the next pair's distance kernel runs.
// (It does not come for free then: beside the fp32 matrix instructions every vector instruction of
// the re-rank still takes its turn on the SIMD; beside the fp16 ones it does overlap.)
template <int PP>
__global__ __launch_bounds__(256, 4) void k_rerank(const PairDev* __restrict__ pairs, double ratio, int engine,
                                                   unsigned long long* __restrict__ rescanned, int want_dist) {
  static_assert(PP == 0 || PP == 64, "two variants");
  __shared__ __attribute__((aligned(16))) float sq[PP > 0 ? PP * RR_SQ_STRIDE : DIM];
  __shared__ uint32_t cand[RR_QUERIES * NCAND];
  __shared__ uint32_t bnd[RR_QUERIES];
  __shared__ float res_r[RR_QUERIES * NCAND];
  __shared__ int unsafe_q[RR_QUERIES];
  __shared__ int mode[RR_QUERIES];
  __shared__ uint32_t acc_col[RR_QUERIES];   // the column of a query accepted from its keys alone
  __shared__ int n_unsafe;
  __shared__ Top2 wave_part[4];
  __shared__ uint16_t live[256];   // PP == 0: the (query, candidate) pairs that need a distance, densely
  __shared__ int wave_live[4];

#if defined(SFM_K2_STOP_AFTER) && SFM_K2_STOP_AFTER == 0
  return;   // diagnostic build: the launch alone
#endif
#ifdef SFM_RERANK_PRIO
  if constexpr (PP == 0) __builtin_amdgcn_s_setprio(SFM_RERANK_PRIO);
#endif
  const PairDev& P = pairs[blockIdx.y];
  const int dir = blockIdx.z;
  const int nq = dir == 0 ? P.na : P.nb;
  const int no = dir == 0 ? P.nb : P.na;
  const int q0 = blockIdx.x * RR_QUERIES;
  if (q0 >= nq) return;
  const float* qbase = dir == 0 ? P.x : P.y;
  const float* obase = dir == 0 ? P.y : P.x;
  const int tid = threadIdx.x;
  if (tid == 0) n_unsafe = 0;
  // what step C needs from memory, requested now: the query's squared norm and the largest norm
  // on the other side
  const float* qn_arr = dir == 0 ? P.xn : P.yn;
  const float my_qn = (tid < RR_QUERIES && q0 + tid < nq) ? qn_arr[q0 + tid] : 0.0f;
  const float other_max = (dir == 0 ? P.yn : P.xn)[dir == 0 ? P.nb_pad : P.na_pad];
  const float own_max = (dir == 0 ? P.xn : P.yn)[dir == 0 ? P.na_pad : P.nb_pad];
  // false: keys of this pair prove nothing (NaN compares false as well) -- no decision from keys
  // below, and every query fails the completeness check and takes the exact scan of step D
  const bool provable = other_max < NORM_LIMIT && own_max < NORM_LIMIT;

  // ---- A
  // First the three smallest keys of the query alone, values only (three v_med3/v_min per key): they
  // decide nearly every query -- "the ratio test fails whatever the exact values are" or "the
  // smallest key's column is the nearest and passes whatever they are" -- and such a query needs the
  // column of its smallest key at most.  Only a wavefront that holds a query they do not decide goes
  // through the full merge (five keys with their columns) below.
  int md_fast = RR_FULL;
  {
    const int ql = tid >> 2, sub = tid & 3;
    const int q = q0 + ql;
    const float nqv = q < nq ? qn_arr[q] : 0.0f;
    uint32_t a0 = SENT, a1 = SENT, a2 = SENT, lbound = SENT, first = 0;   // first: column (dir 0) or row block (dir 1) of a0
    if (q < nq) {
      if (dir == 0) {
        const size_t st = (size_t)P.na_pad;
        for (int c0 = sub; c0 < P.nchunks; c0 += 8) {
          uint32_t k[2][NCAND], ix[2][NCAND], b[2];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int c = c0 + 4 * u;
            const uint32_t* src = P.rowpart + (size_t)c * ROWPART_FIELDS * st + q;
#pragma unroll
            for (int e = 0; e < NCAND; e++) {
              k[u][e] = c < P.nchunks ? src[e * st] : SENT;
              ix[u][e] = c < P.nchunks ? src[(NCAND + e) * st] : 0u;
            }
            b[u] = c < P.nchunks ? src[2 * NCAND * st] : SENT;
          }
#pragma unroll
          for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int e = 0; e < NCAND; e++) {
              first = k[u][e] < a0 ? ix[u][e] : first;
              insert3(a0, a1, a2, k[u][e]);
            }
            lbound = lbound < b[u] ? lbound : b[u];
          }
        }
      } else {
        const size_t st = (size_t)P.nb_pad;
        const int nrb = P.na_pad / BM;
        for (int rb0 = sub; rb0 < nrb; rb0 += 32) {
          uint32_t k[8][3];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int rb = rb0 + 4 * u;
            const uint32_t* src = P.colpart + (size_t)rb * 3 * st + q;
#pragma unroll
            for (int e = 0; e < 3; e++) k[u][e] = rb < nrb ? src[e * st] : SENT;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            first = k[u][0] < a0 ? (uint32_t)(rb0 + 4 * u) : first;   // a list is ascending: its head decides
#pragma unroll
            for (int e = 0; e < 3; e++) insert3(a0, a1, a2, k[u][e]);
            lbound = lbound < k[u][2] ? lbound : k[u][2];
          }
        }
      }
    }
    // the four lanes of a query: two exchanges inside the quad (lanes l^1, then l^2)
#define SFM_QUAD_XCHG(v, ctrl) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, false)
#define SFM_QUAD_ROUND(ctrl)                                                                                   \
    {                                                                                                          \
      const uint32_t u0 = SFM_QUAD_XCHG(a0, ctrl), u1 = SFM_QUAD_XCHG(a1, ctrl), u2 = SFM_QUAD_XCHG(a2, ctrl); \
      const uint32_t ub = SFM_QUAD_XCHG(lbound, ctrl), uf = SFM_QUAD_XCHG(first, ctrl);                        \
      first = u0 < a0 ? uf : first;                                                                            \
      insert3(a0, a1, a2, u0);                                                                                 \
      insert3(a0, a1, a2, u1);                                                                                 \
      insert3(a0, a1, a2, u2);                                                                                 \
      lbound = lbound < ub ? lbound : ub;                                                                      \
    }
    SFM_QUAD_ROUND(0xB1)   // quad_perm [1, 0, 3, 2]
    SFM_QUAD_ROUND(0x4E)   // quad_perm [2, 3, 0, 1]
#undef SFM_QUAD_ROUND
    // The four lanes hold the same keys now (equal keys may have left different columns in `first`:
    // a query with equal smallest keys is never accepted from the keys alone).  One lane decides.
    int md = RR_FULL;
    if (sub == 0 && provable && q < nq && no >= NCAND && lbound != SENT && a2 != SENT &&
        (a2 & 0x7F800000u) != 0x7F800000u && ratio > 0.0 && ratio <= 1.0) {
      // Every column's reference sum lies in [lower bound, upper bound] of its key; the columns
      // outside the lists lie above the lower bound of the smallest list bound, and every column but
      // the two smallest keys' above the lower bound of the third smallest key:
      //   the nearest is >= min(lo_0, lx) and the second nearest <= hi_1:
      //       sqrt(min(lo_0, lx)) >= ratio * sqrt(hi_1)  =>  brute_force_matcher.cc:35 fails;
      //   column 0 <= hi_0 and everything else >= min(lo_1, lx):
      //       sqrt(hi_0) < ratio * sqrt(min(lo_1, lx))   =>  column 0 is the nearest and passes.
      // Square roots are taken of floats rounded outwards (sqrtf is monotone), in double as the rule.
      const uint32_t mask = KEY_MASK;
      const uint32_t xkey = lbound < a2 ? lbound : a2;
      double lo0, lo1, hi0, hi1, lx;
      if (engine == ENGINE_F32) {
        lo0 = excluded_lower_bound(a0, mask, nqv, other_max);
        lo1 = excluded_lower_bound(a1, mask, nqv, other_max);
        lx = excluded_lower_bound(xkey, mask, nqv, other_max);
        hi0 = key_upper_bound(a0, mask, nqv, other_max);
        hi1 = key_upper_bound(a1, mask, nqv, other_max);
      } else {
        const float sq = dir == 0 ? P.xscale[0] : P.yscale[0];
        const float so = dir == 0 ? P.yscale[0] : P.xscale[0];
        lo0 = excluded_lower_bound_split(a0, mask, nqv, other_max, sq, so);
        lo1 = excluded_lower_bound_split(a1, mask, nqv, other_max, sq, so);
        lx = excluded_lower_bound_split(xkey, mask, nqv, other_max, sq, so);
        hi0 = key_upper_bound_split(a0, mask, nqv, other_max, sq, so);
        hi1 = key_upper_bound_split(a1, mask, nqv, other_max, sq, so);
      }
      const double near_lo = fmax(0.0, fmin(lo0, lx)), rest_lo = fmax(0.0, fmin(lo1, lx));
      if ((double)sqrtf(float_below(near_lo)) >= ratio * (double)sqrtf(float_above(hi1))) md = RR_REJECT;
      else if ((double)sqrtf(float_above(hi0)) < ratio * (double)sqrtf(float_below(rest_lo))) md = RR_ACCEPT;
      uint32_t c0 = SENT;
      if (md == RR_ACCEPT) {
        c0 = dir == 0 ? first : first * BM + col_key_row(a0 & COL_LOW);
        if ((int)c0 >= no) md = RR_FULL;   // cannot happen for a finite key; the full merge sorts it out
      }
      if (md != RR_FULL) {
        // An accepted query's exact distance is part of the output only if the caller asked for
        // distances (the reference's Match record holds none): without them it needs no row loads.
        acc_col[ql] = c0;
        cand[tid + 0] = (md == RR_ACCEPT && !want_dist) ? SENT : c0;
        cand[tid + 1] = SENT;
        cand[tid + 2] = SENT;
        cand[tid + 3] = SENT;
        bnd[ql] = xkey;
        mode[ql] = md;
      }
    }
    md_fast = __builtin_amdgcn_update_dpp(0, md, 0x00, 0xF, 0xF, false);   // quad_perm [0, 0, 0, 0]: the deciding lane's
#undef SFM_QUAD_XCHG
  }
  if (__any(md_fast == RR_FULL)) {
    const int ql = tid >> 2, sub = tid & 3;
    const int q = q0 + ql;
    const float qn_early = q < nq ? qn_arr[q] : 0.0f;   // for the decision at the end of this step
    Top5 t;
    top5_init(t);
    uint32_t bound = SENT;
    if (q < nq) {
      if (dir == 0) {
        const size_t st = (size_t)P.na_pad;
        // two lists per trip to memory (loads first, then the inserts)
        for (int c0 = sub; c0 < P.nchunks; c0 += 8) {
          uint32_t k[2][NCAND], ix[2][NCAND], b[2];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int c = c0 + 4 * u;
            const uint32_t* src = P.rowpart + (size_t)c * ROWPART_FIELDS * st + q;
#pragma unroll
            for (int e = 0; e < NCAND; e++) {
              k[u][e] = c < P.nchunks ? src[e * st] : SENT;
              ix[u][e] = c < P.nchunks ? src[(NCAND + e) * st] : 0u;
            }
            b[u] = c < P.nchunks ? src[2 * NCAND * st] : SENT;
          }
#pragma unroll
          for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int e = 0; e < NCAND; e++) top5_put(t, k[u][e], ix[u][e]);
            bound = bound < b[u] ? bound : b[u];
          }
        }
      } else {
        const size_t st = (size_t)P.nb_pad;
        const int nrb = P.na_pad / BM;
        // eight lists per trip to memory (loads first, then the inserts)
        for (int rb0 = sub; rb0 < nrb; rb0 += 32) {
          uint32_t k[8][3];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int rb = rb0 + 4 * u;
            const uint32_t* src = P.colpart + (size_t)rb * 3 * st + q;
#pragma unroll
            for (int e = 0; e < 3; e++) k[u][e] = rb < nrb ? src[e * st] : SENT;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int rb = rb0 + 4 * u;
#pragma unroll
            for (int e = 0; e < 3; e++) {
              const uint32_t key = k[u][e];
              top5_put(t, key, (uint32_t)rb * BM + col_key_row(key & COL_LOW));
            }
            bound = bound < k[u][2] ? bound : k[u][2];   // a list's last key bounds what it left out
          }
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
      Top5 u;
      u.v0 = __shfl_xor(t.v0, o); u.v1 = __shfl_xor(t.v1, o); u.v2 = __shfl_xor(t.v2, o);
      u.v3 = __shfl_xor(t.v3, o); u.v4 = __shfl_xor(t.v4, o);
      u.i0 = __shfl_xor(t.i0, o); u.i1 = __shfl_xor(t.i1, o); u.i2 = __shfl_xor(t.i2, o);
      u.i3 = __shfl_xor(t.i3, o); u.i4 = __shfl_xor(t.i4, o);
      const uint32_t ob = __shfl_xor(bound, o);
      top5_put(t, u.v0, u.i0); top5_put(t, u.v1, u.i1); top5_put(t, u.v2, u.i2);
      top5_put(t, u.v3, u.i3); top5_put(t, u.v4, u.i4);
      bound = bound < ob ? bound : ob;
    }
    bound = bound < t.v4 ? bound : t.v4;
    // The four lanes now hold the same five entries, but equal keys may sit in a different
    // order in each lane: one lane alone names the candidates.
    if (sub == 0 && md_fast == RR_FULL) {
      uint32_t c0 = (t.v0 != SENT && (int)t.i0 < no) ? t.i0 : SENT;
      uint32_t c1 = (t.v1 != SENT && (int)t.i1 < no) ? t.i1 : SENT;
      uint32_t c2 = (t.v2 != SENT && (int)t.i2 < no) ? t.i2 : SENT;
      uint32_t c3 = (t.v3 != SENT && (int)t.i3 < no) ? t.i3 : SENT;
      // Most queries are decided by the keys alone.  Every column's reference sum lies in
      // [lower bound, upper bound] of its key; the columns outside the candidates lie above the
      // lower bound of the bound key.  The keys are sorted, so with lo_i, hi_i the interval of
      // candidate i and lx the excluded columns' lower bound:
      //   the nearest is >= min(lo_0, lx) and the second nearest <= hi_1:
      //       sqrt(min(lo_0, lx)) >= ratio * sqrt(hi_1)  =>  brute_force_matcher.cc:35 fails;
      //   candidate 0 <= hi_0 and everything else >= min(lo_1, lx):
      //       sqrt(hi_0) < ratio * sqrt(min(lo_1, lx))   =>  candidate 0 is the nearest and passes.
      // Square roots are taken of floats rounded outwards (sqrtf is monotone), in double as the rule.
      int md = RR_FULL;
      if (provable && q < nq && c0 != SENT && c1 != SENT && c2 != SENT && c3 != SENT && bound != SENT && ratio > 0.0 &&
          ratio <= 1.0) {
        const uint32_t mask = KEY_MASK;
        const float nqv = qn_early;
        double lo0, lo1, hi0, hi1, lx;
        if (engine == ENGINE_F32) {
          lo0 = excluded_lower_bound(t.v0, mask, nqv, other_max);
          lo1 = excluded_lower_bound(t.v1, mask, nqv, other_max);
          lx = excluded_lower_bound(bound, mask, nqv, other_max);
          hi0 = key_upper_bound(t.v0, mask, nqv, other_max);
          hi1 = key_upper_bound(t.v1, mask, nqv, other_max);
        } else {
          const float sq = dir == 0 ? P.xscale[0] : P.yscale[0];
          const float so = dir == 0 ? P.yscale[0] : P.xscale[0];
          lo0 = excluded_lower_bound_split(t.v0, mask, nqv, other_max, sq, so);
          lo1 = excluded_lower_bound_split(t.v1, mask, nqv, other_max, sq, so);
          lx = excluded_lower_bound_split(bound, mask, nqv, other_max, sq, so);
          hi0 = key_upper_bound_split(t.v0, mask, nqv, other_max, sq, so);
          hi1 = key_upper_bound_split(t.v1, mask, nqv, other_max, sq, so);
        }
        const double near_lo = fmax(0.0, fmin(lo0, lx)), rest_lo = fmax(0.0, fmin(lo1, lx));
        if ((double)sqrtf(float_below(near_lo)) >= ratio * (double)sqrtf(float_above(hi1))) md = RR_REJECT;
        else if ((double)sqrtf(float_above(hi0)) < ratio * (double)sqrtf(float_below(rest_lo))) md = RR_ACCEPT;
      }
      if (md == RR_REJECT) c0 = SENT;
      if (md != RR_FULL) {
        c1 = c2 = c3 = SENT;
        acc_col[ql] = c0;
        if (md == RR_ACCEPT && !want_dist) c0 = SENT;
      }
      cand[tid + 0] = c0;
      cand[tid + 1] = c1;
      cand[tid + 2] = c2;
      cand[tid + 3] = c3;
      bnd[ql] = bound;
      mode[ql] = md;
    }
  }
  __syncthreads();
#if defined(SFM_K2_STOP_AFTER) && SFM_K2_STOP_AFTER == 1
  return;   // diagnostic build: time of the list merge alone
#endif

  // ---- B
  if constexpr (PP == 0) {
    // the pairs that need a distance (typically one per accepted query, a quarter of the slots at
    // most) laid end to end, so that whole wavefronts have nothing to do
    const bool lv = cand[tid] != SENT;
    const unsigned long long bal = __ballot(lv);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wave_live[wv] = __popcll(bal);
    res_r[tid] = __builtin_inff();
    __syncthreads();
    int before = 0, n_live = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      before += k < wv ? wave_live[k] : 0;
      n_live += wave_live[k];
    }
    if (lv) live[before + __popcll(bal & (((unsigned long long)1 << lane) - 1))] = (uint16_t)tid;
    __syncthreads();
    if (tid < n_live) {
      const int p = live[tid];
      res_r[p] = ref_sumsq(qbase + (size_t)(q0 + (p >> 2)) * DIM, obase + (size_t)cand[p] * DIM);
    }
    __syncthreads();
  } else {
    const int w = tid >> 6, l = tid & 63, half = l >> 5, j = l & 31;
    // 256 / PP passes of PP (query, candidate) pairs.  A thread handles, for its 16 bytes of the
    // row, IT consecutive pairs = IT / 4 queries with their four candidates, so a query row is
    // loaded once per query.  A pass's row loads are issued as early as its registers are
    // free -- the next pass's right after this pass's squares went to LDS -- so that they travel
    // while the serial sums run.
    constexpr int IT = PP / 8, NPASS = 256 / PP;
    f32x4 uq[IT / 4], vc[IT];
    uint32_t cis[IT];
    auto issue_loads = [&](int pass) {
      const int p0 = pass * PP + w * (PP / 4) + half * IT;    // first pair of this thread in the pass
#pragma unroll
      for (int it = 0; it < IT; it++) cis[it] = cand[p0 + it];
#pragma unroll
      for (int g = 0; g < IT / 4; g++) {
        // a query decided by its keys needs no rows at all (its first candidate is the one kept, if any)
        const int q = q0 + ((p0 + 4 * g) >> 2);
        if ((cis[4 * g] & cis[4 * g + 1] & cis[4 * g + 2] & cis[4 * g + 3]) != SENT)
          uq[g] = reinterpret_cast<const f32x4*>(qbase + (size_t)(q < nq ? q : q0) * DIM)[j];
      }
#pragma unroll
      for (int it = 0; it < IT; it++)
        if (cis[it] != SENT) vc[it] = reinterpret_cast<const f32x4*>(obase + (size_t)cis[it] * DIM)[j];
    };
    issue_loads(0);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; pass++) {
#pragma unroll
      for (int it = 0; it < IT; it++) {
#pragma clang fp contract(off)
        const int pl = w * (PP / 4) + half * IT + it;      // (query, candidate) inside this pass
        const f32x4 u = uq[it >> 2], v = vc[it];
        f32x4 d;
        d.x = u.x - v.x; d.y = u.y - v.y; d.z = u.z - v.z; d.w = u.w - v.w;
        d.x = d.x * d.x; d.y = d.y * d.y; d.z = d.z * d.z; d.w = d.w * d.w;
        if (cis[it] != SENT) *reinterpret_cast<f32x4*>(sq + pl * RR_SQ_STRIDE + 4 * j) = d;
      }
      if (pass + 1 < NPASS) issue_loads(pass + 1);
      __syncthreads();
      if (tid < PP) {
#pragma clang fp contract(off)
        const int pi = pass * PP + tid;
        float sum = __builtin_inff();
        if (cand[pi] != SENT) {
          const f32x4* s4 = reinterpret_cast<const f32x4*>(sq + tid * RR_SQ_STRIDE);
          sum = 0.0f;
#pragma unroll 8
          for (int k = 0; k < DIM / 4; k++) {
            const f32x4 v = s4[k];
            sum = sum + v.x;
            sum = sum + v.y;
            sum = sum + v.z;
            sum = sum + v.w;
          }
        }
        res_r[pi] = sum;
      }
      __syncthreads();
    }
  }

#if defined(SFM_K2_STOP_AFTER) && SFM_K2_STOP_AFTER == 2
  // diagnostic build: list merge + exact distances (stored, so that they are not optimized away)
  if (q0 + (tid >> 2) < nq) (dir == 0 ? P.fwd_dist : P.rev_dist)[q0 + (tid >> 2)] = res_r[tid];
  return;
#endif
  // ---- C
  if (tid < RR_QUERIES && q0 + tid < nq && mode[tid] != RR_FULL) {
    const int q = q0 + tid;
    int32_t* best = dir == 0 ? P.fwd_best : P.rev_best;
    float* dist = dir == 0 ? P.fwd_dist : P.rev_dist;
    const bool accept = mode[tid] == RR_ACCEPT;
    best[q] = accept ? (int32_t)acc_col[tid] : -1;
    dist[q] = (accept && want_dist) ? sqrtf(res_r[tid * NCAND]) : 0.0f;
  } else if (tid < RR_QUERIES && q0 + tid < nq) {
    const int q = q0 + tid;
    Top2 t;
    top2_init(t);
#pragma unroll
    for (int c = 0; c < NCAND; c++) {
      const uint32_t ci = cand[tid * NCAND + c];
      // A column list may hold one row under two keys (its third key can repeat the second), so two
      // candidates can name the same descriptor: it is one descriptor and enters the top-2 once.
      bool seen = false;
#pragma unroll
      for (int e = 0; e < c; e++) seen = seen || cand[tid * NCAND + e] == ci;
      if (ci != SENT && !seen) {
        const float rsq = res_r[tid * NCAND + c];
        top2_add(t, sqrtf(rsq), rsq, (int32_t)ci);
      }
    }
    const uint32_t bound = bnd[tid];
    // (Few columns on the other side do not make the candidates complete by themselves: the four rows
    // of a four-row image sit in ONE lane's two-entry column list, which drops two of them -- by index
    // when their keys tie.  Only lists that dropped nothing, bound == SENT, need no proof.)
    bool safe = provable && bound == SENT;
    if (provable && !safe) {
      const uint32_t mask = KEY_MASK;
      if (engine == ENGINE_F32) {
        safe = (double)t.r2 <= excluded_lower_bound(bound, mask, my_qn, other_max);
      } else {
        const float sq = dir == 0 ? P.xscale[0] : P.yscale[0];
        const float so = dir == 0 ? P.yscale[0] : P.xscale[0];
        safe = (double)t.r2 <= excluded_lower_bound_split(bound, mask, my_qn, other_max, sq, so);
      }
    }
    const bool accept = (double)t.d1 < ratio * (double)t.d2;   // brute_force_matcher.cc:35
    int32_t* best = dir == 0 ? P.fwd_best : P.rev_best;
    float* dist = dir == 0 ? P.fwd_dist : P.rev_dist;
    best[q] = accept ? t.i1 : -1;
    dist[q] = t.d1;
    if (!safe) unsafe_q[atomicAdd(&n_unsafe, 1)] = q;
  }
  __syncthreads();

  // ---- D
  const int nu = n_unsafe;
  if (nu == 0) return;
  if (tid == 0) atomicAdd(rescanned, (unsigned long long)nu);
  float* qrow = sq;                                        // 128 floats
  for (int e = 0; e < nu; e++) {
    const int q = unsafe_q[e];
    __syncthreads();
    if (tid < DIM) qrow[tid] = qbase[(size_t)q * DIM + tid];
    __syncthreads();
    Top2 t;
    top2_init(t);
    for (int jj = tid; jj < no; jj += 256) {
      const float rsq = ref_sumsq(qrow, obase + (size_t)jj * DIM);
      top2_add(t, sqrtf(rsq), rsq, jj);
    }
    // the 256 partial states: a butterfly inside each wavefront (the two sides of every exchange
    // cover disjoint columns), then the four wavefronts' states through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      Top2 u;
      u.d1 = __shfl_xor(t.d1, o); u.i1 = __shfl_xor(t.i1, o); u.d2 = __shfl_xor(t.d2, o);
      u.r2 = __shfl_xor(t.r2, o); u.r1 = __shfl_xor(t.r1, o);
      top2_merge(t, u);
    }
    if ((tid & 63) == 0) wave_part[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) {
      Top2 r = wave_part[0];
      top2_merge(r, wave_part[1]);
      top2_merge(r, wave_part[2]);
      top2_merge(r, wave_part[3]);
      const bool accept = (double)r.d1 < ratio * (double)r.d2;
      int32_t* best = dir == 0 ? P.fwd_best : P.rev_best;
      float* dist = dir == 0 ? P.fwd_dist : P.rev_dist;
      best[q] = accept ? r.i1 : -1;
      dist[q] = r.d1;
    }
  }
}

// Cross-check (brute_force_matcher.cc:64-80) and ordered compaction.  A forward
// match (i -> j) is kept when the reverse direction matched j back to i, or
// when j has no surviving reverse match and i == 0 (the reference's map_[j]
// default-inserts 0).  Every thread owns eight consecutive rows of a sweep of 8 x THREADS rows, so
// output order is ascending row order.
// THREADS = 1024 for batches: one workgroup per pair (grid (pairs, 1)) goes through the sweeps one
// after the other.  THREADS = 256 (one wave per SIMD, next to no LDS) for launches of a few pairs,
// whose workgroups then fit beside a workgroup of k_distance_topk instead of holding up the one
// that the next launch wants to place on its compute unit; there the grid is (pairs, sweeps): a
// workgroup writes one sweep and first counts what the sweeps before it keep (the same loads, all
// in flight at once), so the pair takes one trip through its dependent loads instead of one per sweep.
// The workgroup of the last sweep in the grid writes the pair's count.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_cross_compact(const PairDev* __restrict__ pairs, int cross_check,
                                                        int32_t* __restrict__ out_counts,
                                                        int32_t* __restrict__ out_lhs,
                                                        int32_t* __restrict__ out_rhs,
                                                        float* __restrict__ out_dist, int64_t out_stride) {
  __shared__ int wsum[THREADS / 64];
  __shared__ int wearlier[THREADS / 64];
  __shared__ int base_s;
  const PairDev& P = pairs[blockIdx.x];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const size_t ob = (size_t)P.out_slot * (size_t)out_stride;
  constexpr int SWEEP = 8 * THREADS;
  const bool one_sweep_each = gridDim.y > 1;
  const int first = one_sweep_each ? (int)blockIdx.y * SWEEP : 0;
  const int last = one_sweep_each ? min(P.na, first + SWEEP) : P.na;

  // fwd arrays are padded to a multiple of 128 rows: vector loads stay in bounds
  auto decide = [&](int ib, int (&j)[8], float (&d)[8], bool (&keep)[8]) {
    if (ib < P.na) {
      const int4 a = *reinterpret_cast<const int4*>(P.fwd_best + ib);
      const int4 b = *reinterpret_cast<const int4*>(P.fwd_best + ib + 4);
      const float4 da = *reinterpret_cast<const float4*>(P.fwd_dist + ib);
      const float4 db = *reinterpret_cast<const float4*>(P.fwd_dist + ib + 4);
      j[0] = a.x; j[1] = a.y; j[2] = a.z; j[3] = a.w; j[4] = b.x; j[5] = b.y; j[6] = b.z; j[7] = b.w;
      d[0] = da.x; d[1] = da.y; d[2] = da.z; d[3] = da.w; d[4] = db.x; d[5] = db.y; d[6] = db.z; d[7] = db.w;
    }
    int rbv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = ib + k;
      keep[k] = ib < P.na && i < P.na && j[k] >= 0;
      rbv[k] = (keep[k] && cross_check) ? P.rev_best[j[k]] : 0;
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = ib + k;
      if (keep[k] && cross_check) keep[k] = rbv[k] == i || (rbv[k] < 0 && i == 0);
      cnt += keep[k] ? 1 : 0;
    }
    return cnt;
  };

  // what the sweeps before this workgroup's keep (nothing when one workgroup does them all)
  int earlier = 0;
  for (int i0 = 0; i0 < min(first, P.na); i0 += SWEEP) {
    int j[8];
    float d[8];
    bool keep[8];
    earlier += decide(i0 + tid * 8, j, d, keep);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) earlier += __shfl_xor(earlier, o);
  if (l == 0) wearlier[wv] = earlier;
  __syncthreads();
  if (tid == 0) {
    int s = 0;
    for (int k = 0; k < THREADS / 64; k++) s += wearlier[k];
    base_s = s;
  }
  __syncthreads();
  for (int i0 = first; i0 < last; i0 += SWEEP) {
    const int ib = i0 + tid * 8;
    int j[8];
    float d[8];
    bool keep[8];
    const int cnt = decide(ib, j, d, keep);
    // exclusive scan of cnt over the workgroup
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (l >= o) incl += v;
    }
    if (l == 63) wsum[wv] = incl;
    __syncthreads();
    int before = base_s;
    for (int k = 0; k < wv; k++) before += wsum[k];
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
      int s = base_s;
      for (int k = 0; k < THREADS / 64; k++) s += wsum[k];
      base_s = s;
    }
    __syncthreads();
  }
  if (tid == 0 && blockIdx.y == gridDim.y - 1) out_counts[P.out_slot] = base_s;
}

}  // namespace sfm
