// gfx950 kernels of the geometric verification of matches: per image pair a fundamental matrix by
// RANSAC over the eight-point fit, and the matches that agree with it.  The GPU side of the
// reference's ComputeFundamentalMatrix (src/sfm_main.cc:17-50), i.e. Ransac<...>::Inference
// (src/ransac.hpp:94-150) over EightPointFundamentalSolverImpl::Fit
// (src/solver/fundamental_solver.cc:13-86) with EpipolarLineError (src/solver/fundamental_solver.hpp:59-78).
//
// The reference runs its rounds one after the other because the number of rounds shrinks as better
// models turn up.  Here a launch evaluates 256 rounds of every unfinished pair at once, later launches 512 to 2,048 (one thread per
// round: sample and fit in one kernel, count in another), then one thread per pair replays the reference's bookkeeping over those
// 256 results in round order -- which yields exactly what the sequential loop would, the rounds
// past the stopping point being wasted work -- and the host launches again while any pair is unfinished.
//
// Arithmetic is fp64, every sum left to right with unfused multiply and add.  The singular vectors
// the reference takes from Eigen's JacobiSVD come from a cyclic Jacobi eigen-solver of the symmetric
// matrices A^T A and F^T F with a fixed number of sweeps, fully unrolled so that both matrices stay
// in registers.  The reference draws its samples from a std::mt19937 it seeds from
// std::random_device for every round; here the same selection-sampling algorithm (ransac.hpp:34-59)
// reads a counter-based stream keyed by (seed, pair, round), so a run can be repeated.
#ifndef SFM_AMD_CSRC_GEOMETRY_KERNELS_HPP_
#define SFM_AMD_CSRC_GEOMETRY_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sfm_geometry {

constexpr int SAMPLE = 8;                 // MINIMUM_DATA_POINT, fundamental_solver.hpp:21
constexpr uint32_t MAX_ROUNDS = 4096;     // MaxEpoch, ransac.hpp:97
constexpr int ROUNDS_PER_LAUNCH = 256;
constexpr int HYP_THREADS = 64;           // one wavefront: the fit needs most of the register file
constexpr int SWEEPS9 = 12, SWEEPS3 = 8;
constexpr int THR_CHUNK = 128;            // positions per table of sampling thresholds (k_geo_fit)

struct GeoState {
  double best_F[9];
  uint32_t rounds_limit;    // N of ransac.hpp:98, only ever lowered
  uint32_t rounds_done;     // sample_count
  uint32_t best_count;      // inliers of the best model so far
  uint32_t done;
};

struct GeoPair {
  const double* lhs;               // [n][2] pixel positions in the first image
  const double* rhs;               // [n][2] in the second
  const uint32_t* rounds_needed;   // [n + 1]: ransac.hpp:128-131 as a function of the inlier count
  GeoState* state;
  double* hyp_F;                   // [256][9] models of the current launch
  int32_t* hyp_idx;                // [256][8] the matches they were fitted to
  uint32_t* hyp_count;             // [256] their inlier counts   (each of the three times the launch's runs of 256 rounds)
  uint8_t* flags;                  // [n] final classification
  uint32_t* n_inliers;
  int32_t n;
  int32_t pad_;
  uint64_t stream;                 // key of this pair's random stream
};

// ---- random numbers: integer arithmetic only, the same bits on every machine ------------------
__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t pair_stream(uint64_t seed, uint64_t pair) {
  return mix64(mix64(seed ^ 0xD1B54A32D192ED03ull) + pair * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ uint64_t round_key(uint64_t stream, uint64_t round) { return stream ^ mix64(round + 0x632BE59BD9B4E019ull); }
__device__ __forceinline__ uint64_t stream_word(uint64_t key, uint64_t j) { return mix64(key + (j + 1) * 0x9E3779B97F4A7C15ull); }

// ---- the fit ----------------------------------------------------------------------------------
// One Jacobi rotation of the symmetric matrix S (upper triangle) in the (P, Q) plane, applied to
// the eigenvector matrix V as well.  Everything is indexed by compile-time constants.
template <int N, int P, int Q>
__device__ __forceinline__ void rotate_pq(double (&S)[N][N], double (&V)[N][N]) {
#pragma clang fp contract(off)
  const double apq = S[P][Q];
  if (apq == 0.0) return;
  // an entry that no longer registers against either diagonal entry is simply dropped; once every
  // lane of the wavefront is there, the later sweeps cost next to nothing
  const double small = 100.0 * fabs(apq);
  if (fabs(S[P][P]) + small == fabs(S[P][P]) && fabs(S[Q][Q]) + small == fabs(S[Q][Q])) {
    S[P][Q] = 0.0;
    return;
  }
  const double theta = (S[Q][Q] - S[P][P]) / (2.0 * apq);
  double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
  if (theta < 0.0) t = -t;
  const double c = 1.0 / sqrt(t * t + 1.0);
  const double s = t * c;
  const double tau = s / (1.0 + c);
  const double h = t * apq;
  S[P][P] = S[P][P] - h;
  S[Q][Q] = S[Q][Q] + h;
  S[P][Q] = 0.0;
#define SFM_GEO_ROTATE(x, y)                \
  {                                         \
    const double g_ = (x), k_ = (y);        \
    (x) = g_ - s * (k_ + g_ * tau);         \
    (y) = k_ + s * (g_ - k_ * tau);         \
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (k < P) SFM_GEO_ROTATE(S[k][P], S[k][Q])
    else if (k > P && k < Q) SFM_GEO_ROTATE(S[P][k], S[k][Q])
    else if (k > Q) SFM_GEO_ROTATE(S[P][k], S[Q][k])
  }
#pragma unroll
  for (int k = 0; k < N; k++) SFM_GEO_ROTATE(V[k][P], V[k][Q])
#undef SFM_GEO_ROTATE
}

template <int N, int P, int Q>
__device__ __forceinline__ void sweep_from(double (&S)[N][N], double (&V)[N][N]) {
  rotate_pq<N, P, Q>(S, V);
  if constexpr (Q + 1 < N) sweep_from<N, P, Q + 1>(S, V);
  else if constexpr (P + 2 < N) sweep_from<N, P + 1, P + 2>(S, V);
}

// Cyclic Jacobi: S becomes (nearly) diagonal, V its eigenvectors; returns the column of V that
// belongs to the diagonal entry of smallest magnitude in `out`.
template <int N>
__device__ __forceinline__ void smallest_eigenvector(double (&S)[N][N], int sweeps, double (&out)[N]) {
  double V[N][N];
#pragma unroll
  for (int a = 0; a < N; a++)
#pragma unroll
    for (int b = 0; b < N; b++) V[a][b] = a == b ? 1.0 : 0.0;
#pragma unroll 1
  for (int sw = 0; sw < sweeps; sw++) sweep_from<N, 0, 1>(S, V);
  int at = 0;
  double least = fabs(S[0][0]);
#pragma unroll
  for (int a = 1; a < N; a++) {
    const double m = fabs(S[a][a]);
    if (m < least) {
      least = m;
      at = a;
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    double v = V[k][0];
#pragma unroll
    for (int a = 1; a < N; a++) v = at == a ? V[k][a] : v;
    out[k] = v;
  }
}

// fundamental_solver.cc:13-53 for eight points (x, y, x', y'); F row-major
__device__ __forceinline__ void fit_eight(const double (&px)[SAMPLE], const double (&py)[SAMPLE], const double (&qx)[SAMPLE],
                                          const double (&qy)[SAMPLE], double (&F)[9]) {
#pragma clang fp contract(off)
  double S[9][9];
  {
    double A[SAMPLE][9];
#pragma unroll
    for (int i = 0; i < SAMPLE; i++) {   // fundamental_solver.cc:25-27
      A[i][0] = qx[i] * px[i]; A[i][1] = qx[i] * py[i]; A[i][2] = qx[i];
      A[i][3] = qy[i] * px[i]; A[i][4] = qy[i] * py[i]; A[i][5] = qy[i];
      A[i][6] = px[i];         A[i][7] = py[i];         A[i][8] = 1.0;
    }
#pragma unroll
    for (int a = 0; a < 9; a++)
#pragma unroll
      for (int b = a; b < 9; b++) {      // fundamental_solver.cc:30, the upper triangle
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < SAMPLE; i++) s = s + A[i][a] * A[i][b];
        S[a][b] = s;
      }
  }
  double f[9];
  smallest_eigenvector<9>(S, SWEEPS9, f);     // fundamental_solver.cc:32-38: last column of V
  // rank two (fundamental_solver.cc:44-52): U diag(s0, s1, 0) V^T = G - (G v) v^T with v the singular
  // vector of the smallest singular value, i.e. the eigenvector of G^T G of the smallest eigenvalue
  double M[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = a; b < 3; b++) M[a][b] = (f[a] * f[b] + f[3 + a] * f[3 + b]) + f[6 + a] * f[6 + b];
  double v[3];
  smallest_eigenvector<3>(M, SWEEPS3, v);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const double gv = (f[3 * r] * v[0] + f[3 * r + 1] * v[1]) + f[3 * r + 2] * v[2];
#pragma unroll
    for (int c = 0; c < 3; c++) F[3 * r + c] = f[3 * r + c] - gv * v[c];
  }
}

// fundamental_solver.hpp:59-72
__device__ __forceinline__ double epipolar_line_error(double x, double y, double xp, double yp, const double (&F)[9]) {
#pragma clang fp contract(off)
  const double lx = (F[0] * x + F[1] * y) + F[2];
  const double ly = (F[3] * x + F[4] * y) + F[5];
  const double rx = (F[0] * xp + F[3] * yp) + F[6];
  const double ry = (F[1] * xp + F[4] * yp) + F[7];
  const double rz = (F[2] * xp + F[5] * yp) + F[8];
  const double num = (rx * x + ry * y) + rz;
  const double deno = 1.0 / (lx * lx + ly * ly) + 1.0 / (rx * rx + ry * ry);
  return (num * num) * deno;
}

// `epipolar_line_error(...) > 3.84` (ransac.hpp:118), decided exactly but usually without the two
// divisions (about fifteen instructions each): with num^2 = nn and the two squared line norms s1, s2
// the error is nn (1/s1 + 1/s2), and in real numbers that exceeds 3.84 exactly when
// nn (s1 + s2) > 3.84 s1 s2.  Both sides are known to a few units in the last place, so unless they
// agree to one part in 10^9 the comparison says what the reference's own arithmetic says; the
// others, and anything out of the ordinary (a NaN fails every comparison and lands there too), take
// the exact route.
__device__ __forceinline__ bool epipolar_error_exceeds(double x, double y, double xp, double yp, const double (&F)[9]) {
#pragma clang fp contract(off)
  const double lx = (F[0] * x + F[1] * y) + F[2];
  const double ly = (F[3] * x + F[4] * y) + F[5];
  const double rx = (F[0] * xp + F[3] * yp) + F[6];
  const double ry = (F[1] * xp + F[4] * yp) + F[7];
  const double rz = (F[2] * xp + F[5] * yp) + F[8];
  const double num = (rx * x + ry * y) + rz;
  const double s1 = lx * lx + ly * ly, s2 = rx * rx + ry * ry, nn = num * num;
  const double lhs = nn * (s1 + s2), rhs = (3.84 * s1) * s2;
  // the right side a normal number: its product neither overflowed nor lost bits to underflow (the left
  // side may do either: it then compares as the infinite or vanishing quantity it stands for)
  const bool tame = __builtin_amdgcn_class(rhs, 0x100);   // positive normal
  if (tame && lhs > rhs * (1.0 + 1e-9)) return true;
  if (tame && lhs < rhs * (1.0 - 1e-9)) return false;
  const double deno = 1.0 / s1 + 1.0 / s2;
  return nn * deno > 3.84;
}

// ---------------------------------------------------------------------------------------------
// 256 rounds (or a few runs of 256) of every unfinished pair, in two kernels because they want opposite things from the
// register file: the fit keeps two matrices in registers and runs one wavefront per SIMD, the
// count needs few registers and hides its divisions behind eight -- and so does the sampler, whose
// walk over the matches is a chain of dependent integer instructions that a wavefront alone on its
// SIMD issues at two thirds of the rate.
//   k_geo_sample thread = one round: the eight matches it is fitted to.  grid: (4 * runs, pairs), 64 threads.
//   k_geo_fit    thread = one round: the fit.  grid: (4 * runs, pairs), 64 threads.
//   k_geo_count  thread = one round, workgroup = the 256 rounds of a pair: the matches pass through
//                LDS in tiles, every thread measures each of them against its model.  grid: (pairs, runs).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HYP_THREADS) void k_geo_sample(const GeoPair* __restrict__ pairs, uint32_t first_round) {
#pragma clang fp contract(off)
  const GeoPair P = pairs[blockIdx.y];
  const GeoState* st = P.state;
  if (st->done) return;
  const uint32_t slot = blockIdx.x * HYP_THREADS + threadIdx.x;
  const uint32_t round = first_round + slot;
  // the limit only ever comes down: a round at or beyond it is never looked at.  Its lane still
  // helps with the threshold table below.
  const bool active = round < st->rounds_limit;
  if (!__any(active)) return;
  const uint32_t n = (uint32_t)P.n;
  const uint64_t key = round_key(P.stream, round);
  // ransac.hpp:34-59: a rotating offset, then selection sampling in index order: match i is taken
  // when u_i < (8 - selected) / (n - i), u_i = (53 bits of the stream's word 1 + i) * 2^-53.
  int idx[SAMPLE];
#if defined(SFM_GEO_ABLATE) && SFM_GEO_ABLATE == 1
  // diagnostic build: no sampling (eight consecutive matches)
  for (int k = 0; k < SAMPLE; k++) idx[k] = (int)((round * 8u + k + (uint32_t)key) % n);
#else
  {
    // The walk looks at some 0.9 n matches per round and takes eight, so what it costs is what it
    // does per match it does NOT take.  u < p, with p the double quotient above, says the same as
    // (word >> 11) < ceil(p * 2^53) in integers (both scalings are exact), and p depends on the
    // position and on how many are taken already, not on the round: the wavefront tabulates the
    // thresholds of 128 positions x 8 counts at a time (sixteen divisions per lane where every lane
    // had 128) and a match is then one hash of the counter, one LDS read and one 64-bit compare.
    __shared__ uint64_t thr[SAMPLE][THR_CHUNK];
    __shared__ int32_t picks[SAMPLE][HYP_THREADS];
    const uint32_t lane = threadIdx.x;
    const uint32_t offset = (uint32_t)(stream_word(key, 0) % (uint64_t)(n + 1));
    uint32_t pos = offset >= n ? offset - n : offset;           // (offset + i) % n, kept up by counting
    uint64_t counter = key + 2 * 0x9E3779B97F4A7C15ull;           // stream_word(key, 1 + i) = mix64(key + (i + 2) * step)
    int selected = active ? 0 : SAMPLE;
    for (uint32_t i0 = 0; i0 < n; i0 += THR_CHUNK) {
      for (uint32_t e = lane; e < SAMPLE * THR_CHUNK; e += HYP_THREADS) {
        const uint32_t taken = e / THR_CHUNK, c = e % THR_CHUNK, i = i0 + c;
        uint64_t t = 0;
        if (i < n) {
          const double select_probability = (double)(SAMPLE - (int)taken) / (double)(n - i);
          t = (uint64_t)ceil(select_probability * 0x1.0p53);   // <= 2^56
        }
        thr[taken][c] = t;
      }
      __syncthreads();
      // Four positions per step, their thresholds for the count on entry and for one more read
      // before the hashes are computed (a wavefront alone on its SIMD has nothing else to cover an
      // LDS read with); a lane that takes two of the four reads again.  Positions past n have
      // threshold 0 and are never taken.
      const uint32_t lim = n - i0 < (uint32_t)THR_CHUNK ? n - i0 : (uint32_t)THR_CHUNK;
      for (uint32_t c = 0; c < lim && __any(selected < SAMPLE); c += 4) {
        const int s0 = selected & (SAMPLE - 1), s1 = (selected + 1) & (SAMPLE - 1);
        uint64_t ta[4], tb[4], word[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          ta[e] = thr[s0][c + e];
          tb[e] = thr[s1][c + e];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) word[e] = mix64(counter + (uint64_t)e * 0x9E3779B97F4A7C15ull) >> 11;
        counter += 4 * 0x9E3779B97F4A7C15ull;
        // the four hashes in one block of straight code, where their chains overlap (left alone the
        // compiler moves each one behind the branch of the position before it)
#pragma unroll
        for (int e = 0; e < 4; e++) asm volatile("" : "+v"(word[e]));
        const int on_entry = selected;
        int resume = 4;     // a lane that has taken two of the four goes on from here, reading again
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint64_t t = selected == on_entry ? ta[e] : tb[e];
          if (resume == 4 && selected < SAMPLE && word[e] < t) {
            const uint32_t at = pos + e;
            picks[selected][lane] = (int32_t)(at >= n ? at - n : at);
            selected++;
            if (selected == on_entry + 2) resume = e + 1;
          }
        }
        if (__any(resume < 4 && selected < SAMPLE)) {
#pragma unroll
          for (int e = 2; e < 4; e++) {
            if (e >= resume && selected < SAMPLE && word[e] < thr[selected & (SAMPLE - 1)][c + e]) {
              const uint32_t at = pos + e;
              picks[selected][lane] = (int32_t)(at >= n ? at - n : at);
              selected++;
            }
          }
        }
        pos += 4;
        pos = pos >= n ? pos - n : pos;
      }
      __syncthreads();
      if (!__any(selected < SAMPLE)) break;
    }
#pragma unroll
    for (int k = 0; k < SAMPLE; k++) idx[k] = k < selected ? picks[k][lane] : 0;
  }
#endif
  if (!active) return;
#pragma unroll
  for (int k = 0; k < SAMPLE; k++) P.hyp_idx[(size_t)slot * SAMPLE + k] = idx[k];
}

__global__ __launch_bounds__(HYP_THREADS) void k_geo_fit(const GeoPair* __restrict__ pairs, uint32_t first_round) {
#pragma clang fp contract(off)
  const GeoPair P = pairs[blockIdx.y];
  const GeoState* st = P.state;
  if (st->done) return;
  const uint32_t slot = blockIdx.x * HYP_THREADS + threadIdx.x;
  if (first_round + slot >= st->rounds_limit) return;   // as k_geo_sample decided
  int idx[SAMPLE];
#pragma unroll
  for (int k = 0; k < SAMPLE; k++) idx[k] = P.hyp_idx[(size_t)slot * SAMPLE + k];
  double px[SAMPLE], py[SAMPLE], qx[SAMPLE], qy[SAMPLE];
#pragma unroll
  for (int k = 0; k < SAMPLE; k++) {
    const double2 a = reinterpret_cast<const double2*>(P.lhs)[idx[k]];
    const double2 b = reinterpret_cast<const double2*>(P.rhs)[idx[k]];
    px[k] = a.x; py[k] = a.y; qx[k] = b.x; qy[k] = b.y;
  }
  double F[9];
#if defined(SFM_GEO_ABLATE) && SFM_GEO_ABLATE == 2
  // diagnostic build: no fit
  for (int k = 0; k < 9; k++) F[k] = px[k & 7] + qy[(k + 3) & 7];
#else
  fit_eight(px, py, qx, qy, F);
#endif
#pragma unroll
  for (int k = 0; k < 9; k++) P.hyp_F[(size_t)slot * 9 + k] = F[k];
}

constexpr int COUNT_TILE = 256;

__global__ __launch_bounds__(ROUNDS_PER_LAUNCH) void k_geo_count(const GeoPair* __restrict__ pairs, uint32_t first_round) {
#pragma clang fp contract(off)
  __shared__ double2 tile_l[COUNT_TILE], tile_r[COUNT_TILE];
  const GeoPair P = pairs[blockIdx.x];
  const GeoState* st = P.state;
  if (st->done) return;
  const uint32_t slot = blockIdx.y * ROUNDS_PER_LAUNCH + threadIdx.x;
  const bool live = first_round + slot < st->rounds_limit;   // the same rounds k_geo_fit worked on
  const int n = P.n;
  double F[9];
  int idx[SAMPLE];
#pragma unroll
  for (int k = 0; k < 9; k++) F[k] = live ? P.hyp_F[(size_t)slot * 9 + k] : 0.0;
#pragma unroll
  for (int k = 0; k < SAMPLE; k++) idx[k] = live ? P.hyp_idx[(size_t)slot * SAMPLE + k] : -1;
  // ransac.hpp:112-124: the sample is in unexamined, everything else unless its error exceeds 3.84.
  // Every match is measured here, the sample's eight among them; what those eight contributed is
  // taken off again below (asking "is this one of the eight" per match costs more than the eight).
  uint32_t count = SAMPLE;
  for (int i0 = 0; i0 < n; i0 += COUNT_TILE) {
    const int m = min(COUNT_TILE, n - i0);
    if ((int)threadIdx.x < m) {
      tile_l[threadIdx.x] = reinterpret_cast<const double2*>(P.lhs)[i0 + threadIdx.x];
      tile_r[threadIdx.x] = reinterpret_cast<const double2*>(P.rhs)[i0 + threadIdx.x];
    }
    __syncthreads();
    if (live) {
#pragma unroll 4
      for (int t = 0; t < m; t++) {
        const double2 a = tile_l[t], b = tile_r[t];
        count += epipolar_error_exceeds(a.x, a.y, b.x, b.y, F) ? 0u : 1u;
      }
    }
    __syncthreads();
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < SAMPLE; k++) {
      bool repeated = false;   // an index drawn twice is one match
#pragma unroll
      for (int j = 0; j < k; j++) repeated |= idx[j] == idx[k];
      if (!repeated && idx[k] >= 0 && idx[k] < n) {
        const double2 a = reinterpret_cast<const double2*>(P.lhs)[idx[k]];
        const double2 b = reinterpret_cast<const double2*>(P.rhs)[idx[k]];
        count -= epipolar_error_exceeds(a.x, a.y, b.x, b.y, F) ? 0u : 1u;
      }
    }
  }
  if (live) P.hyp_count[slot] = count;
}

// ---------------------------------------------------------------------------------------------
// ransac.hpp:126-142 over the launch's results in round order.  One wavefront per pair, four
// consecutive slots per lane.  The sequential loop
//     while (N > rounds) { rounds++; if (count == n) { take it; stop; }
//                          N = min(N, needed[count]); if (count > best) take it; }
// is a prefix minimum (N on entry to a slot), the first slot at which it stops -- before the slot
// when N <= rounds, after it when everything is an inlier -- and, over the slots before that point,
// the largest count at its first occurrence.  Slots the fit skipped (at or beyond N at the launch's
// start) hold stale counts; they lie behind the stopping point and never reach the result.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t u = __shfl_xor(v, o);
    v = u < v ? u : v;
  }
  return v;
}

// A launch holds `chunks` runs of 256 rounds per pair (slots 0 .. 256 * chunks - 1); they are replayed one
// after the other, the state handed on in registers, until the loop has stopped.
// `work` (may be null): two counters that receive the rounds this launch's fit and count worked on
// and the (match, round) pairs the count measured -- for the statistics, not for the result.
__global__ __launch_bounds__(64) void k_geo_bookkeeping(const GeoPair* __restrict__ pairs, int n_pairs, int chunks,
                                                        uint32_t* __restrict__ unfinished,
                                                        unsigned long long* __restrict__ work) {
  static_assert(ROUNDS_PER_LAUNCH == 256, "four slots per lane");
  const int p = blockIdx.x;
  if (p >= n_pairs) return;
  const GeoPair P = pairs[p];
  GeoState* st = P.state;
  if (st->done) return;
  uint32_t limit = st->rounds_limit, done = st->rounds_done, best = st->best_count;
  const int l = threadIdx.x;
  const uint32_t NONE = 0xFFFFFFFFu;
  if (work && l == 0) {   // the launch started at round `done` and skipped rounds at or beyond `limit`
    unsigned long long worked = 0;
    for (int c = 0; c < chunks; c++) {
      const uint32_t from = done + (uint32_t)(c * ROUNDS_PER_LAUNCH);
      if (limit > from) worked += min(limit - from, (uint32_t)ROUNDS_PER_LAUNCH);
    }
    if (worked) {
      atomicAdd(work, worked);
      atomicAdd(work + 1, worked * (unsigned long long)P.n);
    }
  }
  bool finished = false;
  for (int c = 0; c < chunks && !finished; c++) {
    const uint32_t limit0 = limit, done0 = done, best0 = best;
    const uint32_t base = (uint32_t)(c * ROUNDS_PER_LAUNCH);
    uint32_t cnt[4], need[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cnt[k] = P.hyp_count[base + 4 * l + k];
#pragma unroll
    for (int k = 0; k < 4; k++) need[k] = cnt[k] <= (uint32_t)P.n ? P.rounds_needed[cnt[k]] : NONE;
    // N on entry to every slot: limit0 and the needs of all earlier slots
    uint32_t lane_min = min(min(need[0], need[1]), min(need[2], need[3]));
    uint32_t incl = lane_min;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (l >= o) incl = min(incl, v);
    }
    uint32_t entry = __shfl_up(incl, 1);
    if (l == 0) entry = NONE;
    entry = min(entry, limit0);
    // first stop: event 2s = before slot s (N <= rounds so far), 2s + 1 = after slot s (all inliers)
    uint32_t ev = NONE;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t slot = (uint32_t)(4 * l + k);
      if (ev == NONE) {
        if (!(entry > done0 + slot)) ev = 2 * slot;
        else if (cnt[k] == (uint32_t)P.n) ev = 2 * slot + 1;
      }
      entry = min(entry, need[k]);
    }
    const uint32_t first_ev = wave_min_u32(ev);
    const bool stop_before = first_ev != NONE && (first_ev & 1u) == 0;
    const bool stop_after = first_ev != NONE && (first_ev & 1u) != 0;
    const uint32_t consumed = first_ev == NONE ? (uint32_t)ROUNDS_PER_LAUNCH : (stop_after ? first_ev / 2 + 1 : first_ev / 2);
    const uint32_t lowered_by = stop_after ? consumed - 1 : consumed;   // the all-inlier slot leaves N alone
    // N afterwards, and the largest count among the consumed slots at its first occurrence
    uint32_t lim = NONE;
    unsigned long long top = 0;   // count << 32 | (NONE - slot): larger count first, then the earlier slot
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t slot = (uint32_t)(4 * l + k);
      if (slot < lowered_by) lim = min(lim, need[k]);
      if (slot < consumed) {
        const unsigned long long key = ((unsigned long long)cnt[k] << 32) | (unsigned long long)(NONE - slot);
        top = key > top ? key : top;
      }
    }
    lim = min(wave_min_u32(lim), limit0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long u = __shfl_xor(top, o);
      top = u > top ? u : top;
    }
    const uint32_t top_count = (uint32_t)(top >> 32), top_slot = NONE - (uint32_t)(top & 0xFFFFFFFFull);
    const bool replaced = consumed > 0 && top_count > best0;
    const uint32_t done_rounds = done0 + consumed;
    finished = stop_before || stop_after || !(lim > done_rounds);
    if (replaced && l < 9) st->best_F[l] = P.hyp_F[(size_t)(base + top_slot) * 9 + l];
    limit = lim;
    done = done_rounds;
    best = replaced ? top_count : best0;
  }
  if (l == 0) {
    st->rounds_limit = limit;
    st->rounds_done = done;
    st->best_count = best;
    st->done = finished ? 1u : 0u;
    if (!finished) atomicAdd(unfinished, 1u);
  }
}

// ---------------------------------------------------------------------------------------------
// sfm_main.cc:32-42: the matches whose error under the final model is below 3.84.
// grid: (ceil(largest n / 256), pairs)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_geo_classify(const GeoPair* __restrict__ pairs) {
#pragma clang fp contract(off)
  const GeoPair P = pairs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P.n) return;
  double F[9];
#pragma unroll
  for (int k = 0; k < 9; k++) F[k] = P.state->best_F[k];
  const double2 a = reinterpret_cast<const double2*>(P.lhs)[i];
  const double2 b = reinterpret_cast<const double2*>(P.rhs)[i];
  const bool in = epipolar_line_error(a.x, a.y, b.x, b.y, F) < 3.84;
  P.flags[i] = in ? 1 : 0;
  const unsigned long long m = __ballot(in);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(P.n_inliers, (uint32_t)__popcll(m));
}

}  // namespace sfm_geometry

#endif  // SFM_AMD_CSRC_GEOMETRY_KERNELS_HPP_
