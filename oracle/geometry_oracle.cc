// CPU restatement of the reference's geometric verification of matches: GeometryFilter
// (src/sfm_main.cc:259-282) -> ComputeFundamentalMatrix (src/sfm_main.cc:17-50) -> RANSAC
// (src/ransac.hpp:85-150) over the eight-point fit (src/solver/fundamental_solver.cc:13-86) with
// the epipolar-line error (src/solver/fundamental_solver.hpp:59-78).
// TEST INFRASTRUCTURE: used by tests/ only, never by the sfm_amd package.
//
// PARITY UNPINNED, and with the reference itself it can only ever be statistical:
//   * the reference seeds a fresh std::mt19937 from std::random_device for every sample it draws
//     (ransac.hpp:45-46), so no two runs of the reference agree with each other;
//   * its fit takes the null vector from Eigen's JacobiSVD of A^T A and enforces rank two with a
//     second JacobiSVD (fundamental_solver.cc:30-52); Eigen is not available here, the file cannot be
//     built, and the reference holds no test or recorded output for this step.
// What is followed literally: the sampler's algorithm (rotating offset + selection sampling,
// ransac.hpp:34-59) with the random numbers as a parameter; the design matrix and its row order
// (fundamental_solver.cc:16-29); which singular vector is taken and how F is laid out (:36-42); the
// rank-two projection (:44-52); the error (fundamental_solver.hpp:59-72); RANSAC's bookkeeping --
// sample points count as inliers whatever their error, `!(error > 3.84)` otherwise, stop at once when
// everything is an inlier, N = min(N, ceil(log(0.01) / log(1 - w^8))), a later model only replaces
// the best on a strictly larger count, at most 4096 rounds (ransac.hpp:94-142); the final
// classification `error < 3.84` and the verdict `more than 30 inliers` (sfm_main.cc:32-45).
// What is fixed here by convention: the singular vectors come from a cyclic Jacobi eigen-solver of
// the symmetric matrices A^T A and F^T F with a fixed number of sweeps (the vector that belongs to the
// smallest |eigenvalue|; the rank-two projection as F - (F v) v^T, which is U diag(s0, s1, 0) V^T),
// every sum is taken left to right with unfused multiply and add, and two samplers are offered:
// the reference's own with std::mt19937 / std distributions seeded per call from a given seed, and
// the counter-based one the GPU uses (identical bits on both sides).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

namespace {

constexpr int kSample = 8;            // MINIMUM_DATA_POINT, fundamental_solver.hpp:21
constexpr size_t kMaxEpoch = 4096;    // ransac.hpp:97
constexpr double kReject = 3.84;      // fundamental_solver.hpp:74-77, sfm_main.cc:30

struct Point {
  double x, y, xp, yp;   // lhs (x, y), rhs (x', y')
};

// ---- random numbers --------------------------------------------------------------------------
inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// word j of the stream that belongs to (seed, pair, hypothesis)
inline uint64_t stream_word(uint64_t key, uint64_t j) { return mix64(key + (j + 1) * 0x9E3779B97F4A7C15ull); }
inline uint64_t stream_key(uint64_t seed, uint64_t pair, uint64_t hypothesis) {
  return mix64(mix64(seed ^ 0xD1B54A32D192ED03ull) + pair * 0x9E3779B97F4A7C15ull) ^ mix64(hypothesis + 0x632BE59BD9B4E019ull);
}

// ransac.hpp:34-59 with the counter-based stream: word 0 gives the offset, word 1 + i the i-th
// uniform number
void sample_counter(size_t n, uint64_t key, int* out) {
  const size_t offset = (size_t)(stream_word(key, 0) % (uint64_t)(n + 1));
  int selected = 0;
  for (size_t i = 0; i < n && selected < kSample; i++) {
    const double select_probability = double(kSample - selected) / double(n - i);
    const double u = (double)(stream_word(key, 1 + i) >> 11) * 0x1.0p-53;
    if (u < select_probability) out[selected++] = (int)((offset + i) % n);
  }
}

// ransac.hpp:34-59 as written, the per-call std::random_device replaced by a given seed
void sample_std(size_t n, uint32_t seed, int* out) {
  std::uniform_real_distribution<double> uniform_distribution(0.0, 1.0);
  std::uniform_int_distribution<size_t> start_distribu(0, n);
  std::mt19937 engine(seed);
  const size_t offset = start_distribu(engine);
  int selected = 0;
  for (size_t i = 0; i < n; i++) {
    const double select_probability = double(kSample - selected) / (n - i);
    const double sample_probability = uniform_distribution(engine);
    if (sample_probability < select_probability) {
      if (selected < kSample) out[selected] = (int)((offset + i) % n);
      selected++;
    }
  }
}

// ---- the fit ---------------------------------------------------------------------------------
// Cyclic Jacobi on a symmetric n x n matrix of which only the upper triangle (S[a][b], a <= b) is
// used and kept: it becomes (nearly) diagonal, V receives the eigenvectors.  A fixed number of
// sweeps, in which an off-diagonal entry too small to change either of its diagonal entries
// (100 |a_pq| is lost when added to them) is set to zero instead of rotated away; the rotation in the form that updates the two diagonal entries by t * S[p][q] and every
// other affected entry as g - s (h + g tau), h + s (g - h tau).
template <int N>
void jacobi(double (&S)[N][N], double (&V)[N][N], int sweeps) {
  for (int a = 0; a < N; a++)
    for (int b = 0; b < N; b++) V[a][b] = a == b ? 1.0 : 0.0;
  for (int sw = 0; sw < sweeps; sw++) {
    for (int p = 0; p < N - 1; p++) {
      for (int q = p + 1; q < N; q++) {
        const double apq = S[p][q];
        if (apq == 0.0) continue;
        // an entry that no longer registers against either diagonal entry is simply dropped
        const double small = 100.0 * std::fabs(apq);
        if (std::fabs(S[p][p]) + small == std::fabs(S[p][p]) && std::fabs(S[q][q]) + small == std::fabs(S[q][q])) {
          S[p][q] = 0.0;
          continue;
        }
        const double theta = (S[q][q] - S[p][p]) / (2.0 * apq);
        double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        if (theta < 0.0) t = -t;
        const double c = 1.0 / std::sqrt(t * t + 1.0);
        const double s = t * c;
        const double tau = s / (1.0 + c);
        const double h = t * apq;
        S[p][p] = S[p][p] - h;
        S[q][q] = S[q][q] + h;
        S[p][q] = 0.0;
        auto rotate = [&](double& x, double& y) {
          const double g = x, k = y;
          x = g - s * (k + g * tau);
          y = k + s * (g - k * tau);
        };
        for (int k = 0; k < p; k++) rotate(S[k][p], S[k][q]);
        for (int k = p + 1; k < q; k++) rotate(S[p][k], S[k][q]);
        for (int k = q + 1; k < N; k++) rotate(S[p][k], S[q][k]);
        for (int k = 0; k < N; k++) rotate(V[k][p], V[k][q]);
      }
    }
  }
}

template <int N>
int smallest(const double (&S)[N][N]) {
  int at = 0;
  for (int a = 1; a < N; a++)
    if (std::fabs(S[a][a]) < std::fabs(S[at][at])) at = a;
  return at;
}

constexpr int kSweeps9 = 12, kSweeps3 = 8;

// fundamental_solver.cc:13-53 for the eight points idx[0..7]; F row-major
void fit(const Point* pts, const int* idx, double* F) {
  double A[kSample][9];
  for (int i = 0; i < kSample; i++) {
    const Point& p = pts[idx[i]];
    const double row[9] = {p.xp * p.x, p.xp * p.y, p.xp, p.yp * p.x, p.yp * p.y, p.yp, p.x, p.y, 1.0};   // :25-27
    for (int c = 0; c < 9; c++) A[i][c] = row[c];
  }
  double S[9][9], V[9][9];
  for (int a = 0; a < 9; a++)
    for (int b = 0; b < 9; b++) {
      double s = 0.0;
      for (int i = 0; i < kSample; i++) s = s + A[i][a] * A[i][b];                                      // :30
      S[a][b] = s;
    }
  jacobi<9>(S, V, kSweeps9);
  const int at = smallest<9>(S);                                                                      // :38
  double G[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) G[r][c] = V[3 * r + c][at];                                           // :42
  double M[3][3], W[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) M[a][b] = (G[0][a] * G[0][b] + G[1][a] * G[1][b]) + G[2][a] * G[2][b];
  jacobi<3>(M, W, kSweeps3);
  const int w3 = smallest<3>(M);
  const double v[3] = {W[0][w3], W[1][w3], W[2][w3]};
  for (int r = 0; r < 3; r++) {
    const double gv = (G[r][0] * v[0] + G[r][1] * v[1]) + G[r][2] * v[2];
    for (int c = 0; c < 3; c++) F[3 * r + c] = G[r][c] - gv * v[c];                                    // :44-52
  }
}

// fundamental_solver.hpp:59-72
double epipolar_line_error(const Point& p, const double* F) {
  const double lx = (F[0] * p.x + F[1] * p.y) + F[2];
  const double ly = (F[3] * p.x + F[4] * p.y) + F[5];
  const double rx = (F[0] * p.xp + F[3] * p.yp) + F[6];
  const double ry = (F[1] * p.xp + F[4] * p.yp) + F[7];
  const double rz = (F[2] * p.xp + F[5] * p.yp) + F[8];
  const double num = (rx * p.x + ry * p.y) + rz;
  const double deno = 1.0 / (lx * lx + ly * ly) + 1.0 / (rx * rx + ry * ry);
  return (num * num) * deno;
}

// ransac.hpp:128-131 as an integer no larger than the round limit
size_t rounds_needed(size_t inliers, size_t n) {
  const double epsilon = (n - inliers) * 1.0 / n;
  const double t = std::ceil(std::log(1 - 0.99) / std::log(1 - std::pow(1 - epsilon, kSample)));
  // Few inliers: w^8 vanishes against 1, the quotient is -infinity, and the reference's conversion of
  // that to size_t yields (on x86-64) a number far above the limit, so N stays as it is.
  return (t >= 0.0 && t < (double)kMaxEpoch) ? (size_t)t : kMaxEpoch;
}

struct Verdict {
  double F[9];
  size_t rounds;
  size_t best_count;
};

// ransac.hpp:94-150.  sampler(k, out): the eight indices of round k.
template <class Sampler>
Verdict ransac(const Point* pts, size_t n, Sampler sampler) {
  Verdict v{};
  size_t N = kMaxEpoch, sample_count = 0;
  std::vector<char> is_sample(n);
  while (N > sample_count) {
    int idx[kSample];
    sampler(sample_count, idx);
    double F[9];
    fit(pts, idx, F);
    std::fill(is_sample.begin(), is_sample.end(), 0);
    for (int k = 0; k < kSample; k++) is_sample[(size_t)idx[k]] = 1;
    size_t count = kSample;                                   // :112-115: the sample is in, unexamined
    for (size_t i = 0; i < n; i++)
      if (!is_sample[i] && !(epipolar_line_error(pts[i], F) > kReject)) count++;   // :116-124
    sample_count++;
    if (count == n) {                                         // :127-131
      std::memcpy(v.F, F, sizeof(F));
      v.best_count = count;
      break;
    }
    N = std::min(N, rounds_needed(count, n));                 // :132-137
    if (count > v.best_count) {                               // :138-141
      std::memcpy(v.F, F, sizeof(F));
      v.best_count = count;
    }
  }
  v.rounds = sample_count;
  return v;
}

}  // namespace

extern "C" {

// The eight indices the counter-based sampler draws for (seed, pair, hypothesis) among n points.
void geo_sample(int64_t n, uint64_t seed, uint64_t pair, uint64_t hypothesis, int32_t* out8) {
  int idx[kSample];
  sample_counter((size_t)n, stream_key(seed, pair, hypothesis), idx);
  for (int k = 0; k < kSample; k++) out8[k] = idx[k];
}

// The eight-point fit of the points idx8 (xy arrays: 2 doubles per point); F: 9 doubles, row-major.
void geo_fit(const double* lhs_xy, const double* rhs_xy, const int32_t* idx8, double* F) {
  Point pts[kSample];
  int idx[kSample];
  for (int k = 0; k < kSample; k++) {
    pts[k] = Point{lhs_xy[2 * idx8[k]], lhs_xy[2 * idx8[k] + 1], rhs_xy[2 * idx8[k]], rhs_xy[2 * idx8[k] + 1]};
    idx[k] = k;
  }
  fit(pts, idx, F);
}

double geo_error(const double* lhs_xy, const double* rhs_xy, const double* F) {
  return epipolar_line_error(Point{lhs_xy[0], lhs_xy[1], rhs_xy[0], rhs_xy[1]}, F);
}

int64_t geo_rounds_needed(int64_t inliers, int64_t n) { return (int64_t)rounds_needed((size_t)inliers, (size_t)n); }

// ComputeFundamentalMatrix (sfm_main.cc:17-50) for one pair of n matches.  sampler: 0 the
// counter-based stream of (seed, pair); 1 the reference's own sampler with std::mt19937 seeded per
// round from that stream.  Returns 1 when more than 30 matches are inliers; inlier_flags[i] = 1 for
// `error < 3.84` under the final F; *rounds: RANSAC rounds run.  n < 8: returns 0, nothing computed
// (the reference asserts).
int geo_compute_fundamental(const double* lhs_xy, const double* rhs_xy, int64_t n, uint64_t seed, uint64_t pair,
                            int sampler, double* F, uint8_t* inlier_flags, int64_t* n_inliers, int64_t* rounds) {
  for (int k = 0; k < 9; k++) F[k] = 0.0;
  if (n_inliers) *n_inliers = 0;
  if (rounds) *rounds = 0;
  if (inlier_flags) std::memset(inlier_flags, 0, (size_t)std::max<int64_t>(n, 0));
  if (n < kSample) return 0;
  std::vector<Point> pts((size_t)n);
  for (int64_t i = 0; i < n; i++) pts[(size_t)i] = Point{lhs_xy[2 * i], lhs_xy[2 * i + 1], rhs_xy[2 * i], rhs_xy[2 * i + 1]};
  Verdict v;
  if (sampler == 0)
    v = ransac(pts.data(), (size_t)n, [&](size_t k, int* out) { sample_counter((size_t)n, stream_key(seed, pair, k), out); });
  else
    v = ransac(pts.data(), (size_t)n,
               [&](size_t k, int* out) { sample_std((size_t)n, (uint32_t)stream_word(stream_key(seed, pair, k), 0), out); });
  std::memcpy(F, v.F, sizeof(v.F));
  int64_t count = 0;
  for (int64_t i = 0; i < n; i++) {                           // sfm_main.cc:32-42
    const bool in = epipolar_line_error(pts[(size_t)i], v.F) < kReject;
    if (inlier_flags) inlier_flags[i] = in ? 1 : 0;
    count += in ? 1 : 0;
  }
  if (n_inliers) *n_inliers = count;
  if (rounds) *rounds = (int64_t)v.rounds;
  return count > 30 ? 1 : 0;                                  // :44
}

}  // extern "C"
