/*
 * CPU restatement of the reference brute-force descriptor matcher.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library or its host
 * wrappers may include, link or call this file; it exists so that the tests,
 * the smoke check and the benchmark's cpu_baseline leg have something to
 * compare the GPU path against.
 *
 * Every function names the reference lines it restates (paths relative to
 * the reference checkout).  The arithmetic is deliberately naive: one fp32
 * accumulator, terms added in index order, separate multiply and add (the
 * build recipe passes -ffp-contract=off), a correctly rounded sqrtf, and the
 * ratio comparison carried out in double exactly as the reference's
 * `min < 0.8 * second` promotes it.
 *
 * Pinning: bf_l2_metric is checked bit-for-bit against the reference's own
 * metric.hpp compiled from where it lies (oracle/_ref).  The matching
 * functor and the cross-check are checked against the known-answer cases
 * recorded from a run of the reference matcher (tests/golden/kat.json).
 * The reference's full matcher translation unit needs Eigen, which this
 * image does not have, so it is not built here.
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BF_DIM 128

/* src/metric.hpp:15-22 (L2_metric_trivial), reached through
 * L2_metric -> L2_metric_aux (src/metric.hpp:51-60). */
float bf_l2_metric(const float *lhs, const float *rhs) {
  float sum = 0.0f;
  for (int i = 0; i < BF_DIM; i++) {
    float temp = lhs[i] - rhs[i];
    sum += temp * temp;
  }
  return sqrtf(sum);
}

/* Same loop without the final square root; used by tests that need the
 * accumulator value itself. */
float bf_l2_metric_sq(const float *lhs, const float *rhs) {
  float sum = 0.0f;
  for (int i = 0; i < BF_DIM; i++) {
    float temp = lhs[i] - rhs[i];
    sum += temp * temp;
  }
  return sum;
}

/*
 * One direction of the matcher: the lambda at
 * src/feature_matcher/brute_force_matcher.cc:13-41.
 *
 * For every query row i the scan keeps (min, second) with strict '<'
 * (lines 26-33), then accepts when min < ratio * second in double
 * (line 35; the reference hard-codes ratio = 0.8).  Accepted matches are
 * appended in ascending i.  out_lhs / out_rhs / out_dist / out_second need
 * room for n_lhs entries; out_dist and out_second may be NULL.
 * Returns the number of matches written.
 */
int64_t bf_match_direction(const float *lhs, int64_t n_lhs, const float *rhs,
                           int64_t n_rhs, double ratio, int64_t *out_lhs,
                           int64_t *out_rhs, float *out_dist,
                           float *out_second) {
  int64_t count = 0;
  for (int64_t i = 0; i < n_lhs; i++) {
    size_t min_index = (size_t)-1;
    float min_distance = FLT_MAX;
    size_t second_min_index = (size_t)-1;
    float second_min_distance = FLT_MAX;
    const float *q = lhs + BF_DIM * i;
    for (int64_t j = 0; j < n_rhs; j++) {
      float distance = bf_l2_metric(q, rhs + BF_DIM * j);
      if (distance < second_min_distance) {
        second_min_distance = distance;
        second_min_index = (size_t)j;
        if (second_min_distance < min_distance) {
          size_t ti = second_min_index;
          second_min_index = min_index;
          min_index = ti;
          float td = second_min_distance;
          second_min_distance = min_distance;
          min_distance = td;
        }
      }
    }
    (void)second_min_index;
    if ((double)min_distance < ratio * (double)second_min_distance) {
      out_lhs[count] = i;
      out_rhs[count] = (int64_t)min_index;
      if (out_dist) out_dist[count] = min_distance;
      if (out_second) out_second[count] = second_min_distance;
      count++;
    }
  }
  return count;
}

/*
 * Cross-check of src/feature_matcher/brute_force_matcher.cc:64-80.
 *
 * The reference builds std::map<IndexT,IndexT> map_ from the reverse matches
 * ({index in second image -> index in first image}, insert keeps the first
 * occurrence; keys are unique here because each reverse query emits at most
 * one match) and keeps a forward match (i -> j) when map_[j] == i.
 * operator[] default-inserts 0 for a j with no surviving reverse match, so a
 * forward match with i == 0 survives in that case.  rev_of[j] is that map as
 * a dense array, -1 meaning "absent".
 */
int64_t bf_cross_check(const int64_t *fwd_lhs, const int64_t *fwd_rhs,
                       const float *fwd_dist, int64_t n_fwd,
                       const int64_t *rev_lhs, const int64_t *rev_rhs,
                       int64_t n_rev, int64_t n_second, int64_t *out_lhs,
                       int64_t *out_rhs, float *out_dist) {
  int64_t *rev_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_second > 0 ? n_second : 1));
  for (int64_t j = 0; j < n_second; j++) rev_of[j] = -1;
  for (int64_t m = 0; m < n_rev; m++) {
    if (rev_of[rev_lhs[m]] < 0) rev_of[rev_lhs[m]] = rev_rhs[m];
  }
  int64_t count = 0;
  for (int64_t m = 0; m < n_fwd; m++) {
    int64_t j = fwd_rhs[m];
    int64_t mapped = rev_of[j];
    if (mapped < 0) {
      mapped = 0;      /* map_[j] default-inserts 0 ... */
      rev_of[j] = 0;   /* ... and the entry stays for later look-ups */
    }
    if (mapped == fwd_lhs[m]) {
      out_lhs[count] = fwd_lhs[m];
      out_rhs[count] = j;
      if (out_dist) out_dist[count] = fwd_dist[m];
      count++;
    }
  }
  free(rev_of);
  return count;
}

/*
 * One image pair as BruteForceMatcher::Match handles it
 * (src/feature_matcher/brute_force_matcher.cc:46-86): forward direction
 * (first queries second), and with cross_check the swapped direction
 * followed by the cross-check above.  Outputs need room for n_first entries.
 */
int64_t bf_match_pair(const float *first, int64_t n_first, const float *second,
                      int64_t n_second, double ratio, int cross_check,
                      int64_t *out_lhs, int64_t *out_rhs, float *out_dist) {
  size_t cap_f = (size_t)(n_first > 0 ? n_first : 1);
  int64_t *f_l = (int64_t *)malloc(sizeof(int64_t) * cap_f);
  int64_t *f_r = (int64_t *)malloc(sizeof(int64_t) * cap_f);
  float *f_d = (float *)malloc(sizeof(float) * cap_f);
  int64_t nf = bf_match_direction(first, n_first, second, n_second, ratio, f_l, f_r, f_d, NULL);
  int64_t count;
  if (cross_check) {
    size_t cap_r = (size_t)(n_second > 0 ? n_second : 1);
    int64_t *r_l = (int64_t *)malloc(sizeof(int64_t) * cap_r);
    int64_t *r_r = (int64_t *)malloc(sizeof(int64_t) * cap_r);
    int64_t nr = bf_match_direction(second, n_second, first, n_first, ratio, r_l, r_r, NULL, NULL);
    count = bf_cross_check(f_l, f_r, f_d, nf, r_l, r_r, nr, n_second, out_lhs, out_rhs, out_dist);
    free(r_l);
    free(r_r);
  } else {
    count = nf;
    memcpy(out_lhs, f_l, sizeof(int64_t) * (size_t)nf);
    memcpy(out_rhs, f_r, sizeof(int64_t) * (size_t)nf);
    if (out_dist) memcpy(out_dist, f_d, sizeof(float) * (size_t)nf);
  }
  free(f_l);
  free(f_r);
  free(f_d);
  return count;
}

/* ------------------------------------------------------------------------
 * Threaded variant, for the timed CPU baseline only.
 *
 * The reference runs (pair, direction) tasks on a pool of two threads
 * (src/internal/thread_pool.hpp:11, brute_force_matcher.cc:44-60), so one
 * pair is at most two-way parallel.  With n_threads == 2 this mirrors that:
 * one thread per direction.  With more threads each direction's query rows
 * are split into contiguous slices; results are identical because queries
 * are independent and slices are concatenated in order.
 * ---------------------------------------------------------------------- */
typedef struct {
  const float *lhs;
  int64_t lo, hi;
  const float *rhs;
  int64_t n_rhs;
  double ratio;
  int64_t *o_l, *o_r;
  float *o_d;
  int64_t count;
} bf_slice;

static void *bf_slice_run(void *p) {
  bf_slice *s = (bf_slice *)p;
  s->count = bf_match_direction(s->lhs + BF_DIM * s->lo, s->hi - s->lo, s->rhs,
                                s->n_rhs, s->ratio, s->o_l, s->o_r, s->o_d, NULL);
  for (int64_t m = 0; m < s->count; m++) s->o_l[m] += s->lo;
  return NULL;
}

int64_t bf_match_pair_mt(const float *first, int64_t n_first, const float *second,
                         int64_t n_second, double ratio, int cross_check,
                         int n_threads, int64_t *out_lhs, int64_t *out_rhs,
                         float *out_dist) {
  if (n_threads < 1) n_threads = 1;
  int n_dirs = cross_check ? 2 : 1;
  int per_dir = n_threads / n_dirs;
  if (per_dir < 1) per_dir = 1;
  int n_slices = per_dir * n_dirs;
  bf_slice *sl = (bf_slice *)calloc((size_t)n_slices, sizeof(bf_slice));
  pthread_t *th = (pthread_t *)calloc((size_t)n_slices, sizeof(pthread_t));
  for (int d = 0; d < n_dirs; d++) {
    const float *q = d == 0 ? first : second;
    const float *t = d == 0 ? second : first;
    int64_t nq = d == 0 ? n_first : n_second;
    int64_t nt = d == 0 ? n_second : n_first;
    for (int k = 0; k < per_dir; k++) {
      bf_slice *s = &sl[d * per_dir + k];
      s->lhs = q;
      s->lo = nq * k / per_dir;
      s->hi = nq * (k + 1) / per_dir;
      s->rhs = t;
      s->n_rhs = nt;
      s->ratio = ratio;
      size_t cap = (size_t)(s->hi - s->lo > 0 ? s->hi - s->lo : 1);
      s->o_l = (int64_t *)malloc(sizeof(int64_t) * cap);
      s->o_r = (int64_t *)malloc(sizeof(int64_t) * cap);
      s->o_d = (float *)malloc(sizeof(float) * cap);
    }
  }
  for (int k = 0; k < n_slices; k++) pthread_create(&th[k], NULL, bf_slice_run, &sl[k]);
  for (int k = 0; k < n_slices; k++) pthread_join(th[k], NULL);

  size_t cap_f = (size_t)(n_first > 0 ? n_first : 1), cap_r = (size_t)(n_second > 0 ? n_second : 1);
  int64_t *f_l = (int64_t *)malloc(sizeof(int64_t) * cap_f), *f_r = (int64_t *)malloc(sizeof(int64_t) * cap_f);
  float *f_d = (float *)malloc(sizeof(float) * cap_f);
  int64_t nf = 0;
  for (int k = 0; k < per_dir; k++) {
    bf_slice *s = &sl[k];
    memcpy(f_l + nf, s->o_l, sizeof(int64_t) * (size_t)s->count);
    memcpy(f_r + nf, s->o_r, sizeof(int64_t) * (size_t)s->count);
    memcpy(f_d + nf, s->o_d, sizeof(float) * (size_t)s->count);
    nf += s->count;
  }
  int64_t count;
  if (cross_check) {
    int64_t *r_l = (int64_t *)malloc(sizeof(int64_t) * cap_r), *r_r = (int64_t *)malloc(sizeof(int64_t) * cap_r);
    int64_t nr = 0;
    for (int k = 0; k < per_dir; k++) {
      bf_slice *s = &sl[per_dir + k];
      memcpy(r_l + nr, s->o_l, sizeof(int64_t) * (size_t)s->count);
      memcpy(r_r + nr, s->o_r, sizeof(int64_t) * (size_t)s->count);
      nr += s->count;
    }
    count = bf_cross_check(f_l, f_r, f_d, nf, r_l, r_r, nr, n_second, out_lhs, out_rhs, out_dist);
    free(r_l);
    free(r_r);
  } else {
    count = nf;
    memcpy(out_lhs, f_l, sizeof(int64_t) * (size_t)nf);
    memcpy(out_rhs, f_r, sizeof(int64_t) * (size_t)nf);
    if (out_dist) memcpy(out_dist, f_d, sizeof(float) * (size_t)nf);
  }
  for (int k = 0; k < n_slices; k++) {
    free(sl[k].o_l);
    free(sl[k].o_r);
    free(sl[k].o_d);
  }
  free(f_l);
  free(f_r);
  free(f_d);
  free(sl);
  free(th);
  return count;
}

/* All pairs i<j over n_images ids, (min,max) in lexicographic order:
 * src/feature_matcher/exhaustive_pair_builder.cc:3-19 (the std::set orders
 * them).  ids must be ascending, as the std::map keys they come from are.
 * out needs room for n*(n-1) values (first,second interleaved). */
int64_t bf_exhaustive_pairs(const int64_t *ids, int64_t n_images, int64_t *out) {
  int64_t c = 0;
  for (int64_t i = 0; i < n_images; i++)
    for (int64_t j = i + 1; j < n_images; j++) {
      out[2 * c] = ids[i] < ids[j] ? ids[i] : ids[j];
      out[2 * c + 1] = ids[i] < ids[j] ? ids[j] : ids[i];
      c++;
    }
  return c;
}
