// Test infrastructure only.  Wraps the reference's own L2 metric, compiled
// from where it lies in the reference checkout, behind a C symbol so that the
// tests can compare the oracle's restatement with it bit for bit.  The include
// path is supplied by oracle/Makefile; no reference text is copied here.
#include "metric.hpp"  // <reference>/src/metric.hpp:58-60 (L2_metric)

extern "C" float ref_L2_metric(const float* lhs, const float* rhs) {
  return L2_metric(lhs, rhs);
}
