// Test infrastructure only.  Wraps the reference's own L2 metric and Hamming
// distance, compiled from where they lie in the reference checkout, behind C
// symbols so that the tests can compare the oracles' restatements with them bit
// for bit.  The include path is supplied by oracle/Makefile; no reference text
// is copied here.
#include <bitset>
#include <cstdint>

#include "metric.hpp"  // <reference>/src/metric.hpp:58-60 (L2_metric), :61-68 (HammingDistance)

extern "C" float ref_L2_metric(const float* lhs, const float* rhs) {
  return L2_metric(lhs, rhs);
}

// codes: four 32-bit words each, hash bit j = bit j % 32 of word j / 32
extern "C" int ref_HammingDistance128(const uint32_t* lhs, const uint32_t* rhs) {
  std::bitset<128> a, b;
  for (int j = 0; j < 128; j++) {
    a[j] = (lhs[j / 32] >> (j % 32)) & 1u;
    b[j] = (rhs[j / 32] >> (j % 32)) & 1u;
  }
  return HammingDistance(a, b);
}
