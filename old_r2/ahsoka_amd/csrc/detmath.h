// detmath.h -- arithmetic that must come out bit-identical on the host CPU and
// on gfx950: only IEEE-754 basic operations (+,-,*,/ are correctly rounded on
// both; the build uses -ffp-contract=off so no FMA is formed) in a fixed order.
//
// Used by stage S4 (read-pair scoring; the reference calls
// ReadScoring::scoreReadsetLocal at src/alignmentstoreadset.cpp:308-311, whose
// log() comes from libm -- libm and the ROCm device library do not round the
// same way, hence this explicit series).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AHK_HD __host__ __device__ __forceinline__
#else
#define AHK_HD inline
#endif

namespace ahk {

AHK_HD double bits_to_double(uint64_t b) {
  union { uint64_t u; double d; } c;
  c.u = b;
  return c.d;
}
AHK_HD uint64_t double_to_bits(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}

// natural logarithm of a positive normal double: log(x) = e*ln2 + 2*atanh(s),
// s = (m-1)/(m+1), m in (sqrt(1/2), sqrt(2)]; atanh by its odd series to s^23.
AHK_HD double det_log(double x) {
  uint64_t bits = double_to_bits(x);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double m = bits_to_double((bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e += 1;
  }
  double s = (m - 1.0) / (m + 1.0);
  double s2 = s * s;
  double p = 1.0 / 23.0;
  p = p * s2 + 1.0 / 21.0;
  p = p * s2 + 1.0 / 19.0;
  p = p * s2 + 1.0 / 17.0;
  p = p * s2 + 1.0 / 15.0;
  p = p * s2 + 1.0 / 13.0;
  p = p * s2 + 1.0 / 11.0;
  p = p * s2 + 1.0 / 9.0;
  p = p * s2 + 1.0 / 7.0;
  p = p * s2 + 1.0 / 5.0;
  p = p * s2 + 1.0 / 3.0;
  p = p * s2 + 1.0;
  double r = (2.0 * s) * p;
  return r + (double)e * 0.6931471805599453;
}

// floor for |x| < 2^51 without libm
AHK_HD double det_floor(double x) {
  long long i = (long long)x;  // truncates toward zero
  double t = (double)i;
  return (t > x) ? t - 1.0 : t;
}

// similarity score -> float (TriangleSparseMatrix stores float) -> Q15.16
// fixed point, clamped to +-2^30.  0 means "no entry".
AHK_HD int32_t quantise_score(double score) {
  float f = (float)score;
  double x = (double)f * 65536.0;
  double r = det_floor(x + 0.5);
  if (r > 1073741824.0) r = 1073741824.0;
  if (r < -1073741824.0) r = -1073741824.0;
  return (int32_t)r;
}

}  // namespace ahk
