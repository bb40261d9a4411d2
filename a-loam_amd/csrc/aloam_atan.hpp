// a-loam_amd/csrc/aloam_atan.hpp — atanf / atan2f with the bits of glibc 2.35 (FDLIBM float algorithm), branch-free.
//
// `atan2` at reference src/scanRegistration.cpp:141-142,208 resolves to atan2f (using std::atan2, :56) and its result feeds
// `intensity = scanID + scanPeriod * relTime` (:238-239), whose integer part is the ring id downstream: the device libm is not
// bit-identical to glibc's, so the algorithm is restated here.  The range reduction of atanf picks one of four argument
// transformations; a wave sees all of them at once (azimuth sweeps the whole circle), so instead of four divergent branches with a
// division each, numerator and denominator are selected and divided ONCE — the same IEEE operations on the same values, hence the
// same bits.  Compiles for the host too (tests/host/test_atan_port.cpp checks it against glibc).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define ALOAM_HD __device__ __forceinline__
ALOAM_HD int aloam_f2i(float f) { return __float_as_int(f); }
ALOAM_HD float aloam_i2f(int i) { return __int_as_float(i); }
#else
#define ALOAM_HD static inline
ALOAM_HD int aloam_f2i(float f) { int i; memcpy(&i, &f, 4); return i; }
ALOAM_HD float aloam_i2f(int i) { float f; memcpy(&f, &i, 4); return f; }
#endif

namespace aloam {

ALOAM_HD float atanf_port(float x) {
  const int hx = aloam_f2i(x);
  const int ix = hx & 0x7fffffff;
  const float ax = fabsf(x);
  // range class: -1: |x| < 7/16, 0: < 11/16, 1: < 19/16, 2: < 39/16, 3: the rest
  const bool c0 = ix >= 0x3ee00000, c1 = ix >= 0x3f300000, c2 = ix >= 0x3f980000, c3 = ix >= 0x401c0000;
  // numerator / denominator of the reduced argument: (2x-1)/(2+x), (x-1)/(x+1), (x-1.5)/(1+1.5x), -1/x; x/1 below 7/16
  const float n0 = 2.0f * ax - 1.0f, d0 = 2.0f + ax;
  const float n1 = ax - 1.0f, d1 = ax + 1.0f;
  const float n2 = ax - 1.5f, d2 = 1.0f + 1.5f * ax;
  const float num = c3 ? -1.0f : (c2 ? n2 : (c1 ? n1 : (c0 ? n0 : x)));
  const float den = c3 ? ax : (c2 ? d2 : (c1 ? d1 : (c0 ? d0 : 1.0f)));
  const float xr = num / den;
  const float hi = c3 ? 1.5707962513e+00f : (c2 ? 9.8279368877e-01f : (c1 ? 7.8539812565e-01f : 4.6364760399e-01f));
  const float lo = c3 ? 7.5497894159e-08f : (c2 ? 3.4473217170e-08f : (c1 ? 3.7748947079e-08f : 5.0121582440e-09f));
  const float z = xr * xr;
  const float w = z * z;
  const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
  const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
  const float small = xr - xr * (s1 + s2);                                  // class -1 (xr == x)
  const float red = hi - ((xr * (s1 + s2) - lo) - xr);
  float r = c0 ? (hx < 0 ? -red : red) : small;
  if (ix < 0x31000000) r = x;                                               // |x| < 2^-29
  if (ix >= 0x4c000000) r = ix > 0x7f800000 ? x + x : (hx > 0 ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f);   // |x| >= 2^25, NaN
  return r;
}

ALOAM_HD float atan2f_port(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int hx = aloam_f2i(x), hy = aloam_f2i(y);
  const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return atanf_port(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    if (m < 2) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
    }
    switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = atanf_port(fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return aloam_i2f(aloam_f2i(z) ^ (int)0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

}  // namespace aloam
