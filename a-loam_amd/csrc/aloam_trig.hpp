// a-loam_amd/csrc/aloam_trig.hpp — acos / sin / cos in f64 as a fixed sequence of IEEE operations (FDLIBM's algorithms: e_acos.c,
// k_sin.c, k_cos.c, the small and medium ranges of e_rem_pio2.c), compiled for the device AND for the host.
//
// Only the DISTORTION 1 mode needs them (reference src/laserOdometry.cpp:59 ships 0): Eigen's slerp blends with
// sin((1 - s) theta) / sin(theta) and sin(s theta) / sin(theta), theta = acos(|w|) (src/laserOdometry.cpp:120, src/lidarFactor.hpp:29,81),
// and the de-skewed point is rounded to f32 before it is searched for.  The device libm and glibc agree to an ulp or two, not bit
// for bit; with every operation below being +, -, *, / or sqrt on doubles (correctly rounded on both sides, -ffp-contract=off on both
// sides), the device and the CPU restatement the tests check it against — which includes this very file — get the SAME bits for every
// scale, hence the same f32
// query points and the same correspondences by construction rather than by luck.  Accuracy is FDLIBM's (< 1 ulp), i.e. within an ulp
// of what the reference's glibc computes; tests/host/test_trig_port.cpp measures that.
// Valid for |x| <= 2^19 * pi / 2 (sin, cos; the interpolation ratio times an angle below pi / 2 never leaves that range) and
// |x| <= 1 (acos); outside they return NaN.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define ALOAM_TRIG_HD __device__ __forceinline__
ALOAM_TRIG_HD int64_t aloam_d2l(double d) { return __double_as_longlong(d); }
ALOAM_TRIG_HD double aloam_l2d(int64_t l) { return __longlong_as_double(l); }
#else
#define ALOAM_TRIG_HD static inline
ALOAM_TRIG_HD int64_t aloam_d2l(double d) { int64_t l; memcpy(&l, &d, 8); return l; }
ALOAM_TRIG_HD double aloam_l2d(int64_t l) { double d; memcpy(&d, &l, 8); return d; }
#endif

namespace aloam {

ALOAM_TRIG_HD int trig_hi(double x) { return (int)(aloam_d2l(x) >> 32); }                       // FDLIBM's __HI
ALOAM_TRIG_HD double trig_clear_lo(double x) { return aloam_l2d(aloam_d2l(x) & (int64_t)0xffffffff00000000ll); }   // __LO(x) = 0

// __kernel_sin(x, y, iy): sin(x + y) for |x| <= pi / 4, y the tail of x (iy = 0: y is 0)
ALOAM_TRIG_HD double trig_ksin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const int ix = trig_hi(x) & 0x7fffffff;
  if (ix < 0x3e400000) { if ((int)x == 0) return x; }                                          // |x| < 2^-27
  const double z = x * x, v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// __kernel_cos(x, y): cos(x + y) for |x| <= pi / 4
ALOAM_TRIG_HD double trig_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const int ix = trig_hi(x) & 0x7fffffff;
  if (ix < 0x3e400000) { if ((int)x == 0) return 1.0; }
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));                               // |x| < 0.3
  const double qx = ix > 0x3fe90000 ? 0.28125 : aloam_l2d((int64_t)(ix - 0x00200000) << 32);   // x / 4
  const double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

// __ieee754_rem_pio2 for |x| <= 2^19 * pi / 2: n = round(x / (pi / 2)), y0 + y1 = x - n pi / 2; returns n (sign of x), or 1 << 30 when
// x is out of range / not finite.  (The table FDLIBM uses to skip the cancellation check for the first 32 multiples is an optimisation
// only: the check itself gives the same result.)
ALOAM_TRIG_HD int trig_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  const int hx = trig_hi(x), ix = hx & 0x7fffffff;
  if (ix <= 0x3fe921fb) { *y0 = x; *y1 = 0.0; return 0; }                                       // |x| ~<= pi / 4
  if (ix < 0x4002d97c) {                                                                        // |x| < 3 pi / 4: n = +-1
    if (hx > 0) {
      double z = x - pio2_1;
      if (ix != 0x3ff921fb) { *y0 = z - pio2_1t; *y1 = (z - *y0) - pio2_1t; }
      else { z -= pio2_2; *y0 = z - pio2_2t; *y1 = (z - *y0) - pio2_2t; }                        // near pi / 2: one more 33 + 53 bits of pi
      return 1;
    }
    double z = x + pio2_1;
    if (ix != 0x3ff921fb) { *y0 = z + pio2_1t; *y1 = (z - *y0) + pio2_1t; }
    else { z += pio2_2; *y0 = z + pio2_2t; *y1 = (z - *y0) + pio2_2t; }
    return -1;
  }
  if (ix > 0x413921fb) { *y0 = *y1 = aloam_l2d((int64_t)0x7ff8000000000000ll); return 1 << 30; }
  const double t0 = fabs(x);
  const int n = (int)(t0 * invpio2 + 0.5);
  const double fn = (double)n;
  double r = t0 - fn * pio2_1, w = fn * pio2_1t;                                               // first round, good to 85 bits
  const int j = ix >> 20;
  double v0 = r - w;
  int i = j - ((trig_hi(v0) >> 20) & 0x7ff);
  if (i > 16) {                                                                                 // second iteration, good to 118 bits
    double t = r;
    w = fn * pio2_2; r = t - w; w = fn * pio2_2t - ((t - r) - w); v0 = r - w;
    i = j - ((trig_hi(v0) >> 20) & 0x7ff);
    if (i > 49) {                                                                               // third iteration, 151 bits
      t = r;
      w = fn * pio2_3; r = t - w; w = fn * pio2_3t - ((t - r) - w); v0 = r - w;
    }
  }
  const double v1 = (r - v0) - w;
  if (hx < 0) { *y0 = -v0; *y1 = -v1; return -n; }
  *y0 = v0; *y1 = v1;
  return n;
}

ALOAM_TRIG_HD double sin_port(double x) {
  double y0, y1;
  const int n = trig_rem_pio2(x, &y0, &y1);
  if (n == (1 << 30)) return y0;
  if (n == 0 && y1 == 0.0 && (trig_hi(x) & 0x7fffffff) <= 0x3fe921fb) return trig_ksin(x, 0.0, 0);
  switch (n & 3) {
    case 0: return trig_ksin(y0, y1, 1);
    case 1: return trig_kcos(y0, y1);
    case 2: return -trig_ksin(y0, y1, 1);
    default: return -trig_kcos(y0, y1);
  }
}

ALOAM_TRIG_HD double cos_port(double x) {
  double y0, y1;
  const int n = trig_rem_pio2(x, &y0, &y1);
  if (n == (1 << 30)) return y0;
  if (n == 0 && y1 == 0.0 && (trig_hi(x) & 0x7fffffff) <= 0x3fe921fb) return trig_kcos(x, 0.0);
  switch (n & 3) {
    case 0: return trig_kcos(y0, y1);
    case 1: return -trig_ksin(y0, y1, 1);
    case 2: return -trig_kcos(y0, y1);
    default: return trig_ksin(y0, y1, 1);
  }
}

// __ieee754_acos
ALOAM_TRIG_HD double acos_port(double x) {
  const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
               pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const int hx = trig_hi(x), ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) {                                                                       // |x| >= 1
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return aloam_l2d((int64_t)0x7ff8000000000000ll);
  }
  if (ix < 0x3fe00000) {                                                                        // |x| < 0.5
    if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
    const double z = x * x;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) {                                                                                 // x < -0.5
    const double z = (1.0 + x) * 0.5;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double s = sqrt(z), r = p / q, w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double z = (1.0 - x) * 0.5;                                                             // x > 0.5
  const double s = sqrt(z);
  const double df = trig_clear_lo(s);
  const double c = (z - df * df) / (s + df);
  const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const double r = p / q, w = r * s + c;
  return 2.0 * (df + w);
}

}  // namespace aloam
