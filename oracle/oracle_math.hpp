// oracle/oracle_math.hpp — tiny f64 vector / quaternion / dual-number helpers for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped MI355X path; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call it.
//
// The reference leans on Eigen (not installed here) for these few operations; this header restates
// the *arithmetic order* Eigen 3.x uses where it can influence a decision quantity:
//   - Quaternion * Vector3  : uv = 2 (u x v);  r = v + w*uv + u x uv       (used at
//                             reference src/laserOdometry.cpp:123, src/lidarFactor.hpp:33,85,120)
//   - Quaternion::slerp     : acos/sin blend with the (1 - eps) guard      (src/laserOdometry.cpp:120,
//                             src/lidarFactor.hpp:29,81)
//   - Quaternion * Quaternion (Hamilton product, xyzw storage)             (src/laserOdometry.cpp:505)
// Templates are generic in the scalar so the same code runs on double and on Jet<N> (forward-mode
// dual numbers standing in for ceres::Jet, src/lidarFactor.hpp:48-50,96-98,130-132).
#pragma once
#include <cmath>
#include <limits>
// acos / sin / cos of the slerp (DISTORTION 1 only): the fixed IEEE operation sequence the device evaluates too, so that both sides get
// the same bits (FDLIBM algorithms, within 1 ulp of glibc: tests/host/test_trig_port.cpp); the reference build under oracle/_ref keeps glibc.
#include "../a-loam_amd/csrc/aloam_trig.hpp"

namespace orc {

// ----------------------------------------------------------------------------------------------
// Jet<N>: value + N partials (what ceres::AutoDiffCostFunction evaluates the functors on).
// ----------------------------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int k = 0; k < N; ++k) v[k] = 0.0; }
  Jet(double s) : a(s) { for (int k = 0; k < N; ++k) v[k] = 0.0; }  // NOLINT (implicit on purpose)
  Jet(double s, int idx) : a(s) { for (int k = 0; k < N; ++k) v[k] = 0.0; v[idx] = 1.0; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a + y.a; for (int k = 0; k < N; ++k) r.v[k] = x.v[k] + y.v[k]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a - y.a; for (int k = 0; k < N; ++k) r.v[k] = x.v[k] - y.v[k]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x) { Jet<N> r; r.a = -x.a; for (int k = 0; k < N; ++k) r.v[k] = -x.v[k]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a * y.a; for (int k = 0; k < N; ++k) r.v[k] = x.a * y.v[k] + x.v[k] * y.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int k = 0; k < N; ++k) r.v[k] = (x.v[k] - r.a * y.v[k]) * inv;
  return r;
}
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline bool operator<(const Jet<N>& x, const Jet<N>& y) { return x.a < y.a; }
template <int N> inline bool operator>=(const Jet<N>& x, const Jet<N>& y) { return x.a >= y.a; }
template <int N> inline Jet<N> jsqrt(const Jet<N>& x) { Jet<N> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a; for (int k = 0; k < N; ++k) r.v[k] = x.v[k] * d; return r; }
template <int N> inline Jet<N> jsin(const Jet<N>& x) { Jet<N> r; r.a = aloam::sin_port(x.a); const double d = aloam::cos_port(x.a); for (int k = 0; k < N; ++k) r.v[k] = x.v[k] * d; return r; }
template <int N> inline Jet<N> jacos(const Jet<N>& x) { Jet<N> r; r.a = aloam::acos_port(x.a); const double d = -1.0 / std::sqrt(1.0 - x.a * x.a); for (int k = 0; k < N; ++k) r.v[k] = x.v[k] * d; return r; }
template <int N> inline Jet<N> jabs(const Jet<N>& x) { return x.a < 0.0 ? -x : x; }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jsin(double x) { return aloam::sin_port(x); }
inline double jacos(double x) { return aloam::acos_port(x); }
inline double jabs(double x) { return std::fabs(x); }
inline double jval(double x) { return x; }
template <int N> inline double jval(const Jet<N>& x) { return x.a; }

// ----------------------------------------------------------------------------------------------
// 3-vectors and quaternions (xyzw storage like Eigen's coeffs()).
// ----------------------------------------------------------------------------------------------
template <typename T> struct V3 { T x, y, z; };
template <typename T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline V3<T> operator*(const T& s, const V3<T>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> inline T norm(const V3<T>& a) { return jsqrt(dot(a, a)); }

template <typename T> struct Quat { T x, y, z, w; };

// Eigen::Quaternion::_transformVector: v + w*(2 u x v) + u x (2 u x v).
template <typename T> inline V3<T> rotate(const Quat<T>& q, const V3<T>& v) {
  const V3<T> u{q.x, q.y, q.z};
  V3<T> uv = cross(u, v);
  uv = uv + uv;
  return v + q.w * uv + cross(u, uv);
}

// Hamilton product a*b.
template <typename T> inline Quat<T> qmul(const Quat<T>& a, const Quat<T>& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

// Identity.slerp(t, other) as Eigen 3.x evaluates it (QuaternionBase::slerp).
template <typename T> inline Quat<T> slerp_from_identity(const T& t, const Quat<T>& other) {
  const T one = T(1.0 - std::numeric_limits<double>::epsilon());
  const T d = T(0.0) * other.x + T(0.0) * other.y + T(0.0) * other.z + T(1.0) * other.w;
  const T absD = jabs(d);
  T scale0, scale1;
  if (absD >= one) {
    scale0 = T(1.0) - t;
    scale1 = t;
  } else {
    const T theta = jacos(absD);
    const T sinTheta = jsin(theta);
    scale0 = jsin((T(1.0) - t) * theta) / sinTheta;
    scale1 = jsin(t * theta) / sinTheta;
  }
  if (d < T(0.0)) scale1 = -scale1;
  // identity coeffs (0,0,0,1)
  return {scale0 * T(0.0) + scale1 * other.x, scale0 * T(0.0) + scale1 * other.y,
          scale0 * T(0.0) + scale1 * other.z, scale0 * T(1.0) + scale1 * other.w};
}

using V3d = V3<double>;
using Quatd = Quat<double>;

}  // namespace orc
