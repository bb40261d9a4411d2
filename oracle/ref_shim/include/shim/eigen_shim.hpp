// oracle/ref_shim/include/shim/eigen_shim.hpp — stand-in for the slice of Eigen 3 the A-LOAM sources touch.
//
// TEST INFRASTRUCTURE ONLY (see shim/ros_shim.hpp).  Eigen is an un-vendored third-party dependency; the few operations
// the reference uses are restated here with Eigen 3.3's evaluation order where it can matter:
//   Quaternion * Vector3   uv = 2 (u x v);  v + w * uv + u x uv               (QuaternionBase::_transformVector)
//   Quaternion * Quaternion Hamilton product                                  (internal::quat_product)
//   Quaternion::slerp      acos / sin blend with the (1 - epsilon) guard      (QuaternionBase::slerp)
//   inverse()              conjugate / squaredNorm
//   SelfAdjointEigenSolver<Matrix3d>  (here: cyclic Jacobi, ascending eigenvalues)
//   colPivHouseholderQr().solve()     (here: Householder QR with column pivoting, least squares)
// Scalar-generic so that the functors of src/lidarFactor.hpp instantiate on ceres::Jet exactly as they do upstream.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>
#include <utility>

namespace Eigen {

namespace shim {
template <class T> struct Eps { static T value() { return T(std::numeric_limits<double>::epsilon()); } };
}

template <class T, int R, int C>
struct Matrix {
  T m[R * C];   // column-major like Eigen
  Matrix() { for (int i = 0; i < R * C; ++i) m[i] = T(0); }
  Matrix(const T& a, const T& b, const T& c) { static_assert(R * C == 3, "3-vector ctor"); m[0] = a; m[1] = b; m[2] = c; }
  static Matrix Zero() { return Matrix(); }
  static Matrix Ones() { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = T(1); return r; }
  T& operator()(int i) { return m[i]; }
  const T& operator()(int i) const { return m[i]; }
  T& operator[](int i) { return m[i]; }
  const T& operator[](int i) const { return m[i]; }
  T& operator()(int i, int j) { return m[i + j * R]; }
  const T& operator()(int i, int j) const { return m[i + j * R]; }
  T& x() { return m[0]; } T& y() { return m[1]; } T& z() { return m[2]; }
  const T& x() const { return m[0]; } const T& y() const { return m[1]; } const T& z() const { return m[2]; }
  friend Matrix operator+(const Matrix& a, const Matrix& b) { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
  friend Matrix operator-(const Matrix& a, const Matrix& b) { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
  friend Matrix operator-(const Matrix& a) { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = -a.m[i]; return r; }
  friend Matrix operator*(const T& s, const Matrix& a) { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = s * a.m[i]; return r; }
  friend Matrix operator*(const Matrix& a, const T& s) { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = a.m[i] * s; return r; }
  friend Matrix operator/(const Matrix& a, const T& s) { Matrix r; for (int i = 0; i < R * C; ++i) r.m[i] = a.m[i] / s; return r; }
  Matrix<T, C, R> transpose() const { Matrix<T, C, R> r; for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) r(j, i) = (*this)(i, j); return r; }
  Matrix<T, R, 1> col(int j) const { Matrix<T, R, 1> r; for (int i = 0; i < R; ++i) r.m[i] = (*this)(i, j); return r; }
  Matrix cross(const Matrix& b) const {
    static_assert(R * C == 3, "cross of 3-vectors");
    return Matrix(m[1] * b.m[2] - m[2] * b.m[1], m[2] * b.m[0] - m[0] * b.m[2], m[0] * b.m[1] - m[1] * b.m[0]);
  }
  T dot(const Matrix& b) const { T s = m[0] * b.m[0]; for (int i = 1; i < R * C; ++i) s = s + m[i] * b.m[i]; return s; }
  T squaredNorm() const { return dot(*this); }
  T norm() const { using std::sqrt; return sqrt(squaredNorm()); }
  void normalize() { const T n = norm(); for (int i = 0; i < R * C; ++i) m[i] = m[i] / n; }
  Matrix normalized() const { Matrix r = *this; r.normalize(); return r; }
  struct ColPivQR;
  ColPivQR colPivHouseholderQr() const;
};
template <class T, int R, int K, int C>
Matrix<T, R, C> operator*(const Matrix<T, R, K>& a, const Matrix<T, K, C>& b) {
  Matrix<T, R, C> r;
  for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) { T s = a(i, 0) * b(0, j); for (int k = 1; k < K; ++k) s = s + a(i, k) * b(k, j); r(i, j) = s; }
  return r;
}
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 3, 3> Matrix3d;

// Least squares through Householder QR with column pivoting (what colPivHouseholderQr().solve() computes for a
// full-column-rank tall system; used on the 5x3 plane fit at reference src/laserMapping.cpp:663).
template <class T, int R, int C>
struct Matrix<T, R, C>::ColPivQR {
  Matrix A;
  Matrix<T, C, 1> solve(const Matrix<T, R, 1>& b0) const {
    double a[R][C], b[R];
    int perm[C];
    for (int i = 0; i < R; ++i) { b[i] = b0(i); for (int j = 0; j < C; ++j) a[i][j] = A(i, j); }
    for (int j = 0; j < C; ++j) perm[j] = j;
    int rank = 0;
    double maxnorm0 = 0.0;
    for (int k = 0; k < C; ++k) {
      int piv = k; double best = -1.0;
      for (int j = k; j < C; ++j) { double s = 0.0; for (int i = k; i < R; ++i) s += a[i][j] * a[i][j]; if (s > best) { best = s; piv = j; } }
      if (k == 0) maxnorm0 = best;
      if (!(best > maxnorm0 * 1e-30)) break;
      if (piv != k) { for (int i = 0; i < R; ++i) std::swap(a[i][k], a[i][piv]); std::swap(perm[k], perm[piv]); }
      const double alpha = (a[k][k] > 0.0 ? -1.0 : 1.0) * std::sqrt(best);
      double v[R];
      for (int i = 0; i < R; ++i) v[i] = i < k ? 0.0 : a[i][k];
      v[k] -= alpha;
      double vv = 0.0; for (int i = k; i < R; ++i) vv += v[i] * v[i];
      if (vv > 0.0) {
        for (int j = k; j < C; ++j) { double s = 0.0; for (int i = k; i < R; ++i) s += v[i] * a[i][j]; s = 2.0 * s / vv; for (int i = k; i < R; ++i) a[i][j] -= s * v[i]; }
        double s = 0.0; for (int i = k; i < R; ++i) s += v[i] * b[i]; s = 2.0 * s / vv; for (int i = k; i < R; ++i) b[i] -= s * v[i];
      }
      ++rank;
    }
    double y[C];
    for (int k = 0; k < C; ++k) y[k] = 0.0;
    for (int k = rank - 1; k >= 0; --k) { double s = b[k]; for (int j = k + 1; j < rank; ++j) s -= a[k][j] * y[j]; y[k] = s / a[k][k]; }
    Matrix<T, C, 1> x;
    for (int k = 0; k < C; ++k) x(perm[k]) = y[k];
    return x;
  }
};
template <class T, int R, int C>
typename Matrix<T, R, C>::ColPivQR Matrix<T, R, C>::colPivHouseholderQr() const { ColPivQR q; q.A = *this; return q; }

template <class T>
struct Quaternion {
  T c[4];   // x, y, z, w  (Eigen's coeffs() order)
  Quaternion() { c[0] = c[1] = c[2] = T(0); c[3] = T(1); }
  Quaternion(const T& w, const T& x, const T& y, const T& z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
  static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
  T& x() { return c[0]; } T& y() { return c[1]; } T& z() { return c[2]; } T& w() { return c[3]; }
  const T& x() const { return c[0]; } const T& y() const { return c[1]; } const T& z() const { return c[2]; } const T& w() const { return c[3]; }
  Quaternion operator*(const Quaternion& b) const {
    const Quaternion& a = *this;
    return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                      a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                      a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                      a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1>& v) const {
    const Matrix<T, 3, 1> u(c[0], c[1], c[2]);
    Matrix<T, 3, 1> uv = u.cross(v);
    uv = uv + uv;
    return v + c[3] * uv + u.cross(uv);
  }
  T dot(const Quaternion& o) const { return c[0] * o.c[0] + c[1] * o.c[1] + c[2] * o.c[2] + c[3] * o.c[3]; }
  T squaredNorm() const { return dot(*this); }
  Quaternion conjugate() const { return Quaternion(c[3], -c[0], -c[1], -c[2]); }
  Quaternion inverse() const {
    const T n2 = squaredNorm();
    if (n2 > T(0)) return Quaternion(c[3] / n2, -c[0] / n2, -c[1] / n2, -c[2] / n2);
    return Quaternion(T(0), T(0), T(0), T(0));
  }
  void normalize() { using std::sqrt; const T n = sqrt(squaredNorm()); for (int k = 0; k < 4; ++k) c[k] = c[k] / n; }
  Quaternion slerp(const T& t, const Quaternion& other) const {
    using std::abs; using std::acos; using std::sin;
    const T one = T(1) - shim::Eps<T>::value();
    const T d = this->dot(other);
    const T absD = abs(d);
    T scale0, scale1;
    if (absD >= one) {
      scale0 = T(1) - t;
      scale1 = t;
    } else {
      const T theta = acos(absD);
      const T sinTheta = sin(theta);
      scale0 = sin((T(1) - t) * theta) / sinTheta;
      scale1 = sin((t * theta)) / sinTheta;
    }
    if (d < T(0)) scale1 = -scale1;
    Quaternion r;
    for (int k = 0; k < 4; ++k) r.c[k] = scale0 * c[k] + scale1 * other.c[k];
    return r;
  }
};
typedef Quaternion<double> Quaterniond;

// Map<> over caller-owned storage, only for the two types the reference maps (Quaterniond over para_q, Vector3d over para_t).
template <class X> class Map;
template <>
class Map<Quaterniond> {
 public:
  explicit Map(double* p) : p_(p) {}
  operator Quaterniond() const { return Quaterniond(p_[3], p_[0], p_[1], p_[2]); }
  Map& operator=(const Quaterniond& q) { p_[0] = q.x(); p_[1] = q.y(); p_[2] = q.z(); p_[3] = q.w(); return *this; }
  double& x() { return p_[0]; } double& y() { return p_[1]; } double& z() { return p_[2]; } double& w() { return p_[3]; }
  double x() const { return p_[0]; } double y() const { return p_[1]; } double z() const { return p_[2]; } double w() const { return p_[3]; }
  Quaterniond inverse() const { return Quaterniond(*this).inverse(); }
  Quaterniond operator*(const Quaterniond& b) const { return Quaterniond(*this) * b; }
  Vector3d operator*(const Vector3d& v) const { return Quaterniond(*this) * v; }
 private:
  double* p_;
};
template <>
class Map<Vector3d> {
 public:
  explicit Map(double* p) : p_(p) {}
  operator Vector3d() const { return Vector3d(p_[0], p_[1], p_[2]); }
  Map& operator=(const Vector3d& v) { p_[0] = v.x(); p_[1] = v.y(); p_[2] = v.z(); return *this; }
  double& x() { return p_[0]; } double& y() { return p_[1]; } double& z() { return p_[2]; }
  double x() const { return p_[0]; } double y() const { return p_[1]; } double z() const { return p_[2]; }
  double& operator()(int i) { return p_[i]; }
  double operator()(int i) const { return p_[i]; }
 private:
  double* p_;
};

// Symmetric 3x3 eigen-decomposition, ascending eigenvalues, orthonormal eigenvectors in columns (cyclic Jacobi).
template <class M> class SelfAdjointEigenSolver;
template <>
class SelfAdjointEigenSolver<Matrix3d> {
 public:
  explicit SelfAdjointEigenSolver(const Matrix3d& A0) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = 0.5 * (A0(i, j) + A0(j, i));
    for (int sweep = 0; sweep < 50; ++sweep) {
      const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
      if (off <= 1e-22 * (std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]))) break;
      for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk; }
        a[p][q] = 0.0; a[q][p] = 0.0;
        for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = cs * vkp - sn * vkq; v[k][q] = sn * vkp + cs * vkq; }
      }
    }
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (a[order[j]][order[j]] < a[order[i]][order[i]]) std::swap(order[i], order[j]);
    for (int k = 0; k < 3; ++k) { vals_(k) = a[order[k]][order[k]]; for (int i = 0; i < 3; ++i) vecs_(i, k) = v[i][order[k]]; }
  }
  const Vector3d& eigenvalues() const { return vals_; }
  const Matrix3d& eigenvectors() const { return vecs_; }
 private:
  Vector3d vals_;
  Matrix3d vecs_;
};

}  // namespace Eigen
