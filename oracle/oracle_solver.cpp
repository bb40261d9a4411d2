// oracle/oracle_solver.cpp — CPU restatement of the residual functors and of ceres::Solve as the
// reference configures it.  TEST INFRASTRUCTURE ONLY (see aloam_oracle.h).
//   LidarEdgeFactor / LidarPlaneFactor       reference src/lidarFactor.hpp:12-55, 57-104
//   problem set-up, Huber(0.1), quaternion   reference src/laserOdometry.cpp:284-291,380-381,478-479
//   solver options (DENSE_QR, 4 iterations)  reference src/laserOdometry.cpp:494-499
// Ceres 1.12.0 (pinned by reference docker/Dockerfile:3) is not vendored and not installed; its
// trust-region Levenberg-Marquardt loop, HuberLoss, Corrector and EigenQuaternionParameterization
// are restated from their published behaviour.  Spec-to-verify: parity
// against a real Ceres build is UNPINNED.
#include <algorithm>
#include <cmath>
#include <limits>

#include "oracle_internal.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------
// Functors, templated on the scalar exactly like the reference's operator() so that the Jet
// instantiation reproduces what ceres::AutoDiffCostFunction differentiates.  s is the interpolation ratio the reference
// passes to the constructors: 1 with DISTORTION 0 (reference src/laserOdometry.cpp:59), the point's relative time otherwise.
// ---------------------------------------------------------------------------------------------
template <typename T>
static void edge_functor(const EdgeRec& e, const T* q, const T* t, T* residual) {
  const double s = e.s;
  const V3<T> cp{T(e.cp.x), T(e.cp.y), T(e.cp.z)};
  const V3<T> lpa{T(e.a.x), T(e.a.y), T(e.a.z)};
  const V3<T> lpb{T(e.b.x), T(e.b.y), T(e.b.z)};
  Quat<T> q_last_curr{q[0], q[1], q[2], q[3]};
  q_last_curr = slerp_from_identity(T(s), q_last_curr);
  const V3<T> t_last_curr{T(s) * t[0], T(s) * t[1], T(s) * t[2]};
  const V3<T> lp = rotate(q_last_curr, cp) + t_last_curr;
  const V3<T> nu = cross(lp - lpa, lp - lpb);
  const V3<T> de = lpa - lpb;
  residual[0] = nu.x / norm(de);
  residual[1] = nu.y / norm(de);
  residual[2] = nu.z / norm(de);
}

static V3d plane_normal(const PlaneRec& p) {       // LidarPlaneFactor ctor (lidarFactor.hpp:64-65)
  V3d n = cross(p.j - p.l, p.j - p.m);
  const double len = std::sqrt(dot(n, n));
  return V3d{n.x / len, n.y / len, n.z / len};
}

template <typename T>
static void plane_functor(const PlaneRec& p, const V3d& ljm_norm, const T* q, const T* t, T* residual) {
  const double s = p.s;
  const V3<T> cp{T(p.cp.x), T(p.cp.y), T(p.cp.z)};
  const V3<T> lpj{T(p.j.x), T(p.j.y), T(p.j.z)};
  const V3<T> ljm{T(ljm_norm.x), T(ljm_norm.y), T(ljm_norm.z)};
  Quat<T> q_last_curr{q[0], q[1], q[2], q[3]};
  q_last_curr = slerp_from_identity(T(s), q_last_curr);
  const V3<T> t_last_curr{T(s) * t[0], T(s) * t[1], T(s) * t[2]};
  const V3<T> lp = rotate(q_last_curr, cp) + t_last_curr;
  residual[0] = dot(lp - lpj, ljm);
}

// EigenQuaternionParameterization (Ceres): Plus(x, d) = (cos|d|, sin|d| d/|d|) * x ; 4x3 Jacobian at d = 0.
void quat_plus(const double q[4], const double delta[3], double out[4]) {
  const double nd = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (nd > 0.0) {
    const double k = std::sin(nd) / nd;
    const Quatd dq{k * delta[0], k * delta[1], k * delta[2], std::cos(nd)};
    const Quatd r = qmul(dq, Quatd{q[0], q[1], q[2], q[3]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
  } else {
    for (int k = 0; k < 4; ++k) out[k] = q[k];
  }
}
static void quat_plus_jacobian(const double x[4], double P[12]) {   // row-major 4x3, rows = x,y,z,w
  P[0] = x[3];  P[1] = x[2];   P[2] = -x[1];
  P[3] = -x[2]; P[4] = x[3];   P[5] = x[0];
  P[6] = x[1];  P[7] = -x[0];  P[8] = x[3];
  P[9] = -x[0]; P[10] = -x[1]; P[11] = -x[2];
}

// Closed form used by the HIP path: lp = R cp + t, d lp / d delta = -2 [R cp]x, d lp / d t = I.
static void closed_form_lp(const V3d& cp, const double q[4], const double t[3], V3d* lp, V3d* rcp) {
  *rcp = rotate(Quatd{q[0], q[1], q[2], q[3]}, cp);
  *lp = *rcp + V3d{t[0], t[1], t[2]};
}

void factor_eval_edge(const EdgeRec& e, const double q[4], const double t[3], bool analytic, double r[3], double J[18]) {
  if (!analytic || e.s != 1.0) {                      // the closed form below is the s = 1 case
    typedef Jet<7> J7;
    J7 jq[4], jt[3], jr[3];
    for (int k = 0; k < 4; ++k) jq[k] = J7(q[k], k);
    for (int k = 0; k < 3; ++k) jt[k] = J7(t[k], 4 + k);
    edge_functor<J7>(e, jq, jt, jr);
    double P[12];
    quat_plus_jacobian(q, P);
    for (int row = 0; row < 3; ++row) {
      r[row] = jr[row].a;
      for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
        for (int k = 0; k < 4; ++k) acc += jr[row].v[k] * P[k * 3 + c];
        J[row * 6 + c] = acc;
        J[row * 6 + 3 + c] = jr[row].v[4 + c];
      }
    }
    return;
  }
  V3d lp, rcp;
  closed_form_lp(e.cp, q, t, &lp, &rcp);
  const V3d de = e.a - e.b;
  const double inv = 1.0 / std::sqrt(dot(de, de));
  const V3d nu = cross(lp - e.a, lp - e.b);
  r[0] = nu.x * inv; r[1] = nu.y * inv; r[2] = nu.z * inv;
  // r = (lp x (a - b) ... ) affine in lp:  d r / d lp = [b - a]x / |a - b|
  const V3d w{(e.b.x - e.a.x) * inv, (e.b.y - e.a.y) * inv, (e.b.z - e.a.z) * inv};
  const double A[9] = {0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0};                 // [w]x
  const double B[9] = {0, 2 * rcp.z, -2 * rcp.y, -2 * rcp.z, 0, 2 * rcp.x, 2 * rcp.y, -2 * rcp.x, 0};  // -2 [R cp]x
  for (int row = 0; row < 3; ++row)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += A[row * 3 + k] * B[k * 3 + c];
      J[row * 6 + c] = acc;
      J[row * 6 + 3 + c] = A[row * 3 + c];
    }
}

void factor_eval_plane(const PlaneRec& p, const double q[4], const double t[3], bool analytic, double r[1], double J[6]) {
  const V3d n = plane_normal(p);
  if (!analytic || p.s != 1.0) {
    typedef Jet<7> J7;
    J7 jq[4], jt[3], jr[1];
    for (int k = 0; k < 4; ++k) jq[k] = J7(q[k], k);
    for (int k = 0; k < 3; ++k) jt[k] = J7(t[k], 4 + k);
    plane_functor<J7>(p, n, jq, jt, jr);
    double P[12];
    quat_plus_jacobian(q, P);
    r[0] = jr[0].a;
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += jr[0].v[k] * P[k * 3 + c];
      J[c] = acc;
      J[3 + c] = jr[0].v[4 + c];
    }
    return;
  }
  V3d lp, rcp;
  closed_form_lp(p.cp, q, t, &lp, &rcp);
  r[0] = dot(lp - p.j, n);
  // n^T (-2 [R cp]x) = 2 (R cp) x n, as a row vector
  J[0] = 2.0 * (n.z * rcp.y - n.y * rcp.z);
  J[1] = 2.0 * (n.x * rcp.z - n.z * rcp.x);
  J[2] = 2.0 * (n.y * rcp.x - n.x * rcp.y);
  J[3] = n.x; J[4] = n.y; J[5] = n.z;
}

// LidarPlaneNormFactor (lidarFactor.hpp:106-138): r = n . (q * cp + t) + d.  Functor instantiated on Jet<7> like
// ceres::AutoDiffCostFunction<LidarPlaneNormFactor, 1, 4, 3>, or the closed form the HIP path uses.
template <typename T>
static void norm_functor(const NormRec& p, const T* q, const T* t, T* residual) {
  const Quat<T> q_w_curr{q[0], q[1], q[2], q[3]};
  const V3<T> t_w_curr{t[0], t[1], t[2]};
  const V3<T> cp{T(p.cp.x), T(p.cp.y), T(p.cp.z)};
  const V3<T> point_w = rotate(q_w_curr, cp) + t_w_curr;
  const V3<T> norm{T(p.n.x), T(p.n.y), T(p.n.z)};
  residual[0] = dot(norm, point_w) + T(p.d);
}
void factor_eval_norm(const NormRec& p, const double q[4], const double t[3], bool analytic, double r[1], double J[6]) {
  if (!analytic) {
    typedef Jet<7> J7;
    J7 jq[4], jt[3], jr[1];
    for (int k = 0; k < 4; ++k) jq[k] = J7(q[k], k);
    for (int k = 0; k < 3; ++k) jt[k] = J7(t[k], 4 + k);
    norm_functor<J7>(p, jq, jt, jr);
    double P[12];
    quat_plus_jacobian(q, P);
    r[0] = jr[0].a;
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += jr[0].v[k] * P[k * 3 + c];
      J[c] = acc;
      J[3 + c] = jr[0].v[4 + c];
    }
    return;
  }
  V3d lp, rcp;
  closed_form_lp(p.cp, q, t, &lp, &rcp);
  r[0] = dot(p.n, lp) + p.d;
  J[0] = 2.0 * (p.n.z * rcp.y - p.n.y * rcp.z);
  J[1] = 2.0 * (p.n.x * rcp.z - p.n.z * rcp.x);
  J[2] = 2.0 * (p.n.y * rcp.x - p.n.x * rcp.y);
  J[3] = p.n.x; J[4] = p.n.y; J[5] = p.n.z;
}

// HuberLoss(a = 0.1) : rho, rho'  (Ceres loss_function.cc)
static inline void huber(double s, double* rho0, double* rho1) {
  const double a = 0.1, b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    *rho0 = 2.0 * a * r - b;
    *rho1 = std::max(std::numeric_limits<double>::min(), a / r);
  } else {
    *rho0 = s;
    *rho1 = 1.0;
  }
}

namespace {
struct Problem {
  const std::vector<EdgeRec>& edges;
  const std::vector<PlaneRec>& planes;
  bool analytic;
  const std::vector<NormRec>* norms = nullptr;     // LidarPlaneNormFactor blocks (mapping only)
  int rows() const { return 3 * (int)edges.size() + (int)planes.size() + (norms ? (int)norms->size() : 0); }
  // cost = 1/2 sum rho(|r_block|^2); residuals / Jacobian rows scaled by sqrt(rho') (Corrector with rho'' <= 0).
  double evaluate(const double q[4], const double t[3], std::vector<double>* res, std::vector<double>* jac) const {
    double cost = 0.0;
    int row = 0;
    if (res) res->assign(rows(), 0.0);
    if (jac) jac->assign((size_t)rows() * 6, 0.0);
    for (const EdgeRec& e : edges) {
      double r[3], J[18];
      factor_eval_edge(e, q, t, analytic, r, J);
      const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      double rho0, rho1;
      huber(s, &rho0, &rho1);
      cost += 0.5 * rho0;
      const double sc = std::sqrt(rho1);
      for (int k = 0; k < 3; ++k) {
        if (res) (*res)[row + k] = sc * r[k];
        if (jac) for (int c = 0; c < 6; ++c) (*jac)[(size_t)(row + k) * 6 + c] = sc * J[k * 6 + c];
      }
      row += 3;
    }
    for (const PlaneRec& p : planes) {
      double r[1], J[6];
      factor_eval_plane(p, q, t, analytic, r, J);
      const double s = r[0] * r[0];
      double rho0, rho1;
      huber(s, &rho0, &rho1);
      cost += 0.5 * rho0;
      const double sc = std::sqrt(rho1);
      if (res) (*res)[row] = sc * r[0];
      if (jac) for (int c = 0; c < 6; ++c) (*jac)[(size_t)row * 6 + c] = sc * J[c];
      row += 1;
    }
    if (norms) for (const NormRec& p : *norms) {
      double r[1], J[6];
      factor_eval_norm(p, q, t, analytic, r, J);
      double rho0, rho1;
      huber(r[0] * r[0], &rho0, &rho1);
      cost += 0.5 * rho0;
      const double sc = std::sqrt(rho1);
      if (res) (*res)[row] = sc * r[0];
      if (jac) for (int c = 0; c < 6; ++c) (*jac)[(size_t)row * 6 + c] = sc * J[c];
      row += 1;
    }
    return cost;
  }
};

// least squares  min |A y - b|  by Householder QR, A is m x 6 row-major (destroyed), b length m (destroyed).
bool qr_solve6(std::vector<double>& A, std::vector<double>& b, int m, double y[6]) {
  const int n = 6;
  for (int k = 0; k < n; ++k) {
    double nrm = 0.0;
    for (int i = k; i < m; ++i) nrm += A[(size_t)i * n + k] * A[(size_t)i * n + k];
    nrm = std::sqrt(nrm);
    if (nrm == 0.0) return false;
    const double alpha = A[(size_t)k * n + k] > 0 ? -nrm : nrm;
    // v = x - alpha e1
    std::vector<double> v(m - k);
    for (int i = k; i < m; ++i) v[i - k] = A[(size_t)i * n + k];
    v[0] -= alpha;
    double vnorm2 = 0.0;
    for (double x : v) vnorm2 += x * x;
    if (vnorm2 == 0.0) continue;
    for (int c = k; c < n; ++c) {
      double d = 0.0;
      for (int i = k; i < m; ++i) d += v[i - k] * A[(size_t)i * n + c];
      const double f = 2.0 * d / vnorm2;
      for (int i = k; i < m; ++i) A[(size_t)i * n + c] -= f * v[i - k];
    }
    double d = 0.0;
    for (int i = k; i < m; ++i) d += v[i - k] * b[i];
    const double f = 2.0 * d / vnorm2;
    for (int i = k; i < m; ++i) b[i] -= f * v[i - k];
  }
  for (int k = n - 1; k >= 0; --k) {
    double acc = b[k];
    for (int c = k + 1; c < n; ++c) acc -= A[(size_t)k * n + c] * y[c];
    const double diag = A[(size_t)k * n + k];
    if (diag == 0.0) return false;
    y[k] = acc / diag;
  }
  for (int k = 0; k < n; ++k) if (!std::isfinite(y[k])) return false;
  return true;
}
}  // namespace

double robust_cost(const std::vector<EdgeRec>& edges, const std::vector<PlaneRec>& planes, const double q[4], const double t[3]) {
  Problem pb{edges, planes, true};
  return pb.evaluate(q, t, nullptr, nullptr);
}

// ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT / DENSE_QR, jacobi_scaling on, monotonic steps,
// defaults otherwise, max_num_iterations as given.
LmSummary lm_solve(const std::vector<EdgeRec>& edges, const std::vector<PlaneRec>& planes, double q[4], double t[3],
                   int max_iterations, bool analytic, bool apply_converged_step, const std::vector<NormRec>* norms) {
  LmSummary sm;
  Problem pb{edges, planes, analytic, norms};
  const int m = pb.rows();
  if (m == 0) { sm.termination = 4; return sm; }

  const double kFunctionTol = 1e-6, kGradientTol = 1e-10, kParameterTol = 1e-8, kMinRelDecrease = 1e-3;
  const double kMinDiag = 1e-6, kMaxDiag = 1e32, kMaxRadius = 1e16, kMinRadius = 1e-32;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int n_invalid = 0;

  std::vector<double> r, J;
  double cost = pb.evaluate(q, t, &r, &J);
  sm.initial_cost = cost;
  const double q_entry[4] = {q[0], q[1], q[2], q[3]}, t_entry[3] = {t[0], t[1], t[2]};   // ceres::Solve puts these back after a FAILURE (solution not usable)
  // Ceres' ResidualBlock::Evaluate rejects non-finite residuals / Jacobians: "Residual and Jacobian evaluation failed" -> FAILURE
  // before the first iteration, parameters untouched (e.g. LidarEdgeFactor with a == b: 0 * inf).
  bool jac_finite = true;
  for (double v : J) jac_finite = jac_finite && std::isfinite(v);
  if (!std::isfinite(cost) || !jac_finite) { sm.termination = 5; sm.final_cost = cost; return sm; }
  // Ceres 1.12 trust_region_minimizer.cc: gradient_max_norm = |x - Plus(x, -g)|_inf, g = J^T r of the unscaled Jacobian in the
  // tangent space (SURVEY.md Appendix A "projected through Plus"); q, t are the current iterate when this is called.
  auto gradient_max = [&](const std::vector<double>& Ju, const std::vector<double>& ru) {
    double g[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < m; ++i) for (int c = 0; c < 6; ++c) g[c] += Ju[(size_t)i * 6 + c] * ru[i];
    const double ng[3] = {-g[0], -g[1], -g[2]};
    double qg[4];
    quat_plus(q, ng, qg);
    double mx = 0.0;
    for (int k = 0; k < 4; ++k) mx = std::max(mx, std::fabs(q[k] - qg[k]));
    for (int k = 0; k < 3; ++k) { const double tg = t[k] + (-g[3 + k]); mx = std::max(mx, std::fabs(t[k] - tg)); }   // literally |t - Plus(t, -g_t)|, quantised by ulp(t)
    return mx;
  };
  double gmax = gradient_max(J, r);
  double scale[6];
  for (int c = 0; c < 6; ++c) {                         // Jacobi scaling from the first Jacobian
    double s2 = 0.0;
    for (int i = 0; i < m; ++i) s2 += J[(size_t)i * 6 + c] * J[(size_t)i * 6 + c];
    scale[c] = 1.0 / (1.0 + std::sqrt(s2));
  }
  auto apply_scale = [&](std::vector<double>& Jm) { for (int i = 0; i < m; ++i) for (int c = 0; c < 6; ++c) Jm[(size_t)i * 6 + c] *= scale[c]; };
  apply_scale(J);
  double x_norm = 0.0;
  auto update_x_norm = [&]() { x_norm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]); };
  update_x_norm();
  double diag[6] = {0, 0, 0, 0, 0, 0};

  int iter = 0;
  while (true) {
    if (iter >= max_iterations) { sm.termination = 0; break; }
    log_decision(kDecLmGradTol, gmax, kGradientTol);
    if (gmax <= kGradientTol) { sm.termination = 3; break; }
    // Ceres 1.12 TrustRegionMinimizer::MinTrustRegionRadiusReached(): Radius() <= min_trust_region_radius ends the solve as CONVERGENCE
    // ("Minimum trust region radius reached"), not as a failure.  Unreachable from 1e4 in the reference's 4 iterations (and in practice at
    // all: a shrinking step meets the function tolerance first); kept literal.
    if (radius <= kMinRadius) { sm.termination = 6; break; }
    ++iter;
    sm.iterations = iter;
    // -- LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal) {
      for (int c = 0; c < 6; ++c) {
        double s2 = 0.0;
        for (int i = 0; i < m; ++i) s2 += J[(size_t)i * 6 + c] * J[(size_t)i * 6 + c];
        diag[c] = std::min(std::max(s2, kMinDiag), kMaxDiag);
      }
    }
    std::vector<double> A((size_t)(m + 6) * 6, 0.0), b(m + 6, 0.0);
    std::copy(J.begin(), J.end(), A.begin());
    std::copy(r.begin(), r.end(), b.begin());
    for (int c = 0; c < 6; ++c) A[(size_t)(m + c) * 6 + c] = std::sqrt(diag[c] / radius);
    double step[6];
    bool ok = qr_solve6(A, b, m + 6, step);
    for (int c = 0; c < 6; ++c) step[c] = -step[c];
    reuse_diagonal = true;
    double model_change = 0.0;
    if (ok) {
      for (int i = 0; i < m; ++i) {
        double mi = 0.0;
        for (int c = 0; c < 6; ++c) mi += J[(size_t)i * 6 + c] * step[c];
        model_change -= mi * (r[i] + mi / 2.0);
      }
    }
    if (ok) log_decision(kDecLmModel, model_change, 0.0);
    if (!ok || !(model_change > 0.0)) {               // invalid step
      if (++n_invalid >= 5) { sm.termination = 5; break; }
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
      continue;
    }
    n_invalid = 0;
    double delta[6];
    for (int c = 0; c < 6; ++c) delta[c] = step[c] * scale[c];
    double qc[4], tc[3];
    quat_plus(q, delta, qc);
    for (int k = 0; k < 3; ++k) tc[k] = t[k] + delta[3 + k];
    const double cost_c = pb.evaluate(qc, tc, nullptr, nullptr);
    // -- parameter tolerance (ambient-space step norm)
    double sn = 0.0;
    for (int k = 0; k < 4; ++k) sn += (q[k] - qc[k]) * (q[k] - qc[k]);
    for (int k = 0; k < 3; ++k) sn += (t[k] - tc[k]) * (t[k] - tc[k]);
    sn = std::sqrt(sn);
    const bool converged_param = sn <= kParameterTol * (x_norm + kParameterTol);
    const bool converged_func = std::fabs(cost - cost_c) <= kFunctionTol * cost;
    log_decision(kDecLmParamTol, sn, kParameterTol * (x_norm + kParameterTol));
    if (!converged_param) log_decision(kDecLmFuncTol, std::fabs(cost - cost_c), kFunctionTol * cost);
    if (converged_param || converged_func) {
      if (apply_converged_step && cost_c < cost) { for (int k = 0; k < 4; ++k) q[k] = qc[k]; for (int k = 0; k < 3; ++k) t[k] = tc[k]; cost = cost_c; }
      sm.termination = converged_param ? 1 : 2;
      break;
    }
    const double rel = (cost - cost_c) / model_change;
    log_decision(kDecLmAccept, rel, kMinRelDecrease);
    if (rel > kMinRelDecrease) {                      // successful step
      for (int k = 0; k < 4; ++k) q[k] = qc[k];
      for (int k = 0; k < 3; ++k) t[k] = tc[k];
      update_x_norm();
      cost = pb.evaluate(q, t, &r, &J);
      // HandleSuccessfulStep(): x is the candidate now; if residuals / Jacobian cannot be evaluated there the solve ends as FAILURE
      bool jf = std::isfinite(cost);
      for (double v : J) jf = jf && std::isfinite(v);
      if (!jf) { sm.termination = 5; break; }
      gmax = gradient_max(J, r);
      apply_scale(J);
      sm.successful++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::min(kMaxRadius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {                                          // rejected step
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  if (sm.termination == 5) {                            // FAILURE (invalid steps / Jacobian at an accepted point): parameters as at entry, Solver::Summary::IsSolutionUsable()
    for (int k = 0; k < 4; ++k) q[k] = q_entry[k];
    for (int k = 0; k < 3; ++k) t[k] = t_entry[k];
    cost = sm.initial_cost;
  }
  sm.final_cost = cost;
  return sm;
}

}  // namespace orc
