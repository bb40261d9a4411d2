// oracle/ref_shim/include/shim/ceres_shim.hpp — stand-in for the slice of Ceres Solver the A-LOAM sources touch.
//
// TEST INFRASTRUCTURE ONLY (see shim/ros_shim.hpp).  Ceres (pinned to 1.12.0 only by reference docker/Dockerfile:3) is
// an un-vendored third-party dependency that is not installed here, so its behaviour is restated from its published
// implementation (SURVEY.md Appendix A lists the semantics and marks them spec-to-verify):
//   Jet<T, N>, AutoDiffCostFunction         forward-mode duals, one pass over all parameter blocks
//   HuberLoss, Corrector                    rho(s) and the sqrt(rho') row scaling (rho'' <= 0 => no curvature term)
//   EigenQuaternionParameterization         Plus(x, d) = (cos|d|, sin|d| d/|d|) * x, 4x3 Jacobian at d = 0
//   Solve()                                 trust-region Levenberg-Marquardt with Jacobi scaling and DENSE_QR
//                                           (Householder QR of the stacked [J; sqrt(D)] system), Ceres >= 1.12 loop order
// This implementation is written independently of oracle/oracle_solver.cpp so that the two can be compared.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <typeinfo>
#include <limits>
#include <memory>
#include <string>
#include <vector>

namespace ceres {

template <class T, int N>
struct Jet {
  T a;
  T v[N];
  Jet() : a(T(0)) { for (int k = 0; k < N; ++k) v[k] = T(0); }
  Jet(const T& s) : a(s) { for (int k = 0; k < N; ++k) v[k] = T(0); }   // NOLINT: implicit like ceres::Jet
  Jet(int s) : a(T(s)) { for (int k = 0; k < N; ++k) v[k] = T(0); }     // NOLINT
  Jet(const T& s, int idx) : a(s) { for (int k = 0; k < N; ++k) v[k] = T(0); v[idx] = T(1); }
};
#define CERES_SHIM_J template <class T, int N> inline
CERES_SHIM_J Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a + g.a; for (int k = 0; k < N; ++k) r.v[k] = f.v[k] + g.v[k]; return r; }
CERES_SHIM_J Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a - g.a; for (int k = 0; k < N; ++k) r.v[k] = f.v[k] - g.v[k]; return r; }
CERES_SHIM_J Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> r; r.a = -f.a; for (int k = 0; k < N; ++k) r.v[k] = -f.v[k]; return r; }
CERES_SHIM_J Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a * g.a; for (int k = 0; k < N; ++k) r.v[k] = f.a * g.v[k] + f.v[k] * g.a; return r; }
CERES_SHIM_J Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
  Jet<T, N> r; const T inv = T(1) / g.a; const T q = f.a * inv; r.a = q;
  for (int k = 0; k < N; ++k) r.v[k] = (f.v[k] - q * g.v[k]) * inv;
  return r;
}
CERES_SHIM_J Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a = f.a + s; return r; }
CERES_SHIM_J Jet<T, N> operator+(T s, const Jet<T, N>& f) { Jet<T, N> r = f; r.a = s + f.a; return r; }
CERES_SHIM_J Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a = f.a - s; return r; }
CERES_SHIM_J Jet<T, N> operator-(T s, const Jet<T, N>& f) { Jet<T, N> r = -f; r.a = s - f.a; return r; }
CERES_SHIM_J Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> r; r.a = f.a * s; for (int k = 0; k < N; ++k) r.v[k] = f.v[k] * s; return r; }
CERES_SHIM_J Jet<T, N> operator*(T s, const Jet<T, N>& f) { Jet<T, N> r; r.a = s * f.a; for (int k = 0; k < N; ++k) r.v[k] = s * f.v[k]; return r; }
CERES_SHIM_J Jet<T, N> operator/(const Jet<T, N>& f, T s) { Jet<T, N> r; const T inv = T(1) / s; r.a = f.a * inv; for (int k = 0; k < N; ++k) r.v[k] = f.v[k] * inv; return r; }
CERES_SHIM_J bool operator<(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a; }
CERES_SHIM_J bool operator>(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a > g.a; }
CERES_SHIM_J bool operator<=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a <= g.a; }
CERES_SHIM_J bool operator>=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a >= g.a; }
CERES_SHIM_J Jet<T, N> abs(const Jet<T, N>& f) { return f.a < T(0) ? -f : f; }
CERES_SHIM_J Jet<T, N> sqrt(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::sqrt(f.a); const T d = T(1) / (T(2) * r.a); for (int k = 0; k < N; ++k) r.v[k] = f.v[k] * d; return r; }
CERES_SHIM_J Jet<T, N> sin(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::sin(f.a); const T d = std::cos(f.a); for (int k = 0; k < N; ++k) r.v[k] = f.v[k] * d; return r; }
CERES_SHIM_J Jet<T, N> cos(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::cos(f.a); const T d = -std::sin(f.a); for (int k = 0; k < N; ++k) r.v[k] = f.v[k] * d; return r; }
CERES_SHIM_J Jet<T, N> acos(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::acos(f.a); const T d = -T(1) / std::sqrt(T(1) - f.a * f.a); for (int k = 0; k < N; ++k) r.v[k] = f.v[k] * d; return r; }
#undef CERES_SHIM_J

class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  int num_residuals() const { return num_residuals_; }
  const std::vector<int>& parameter_block_sizes() const { return sizes_; }
  // test drivers look at what the reference put into its residual blocks (the functor of an AutoDiffCostFunction, e.g. a
  // LidarEdgeFactor with the correspondence it was built from); not part of Ceres
  virtual const void* shim_functor() const { return nullptr; }
  virtual const std::type_info* shim_functor_type() const { return nullptr; }
 protected:
  int num_residuals_ = 0;
  std::vector<int> sizes_;
};

template <class Functor, int kNumResiduals, int N0, int N1>
class AutoDiffCostFunction : public CostFunction {
 public:
  explicit AutoDiffCostFunction(Functor* f) : functor_(f) { num_residuals_ = kNumResiduals; sizes_ = {N0, N1}; }
  const void* shim_functor() const override { return functor_.get(); }
  const std::type_info* shim_functor_type() const override { return &typeid(Functor); }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    if (!jacobians) return (*functor_)(p[0], p[1], residuals);
    typedef Jet<double, N0 + N1> J;
    J x0[N0], x1[N1], r[kNumResiduals];
    for (int k = 0; k < N0; ++k) x0[k] = J(p[0][k], k);
    for (int k = 0; k < N1; ++k) x1[k] = J(p[1][k], N0 + k);
    if (!(*functor_)(x0, x1, r)) return false;
    for (int i = 0; i < kNumResiduals; ++i) {
      residuals[i] = r[i].a;
      if (jacobians[0]) for (int k = 0; k < N0; ++k) jacobians[0][i * N0 + k] = r[i].v[k];
      if (jacobians[1]) for (int k = 0; k < N1; ++k) jacobians[1][i * N1 + k] = r[i].v[N0 + k];
    }
    return true;
  }
 private:
  std::unique_ptr<Functor> functor_;
};

class LossFunction { public: virtual ~LossFunction() {} virtual void Evaluate(double s, double rho[3]) const = 0; };
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
 private:
  double a_, b_;
};

class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;   // GlobalSize x LocalSize, row-major
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
class EigenQuaternionParameterization : public LocalParameterization {
 public:
  bool Plus(const double* x, const double* d, double* out) const override {
    const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd > 0.0) {
      const double s = std::sin(nd) / nd;
      const double qx = s * d[0], qy = s * d[1], qz = s * d[2], qw = std::cos(nd);     // delta quaternion, then delta * x
      out[0] = qw * x[0] + qx * x[3] + qy * x[2] - qz * x[1];
      out[1] = qw * x[1] + qy * x[3] + qz * x[0] - qx * x[2];
      out[2] = qw * x[2] + qz * x[3] + qx * x[1] - qy * x[0];
      out[3] = qw * x[3] - qx * x[0] - qy * x[1] - qz * x[2];
    } else { for (int k = 0; k < 4; ++k) out[k] = x[k]; }
    return true;
  }
  bool ComputeJacobian(const double* x, double* J) const override {
    J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
    J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
    J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
    J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
    return true;
  }
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 3; }
};

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

class Problem {
 public:
  struct Options {};
  Problem() {}
  explicit Problem(const Options&) {}
  ~Problem() { /* the reference leaks nothing observable; objects are owned here like in Ceres */ }
  void AddParameterBlock(double* values, int size, LocalParameterization* lp = nullptr) {
    blocks_.push_back(Block{values, size, std::shared_ptr<LocalParameterization>(lp)});
  }
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1) {
    if (loss && (losses_.empty() || losses_.back().get() != loss)) {
      bool known = false;
      for (auto& l : losses_) known = known || l.get() == loss;
      if (!known) losses_.emplace_back(loss);
    }
    residuals_.push_back(Residual{std::shared_ptr<CostFunction>(cost), loss, {index_of(x0), index_of(x1)}});
  }
  struct Block { double* values; int size; std::shared_ptr<LocalParameterization> lp; };
  struct Residual { std::shared_ptr<CostFunction> cost; LossFunction* loss; int block[2]; };
  std::vector<Block> blocks_;
  std::vector<Residual> residuals_;
  std::vector<std::shared_ptr<LossFunction>> losses_;
 private:
  int index_of(const double* p) const {
    for (size_t i = 0; i < blocks_.size(); ++i) if (blocks_[i].values == p) return static_cast<int>(i);
    std::fprintf(stderr, "ceres shim: residual block uses an unknown parameter block\n");
    std::abort();
  }
};

struct Solver {
  struct Options {
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    int max_num_iterations = 50;
    bool minimizer_progress_to_stdout = false;
    bool check_gradients = false;
    double gradient_check_relative_precision = 1e-8;
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    int max_num_consecutive_invalid_steps = 5;
    bool jacobi_scaling = true;
  };
  struct Summary {
    double initial_cost = 0, final_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0, iterations = 0;
    TerminationType termination_type = NO_CONVERGENCE;
    int shim_termination = 0;   // 0 max-iter, 1 parameter tol, 2 function tol, 3 gradient tol, 4 no residuals, 5 failure, 6 minimum trust-region radius
    std::string message;
    std::string BriefReport() const { return message; }
    std::string FullReport() const { return message; }
  };
};

namespace shim {

struct Evaluator {
  Problem* p;
  int n_local = 0, n_global = 0, n_rows = 0;
  std::vector<int> local_off, global_off;
  explicit Evaluator(Problem* pr) : p(pr) {
    for (auto& b : p->blocks_) {
      local_off.push_back(n_local); global_off.push_back(n_global);
      n_local += b.lp ? b.lp->LocalSize() : b.size;
      n_global += b.size;
    }
    for (auto& r : p->residuals_) n_rows += r.cost->num_residuals();
  }
  void get(std::vector<double>* x) const { x->resize(n_global); for (size_t i = 0; i < p->blocks_.size(); ++i) for (int k = 0; k < p->blocks_[i].size; ++k) (*x)[global_off[i] + k] = p->blocks_[i].values[k]; }
  void plus(const std::vector<double>& x, const std::vector<double>& delta, std::vector<double>* out) const {
    out->resize(n_global);
    for (size_t i = 0; i < p->blocks_.size(); ++i) {
      const auto& b = p->blocks_[i];
      if (b.lp) b.lp->Plus(&x[global_off[i]], &delta[local_off[i]], &(*out)[global_off[i]]);
      else for (int k = 0; k < b.size; ++k) (*out)[global_off[i] + k] = x[global_off[i] + k] + delta[local_off[i] + k];
    }
  }
  // cost = 1/2 sum rho(|r|^2); optionally the corrected residual vector and Jacobian (rows x n_local, row-major) in the tangent space
  bool evaluate(const std::vector<double>& x, double* cost, std::vector<double>* res, std::vector<double>* jac) const {
    *cost = 0.0;
    if (res) res->assign(n_rows, 0.0);
    if (jac) jac->assign(static_cast<size_t>(n_rows) * n_local, 0.0);
    int row = 0;
    for (auto& rb : p->residuals_) {
      const int nr = rb.cost->num_residuals();
      const double* params[2] = {&x[global_off[rb.block[0]]], &x[global_off[rb.block[1]]]};
      double r[8];
      std::vector<double> j0(static_cast<size_t>(nr) * p->blocks_[rb.block[0]].size), j1(static_cast<size_t>(nr) * p->blocks_[rb.block[1]].size);
      double* jp[2] = {j0.data(), j1.data()};
      if (!rb.cost->Evaluate(params, r, jac ? jp : nullptr)) return false;
      double sq = 0.0;
      for (int i = 0; i < nr; ++i) sq += r[i] * r[i];
      double scale = 1.0;
      if (rb.loss) {
        double rho[3];
        rb.loss->Evaluate(sq, rho);
        *cost += 0.5 * rho[0];
        // Corrector: rho'' <= 0 (or s = 0) -> residual and Jacobian rows scaled by sqrt(rho')
        scale = std::sqrt(rho[1]);
        if (sq != 0.0 && rho[2] > 0.0) { std::fprintf(stderr, "ceres shim: rho'' > 0 not supported\n"); std::abort(); }
      } else {
        *cost += 0.5 * sq;
      }
      for (int i = 0; i < nr; ++i) {
        if (res) (*res)[row + i] = scale * r[i];
        if (jac) {
          for (int bi = 0; bi < 2; ++bi) {
            const auto& b = p->blocks_[rb.block[bi]];
            const double* jb = jp[bi] + static_cast<size_t>(i) * b.size;
            double* dst = &(*jac)[static_cast<size_t>(row + i) * n_local + local_off[rb.block[bi]]];
            if (b.lp) {
              std::vector<double> P(static_cast<size_t>(b.size) * b.lp->LocalSize());
              b.lp->ComputeJacobian(params[bi], P.data());
              for (int c = 0; c < b.lp->LocalSize(); ++c) { double s = 0.0; for (int k = 0; k < b.size; ++k) s += jb[k] * P[k * b.lp->LocalSize() + c]; dst[c] = scale * s; }
            } else {
              for (int c = 0; c < b.size; ++c) dst[c] = scale * jb[c];
            }
          }
        }
      }
      row += nr;
    }
    return true;
  }
};

// min |A y - b|^2 by Householder QR (A is rows x n, row-major, overwritten).
inline bool qr_least_squares(std::vector<double>& A, std::vector<double>& b, int rows, int n, std::vector<double>* y) {
  for (int k = 0; k < n; ++k) {
    double s = 0.0;
    for (int i = k; i < rows; ++i) s += A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(i) * n + k];
    if (!(s > 0.0)) return false;
    const double akk = A[static_cast<size_t>(k) * n + k];
    const double alpha = (akk > 0.0 ? -1.0 : 1.0) * std::sqrt(s);
    std::vector<double> v(rows - k);
    for (int i = k; i < rows; ++i) v[i - k] = A[static_cast<size_t>(i) * n + k];
    v[0] -= alpha;
    double vv = 0.0;
    for (double e : v) vv += e * e;
    if (vv > 0.0) {
      for (int j = k; j < n; ++j) {
        double d = 0.0;
        for (int i = k; i < rows; ++i) d += v[i - k] * A[static_cast<size_t>(i) * n + j];
        d = 2.0 * d / vv;
        for (int i = k; i < rows; ++i) A[static_cast<size_t>(i) * n + j] -= d * v[i - k];
      }
      double d = 0.0;
      for (int i = k; i < rows; ++i) d += v[i - k] * b[i];
      d = 2.0 * d / vv;
      for (int i = k; i < rows; ++i) b[i] -= d * v[i - k];
    }
  }
  y->assign(n, 0.0);
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A[static_cast<size_t>(k) * n + j] * (*y)[j];
    (*y)[k] = s / A[static_cast<size_t>(k) * n + k];
    if (!std::isfinite((*y)[k])) return false;
  }
  return true;
}

}  // namespace shim

// test drivers may look at a problem when the reference hands it to Solve (e.g. to dump the correspondences behind its residual blocks)
inline std::function<void(const Problem&)>& shim_solve_hook() { static std::function<void(const Problem&)> h; return h; }

inline void Solve(const Solver::Options& o, Problem* problem, Solver::Summary* sum) {
  if (shim_solve_hook()) shim_solve_hook()(*problem);
  shim::Evaluator ev(problem);
  const int n = ev.n_local, m = ev.n_rows;
  std::vector<double> x, xc, r, J;
  ev.get(&x);
  const std::vector<double> x_entry = x;        // solver.cc Minimize(): a solution that is not usable (FAILURE) is replaced by the original parameters
  double cost = 0.0;
  *sum = Solver::Summary();
  if (m == 0) { sum->shim_termination = 4; sum->termination_type = CONVERGENCE; return; }
  ev.evaluate(x, &cost, &r, &J);
  sum->initial_cost = cost;
  bool jac_finite = true;
  for (double v : J) jac_finite = jac_finite && std::isfinite(v);
  if (!std::isfinite(cost) || !jac_finite) {   // ResidualBlock::Evaluate's validity check (residuals AND Jacobians) fails: FAILURE before the first iteration, x untouched
    sum->final_cost = cost; sum->iterations = 0; sum->shim_termination = 5; sum->termination_type = FAILURE;
    return;
  }
  std::vector<double> scale(n, 1.0);
  if (o.jacobi_scaling) for (int c = 0; c < n; ++c) { double s = 0.0; for (int i = 0; i < m; ++i) s += J[static_cast<size_t>(i) * n + c] * J[static_cast<size_t>(i) * n + c]; scale[c] = 1.0 / (1.0 + std::sqrt(s)); }
  auto scale_jac = [&]() { for (int i = 0; i < m; ++i) for (int c = 0; c < n; ++c) J[static_cast<size_t>(i) * n + c] *= scale[c]; };
  // Ceres 1.12 trust_region_minimizer.cc: gradient_max_norm = |x - Plus(x, -g)|_inf with g = J^T r the (unscaled) tangent-space
  // gradient; J is still unscaled when this is called.  (For a Euclidean block this is |g|_inf; for the quaternion block the
  // step goes through the parameterization.)
  auto grad_max = [&]() {
    std::vector<double> ng(n), xg;
    for (int c = 0; c < n; ++c) { double g = 0.0; for (int i = 0; i < m; ++i) g += J[static_cast<size_t>(i) * n + c] * r[i]; ng[c] = -g; }
    ev.plus(x, ng, &xg);
    double mx = 0.0;
    for (size_t k = 0; k < x.size(); ++k) mx = std::max(mx, std::fabs(x[k] - xg[k]));
    return mx;
  };
  double gmax = grad_max();
  scale_jac();
  auto norm_of = [](const std::vector<double>& v) { double s = 0.0; for (double e : v) s += e * e; return std::sqrt(s); };
  double x_norm = norm_of(x);
  double radius = o.initial_trust_region_radius, decrease = 2.0;
  std::vector<double> diag(n, 0.0);
  bool reuse = false;
  int invalid = 0, iter = 0, term = 0;
  while (true) {
    if (iter >= o.max_num_iterations) { term = 0; break; }
    if (gmax <= o.gradient_tolerance) { term = 3; break; }
    if (radius <= o.min_trust_region_radius) { term = 6; break; }   // MinTrustRegionRadiusReached(): `<=`, CONVERGENCE
    ++iter;
    if (!reuse) for (int c = 0; c < n; ++c) { double s = 0.0; for (int i = 0; i < m; ++i) s += J[static_cast<size_t>(i) * n + c] * J[static_cast<size_t>(i) * n + c]; diag[c] = std::min(std::max(s, o.min_lm_diagonal), o.max_lm_diagonal); }
    reuse = true;
    // stacked system [J; sqrt(diag / radius)] y = [r; 0]; step = -y
    std::vector<double> A(static_cast<size_t>(m + n) * n, 0.0), b(m + n, 0.0), y;
    for (int i = 0; i < m; ++i) { for (int c = 0; c < n; ++c) A[static_cast<size_t>(i) * n + c] = J[static_cast<size_t>(i) * n + c]; b[i] = r[i]; }
    for (int c = 0; c < n; ++c) A[static_cast<size_t>(m + c) * n + c] = std::sqrt(diag[c] / radius);
    const bool ok = shim::qr_least_squares(A, b, m + n, n, &y);
    std::vector<double> step(n);
    double model_change = 0.0;
    if (ok) {
      for (int c = 0; c < n; ++c) step[c] = -y[c];
      for (int i = 0; i < m; ++i) { double ms = 0.0; for (int c = 0; c < n; ++c) ms += J[static_cast<size_t>(i) * n + c] * step[c]; model_change -= ms * (r[i] + ms / 2.0); }
    }
    if (!ok || !(model_change > 0.0)) {
      if (++invalid >= o.max_num_consecutive_invalid_steps) { term = 5; break; }
      radius /= decrease; decrease *= 2.0;
      continue;
    }
    invalid = 0;
    std::vector<double> delta(n);
    for (int c = 0; c < n; ++c) delta[c] = step[c] * scale[c];
    ev.plus(x, delta, &xc);
    double cost_c = 0.0;
    if (!ev.evaluate(xc, &cost_c, nullptr, nullptr)) cost_c = std::numeric_limits<double>::max();
    std::vector<double> dx(x.size());
    for (size_t k = 0; k < x.size(); ++k) dx[k] = x[k] - xc[k];
    if (norm_of(dx) <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { term = 1; break; }
    if (std::fabs(cost - cost_c) <= o.function_tolerance * cost) { term = 2; break; }
    const double rel = (cost - cost_c) / model_change;
    if (rel > o.min_relative_decrease) {
      x = xc; x_norm = norm_of(x);
      ev.evaluate(x, &cost, &r, &J);
      {   // HandleSuccessfulStep(): residuals / Jacobian must evaluate at the accepted point, else FAILURE (the step is not counted)
        bool jf = std::isfinite(cost);
        for (double v : J) jf = jf && std::isfinite(v);
        if (!jf) { term = 5; break; }
      }
      ++sum->num_successful_steps;
      gmax = grad_max();
      scale_jac();
      const double c3 = 2.0 * rel - 1.0;
      radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - c3 * c3 * c3));
      decrease = 2.0; reuse = false;
    } else {
      ++sum->num_unsuccessful_steps;
      radius /= decrease; decrease *= 2.0;
    }
  }
  if (term == 5) { x = x_entry; cost = sum->initial_cost; }   // Summary::IsSolutionUsable() is false: the user's parameter blocks keep their entry values
  for (size_t i = 0; i < problem->blocks_.size(); ++i) for (int k = 0; k < problem->blocks_[i].size; ++k) problem->blocks_[i].values[k] = x[ev.global_off[i] + k];
  sum->final_cost = cost; sum->iterations = iter; sum->shim_termination = term;
  sum->termination_type = (term == 0) ? NO_CONVERGENCE : (term == 5 ? FAILURE : CONVERGENCE);
}

}  // namespace ceres
