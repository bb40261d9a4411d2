// a-loam_amd/csrc/lm_device.hpp — device-side pieces shared by the odometry and mapping solvers: quaternion helpers, Huber
// re-weighting, the 6x6 normal-equation accumulation, block reductions in f64 and the Levenberg-Marquardt trust-region
// loop that stands in for ceres::Solve as the reference configures it (DENSE_QR, max_num_iterations = 4, everything else
// default; reference src/laserOdometry.cpp:494-499, src/laserMapping.cpp:712-720).
#pragma once
#include "aloam_device.hpp"

namespace aloam {

// q * v as Eigen evaluates it (uv = 2 u x v; v + w uv + u x uv), f64.
__device__ __forceinline__ void quat_rotate(const double q[4], double vx, double vy, double vz, double out[3]) {
  double ux = q[1] * vz - q[2] * vy, uy = q[2] * vx - q[0] * vz, uz = q[0] * vy - q[1] * vx;
  ux += ux; uy += uy; uz += uz;
  out[0] = vx + q[3] * ux + (q[1] * uz - q[2] * uy);
  out[1] = vy + q[3] * uy + (q[2] * ux - q[0] * uz);
  out[2] = vz + q[3] * uz + (q[0] * uy - q[1] * ux);
}

// -------------------------------------------------------------------------------------------------------
// Robust Gauss-Newton sums of one evaluation point.  acc[0..20] upper triangle of J^T J (row-major), acc[21..26]
// J^T r, acc[27] cost; rows are already scaled by sqrt(rho') (Ceres Corrector with rho'' <= 0).
__device__ __forceinline__ void huber(double s, double* rho0, double* rho1) {      // HuberLoss(0.1)
  const double aa = 0.1, bb = aa * aa;
  if (s > bb) { const double r = sqrt(s); *rho0 = 2.0 * aa * r - bb; *rho1 = fmax(2.2250738585072014e-308, aa / r); }
  else { *rho0 = s; *rho1 = 1.0; }
}

__device__ __forceinline__ void add_row(double* acc, const double J[6], double r, double w) {
  int o = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = i; j < 6; ++j) acc[o++] += w * J[i] * J[j];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] += w * J[i] * r;
}

// block-wide sum of NV doubles per thread over NW waves; result broadcast to every thread (s_red: [NW][NV] doubles of LDS).
template <int NV, int NW = 4>
__device__ __forceinline__ void block_sum(double* v, double* s_red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
    for (int d = 32; d > 0; d >>= 1) x += shfl_down_f64(x, d);
    if (lane == 0) s_red[wave * NV + k] = x;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (NW == 4) v[k] = (s_red[k] + s_red[NV + k]) + (s_red[2 * NV + k] + s_red[3 * NV + k]);
    else { double x = s_red[k]; for (int w = 1; w < NW; ++w) x += s_red[w * NV + k]; v[k] = x; }
  }
  __syncthreads();
}

// EigenQuaternionParameterization::Plus: (cos|d|, sin|d| d/|d|) * q  (Ceres local_parameterization.cc)
__device__ __forceinline__ void quat_plus(const double q[4], const double d[3], double out[4]) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd > 0.0) {
    const double k = sin(nd) / nd;
    const double ax = k * d[0], ay = k * d[1], az = k * d[2], aw = cos(nd);
    out[0] = aw * q[0] + ax * q[3] + ay * q[2] - az * q[1];
    out[1] = aw * q[1] + ay * q[3] + az * q[0] - ax * q[2];
    out[2] = aw * q[2] + az * q[3] + ax * q[1] - ay * q[0];
    out[3] = aw * q[3] - ax * q[0] - ay * q[1] - az * q[2];
  } else {
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  }
}

// Solve (H + diag(D2)) y = g for the 6x6 SPD system by Cholesky; returns false if not positive definite.
__device__ __forceinline__ bool chol_solve6(const double H[6][6], const double D2[6], const double g[6], double y[6]) {
  double Lm[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = H[i][j] + (i == j ? D2[i] : 0.0);
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        Lm[i][i] = sqrt(s);
      } else {
        Lm[i][j] = s / Lm[j][j];
      }
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= Lm[i][k] * z[k];
    z[i] = s / Lm[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= Lm[k][i] * y[k];
    y[i] = s / Lm[i][i];
  }
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; ++i) ok = ok && isfinite(y[i]);
  return ok;
}


struct LmResult { int iterations, successful, termination, n_a, n_b; double initial_cost, final_cost; };

// The whole ceres::Solve stand-in for one sequence, executed by one 256-thread workgroup.  Every thread runs the (uniform)
// scalar LM logic redundantly; only the evaluations are distributed.  `eval(with_jac, q, t, acc, &n_a, &n_b)` adds this
// thread's share of the robustified sums at (q, t): acc[0..20] upper triangle of J^T J, acc[21..26] J^T r, acc[27] cost, and
// counts the residual blocks of the two factor classes it visited.  q (xyzw) and t are updated in place; after termination 5 (FAILURE) the
// caller must put the values it passed in back, as ceres::Solve does with a solution that is not usable.
__device__ __forceinline__ bool all_finite(const double* v, int n) { bool f = true; for (int k = 0; k < n; ++k) f = f && isfinite(v[k]); return f; }

template <int NW = 4, class Eval>
__device__ __forceinline__ LmResult lm_solve_block(Eval&& eval, double q[4], double t[3], int lm_max_iterations, double* s_red) {
  const double kFunctionTol = 1e-6, kGradientTol = 1e-10, kParameterTol = 1e-8, kMinRelDecrease = 1e-3;
  const double kMinDiag = 1e-6, kMaxDiag = 1e32, kMaxRadius = 1e16, kMinRadius = 1e-32;

  double acc[28];
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  int ne = 0, np = 0;
  eval(true, q, t, acc, &ne, &np);
  double cnt[2] = {(double)ne, (double)np};
  block_sum<28, NW>(acc, s_red);
  block_sum<2, NW>(cnt, s_red);
  const int n_edges = (int)cnt[0], n_planes = (int)cnt[1];

  int iterations = 0, successful = 0, termination = 0;
  double cost = acc[27];
  const double initial_cost = cost;

  if (n_edges + n_planes == 0) {
    termination = 4;
  } else if (!isfinite(cost) || !all_finite(acc, 27)) {
    termination = 5;              // Ceres: "Residual and Jacobian evaluation failed" (non-finite residual or Jacobian: J^T J / J^T r not finite — the sums,
                                  // see the note at the accepted step), parameters untouched
  } else {
    double H[6][6], g[6], scale[6];
    auto unpack = [&](const double* s) {
      int o = 0;
      for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { H[i][j] = s[o]; H[j][i] = s[o]; ++o; }
      for (int i = 0; i < 6; ++i) g[i] = s[21 + i];
    };
    unpack(acc);
    for (int c = 0; c < 6; ++c) scale[c] = 1.0 / (1.0 + sqrt(H[c][c]));           // Jacobi scaling, first Jacobian only
    // Ceres 1.12: gradient_max_norm = |x - Plus(x, -g)|_inf with the unscaled tangent-space gradient g (called before apply_scale)
    auto gradient_max = [&]() {
      const double ng[3] = {-g[0], -g[1], -g[2]};
      double qg[4];
      quat_plus(q, ng, qg);
      double mx = 0.0;
      for (int k = 0; k < 4; ++k) mx = fmax(mx, fabs(q[k] - qg[k]));
      for (int k = 0; k < 3; ++k) { const double tg = t[k] + (-g[3 + k]); mx = fmax(mx, fabs(t[k] - tg)); }   // literally |x - Plus(x, -g)|: quantised by ulp(t) like Ceres' own
      return mx;
    };
    double gmax = gradient_max();
    auto apply_scale = [&]() {
      for (int i = 0; i < 6; ++i) { g[i] *= scale[i]; for (int j = 0; j < 6; ++j) H[i][j] *= scale[i] * scale[j]; }
    };
    apply_scale();
    double x_norm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double radius = 1e4, decrease_factor = 2.0, diag[6] = {0, 0, 0, 0, 0, 0};
    bool reuse_diagonal = false;
    int n_invalid = 0;
    int iter = 0;
    while (true) {
      if (iter >= lm_max_iterations) { termination = 0; break; }
      if (gmax <= kGradientTol) { termination = 3; break; }
      if (radius <= kMinRadius) { termination = 6; break; }   // Ceres: MinTrustRegionRadiusReached(), `<=`, CONVERGENCE (not reachable in 4 iterations from 1e4)
      ++iter;
      iterations = iter;
      if (!reuse_diagonal) for (int c = 0; c < 6; ++c) diag[c] = fmin(fmax(H[c][c], kMinDiag), kMaxDiag);
      double D2[6], y[6], step[6];
      for (int c = 0; c < 6; ++c) D2[c] = diag[c] / radius;
      const bool ok = chol_solve6(H, D2, g, y);
      for (int c = 0; c < 6; ++c) step[c] = -y[c];
      reuse_diagonal = true;
      double model_change = 0.0;
      if (ok) {
        double sg = 0.0, shs = 0.0;
        for (int i = 0; i < 6; ++i) { sg += step[i] * g[i]; double hs = 0.0; for (int j = 0; j < 6; ++j) hs += H[i][j] * step[j]; shs += step[i] * hs; }
        model_change = -sg - 0.5 * shs;                   // -(J s)^T (r + J s / 2)
      }
      if (!ok || !(model_change > 0.0)) {
        if (++n_invalid >= 5) { termination = 5; cost = initial_cost; break; }   // FAILURE: the caller restores the entry parameters (see the accepted step)
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
        continue;
      }
      n_invalid = 0;
      double delta[6], qc[4], tc[3];
      for (int c = 0; c < 6; ++c) delta[c] = step[c] * scale[c];
      quat_plus(q, delta, qc);
      for (int k = 0; k < 3; ++k) tc[k] = t[k] + delta[3 + k];
      // Ceres evaluates the cost at the candidate and, if the step is accepted, residuals + Jacobian there once more.  Here one
      // pass at the candidate yields both: the cost is the same sum of the same terms, an accepted step needs no second pass over
      // the correspondence records, and only a rejected step (rare) has computed a Jacobian for nothing.
      double cacc[28];
      for (int k = 0; k < 28; ++k) cacc[k] = 0.0;
      eval(true, qc, tc, cacc, &ne, &np);
      block_sum<28, NW>(cacc, s_red);
      const double cost_c = cacc[27];
      double sn = 0.0;
      for (int k = 0; k < 4; ++k) sn += (q[k] - qc[k]) * (q[k] - qc[k]);
      for (int k = 0; k < 3; ++k) sn += (t[k] - tc[k]) * (t[k] - tc[k]);
      sn = sqrt(sn);
      if (sn <= kParameterTol * (x_norm + kParameterTol)) { termination = 1; break; }   // x NOT updated (Ceres >= 1.12)
      if (fabs(cost - cost_c) <= kFunctionTol * cost) { termination = 2; break; }
      const double rel = (cost - cost_c) / model_change;
      if (rel > kMinRelDecrease) {
        // HandleSuccessfulStep(): a Jacobian that cannot be evaluated at the accepted point ends the solve as FAILURE, and a failed solve
        // leaves the user's parameters as they were at entry (Ceres writes x back only from a usable solution): the caller re-reads them
        // (termination 5), the step is not counted, the cost reported is the initial one.  (The sums are tested, not the entries: a finite
        // entry beyond ~1e154 overflows in J^T J and reads as non-finite here, where Ceres and the CPU restatements used by the tests
        // would go on — no real sweep comes within 150 orders of magnitude of that.)
        if (!all_finite(cacc, 28)) { termination = 5; cost = initial_cost; break; }
        for (int k = 0; k < 4; ++k) q[k] = qc[k];
        for (int k = 0; k < 3; ++k) t[k] = tc[k];
        x_norm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        for (int k = 0; k < 28; ++k) acc[k] = cacc[k];
        cost = acc[27];
        ++successful;
        unpack(acc);
        gmax = gradient_max();
        apply_scale();
        const double c3 = 2.0 * rel - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - c3 * c3 * c3);
        radius = fmin(kMaxRadius, radius);
        decrease_factor = 2.0;
        reuse_diagonal = false;
      } else {
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
        reuse_diagonal = true;
      }
    }
  }

  LmResult res;
  res.iterations = iterations; res.successful = successful; res.termination = termination; res.n_a = n_edges; res.n_b = n_planes;
  res.initial_cost = initial_cost; res.final_cost = cost;
  return res;
}

}  // namespace aloam
