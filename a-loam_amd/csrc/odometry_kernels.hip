// a-loam_amd/csrc/odometry_kernels.hip — gfx950 kernels for A-LOAM scan-to-scan odometry.
//
// Replaces, for a BATCH of independent sequences, the solve part of the laserOdometry main loop
// (reference src/laserOdometry.cpp:274-506) and the third-party calls inside it:
//   k_nn_search   TransformToStart (:111-129) + pcl::KdTreeFLANN::nearestKSearch(k=1) (:302,390): exact brute force
//                 over target tiles staged in LDS; one lane per query, broadcast ds_read_b128 per target, the f32
//                 distance ((dx*dx+dy*dy)+dz*dz) FLANN's L2_Simple accumulates; tiles merge through a packed
//                 (distance bits, index) 64-bit atomicMin, so the lowest index wins exact ties.
//   k_walk_corner the ring-adjacent second neighbour walk for edge features (:304-384), one wave per query
//   k_walk_plane  the two-neighbour walk for planar features (:392-482), one wave per query
//   k_solve       ceres::Problem + ceres::Solve (:284-291,380-381,478-479,494-499): per-correspondence
//                 LidarEdgeFactor / LidarPlaneFactor residual + closed-form Jacobian (reference src/lidarFactor.hpp:
//                 18-43,68-90), Huber(0.1) re-weighting, reduction to the 6x6 J^T J / J^T r / cost with wave64
//                 shuffles in f64, and the whole Levenberg-Marquardt trust-region loop (Jacobi scaling, damped
//                 6x6 Cholesky, step quality, radius update, <= 4 iterations) on device; then the pose
//                 integration (:504-505).  One workgroup per sequence.
// No step is a dense contraction (largest "matrix" is 3060 x 6), so no MFMA.
#include "aloam_device.hpp"
#include "odometry_kernels.hpp"

namespace aloam {

// q * v as Eigen evaluates it (uv = 2 u x v; v + w uv + u x uv), f64.
__device__ __forceinline__ void quat_rotate(const double q[4], double vx, double vy, double vz, double out[3]) {
  double ux = q[1] * vz - q[2] * vy, uy = q[2] * vx - q[0] * vz, uz = q[0] * vy - q[1] * vx;
  ux += ux; uy += uy; uz += uz;
  out[0] = vx + q[3] * ux + (q[1] * uz - q[2] * uy);
  out[1] = vy + q[3] * uy + (q[2] * ux - q[0] * uz);
  out[2] = vz + q[3] * uz + (q[0] * uy - q[1] * ux);
}

// TransformToStart with DISTORTION 0: Identity.slerp(1, q) is exactly +-q (sign flips when w < 0), rotation in
// f64, result stored back to f32 (reference src/laserOdometry.cpp:111-129).
__device__ __forceinline__ float4 transform_to_start(const float4& p, const OdomState& st) {
  double q[4] = {st.para_q[0], st.para_q[1], st.para_q[2], st.para_q[3]};
  if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  double o[3];
  quat_rotate(q, (double)p.x, (double)p.y, (double)p.z, o);
  return make_float4((float)(o[0] + st.para_t[0]), (float)(o[1] + st.para_t[1]), (float)(o[2] + st.para_t[2]), p.w);
}

// -------------------------------------------------------------------------------------------------------
// grid (tile slots, query tiles, B); which = 0 corners (sharp vs corner_last), 1 planes (flat vs surf_last).
// Each workgroup strides over the target tiles (the true target count is only known on the device).
__global__ __launch_bounds__(256) void k_nn_search(OdomArgs a, int which) {
  const int b = blockIdx.z, tid = threadIdx.x;
  const SeqMeta m = a.meta[b];
  const int nq = which == 0 ? m.n_sharp : m.n_flat;
  const int nt = which == 0 ? m.n_corner_last : m.n_surf_last;
  const int q0 = blockIdx.y * 256;
  if (q0 >= nq || (int)blockIdx.x * kNnTile >= nt) return;
  const float4* queries = which == 0 ? a.sharp + (long long)b * a.R * 12 : a.flat + (long long)b * a.R * 24;
  const float4* targets = which == 0 ? a.corner_last + (long long)b * a.R * 120 : a.surf_last + (long long)b * a.cap;
  unsigned long long* nn = which == 0 ? a.nn_corner + (long long)b * a.R * 12 : a.nn_surf + (long long)b * a.R * 24;
  __shared__ float4 tile[kNnTile];
  const int qi = q0 + tid;
  float4 sel = make_float4(0.f, 0.f, 0.f, 0.f);
  if (qi < nq) sel = transform_to_start(queries[qi], a.state[b]);
  float best = __int_as_float(0x7f800000);
  int besti = -1;
  for (int t0 = blockIdx.x * kNnTile; t0 < nt; t0 += gridDim.x * kNnTile) {
    const int tc = (nt - t0 < kNnTile) ? nt - t0 : kNnTile;
    __syncthreads();
    for (int t = tid; t < tc; t += 256) tile[t] = targets[t0 + t];
    __syncthreads();
#pragma unroll 4
    for (int t = 0; t < tc; ++t) {
      const float4 p = tile[t];
      const float dx = p.x - sel.x, dy = p.y - sel.y, dz = p.z - sel.z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d < best) { best = d; besti = t0 + t; }
    }
  }
  if (qi < nq && besti >= 0) {
    const unsigned long long packed = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)besti;
    atomicMin(&nn[qi], packed);
  }
}

__device__ __forceinline__ float walk_dist(const float4& p, const float4& sel) {     // f32 expression (:322-327)
  return (p.x - sel.x) * (p.x - sel.x) + (p.y - sel.y) * (p.y - sel.y) + (p.z - sel.z) * (p.z - sel.z);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  for (int msk = 32; msk > 0; msk >>= 1) { const unsigned long long o = shfl_xor_u64(v, msk); v = o < v ? o : v; }
  return v;
}

// -------------------------------------------------------------------------------------------------------
// One wave per sharp query.  Candidates are ordered exactly as the reference visits them (upward walk from
// closest+1, then downward from closest-1); "first strictly smaller wins" becomes a lexicographic
// (distance, visit order) minimum.
__global__ __launch_bounds__(256) void k_walk_corner(OdomArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  const SeqMeta m = a.meta[b];
  if (qi >= m.n_sharp) return;
  const float4* CL = a.corner_last + (long long)b * a.R * 120;
  const int nt = m.n_corner_last;
  EdgeRec* rec = a.edges + (long long)b * a.R * 12 + qi;
  const float4 raw = a.sharp[(long long)b * a.R * 12 + qi];
  const unsigned long long packed = a.nn_corner[(long long)b * a.R * 12 + qi];
  const unsigned idx = (unsigned)packed;
  const float nnd = __uint_as_float((unsigned)(packed >> 32));
  int valid = 0;
  int closest = -1, min2 = -1;
  if (idx != 0xffffffffu && (double)nnd < 25.0) {                     // DISTANCE_SQ_THRESHOLD (:65,305)
    closest = (int)idx;
    const float4 sel = transform_to_start(raw, a.state[b]);
    const int cid = (int)CL[closest].w;                               // closestPointScanID (:308)
    unsigned long long best = ~0ull;
    // upward (:312-335)
    for (int base = closest + 1; base < nt; base += 64) {
      const int j = base + lane;
      bool stop = false, cand = false;
      float d = 0.f;
      if (j < nt) {
        const float4 p = CL[j];
        const int key = (int)p.w;
        if (key > cid) {                                              // `<= cid` -> continue
          if ((double)key > (double)cid + 2.5) stop = true;           // NEARBY_SCAN (:66,319)
          else { d = walk_dist(p, sel); cand = (double)d < 25.0; }
        }
      }
      const unsigned long long sm = __ballot(stop);
      const int first_stop = sm ? (__ffsll((long long)sm) - 1) : 64;
      if (cand && lane < first_stop) {
        const unsigned long long v = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(j - closest);
        if (v < best) best = v;
      }
      if (sm) break;
    }
    // downward (:338-361)
    for (int base = closest - 1; base >= 0; base -= 64) {
      const int j = base - lane;
      bool stop = false, cand = false;
      float d = 0.f;
      if (j >= 0) {
        const float4 p = CL[j];
        const int key = (int)p.w;
        if (key < cid) {                                              // `>= cid` -> continue
          if ((double)key < (double)cid - 2.5) stop = true;
          else { d = walk_dist(p, sel); cand = (double)d < 25.0; }
        }
      }
      const unsigned long long sm = __ballot(stop);
      const int first_stop = sm ? (__ffsll((long long)sm) - 1) : 64;
      if (cand && lane < first_stop) {
        const unsigned long long v = ((unsigned long long)__float_as_uint(d) << 32) | (0x40000000u + (unsigned)(closest - j));
        if (v < best) best = v;
      }
      if (sm) break;
    }
    best = wave_min_u64(best);
    if (best != ~0ull) {
      const unsigned seq = (unsigned)best;
      min2 = seq >= 0x40000000u ? closest - (int)(seq - 0x40000000u) : closest + (int)seq;
      valid = 1;
    }
  }
  if (lane == 0) {
    EdgeRec e;
    e.valid = valid;
    e.pad[0] = e.pad[1] = 0;
    e.cp[0] = raw.x; e.cp[1] = raw.y; e.cp[2] = raw.z;                // raw, untransformed point (:365-367)
    if (valid) {
      const float4 pa = CL[closest], pb = CL[min2];
      e.a[0] = pa.x; e.a[1] = pa.y; e.a[2] = pa.z;
      e.b[0] = pb.x; e.b[1] = pb.y; e.b[2] = pb.z;
    } else {
      e.a[0] = e.a[1] = e.a[2] = e.b[0] = e.b[1] = e.b[2] = 0.f;
    }
    *rec = e;
  }
}

// -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_walk_plane(OdomArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  const SeqMeta m = a.meta[b];
  if (qi >= m.n_flat) return;
  const float4* SL = a.surf_last + (long long)b * a.cap;
  const int nt = m.n_surf_last;
  PlaneRec* rec = a.planes + (long long)b * a.R * 24 + qi;
  const float4 raw = a.flat[(long long)b * a.R * 24 + qi];
  const unsigned long long packed = a.nn_surf[(long long)b * a.R * 24 + qi];
  const unsigned idx = (unsigned)packed;
  const float nnd = __uint_as_float((unsigned)(packed >> 32));
  int valid = 0;
  int closest = -1, min2 = -1, min3 = -1;
  if (idx != 0xffffffffu && (double)nnd < 25.0) {
    closest = (int)idx;
    const float4 sel = transform_to_start(raw, a.state[b]);
    const int cid = (int)SL[closest].w;
    unsigned long long best2 = ~0ull, best3 = ~0ull;
    // upward (:402-427): same-or-lower ring -> min2, higher ring -> min3
    for (int base = closest + 1; base < nt; base += 64) {
      const int j = base + lane;
      bool stop = false, c2 = false, c3 = false;
      float d = 0.f;
      if (j < nt) {
        const float4 p = SL[j];
        const int key = (int)p.w;
        if ((double)key > (double)cid + 2.5) stop = true;
        else { d = walk_dist(p, sel); const bool in = (double)d < 25.0; c2 = in && key <= cid; c3 = in && key > cid; }
      }
      const unsigned long long sm = __ballot(stop);
      const int first_stop = sm ? (__ffsll((long long)sm) - 1) : 64;
      if (lane < first_stop) {
        const unsigned long long v = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(j - closest);
        if (c2 && v < best2) best2 = v;
        if (c3 && v < best3) best3 = v;
      }
      if (sm) break;
    }
    // downward (:430-455): same-or-higher ring -> min2, lower ring -> min3
    for (int base = closest - 1; base >= 0; base -= 64) {
      const int j = base - lane;
      bool stop = false, c2 = false, c3 = false;
      float d = 0.f;
      if (j >= 0) {
        const float4 p = SL[j];
        const int key = (int)p.w;
        if ((double)key < (double)cid - 2.5) stop = true;
        else { d = walk_dist(p, sel); const bool in = (double)d < 25.0; c2 = in && key >= cid; c3 = in && key < cid; }
      }
      const unsigned long long sm = __ballot(stop);
      const int first_stop = sm ? (__ffsll((long long)sm) - 1) : 64;
      if (lane < first_stop) {
        const unsigned long long v = ((unsigned long long)__float_as_uint(d) << 32) | (0x40000000u + (unsigned)(closest - j));
        if (c2 && v < best2) best2 = v;
        if (c3 && v < best3) best3 = v;
      }
      if (sm) break;
    }
    best2 = wave_min_u64(best2);
    best3 = wave_min_u64(best3);
    if (best2 != ~0ull && best3 != ~0ull) {                            // :457
      const unsigned s2 = (unsigned)best2, s3 = (unsigned)best3;
      min2 = s2 >= 0x40000000u ? closest - (int)(s2 - 0x40000000u) : closest + (int)s2;
      min3 = s3 >= 0x40000000u ? closest - (int)(s3 - 0x40000000u) : closest + (int)s3;
      valid = 1;
    }
  }
  if (lane == 0) {
    PlaneRec e;
    e.valid = valid;
    e.pad[0] = e.pad[1] = e.pad[2] = 0;
    e.cp[0] = raw.x; e.cp[1] = raw.y; e.cp[2] = raw.z;
    if (valid) {
      const float4 pj = SL[closest], pl = SL[min2], pm = SL[min3];
      e.j[0] = pj.x; e.j[1] = pj.y; e.j[2] = pj.z;
      e.l[0] = pl.x; e.l[1] = pl.y; e.l[2] = pl.z;
      e.m[0] = pm.x; e.m[1] = pm.y; e.m[2] = pm.z;
    } else {
      for (int k = 0; k < 3; ++k) e.j[k] = e.l[k] = e.m[k] = 0.f;
    }
    *rec = e;
  }
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

template <bool WITH_JAC>
__device__ void evaluate(const OdomArgs& a, int b, const double q[4], const double t[3], double* acc, int* n_edge, int* n_plane) {
  const int tid = threadIdx.x;
  const SeqMeta m = a.meta[b];
  const EdgeRec* E = a.edges + (long long)b * a.R * 12;
  const PlaneRec* P = a.planes + (long long)b * a.R * 24;
  int ne = 0, np = 0;
  for (int i = tid; i < m.n_sharp; i += 256) {
    const EdgeRec e = E[i];
    if (!e.valid) continue;
    ++ne;
    double rcp[3];
    quat_rotate(q, (double)e.cp[0], (double)e.cp[1], (double)e.cp[2], rcp);
    const double lp[3] = {rcp[0] + t[0], rcp[1] + t[1], rcp[2] + t[2]};
    const double ax = e.a[0], ay = e.a[1], az = e.a[2], bx = e.b[0], by = e.b[1], bz = e.b[2];
    const double dex = ax - bx, dey = ay - by, dez = az - bz;
    const double inv = 1.0 / sqrt(dex * dex + dey * dey + dez * dez);
    const double ux = lp[0] - ax, uy = lp[1] - ay, uz = lp[2] - az, vx = lp[0] - bx, vy = lp[1] - by, vz = lp[2] - bz;
    const double r0 = (uy * vz - uz * vy) * inv, r1 = (uz * vx - ux * vz) * inv, r2 = (ux * vy - uy * vx) * inv;
    double rho0, rho1;
    huber(r0 * r0 + r1 * r1 + r2 * r2, &rho0, &rho1);
    acc[27] += 0.5 * rho0;
    if (WITH_JAC) {
      // d r / d lp = [w]x, w = (b - a)/|a-b|;  d lp / d delta = -2 [R cp]x;  d lp / d t = I
      const double wx = -dex * inv, wy = -dey * inv, wz = -dez * inv;
      const double A[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
      const double Bm[3][3] = {{0, 2 * rcp[2], -2 * rcp[1]}, {-2 * rcp[2], 0, 2 * rcp[0]}, {2 * rcp[1], -2 * rcp[0], 0}};
      const double rr[3] = {r0, r1, r2};
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        double J[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          J[c] = A[row][0] * Bm[0][c] + A[row][1] * Bm[1][c] + A[row][2] * Bm[2][c];
          J[3 + c] = A[row][c];
        }
        add_row(acc, J, rr[row], rho1);
      }
    }
  }
  for (int i = tid; i < m.n_flat; i += 256) {
    const PlaneRec p = P[i];
    if (!p.valid) continue;
    ++np;
    // LidarPlaneFactor ctor: n = normalize((j - l) x (j - m))  (reference src/lidarFactor.hpp:64-65)
    const double jx = p.j[0], jy = p.j[1], jz = p.j[2];
    const double e1x = jx - (double)p.l[0], e1y = jy - (double)p.l[1], e1z = jz - (double)p.l[2];
    const double e2x = jx - (double)p.m[0], e2y = jy - (double)p.m[1], e2z = jz - (double)p.m[2];
    double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    nx /= len; ny /= len; nz /= len;
    double rcp[3];
    quat_rotate(q, (double)p.cp[0], (double)p.cp[1], (double)p.cp[2], rcp);
    const double r = (rcp[0] + t[0] - jx) * nx + (rcp[1] + t[1] - jy) * ny + (rcp[2] + t[2] - jz) * nz;
    double rho0, rho1;
    huber(r * r, &rho0, &rho1);
    acc[27] += 0.5 * rho0;
    if (WITH_JAC) {
      const double J[6] = {2.0 * (nz * rcp[1] - ny * rcp[2]), 2.0 * (nx * rcp[2] - nz * rcp[0]), 2.0 * (ny * rcp[0] - nx * rcp[1]), nx, ny, nz};
      add_row(acc, J, r, rho1);
    }
  }
  *n_edge = ne;
  *n_plane = np;
}

// block-wide sum of NV doubles per thread; result broadcast to every thread (s_red: [4][NV] doubles of LDS).
template <int NV>
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
  for (int k = 0; k < NV; ++k) v[k] = (s_red[k] + s_red[NV + k]) + (s_red[2 * NV + k] + s_red[3 * NV + k]);
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

// One workgroup per sequence: the whole ceres::Solve stand-in + pose integration.
// Every thread runs the (uniform) scalar LM logic redundantly; only the evaluations are distributed.
__global__ __launch_bounds__(256) void k_solve(OdomArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double s_red[4 * 28];
  OdomState& st = a.state[b];
  double q[4] = {st.para_q[0], st.para_q[1], st.para_q[2], st.para_q[3]};
  double t[3] = {st.para_t[0], st.para_t[1], st.para_t[2]};

  const double kFunctionTol = 1e-6, kGradientTol = 1e-10, kParameterTol = 1e-8, kMinRelDecrease = 1e-3;
  const double kMinDiag = 1e-6, kMaxDiag = 1e32, kMaxRadius = 1e16, kMinRadius = 1e-32;

  double acc[28];
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  int ne = 0, np = 0;
  evaluate<true>(a, b, q, t, acc, &ne, &np);
  double cnt[2] = {(double)ne, (double)np};
  block_sum<28>(acc, s_red);
  block_sum<2>(cnt, s_red);
  const int n_edges = (int)cnt[0], n_planes = (int)cnt[1];

  int iterations = 0, successful = 0, termination = 0;
  double cost = acc[27];
  const double initial_cost = cost;

  if (n_edges + n_planes == 0) {
    termination = 4;
  } else {
    double H[6][6], g[6], scale[6];
    auto unpack = [&](const double* s) {
      int o = 0;
      for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { H[i][j] = s[o]; H[j][i] = s[o]; ++o; }
      for (int i = 0; i < 6; ++i) g[i] = s[21 + i];
    };
    unpack(acc);
    for (int c = 0; c < 6; ++c) scale[c] = 1.0 / (1.0 + sqrt(H[c][c]));           // Jacobi scaling, first Jacobian only
    auto gradient_max = [&]() { double mx = 0.0; for (int c = 0; c < 6; ++c) mx = fmax(mx, fabs(g[c])); return mx; };
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
      if (iter >= a.lm_max_iterations) { termination = 0; break; }
      if (gmax <= kGradientTol) { termination = 3; break; }
      if (radius < kMinRadius) { termination = 5; break; }
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
        if (++n_invalid >= 5) { termination = 5; break; }
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
        continue;
      }
      n_invalid = 0;
      double delta[6], qc[4], tc[3];
      for (int c = 0; c < 6; ++c) delta[c] = step[c] * scale[c];
      quat_plus(q, delta, qc);
      for (int k = 0; k < 3; ++k) tc[k] = t[k] + delta[3 + k];
      double cacc[28];
      for (int k = 0; k < 28; ++k) cacc[k] = 0.0;
      evaluate<false>(a, b, qc, tc, cacc, &ne, &np);
      double cc[1] = {cacc[27]};
      block_sum<1>(cc, s_red);
      const double cost_c = cc[0];
      double sn = 0.0;
      for (int k = 0; k < 4; ++k) sn += (q[k] - qc[k]) * (q[k] - qc[k]);
      for (int k = 0; k < 3; ++k) sn += (t[k] - tc[k]) * (t[k] - tc[k]);
      sn = sqrt(sn);
      if (sn <= kParameterTol * (x_norm + kParameterTol)) { termination = 1; break; }   // x NOT updated (Ceres >= 1.12)
      if (fabs(cost - cost_c) <= kFunctionTol * cost) { termination = 2; break; }
      const double rel = (cost - cost_c) / model_change;
      if (rel > kMinRelDecrease) {
        for (int k = 0; k < 4; ++k) q[k] = qc[k];
        for (int k = 0; k < 3; ++k) t[k] = tc[k];
        x_norm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        for (int k = 0; k < 28; ++k) acc[k] = 0.0;
        evaluate<true>(a, b, q, t, acc, &ne, &np);
        block_sum<28>(acc, s_red);
        cost = acc[27];
        unpack(acc);
        gmax = gradient_max();
        apply_scale();
        ++successful;
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

  if (tid == 0) {
    for (int k = 0; k < 4; ++k) st.para_q[k] = q[k];
    for (int k = 0; k < 3; ++k) st.para_t[k] = t[k];
    const int o = a.outer < 2 ? a.outer : 1;
    st.corner_corr[o] = n_edges;
    st.plane_corr[o] = n_planes;
    st.lm_iterations[o] = iterations;
    st.lm_successful[o] = successful;
    st.initial_cost[o] = initial_cost;
    st.final_cost[o] = cost;
    st.termination[o] = termination;
    if (a.last_outer) {                                   // t_w += q_w * t_lc;  q_w = q_w * q_lc  (:504-505)
      const double qw[4] = {st.q_w[0], st.q_w[1], st.q_w[2], st.q_w[3]};
      double rt[3];
      quat_rotate(qw, t[0], t[1], t[2], rt);
      st.t_w[0] += rt[0]; st.t_w[1] += rt[1]; st.t_w[2] += rt[2];
      st.q_w[0] = qw[3] * q[0] + qw[0] * q[3] + qw[1] * q[2] - qw[2] * q[1];
      st.q_w[1] = qw[3] * q[1] + qw[1] * q[3] + qw[2] * q[0] - qw[0] * q[2];
      st.q_w[2] = qw[3] * q[2] + qw[2] * q[3] + qw[0] * q[1] - qw[1] * q[0];
      st.q_w[3] = qw[3] * q[3] - qw[0] * q[0] - qw[1] * q[1] - qw[2] * q[2];
    }
  }
}

// Cloud swap bookkeeping (reference src/laserOdometry.cpp:554-563): the buffers are swapped by the host (pointer
// flip), the counts here.
__global__ void k_advance(SeqMeta* meta, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { meta[b].n_corner_last = meta[b].n_less_sharp; meta[b].n_surf_last = meta[b].n_less_flat; }
}
void launch_advance(SeqMeta* meta, int B, hipStream_t s) { hipLaunchKernelGGL(k_advance, dim3((B + 63) / 64), dim3(64), 0, s, meta, B); }

// -------------------------------------------------------------------------------------------------------
void launch_nn_search(const OdomArgs& a, int which, int max_queries, int max_targets, hipStream_t s) {
  int tiles = (max_targets + kNnTile - 1) / kNnTile;
  if (tiles > 48) tiles = 48;
  const dim3 grid(tiles, (max_queries + 255) / 256, a.B);
  hipLaunchKernelGGL(k_nn_search, grid, dim3(256), 0, s, a, which);
}
void launch_walk_corner(const OdomArgs& a, int max_queries, hipStream_t s) {
  hipLaunchKernelGGL(k_walk_corner, dim3((max_queries + 3) / 4, a.B), dim3(256), 0, s, a);
}
void launch_walk_plane(const OdomArgs& a, int max_queries, hipStream_t s) {
  hipLaunchKernelGGL(k_walk_plane, dim3((max_queries + 3) / 4, a.B), dim3(256), 0, s, a);
}
void launch_solve(const OdomArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_solve, dim3(a.B), dim3(256), 0, s, a); }

}  // namespace aloam
