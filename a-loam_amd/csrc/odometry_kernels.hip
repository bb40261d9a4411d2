// a-loam_amd/csrc/odometry_kernels.hip — gfx950 kernels for A-LOAM scan-to-scan odometry.
//
// Replaces, for a BATCH of independent sequences, the solve part of the laserOdometry main loop
// (reference src/laserOdometry.cpp:274-506) and the third-party calls inside it:
//   k_build_grids_fused / k_build_grids
//                 pcl::KdTreeFLANN::setInputCloud (:567-568): LDS counting-sort of each "last" cloud into two spatial
//                 hash grids (3-D cells on two levels, (x, y, ring) cells) + the flag that says whether the cloud is ring-sorted, which is what turns
//                 the reference's walk-until-break loops into "ring key within +-2" tests
//   k_associate_pair (round 4; k_associate = the one-query-per-wave form of rounds 1-3, kept for A/B builds and, as k_associate_flagged, for clouds that
//                 are not ring-sorted)
//                 nearestKSearch(k=1) (:302,390) + the ring-adjacent second / third neighbour walks (:304-384, :392-482) on the features
//                 k_transform_queries moved to the sweep start (:111-129), TWO queries per wave: exact 1-NN over the 3x3x3 block of fine hash
//                 cells with the f32 distance ((dx*dx+dy*dy)+dz*dz) FLANN's L2_Simple accumulates (lowest index wins exact ties), the walk
//                 candidates from the keys that block left in registers; coarse shells / the (x, y, ring) grid as tails for the queries whose
//                 neighbours lie beyond the block.  The association does not depend on aloam_config.distortion (the features arrive
//                 de-skewed from k_transform_queries; the interpolation ratio travels in the record for k_solve)
//   k_solve       ceres::Problem + ceres::Solve (:284-291,380-381,478-479,494-499): per-correspondence
//                 LidarEdgeFactor / LidarPlaneFactor residual + closed-form Jacobian (reference src/lidarFactor.hpp:
//                 18-43,68-90), Huber(0.1) re-weighting, reduction to the 6x6 J^T J / J^T r / cost with wave64
//                 shuffles in f64, and the whole Levenberg-Marquardt trust-region loop (Jacobi scaling, damped
//                 6x6 Cholesky, step quality, radius update, <= 4 iterations) on device; then the pose
//                 integration (:504-505).  One workgroup per sequence.
// No step is a dense contraction (largest "matrix" is 3060 x 6), so no MFMA.
#include "aloam_device.hpp"
#include "aloam_trig.hpp"
#include "lm_device.hpp"
#include "odometry_kernels.hpp"

namespace aloam {

// markers in the generated code for tools/isa_phases.py (comments only: no instruction is emitted)
#define ALOAM_PHASE(name) asm volatile("; ##PHASE " name)
#ifdef ALOAM_PHASE_CLOCK   // debug builds: shader clock at numbered points of k_build_grids_fused (surf class), summed over the workgroups (tools/ab_check.py)
__device__ unsigned long long g_phase_clock_odo[2][32];
#define ALOAM_BG_CLOCK(slot) do { if (threadIdx.x == 0 && blockIdx.x == 1) { atomicAdd(&g_phase_clock_odo[0][slot], (unsigned long long)__builtin_readcyclecounter()); atomicAdd(&g_phase_clock_odo[1][slot], 1ull); } } while (0)
#else
#define ALOAM_BG_CLOCK(slot) do { } while (0)
#endif
// debug builds (-DALOAM_ASSOC_STATS): what the association waves actually do, summed over a run (tools/ab_check.py stats)
#ifdef ALOAM_ASSOC_STATS
__device__ unsigned long long g_assoc_stats[2][32];
#define ALOAM_STAT(cls, i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_assoc_stats[cls][i], (unsigned long long)(v)); } while (0)
#else
#define ALOAM_STAT(cls, i, v) do { } while (0)
#endif

// TransformToStart with DISTORTION 0: Identity.slerp(1, q) is exactly +-q (sign flips when w < 0), rotation in
// f64, result stored back to f32 (reference src/laserOdometry.cpp:111-129).
__device__ __forceinline__ float4 transform_to_start(const float4& p, const OdomState& st) {
  double q[4] = {st.para_q[0], st.para_q[1], st.para_q[2], st.para_q[3]};
  if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  double o[3];
  quat_rotate(q, (double)p.x, (double)p.y, (double)p.z, o);
  return make_float4((float)(o[0] + st.para_t[0]), (float)(o[1] + st.para_t[1]), (float)(o[2] + st.para_t[2]), p.w);
}

// ---- DISTORTION 1 (reference src/laserOdometry.cpp:59 ships 0) ---------------------------------------------------------
// Interpolation ratio of a point: (intensity - int(intensity)) / SCAN_PERIOD, an f32 difference divided by the double 0.1
// (:115-116, :376-377, :474-475).
__device__ __forceinline__ double interpolation_ratio(float frac) { return (double)frac / 0.1; }

// Identity.slerp(s, q) as Eigen's QuaternionBase::slerp evaluates it: the result is scale0 * Identity + scale1 * q (a
// coefficient blend, not re-normalised); both scales depend on q only through d = q.w, so their derivatives do too.
__device__ __forceinline__ void slerp_scales(double w, double s, double* c0, double* c1, double* dc0, double* dc1) {
  const double one = 1.0 - 2.220446049250313e-16;
  const double absD = fabs(w);
  if (absD >= one) { *c0 = 1.0 - s; *c1 = s; *dc0 = 0.0; *dc1 = 0.0; }
  else {
    // acos / sin / cos of aloam_trig.hpp: the same IEEE operations as the CPU side performs, so the scales — and with them the f32
    // query points and the correspondences — are bit-identical by construction, not merely to an ulp of the device libm
    const double theta = acos_port(absD), st = sin_port(theta), ct = cos_port(theta);
    const double a0 = (1.0 - s) * theta, a1 = s * theta;
    const double s0 = sin_port(a0), s1 = sin_port(a1);
    *c0 = s0 / st;
    *c1 = s1 / st;
    // d/dtheta of sin(k theta) / sin(theta), then d theta / d absD = -1 / sin(theta), d absD / d w = sign(w)
    const double g = (w < 0.0 ? 1.0 : -1.0) / st;
    *dc0 = ((1.0 - s) * cos_port(a0) * st - s0 * ct) / (st * st) * g;
    *dc1 = (s * cos_port(a1) * st - s1 * ct) / (st * st) * g;
  }
  if (w < 0.0) { *c1 = -*c1; *dc1 = -*dc1; }
}

// lp = slerp(I, q, s) * cp + s t (reference src/lidarFactor.hpp:27-32 / :79-84 and TransformToStart) and, if M is given, the
// 3x4 matrix d lp / d (qx, qy, qz, qw) that forward-mode autodiff of those lines produces.
__device__ __forceinline__ void deskew_point(const double q[4], const double t[3], double s, double vx, double vy, double vz,
                                             double lp[3], double (*M)[4]) {
  double c0, c1, dc0, dc1;
  slerp_scales(q[3], s, &c0, &c1, &dc0, &dc1);
  const double u[3] = {c1 * q[0], c1 * q[1], c1 * q[2]}, w = c0 + c1 * q[3];
  const double v[3] = {vx, vy, vz};
  // Eigen: uv = 2 u x v; result = v + w uv + u x uv
  const double uv[3] = {2.0 * (u[1] * v[2] - u[2] * v[1]), 2.0 * (u[2] * v[0] - u[0] * v[2]), 2.0 * (u[0] * v[1] - u[1] * v[0])};
  lp[0] = v[0] + w * uv[0] + (u[1] * uv[2] - u[2] * uv[1]) + s * t[0];
  lp[1] = v[1] + w * uv[1] + (u[2] * uv[0] - u[0] * uv[2]) + s * t[1];
  lp[2] = v[2] + w * uv[2] + (u[0] * uv[1] - u[1] * uv[0]) + s * t[2];
  if (!M) return;
  // d result / d u_k = 2 w (e_k x v) + e_k x uv + u x (2 e_k x v);   d result / d w = uv
  double Du[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double e[3] = {0.0, 0.0, 0.0};
    e[k] = 1.0;
    const double ev[3] = {2.0 * (e[1] * v[2] - e[2] * v[1]), 2.0 * (e[2] * v[0] - e[0] * v[2]), 2.0 * (e[0] * v[1] - e[1] * v[0])};
    Du[0][k] = w * ev[0] + (e[1] * uv[2] - e[2] * uv[1]) + (u[1] * ev[2] - u[2] * ev[1]);
    Du[1][k] = w * ev[1] + (e[2] * uv[0] - e[0] * uv[2]) + (u[2] * ev[0] - u[0] * ev[2]);
    Du[2][k] = w * ev[2] + (e[0] * uv[1] - e[1] * uv[0]) + (u[0] * ev[1] - u[1] * ev[0]);
  }
  const double dw = dc0 + dc1 * q[3] + c1;                                   // d w / d qw
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    M[r][0] = c1 * Du[r][0];
    M[r][1] = c1 * Du[r][1];
    M[r][2] = c1 * Du[r][2];
    M[r][3] = dc1 * (Du[r][0] * q[0] + Du[r][1] * q[1] + Du[r][2] * q[2]) + uv[r] * dw;
  }
}

// Row of the residual Jacobian in the tangent space Ceres solves in: (d r / d lp) (d lp / d q) Plus'(q), and d lp / d t = s I.
__device__ __forceinline__ void deskew_jacobian_row(const double a[3], const double (*M)[4], const double q[4], double s, double J[6]) {
  double g[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = a[0] * M[0][j] + a[1] * M[1][j] + a[2] * M[2][j];
  // EigenQuaternionParameterization Plus Jacobian at delta = 0 (rows x, y, z, w)
  J[0] = g[0] * q[3] - g[1] * q[2] + g[2] * q[1] - g[3] * q[0];
  J[1] = g[0] * q[2] + g[1] * q[3] - g[2] * q[0] - g[3] * q[1];
  J[2] = -g[0] * q[1] + g[1] * q[0] + g[2] * q[3] - g[3] * q[2];
  J[3] = a[0] * s; J[4] = a[1] * s; J[5] = a[2] * s;
}

// -------------------------------------------------------------------------------------------------------
// Spatial hash grids over the "last" clouds (stand-in for pcl::KdTreeFLANN::setInputCloud, reference
// src/laserOdometry.cpp:567-568).  Two grids per cloud, both built by one 1024-thread workgroup per (sequence, cloud)
// with LDS counting sort (count -> exclusive scan -> fill):
//   G3: key (ix, iy, iz)      cell kCell3 + a coarse level — exact 1-NN by expanding cubic shells
//   G2: key (ix, iy, ringkey) cell kCell2                  — the ring-adjacent second / third neighbour search
// A grid entry is {x, y, z, bits(idx | (ringkey + 1) << 20)}: one 16-B load per candidate.  Cell sizes have few
// mantissa bits (0.5, 0.75, 2, 3, 2.625 = 21/8), so cell borders are exact f32 values; skipping a cell by its distance still leaves a margin.
// Also per cloud: flags[1] = the cloud is NOT ring-sorted (ring key = int(intensity) never decreasing with the index, the
// way scan registration emits it).  On ring-sorted clouds the reference's walk-until-break loops visit exactly the points
// whose key lies within +-2 of the closest point's; clouds that are not sorted (possible through aloam_set_last) take the
// literal walks, and clouds with huge coordinates or keys (flags[0]) the literal brute-force search as well.
constexpr float kCell3Surf = 0.5f;   // values with few mantissa bits only: cell borders must be exact f32 numbers
constexpr float kCell3Corner = 0.75f;   // measured (k_associate[corner], two launches, batch 1024): 1.25 m 1.13 ms, 1.0 m 1.106, 0.75 m 1.048, 0.625 m 1.074, 0.5 m 1.159; the planar class: 0.375 m 2.90 ms, 0.5 m 2.84, 0.625 m 3.01
// A 1-NN query whose neighbour is not inside the first block of fine cells continues on cells four times as large, so the work
// of a far query is bounded by a few dozen bucket look-ups instead of growing with the cube of the radius.  The ring grid has
// one level of 2.625 m cells: its 3x3 block already settles 95 % of the searches that reach it (most never do: the fine 1-NN
// block answers them), the 5x5 block (two cells = 5.25 m) covers the whole DISTANCE_SQ_THRESHOLD = 5 m radius.  A finer ring
// level was measured: it saves the association 0.1 ms per step and costs the grid build 0.2 ms.
constexpr float kCell3CoarseFactor = 4.0f, kCell2 = 2.625f;
constexpr unsigned kIdxMask = (1u << 20) - 1u;
__device__ __forceinline__ float cell3_of(int which) { return which == 0 ? kCell3Corner : kCell3Surf; }

__device__ __forceinline__ unsigned hash3(int a, int b, int c) {
  return ((unsigned)a * 73856093u) ^ ((unsigned)b * 19349663u) ^ ((unsigned)c * 83492791u);
}

// ---- the same three grids of a cloud from TWO reads of it instead of six ---------------------------------------------------------
// One 1024-thread workgroup per (sequence, cloud) keeps all three bucket tables in LDS at once — 16-bit counters, two to a word,
// which is what makes 3 x 16384 buckets fit (96 KiB) — so the cloud is read once to count and once to fill, and the three
// scattered 16-byte copies are written from the same registers.  16-bit offsets hold clouds of up to 65535 points (every synthetic
// and KITTI-sized HDL-64 cloud); larger clouds (128-ring stress input) and tables above 16384 buckets keep the per-grid workgroups
// of k_build_grids.  Traffic per point: 2 x 16 B read + 3 x 16 B written, against 6 x 16 + 3 x 16.
constexpr int kFusedMaxN = 65535, kFusedMaxH = 16384;
constexpr int kWalkKeys = kMaxRings + 8;

// Clouds whose ring keys are ALMOST ascending.  int(intensity) is the reference's ring id (src/laserOdometry.cpp:308,398), intensity = scanID + 0.1 relTime
// (src/scanRegistration.cpp:239), and relTime is slightly negative for the points of a ring that lie before the azimuth of the sweep's FIRST point
// (:211-214): as soon as that first ray had no return (any real sweep; every synthetic one with a range limit), a few points of ring r carry the key
// r - 1 in the middle of ring r's stretch.  The reference's walks (:315-361, :410-455) go up from the closest point until a key exceeds closest + 2 and
// down until one falls below closest - 2.  If no key is more than 2 below an EARLIER key, the first key >= c + 3 above the closest point is the first
// one in the whole cloud and the last key <= c - 3 below it the last one in the whole cloud, so the walks still visit ONE index range, whose ends
// depend on c alone: first[c + 3] and last[c - 3], two tables of R + 8 entries.  The pair kernel then takes "index inside the range" for "key within
// +-2" and the direction-dependent class rules for "same ring / other ring" (consider2<.., true>), exactly; only clouds with deeper descents
// (possible through aloam_set_last) keep the literal one-query walks.   s_flag[1]: 1 = nearly sorted (tables written), 2 = not.
__device__ __forceinline__ void walk_tables(const float4* __restrict__ pts, int n, int R, int* __restrict__ walk, int* s_walk, int* s_flag, int tid) {
  const int S = R + 8, lane = tid & 63;
  int* s_first = s_walk;
  int* s_last = s_walk + kWalkKeys;
  for (int k = tid; k < kWalkKeys; k += 1024) { s_first[k] = 0x7fffffff; s_last[k] = -1; }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {              // consecutive lanes hold consecutive points: only the ends of a run of equal keys touch the LDS
    const int i = base + tid;
    const int key = i < n ? (int)pts[i].w : -1;
    const int prev = __shfl_up(key, 1, 64), next = __shfl_down(key, 1, 64);
    if (key >= 0 && key < S) {
      if (lane == 0 || prev != key) atomicMin(&s_first[key], i);
      if (lane == 63 || next != key) atomicMax(&s_last[key], i);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int run = n;
    for (int k = S - 1; k >= 0; --k) { run = min(run, s_first[k]); s_first[k] = run; walk[k] = run; }               // first index with key >= k (n: none)
    run = -1;
    for (int k = 0; k < S; ++k) { run = max(run, s_last[k]); s_last[k] = run; walk[S + k] = run; }                  // last index with key <= k (-1: none)
    bool nearly = true;
    for (int k = 3; k < S; ++k) nearly = nearly && s_last[k - 3] < s_first[k];                                      // no key <= k - 3 after a key >= k
    s_flag[1] = nearly ? 1 : 2;
  }
  __syncthreads();
}
__host__ __device__ __forceinline__ bool fused_takes(int n, int H) { return n <= kFusedMaxN && H <= kFusedMaxH; }

__global__ __launch_bounds__(1024) void k_build_grids_fused(OdomArgs a) {
  constexpr int U = 4;
  const int b = blockIdx.y, which = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SeqMeta m = a.meta[b];
  const int n = which == 0 ? m.n_corner_last : m.n_surf_last;
  const float4* pts = which == 0 ? a.corner_last + (long long)b * a.R * 120 : a.surf_last + (long long)b * a.cap;
  const GridView g = grid_view(a, b, which);
  const int H = g.H;
  if (!fused_takes(n, H)) return;
  extern __shared__ __attribute__((aligned(16))) int lds[];
  unsigned* tab = reinterpret_cast<unsigned*>(lds);     // [3][H / 2]: bucket h of table t lives in half (h & 1) of word t * H / 2 + h / 2
  int* s_part = lds + 3 * (H / 2);                      // [3][16] wave totals of the scans
  int* s_flag = s_part + 48;                            // bad, unsorted
  int* s_walk = s_flag + 8;                             // [2][kWalkKeys] first / last index of every ring key (clouds with a descending key only)
  int* const starts[3] = {g.start3, g.start3c, g.start2};
  if (n == 0) {
    for (int t = 0; t < 3; ++t) for (int h = tid; h <= H; h += 1024) starts[t][h] = 0;
    if (tid < 3) g.flags[tid] = 0;
    return;
  }
  ALOAM_BG_CLOCK(0);
  for (int w = tid; w < 3 * (H / 2); w += 1024) tab[w] = 0u;
  if (tid < 2) s_flag[tid] = 0;
  __syncthreads();
  ALOAM_BG_CLOCK(1);
  const float inv0 = 1.0f / cell3_of(which), inv1 = 1.0f / (cell3_of(which) * kCell3CoarseFactor), inv2 = 1.0f / kCell2;
  const unsigned hm = (unsigned)(H - 1);
  auto buckets = [&](const float4& p, int key, unsigned* h) {
    h[0] = hash3((int)floorf(p.x * inv0), (int)floorf(p.y * inv0), (int)floorf(p.z * inv0)) & hm;
    h[1] = hash3((int)floorf(p.x * inv1), (int)floorf(p.y * inv1), (int)floorf(p.z * inv1)) & hm;
    h[2] = hash3((int)floorf(p.x * inv2), (int)floorf(p.y * inv2), key) & hm;
  };
  auto fetch = [&](int base, float4* p) {
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = base + u * 1024 + tid; p[u] = pts[i < n ? i : n - 1]; }
  };
  // ---- count (+ the sanity / ring-sorted flags of the cloud)
  {
    int bad = 0, unsorted = 0;
    for (int base = 0; base < n; base += U * 1024) {
      float4 p[U];
      float pw[U];
      fetch(base, p);
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = base + u * 1024 + tid; pw[u] = pts[i < n ? (i > 0 ? i - 1 : 0) : n - 1].w; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (base + u * 1024 + tid >= n) continue;
        const int key = (int)p[u].w;
        if (key < 0 || key > a.R || !(fabsf(p[u].x) < 4096.f && fabsf(p[u].y) < 4096.f && fabsf(p[u].z) < 4096.f)) bad = 1;
        if (key < (int)pw[u]) unsorted = 1;
        unsigned h[3];
        buckets(p[u], key, h);
#pragma unroll
        for (int t = 0; t < 3; ++t) atomicAdd(&tab[t * (H / 2) + (h[t] >> 1)], 1u << ((h[t] & 1u) * 16));
      }
    }
    if (bad) atomicOr(&s_flag[0], 1);
    if (unsorted) atomicOr(&s_flag[1], 1);
  }
  __syncthreads();
  ALOAM_BG_CLOCK(2);
  if (s_flag[1] && !s_flag[0]) walk_tables(pts, n, a.R, g.walk, s_walk, s_flag, tid);    // some key is lower than its predecessor's: nearly sorted or not at all?
  if (tid < 3) g.flags[tid] = tid == 2 ? 0 : s_flag[tid];
  ALOAM_BG_CLOCK(3);
  // ---- exclusive scans of the three tables: every thread owns H / 1024 consecutive buckets (= H / 2048 words) of each
  {
    const int wpt = H / 2048;                                              // words per thread and table (2 at H = 4096, 8 at 16384)
    int local[3], inc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      local[t] = 0;
      for (int k = 0; k < wpt; ++k) { const unsigned w = tab[t * (H / 2) + tid * wpt + k]; local[t] += (int)(w & 0xffffu) + (int)(w >> 16); }
      inc[t] = wave_scan_i32<false>(local[t]);
      if (lane == 63) s_part[t * 16 + wave] = inc[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      int run = inc[t] - local[t];
#pragma unroll 1
      for (int w = 0; w < wave; ++w) run += s_part[t * 16 + w];
      int* st = starts[t] + tid * wpt * 2;
      for (int k = 0; k < wpt; ++k) {
        const unsigned w = tab[t * (H / 2) + tid * wpt + k];
        const int c0 = (int)(w & 0xffffu), c1 = (int)(w >> 16);
        st[2 * k] = run; st[2 * k + 1] = run + c0;
        tab[t * (H / 2) + tid * wpt + k] = (unsigned)run | ((unsigned)(run + c0) << 16);   // running offsets (<= n <= 65535)
        run += c0 + c1;
      }
      if (tid == 1023) starts[t][H] = run;
    }
  }
  __syncthreads();
  ALOAM_BG_CLOCK(4);
  // ---- fill: the three copies of every point from one read.  (Measured in round 6, phase clock: count 90 k cycles, scans 26 k, fill 360 k of a surf
  // workgroup's 480 k.  The fill is bound by what it writes - 3 x 16 B per point to scattered places, ~2.9 TB/s over the chip while it runs - not by the LDS
  // atomics: handing the positions out per RUN of equal buckets, one atomic per run, changed nothing: 1.392 / 1.522 against 1.363 / 1.485 ms.)
  for (int base = 0; base < n; base += U * 1024) {
    float4 p[U];
    fetch(base, p);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 1024 + tid;
      if (i >= n) continue;
      const int key = (int)p[u].w;
      unsigned h[3];
      buckets(p[u], key, h);
      unsigned pos[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned sh = (h[t] & 1u) * 16;
        pos[t] = (atomicAdd(&tab[t * (H / 2) + (h[t] >> 1)], 1u << sh) >> sh) & 0xffffu;
      }
      const float4 e = make_float4(p[u].x, p[u].y, p[u].z, __uint_as_float((unsigned)i | ((unsigned)(key + 1) << 20)));
      g.sorted3[pos[0]] = e;
      g.sorted3c[pos[1]] = e;
      g.sorted2[pos[2]] = e;
    }
  }
  ALOAM_BG_CLOCK(5);
#ifdef ALOAM_PHASE_CLOCK
  __syncthreads();
  ALOAM_BG_CLOCK(6);
#endif
}

constexpr int kBgWaves = 4;   // waves per SIMD the register budget is sized for
constexpr int kBgUnroll = 4;
__global__ __launch_bounds__(1024, kBgWaves) void k_build_grids(OdomArgs a) {
  constexpr int U = kBgUnroll;                                     // loads in flight per thread
  // one workgroup per (sequence, cloud, grid): the three grids of a cloud are independent, and six workgroups per sequence
  // overlap each other's load / LDS-atomic / scattered-store phases better than two that run three passes back to back
  const int b = blockIdx.y, which = blockIdx.x / 3, tid = threadIdx.x;      // which: 0 corner_last, 1 surf_last
  const int pass_lo = blockIdx.x % 3, pass_hi = pass_lo + 1;
  const SeqMeta m = a.meta[b];
  const int n = which == 0 ? m.n_corner_last : m.n_surf_last;
  const float4* pts = which == 0 ? a.corner_last + (long long)b * a.R * 120 : a.surf_last + (long long)b * a.cap;
  const GridView g = grid_view(a, b, which);
  if (fused_takes(n, g.H)) return;        // k_build_grids_fused built this cloud's grids already
  extern __shared__ __attribute__((aligned(16))) int lds[];
  int* cnt = lds;                         // [H]
  int* part = lds + g.H;                  // [1024]
  int* s_flag = part + 1024;              // bad, unsorted
  if (tid < 2) s_flag[tid] = n >= (1 << 20) && tid == 0 ? 1 : 0;
  if (n == 0) {
    for (int pass = pass_lo; pass < pass_hi; ++pass) { int* st = pass == 0 ? g.start3 : pass == 1 ? g.start3c : g.start2; for (int h = tid; h <= g.H; h += 1024) st[h] = 0; }
    if (tid < 3 && pass_lo == 0) g.flags[tid] = 0;
    return;
  }
  // The loops below keep the loads of the next round in flight while the current one is binned: on this ISA a wait for
  // loaded data also waits for every older store, so loads are always issued ahead of the stores they must not wait for.
  auto fetch = [&](int base, float4* p) {
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = base + u * 1024 + tid; p[u] = pts[i < n ? i : n - 1]; }
  };
  for (int pass = pass_lo; pass < pass_hi; ++pass) {          // 0: G3, 1: G3 coarse, 2: G2
    const bool g3 = pass < 2;
    const float cell = pass == 0 ? cell3_of(which) : pass == 1 ? cell3_of(which) * kCell3CoarseFactor : kCell2;
    const float inv = 1.0f / cell;
    int* start = pass == 0 ? g.start3 : pass == 1 ? g.start3c : g.start2;
    float4* sorted = pass == 0 ? g.sorted3 : pass == 1 ? g.sorted3c : g.sorted2;
    __syncthreads();
    for (int h = tid; h < g.H; h += 1024) cnt[h] = 0;
    __syncthreads();
    float4 p[U], pn[U];
    float pw[U], pwn[U];                          // pass 0: intensity of the point before, for the ring-sorted test
    fetch(0, p);
    if (pass == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = u * 1024 + tid; pw[u] = pts[i < n ? (i > 0 ? i - 1 : 0) : n - 1].w; }
    }
    int bad = 0, unsorted = 0;
    for (int base = 0; base < n; base += U * 1024) {
      if (base + U * 1024 < n) {
        fetch(base + U * 1024, pn);
        if (pass == 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) { const int i = base + U * 1024 + u * 1024 + tid; pwn[u] = pts[i < n ? i - 1 : n - 1].w; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (base + u * 1024 + tid >= n) continue;
        const int key = (int)p[u].w;
        if (pass == 0) {
          // sanity: ring keys 0..R and coordinates the cell arithmetic is exact for; ring-sorted: keys never decrease
          if (key < 0 || key > a.R || !(fabsf(p[u].x) < 4096.f && fabsf(p[u].y) < 4096.f && fabsf(p[u].z) < 4096.f)) bad = 1;
          if (key < (int)pw[u]) unsorted = 1;
        }
        const int ix = (int)floorf(p[u].x * inv), iy = (int)floorf(p[u].y * inv);
        const int iz = g3 ? (int)floorf(p[u].z * inv) : key;
        atomicAdd(&cnt[hash3(ix, iy, iz) & (g.H - 1)], 1);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { p[u] = pn[u]; pw[u] = pwn[u]; }
    }
    if (pass == 0) {
      if (bad) atomicOr(&s_flag[0], 1);
      if (unsorted) atomicOr(&s_flag[1], 1);
    }
    __syncthreads();
    if (pass == 0 && tid < 3) g.flags[tid] = tid == 2 ? 0 : (tid == 1 ? 2 * s_flag[1] : s_flag[0]);   // (clouds beyond the fused build: any descending key -> literal walks)
    // exclusive scan of cnt[H]: per-thread run of H/1024 consecutive buckets + scan of the 1024 partial sums
    const int per = g.H / 1024;
    int local = 0;
    for (int k = 0; k < per; ++k) local += cnt[tid * per + k];
    part[tid] = local;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int v = tid >= d ? part[tid - d] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    int run = part[tid] - local;
    for (int k = 0; k < per; ++k) { const int c = cnt[tid * per + k]; cnt[tid * per + k] = run; start[tid * per + k] = run; run += c; }
    if (tid == 1023) start[g.H] = run;
    __syncthreads();
    fetch(0, p);
    for (int base = 0; base < n; base += U * 1024) {
      if (base + U * 1024 < n) fetch(base + U * 1024, pn);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 1024 + tid;
        if (i >= n) continue;
        const int key = (int)p[u].w;
        const int ix = (int)floorf(p[u].x * inv), iy = (int)floorf(p[u].y * inv);
        const int iz = g3 ? (int)floorf(p[u].z * inv) : key;
        const int pos = atomicAdd(&cnt[hash3(ix, iy, iz) & (g.H - 1)], 1);
        sorted[pos] = make_float4(p[u].x, p[u].y, p[u].z, __uint_as_float((unsigned)i | ((unsigned)(key + 1) << 20)));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = pn[u];
    }
  }
}

__device__ __forceinline__ float walk_dist(const float4& p, const float4& sel) {     // f32 expression (:322-327)
  return (p.x - sel.x) * (p.x - sel.x) + (p.y - sel.y) * (p.y - sel.y) + (p.z - sel.z) * (p.z - sel.z);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  return wave_min_packed(v);                                               // DPP ladder on the distance word, tie-break only on exact ties
}

// ---- shell enumeration -------------------------------------------------------------------------------------
// Square ring of Chebyshev radius r >= 1 in 2-D: 8 r cells, t in [0, 8r).
__device__ __forceinline__ void ring2d(int r, int t, int* dx, int* dy) {
  const int w = 2 * r + 1;
  if (t < w) { *dx = -r + t; *dy = -r; }
  else if (t < 2 * w) { *dx = -r + (t - w); *dy = r; }
  else { const int u = t - 2 * w, side = u / (w - 2); *dy = -r + 1 + u % (w - 2); *dx = side ? r : -r; }
}
// Cubic shell of Chebyshev radius r >= 1 in 3-D: 24 r^2 + 2 cells.
__device__ __forceinline__ void shell3d(int r, int c, int* dx, int* dy, int* dz) {
  const int w = 2 * r + 1, nface = w * w;
  if (c < 2 * nface) { const int f = c / nface, q = c % nface; *dz = f ? r : -r; *dy = q / w - r; *dx = q % w - r; }
  else { const int q = c - 2 * nface; *dz = -r + 1 + q / (8 * r); ring2d(r, q % (8 * r), dx, dy); }
}

// Load-balanced sweep over the buckets the lanes looked up: lane L owns bucket [s0, s0 + cnt); all 64 lanes then share
// the concatenated candidate list, so every candidate costs one independent 16-B load instead of a per-lane serial chain.
// Position i belongs to the last lane whose exclusive prefix is <= i: the owners drop their lane id at their first position
// in a 64-entry LDS row of the wave and an inclusive max-scan spreads it (prefix and spread run on the DPP network; the only
// LDS round trips per 64 candidates are that row and the two bpermutes that fetch the owner's bucket).
// What a lane still holds of a sweep that needed a single round (<= kSweep * 64 candidates): the caller can look at the
// same candidates again under another criterion without touching memory.
// kSweep = independent 16-B loads in flight per lane: 3 for the planar class (its fine blocks often hold 130-190 candidates),
// 2 for the corner class (measured: a third row only costs there).
template <int kSweep> struct Kept { float4 p[kSweep]; bool a[kSweep]; bool ok; };

template <int kSweep, class F>
__device__ __forceinline__ void wave_sweep(const float4* __restrict__ sorted, int s0, int cnt, int lane, int* row, F&& f, Kept<kSweep>* keep = nullptr) {
  const int incl = wave_scan_i32<false>(cnt);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - cnt;
  int carry = 0;                                                           // owner + 1 of the position before `base`
  for (int base = 0; base < total; base += kSweep * 64) {
    const int slot = excl - base;
    const int rows = total - base > (kSweep - 1) * 64 ? kSweep : (total - base + 63) >> 6;   // uniform: rows in use this round
#pragma unroll
    for (int u = 0; u < kSweep; ++u) if (u < rows) row[u * 64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (cnt > 0 && slot >= 0 && slot < rows * 64) row[slot] = lane + 1;
    __builtin_amdgcn_wave_barrier();
    int own[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) own[u] = u < rows ? row[u * 64 + lane] : 0;
    __builtin_amdgcn_wave_barrier();
    float4 p[kSweep];
#pragma unroll
    for (int u = 0; u < kSweep; ++u) {
      p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < rows) {
        if (lane == 0 && carry > own[u]) own[u] = carry;
        own[u] = wave_scan_i32<true>(own[u]);
        carry = __builtin_amdgcn_readlane(own[u], 63);
        const int i = base + u * 64 + lane;
        const int os = __shfl(s0, own[u] - 1, 64), oe = __shfl(excl, own[u] - 1, 64);
        const int at = i < total ? os + (i - oe) : 0;
        p[u] = sorted[at];
      }
    }
#pragma unroll
    for (int u = 0; u < kSweep; ++u) {
      const bool act = u < rows && base + u * 64 + lane < total;
      if (keep) { keep->p[u] = p[u]; keep->a[u] = act; }
      if (act) f(p[u]);
    }
  }
  if (keep) keep->ok = total <= kSweep * 64;
}

// Distance from coordinate s to the cell [c * cell, (c + 1) * cell) along one axis (0 inside).  Points are
// binned with floorf(p / cell); the gap is shortened by 1 mm and callers leave another 0.1 % on the squared distance, so a
// skipped cell provably holds nothing that could tie with or beat the current best.
__device__ __forceinline__ float cell_gap(float s, int c, float cell) {
  const float lo = (float)c * cell, hi = lo + cell;
  const float gap = s < lo ? lo - s : (s > hi ? s - hi : 0.f);
  return gap > 1e-3f ? gap - 1e-3f : 0.f;                                 // 1 mm: far above the rounding of p / cell for |p| < 4096
}

// A lane's best candidate so far: packed (f32 distance bits << 32 | tie-break order) and the point it belongs to.  After the
// wave-wide minimum the lane whose own value equals it still holds the winner's coordinates and stores them itself, so no
// index has to be chased through the cloud afterwards.
struct Track { unsigned long long v; float x, y, z; };
__device__ __forceinline__ void track_take(Track& t, bool ok, unsigned long long v, const float4& p) {
  const bool k = ok && v < t.v;                                             // selects, not branches: the trackers stay in registers
  t.v = k ? v : t.v; t.x = k ? p.x : t.x; t.y = k ? p.y : t.y; t.z = k ? p.z : t.z;
}

// Exact 1-NN of `sel` (pcl::KdTreeFLANN::nearestKSearch(k = 1), reference src/laserOdometry.cpp:302,390) by one wave.
// Returns packed (f32 distance bits << 32 | index << 12 | ring key + 1), ~0 if nothing was found; `mine` is this lane's
// share.  Distances >= 25 are not needed by the caller (DISTANCE_SQ_THRESHOLD), so the grid search stops once every
// unvisited point is provably >= 5 m away.
template <int kSweep>
__device__ __forceinline__ unsigned long long wave_nn(const GridView& g, bool bad, int which, const float4* pts, int n, const float4& sel, int lane, int* row, Track& mine, Kept<kSweep>& kept) {
  unsigned long long best = ~0ull;
  auto visit = [&](const float4& p) {
    const float ddx = p.x - sel.x, ddy = p.y - sel.y, ddz = p.z - sel.z;
    const float d = (ddx * ddx + ddy * ddy) + ddz * ddz;
    const unsigned wb = __float_as_uint(p.w);
    track_take(mine, true, ((unsigned long long)__float_as_uint(d) << 32) | ((wb & kIdxMask) << 12) | (wb >> 20), p);
  };
  if (!bad) {
    const float cell = cell3_of(which);
    {  // level 0: the 3x3x3 block of fine cells
      const float inv = 1.0f / cell;
      const int cx = (int)floorf(sel.x * inv), cy = (int)floorf(sel.y * inv), cz = (int)floorf(sel.z * inv);
      int s0 = 0, cnt = 0;
      if (lane < 27) {
        const unsigned h = hash3(cx + lane % 3 - 1, cy + (lane % 9) / 3 - 1, cz + lane / 9 - 1) & (unsigned)(g.H - 1);
        s0 = g.start3[h];
        cnt = g.start3[h + 1] - s0;
      }
      wave_sweep<kSweep>(g.sorted3, s0, cnt, lane, row, visit, &kept);
      best = wave_min_u64(mine.v);
      const float bound = (1.0f - 0.01f) * cell;                       // every unvisited point is farther than `bound`
      if (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) <= bound * bound) return best;
    }
    // level 1: expanding cubic shells of coarse cells (at most three steps reach DISTANCE_SQ_THRESHOLD)
    const float cellc = cell * kCell3CoarseFactor, invc = 1.0f / cellc;
    const int ux = (int)floorf(sel.x * invc), uy = (int)floorf(sel.y * invc), uz = (int)floorf(sel.z * invc);
    for (int r = 1;; ++r) {
      const int ncell = r == 1 ? 27 : 24 * r * r + 2;
      const float limit = best != ~0ull ? fminf(__uint_as_float((unsigned)(best >> 32)), 25.0f) : 25.0f;
      for (int cb = 0; cb < ncell; cb += 64) {
        const int c = cb + lane;
        int s0 = 0, cnt = 0;
        if (c < ncell) {
          int dx, dy, dz;
          if (r == 1) { dz = c / 9 - 1; dy = (c % 9) / 3 - 1; dx = c % 3 - 1; }
          else shell3d(r, c, &dx, &dy, &dz);
          // a cell farther away than the best candidate so far (or than DISTANCE_SQ_THRESHOLD) cannot change the answer
          const float gx = cell_gap(sel.x, ux + dx, cellc), gy = cell_gap(sel.y, uy + dy, cellc), gz = cell_gap(sel.z, uz + dz, cellc);
          if (((gx * gx + gy * gy) + gz * gz) * 0.999f <= limit) {
            const unsigned h = hash3(ux + dx, uy + dy, uz + dz) & (unsigned)(g.H - 1);
            s0 = g.start3c[h];
            cnt = g.start3c[h + 1] - s0;
          }
        }
        wave_sweep<kSweep>(g.sorted3c, s0, cnt, lane, row, visit);
      }
      best = wave_min_u64(mine.v);
      const float bc = ((float)r - 0.01f) * cellc, b2 = bc * bc;
      if (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) <= b2) break;
      if (b2 >= 25.0f) break;                                         // nothing within DISTANCE_SQ_THRESHOLD is left
    }
    return best;
  }
  for (int base = 0; base < n; base += 64) {                      // literal brute force (flagged clouds only)
    const int j = base + lane;
    if (j < n) {
      const float4 p = pts[j];
      visit(make_float4(p.x, p.y, p.z, __uint_as_float((unsigned)j)));
    }
  }
  return wave_min_u64(mine.v);
}

// -------------------------------------------------------------------------------------------------------
// TransformToStart of every corner / planar feature of the current sweep with the current pose estimate (reference
// src/laserOdometry.cpp:111-129, called at :300 and :388), one lane per feature, once per outer iteration.  The association
// waves then load the transformed point instead of every one of their 64 lanes redoing the same f64 rotation.
template <bool DISTORT>
__global__ __launch_bounds__(256) void k_transform_queries(OdomArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const SeqMeta m = a.meta[b];
  if (i >= m.n_sharp + m.n_flat) return;
  const bool plane = i >= m.n_sharp;
  const int qi = plane ? i - m.n_sharp : i;
  const long long o = plane ? (long long)b * a.R * 24 + qi : (long long)b * a.R * 12 + qi;
  const float4 raw = plane ? a.flat[o] : a.sharp[o];
  float4 sel;
  if (DISTORT) {
    const OdomState& st = a.state[b];
    const double q[4] = {st.para_q[0], st.para_q[1], st.para_q[2], st.para_q[3]}, t[3] = {st.para_t[0], st.para_t[1], st.para_t[2]};
    const float frac = raw.w - (float)(int)raw.w;                            // relTime of the point (:116)
    double lp[3];
    deskew_point(q, t, interpolation_ratio(frac), (double)raw.x, (double)raw.y, (double)raw.z, lp, nullptr);
    sel = make_float4((float)lp[0], (float)lp[1], (float)lp[2], raw.w);
  } else sel = transform_to_start(raw, a.state[b]);
  (plane ? a.sel_flat : a.sel_sharp)[o] = sel;
}

// -------------------------------------------------------------------------------------------------------
// Data association for one feature class, one wave per query (4 queries per workgroup):
//   PLANE = false: corner features  (reference src/laserOdometry.cpp:299-384)  -> EdgeRec
//   PLANE = true : planar features  (reference src/laserOdometry.cpp:387-483)  -> PlaneRec
// Candidates of the ring walk are ordered exactly as the reference visits them (upward from closest+1, then
// downward from closest-1); "first strictly smaller wins" is the lexicographic (distance, visit order) minimum.
// The kernel is a chain of dependent memory round trips per query (bucket bounds -> bucket entries, twice), so everything
// that does not depend on a search result is loaded up front and nothing is fetched by index afterwards: grid entries
// carry their ring key, and the lanes that saw the winners store the record fields themselves.
// Launch geometry: one single-wave workgroup per query slot.  Measured at batch 512 (planar class, both launches): 8 waves per
// workgroup 1.90 ms, 4: 1.63, 2: 1.53, 1: 1.47 — a freed SIMD slot is refilled soonest when nothing else has to retire with it.
// Persistent waves (exactly the resident number, each walking through 1 / W of the queries of its XCD's sequences) were measured
// too: 1.84 ms — the loop carries 75 VGPRs instead of 64 (6 waves per SIMD instead of 8) and the kernel lives off occupancy.
constexpr int kAssocSweepWide = 6;   // rows of 64 candidates in flight per wave, planar class, sensors with more than 64 rings.  Measured on the 128 x 2048 stress sweeps (two launches, batch 1024): 3 rows 7.65 ms, 4: 7.70, 5: 7.14, 6: 6.64, 8: 8.69, 10: 11.8
constexpr int kAssocSweepPlane = 3;   // the same for sensors with up to 64 rings (HDL-64, batch 1024: 3 rows 2.83 ms, 4 rows 3.03 ms)
template <bool PLANE, bool WIDE> struct SweepRows { static constexpr int value = WIDE ? (PLANE ? kAssocSweepWide : 3) : (PLANE ? kAssocSweepPlane : 2); };

template <bool PLANE, bool WIDE>
__device__ __forceinline__ void associate_one(const OdomArgs& a, int b, int qi, const SeqMeta& m, const GridView& g, int lane, int* row) {
  constexpr int kSweep = SweepRows<PLANE, WIDE>::value;
  const int qcap = PLANE ? a.R * 24 : a.R * 12;
  const float4* Q = (PLANE ? a.flat : a.sharp) + (long long)b * qcap;
  const float4 raw = Q[qi];
  const float4 sel = ((PLANE ? a.sel_flat : a.sel_sharp) + (long long)b * qcap)[qi];   // k_transform_queries
  const bool bad = g.flags[0] != 0, unsorted = g.flags[1] != 0;
  const float frac = raw.w - (float)(int)raw.w;                              // relTime of the point (:116)
  const int nt = PLANE ? m.n_surf_last : m.n_corner_last;
  const float4* T = PLANE ? a.surf_last + (long long)b * a.cap : a.corner_last + (long long)b * a.R * 120;
  int valid = 0;
  Track t1 = {~0ull, 0.f, 0.f, 0.f}, t2 = t1, t3 = t1;
  Kept<kSweep> kept;
  kept.ok = false;
#pragma unroll
  for (int u = 0; u < kSweep; ++u) { kept.a[u] = false; kept.p[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
  unsigned long long best2 = ~0ull, best3 = ~0ull;
  const unsigned long long nn = nt > 0 ? wave_nn(g, bad, PLANE ? 1 : 0, T, nt, sel, lane, row, t1, kept) : ~0ull;
  const float nnd = __uint_as_float((unsigned)(nn >> 32));
  if (nn != ~0ull && (double)nnd < 25.0) {                            // DISTANCE_SQ_THRESHOLD (:65,305,393)
    const int closest = (int)((unsigned)nn >> 12);
    const int cid = bad ? (int)T[closest].w : (int)((unsigned)nn & 0xfffu) - 1;     // closestPointScanID (:308,398)
    auto consider = [&](const float4& p, int j, int key) {
      if (j == closest) return;
      const bool up = j > closest;
      bool c2, c3;
      if (PLANE) { c2 = up ? key <= cid : key >= cid; c3 = !c2; }     // :416-426, :444-454
      else { c2 = up ? key > cid : key < cid; c3 = false; }           // :315-316, :341-342 (`continue` on the same side)
      const float d = walk_dist(p, sel);
      if (!((double)d < 25.0)) return;
      const unsigned seq = up ? (unsigned)(j - closest) : 0x40000000u + (unsigned)(closest - j);
      const unsigned long long v = ((unsigned long long)__float_as_uint(d) << 32) | seq;
      track_take(t2, c2, v, p);
      if (PLANE) track_take(t3, c3, v, p);
    };
    if (!bad && !unsorted) {
      // window form: the walks visit exactly the points whose ring key lies in cid-2 .. cid+2 (minus the closest point)
      auto visit = [&](const float4& p) {
        const unsigned wb = __float_as_uint(p.w);
        const int j = (int)(wb & kIdxMask), pk = (int)(wb >> 20) - 1;
        if (pk >= cid - 2 && pk <= cid + 2) consider(p, j, pk);
      };
      // On a ring-sorted cloud the second neighbour of a planar feature is the nearest point of the closest point's own ring
      // and the third the nearest of the other rings; a corner feature's second neighbour comes from the other rings only.
      bool done2 = false, done3 = !PLANE;
      float lim2 = 25.0f, lim3 = 25.0f;
      if (kept.ok) {
        // The candidates of the fine 1-NN block are still in registers: every point within (almost) one cell of the query is
        // among them, so neighbours found there within that radius are final and the ring grid is not needed for them.
#pragma unroll
        for (int u = 0; u < kSweep; ++u) if (kept.a[u]) visit(kept.p[u]);
        best2 = wave_min_u64(t2.v);
        if (PLANE) best3 = wave_min_u64(t3.v);
        const float bound = (1.0f - 0.01f) * cell3_of(PLANE ? 1 : 0), b2 = bound * bound;
        if (best2 != ~0ull) lim2 = __uint_as_float((unsigned)(best2 >> 32));
        if (PLANE && best3 != ~0ull) lim3 = __uint_as_float((unsigned)(best3 >> 32));
        done2 = best2 != ~0ull && lim2 <= b2;
        done3 = !PLANE || (best3 != ~0ull && lim3 <= b2);
      }
      // ring grid: first the 3x3 block of cells around the query (5 ring keys each, one look-up per lane), then, only if a
      // neighbour may still lie farther than that block reaches, the 16 cells around it: two cells = 5.25 m cover
      // DISTANCE_SQ_THRESHOLD.  Cells that cannot hold anything closer than the class's best so far are skipped.
      for (int level = 0; level < 2 && !(done2 && done3); ++level) {
        const float cell = kCell2, inv = 1.0f / cell;
        const int cx = (int)floorf(sel.x * inv), cy = (int)floorf(sel.y * inv);
        const int nlook = level == 0 ? 45 : 80;
        for (int lb = 0; lb < nlook; lb += 64) {
          const int l = lb + lane;
          int s0 = 0, cnt = 0;
          if (l < nlook) {
            const int key = cid + l % 5 - 2, cc = l / 5;
            int ddx, ddy;
            if (level == 0) { ddx = cc % 3 - 1; ddy = cc / 3 - 1; }
            else ring2d(2, cc, &ddx, &ddy);
            const bool second = PLANE ? key == cid : key != cid;
            const bool wanted = second ? !done2 : (PLANE && !done3);
            const float gx = cell_gap(sel.x, cx + ddx, cell), gy = cell_gap(sel.y, cy + ddy, cell);
            if (key >= 0 && wanted && (gx * gx + gy * gy) * 0.999f <= (second ? lim2 : lim3)) {
              const unsigned h = hash3(cx + ddx, cy + ddy, key) & (unsigned)(g.H - 1);
              s0 = g.start2[h];
              cnt = g.start2[h + 1] - s0;
            }
          }
          wave_sweep<kSweep>(g.sorted2, s0, cnt, lane, row, visit);
        }
        best2 = wave_min_u64(t2.v);
        if (PLANE) best3 = wave_min_u64(t3.v);
        const float bound = ((float)(level + 1) - 0.01f) * cell, b2 = bound * bound;
        if (b2 >= 25.0f) break;
        if (best2 != ~0ull) lim2 = __uint_as_float((unsigned)(best2 >> 32));
        if (PLANE && best3 != ~0ull) lim3 = __uint_as_float((unsigned)(best3 >> 32));
        done2 = best2 != ~0ull && lim2 <= b2;
        done3 = !PLANE || (best3 != ~0ull && lim3 <= b2);
      }
    } else {
      // literal walks (:312-361 / :402-455) for clouds that are not ring-sorted
      for (int base = closest + 1; base < nt; base += 64) {
        const int j = base + lane;
        bool stop = false;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        int key = 0;
        if (j < nt) { p = T[j]; key = (int)p.w; stop = (double)key > (double)cid + 2.5; }       // NEARBY_SCAN (:66)
        const unsigned long long sm = __ballot(stop);
        const int first_stop = sm ? (__ffsll((long long)sm) - 1) : 64;
        if (j < nt && lane < first_stop) consider(p, j, key);
        if (sm) break;
      }
      for (int base = closest - 1; base >= 0; base -= 64) {
        const int j = base - lane;
        bool stop = false;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        int key = 0;
        if (j >= 0) { p = T[j]; key = (int)p.w; stop = (double)key < (double)cid - 2.5; }
        const unsigned long long sm = __ballot(stop);
        const int first_stop = sm ? (__ffsll((long long)sm) - 1) : 64;
        if (j >= 0 && lane < first_stop) consider(p, j, key);
        if (sm) break;
      }
      best2 = wave_min_u64(t2.v);
      if (PLANE) best3 = wave_min_u64(t3.v);
    }
    valid = best2 != ~0ull && (!PLANE || best3 != ~0ull);               // :363 / :457
  }
  // the record: lane 0 stores the raw, untransformed point (:365-367 / :460-462) and the flag, the lanes that saw the
  // winners store them (a point seen on two grid levels is stored twice with the same value)
  if (PLANE) {
    PlaneRec* e = a.planes + (long long)b * a.R * 24 + qi;
    if (lane == 0) {
      e->cp[0] = raw.x; e->cp[1] = raw.y; e->cp[2] = raw.z;
      e->valid = valid; e->pad[0] = __float_as_int(frac); e->pad[1] = e->pad[2] = 0;
      if (!valid) for (int k = 0; k < 3; ++k) e->j[k] = e->l[k] = e->m[k] = 0.f;
    }
    if (valid) {
      if (t1.v == nn) { e->j[0] = t1.x; e->j[1] = t1.y; e->j[2] = t1.z; }
      if (t2.v == best2) { e->l[0] = t2.x; e->l[1] = t2.y; e->l[2] = t2.z; }
      if (t3.v == best3) { e->m[0] = t3.x; e->m[1] = t3.y; e->m[2] = t3.z; }
    }
  } else {
    EdgeRec* e = a.edges + (long long)b * a.R * 12 + qi;
    if (lane == 0) {
      e->cp[0] = raw.x; e->cp[1] = raw.y; e->cp[2] = raw.z;
      e->valid = valid; e->pad[0] = __float_as_int(frac); e->pad[1] = 0;
      if (!valid) for (int k = 0; k < 3; ++k) e->a[k] = e->b[k] = 0.f;
    }
    if (valid) {
      if (t1.v == nn) { e->a[0] = t1.x; e->a[1] = t1.y; e->a[2] = t1.z; }
      if (t2.v == best2) { e->b[0] = t2.x; e->b[1] = t2.y; e->b[2] = t2.z; }
    }
  }
}


// -------------------------------------------------------------------------------------------------------
// TWO queries per wave (round 4).  The wave-per-query kernel above spends two thirds of its ~600 VALU instructions per query on
// work that does not depend on the lane: bucket hashes, scans, the owner spread of every row of candidates, five wave minima, the
// ring-grid look-ups.  Here a wave serves two consecutive queries of one sequence, one per HALF of 32 lanes: every one of those
// instructions now does its work for two queries, the per-candidate work (one lane = one candidate) is unchanged.  Scans and minima
// stop at the half border (the row_bcast:31 step of the DPP ladders is left out); what the one-query form keeps in scalar registers
// (closest point, its ring, the class bounds, "done") is half-uniform data in vector registers, broadcast from the last lane of the
// half with one ds_bpermute; early exits become per-half masks and the wave leaves a phase when neither half needs it.
// Trackers hold the packed key only: the winners' coordinates are read from the ring-ordered cloud by index at the very end (grid
// entries are copies of those points), and the candidates a half keeps from its fine block are (distance, index | ring) pairs, so six
// rows of 32 cost the registers three rows of 64 cost before.  Sequences whose clouds are flagged (not ring-sorted, coordinates
// or keys out of range) are left to k_associate_flagged, which runs the literal one-query code above.
template <bool MAX>
__device__ __forceinline__ int half_scan_i32(int v) {                         // inclusive scan inside each half of 32 lanes
  constexpr int identity = MAX ? (int)0x80000000 : 0;
  auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : a + b; };
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xF, 0xF, false));   // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xF, 0xF, false));   // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xF, 0xF, false));   // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xF, 0xF, false));   // row_shr:8
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  return v;
}
__device__ __forceinline__ unsigned half_min_ladder(unsigned v) {             // lanes 31 and 63 end up with the minimum of their half
  auto op = [](unsigned a, unsigned b) { return a < b ? a : b; };
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xF, 0xF, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xF, 0xF, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xF, 0xF, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xF, 0xF, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));
  return v;
}
// value of the last lane of the own half, for every lane (`last4` = (lane | 31) * 4)
__device__ __forceinline__ int half_last(int v, int last4) { return __builtin_amdgcn_ds_bpermute(last4, v); }
// minimum of packed (distance bits << 32 | tie-break) keys per half, in every lane of the half
__device__ __forceinline__ unsigned long long half_min_packed(unsigned long long v, int last4) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = (unsigned)half_last((int)half_min_ladder(hi), last4);
  const unsigned ml = (unsigned)half_last((int)half_min_ladder(hi == mh ? lo : 0xffffffffu), last4);
  return ((unsigned long long)mh << 32) | ml;
}

// ---- sweeps whose VALU cost per row of candidates is a few instructions --------------------------------------------------------
// The kernels are bound by the number of vector instructions a wave issues (SQ counters: 97 % of the VALU issue cycles busy), so the
// owner of a list position is found with LDS and scalar work instead of a max-scan on the DPP network:
//  * every non-empty bucket sets ONE bit, at the last position it owns, in a bit mask per row (ds_or); the number of buckets in front
//    of position i is then v_mbcnt of the row's mask plus a running popcount — its rank among the non-empty buckets;
//  * the non-empty buckets leave (first entry - first position) in a table at their rank (ballot + v_mbcnt), so a position needs one
//    LDS read and one addition to get the index of its grid entry;
//  * lanes beyond the end of the list do not idle behind a flag: their index is clamped to the cloud and they look at whatever grid
//    entry it names.  Every entry is a point of the cloud with its true index and ring, and every search here is a minimum over a SUPERSET
//    of the candidates it needs (the reference's walks look at the whole ring window), so extra real points cannot change a result.
// HALVES: two independent lists, one per half of 32 lanes (rows of 32 positions, side by side); otherwise one list, rows of 64.
// LDS of a wave: mark slots of four dwords {mask of lanes 0-31, mask of lanes 32-63, both, -} — HALVES: slot (row, half), the half's 32-bit
// mask in the dword that v_mbcnt reads for its lanes; else slot (row) — then table[72].
template <int kRows> struct KeptPair { unsigned long long k[kRows]; bool ok; };                 // packed 1-NN keys (distance, index << 12 | ring + 1)
template <int kRows> constexpr int sweep2_lds_ints() { return kRows * 8 + 72; }

typedef float float2v __attribute__((ext_vector_type(2)));
// bounds of bucket h: start[h] and start[h + 1] by ONE 8-byte load (4-byte aligned: the hardware takes it) at a 32-bit offset from the table
struct __attribute__((packed, aligned(4))) IntPair { int a, b; };
__device__ __forceinline__ void bucket_bounds(const int* __restrict__ start, unsigned h, int& s0, int& cnt) {
  const IntPair v = *reinterpret_cast<const IntPair*>(reinterpret_cast<const char*>(start) + (h << 2));
  s0 = v.a; cnt = v.b - v.a;
}
__device__ __forceinline__ float dist_to(const float4& p, float2v sxy, float sz) {
  const float2v pxy = {p.x, p.y};
  const float2v dxy = pxy - sxy, qxy = dxy * dxy;                             // packed f32: the same two subtractions and products
  const float ddz = p.z - sz;
  return (qxy.x + qxy.y) + ddz * ddz;                                         // FLANN's L2_Simple sum = the walk's f32 expression (:322-327)
}

template <bool HALVES, int kRows, class F>
__device__ __forceinline__ void sweep2(const float4* __restrict__ sorted, unsigned last_index, int s0, int cnt, int lane, int last4, int* lds, F&& f, bool* single_round = nullptr, int* rows_first = nullptr, int stat_cls = 0, int stat_slot = -1) {
  constexpr int W = HALVES ? 32 : 64, LOGW = HALVES ? 5 : 6, kSlots = HALVES ? 2 * kRows : kRows;
  static_assert(kRows <= 8, "the mark slots of a round are zeroed by one store of the wave: 4 * kSlots <= 64 lanes");
  unsigned* marks = reinterpret_cast<unsigned*>(lds);
  int* table = lds + kRows * 8;
  // A lane past the end of its half's list reads table[rank] of a rank no bucket owns: zero the table once per call, so that such a lane
  // looks at the (clamped) entry `position` of the cloud — a real point, a harmless extra candidate — and never at an uninitialised word.
  table[lane] = 0;
  if (lane < 8) table[64 + lane] = 0;
  const int h = HALVES ? lane >> 5 : 0, l = lane & (W - 1);
  const int incl = HALVES ? half_scan_i32<false>(cnt) : wave_scan_i32<false>(cnt);
  int total, tmax;
  if (HALVES) {
    total = half_last(incl, last4);
    const int t0 = __builtin_amdgcn_readlane(incl, 31), t1 = __builtin_amdgcn_readlane(incl, 63);
    tmax = t0 > t1 ? t0 : t1;
  } else total = tmax = __builtin_amdgcn_readlane(incl, 63);
  if (single_round) *single_round = total <= kRows * W;
  if (rows_first) *rows_first = tmax > (kRows - 1) * W ? kRows : (tmax + W - 1) >> LOGW;      // uniform: rows the first round fills
  if (tmax <= 0) return;
  const bool nonempty = cnt > 0;
  const unsigned long long ne = __ballot(nonempty);
#ifdef ALOAM_ASSOC_STATS
  if (stat_slot >= 0) { ALOAM_STAT(stat_cls, stat_slot, 1); ALOAM_STAT(stat_cls, stat_slot + 1, __builtin_popcountll(ne)); ALOAM_STAT(stat_cls, stat_slot + 2, HALVES ? __builtin_amdgcn_readlane(incl, 31) + __builtin_amdgcn_readlane(incl, 63) : tmax); ALOAM_STAT(stat_cls, stat_slot + 3, (tmax + W - 1) >> LOGW); }
#endif
  const int rank_own = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ne >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ne, 0u));
  if (nonempty) table[rank_own] = s0 - (incl - cnt);                          // entry index = position + this
  const int rank0 = HALVES && h ? __builtin_popcount((unsigned)ne) : 0;       // rank of the half's first non-empty bucket
  const int pl = incl - 1;                                                    // last position the bucket owns
  const char* const bytes = reinterpret_cast<const char*>(sorted);
  for (int base = 0; base < tmax; base += kRows * W) {
    const int rows = tmax - base > (kRows - 1) * W ? kRows : (tmax - base + W - 1) >> LOGW;   // uniform: rows in use this round
    if (lane < kSlots * 4) marks[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const int pr = pl - base;
    if (nonempty && pr >= 0 && pr < kRows * W) {
      const int r = pr >> LOGW, bit = pr & (W - 1);
      unsigned* slot = marks + (HALVES ? (r * 2 + h) * 4 : r * 4);
      atomicOr(slot + (HALVES ? h : bit >> 5), 1u << (bit & 31));
      if (HALVES) atomicOr(slot + 2, 1u << (bit & 31));
    }
    __builtin_amdgcn_wave_barrier();
    int carry = rank0;                                                        // rank of the owner of the first position of the row
    if (base > 0) {                                                           // buckets that ended in earlier rounds
      const unsigned long long before = __ballot(nonempty && pl < base);
      carry += HALVES ? (h ? __builtin_popcount((unsigned)(before >> 32)) : __builtin_popcount((unsigned)before)) : __builtin_popcountll(before);
    }
    float4 p[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < rows) {
        const unsigned* slot = marks + (HALVES ? (u * 2 + h) * 4 : u * 4);
        int rank;
        if (HALVES) {
          const uint3 m = *reinterpret_cast<const uint3*>(slot);
          rank = (int)__builtin_amdgcn_mbcnt_hi(m.y, __builtin_amdgcn_mbcnt_lo(m.x, (unsigned)carry));
          carry = __builtin_popcount(m.z) + carry;
        } else {
          const uint2 m = *reinterpret_cast<const uint2*>(slot);
          rank = (int)__builtin_amdgcn_mbcnt_hi(m.y, __builtin_amdgcn_mbcnt_lo(m.x, (unsigned)carry));
          carry = __builtin_popcount(m.x) + carry;
          carry = __builtin_popcount(m.y) + carry;
        }
        const unsigned at = (unsigned)(table[rank] + (base + u * W + l));
        p[u] = *reinterpret_cast<const float4*>(bytes + ((at < last_index ? at : last_index) << 4));
      }
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) if (u < rows) f(p[u], u, base == 0);
  }
}

constexpr int kPairRowsPlane = 6;   // rows of 32 candidates per half in flight / kept
constexpr int kPairRowsCorner = 4;
constexpr int kPairRowsWide = 8;
template <bool PLANE, bool WIDE> struct PairRows { static constexpr int value = WIDE ? (PLANE ? kPairRowsWide : kPairRowsPlane) : (PLANE ? kPairRowsPlane : kPairRowsCorner);
                                                   static_assert(value >= 1 && value <= 8, "kPairRows*: 1 .. 8 rows (sweep2's mark slots)"); };

__device__ __forceinline__ unsigned long long nn_key(float d, unsigned wb) { return ((unsigned long long)__float_as_uint(d) << 32) | ((wb & kIdxMask) << 12) | (wb >> 20); }
__device__ __forceinline__ void take_min(unsigned long long& t, unsigned long long v) { t = v < t ? v : t; }

// Second / third neighbour of the window form (ring-sorted clouds: j > closest implies key >= cid, so "key <= cid on the way up, key >= cid on
// the way down" (:416-426, :444-454) is key == cid, and the corner class's "key > cid up, key < cid down" (:315-316, :341-342) is key != cid).
// Visit order: upward from closest + 1, then downward from closest - 1; first strictly smaller distance wins = lexicographic (distance, order).
// NEARLY (clouds whose keys are almost ascending, walk_tables above): the walks visit the index range (lo, hi); on the way up a key <= cid is "same
// ring" for the planar class and skipped by the corner class, on the way down a key >= cid (:315-316,341-342 / :416-426,444-454) - the literal rules.
template <bool PLANE, bool NEARLY = false>
__device__ __forceinline__ void consider2(float d, int j, int key, int closest, int cid, unsigned long long& t2, unsigned long long& t3, int lo = 0, int hi = 0) {
  const int t = j - closest, nt = closest - j;
  const unsigned seq = (unsigned)(t > nt ? t : nt) | ((unsigned)t & 0x80000000u);   // up: j - closest; down: 2^31 + closest - j
  bool ok, own;
  if (NEARLY) {
    ok = j != closest && j > lo && j < hi && d < 25.0f;
    own = t > 0 ? key <= cid : key >= cid;
  } else {
    ok = j != closest && (unsigned)(key - cid + 2) <= 4u && d < 25.0f;               // (double)d < 25.0 (:305,393) is the same test: 25 is an f32 number
    own = key == cid;
  }
  // a candidate outside its class enters the minimum with distance word ~0: "nothing found" is a key whose HIGH word is ~0 (none() below)
  const unsigned db = __float_as_uint(d);
  if (PLANE) {
    take_min(t2, ((unsigned long long)(ok && own ? db : ~0u) << 32) | seq);
    take_min(t3, ((unsigned long long)(ok && !own ? db : ~0u) << 32) | seq);
  } else take_min(t2, ((unsigned long long)(ok && !own ? db : ~0u) << 32) | seq);
}
__device__ __forceinline__ bool none(unsigned long long best) { return (unsigned)(best >> 32) == ~0u; }
__device__ __forceinline__ int index_of(unsigned long long best, int closest) {
  const unsigned seq = (unsigned)best;
  return (seq & 0x80000000u) ? closest - (int)(seq & 0x7fffffffu) : closest + (int)seq;
}

// ---- the tails, one query at a time on all 64 lanes (state in scalar registers) ----
// Which cell a lane looks up is a property of the lane: the offsets come from constant tables (one load, no integer division on the vector
// unit), and the distance from the query to a cell d cells away along an axis is |d| cell + (the distance to the own cell's face on that side
// - cell - 1 mm), clamped at 0 — the quantity cell_gap() computes, with the query-dependent part hoisted into two wave-uniform numbers per axis.
struct alignas(16) CellOff2 { int dx, dy; float ax, ay; };                    // ax = |dx| * kCell2
struct alignas(16) CellOff3 { int dx, dy, dz; int pad; };
#define ALOAM_C2(x, y) {x, y, (x < 0 ? -x : x) * 2.625f, (y < 0 ? -y : y) * 2.625f}
__device__ __constant__ CellOff2 kRingCells[24] = {
    // the eight cells around the own cell
    ALOAM_C2(-1, -1), ALOAM_C2(0, -1), ALOAM_C2(1, -1), ALOAM_C2(-1, 0), ALOAM_C2(1, 0), ALOAM_C2(-1, 1), ALOAM_C2(0, 1), ALOAM_C2(1, 1),
    // the sixteen around those
    ALOAM_C2(-2, -2), ALOAM_C2(-1, -2), ALOAM_C2(0, -2), ALOAM_C2(1, -2), ALOAM_C2(2, -2), ALOAM_C2(-2, 2), ALOAM_C2(-1, 2), ALOAM_C2(0, 2), ALOAM_C2(1, 2), ALOAM_C2(2, 2),
    ALOAM_C2(-2, -1), ALOAM_C2(-2, 0), ALOAM_C2(-2, 1), ALOAM_C2(2, -1), ALOAM_C2(2, 0), ALOAM_C2(2, 1)};
#undef ALOAM_C2
static_assert(kCell2 == 2.625f, "kRingCells holds |d| * kCell2");
__device__ __constant__ CellOff3 kBlock27[27] = {
    {-1, -1, -1, 0}, {0, -1, -1, 0}, {1, -1, -1, 0}, {-1, 0, -1, 0}, {0, 0, -1, 0}, {1, 0, -1, 0}, {-1, 1, -1, 0}, {0, 1, -1, 0}, {1, 1, -1, 0},
    {-1, -1, 0, 0},  {0, -1, 0, 0},  {1, -1, 0, 0},  {-1, 0, 0, 0},  {0, 0, 0, 0},  {1, 0, 0, 0},  {-1, 1, 0, 0},  {0, 1, 0, 0},  {1, 1, 0, 0},
    {-1, -1, 1, 0},  {0, -1, 1, 0},  {1, -1, 1, 0},  {-1, 0, 1, 0},  {0, 0, 1, 0},  {1, 0, 1, 0},  {-1, 1, 1, 0},  {0, 1, 1, 0},  {1, 1, 1, 0}};
// the two wave-uniform numbers of an axis: (own cell's upper face - s) - cell - 1 mm and (s - lower face) - cell - 1 mm
struct AxisGap { float up, dn; };
__device__ __forceinline__ AxisGap axis_gap_of(float s, int c, float cell) {
  const float lo = (float)c * cell;
  return {((lo + cell) - s) - cell - 1e-3f, (s - lo) - cell - 1e-3f};
}
__device__ __forceinline__ float axis_gap(int d, float ad_cell, const AxisGap& a) { return fmaxf(ad_cell + (d > 0 ? a.up : a.dn), 0.f); }

// The tails exist in two forms: HALVES = false — one query on all 64 lanes, its state wave-uniform (scalar registers); HALVES = true —
// the two queries of the wave side by side, 32 lanes each, when BOTH need the tail (a quarter of the pairs): one set of look-ups, scans,
// minima and bounds for two queries.  (One query alone on half-width rows would need twice the rows, hence the two forms.)
template <bool HALVES> __device__ __forceinline__ unsigned long long seg_min(unsigned long long v, int last4) { return HALVES ? half_min_packed(v, last4) : wave_min_u64(v); }
// maximum over the queries of the wave of a per-query look-up count (uniform)
template <bool HALVES> __device__ __forceinline__ int seg_max_count(int n) {
  if (!HALVES) return __builtin_amdgcn_readfirstlane(n);
  const int a = __builtin_amdgcn_readlane(n, 0), b = __builtin_amdgcn_readlane(n, 32);
  return a > b ? a : b;
}

// expanding cubic shells of coarse cells around a query whose neighbour is not inside its fine block (at most three steps reach
// DISTANCE_SQ_THRESHOLD); returns the improved packed 1-NN key.  `need`: this query takes part (HALVES: the other may be done already)
template <int kRows, bool HALVES>
__device__ __forceinline__ unsigned long long coarse_shells(const GridView& g, unsigned last_index, float cellc, float sx, float sy, float sz, unsigned long long nn, bool need,
                                                            int lane, int last4, int* lds, int stat_cls = 0) {
  constexpr int W = HALVES ? 32 : 64;
  const int l = lane & (W - 1);
  const float invc = 1.0f / cellc;
  const float2v sxy = {sx, sy};
  const int ux = (int)floorf(sx * invc), uy = (int)floorf(sy * invc), uz = (int)floorf(sz * invc);
  const AxisGap ax = axis_gap_of(sx, ux, cellc), ay = axis_gap_of(sy, uy, cellc), az = axis_gap_of(sz, uz, cellc);
  const unsigned hm = (unsigned)(g.H - 1);
  unsigned long long t1 = nn;
  ALOAM_STAT(stat_cls, 28, 1); (void)stat_cls;       // (round 5's instrumented build counted both classes in the corner slot)
  for (int r = 1;; ++r) {
    const int ncell = r == 1 ? 27 : 24 * r * r + 2;
    const float limit = fminf(__uint_as_float((unsigned)(nn >> 32)), 25.0f);
    for (int cb = 0; cb < ncell; cb += W) {
      const int c = cb + l;
      int s0 = 0, cnt = 0;
      if (need && c < ncell) {
        int dx, dy, dz;
        if (r == 1) { const CellOff3 o = kBlock27[c]; dx = o.dx; dy = o.dy; dz = o.dz; }
        else shell3d(r, c, &dx, &dy, &dz);
        // a cell farther away than the best candidate so far (or than DISTANCE_SQ_THRESHOLD) cannot change the answer
        const float gx = axis_gap(dx, fabsf((float)dx) * cellc, ax), gy = axis_gap(dy, fabsf((float)dy) * cellc, ay), gz = axis_gap(dz, fabsf((float)dz) * cellc, az);
        if (((gx * gx + gy * gy) + gz * gz) * 0.999f <= limit) {
          const unsigned hh = hash3(ux + dx, uy + dy, uz + dz) & hm;
          bucket_bounds(g.start3c, hh, s0, cnt);
        }
      }
      sweep2<HALVES, kRows>(g.sorted3c, last_index, s0, cnt, lane, last4, lds, [&](const float4& p, int, bool) { take_min(t1, nn_key(dist_to(p, sxy, sz), __float_as_uint(p.w))); });
    }
    nn = seg_min<HALVES>(t1, last4);
    const float bc = ((float)r - 0.01f) * cellc, b2 = bc * bc;
    if (__uint_as_float((unsigned)(nn >> 32)) <= b2) need = false;             // found
    if (b2 >= 25.0f || !__ballot(need)) break;                                 // nothing within DISTANCE_SQ_THRESHOLD is left / every query is served
  }
  return nn;
}

// ring grid, three stages: the query's own cell, the eight cells around it, the sixteen around those (two cells = 5.25 m cover
// DISTANCE_SQ_THRESHOLD); per cell the ring keys cid -1, +1, -2, +2 (the other rings) and, planar class, cid itself.  After a stage everything
// within the distance to the border of what has been visited is known; cells that cannot hold anything closer than the class's best so far
// are skipped — the own cell first because its ~50 candidates usually bound the search to one or two of the eight cells (and their ~400 candidates).
// Measured (batch 1024, two launches; per query pair the planar class sweeps 47 + 95 candidates in 0.98 + 0.85 stages instead of 214 in 1.04):
// planar class 2.23 -> 2.17 ms, corner class (a third of the candidates, so the extra stage costs more than it prunes) 0.754 -> 0.817 ms.
template <bool PLANE, int kRows, bool HALVES, bool NEARLY>
__device__ __forceinline__ void ring_grid(const GridView& g, unsigned last_index, float sx, float sy, float sz, int closest, int cid, bool want2, bool want3,
                                          unsigned long long& best2, unsigned long long& best3, int lane, int last4, int* lds, int lo, int hi) {
  constexpr bool kOwnFirst = PLANE;                                     // the own cell as a stage of its own: pays for the planar class only (corner: 0.754 -> 0.817 ms)
  constexpr int W = HALVES ? 32 : 64;
  const int l = lane & (W - 1);
  const unsigned hm = (unsigned)(g.H - 1);
  const float2v sxy = {sx, sy};
  unsigned long long t2 = best2, t3 = best3;
  float lim2 = !none(best2) ? __uint_as_float((unsigned)(best2 >> 32)) : 25.0f, lim3 = !none(best3) ? __uint_as_float((unsigned)(best3 >> 32)) : 25.0f;
  const float rc = kCell2, inv = 1.0f / rc;
  const int cx = (int)floorf(sx * inv), cy = (int)floorf(sy * inv);
  const AxisGap ax = axis_gap_of(sx, cx, rc), ay = axis_gap_of(sy, cy, rc);
  const bool track2 = __ballot(want2) != 0ull;                          // class 2 open at all (else its minimum is final and not taken again)
  ALOAM_STAT(PLANE ? 1 : 0, 20, 1); ALOAM_STAT(PLANE ? 1 : 0, 21, want2); ALOAM_STAT(PLANE ? 1 : 0, 22, want3); ALOAM_STAT(PLANE ? 1 : 0, 23, !none(best3) || (!PLANE && !none(best2)));
  for (int stage = kOwnFirst ? 0 : 1; stage < 3 && __ballot(want2 || want3); ++stage) {
    ALOAM_STAT(PLANE ? 1 : 0, 24 + stage, 1);
    // look-ups: lane = slot * cells + cell; slots 0..3 = the rings cid -1, +1, -2, +2, slot 4 = cid (planar class).  Stage 1 without the own-cell
    // stage has 9 cells: the own cell is looked up by the lanes behind the 8 x slots block
    // other rings wanted / own ring wanted (nearly-sorted clouds: a "same ring" neighbour may carry a neighbouring key, so either wish looks everywhere)
    const bool wo = PLANE ? (NEARLY ? want2 || want3 : want3) : want2, ws = PLANE && (NEARLY ? want2 || want3 : want2);
    const int lc = stage == 0 ? 0 : stage == 1 ? 3 : 4, ncell = 1 << lc, nslot = PLANE ? 5 : 4;
    const int n_look = ncell * nslot + (stage == 1 && !kOwnFirst ? nslot : 0);
    ALOAM_PHASE("ring_lookup");
    for (int lb = 0; lb < n_look; lb += W) {
      const int k = lb + l;
      int s0 = 0, cnt = 0;
      const bool centre = k >= ncell * nslot;                            // (stage 1 without the own-cell stage only)
      const int slot = centre ? k - ncell * nslot : k >> lc;
      const bool other = slot < 4;
      if (k < n_look && (other ? wo : ws)) {
        const int key = cid - 2 + (int)((0x2819u >> (slot * 3)) & 7u);             // slots 0..4: cid -1, +1, -2, +2, 0 (3 bits each, offset by 2)
        int ddx = 0, ddy = 0;
        float gx = 0.f, gy = 0.f;
        if (stage > 0 && !centre) {
          const CellOff2 o = kRingCells[(stage == 1 ? 0 : 8) + (k & (ncell - 1))];
          ddx = o.dx; ddy = o.dy;
          gx = axis_gap(ddx, o.ax, ax); gy = axis_gap(ddy, o.ay, ay);
        }
        const bool second = PLANE ? !other : true;
        if (key >= 0 && (gx * gx + gy * gy) * 0.999f <= (second ? lim2 : lim3)) {
          const unsigned hh = hash3(cx + ddx, cy + ddy, key) & hm;
          bucket_bounds(g.start2, hh, s0, cnt);
        }
      }
      ALOAM_PHASE("ring_sweep");
      sweep2<HALVES, kRows>(g.sorted2, last_index, s0, cnt, lane, last4, lds, [&](const float4& p, int, bool) {
        const unsigned wb = __float_as_uint(p.w);
        consider2<PLANE, NEARLY>(dist_to(p, sxy, sz), (int)(wb & kIdxMask), (int)(wb >> 20) - 1, closest, cid, t2, t3, lo, hi);
      }, nullptr, nullptr, PLANE ? 1 : 0, 8 + 4 * stage);
    }
    ALOAM_PHASE("ring_mins");
    if (!PLANE || track2) best2 = seg_min<HALVES>(t2, last4);
    if (PLANE) best3 = seg_min<HALVES>(t3, last4);
    ALOAM_PHASE("ring_bounds");
    if (stage == 2) break;                                               // two cells = 5.25 m: everything within DISTANCE_SQ_THRESHOLD has been seen
    float bound;                                                          // every point not visited so far is farther than this
    if (stage == 0) bound = (1.0f - 0.01f) * fmaxf(fminf(fminf(ax.up, ax.dn), fminf(ay.up, ay.dn)) + rc, 0.f);   // nearest face of the own cell, minus 1 mm
    else bound = ((float)stage - 0.01f) * rc;
    const float b2 = bound * bound;
    if (!none(best2)) lim2 = __uint_as_float((unsigned)(best2 >> 32));
    if (PLANE && !none(best3)) lim3 = __uint_as_float((unsigned)(best3 >> 32));
    if (!none(best2) && lim2 <= b2) want2 = false;
    if (PLANE && !none(best3) && lim3 <= b2) want3 = false;
  }
}

__device__ __forceinline__ unsigned long long read_u64(unsigned long long v, int src) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
}
__device__ __forceinline__ float read_f32(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

template <bool PLANE, int kRows, bool PAIRED_TAILS, bool NEARLY>
__device__ __forceinline__ void associate_pair(const OdomArgs& a, int b, int qi0, int nq, int nt, const GridView& g, int lane, int* lds) {
  constexpr int kRows1 = (kRows + 1) / 2;                                    // rows of 64 of the one-query tails
  const int l = lane & 31, hsel = lane >> 5, last4 = (lane | 31) << 2;
  const int qi = qi0 + hsel;
  const bool qact = qi < nq;
  const int qcap = PLANE ? a.R * 24 : a.R * 12;
  const long long qo = (long long)b * qcap + (qact ? qi : qi0);
  const float4 raw = (PLANE ? a.flat : a.sharp)[qo];
  const float4 sel = (PLANE ? a.sel_flat : a.sel_sharp)[qo];                 // k_transform_queries
  const float frac = raw.w - (float)(int)raw.w;                              // relTime of the point (:116)
  const float4* T = PLANE ? a.surf_last + (long long)b * a.cap : a.corner_last + (long long)b * a.R * 120;
  const unsigned hm = (unsigned)(g.H - 1), last_index = (unsigned)(nt - 1);
  const float cell = cell3_of(PLANE ? 1 : 0);
  const float2v selxy = {sel.x, sel.y};

  ALOAM_PHASE("setup");
  // ---- exact 1-NN: the 3x3x3 block of fine cells of both queries, the candidates kept as (distance, index | ring)
  unsigned long long t1 = ~0ull;
  KeptPair<kRows> kept;
  int kept_rows = 0;                                                         // uniform
#pragma unroll
  for (int u = 0; u < kRows; ++u) kept.k[u] = ~0ull;                             // a distance that fails every test
  {
    const float inv = 1.0f / cell;
    const int cx = (int)floorf(sel.x * inv), cy = (int)floorf(sel.y * inv), cz = (int)floorf(sel.z * inv);
    int s0 = 0, cnt = 0;
    if (l < 27 && qact) {
      const unsigned hh = hash3(cx + l % 3 - 1, cy + (l % 9) / 3 - 1, cz + l / 9 - 1) & hm;
      bucket_bounds(g.start3, hh, s0, cnt);
    }
    sweep2<true, kRows>(g.sorted3, last_index, s0, cnt, lane, last4, lds, [&](const float4& p, int u, bool first) {
      const unsigned long long v = nn_key(dist_to(p, selxy, sel.z), __float_as_uint(p.w));
      if (first) kept.k[u] = v;
      take_min(t1, v);
    }, &kept.ok, &kept_rows, PLANE ? 1 : 0, 4);
  }
  ALOAM_PHASE("nn_min");
  unsigned long long nn = half_min_packed(t1, last4);
  // every point outside the 3x3x3 block is farther away than one cell plus the distance to the nearest face of the query's own cell
  // (minus the margins of cell_gap): between 1 and 1.5 cells, by query
  float fine_b2;
  {
    const float inv = 1.0f / cell;
    const float lx = floorf(sel.x * inv) * cell, ly = floorf(sel.y * inv) * cell, lz = floorf(sel.z * inv) * cell;
    const float gap = fminf(fminf(fminf(sel.x - lx, (lx + cell) - sel.x), fminf(sel.y - ly, (ly + cell) - sel.y)), fminf(sel.z - lz, (lz + cell) - sel.z));
    const float b = (1.0f - 0.01f) * (cell + (gap > 1e-3f ? gap - 1e-3f : 0.f));
    fine_b2 = b * b;
  }
  {
    const float bound2 = fine_b2;
    const bool need1 = qact && !(nn != ~0ull && __uint_as_float((unsigned)(nn >> 32)) <= bound2);
    const unsigned long long need = __ballot(need1);
    const bool n0 = need & 1ull, n1 = (need >> 32) & 1ull;
    if (PAIRED_TAILS && n0 && n1) nn = coarse_shells<kRows1, true>(g, last_index, cell * kCell3CoarseFactor, sel.x, sel.y, sel.z, nn, need1, lane, last4, lds, PLANE ? 1 : 0);
    else if (n0 || n1) {
      const int q = n0 ? 0 : 1;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {                                       // (both, one after the other, in builds without the paired form)
        if (PAIRED_TAILS ? qq != q : !((need >> (32 * qq)) & 1ull)) continue;
        const unsigned long long r = coarse_shells<kRows1, false>(g, last_index, cell * kCell3CoarseFactor, read_f32(sel.x, 32 * qq), read_f32(sel.y, 32 * qq), read_f32(sel.z, 32 * qq),
                                                                    read_u64(nn, 32 * qq), true, lane, 0, lds, PLANE ? 1 : 0);
        if (hsel == qq) nn = r;
      }
    }
  }
  ALOAM_PHASE("classes");
  // ---- second / third neighbour (:304-361 / :392-455)
  const float nnd = __uint_as_float((unsigned)(nn >> 32));
  const bool has1 = qact && nn != ~0ull && nnd < 25.0f;                      // DISTANCE_SQ_THRESHOLD (:65,305,393)
  const int closest = (int)((unsigned)nn >> 12);
  const int cid = (int)((unsigned)nn & 0xfffu) - 1;                          // closestPointScanID (:308,398)
  unsigned long long best2 = ~0ull, best3 = ~0ull;
  int wlo = -1, whi = nt;                                                    // nearly-sorted clouds: the index range the reference's walks visit
  if (NEARLY && has1) {
    const int S = a.R + 8, c = min(max(cid, 0), a.R);
    whi = g.walk[c + 3];
    wlo = c >= 3 ? g.walk[S + c - 3] : -1;
  }
  ALOAM_STAT(PLANE ? 1 : 0, 0, 1); ALOAM_STAT(PLANE ? 1 : 0, 1, (int)((__ballot(qact) & 1) + ((__ballot(qact) >> 32) & 1))); ALOAM_STAT(PLANE ? 1 : 0, 2, (int)((__ballot(has1) & 1) + ((__ballot(has1) >> 32) & 1)));
  ALOAM_STAT(PLANE ? 1 : 0, 3, (int)((__ballot(has1 && kept.ok) & 1) + ((__ballot(has1 && kept.ok) >> 32) & 1)));
  if (__ballot(has1)) {
    bool want2 = has1, want3 = PLANE && has1;
    if (__ballot(has1 && kept.ok)) {
      // the candidates of the fine block are still in registers: neighbours found there within (almost) one cell are final
      unsigned long long t2 = ~0ull, t3 = ~0ull;
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        if (u >= kept_rows) continue;
        const unsigned lo = (unsigned)kept.k[u];
        consider2<PLANE, NEARLY>(__uint_as_float((unsigned)(kept.k[u] >> 32)), (int)(lo >> 12), (int)(lo & 0xfffu) - 1, closest, cid, t2, t3, wlo, whi);
      }
      const bool use = has1 && kept.ok;
      best2 = half_min_packed(use ? t2 : ~0ull, last4);
      if (PLANE) best3 = half_min_packed(use ? t3 : ~0ull, last4);
      const float b2 = fine_b2;
      if (!none(best2) && __uint_as_float((unsigned)(best2 >> 32)) <= b2) want2 = false;
      if (PLANE && !none(best3) && __uint_as_float((unsigned)(best3 >> 32)) <= b2) want3 = false;
    }
    ALOAM_PHASE("ring_tail");
    const unsigned long long w2 = __ballot(want2), w3 = __ballot(want3);
    const bool r0 = (w2 | w3) & 1ull, r1 = ((w2 | w3) >> 32) & 1ull;
    if (PAIRED_TAILS && r0 && r1) ring_grid<PLANE, kRows1, true, NEARLY>(g, last_index, sel.x, sel.y, sel.z, closest, cid, want2, want3, best2, best3, lane, last4, lds, wlo, whi);
    else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bool q2 = (w2 >> (32 * q)) & 1ull, q3 = (w3 >> (32 * q)) & 1ull;
        if (!(q2 || q3)) continue;
        unsigned long long r2 = read_u64(best2, 32 * q), r3 = read_u64(best3, 32 * q);
        ring_grid<PLANE, kRows1, false, NEARLY>(g, last_index, read_f32(sel.x, 32 * q), read_f32(sel.y, 32 * q), read_f32(sel.z, 32 * q), __builtin_amdgcn_readlane(closest, 32 * q),
                                                __builtin_amdgcn_readlane(cid, 32 * q), q2, q3, r2, r3, lane, 0, lds, __builtin_amdgcn_readlane(wlo, 32 * q), __builtin_amdgcn_readlane(whi, 32 * q));
        if (hsel == q) { best2 = r2; best3 = r3; }
      }
    }
  }
  ALOAM_PHASE("record");
  const bool valid = has1 && !none(best2) && (!PLANE || !none(best3));   // :363 / :457

  // ---- the record: lane 0 of the half stores the raw, untransformed point (:365-367 / :460-462), lanes 1.. the neighbours (read from the
  // ring-ordered cloud by index: the grid entries are copies of exactly those points), the next lane the flag and the relTime
  if (qact) {
    constexpr int kPts = PLANE ? 4 : 3;
    float* e = PLANE ? reinterpret_cast<float*>(a.planes + (long long)b * a.R * 24 + qi) : reinterpret_cast<float*>(a.edges + (long long)b * a.R * 12 + qi);
    if (l < kPts) {
      float x = raw.x, y = raw.y, z = raw.z;
      if (l > 0) {
        x = y = z = 0.f;
        if (valid) {
          const int j = l == 1 ? closest : index_of(l == 2 ? best2 : best3, closest);
          const float4 p = T[j];
          x = p.x; y = p.y; z = p.z;
        }
      }
      e[3 * l] = x; e[3 * l + 1] = y; e[3 * l + 2] = z;
    } else if (l == kPts) {
      int* ei = reinterpret_cast<int*>(e) + 3 * kPts;
      ei[0] = valid ? 1 : 0; ei[1] = __float_as_int(frac); ei[2] = 0;
      if (PLANE) ei[3] = 0;
    }
  }
}

template <bool PLANE, bool WIDE>
__global__ __launch_bounds__(64) void k_associate_pair(OdomArgs a) {
  constexpr int kRows = PairRows<PLANE, WIDE>::value;
  __shared__ int lds[sweep2_lds_ints<kRows>()];
  // XCD-aware work mapping as in k_associate: XCD x works through sequences x, x + 8, ...
  const int lane = threadIdx.x, L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int pairs = PLANE ? a.R * 12 : a.R * 6;
  const int b = (slot / pairs) * 8 + xcd, qi0 = (slot % pairs) * 2;
  if (b >= a.B) return;
  const SeqMeta m = a.meta[b];
  const int nq = PLANE ? m.n_flat : m.n_sharp;
  if (qi0 >= nq) return;
  const GridView g = grid_view(a, b, PLANE ? 1 : 0);
  const int nt = PLANE ? m.n_surf_last : m.n_corner_last;
  if (g.flags[0] != 0 || g.flags[1] != 0 || nt <= 0) return;                 // k_associate_nearly / k_associate_flagged own this sequence
  // (128-ring sensors: the paired tails would take the planar class from 79 to 85 registers, six waves per SIMD to five, for their ~1 %)
  associate_pair<PLANE, kRows, !WIDE, false>(a, b, qi0, nq, nt, g, lane, lds);
}

// Sequences whose last cloud is NEARLY ring-sorted (flags[1] == 1, walk_tables above: a sweep whose first ray had no return - a handful of
// sequences per step on range-limited data, most sweeps of a real sensor): the same pair code with the index-range form of the walk window.  A
// kernel of its own so that its three extra registers do not cost the ring-sorted majority a wave per SIMD (71 -> 74 / 62 -> 65 registers when the
// two forms share a kernel); every wave takes kPairsPerWave pairs of its sequence in turn, so the launch is small enough to be free when nothing is flagged.
constexpr int kPairsPerWave = 8;
template <bool PLANE, bool WIDE>
__global__ __launch_bounds__(64) void k_associate_nearly(OdomArgs a) {
  constexpr int kRows = PairRows<PLANE, WIDE>::value;
  __shared__ int lds[sweep2_lds_ints<kRows>()];
  const int lane = threadIdx.x, b = blockIdx.y;
  const GridView g = grid_view(a, b, PLANE ? 1 : 0);
  if (g.flags[0] != 0 || g.flags[1] != 1) return;
  const SeqMeta m = a.meta[b];
  const int nq = PLANE ? m.n_flat : m.n_sharp, nt = PLANE ? m.n_surf_last : m.n_corner_last;
  if (nt <= 0) return;
  for (int k = 0; k < kPairsPerWave; ++k) {
    const int qi0 = (blockIdx.x * kPairsPerWave + k) * 2;
    if (qi0 >= nq) return;
    associate_pair<PLANE, kRows, !WIDE, true>(a, b, qi0, nq, nt, g, lane, lds);
  }
}

// Sequences the pair kernel leaves alone: clouds that are not ring-sorted or hold keys / coordinates outside the range the grids are
// exact for (possible through aloam_set_last), and empty clouds.  One workgroup of four waves per sequence walks through its queries with
// the literal one-query code; for every other sequence the workgroup returns at once.
template <bool PLANE, bool WIDE>
__global__ __launch_bounds__(256) void k_associate_flagged(OdomArgs a) {
  constexpr int kSweep = SweepRows<PLANE, WIDE>::value;
  __shared__ int rows[4][kSweep * 64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const SeqMeta m = a.meta[b];
  const GridView g = grid_view(a, b, PLANE ? 1 : 0);
  const int nt = PLANE ? m.n_surf_last : m.n_corner_last;
  if (!(g.flags[0] != 0 || g.flags[1] == 2 || nt <= 0)) return;
  const int nq = PLANE ? m.n_flat : m.n_sharp;
  for (int qi = wave; qi < nq; qi += 4) associate_one<PLANE, WIDE>(a, b, qi, m, g, lane, rows[wave]);
}

// -------------------------------------------------------------------------------------------------------
constexpr int kSolveWaves = 2;   // waves per sequence in k_solve; measured at batch 512: 1: 0.62 ms, 2: 0.44, 4: 0.53, 8: 0.87 (two launches)
constexpr int kSolveThreads = 64 * kSolveWaves;
template <bool WITH_JAC, bool DISTORT>
__device__ void evaluate(const OdomArgs& a, int b, const double q[4], const double t[3], double* acc, int* n_edge, int* n_plane) {
  const int tid = threadIdx.x;
  const SeqMeta m = a.meta[b];
  const EdgeRec* E = a.edges + (long long)b * a.R * 12;
  const PlaneRec* P = a.planes + (long long)b * a.R * 24;
  int ne = 0, np = 0;
  for (int i = tid; i < m.n_sharp; i += kSolveThreads) {
    const EdgeRec e = E[i];
    if (!e.valid) continue;
    ++ne;
    double rcp[3], lp[3], M[3][4];
    const double s = DISTORT ? interpolation_ratio(__int_as_float(e.pad[0])) : 1.0;
    if (DISTORT) deskew_point(q, t, s, (double)e.cp[0], (double)e.cp[1], (double)e.cp[2], lp, WITH_JAC ? M : nullptr);
    else {
      quat_rotate(q, (double)e.cp[0], (double)e.cp[1], (double)e.cp[2], rcp);
      lp[0] = rcp[0] + t[0]; lp[1] = rcp[1] + t[1]; lp[2] = rcp[2] + t[2];
    }
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
        if (DISTORT) deskew_jacobian_row(A[row], M, q, s, J);
        else {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            J[c] = A[row][0] * Bm[0][c] + A[row][1] * Bm[1][c] + A[row][2] * Bm[2][c];
            J[3 + c] = A[row][c];
          }
        }
        add_row(acc, J, rr[row], rho1);
      }
    }
  }
  for (int i = tid; i < m.n_flat; i += kSolveThreads) {
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
    double rcp[3] = {0.0, 0.0, 0.0}, lp[3], M[3][4];
    const double s = DISTORT ? interpolation_ratio(__int_as_float(p.pad[0])) : 1.0;
    if (DISTORT) deskew_point(q, t, s, (double)p.cp[0], (double)p.cp[1], (double)p.cp[2], lp, WITH_JAC ? M : nullptr);
    else {
      quat_rotate(q, (double)p.cp[0], (double)p.cp[1], (double)p.cp[2], rcp);
      lp[0] = rcp[0] + t[0]; lp[1] = rcp[1] + t[1]; lp[2] = rcp[2] + t[2];
    }
    const double r = (lp[0] - jx) * nx + (lp[1] - jy) * ny + (lp[2] - jz) * nz;
    double rho0, rho1;
    huber(r * r, &rho0, &rho1);
    acc[27] += 0.5 * rho0;
    if (WITH_JAC) {
      double J[6];
      const double nn[3] = {nx, ny, nz};
      if (DISTORT) deskew_jacobian_row(nn, M, q, s, J);
      else { J[0] = 2.0 * (nz * rcp[1] - ny * rcp[2]); J[1] = 2.0 * (nx * rcp[2] - nz * rcp[0]); J[2] = 2.0 * (ny * rcp[0] - nx * rcp[1]); J[3] = nx; J[4] = ny; J[5] = nz; }
      add_row(acc, J, r, rho1);
    }
  }
  *n_edge = ne;
  *n_plane = np;
}

// One workgroup per sequence: the whole ceres::Solve stand-in + pose integration.
// Every thread runs the (uniform) scalar LM logic redundantly; only the evaluations are distributed.
template <bool DISTORT>
__global__ __launch_bounds__(kSolveThreads) void k_solve(OdomArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double s_red[kSolveWaves * 28];
  OdomState& st = a.state[b];
  double q[4] = {st.para_q[0], st.para_q[1], st.para_q[2], st.para_q[3]};
  double t[3] = {st.para_t[0], st.para_t[1], st.para_t[2]};

  const LmResult lm = lm_solve_block<kSolveWaves>([&](bool with_jac, const double* qq, const double* tt, double* acc, int* ne, int* np) {
    if (with_jac) evaluate<true, DISTORT>(a, b, qq, tt, acc, ne, np); else evaluate<false, DISTORT>(a, b, qq, tt, acc, ne, np);
  }, q, t, a.lm_max_iterations, s_red);
  const int n_edges = lm.n_a, n_planes = lm.n_b, iterations = lm.iterations, successful = lm.successful, termination = lm.termination;
  const double initial_cost = lm.initial_cost, cost = lm.final_cost;

  if (tid == 0) {
    if (termination == 5) {                               // FAILURE: para_q / para_t stay what they were (ceres::Solve restores them)
      for (int k = 0; k < 4; ++k) q[k] = st.para_q[k];
      for (int k = 0; k < 3; ++k) t[k] = st.para_t[k];
    }
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
size_t build_grids_lds_bytes(int H, int R) { return sizeof(int) * ((size_t)H + 1024 + 2 * (R + 8) + 4); }
static size_t build_grids_fused_lds_bytes(int H) { return sizeof(int) * (3 * (size_t)(H / 2) + 48 + 8 + 2 * kWalkKeys); }
// The surf grid needs > 64 KiB of dynamic LDS: the attribute belongs to the function ON the current device; aloam_create sets it
// once per context (no process-global state).
int prepare_build_grids(int H_surf) {
  if (hipFuncSetAttribute((const void*)k_build_grids_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)build_grids_fused_lds_bytes(kFusedMaxH)) != hipSuccess) return -1;
  return hipFuncSetAttribute((const void*)k_build_grids, hipFuncAttributeMaxDynamicSharedMemorySize, (int)build_grids_lds_bytes(H_surf, kMaxRings)) == hipSuccess ? 0 : -1;
}
void launch_build_grids(const OdomArgs& a, hipStream_t s) {
  // clouds of up to 65535 points: all three grids by one workgroup from two reads; the per-grid workgroups below return at once for those
  // (always launched: which kernel owns a cloud is decided per cloud by fused_takes() in both kernels; LDS sized for the larger table
  // that can be fused)
  const int hc = a.grid_H_corner <= kFusedMaxH ? a.grid_H_corner : 0, hs = a.grid_H_surf <= kFusedMaxH ? a.grid_H_surf : 0;
  if (hc || hs) hipLaunchKernelGGL(k_build_grids_fused, dim3(2, a.B), dim3(1024), build_grids_fused_lds_bytes(hc > hs ? hc : hs), s, a);
  hipLaunchKernelGGL(k_build_grids, dim3(6, a.B), dim3(1024), build_grids_lds_bytes(a.grid_H_surf, a.R), s, a);
}
void launch_transform_queries(const OdomArgs& a, hipStream_t s) {
  const dim3 grid((a.R * 36 + 255) / 256, a.B);
  if (a.distortion) hipLaunchKernelGGL(k_transform_queries<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_transform_queries<false>, grid, dim3(256), 0, s, a);
}
void launch_associate(const OdomArgs& a, bool plane, hipStream_t s) {
  const int by = (a.B + 7) / 8 * 8;      // padded so that every (XCD, sequence slot) pair exists (see k_associate_pair)
  const int qcap = plane ? a.R * 24 : a.R * 12;   // the kernel decodes (sequence, query) from blockIdx.x with exactly this slot count
  // sensors with more than 64 rings: the ring grid's +-2-ring window and the fine blocks hold about twice the candidates, so the
  // waves keep more rows of candidates in flight per sweep round (rows by class AND ring count)
  const bool wide = a.R > 64;
  // every sequence belongs to exactly one of the three kernels (flags of its last cloud, k_build_grids_fused): ring-sorted -> pair kernel,
  // nearly ring-sorted -> pair code with the index-range walk window, anything else -> literal one-query walks; the other two return at once
  const dim3 grid((unsigned)(qcap / 2 * by)), block(64), gridf((unsigned)a.B), blockf(256);
  const dim3 gridn((unsigned)((qcap / 2 + kPairsPerWave - 1) / kPairsPerWave), (unsigned)a.B);
  if (wide) {
    if (plane) { hipLaunchKernelGGL((k_associate_pair<true, true>), grid, block, 0, s, a); hipLaunchKernelGGL((k_associate_nearly<true, true>), gridn, block, 0, s, a); hipLaunchKernelGGL((k_associate_flagged<true, true>), gridf, blockf, 0, s, a); }
    else { hipLaunchKernelGGL((k_associate_pair<false, true>), grid, block, 0, s, a); hipLaunchKernelGGL((k_associate_nearly<false, true>), gridn, block, 0, s, a); hipLaunchKernelGGL((k_associate_flagged<false, true>), gridf, blockf, 0, s, a); }
  } else {
    if (plane) { hipLaunchKernelGGL((k_associate_pair<true, false>), grid, block, 0, s, a); hipLaunchKernelGGL((k_associate_nearly<true, false>), gridn, block, 0, s, a); hipLaunchKernelGGL((k_associate_flagged<true, false>), gridf, blockf, 0, s, a); }
    else { hipLaunchKernelGGL((k_associate_pair<false, false>), grid, block, 0, s, a); hipLaunchKernelGGL((k_associate_nearly<false, false>), gridn, block, 0, s, a); hipLaunchKernelGGL((k_associate_flagged<false, false>), gridf, blockf, 0, s, a); }
  }
}
void launch_solve(const OdomArgs& a, hipStream_t s) {
  if (a.distortion) hipLaunchKernelGGL(k_solve<true>, dim3(a.B), dim3(kSolveThreads), 0, s, a);
  else hipLaunchKernelGGL(k_solve<false>, dim3(a.B), dim3(kSolveThreads), 0, s, a);
}

}  // namespace aloam

#ifdef ALOAM_PHASE_CLOCK
extern "C" int aloam_debug_phase_clock_odo(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(aloam::g_phase_clock_odo), sizeof(unsigned long long) * 64); }
#endif
#ifdef ALOAM_ASSOC_STATS
extern "C" int aloam_debug_assoc_stats(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(aloam::g_assoc_stats), sizeof(unsigned long long) * 64); }
#endif
