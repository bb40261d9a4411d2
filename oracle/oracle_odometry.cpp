// oracle/oracle_odometry.cpp — CPU restatement of A-LOAM scan-to-scan odometry (one pass of the
// main-loop body).  TEST INFRASTRUCTURE ONLY (see aloam_oracle.h).  Follows reference
// src/laserOdometry.cpp:
//   TransformToStart                       :111-129
//   first-frame gate                       :267-271
//   corner association (1-NN + ring walk)  :299-384
//   plane  association (1-NN + ring walk)  :387-483
//   Ceres problem / solve, twice           :278-291,494-501   (solver: oracle_solver.cpp)
//   pose integration                       :504-505
//   cloud swap + kd-tree rebuild           :554-568
// pcl::KdTreeFLANN (exact, eps = 0) is stood in for by an exact kd-tree / brute force that evaluates
// the same f32 squared distance ((dx*dx + dy*dy) + dz*dz) FLANN's L2_Simple accumulates (SURVEY.md
// Appendix C); equal-distance ties go to the lowest index (FLANN leaves them to traversal order).
#include <algorithm>
#include <cmath>
#include <limits>

#include "oracle_internal.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------
// exact nearest neighbour
// ---------------------------------------------------------------------------------------------
static inline float sqdist(const P4& a, const float q[3]) {
  const float dx = a.x - q[0], dy = a.y - q[1], dz = a.z - q[2];
  return (dx * dx + dy * dy) + dz * dz;
}

void NnIndex::build(const std::vector<P4>& cloud) {
  pts = cloud;
  perm.resize(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) perm[i] = (int)i;
  nodes.clear();
  if (!pts.empty()) build_rec(0, (int)pts.size());
}

int NnIndex::build_rec(int lo, int hi) {
  const int id = (int)nodes.size();
  nodes.push_back(Node{lo, hi, -1, -1, -1, 0.f});
  if (hi - lo <= 12) return id;
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
  for (int k = lo; k < hi; ++k) {
    const P4& p = pts[perm[k]];
    const float v[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], v[a]); mx[a] = std::max(mx[a], v[a]); }
  }
  int axis = 0;
  if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
  if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
  if (!(mx[axis] > mn[axis])) return id;   // all coincident: keep as a leaf
  auto coord = [&](int i) { const P4& p = pts[i]; return axis == 0 ? p.x : axis == 1 ? p.y : p.z; };
  const int mid = (lo + hi) / 2;
  std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi, [&](int a, int b) { return coord(a) < coord(b); });
  const float split = coord(perm[mid]);
  const int l = build_rec(lo, mid);
  const int r = build_rec(mid, hi);
  nodes[id].axis = axis; nodes[id].split = split; nodes[id].left = l; nodes[id].right = r;
  return id;
}

void NnIndex::search(int node, const float q[3], int* best, float* bestd) const {
  const Node& nd = nodes[node];
  if (nd.axis < 0) {
    for (int k = nd.lo; k < nd.hi; ++k) {
      const int i = perm[k];
      const float d = sqdist(pts[i], q);
      if (d < *bestd || (d == *bestd && i < *best)) { *bestd = d; *best = i; }
    }
    return;
  }
  const double diff = (double)q[nd.axis] - (double)nd.split;
  const int near = diff < 0 ? nd.left : nd.right;
  const int far = diff < 0 ? nd.right : nd.left;
  search(near, q, best, bestd);
  // conservative prune: the f32 distance of any far-side point is >= diff^2 * (1 - 1e-6)
  if (diff * diff * (1.0 - 1e-6) <= (double)*bestd) search(far, q, best, bestd);
}

void NnIndex::query(const P4& qp, bool brute, int* idx, float* d2) const {
  const float q[3] = {qp.x, qp.y, qp.z};
  int best = -1;
  float bestd = std::numeric_limits<float>::infinity();
  if (brute || nodes.empty()) {
    for (size_t i = 0; i < pts.size(); ++i) {
      const float d = sqdist(pts[i], q);
      if (d < bestd) { bestd = d; best = (int)i; }
    }
  } else {
    search(0, q, &best, &bestd);
  }
  *idx = best;
  *d2 = bestd;
}

// ---------------------------------------------------------------------------------------------
// TransformToStart (:111-129): s = 1 with DISTORTION 0 (:59), else the point's relative time / SCAN_PERIOD (:115-116, f32
// difference divided by the double 0.1).
// ---------------------------------------------------------------------------------------------
static double interpolation_ratio(const P4& pi, int distortion) {
  const double SCAN_PERIOD = 0.1;            // :64
  return distortion ? (pi.i - int(pi.i)) / SCAN_PERIOD : 1.0;
}
static P4 transform_to_start(const P4& pi, const double para_q[4], const double para_t[3], int distortion) {
  const double s = interpolation_ratio(pi, distortion);
  const Quatd q_last_curr{para_q[0], para_q[1], para_q[2], para_q[3]};
  const Quatd q_point_last = slerp_from_identity(s, q_last_curr);
  const V3d t_point_last{s * para_t[0], s * para_t[1], s * para_t[2]};
  const V3d point{pi.x, pi.y, pi.z};
  const V3d un = rotate(q_point_last, point) + t_point_last;
  return P4{(float)un.x, (float)un.y, (float)un.z, pi.i};
}

static inline float walkdist(const P4& a, const P4& sel) {   // f32 expression assigned to a double at :322-327 etc.
  return (a.x - sel.x) * (a.x - sel.x) + (a.y - sel.y) * (a.y - sel.y) + (a.z - sel.z) * (a.z - sel.z);
}

int odometry_step(const orc_config& cfg, const std::vector<P4>& sharp, const std::vector<P4>& less_sharp,
                  const std::vector<P4>& flat, const std::vector<P4>& less_flat, OdomState* st, std::string* err) {
  (void)err;
  const double DISTANCE_SQ_THRESHOLD = 25;   // :65
  const double NEARBY_SCAN = 2.5;            // :66
  st->stats = orc_odom_stats{};
  if (!st->inited) {
    st->inited = true;                       // :267-271
  } else {
    const std::vector<P4>& CL = st->corner_last;
    const std::vector<P4>& SL = st->surf_last;
    for (int opti = 0; opti < cfg.outer_iterations; ++opti) {
      st->edges.clear();
      st->planes.clear();
      // ---- corners (:299-384)
      for (int i = 0; i < (int)sharp.size() && !CL.empty(); ++i) {
        const P4 sel = transform_to_start(sharp[i], st->para_q, st->para_t, cfg.distortion);
        int nn; float nnd;
        st->tree_corner.query(sel, cfg.nn_brute != 0, &nn, &nnd);
        int closest = -1, min2 = -1;
        log_decision(kDecOdomNN, nnd, DISTANCE_SQ_THRESHOLD);
        if (nnd < DISTANCE_SQ_THRESHOLD) {
          closest = nn;
          const int cid = (int)CL[closest].i;
          double minD2 = DISTANCE_SQ_THRESHOLD;
          for (int j = closest + 1; j < (int)CL.size(); ++j) {
            if ((int)CL[j].i <= cid) continue;
            if ((int)CL[j].i > cid + NEARBY_SCAN) break;
            const double d = walkdist(CL[j], sel);
            if (d < minD2) { minD2 = d; min2 = j; }
          }
          for (int j = closest - 1; j >= 0; --j) {
            if ((int)CL[j].i >= cid) continue;
            if ((int)CL[j].i < cid - NEARBY_SCAN) break;
            const double d = walkdist(CL[j], sel);
            if (d < minD2) { minD2 = d; min2 = j; }
          }
        }
        if (min2 >= 0) {
          EdgeRec e;
          e.cp = V3d{sharp[i].x, sharp[i].y, sharp[i].z};            // raw (untransformed) point, :365-367
          e.a = V3d{CL[closest].x, CL[closest].y, CL[closest].z};
          e.b = V3d{CL[min2].x, CL[min2].y, CL[min2].z};
          e.query = i;
          e.s = interpolation_ratio(sharp[i], cfg.distortion);      // :376-377
          st->edges.push_back(e);
        }
      }
      // ---- planes (:387-483)
      for (int i = 0; i < (int)flat.size() && !SL.empty(); ++i) {
        const P4 sel = transform_to_start(flat[i], st->para_q, st->para_t, cfg.distortion);
        int nn; float nnd;
        st->tree_surf.query(sel, cfg.nn_brute != 0, &nn, &nnd);
        int closest = -1, min2 = -1, min3 = -1;
        log_decision(kDecOdomNN, nnd, DISTANCE_SQ_THRESHOLD);
        if (nnd < DISTANCE_SQ_THRESHOLD) {
          closest = nn;
          const int cid = (int)SL[closest].i;
          double minD2 = DISTANCE_SQ_THRESHOLD, minD3 = DISTANCE_SQ_THRESHOLD;
          for (int j = closest + 1; j < (int)SL.size(); ++j) {
            if ((int)SL[j].i > cid + NEARBY_SCAN) break;
            const double d = walkdist(SL[j], sel);
            if ((int)SL[j].i <= cid && d < minD2) { minD2 = d; min2 = j; }
            else if ((int)SL[j].i > cid && d < minD3) { minD3 = d; min3 = j; }
          }
          for (int j = closest - 1; j >= 0; --j) {
            if ((int)SL[j].i < cid - NEARBY_SCAN) break;
            const double d = walkdist(SL[j], sel);
            if ((int)SL[j].i >= cid && d < minD2) { minD2 = d; min2 = j; }
            else if ((int)SL[j].i < cid && d < minD3) { minD3 = d; min3 = j; }
          }
          if (min2 >= 0 && min3 >= 0) {
            PlaneRec p;
            p.cp = V3d{flat[i].x, flat[i].y, flat[i].z};
            p.j = V3d{SL[closest].x, SL[closest].y, SL[closest].z};
            p.l = V3d{SL[min2].x, SL[min2].y, SL[min2].z};
            p.m = V3d{SL[min3].x, SL[min3].y, SL[min3].z};
            p.query = i;
            p.s = interpolation_ratio(flat[i], cfg.distortion);     // :474-475
            st->planes.push_back(p);
          }
        }
      }
      const int slot = opti < 2 ? opti : 1;     // [0] first outer iteration, [1] second or, with more than two, the last one
      st->stats.corner_corr[slot] = (int)st->edges.size();
      st->stats.plane_corr[slot] = (int)st->planes.size();
      // ---- ceres::Solve (:494-499)
      const LmSummary sm = lm_solve(st->edges, st->planes, st->para_q, st->para_t, cfg.lm_max_iterations,
                                    cfg.analytic_jacobian != 0, cfg.apply_converged_step != 0);
      st->stats.lm_iterations[slot] = sm.iterations;
      st->stats.lm_successful[slot] = sm.successful;
      st->stats.initial_cost[slot] = sm.initial_cost;
      st->stats.final_cost[slot] = sm.final_cost;
      st->stats.termination[slot] = sm.termination;
    }
    // ---- pose integration (:504-505), no renormalisation
    const Quatd q_lc{st->para_q[0], st->para_q[1], st->para_q[2], st->para_q[3]};
    const V3d t_lc{st->para_t[0], st->para_t[1], st->para_t[2]};
    st->t_w = st->t_w + rotate(st->q_w, t_lc);
    st->q_w = qmul(st->q_w, q_lc);
  }
  // ---- swap + rebuild (:554-568)
  st->corner_last = less_sharp;
  st->surf_last = less_flat;
  st->tree_corner.build(st->corner_last);
  st->tree_surf.build(st->surf_last);
  return 0;
}

}  // namespace orc
