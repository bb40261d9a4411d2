// oracle/oracle_mapping.cpp — CPU restatement of the scan-to-map refinement, body of process() in the reference's
// src/laserMapping.cpp:231-893 for ONE frame, run no-drop (every frame is processed; the live node drops queued frames,
// :299-303).  TEST INFRASTRUCTURE ONLY (see aloam_oracle.h).
//   transformAssociateToMap / transformUpdate / pointAssociateToMap     :142-163
//   cube window, shifts, valid set, submap gather                       :307-539
//   VoxelGrid of the incoming clouds, per-cube re-filter                :542-550, :788-801
//   5-NN, line fit (3x3 eigen) / plane fit (5x3 least squares), factors :556-706
//   ceres::Solve (DENSE_QR, 4 iterations), twice                        :562-572, 710-729
//   map insertion                                                       :737-783
// Third-party pieces (not vendored, not installed): pcl::VoxelGrid / KdTreeFLANN (oracle_registration.cpp /
// oracle_odometry.cpp restatements), ceres::Solve (oracle_solver.cpp), Eigen's SelfAdjointEigenSolver<Matrix3d> and
// colPivHouseholderQr — for those two this file uses the same stand-in routines as oracle/ref_shim (cyclic Jacobi /
// pivoted Householder least squares), so that oracle and oracle/_ref agree bit-for-bit; they are NOT Eigen's own
// algorithms (tridiagonal QL / blocked Householder), results differ from a real Eigen build at the 1e-15 level.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "oracle_internal.hpp"
#include "ref_shim/include/shim/eigen_shim.hpp"

namespace orc {

void sym_eigen3(const double A[9], double vals[3], double vecs[9]) {
  Eigen::Matrix3d M;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i, j) = A[i * 3 + j];
  Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> s(M);
  for (int k = 0; k < 3; ++k) { vals[k] = s.eigenvalues()(k); for (int i = 0; i < 3; ++i) vecs[i * 3 + k] = s.eigenvectors()(i, k); }
}
void lstsq_5x3(const double A[15], const double b[5], double x[3]) {
  Eigen::Matrix<double, 5, 3> M;
  Eigen::Matrix<double, 5, 1> B;
  for (int i = 0; i < 5; ++i) { B(i) = b[i]; for (int j = 0; j < 3; ++j) M(i, j) = A[i * 3 + j]; }
  const Eigen::Vector3d r = M.colPivHouseholderQr().solve(B);
  x[0] = r(0); x[1] = r(1); x[2] = r(2);
}

// ---- exact k-NN (ascending distance, lowest index on ties) ---------------------------------------------------
static inline float sqd(const P4& p, const float q[3]) {
  const float dx = p.x - q[0], dy = p.y - q[1], dz = p.z - q[2];
  return (dx * dx + dy * dy) + dz * dz;
}
namespace {
struct Cand { float d; int i; };
inline bool better(const Cand& a, const Cand& b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }
void knn_insert(Cand* best, int* n, int k, Cand c) {
  if (*n == k && !better(c, best[k - 1])) return;
  int pos = *n < k ? (*n)++ : k - 1;
  while (pos > 0 && better(c, best[pos - 1])) { best[pos] = best[pos - 1]; --pos; }
  best[pos] = c;
}
void knn_search(const NnIndex& t, int node, const float q[3], Cand* best, int* n, int k) {
  const NnIndex::Node& nd = t.nodes[node];
  if (nd.axis < 0) {
    for (int j = nd.lo; j < nd.hi; ++j) { const int i = t.perm[j]; knn_insert(best, n, k, Cand{sqd(t.pts[i], q), i}); }
    return;
  }
  const double diff = (double)q[nd.axis] - (double)nd.split;
  const int near = diff < 0 ? nd.left : nd.right, far = diff < 0 ? nd.right : nd.left;
  knn_search(t, near, q, best, n, k);
  if (*n < k || diff * diff * (1.0 - 1e-6) <= (double)best[*n - 1].d) knn_search(t, far, q, best, n, k);
}
}  // namespace
int KnnIndex::query(const P4& qp, int k, bool brute, int* idx, float* d2) const {
  const float q[3] = {qp.x, qp.y, qp.z};
  k = std::min<int>(k, (int)tree.pts.size());
  if (k <= 0) return 0;
  std::vector<Cand> best(k);
  int n = 0;
  if (brute || tree.nodes.empty()) for (size_t i = 0; i < tree.pts.size(); ++i) knn_insert(best.data(), &n, k, Cand{sqd(tree.pts[i], q), (int)i});
  else knn_search(tree, 0, q, best.data(), &n, k);
  for (int j = 0; j < n; ++j) { idx[j] = best[j].i; d2[j] = best[j].d; }
  return n;
}

// ---- helpers ------------------------------------------------------------------------------------------------
static inline P4 associate_to_map(const P4& pi, const double par[7]) {                 // pointAssociateToMap (:157-166)
  const V3d w = rotate(Quatd{par[0], par[1], par[2], par[3]}, V3d{pi.x, pi.y, pi.z}) + V3d{par[4], par[5], par[6]};
  return P4{(float)w.x, (float)w.y, (float)w.z, pi.i};
}
static inline int cube_coord(double v, int cen) {        // int((v + 25.0) / 50.0) + cen, then -1 if v + 25 < 0 (:312-321, :741-750)
  int c = int((v + 25.0) / 50.0) + cen;
  if (v + 25.0 < 0) c--;
  return c;
}

// One shift of the cube window along `axis` (0: i, 1: j, 2: k).  dir = +1: contents move towards higher index, the
// slab that falls off the top re-enters cleared at index 0 (the `while (centerCube < 3)` loops, :323-353 etc.);
// dir = -1: the opposite (`while (centerCube >= dim - 3)`, :355-385 etc.).
static void shift_cubes(MapState* st, int axis, int dir) {
  const int dim[3] = {MapState::W, MapState::H, MapState::D};
  auto at = [&](int a, int u, int v) {
    int ijk[3];
    ijk[axis] = a; ijk[(axis + 1) % 3] = u; ijk[(axis + 2) % 3] = v;
    return ijk[0] + MapState::W * ijk[1] + MapState::W * MapState::H * ijk[2];
  };
  const int n = dim[axis], nu = dim[(axis + 1) % 3], nv = dim[(axis + 2) % 3];
  for (int u = 0; u < nu; ++u) for (int v = 0; v < nv; ++v) {
    for (auto* arr : {&st->corner, &st->surf}) {
      if (dir > 0) {
        std::vector<P4> keep = std::move((*arr)[at(n - 1, u, v)]);
        for (int a = n - 1; a >= 1; --a) (*arr)[at(a, u, v)] = std::move((*arr)[at(a - 1, u, v)]);
        keep.clear();
        (*arr)[at(0, u, v)] = std::move(keep);
      } else {
        std::vector<P4> keep = std::move((*arr)[at(0, u, v)]);
        for (int a = 0; a < n - 1; ++a) (*arr)[at(a, u, v)] = std::move((*arr)[at(a + 1, u, v)]);
        keep.clear();
        (*arr)[at(n - 1, u, v)] = std::move(keep);
      }
    }
  }
}

int mapping_step(const orc_config& cfg, MapState* st, const double q_wodom[4], const double t_wodom[3], const std::vector<P4>& corner_last,
                 const std::vector<P4>& surf_last, const std::vector<P4>& full_res) {
  const bool canonical = cfg.canonical_order != 0;
  double* par = st->parameters;
  const Quatd q_wodom_curr{q_wodom[0], q_wodom[1], q_wodom[2], q_wodom[3]};
  const V3d t_wodom_curr{t_wodom[0], t_wodom[1], t_wodom[2]};
  {  // transformAssociateToMap (:142-146)
    const Quatd q = qmul(st->q_wmap_wodom, q_wodom_curr);
    const V3d t = rotate(st->q_wmap_wodom, t_wodom_curr) + st->t_wmap_wodom;
    par[0] = q.x; par[1] = q.y; par[2] = q.z; par[3] = q.w; par[4] = t.x; par[5] = t.y; par[6] = t.z;
  }
  // ---- cube window (:311-507)
  int cI = cube_coord(par[4], st->cenW), cJ = cube_coord(par[5], st->cenH), cK = cube_coord(par[6], st->cenD);
  while (cI < 3) { shift_cubes(st, 0, +1); cI++; st->cenW++; }
  while (cI >= MapState::W - 3) { shift_cubes(st, 0, -1); cI--; st->cenW--; }
  while (cJ < 3) { shift_cubes(st, 1, +1); cJ++; st->cenH++; }
  while (cJ >= MapState::H - 3) { shift_cubes(st, 1, -1); cJ--; st->cenH--; }
  while (cK < 3) { shift_cubes(st, 2, +1); cK++; st->cenD++; }
  while (cK >= MapState::D - 3) { shift_cubes(st, 2, -1); cK--; st->cenD--; }
  // ---- valid cubes, submap (:509-539)
  st->valid.clear();
  for (int i = cI - 2; i <= cI + 2; i++) for (int j = cJ - 2; j <= cJ + 2; j++) for (int k = cK - 1; k <= cK + 1; k++)
    if (i >= 0 && i < MapState::W && j >= 0 && j < MapState::H && k >= 0 && k < MapState::D) st->valid.push_back(i + MapState::W * j + MapState::W * MapState::H * k);
  std::vector<P4> corner_from_map, surf_from_map;
  for (int ind : st->valid) {
    corner_from_map.insert(corner_from_map.end(), st->corner[ind].begin(), st->corner[ind].end());
    surf_from_map.insert(surf_from_map.end(), st->surf[ind].begin(), st->surf[ind].end());
  }
  st->from_map_corner = (int)corner_from_map.size();
  st->from_map_surf = (int)surf_from_map.size();
  // ---- down-sample the incoming clouds (:542-550)
  voxel_filter(corner_last, st->line_res, canonical, &st->corner_stack);
  voxel_filter(surf_last, st->plane_res, canonical, &st->surf_stack);
  st->corner_num[0] = st->corner_num[1] = st->surf_num[0] = st->surf_num[1] = 0;
  st->lm[0] = st->lm[1] = LmSummary();
  st->edges.clear(); st->norms.clear();

  if (st->from_map_corner > 10 && st->from_map_surf > 50) {                          // :554
    KnnIndex tree_corner, tree_surf;
    tree_corner.build(corner_from_map);                                                // :558-559
    tree_surf.build(surf_from_map);
    for (int iter = 0; iter < 2; ++iter) {                                             // :562
      st->edges.clear(); st->norms.clear();
      int idx[5]; float d2[5];
      for (size_t i = 0; i < st->corner_stack.size(); ++i) {                           // :576-640
        const P4 ori = st->corner_stack[i];
        const P4 sel = associate_to_map(ori, par);
        const int n = tree_corner.query(sel, 5, cfg.nn_brute != 0, idx, d2);
        if (n >= 5) log_decision(kDecMapKnn, d2[4], 1.0);
        if (n < 5 || !(d2[4] < 1.0)) continue;
        V3d near[5], center{0, 0, 0};
        for (int j = 0; j < 5; ++j) { near[j] = V3d{corner_from_map[idx[j]].x, corner_from_map[idx[j]].y, corner_from_map[idx[j]].z}; center = center + near[j]; }
        center = V3d{center.x / 5.0, center.y / 5.0, center.z / 5.0};
        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < 5; ++j) {
          const V3d z = near[j] - center;
          const double zz[3] = {z.x, z.y, z.z};
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a * 3 + b] = cov[a * 3 + b] + zz[a] * zz[b];
        }
        double vals[3], vecs[9];
        sym_eigen3(cov, vals, vecs);
        log_decision(kDecEigRatio, vals[2], 3 * vals[1]);
        if (vals[2] > 3 * vals[1]) {                                                   // :611
          const V3d dir{vecs[0 * 3 + 2], vecs[1 * 3 + 2], vecs[2 * 3 + 2]};
          EdgeRec e;
          e.cp = V3d{ori.x, ori.y, ori.z};
          e.a = 0.1 * dir + center;
          e.b = -0.1 * dir + center;
          e.query = (int)i;
          st->edges.push_back(e);
        }
      }
      for (size_t i = 0; i < st->surf_stack.size(); ++i) {                             // :642-706
        const P4 ori = st->surf_stack[i];
        const P4 sel = associate_to_map(ori, par);
        const int n = tree_surf.query(sel, 5, cfg.nn_brute != 0, idx, d2);
        if (n >= 5) log_decision(kDecMapKnn, d2[4], 1.0);
        if (n < 5 || !(d2[4] < 1.0)) continue;
        double A[15], B[5] = {-1, -1, -1, -1, -1}, x[3];
        for (int j = 0; j < 5; ++j) { A[j * 3 + 0] = surf_from_map[idx[j]].x; A[j * 3 + 1] = surf_from_map[idx[j]].y; A[j * 3 + 2] = surf_from_map[idx[j]].z; }
        lstsq_5x3(A, B, x);
        V3d nrm{x[0], x[1], x[2]};
        const double len = std::sqrt(dot(nrm, nrm));
        const double negative_OA_dot_norm = 1 / len;
        nrm = V3d{nrm.x / len, nrm.y / len, nrm.z / len};
        bool valid = true;
        for (int j = 0; j < 5; ++j) {
          const P4& p = surf_from_map[idx[j]];
          log_decision(kDecPlaneFit, std::fabs(nrm.x * p.x + nrm.y * p.y + nrm.z * p.z + negative_OA_dot_norm), 0.2);
          if (std::fabs(nrm.x * p.x + nrm.y * p.y + nrm.z * p.z + negative_OA_dot_norm) > 0.2) { valid = false; break; }
        }
        if (valid) st->norms.push_back(NormRec{V3d{ori.x, ori.y, ori.z}, nrm, negative_OA_dot_norm, (int)i});
      }
      st->corner_num[iter] = (int)st->edges.size();
      st->surf_num[iter] = (int)st->norms.size();
      static const std::vector<PlaneRec> no_planes;
      st->lm[iter] = lm_solve(st->edges, no_planes, par, par + 4, cfg.lm_max_iterations, cfg.analytic_jacobian != 0, cfg.apply_converged_step != 0, &st->norms);
    }
  }
  {  // transformUpdate (:148-152)
    const Quatd q_w{par[0], par[1], par[2], par[3]};
    const double n2 = q_wodom_curr.x * q_wodom_curr.x + q_wodom_curr.y * q_wodom_curr.y + q_wodom_curr.z * q_wodom_curr.z + q_wodom_curr.w * q_wodom_curr.w;
    const Quatd inv{-q_wodom_curr.x / n2, -q_wodom_curr.y / n2, -q_wodom_curr.z / n2, q_wodom_curr.w / n2};   // Eigen inverse(): conjugate / squaredNorm
    st->q_wmap_wodom = qmul(q_w, inv);
    st->t_wmap_wodom = V3d{par[4], par[5], par[6]} - rotate(st->q_wmap_wodom, t_wodom_curr);
  }
  // ---- insert the new points into their cubes (:737-783)
  for (int cls = 0; cls < 2; ++cls) {
    const std::vector<P4>& stack = cls == 0 ? st->corner_stack : st->surf_stack;
    std::vector<std::vector<P4>>& arr = cls == 0 ? st->corner : st->surf;
    for (const P4& p : stack) {
      const P4 sel = associate_to_map(p, par);
      const int ci = cube_coord(sel.x, st->cenW), cj = cube_coord(sel.y, st->cenH), ck = cube_coord(sel.z, st->cenD);
      if (ci >= 0 && ci < MapState::W && cj >= 0 && cj < MapState::H && ck >= 0 && ck < MapState::D) arr[ci + MapState::W * cj + MapState::W * MapState::H * ck].push_back(sel);
    }
  }
  // ---- re-filter the valid cubes (:788-801)
  for (int ind : st->valid) {
    std::vector<P4> tmp;
    voxel_filter(st->corner[ind], st->line_res, canonical, &tmp);
    st->corner[ind].swap(tmp);
    voxel_filter(st->surf[ind], st->plane_res, canonical, &tmp);
    st->surf[ind].swap(tmp);
  }
  // ---- registered full-resolution cloud (:836-846)
  st->registered.resize(full_res.size());
  for (size_t i = 0; i < full_res.size(); ++i) st->registered[i] = associate_to_map(full_res[i], par);
  st->frame_count++;
  return 0;
}

}  // namespace orc
