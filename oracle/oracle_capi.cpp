// oracle/oracle_capi.cpp — extern "C" surface of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
#include <algorithm>
#include <cstring>

#include "oracle_internal.hpp"

using orc::P4;

static void copy_cloud(const std::vector<P4>& v, float* out, int cap) {
  const int n = std::min((int)v.size(), cap);
  if (n > 0) std::memcpy(out, v.data(), (size_t)n * sizeof(P4));
}
static std::vector<P4> to_vec(const float* p, int n) {
  std::vector<P4> v((size_t)std::max(n, 0));
  if (n > 0) std::memcpy(v.data(), p, (size_t)n * sizeof(P4));
  return v;
}
static const std::vector<P4>* pick(const orc_ctx* c, int which) {
  switch (which) {
    case ORC_CLOUD_FULL: return &c->reg.cloud;
    case ORC_CLOUD_SHARP: return &c->reg.sharp;
    case ORC_CLOUD_LESS_SHARP: return &c->reg.less_sharp;
    case ORC_CLOUD_FLAT: return &c->reg.flat;
    case ORC_CLOUD_LESS_FLAT: return &c->reg.less_flat;
    case ORC_CLOUD_CORNER_LAST: return &c->odom.corner_last;
    case ORC_CLOUD_SURF_LAST: return &c->odom.surf_last;
    default: return nullptr;
  }
}

extern "C" {

void orc_default_config(orc_config* cfg) {
  cfg->n_scans = 64;
  cfg->min_range = 5.0f;          // launch/aloam_velodyne_HDL_64.launch minimum_range
  cfg->ring_from_field = 0;
  cfg->canonical_order = 1;
  cfg->nn_brute = 0;
  cfg->analytic_jacobian = 0;
  cfg->apply_converged_step = 0;
  cfg->lm_max_iterations = 4;
  cfg->outer_iterations = 2;
  cfg->distortion = 0;
}

orc_ctx* orc_create(const orc_config* cfg) {
  orc_ctx* c = new orc_ctx();
  c->cfg = *cfg;
  return c;
}
void orc_destroy(orc_ctx* ctx) { delete ctx; }
const char* orc_last_error(const orc_ctx* ctx) { return ctx->err.c_str(); }

int orc_scan_register(orc_ctx* ctx, const void* pts, int n_in, int stride_bytes) {
  return orc::register_scan(ctx->cfg, pts, n_in, stride_bytes, &ctx->reg, &ctx->err);
}
int orc_cloud_size(const orc_ctx* ctx, int which) {
  const std::vector<P4>* v = pick(ctx, which);
  return v ? (int)v->size() : -1;
}
int orc_get_cloud(const orc_ctx* ctx, int which, float* out, int cap) {
  const std::vector<P4>* v = pick(ctx, which);
  if (!v) return -1;
  copy_cloud(*v, out, cap);
  return (int)v->size();
}
int orc_get_ring_ranges(const orc_ctx* ctx, int* start, int* count) {
  for (size_t r = 0; r < ctx->reg.ring_start.size(); ++r) { start[r] = ctx->reg.ring_start[r]; count[r] = ctx->reg.ring_count[r]; }
  return (int)ctx->reg.ring_start.size();
}
int orc_get_curvature(const orc_ctx* ctx, float* out, int cap) {
  const int n = std::min((int)ctx->reg.curvature.size(), cap);
  std::copy(ctx->reg.curvature.begin(), ctx->reg.curvature.begin() + n, out);
  return (int)ctx->reg.curvature.size();
}
int orc_get_labels(const orc_ctx* ctx, int* out, int cap) {
  const int n = std::min((int)ctx->reg.label.size(), cap);
  std::copy(ctx->reg.label.begin(), ctx->reg.label.begin() + n, out);
  return (int)ctx->reg.label.size();
}
int orc_get_picked(const orc_ctx* ctx, int* out, int cap) {
  const int n = std::min((int)ctx->reg.picked.size(), cap);
  std::copy(ctx->reg.picked.begin(), ctx->reg.picked.begin() + n, out);
  return (int)ctx->reg.picked.size();
}

int orc_set_features(orc_ctx* ctx, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp,
                     const float* flat, int n_flat, const float* less_flat, int n_less_flat) {
  ctx->reg.sharp = to_vec(sharp, n_sharp);
  ctx->reg.less_sharp = to_vec(less_sharp, n_less_sharp);
  ctx->reg.flat = to_vec(flat, n_flat);
  ctx->reg.less_flat = to_vec(less_flat, n_less_flat);
  return 0;
}
int orc_set_last(orc_ctx* ctx, const float* corner_last, int n_corner, const float* surf_last, int n_surf) {
  ctx->odom.corner_last = to_vec(corner_last, n_corner);
  ctx->odom.surf_last = to_vec(surf_last, n_surf);
  ctx->odom.tree_corner.build(ctx->odom.corner_last);
  ctx->odom.tree_surf.build(ctx->odom.surf_last);
  return 0;
}
int orc_set_state(orc_ctx* ctx, const double para_q[4], const double para_t[3], const double q_w[4], const double t_w[3],
                  int system_inited) {
  for (int k = 0; k < 4; ++k) ctx->odom.para_q[k] = para_q[k];
  for (int k = 0; k < 3; ++k) ctx->odom.para_t[k] = para_t[k];
  ctx->odom.q_w = orc::Quatd{q_w[0], q_w[1], q_w[2], q_w[3]};
  ctx->odom.t_w = orc::V3d{t_w[0], t_w[1], t_w[2]};
  ctx->odom.inited = system_inited != 0;
  return 0;
}

int orc_odometry_step(orc_ctx* ctx) {
  return orc::odometry_step(ctx->cfg, ctx->reg.sharp, ctx->reg.less_sharp, ctx->reg.flat, ctx->reg.less_flat, &ctx->odom, &ctx->err);
}
int orc_get_pose(const orc_ctx* ctx, double q_w[4], double t_w[3], double q_lc[4], double t_lc[3]) {
  q_w[0] = ctx->odom.q_w.x; q_w[1] = ctx->odom.q_w.y; q_w[2] = ctx->odom.q_w.z; q_w[3] = ctx->odom.q_w.w;
  t_w[0] = ctx->odom.t_w.x; t_w[1] = ctx->odom.t_w.y; t_w[2] = ctx->odom.t_w.z;
  for (int k = 0; k < 4; ++k) q_lc[k] = ctx->odom.para_q[k];
  for (int k = 0; k < 3; ++k) t_lc[k] = ctx->odom.para_t[k];
  return 0;
}
int orc_get_odom_stats(const orc_ctx* ctx, orc_odom_stats* out) { *out = ctx->odom.stats; return 0; }

int orc_get_correspondences(const orc_ctx* ctx, double* edges, int cap_edges, int* n_edges, double* planes, int cap_planes,
                            int* n_planes, int* edge_query_index, int* plane_query_index) {
  const auto& E = ctx->odom.edges;
  const auto& P = ctx->odom.planes;
  *n_edges = (int)E.size();
  *n_planes = (int)P.size();
  for (int i = 0; i < (int)E.size() && i < cap_edges; ++i) {
    const double v[9] = {E[i].cp.x, E[i].cp.y, E[i].cp.z, E[i].a.x, E[i].a.y, E[i].a.z, E[i].b.x, E[i].b.y, E[i].b.z};
    std::memcpy(edges + (size_t)i * 9, v, sizeof(v));
    if (edge_query_index) edge_query_index[i] = E[i].query;
  }
  for (int i = 0; i < (int)P.size() && i < cap_planes; ++i) {
    const double v[12] = {P[i].cp.x, P[i].cp.y, P[i].cp.z, P[i].j.x, P[i].j.y, P[i].j.z,
                          P[i].l.x, P[i].l.y, P[i].l.z, P[i].m.x, P[i].m.y, P[i].m.z};
    std::memcpy(planes + (size_t)i * 12, v, sizeof(v));
    if (plane_query_index) plane_query_index[i] = P[i].query;
  }
  return 0;
}

int orc_voxel_filter(const float* xyzi, int n, float leaf, int canonical, float* out_xyzi, int cap) {
  std::vector<P4> out;
  orc::voxel_filter(to_vec(xyzi, n), leaf, canonical != 0, &out);
  copy_cloud(out, out_xyzi, cap);
  return (int)out.size();
}

int orc_nn_search(const float* target_xyzi, int n_target, const float* query_xyzi, int n_query, int brute, int* idx, float* d2) {
  orc::NnIndex index;
  index.build(to_vec(target_xyzi, n_target));
  for (int i = 0; i < n_query; ++i) {
    P4 q;
    std::memcpy(&q, query_xyzi + (size_t)i * 4, sizeof(P4));
    index.query(q, brute != 0, &idx[i], &d2[i]);
  }
  return 0;
}

static std::vector<orc::EdgeRec> edges_from(const double* e, int n) {
  std::vector<orc::EdgeRec> v(n);
  for (int i = 0; i < n; ++i) {
    const double* p = e + (size_t)i * 9;
    v[i].cp = {p[0], p[1], p[2]}; v[i].a = {p[3], p[4], p[5]}; v[i].b = {p[6], p[7], p[8]}; v[i].query = i;
  }
  return v;
}
static std::vector<orc::PlaneRec> planes_from(const double* e, int n) {
  std::vector<orc::PlaneRec> v(n);
  for (int i = 0; i < n; ++i) {
    const double* p = e + (size_t)i * 12;
    v[i].cp = {p[0], p[1], p[2]}; v[i].j = {p[3], p[4], p[5]}; v[i].l = {p[6], p[7], p[8]}; v[i].m = {p[9], p[10], p[11]}; v[i].query = i;
  }
  return v;
}

int orc_factor_eval(int kind, const double* consts, const double q[4], const double t[3], int analytic, double* r, double* J) {
  if (kind == 0) { orc::factor_eval_edge(edges_from(consts, 1)[0], q, t, analytic != 0, r, J); return 3; }
  if (kind == 1) { orc::factor_eval_plane(planes_from(consts, 1)[0], q, t, analytic != 0, r, J); return 1; }
  return -1;
}
int orc_factor_eval_s(int kind, const double* consts, double s, const double q[4], const double t[3], double* r, double* J) {
  if (kind == 0) { orc::EdgeRec e = edges_from(consts, 1)[0]; e.s = s; orc::factor_eval_edge(e, q, t, false, r, J); return 3; }
  if (kind == 1) { orc::PlaneRec p = planes_from(consts, 1)[0]; p.s = s; orc::factor_eval_plane(p, q, t, false, r, J); return 1; }
  return -1;
}
double orc_cost(int n_edges, const double* edges, int n_planes, const double* planes, const double q[4], const double t[3]) {
  return orc::robust_cost(edges_from(edges, n_edges), planes_from(planes, n_planes), q, t);
}
int orc_lm_solve(int n_edges, const double* edges, int n_planes, const double* planes, double q[4], double t[3],
                 int max_iterations, int analytic, int apply_converged_step, int* iterations, int* successful,
                 double* initial_cost, double* final_cost, int* termination) {
  const orc::LmSummary sm = orc::lm_solve(edges_from(edges, n_edges), planes_from(planes, n_planes), q, t, max_iterations,
                                          analytic != 0, apply_converged_step != 0);
  if (iterations) *iterations = sm.iterations;
  if (successful) *successful = sm.successful;
  if (initial_cost) *initial_cost = sm.initial_cost;
  if (final_cost) *final_cost = sm.final_cost;
  if (termination) *termination = sm.termination;
  return 0;
}
float orc_atan2f_port(float y, float x) { return orc::atan2f_port(y, x); }
void orc_quat_plus(const double q[4], const double delta[3], double out[4]) { orc::quat_plus(q, delta, out); }

}  // extern "C"

// ---- scan-to-map refinement -------------------------------------------------------------------------------------
extern "C" {
int orc_map_config(orc_ctx* ctx, float line_res, float plane_res) { ctx->map.line_res = line_res; ctx->map.plane_res = plane_res; return 0; }
int orc_mapping_step(orc_ctx* ctx, const double q_wodom[4], const double t_wodom[3], const float* corner_last, int n_corner, const float* surf_last,
                     int n_surf, const float* full_res, int n_full) {
  return orc::mapping_step(ctx->cfg, &ctx->map, q_wodom, t_wodom, to_vec(corner_last, n_corner), to_vec(surf_last, n_surf), to_vec(full_res, n_full));
}
int orc_map_get_pose(const orc_ctx* ctx, double q_w_curr[4], double t_w_curr[3], double q_wmap_wodom[4], double t_wmap_wodom[3]) {
  const orc::MapState& m = ctx->map;
  for (int k = 0; k < 4; ++k) q_w_curr[k] = m.parameters[k];
  for (int k = 0; k < 3; ++k) t_w_curr[k] = m.parameters[4 + k];
  q_wmap_wodom[0] = m.q_wmap_wodom.x; q_wmap_wodom[1] = m.q_wmap_wodom.y; q_wmap_wodom[2] = m.q_wmap_wodom.z; q_wmap_wodom[3] = m.q_wmap_wodom.w;
  t_wmap_wodom[0] = m.t_wmap_wodom.x; t_wmap_wodom[1] = m.t_wmap_wodom.y; t_wmap_wodom[2] = m.t_wmap_wodom.z;
  return 0;
}
int orc_map_get_info(const orc_ctx* ctx, int out[16]) {
  const orc::MapState& m = ctx->map;
  const int v[16] = {m.cenW, m.cenH, m.cenD, m.frame_count, m.from_map_corner, m.from_map_surf, (int)m.corner_stack.size(), (int)m.surf_stack.size(),
                     m.corner_num[0], m.corner_num[1], m.surf_num[0], m.surf_num[1], m.lm[0].iterations, m.lm[1].iterations, m.lm[0].termination, m.lm[1].termination};
  std::memcpy(out, v, sizeof(v));
  return 0;
}
/* which: 0 corner cube, 1 surf cube (cube = index into the 21 x 21 x 11 window), 2 registered cloud, 3 corner stack, 4 surf stack */
static const std::vector<P4>* map_pick(const orc_ctx* c, int which, int cube) {
  switch (which) {
    case 0: return (cube >= 0 && cube < orc::MapState::NUM) ? &c->map.corner[cube] : nullptr;
    case 1: return (cube >= 0 && cube < orc::MapState::NUM) ? &c->map.surf[cube] : nullptr;
    case 2: return &c->map.registered;
    case 3: return &c->map.corner_stack;
    case 4: return &c->map.surf_stack;
    default: return nullptr;
  }
}
int orc_map_cloud_size(const orc_ctx* ctx, int which, int cube) { const std::vector<P4>* v = map_pick(ctx, which, cube); return v ? (int)v->size() : -1; }
int orc_map_get_cloud(const orc_ctx* ctx, int which, int cube, float* out, int cap) {
  const std::vector<P4>* v = map_pick(ctx, which, cube);
  if (!v) return -1;
  copy_cloud(*v, out, cap);
  return (int)v->size();
}
int orc_map_cube_counts(const orc_ctx* ctx, int cls, int* out) {     /* out[4851] */
  const auto& arr = cls == 0 ? ctx->map.corner : ctx->map.surf;
  for (int i = 0; i < orc::MapState::NUM; ++i) out[i] = (int)arr[i].size();
  return orc::MapState::NUM;
}
int orc_knn_search(const float* target_xyzi, int n_target, const float* query_xyzi, int n_query, int k, int brute, int* idx, float* d2) {
  orc::KnnIndex t;
  t.build(to_vec(target_xyzi, n_target));
  for (int i = 0; i < n_query; ++i) {
    P4 q; std::memcpy(&q, query_xyzi + 4 * (size_t)i, sizeof(P4));
    const int n = t.query(q, k, brute != 0, idx + (size_t)i * k, d2 + (size_t)i * k);
    for (int j = n; j < k; ++j) { idx[(size_t)i * k + j] = -1; d2[(size_t)i * k + j] = 0.f; }
  }
  return 0;
}
void orc_sym_eigen3(const double A[9], double vals[3], double vecs[9]) { orc::sym_eigen3(A, vals, vecs); }
void orc_lstsq_5x3(const double A[15], const double b[5], double x[3]) { orc::lstsq_5x3(A, b, x); }
}

// ---- decision log (thread-local; see oracle_internal.hpp) ----------------------------------------------------------------------
namespace orc { thread_local std::vector<Decision>* g_decision_log = nullptr; }
static thread_local std::vector<orc::Decision> t_decisions;
extern "C" void orc_decision_log(int enable) { t_decisions.clear(); orc::g_decision_log = enable ? &t_decisions : nullptr; }
extern "C" long long orc_decision_log_size(void) { return (long long)t_decisions.size(); }
extern "C" long long orc_decision_log_fetch(int* kinds, double* values, double* thresholds, long long cap) {
  const long long n = (long long)t_decisions.size() < cap ? (long long)t_decisions.size() : cap;
  for (long long i = 0; i < n; ++i) { kinds[i] = t_decisions[i].kind; values[i] = t_decisions[i].value; thresholds[i] = t_decisions[i].threshold; }
  return (long long)t_decisions.size();
}
