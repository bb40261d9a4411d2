// oracle/oracle_internal.hpp — shared state of the CPU oracle (TEST INFRASTRUCTURE ONLY).
#pragma once
#include <string>
#include <vector>

#include "aloam_oracle.h"
#include "oracle_math.hpp"

namespace orc {

struct P4 { float x, y, z, i; };   // pcl::PointXYZI payload (reference include/aloam_velodyne/common.h:43)

struct RegistrationResult {
  std::vector<P4> cloud;                 // ring-ordered "laserCloud"   (scanRegistration.cpp:246-252)
  std::vector<int> ring_start, ring_count;
  std::vector<float> curvature;          // cloudCurvature              (scanRegistration.cpp:66,262)
  std::vector<int> label, picked;        // cloudLabel / cloudNeighborPicked (scanRegistration.cpp:68-69)
  std::vector<P4> sharp, less_sharp, flat, less_flat;
};

int register_scan(const orc_config& cfg, const void* pts, int n_in, int stride_bytes, RegistrationResult* out, std::string* err);
void voxel_filter(const std::vector<P4>& in, float leaf, bool canonical, std::vector<P4>* out);
float atan2f_port(float y, float x);

// exact 1-NN over x,y,z with the f32 distance ((dx*dx+dy*dy)+dz*dz); lowest index wins ties.
struct NnIndex {
  std::vector<P4> pts;
  std::vector<int> perm;      // kd-tree: point order
  struct Node { int lo, hi, axis, left, right; float split; };
  std::vector<Node> nodes;
  void build(const std::vector<P4>& cloud);
  void query(const P4& q, bool brute, int* idx, float* d2) const;
 private:
  int build_rec(int lo, int hi);
  void search(int node, const float q[3], int* best, float* bestd) const;
};

struct EdgeRec { V3d cp, a, b; int query; double s = 1.0; };          // LidarEdgeFactor ctor args  (lidarFactor.hpp:14-16); s: interpolation ratio
struct PlaneRec { V3d cp, j, l, m; int query; double s = 1.0; };       // LidarPlaneFactor ctor args (lidarFactor.hpp:59-62)

struct NormRec { V3d cp, n; double d; int query; };   // LidarPlaneNormFactor ctor args (lidarFactor.hpp:109-111)

struct LmSummary { int iterations = 0, successful = 0, termination = 0; double initial_cost = 0, final_cost = 0; };

void factor_eval_edge(const EdgeRec& e, const double q[4], const double t[3], bool analytic, double r[3], double J[18]);
void factor_eval_plane(const PlaneRec& p, const double q[4], const double t[3], bool analytic, double r[1], double J[6]);
double robust_cost(const std::vector<EdgeRec>& edges, const std::vector<PlaneRec>& planes, const double q[4], const double t[3]);
void quat_plus(const double q[4], const double delta[3], double out[4]);
LmSummary lm_solve(const std::vector<EdgeRec>& edges, const std::vector<PlaneRec>& planes, double q[4], double t[3],
                   int max_iterations, bool analytic, bool apply_converged_step, const std::vector<NormRec>* norms = nullptr);
void factor_eval_norm(const NormRec& p, const double q[4], const double t[3], bool analytic, double r[1], double J[6]);

struct OdomState {
  double para_q[4] = {0, 0, 0, 1};        // laserOdometry.cpp:97
  double para_t[3] = {0, 0, 0};           // laserOdometry.cpp:98
  Quatd q_w{0, 0, 0, 1};                  // laserOdometry.cpp:93 (xyzw storage here)
  V3d t_w{0, 0, 0};                       // laserOdometry.cpp:94
  bool inited = false;                    // laserOdometry.cpp:69
  std::vector<P4> corner_last, surf_last; // laserOdometry.cpp:85-86
  NnIndex tree_corner, tree_surf;         // laserOdometry.cpp:77-78
  std::vector<EdgeRec> edges;             // last outer iteration
  std::vector<PlaneRec> planes;
  orc_odom_stats stats{};
};

int odometry_step(const orc_config& cfg, const std::vector<P4>& sharp, const std::vector<P4>& less_sharp,
                  const std::vector<P4>& flat, const std::vector<P4>& less_flat, OdomState* st, std::string* err);

// ---- scan-to-map refinement (reference src/laserMapping.cpp) -----------------------------------------------
struct KnnIndex {                       // exact k-NN, ascending (f32 distance, index): pcl::KdTreeFLANN::nearestKSearch stand-in
  NnIndex tree;
  void build(const std::vector<P4>& cloud) { tree.build(cloud); }
  int query(const P4& q, int k, bool brute, int* idx, float* d2) const;
};

struct MapState {
  static constexpr int W = 21, H = 21, D = 11, NUM = W * H * D;       // laserMapping.cpp:75-80
  int cenW = 10, cenH = 10, cenD = 5;                                 // :72-74
  float line_res = 0.4f, plane_res = 0.8f;                            // :900-901
  std::vector<std::vector<P4>> corner, surf;                          // laserCloudCornerArray / SurfArray (:102-103)
  double parameters[7] = {0, 0, 0, 1, 0, 0, 0};                       // q_w_curr (xyzw), t_w_curr (:109-111)
  Quatd q_wmap_wodom{0, 0, 0, 1};                                     // :115-116
  V3d t_wmap_wodom{0, 0, 0};
  std::vector<P4> registered;                                         // /velodyne_cloud_registered payload (:836-846)
  std::vector<P4> corner_stack, surf_stack;                           // down-sampled inputs of the last frame (:542-550)
  std::vector<int> valid;                                             // laserCloudValidInd of the last frame
  std::vector<EdgeRec> edges;                                         // factors of the last iteration
  std::vector<NormRec> norms;
  int frame_count = 0;
  int corner_num[2] = {0, 0}, surf_num[2] = {0, 0};                   // factors per iteration
  int from_map_corner = 0, from_map_surf = 0;
  LmSummary lm[2];
  MapState() : corner(NUM), surf(NUM) {}
};
// ---- decision log (tests/test_decision_margins.py): when enabled, every threshold decision of the path records the quantity it
// compared and the threshold, so that a test can ask how far the data sits from the points where a last-bit difference of a
// third-party routine (Eigen's eigen-solver / QR, FLANN's distance sum) or of the device arithmetic could flip a decision.
enum DecisionKind { kDecCurvCorner = 0,   // cloudCurvature > 0.1      (reference src/scanRegistration.cpp:296)
                    kDecCurvFlat = 1,     // cloudCurvature < 0.1      (:351)
                    kDecGap = 2,          // neighbour step^2 > 0.05   (:324,:336,:371,:383)
                    kDecRange = 3,        // x^2+y^2+z^2 < minimum_range^2 (:99)
                    kDecOdomNN = 4,       // pointSearchSqDis[0] < DISTANCE_SQ_THRESHOLD = 25 (reference src/laserOdometry.cpp:305,393)
                    kDecMapKnn = 5,       // pointSearchSqDis[4] < 1.0 (reference src/laserMapping.cpp:582,650)
                    kDecEigRatio = 6,     // saes.eigenvalues()[2] > 3 * saes.eigenvalues()[1] (:611)
                    kDecPlaneFit = 7,     // fabs(norm . p + negative_OA_dot_norm) > 0.2 (:672-681)
                    // the decisions of ceres::Solve's trust-region loop that a different dense solve (Eigen's householderQr against this
                    // repo's Householder / the device's Cholesky) could flip; value / threshold as compared (reference src/laserOdometry.cpp:494-499)
                    kDecLmParamTol = 8,   // |x - x_candidate| <= parameter_tolerance (|x| + parameter_tolerance)
                    kDecLmFuncTol = 9,    // |cost - cost_candidate| <= function_tolerance cost
                    kDecLmAccept = 10,    // relative decrease > min_relative_decrease = 1e-3
                    kDecLmGradTol = 11,   // gradient max-norm <= gradient_tolerance = 1e-10
                    kDecLmModel = 12 };   // model_cost_change > 0 (step validity)
struct Decision { int kind; double value, threshold; };
extern thread_local std::vector<Decision>* g_decision_log;
inline void log_decision(int kind, double value, double threshold) { if (g_decision_log) g_decision_log->push_back(Decision{kind, value, threshold}); }

int mapping_step(const orc_config& cfg, MapState* st, const double q_wodom[4], const double t_wodom[3], const std::vector<P4>& corner_last,
                 const std::vector<P4>& surf_last, const std::vector<P4>& full_res);
void sym_eigen3(const double A[9], double vals[3], double vecs[9]);           // SelfAdjointEigenSolver<Matrix3d> stand-in (ascending)
void lstsq_5x3(const double A[15], const double b[5], double x[3]);           // colPivHouseholderQr().solve stand-in

}  // namespace orc

struct orc_ctx {
  orc_config cfg;
  orc::RegistrationResult reg;
  orc::OdomState odom;
  orc::MapState map;
  std::string err;
};
