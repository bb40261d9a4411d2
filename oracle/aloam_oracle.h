/* oracle/aloam_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * What this is: a dependency-free, single-threaded CPU restatement of the A-LOAM hot path
 *   scan registration  : reference src/scanRegistration.cpp:85-112,127-411
 *   scan-to-scan odom  : reference src/laserOdometry.cpp:59-66,93-129,265-506,554-568
 *   residual functors  : reference src/lidarFactor.hpp:12-104
 * plus the third-party pieces those files call and that are NOT vendored in the reference tree
 * (PCL 1.8.0 KdTreeFLANN / VoxelGrid, Ceres 1.12.0 Solve, Eigen 3 — pinned only by
 * reference docker/Dockerfile:3-4), restated from their published behaviour.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or sample data and cannot be built
 * in this image (needs ROS-1, PCL, FLANN, Ceres, Eigen — none installed, no network).  The oracle
 * is therefore pinned only by (i) independent cross-checks in tests/ (scipy cKDTree, central
 * differences, dual-number autodiff vs closed form, scipy minimisers, synthetic ground truth) and
 * (ii) self-generated goldens under tests/golden/.
 *
 * Who may call this: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
 * library (libaloam_mi355x.so) never links or loads it.
 */
#ifndef ALOAM_ORACLE_H_
#define ALOAM_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct orc_config {
  int n_scans;              /* 16 / 32 / 64 (scanRegistration.cpp:466-476) or any value with ring_from_field */
  float min_range;          /* MINIMUM_RANGE (scanRegistration.cpp:83,468) */
  int ring_from_field;      /* 0: elevation formulas (scanRegistration.cpp:166-205); 1: ring = int(point[3]) (config 4: 128 rows) */
  int canonical_order;      /* 1: curvature ties broken by index, voxel members summed in input order (what the HIP path does);
                               0: std::sort exactly as scanRegistration.cpp:288 / PCL voxel_grid.hpp (libstdc++ introsort tie order) */
  int nn_brute;             /* 0: exact kd-tree (KdTreeFLANN stand-in); 1: brute force */
  int analytic_jacobian;    /* 0: dual-number autodiff of the functors as written (what Ceres does); 1: closed form */
  int apply_converged_step; /* 0: Ceres >= 1.12 loop order (tolerances tested before the step is applied) */
  int lm_max_iterations;    /* 4  (laserOdometry.cpp:496) */
  int outer_iterations;     /* 2  (laserOdometry.cpp:278) */
} orc_config;

typedef struct orc_odom_stats {
  int corner_corr[2];       /* corner_correspondence per outer iteration (laserOdometry.cpp:382) */
  int plane_corr[2];        /* plane_correspondence  per outer iteration (laserOdometry.cpp:480) */
  int lm_iterations[2];     /* LM iterations executed (successful + rejected + invalid) */
  int lm_successful[2];
  double initial_cost[2];
  double final_cost[2];
  int termination[2];       /* 0 max-iter, 1 param tol, 2 function tol, 3 gradient tol, 4 no residuals, 5 failure */
} orc_odom_stats;

enum { ORC_CLOUD_FULL = 0, ORC_CLOUD_SHARP = 1, ORC_CLOUD_LESS_SHARP = 2, ORC_CLOUD_FLAT = 3, ORC_CLOUD_LESS_FLAT = 4,
       ORC_CLOUD_CORNER_LAST = 5, ORC_CLOUD_SURF_LAST = 6 };

void orc_default_config(orc_config* cfg);
orc_ctx* orc_create(const orc_config* cfg);
void orc_destroy(orc_ctx* ctx);
const char* orc_last_error(const orc_ctx* ctx);

/* Stage 1: body of laserCloudHandler (scanRegistration.cpp:127-411).  pts = n_in records of
 * stride_bytes (>= 16), each starting with float x,y,z,(field).  Returns 0, or <0 on error. */
int orc_scan_register(orc_ctx* ctx, const void* pts, int n_in, int stride_bytes);
int orc_cloud_size(const orc_ctx* ctx, int which);
int orc_get_cloud(const orc_ctx* ctx, int which, float* out_xyzi, int cap_points);
/* per ring: offset of its first point in the ring-ordered cloud and its point count */
int orc_get_ring_ranges(const orc_ctx* ctx, int* start, int* count);
/* per point of the ring-ordered cloud (valid where the reference writes them, 0 elsewhere) */
int orc_get_curvature(const orc_ctx* ctx, float* out, int cap);
int orc_get_labels(const orc_ctx* ctx, int* out, int cap);
int orc_get_picked(const orc_ctx* ctx, int* out, int cap);

/* Teacher forcing */
int orc_set_features(orc_ctx* ctx, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp,
                     const float* flat, int n_flat, const float* less_flat, int n_less_flat);
int orc_set_last(orc_ctx* ctx, const float* corner_last, int n_corner, const float* surf_last, int n_surf);
int orc_set_state(orc_ctx* ctx, const double para_q[4], const double para_t[3], const double q_w[4], const double t_w[3],
                  int system_inited);

/* Stage 2: one pass of the odometry main loop body (laserOdometry.cpp:265-506,554-568) on the
 * features currently held by the context. */
int orc_odometry_step(orc_ctx* ctx);
int orc_get_pose(const orc_ctx* ctx, double q_w_curr[4], double t_w_curr[3], double q_last_curr[4], double t_last_curr[3]);
int orc_get_odom_stats(const orc_ctx* ctx, orc_odom_stats* out);
/* correspondence records of the LAST outer iteration: edges 9 doubles (cp,a,b), planes 12 doubles (cp,j,l,m) */
int orc_get_correspondences(const orc_ctx* ctx, double* edges, int cap_edges, int* n_edges, double* planes, int cap_planes, int* n_planes,
                            int* edge_query_index, int* plane_query_index);

/* Building blocks exposed for unit tests */
int orc_voxel_filter(const float* xyzi, int n, float leaf, int canonical, float* out_xyzi, int cap);
int orc_nn_search(const float* target_xyzi, int n_target, const float* query_xyzi, int n_query, int brute, int* idx, float* d2);
/* kind 0: LidarEdgeFactor consts = cp,a,b (9), 3 residuals; kind 1: LidarPlaneFactor consts = cp,j,l,m (12), 1 residual.
 * J is rows x 6 row-major in the tangent space (delta[3], dt[3]) of EigenQuaternionParameterization. */
int orc_factor_eval(int kind, const double* consts, const double q_xyzw[4], const double t[3], int analytic, double* r, double* J);
/* robust cost 1/2 sum rho(|r|^2) at (q,t) */
double orc_cost(int n_edges, const double* edges, int n_planes, const double* planes, const double q_xyzw[4], const double t[3]);
int orc_lm_solve(int n_edges, const double* edges, int n_planes, const double* planes, double q_xyzw[4], double t[3],
                 int max_iterations, int analytic, int apply_converged_step, int* iterations, int* successful,
                 double* initial_cost, double* final_cost, int* termination);
float orc_atan2f_port(float y, float x);   /* FDLIBM-style atan2f the HIP path uses; tests compare it with libm's atan2f */
void orc_quat_plus(const double q_xyzw[4], const double delta[3], double out[4]);

#ifdef __cplusplus
}
#endif
#endif
