/* include/aloam_mi355x.h — C ABI of libaloam_mi355x.so: the MI355X (gfx950) drop-in for A-LOAM's hot path.
 *
 * The reference (HKUST-Aerial-Robotics/A-LOAM) has no plugin / FFI interface: each stage is a ROS node whose
 * per-scan work sits in one function body over file-scope globals.  This ABI cuts exactly at those bodies
 * (SURVEY.md §8(b) "B2 in-process function seams"); every entry point below names the reference code it
 * replaces.  A ROS node keeps its subscribe / sync / publish code and calls these instead of PCL + Ceres
 * (INTEGRATION.md shows the ~50-line change per node).
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; return 0 = ok, negative = error (aloam_last_error()).
 *   - one aloam_ctx = `batch` independent sequences advanced in lock-step on ONE device and ONE HIP stream
 *     (batch = 1 is the reference's single-sensor node).  A context is not thread-safe; distinct contexts
 *     are independent (one per GPU / per process for multi-GPU; no collectives — sequences never exchange data).
 *     Every call runs on the context's device and restores the calling thread's current HIP device before returning.
 *   - points are 16-byte records {float x, y, z, w}; w = `intensity` of pcl::PointXYZI
 *     (reference include/aloam_velodyne/common.h:43).  Input records may use any stride >= 16 bytes that is a multiple of 4
 *     (16 = KITTI .bin, 32 = the PointCloud2 point_step pcl::toROSMsg<PointXYZI> produces, reference src/kittiHelper.cpp:153-154),
 *     or stride 12 = {x, y, z} only: scan registration never reads the 4th float of its input (it overwrites intensity with
 *     scanID + relTime, reference src/scanRegistration.cpp:132-133,239), so a driver may leave it off the wire (a quarter less
 *     PCIe traffic for the host-fed entries).  Not with ring_from_field, which IS the 4th float.
 *   - quaternions are (x, y, z, w) like para_q (reference src/laserOdometry.cpp:96-100).
 *   - there is NO CPU fallback: every entry point fails with ALOAM_E_HIP if the HIP runtime / a gfx950 device
 *     is unavailable.
 */
#ifndef ALOAM_MI355X_H_
#define ALOAM_MI355X_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aloam_ctx aloam_ctx;

enum {
  ALOAM_OK = 0,
  ALOAM_E_ARG = -1,        /* bad argument */
  ALOAM_E_SCAN_LINES = -2, /* n_scans not 16/32/64 without ring_from_field (reference src/scanRegistration.cpp:472-476) */
  ALOAM_E_EMPTY = -3,      /* no point of some scan survives the NaN / minimum-range filter */
  ALOAM_E_CAPACITY = -4,   /* a scan exceeds max_points, a ring exceeds max_ring_points, or (mapping) the map pool / voxel scratch was too small
                              in some step since the last aloam_synchronize: the steps have still run, the points that did not fit are
                              missing from the map */
  ALOAM_E_HIP = -5,        /* HIP runtime error / no device / internal device-side time-out */
  ALOAM_E_STATE = -6       /* call order (e.g. odometry before any registration) */
};

/* Launch-file parameters (reference launch/aloam_velodyne_HDL_64.launch:3-13) + sizing of the device buffers. */
typedef struct aloam_config {
  int n_scans;             /* `scan_line`   : 16 / 32 / 64 (reference src/scanRegistration.cpp:466)                       */
  float min_range;         /* `minimum_range` (reference src/scanRegistration.cpp:468): 0.3 VLP-16/HDL-32, 5 HDL-64          */
  int ring_from_field;     /* 0: ring from the elevation formulas (src/scanRegistration.cpp:166-205); 1: ring = int(w)     */
  int batch;               /* independent sequences processed per call                                                      */
  int max_points;          /* capacity per scan; the reference's global arrays hold 400000 (src/scanRegistration.cpp:66-69) */
  int max_ring_points;     /* capacity per ring: 2059 or 4107 (LDS sizing of the per-ring selection kernel)                 */
  int device;              /* HIP device ordinal                                                                            */
  int lm_max_iterations;   /* options.max_num_iterations = 4 (reference src/laserOdometry.cpp:496)                          */
  int outer_iterations;    /* opti_counter loop = 2 (reference src/laserOdometry.cpp:278)                                   */
  int distortion;          /* 0 = #define DISTORTION 0 (reference src/laserOdometry.cpp:59, the shipped setting); 1: every point is
                              moved / constrained with its own interpolation ratio s = (intensity - int(intensity)) / SCAN_PERIOD
                              (src/laserOdometry.cpp:115-116,376-377,474-475; src/lidarFactor.hpp:29-30,81-82)              */
} aloam_config;

/* Which cloud of a sequence (topic names of reference src/scanRegistration.cpp:480-488, src/laserOdometry.cpp:205-209). */
enum {
  ALOAM_CLOUD_FULL = 0,        /* /velodyne_cloud_2        ring-ordered laserCloud (src/scanRegistration.cpp:246-252).  The device keeps one slab per
                                  ring; the dense ring-by-ring cloud is assembled when this id is first asked for after a registration  */
  ALOAM_CLOUD_SHARP = 1,       /* /laser_cloud_sharp                                              */
  ALOAM_CLOUD_LESS_SHARP = 2,  /* /laser_cloud_less_sharp                                         */
  ALOAM_CLOUD_FLAT = 3,        /* /laser_cloud_flat                                               */
  ALOAM_CLOUD_LESS_FLAT = 4,   /* /laser_cloud_less_flat                                          */
  ALOAM_CLOUD_CORNER_LAST = 5, /* /laser_cloud_corner_last (laserCloudCornerLast after the swap)  */
  ALOAM_CLOUD_SURF_LAST = 6    /* /laser_cloud_surf_last   (laserCloudSurfLast after the swap)    */
};

typedef struct aloam_odom_stats {
  int corner_corr[2];      /* corner_correspondence (reference src/laserOdometry.cpp:382): [0] first outer iteration, [1] second — or the last one when outer_iterations > 2 */
  int plane_corr[2];       /* plane_correspondence  per outer iteration (reference src/laserOdometry.cpp:480) */
  int lm_iterations[2];    /* LM iterations executed per ceres::Solve stand-in                                 */
  int lm_successful[2];
  double initial_cost[2];
  double final_cost[2];
  int termination[2];      /* 0 max-iter, 1 parameter tol, 2 function tol, 3 gradient tol, 4 no residuals, 5 failure, 6 minimum trust-region radius (Ceres: CONVERGENCE) */
} aloam_odom_stats;

/* ---- lifetime -------------------------------------------------------------------------------------------- */
void aloam_default_config(aloam_config* cfg);                       /* HDL-64 launch values, batch 1           */
int aloam_create(const aloam_config* cfg, aloam_ctx** out);          /* replaces the nodes' global state (src/scanRegistration.cpp:60-83, src/laserOdometry.cpp:59-108) */
/* The reference runs its three stages as three processes; a node that hosts one stage only needs that stage's device buffers.
 * `stages` = any combination of ALOAM_STAGE_*; entry points of a stage that was left out fail with ALOAM_E_STATE.
 * aloam_create == aloam_create_stages(cfg, ALOAM_STAGE_ALL, out). */
enum { ALOAM_STAGE_REGISTRATION = 1, ALOAM_STAGE_ODOMETRY = 2, ALOAM_STAGE_MAPPING = 4, ALOAM_STAGE_ALL = 7 };
int aloam_create_stages(const aloam_config* cfg, int stages, aloam_ctx** out);
void aloam_destroy(aloam_ctx* ctx);
const char* aloam_last_error(const aloam_ctx* ctx);                  /* replaces printf / ROS_BREAK diagnostics */
void* aloam_stream(aloam_ctx* ctx);                                  /* the hipStream_t all work is queued on   */
int aloam_synchronize(aloam_ctx* ctx);                               /* waits + surfaces device-side error flags */
/* pcl::VoxelGrid (src/scanRegistration.cpp:402-405, src/laserMapping.cpp:543-549,793-799) sums the members of a voxel in the order an UNSTABLE
 * std::sort leaves its index vector in.  ALOAM_SUM_INPUT_ORDER (default, the throughput path) sums them in input order: centroids of three or more
 * points may differ from the reference's in their last bits (<= 4 ulp), nothing else does.  ALOAM_SUM_REFERENCE_ORDER replays libstdc++'s introsort
 * on the device (one workgroup per filter call) and sums in the order it produces: the reference's bits, about 4x slower (registration + odometry; more with mapping at large batches) -
 * a validation mode for comparing long free-running sequences with the reference's own output.  Takes effect from the next call; not to be changed mid-sequence. */
enum { ALOAM_SUM_INPUT_ORDER = 0, ALOAM_SUM_REFERENCE_ORDER = 1 };
int aloam_set_voxel_sum_order(aloam_ctx* ctx, int order);

/* ---- stage 1: body of laserCloudHandler (reference src/scanRegistration.cpp:127-411) ----------------------- */
/* Host input: scans[b] points to n_in[b] records of stride_bytes.  Blocking w.r.t. the host buffers only.    */
int aloam_scan_register(aloam_ctx* ctx, const void* const* scans, const int* n_in, int stride_bytes);
/* Device-resident input: sequence b starts at d_scans + b * seq_stride_bytes.  Fully asynchronous.           */
int aloam_scan_register_device(aloam_ctx* ctx, const void* d_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes);

/* Host-resident batch in ONE buffer (sequence b at h_scans + b * seq_stride_bytes, n_in[b] * stride_bytes readable bytes each;
 * rows 0 .. batch-2 are copied with the batch-wide maximum length, which stays inside the buffer because another row follows):
 * what a driver thread that receives the
 * sensor messages (reference src/scanRegistration.cpp:114-133) hands over.  One batched H2D copy per call on a dedicated copy
 * stream into one of two device slabs, so the copy of call k + 1 runs under the kernels of call k; asynchronous when the buffer is
 * pinned (hipHostMalloc / hipHostRegister).  The buffer must stay unmodified until aloam_input_consumed() or aloam_synchronize(). */
int aloam_scan_register_host(aloam_ctx* ctx, const void* h_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes);
int aloam_input_consumed(aloam_ctx* ctx);                            /* waits until every host buffer handed over so far has been copied and read */

/* ---- stage 2: odometry main-loop body (reference src/laserOdometry.cpp:265-506,554-568) -------------------- */
int aloam_odometry_step(aloam_ctx* ctx);                             /* asynchronous; all sequences             */

/* ---- throughput entry: stage 1 + stage 2 for one sweep of every sequence, asynchronous -------------------- */
int aloam_process_device(aloam_ctx* ctx, const void* d_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes);

int aloam_process_host(aloam_ctx* ctx, const void* h_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes);   /* same from a host buffer (see aloam_scan_register_host) */

/* ---- stage 3: scan-to-map refinement, body of process() (reference src/laserMapping.cpp:231-893), no frame dropping ---- */
/* Replaces the node's globals (cube arrays laserCloudCornerArray / SurfArray[4851], q_wmap_wodom, t_wmap_wodom, `parameters`,
 * laserCloudCen*; src/laserMapping.cpp:72-116) and reads the launch parameters mapping_line_resolution / mapping_plane_resolution
 * (:898-905).  pool_points = capacity of the device-resident map per sequence and feature class.  Call once, before the first step. */
/* pool_points is where the map STARTS: the reference's cubes are std::vectors that grow as long as the sensor travels
 * (src/laserMapping.cpp:737-783), so the pools (one per sequence and class; cubes grow by doubling inside it, the pool is compacted when
 * fragmented; one cube may hold up to the whole pool) are doubled - between steps, contents moved, at aloam_mapping_step - whenever the
 * live points plus what the queued steps can add would no longer fit, up to aloam_mapping_set_pool_limit (default 2^26 points, or
 * whatever the device memory holds).  Only at that ceiling do points get dropped: such frames are counted on the device and the first
 * aloam_synchronize after one returns ALOAM_E_CAPACITY once (however many asynchronous steps were queued in between), then ALOAM_OK
 * again until it happens anew. */
int aloam_mapping_enable(aloam_ctx* ctx, float mapping_line_resolution, float mapping_plane_resolution, int pool_points);
int aloam_mapping_set_pool_limit(aloam_ctx* ctx, int max_pool_points);   /* ceiling of the pool growth, per sequence and class; before or after enable */
/* out: current pool_points, growths so far, the limit, live points of the fullest (sequence, class) pool after the last finished step */
int aloam_get_map_pool_info(aloam_ctx* ctx, int out[4]);
/* One frame for every sequence, asynchronous.  Consumes what the odometry node publishes for the frame — /laser_cloud_corner_last,
 * /laser_cloud_surf_last, /velodyne_cloud_3, /laser_odom_to_init (src/laserOdometry.cpp:508-591) — straight from the context
 * (call after aloam_odometry_step / aloam_process_device), or as injected through aloam_set_last / aloam_set_full_cloud / aloam_set_state. */
int aloam_mapping_step(aloam_ctx* ctx);
int aloam_set_full_cloud(aloam_ctx* ctx, int seq, const float* cloud_xyzw, int n);          /* /velodyne_cloud_3 (src/laserMapping.cpp:189-194) */
/* State injection, mapping node: laserCloudCornerArray / SurfArray (feature_class 0 / 1) of one sequence replaced by n_cubes cubes
 * (window indices i + 21 j + 441 k as src/laserMapping.cpp:527, their populations, their points back to back), and
 * laserCloudCenWidth / Height / Depth, q_wmap_wodom, t_wmap_wodom, frameCount (src/laserMapping.cpp:72-74,84-91,115-116). */
int aloam_set_map(aloam_ctx* ctx, int seq, int feature_class, const int* cube_ids, const int* counts, int n_cubes, const float* points_xyzw);
int aloam_set_map_frame(aloam_ctx* ctx, int seq, const int cen[3], const double q_wmap_wodom[4], const double t_wmap_wodom[3], int frame_count);
/* /aft_mapped_to_init pose = q_w_curr, t_w_curr (src/laserMapping.cpp:851-863) and the map<-odom correction (:148-152) */
int aloam_get_map_pose(aloam_ctx* ctx, int seq, double q_w_curr[4], double t_w_curr[3], double q_wmap_wodom[4], double t_wmap_wodom[3]);
/* laserCloudCenWidth/Height/Depth, frameCount, submap sizes (corner, surf), stack sizes (corner, surf), factors per iteration
 * (corner[2], surf[2]), LM iterations[2], termination of the first solve, pool compactions so far */
int aloam_get_map_info(aloam_ctx* ctx, int seq, int out[16]);
int aloam_map_cube_counts(aloam_ctx* ctx, int seq, int feature_class, int* out_4851);      /* points per cube of the 21 x 21 x 11 window */
int aloam_get_map_cube(aloam_ctx* ctx, int seq, int feature_class, int cube, float* out_xyzw, int cap_points);   /* laserCloud*Array[cube] */
enum { ALOAM_MAP_REGISTERED = 2, ALOAM_MAP_CORNER_STACK = 3, ALOAM_MAP_SURF_STACK = 4 };    /* /velodyne_cloud_registered (:836-846); laserCloud*Stack (:542-550) */
int aloam_get_map_cloud(aloam_ctx* ctx, int seq, int which, float* out_xyzw, int cap_points);

/* ---- results (each synchronises the stream) ---------------------------------------------------------------- */
int aloam_cloud_size(aloam_ctx* ctx, int seq, int which);            /* replaces cloud.points.size()            */
int aloam_get_cloud(aloam_ctx* ctx, int seq, int which, float* out_xyzw, int cap_points);  /* what pcl::toROSMsg publishes (src/scanRegistration.cpp:413-441, src/laserOdometry.cpp:574-590) */
int aloam_get_pose(aloam_ctx* ctx, int seq, double q_w_curr[4], double t_w_curr[3], double q_last_curr[4], double t_last_curr[3]); /* /laser_odom_to_init (src/laserOdometry.cpp:511-522) + para_q / para_t */
int aloam_get_odom_stats(aloam_ctx* ctx, int seq, aloam_odom_stats* out);

/* ---- teacher forcing / state injection (what the topic hand-over between the nodes allows) ----------------- */
int aloam_set_features(aloam_ctx* ctx, int seq, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp,
                       const float* flat, int n_flat, const float* less_flat, int n_less_flat);   /* the 4 feature topics (src/laserOdometry.cpp:243-258) */
int aloam_set_last(aloam_ctx* ctx, int seq, const float* corner_last, int n_corner, const float* surf_last, int n_surf); /* laserCloudCornerLast / SurfLast + kd-tree input (src/laserOdometry.cpp:554-568) */
int aloam_set_state(aloam_ctx* ctx, int seq, const double para_q[4], const double para_t[3], const double q_w_curr[4],
                    const double t_w_curr[3]);                       /* src/laserOdometry.cpp:93-98              */
int aloam_set_system_inited(aloam_ctx* ctx, int inited);             /* systemInited (src/laserOdometry.cpp:69,267-271) */

/* ---- intermediate arrays, for parity tests ----------------------------------------------------------------- */
/* cloudCurvature / cloudLabel are kept by the aloam_scan_register* entries only; after aloam_process_device / aloam_process_host
 * (which skip those 5 bytes per point) the two getters fail with ALOAM_E_STATE. */
int aloam_get_ring_ranges(aloam_ctx* ctx, int seq, int* start, int* count);     /* scanStartInd-5 / ring sizes (src/scanRegistration.cpp:246-252) */
int aloam_get_curvature(aloam_ctx* ctx, int seq, float* out, int cap);          /* cloudCurvature (src/scanRegistration.cpp:66,262)               */
int aloam_get_labels(aloam_ctx* ctx, int seq, int* out, int cap);               /* cloudLabel (src/scanRegistration.cpp:69)                       */
/* correspondences of the last outer iteration: edges 9 floats (cp,a,b), planes 12 floats (cp,j,l,m), plus the
 * index of the query feature each belongs to (src/laserOdometry.cpp:365-381,460-479). */
int aloam_get_correspondences(aloam_ctx* ctx, int seq, float* edges, int cap_edges, int* n_edges, int* edge_query,
                              float* planes, int cap_planes, int* n_planes, int* plane_query);

/* How the ring ids of the clouds the last aloam_odometry_step searched (laserCloudCornerLast, laserCloudSurfLast of that step) are ordered, as
 * found when their kd-tree stand-ins were built (src/laserOdometry.cpp:567-568): 0 = int(intensity) never decreases with the index; 1 = it decreases, but never by more than 2 below an earlier
 * value (sweeps whose first ray had no return: relTime < 0 for the points before it, src/scanRegistration.cpp:211-214,239) - the reference's
 * neighbour walks (src/laserOdometry.cpp:315-361,410-455) then still visit one index range and run on the fast path; 2 = neither (literal walks);
 * -1 = ring ids / coordinates outside the range the grids are exact for (literal search). */
int aloam_get_last_cloud_order(aloam_ctx* ctx, int seq, int out[2]);

/* ---- per-kernel timing (hipEvents on the context's stream), for bench.py's roofline object ----------------- */
int aloam_profile_enable(aloam_ctx* ctx, int on);
int aloam_profile_kernel_count(void);
const char* aloam_profile_kernel_name(int kernel);
/* accumulated since the last enable: total milliseconds, launches, algorithmic bytes moved */
int aloam_profile_get(aloam_ctx* ctx, int kernel, double* total_ms, long long* launches, double* algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif
