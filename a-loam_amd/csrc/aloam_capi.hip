// a-loam_amd/csrc/aloam_capi.hip — host side of libaloam_mi355x.so: context, device buffers, launch sequencing and
// the extern "C" surface declared in include/aloam_mi355x.h.  There is no CPU fallback anywhere in this file:
// without a HIP device every entry point fails with ALOAM_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aloam_mi355x.h"
#include "aloam_device.hpp"
#include "mapping_kernels.hpp"
#include "odometry_kernels.hpp"
#include "registration_kernels.hpp"

using namespace aloam;

namespace {
enum KernelId { K_FIND_ENDS = 0, K_CLASSIFY, K_RING_OFFSETS, K_SCATTER, K_RING_FEATURES, K_BUILD_GRIDS, K_TRANSFORM, K_ASSOC_CORNER,
                K_ASSOC_PLANE, K_SOLVE, K_ADVANCE, K_MAP_BEGIN, K_MAP_VOXEL_STACK, K_MAP_GRID, K_MAP_ASSOC, K_MAP_SOLVE, K_MAP_INSERT,
                K_MAP_VOXEL_CUBES, K_MAP_REGISTER, K_COUNT };
const char* kKernelNames[K_COUNT] = {"k_find_ends", "k_front", "k_ring_starts", "k_dense_cloud", "k_ring_features",
                                     "k_build_grids", "k_transform_queries", "k_associate[corner]", "k_associate[plane]",
                                     "k_solve", "k_advance", "map_begin", "map_voxel[stacks]", "map_grid", "map_associate", "map_solve",
                                     "map_insert", "map_voxel[cubes]", "map_register"};
struct ProfRec { int kernel; hipEvent_t e0, e1; };
constexpr int kNinSlots = 8;
}  // namespace

struct aloam_ctx {
  aloam_config cfg{};
  int stages = ALOAM_STAGE_ALL;      // which stages this context has buffers for (aloam_create_stages)
  int B = 0, cap = 0, R = 0, NB = 0, npad = 0;   // cap: points per sequence the big buffers are laid out for = max_points + padding (below)
  int max_points = 0;                   // what the caller may hand in (aloam_config.max_points)
  hipStream_t stream = nullptr;
  std::string err;
  // input staging (host-input path only)
  // two device slabs: the H2D copy of call k + 1 (copy stream) overlaps the kernels of call k (compute stream)
  char* d_in[2] = {nullptr, nullptr}; size_t d_in_bytes[2] = {0, 0};
  int in_slot = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t in_copied[2] = {}, in_consumed[2] = {}; bool in_used[2] = {false, false};
  char* h_pin = nullptr; size_t h_pin_bytes = 0;    // pinned bounce buffer for pageable callers of aloam_scan_register
  int* d_nin = nullptr;
  int* h_nin = nullptr; int h_nin_slot = 0;         // pinned ring of kNinSlots x B counts: an async H2D copy reads its slot later
  hipEvent_t nin_done[8] = {}; bool nin_used[8] = {};
  SeqMeta* d_meta = nullptr;
  float4* d_slabs = nullptr; int slab = 0;          // ring-ordered points, one slab per (sequence, ring): what k_front writes and the feature kernels read
  unsigned long long* d_front_lb = nullptr; int* d_front_ticket = nullptr;
  bool dense_valid = true;                          // d_cloud holds the dense concatenation of the current slabs (k_dense_cloud, on demand)
  int* d_ringstart = nullptr;
  float4* d_cloud = nullptr; float* d_curv = nullptr; int8_t* d_label = nullptr;
  unsigned long long* d_lookback = nullptr; unsigned reg_epoch = 0;   // ring-count granules of k_ring_features, launch counter
  int* d_ring_ticket = nullptr;                                       // per sweep: rings handed out to the workgroups of the running k_ring_features
  bool debug_arrays = false;                                         // the last registration wrote curvature / labels
  float4 *d_sharp = nullptr, *d_flat = nullptr;
  float4* d_less_sharp[2] = {nullptr, nullptr};
  float4* d_less_flat[2] = {nullptr, nullptr};
  int cur = 0;                       // which of the double buffers holds the CURRENT sweep's less-sharp / less-flat
  OdomState* d_state = nullptr;
  float4* d_grid_sorted3[2] = {nullptr, nullptr}; float4* d_grid_sorted2[2] = {nullptr, nullptr};
  int* d_grid_start3[2] = {nullptr, nullptr}; int* d_grid_start2[2] = {nullptr, nullptr};
  float4* d_grid_sorted3c[2] = {nullptr, nullptr};   // coarse level of the 3-D grid
  int* d_grid_start3c[2] = {nullptr, nullptr};
  int* d_grid_flags[2] = {nullptr, nullptr}; int* d_grid_walk[2] = {nullptr, nullptr};
  int grid_H[2] = {4096, 16384};
  EdgeRec* d_edges = nullptr; PlaneRec* d_planes = nullptr;
  float4 *d_sel_sharp = nullptr, *d_sel_flat = nullptr;
  // scan-to-map refinement (allocated by aloam_mapping_enable)
  bool map_on = false;
  long long map_err_reported = 0;    // voxel-scratch capacity events (vox counters[3]) aloam_synchronize has already returned
  std::vector<long long> map_err_seen;   // per sequence: pool capacity events (MapSeq.err_steps) already returned
  float map_line_res = 0.4f, map_plane_res = 0.8f;
  int map_pool = 0, map_H = 0, map_levels = 0, map_cube_levels = 0, map_tile_cap = 0, map_tile_bound[2] = {0, 0}, map_nsegs_max = 0;
  long long map_key_cap = 0;
  // pool growth (map_ensure_capacity): the reference's cubes are std::vectors that grow without bound (src/laserMapping.cpp:737-783)
  int map_pool_limit = 1 << 26;      // ceiling per sequence and class (aloam_mapping_set_pool_limit); ALOAM_E_CAPACITY only there
  int map_growths = 0;
  long long map_steps = 0;           // mapping steps queued so far
  int nin_max = 0;                   // largest scan handed to the last registration call (bounds what one step can add to a map)
  int inject_max = 0;                // largest cloud injected through aloam_set_last since the last mapping step (the same bound for a mapping-only context)
  volatile int* h_map_report = nullptr;   // pinned: {step, live corner, live surf, stack corner, stack surf} of the last finished step
  int* d_map_report_host = nullptr;       // the same memory as the device sees it
  int* d_map_report = nullptr; int* d_map_live = nullptr;
  hipEvent_t map_step_done[4] = {};
  MapSeq* d_mapseq = nullptr; CubeDesc* d_cubes = nullptr; float4* d_pool[2] = {nullptr, nullptr}; int* d_maptab = nullptr;
  float4* d_stack[2] = {nullptr, nullptr}; float4* d_stack_world[2] = {nullptr, nullptr}; int* d_stack_cube[2] = {nullptr, nullptr};
  int *d_addcnt = nullptr, *d_cursor = nullptr, *d_compact_flag = nullptr;
  float4* d_mgrid_sorted[2] = {nullptr, nullptr}; int* d_mgrid_start[2] = {nullptr, nullptr};
  MapEdgeRec* d_medges = nullptr; MapNormRec* d_mnorms = nullptr; float4* d_registered = nullptr; float4* d_knn = nullptr;
  int* d_vox_lists = nullptr;
  int* d_rec_tiles = nullptr; int rec_tiles_corner = 0, rec_tiles_per_seq = 0;
  VoxSeg* d_segs = nullptr; int *d_tile_seg = nullptr, *d_tile_heads = nullptr, *d_tile_pref = nullptr, *d_vox_counters = nullptr, *d_bbox = nullptr;
  unsigned long long* d_keys[2] = {nullptr, nullptr}; float4* d_voxtmp = nullptr;
  bool system_inited = false;        // reference src/laserOdometry.cpp:69
  int sum_order = 0;                 // ALOAM_SUM_INPUT_ORDER / ALOAM_SUM_REFERENCE_ORDER (aloam_set_voxel_sum_order)
  // small batches (the ROS shims run batch 1): the ~15 dependent launches of an odometry step as ONE hipGraph launch per buffer parity
  hipGraphExec_t odom_graph[2] = {nullptr, nullptr};
  bool use_graph = false;            // batch <= ALOAM_GRAPH_MAX_BATCH (environment, default 0 = off), read once at creation
  bool have_features = false;
  // profiling
  bool prof_on = false;
  bool debug_sync = false;           // environment ALOAM_DEBUG_SYNC, read once at creation
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_free;
  double prof_ms[K_COUNT] = {0};
  long long prof_launches[K_COUNT] = {0};
};

#define HIP_TRY(ctx, expr)                                                                                   \
  do {                                                                                                       \
    hipError_t e__ = (expr);                                                                                 \
    if (e__ != hipSuccess) {                                                                                 \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                       \
      return ALOAM_E_HIP;                                                                                    \
    }                                                                                                        \
  } while (0)

namespace {

template <typename T>
int dmalloc(aloam_ctx* c, T** p, size_t count) {
  HIP_TRY(c, hipMalloc((void**)p, count * sizeof(T)));
  HIP_TRY(c, hipMemsetAsync(*p, 0, count * sizeof(T), c->stream));
  return ALOAM_OK;
}

hipEvent_t prof_event(aloam_ctx* c) {
  if (!c->prof_free.empty()) { hipEvent_t e = c->prof_free.back(); c->prof_free.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct ProfScope {
  aloam_ctx* c; int k; hipEvent_t e0 = nullptr;
  ProfScope(aloam_ctx* c_, int k_) : c(c_), k(k_) {
    if (c->prof_on) { e0 = prof_event(c); (void)hipEventRecord(e0, c->stream); }
  }
  ~ProfScope() {
    if (c->prof_on) { hipEvent_t e1 = prof_event(c); (void)hipEventRecord(e1, c->stream); c->prof_pending.push_back({k, e0, e1}); }
    if (c->debug_sync) {   // ALOAM_DEBUG_SYNC=1: wait after every stage and name it, so that a device fault can be pinned on a kernel
      const hipError_t e = hipStreamSynchronize(c->stream);
      std::fprintf(stderr, "[aloam] %-22s %s\n", kKernelNames[k], e == hipSuccess ? "ok" : hipGetErrorString(e));
    }
  }
};
// Every entry point runs on the context's device whatever the calling thread's current device is, and leaves the caller's
// choice as it found it (several contexts on several devices in one process; frameworks that switch devices behind our back).
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(const aloam_ctx* c) {
    int cur = -1;
    if (c && hipGetDevice(&cur) == hipSuccess && cur != c->cfg.device && hipSetDevice(c->cfg.device) == hipSuccess) prev = cur;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

void prof_resolve(aloam_ctx* c) {
  for (ProfRec& r : c->prof_pending) {
    float ms = 0.f;
    (void)hipEventSynchronize(r.e1);
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    c->prof_ms[r.kernel] += ms;
    c->prof_launches[r.kernel] += 1;
    c->prof_free.push_back(r.e0);
    c->prof_free.push_back(r.e1);
  }
  c->prof_pending.clear();
}

RegArgs reg_args(aloam_ctx* c, const void* d_scans, long long seq_stride, int pt_stride) {
  RegArgs a{};
  a.in = (const char*)d_scans; a.seq_stride = seq_stride; a.pt_stride = pt_stride;
  a.B = c->B; a.cap = c->cap; a.R = c->R; a.NB = c->NB;
  a.ring_from_field = c->cfg.ring_from_field; a.min_range = c->cfg.min_range;
  a.meta = c->d_meta; a.slabs = c->d_slabs; a.slab = c->slab; a.front_lb = c->d_front_lb; a.front_ticket = c->d_front_ticket;
  a.ringstart = c->d_ringstart; a.cloud = c->d_cloud; a.curv = c->d_curv; a.label = c->d_label;
  a.lookback = c->d_lookback; a.epoch = c->reg_epoch; a.store_debug = c->debug_arrays ? 1 : 0; a.ring_ticket = c->d_ring_ticket;
  a.sharp = c->d_sharp; a.less_sharp = c->d_less_sharp[c->cur]; a.flat = c->d_flat; a.less_flat = c->d_less_flat[c->cur];
  return a;
}

// The dense ring-by-ring cloud (laserCloud of src/scanRegistration.cpp:246-252) is made from the slabs when a consumer of the FULL cloud asks for it.
int ensure_dense(aloam_ctx* c) {
  if (c->dense_valid) return ALOAM_OK;                  // (also: nothing registered yet, or the cloud was set from outside)
  { ProfScope p(c, K_SCATTER); launch_dense_cloud(reg_args(c, nullptr, 0, 16), c->stream); }
  HIP_TRY(c, hipGetLastError());
  c->dense_valid = true;
  return ALOAM_OK;
}

OdomArgs odom_args(aloam_ctx* c) {
  OdomArgs a{};
  a.B = c->B; a.cap = c->cap; a.R = c->R;
  a.meta = c->d_meta; a.state = c->d_state;
  a.sharp = c->d_sharp; a.flat = c->d_flat;
  a.corner_last = c->d_less_sharp[1 - c->cur]; a.surf_last = c->d_less_flat[1 - c->cur];
  for (int k = 0; k < 2; ++k) {
    a.grid_sorted3[k] = c->d_grid_sorted3[k]; a.grid_sorted2[k] = c->d_grid_sorted2[k]; a.grid_start3[k] = c->d_grid_start3[k];
    a.grid_sorted3c[k] = c->d_grid_sorted3c[k]; a.grid_start3c[k] = c->d_grid_start3c[k];
    a.grid_start2[k] = c->d_grid_start2[k];
    a.grid_flags[k] = c->d_grid_flags[k]; a.grid_walk[k] = c->d_grid_walk[k];
  }
  a.grid_H_corner = c->grid_H[0]; a.grid_H_surf = c->grid_H[1];
  a.edges = c->d_edges; a.planes = c->d_planes;
  a.sel_sharp = c->d_sel_sharp; a.sel_flat = c->d_sel_flat;
  a.lm_max_iterations = c->cfg.lm_max_iterations;
  a.distortion = c->cfg.distortion != 0;
  return a;
}

int check_seq(aloam_ctx* c, int seq) {
  if (!c) return ALOAM_E_ARG;
  if (seq < 0 || seq >= c->B) { c->err = "sequence index out of range"; return ALOAM_E_ARG; }
  return ALOAM_OK;
}

int sync_and_check(aloam_ctx* c) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ALOAM_OK;
}

int fetch_meta(aloam_ctx* c, int seq, SeqMeta* m) {
  HIP_TRY(c, hipMemcpyAsync(m, c->d_meta + seq, sizeof(SeqMeta), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ALOAM_OK;
}

// debug_arrays: also write cloudCurvature / cloudLabel (the per-point entry points aloam_get_curvature / aloam_get_labels);
// the throughput entries (aloam_process_device / aloam_process_host) leave those 5 bytes per point out.
int register_launch(aloam_ctx* c, const void* d_scans, long long seq_stride, const int* n_in, int stride_bytes, int slot = -1, bool debug_arrays = true) {
  if (!(c->stages & ALOAM_STAGE_REGISTRATION)) { c->err = "this context was created without ALOAM_STAGE_REGISTRATION"; return ALOAM_E_STATE; }
  if (stride_bytes < 12 || (stride_bytes & 3)) { c->err = "stride_bytes must be 12 (x, y, z only) or >= 16, and a multiple of 4"; return ALOAM_E_ARG; }
  if (stride_bytes == 12 && c->cfg.ring_from_field) { c->err = "ring_from_field needs the 4th float of every record: stride_bytes >= 16"; return ALOAM_E_ARG; }
  int nin_max = 0;
  for (int b = 0; b < c->B; ++b) {
    if (n_in[b] < 0) { c->err = "negative point count"; return ALOAM_E_ARG; }
    if (n_in[b] > c->max_points) { c->err = "scan exceeds max_points"; return ALOAM_E_CAPACITY; }
    nin_max = std::max(nin_max, n_in[b]);
  }
  c->nin_max = nin_max;
  const int ns = c->h_nin_slot;
  c->h_nin_slot = (ns + 1) % kNinSlots;
  int* nin_slot = c->h_nin + (size_t)ns * c->B;
  if (c->nin_used[ns]) HIP_TRY(c, hipEventSynchronize(c->nin_done[ns]));   // the copy queued kNinSlots launches ago has read it
  std::memcpy(nin_slot, n_in, sizeof(int) * c->B);
  HIP_TRY(c, hipMemcpyAsync(c->d_nin, nin_slot, sizeof(int) * c->B, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->nin_done[ns], c->stream));
  c->nin_used[ns] = true;
  c->debug_arrays = debug_arrays || c->sum_order != 0;      // the reference-order pass reads cloudLabel
  if (((++c->reg_epoch) & 0x7fffffffu) == 0) ++c->reg_epoch;                 // 31 bits of it tag the look-back granules; 0 = "never written"
  const RegArgs a = reg_args(c, d_scans, seq_stride, stride_bytes);
  { ProfScope p(c, K_FIND_ENDS); launch_find_ends(a, c->d_nin, c->stream); }
  { ProfScope p(c, K_CLASSIFY); launch_front(a, c->stream); }
  { ProfScope p(c, K_RING_OFFSETS); launch_ring_starts(a, c->stream); }
  c->dense_valid = false;
  if (slot >= 0) { HIP_TRY(c, hipEventRecord(c->in_consumed[slot], c->stream)); c->in_used[slot] = true; }   // the raw sweep is not read after this
  { ProfScope p(c, K_RING_FEATURES); launch_ring_features(a, c->npad, 0.2f, c->stream);     // leaf 0.2 (src/scanRegistration.cpp:404)
    if (c->sum_order) launch_less_flat_reference_order(reg_args(c, d_scans, seq_stride, stride_bytes), c->npad, 0.2f, c->stream); }
  HIP_TRY(c, hipGetLastError());
  c->have_features = true;
  return ALOAM_OK;
}

}  // namespace

extern "C" {

void aloam_default_config(aloam_config* cfg) {
  cfg->n_scans = 64;            // launch/aloam_velodyne_HDL_64.launch: scan_line
  cfg->min_range = 5.0f;        // launch/aloam_velodyne_HDL_64.launch: minimum_range
  cfg->ring_from_field = 0;
  cfg->batch = 1;
  cfg->max_points = 140000;
  cfg->max_ring_points = 4107;
  cfg->device = 0;
  cfg->lm_max_iterations = 4;
  cfg->outer_iterations = 2;
  cfg->distortion = 0;
}

int aloam_create(const aloam_config* cfg, aloam_ctx** out) { return aloam_create_stages(cfg, ALOAM_STAGE_ALL, out); }

int aloam_create_stages(const aloam_config* cfg, int stages, aloam_ctx** out) {
  if (!cfg || !out) return ALOAM_E_ARG;
  *out = nullptr;
  aloam_ctx* c = new aloam_ctx();
  c->cfg = *cfg;
  c->stages = stages;
  { const char* e = std::getenv("ALOAM_DEBUG_SYNC"); c->debug_sync = e && e[0] == '1'; }
  *out = c;   // returned even on failure so that aloam_last_error() works; caller destroys it
  if ((stages & ~ALOAM_STAGE_ALL) || !(stages & ALOAM_STAGE_ALL)) { c->err = "stages must be a non-empty combination of ALOAM_STAGE_*"; return ALOAM_E_ARG; }
  if (cfg->batch < 1 || cfg->max_points < 32 || cfg->max_points > 400000 || cfg->lm_max_iterations < 0 || cfg->outer_iterations < 1 ||
      cfg->outer_iterations > 64) { c->err = "bad configuration value"; return ALOAM_E_ARG; }
  if (!cfg->ring_from_field && cfg->n_scans != 16 && cfg->n_scans != 32 && cfg->n_scans != 64) {
    c->err = "only support velodyne with 16, 32 or 64 scan line (or ring_from_field)";   // src/scanRegistration.cpp:472-476
    return ALOAM_E_SCAN_LINES;
  }
  if (cfg->n_scans < 1 || cfg->n_scans > kMaxRings) { c->err = "n_scans out of range"; return ALOAM_E_ARG; }
  if (cfg->max_ring_points < 17 || cfg->max_ring_points > 4107) { c->err = "max_ring_points must be in [17, 4107]"; return ALOAM_E_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { c->err = "no HIP device available (this library has no CPU fallback)"; return ALOAM_E_HIP; }
  if (cfg->device < 0 || cfg->device >= ndev) { c->err = "device ordinal out of range"; return ALOAM_E_ARG; }
  DeviceScope device_scope(c);                      // the caller's current device is restored on every return path
  HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(c, hipEventCreateWithFlags(&c->in_copied[k], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->in_consumed[k], hipEventDisableTiming));
  }
  c->B = cfg->batch; c->max_points = cfg->max_points; c->R = cfg->n_scans;
  // The per-sequence stride of every [B][cap] buffer is kept OFF the powers of two (131 072 points x 16 B = 2 MiB apart, the workgroups of a launch - one
  // per sequence, all at about the same offset of their sequence - meet on the same memory channels): + 1/32 + 16 points.  Measured on k_build_grids_fused at
  // batch 1024, one box: 1.62 - 1.65 ms at the power-of-two stride, 1.48 - 1.52 ms with 1040 / 4112 / 16 400 points of padding.
  c->cap = cfg->max_points + (((cfg->max_points / 32 + 15) & ~15) + 16);
  { const char* e = std::getenv("ALOAM_GRAPH_MAX_BATCH"); c->use_graph = c->B <= (e ? std::atoi(e) : 0); }   // off unless asked for: measured no gain (below)
  c->NB = (c->cap + kBlockPts - 1) / kBlockPts;
  c->npad = cfg->max_ring_points <= 2059 ? 2048 : 4096;
  const size_t B = c->B, cap = c->cap, R = c->R, NB = c->NB;
  int rc = 0;
  HIP_TRY(c, hipHostMalloc((void**)&c->h_nin, sizeof(int) * B * kNinSlots, hipHostMallocDefault));
  for (hipEvent_t& e : c->nin_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const bool reg = stages & ALOAM_STAGE_REGISTRATION, odo = stages & ALOAM_STAGE_ODOMETRY, map = stages & ALOAM_STAGE_MAPPING;
  // hash tables of the correspondence search sized by the clouds they index (power of two; the surf table must fit k_build_grids' LDS)
  c->grid_H[0] = R > 64 ? 8192 : 4096;
  c->grid_H[1] = c->max_points > 160000 ? 32768 : 16384;
  if ((rc = dmalloc(c, &c->d_nin, B))) return rc;
  if ((rc = dmalloc(c, &c->d_meta, B))) return rc;
  if ((rc = dmalloc(c, &c->d_state, B))) return rc;
  if ((rc = dmalloc(c, &c->d_cloud, B * cap))) return rc;                    // /velodyne_cloud_2 -> _3 -> mapping's full-resolution input
  if (reg) {                                                                 // working set of scan registration
    c->slab = c->npad + 16;                                                  // >= the longest ring k_ring_features accepts (npad + 11)
    if ((rc = dmalloc(c, &c->d_slabs, B * R * (size_t)c->slab))) return rc;
    if ((rc = dmalloc(c, &c->d_front_lb, B * NB * (size_t)kFrontSlots))) return rc;
    if ((rc = dmalloc(c, &c->d_front_ticket, B))) return rc;
    if ((rc = dmalloc(c, &c->d_ringstart, B * (R + 1)))) return rc;
    if ((rc = dmalloc(c, &c->d_curv, B * cap))) return rc;
    if ((rc = dmalloc(c, &c->d_label, B * cap))) return rc;
    if ((rc = dmalloc(c, &c->d_lookback, B * 4 * R))) return rc;
    if ((rc = dmalloc(c, &c->d_ring_ticket, B))) return rc;
  }
  if (reg || odo) {
    if ((rc = dmalloc(c, &c->d_sharp, B * R * 12))) return rc;
    if ((rc = dmalloc(c, &c->d_flat, B * R * 24))) return rc;
  }
  for (int k = 0; k < 2; ++k) {
    // [cur = 0] receives the sweep being registered, [1] is what a mapping-only context is handed as the "last" clouds; odometry flips between both
    if (!(odo || (k == 0 && reg) || (k == 1 && map))) continue;
    if ((rc = dmalloc(c, &c->d_less_sharp[k], B * R * 120))) return rc;
    if ((rc = dmalloc(c, &c->d_less_flat[k], B * cap))) return rc;
  }
  if (odo) {
    for (int k = 0; k < 2; ++k) {
      const size_t per = k == 0 ? R * 120 : cap;
      if ((rc = dmalloc(c, &c->d_grid_sorted3[k], B * per))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_sorted2[k], B * per))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_start3[k], B * (c->grid_H[k] + 1)))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_start2[k], B * (c->grid_H[k] + 1)))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_sorted3c[k], B * per))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_start3c[k], B * (c->grid_H[k] + 1)))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_flags[k], B * 4))) return rc;
      if ((rc = dmalloc(c, &c->d_grid_walk[k], B * 2 * (R + 8)))) return rc;
    }
    if ((rc = dmalloc(c, &c->d_edges, B * R * 12))) return rc;
    if ((rc = dmalloc(c, &c->d_planes, B * R * 24))) return rc;
    if ((rc = dmalloc(c, &c->d_sel_sharp, B * R * 12))) return rc;
    if ((rc = dmalloc(c, &c->d_sel_flat, B * R * 24))) return rc;
    if ((rc = prepare_build_grids(c->grid_H[1]))) { c->err = "k_build_grids: dynamic LDS size rejected"; return ALOAM_E_HIP; }
  }
  // identity poses (src/laserOdometry.cpp:93-98)
  std::vector<OdomState> init(B);
  std::memset(init.data(), 0, sizeof(OdomState) * B);
  for (size_t b = 0; b < B; ++b) { init[b].para_q[3] = 1.0; init[b].q_w[3] = 1.0; }
  HIP_TRY(c, hipMemcpyAsync(c->d_state, init.data(), sizeof(OdomState) * B, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ALOAM_OK;
}

void aloam_destroy(aloam_ctx* c) {
  DeviceScope device_scope(c);
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  prof_resolve(c);
  for (hipGraphExec_t& ge : c->odom_graph) if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
  for (hipEvent_t e : c->prof_free) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  void* bufs[] = {c->d_in[0], c->d_in[1], c->d_nin, c->d_meta, c->d_slabs, c->d_front_lb, c->d_front_ticket, c->d_ringstart, c->d_cloud, c->d_curv,
                  c->d_label, c->d_lookback, c->d_ring_ticket, c->d_sharp,
                  c->d_flat, c->d_less_sharp[0], c->d_less_sharp[1], c->d_less_flat[0], c->d_less_flat[1], c->d_state, c->d_edges, c->d_planes, c->d_sel_sharp, c->d_sel_flat,
                  c->d_grid_sorted3[0], c->d_grid_sorted3[1], c->d_grid_sorted2[0], c->d_grid_sorted2[1], c->d_grid_start3[0], c->d_grid_start3[1],
                  c->d_grid_start2[0], c->d_grid_start2[1],
                  c->d_grid_flags[0], c->d_grid_flags[1], c->d_grid_walk[0], c->d_grid_walk[1], c->d_grid_sorted3c[0], c->d_grid_sorted3c[1],
                  c->d_grid_start3c[0], c->d_grid_start3c[1],
                  c->d_mapseq, c->d_cubes, c->d_pool[0], c->d_pool[1], c->d_maptab, c->d_stack[0], c->d_stack[1], c->d_stack_world[0], c->d_stack_world[1],
                  c->d_stack_cube[0], c->d_stack_cube[1], c->d_addcnt, c->d_cursor, c->d_mgrid_sorted[0], c->d_mgrid_sorted[1], c->d_mgrid_start[0],
                  c->d_mgrid_start[1], c->d_map_report, c->d_map_live, c->d_medges, c->d_mnorms, c->d_registered, c->d_segs, c->d_tile_seg,
                  c->d_tile_heads, c->d_tile_pref, c->d_vox_counters, c->d_bbox, c->d_keys[0], c->d_keys[1], c->d_voxtmp, c->d_knn, c->d_compact_flag, c->d_vox_lists, c->d_rec_tiles};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_nin) (void)hipHostFree(c->h_nin);
  if (c->h_map_report) (void)hipHostFree((void*)c->h_map_report);
  for (hipEvent_t e : c->map_step_done) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->nin_done) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < 2; ++k) { if (c->in_copied[k]) (void)hipEventDestroy(c->in_copied[k]); if (c->in_consumed[k]) (void)hipEventDestroy(c->in_consumed[k]); }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* aloam_last_error(const aloam_ctx* c) { return c ? c->err.c_str() : "null context"; }
void* aloam_stream(aloam_ctx* c) { return c ? (void*)c->stream : nullptr; }

int aloam_synchronize(aloam_ctx* c) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::vector<SeqMeta> m(c->B);
  HIP_TRY(c, hipMemcpy(m.data(), c->d_meta, sizeof(SeqMeta) * c->B, hipMemcpyDeviceToHost));
  for (int b = 0; b < c->B; ++b) {
    if (m[b].err & kErrEmpty) { c->err = "sequence " + std::to_string(b) + ": no point survives the NaN / minimum-range filter"; return ALOAM_E_EMPTY; }
    if (m[b].err & (kErrRingCap | kErrPointCap)) { c->err = "sequence " + std::to_string(b) + ": a ring exceeds max_ring_points or the scan exceeds max_points"; return ALOAM_E_CAPACITY; }
    if (m[b].err & kErrInternal) { c->err = "sequence " + std::to_string(b) + ": internal error, a ring workgroup of k_ring_features never published its counts (look-back wait timed out)"; return ALOAM_E_HIP; }
  }
  if (c->map_on) {
    std::vector<MapSeq> ms(c->B);
    int vc[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpy(ms.data(), c->d_mapseq, sizeof(MapSeq) * c->B, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vc, c->d_vox_counters, sizeof(vc), hipMemcpyDeviceToHost));
    // The per-step flags (MapSeq.err, counters[1]) are folded into running counts when the next step starts (k_map_begin), so a
    // caller that queues many steps and synchronises once still hears about every step that dropped points: reported once, at the
    // first aloam_synchronize after it happened.  The steps themselves have run: poses and map are valid, the points that did not
    // fit were left out of the map.
    // Per-sequence deltas against what has already been reported, pool events and voxel-scratch events apart, so the message names a sequence
    // that dropped points SINCE the last call and says which resource ran out.
    if ((int)c->map_err_seen.size() != c->B) c->map_err_seen.assign(c->B, 0);
    long long fresh_pool = 0;
    int first_seq = -1;
    for (int b = 0; b < c->B; ++b) {
      const long long n = ms[b].err_steps + ((ms[b].err & kMapErrPool) ? 1 : 0);
      if (n > c->map_err_seen[b]) { fresh_pool += n - c->map_err_seen[b]; if (first_seq < 0) first_seq = b; }
      c->map_err_seen[b] = n;
    }
    const long long vox = vc[3] + (vc[1] ? 1 : 0), fresh_vox = vox > c->map_err_reported ? vox - c->map_err_reported : 0;
    c->map_err_reported = vox;
    if (fresh_pool + fresh_vox > 0) {
      c->err = "mapping, since the last aloam_synchronize:";
      if (fresh_pool) c->err += " " + std::to_string(fresh_pool) + " (sequence, step) pair(s) ran out of map pool (first: sequence " + std::to_string(first_seq) + ")";
      if (fresh_vox) c->err += std::string(fresh_pool ? " and" : "") + " " + std::to_string(fresh_vox) + " step(s) ran out of voxel-filter scratch";
      c->err += "; the points that did not fit were not inserted (raise pool_points)";
      return ALOAM_E_CAPACITY;
    }
  }
  return ALOAM_OK;
}

int aloam_scan_register_device(aloam_ctx* c, const void* d_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes) {
  DeviceScope device_scope(c);
  if (!c || !d_scans || !n_in) return ALOAM_E_ARG;
  return register_launch(c, d_scans, seq_stride_bytes, n_in, stride_bytes);
}

// Next device staging slab for a host-resident batch: waits (host side) until the kernels that read the slab two calls ago
// are done with it, grows it if needed.
static int acquire_slab(aloam_ctx* c, size_t need, int* slot_out) {
  const int s = c->in_slot;
  c->in_slot ^= 1;
  if (c->in_used[s]) HIP_TRY(c, hipEventSynchronize(c->in_consumed[s]));
  if (c->d_in_bytes[s] < need) {
    if (c->d_in[s]) { char* old = c->d_in[s]; c->d_in[s] = nullptr; c->d_in_bytes[s] = 0; HIP_TRY(c, hipFree(old)); }
    HIP_TRY(c, hipMalloc((void**)&c->d_in[s], need));
    c->d_in_bytes[s] = need;
  }
  *slot_out = s;
  return ALOAM_OK;
}

int aloam_scan_register(aloam_ctx* c, const void* const* scans, const int* n_in, int stride_bytes) {
  DeviceScope device_scope(c);
  if (!c || !scans || !n_in) return ALOAM_E_ARG;
  if (!(c->stages & ALOAM_STAGE_REGISTRATION)) { c->err = "this context was created without ALOAM_STAGE_REGISTRATION"; return ALOAM_E_STATE; }
  if (stride_bytes < 12 || (stride_bytes & 3)) { c->err = "stride_bytes must be 12 (x, y, z only) or >= 16, and a multiple of 4"; return ALOAM_E_ARG; }
  const size_t seq_stride = (size_t)c->cap * stride_bytes;
  for (int b = 0; b < c->B; ++b) if (n_in[b] > c->max_points) { c->err = "scan exceeds max_points"; return ALOAM_E_CAPACITY; }
  int slot = 0;
  int rc = acquire_slab(c, seq_stride * c->B, &slot);
  if (rc) return rc;
  for (int b = 0; b < c->B; ++b)
    if (n_in[b] > 0) HIP_TRY(c, hipMemcpyAsync(c->d_in[slot] + b * seq_stride, scans[b], (size_t)n_in[b] * stride_bytes, hipMemcpyHostToDevice, c->stream));
  rc = register_launch(c, c->d_in[slot], (long long)seq_stride, n_in, stride_bytes, slot);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the host buffers may be reused on return
  return ALOAM_OK;
}

// Host-resident batch in ONE buffer (sequence b at h_scans + b * seq_stride_bytes): one batched H2D copy on the context's copy
// stream into the next of two device slabs, the kernels wait for it on the compute stream — so the copy of call k + 1 runs
// under the kernels of call k.  Truly asynchronous only from pinned memory (hipHostMalloc / hipHostRegister); the runtime stages
// pageable memory synchronously.  The buffer must stay unmodified until aloam_input_consumed() / aloam_synchronize().
static int stage_and_register(aloam_ctx* c, const void* h_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes, bool debug_arrays) {
  if (!c || !h_scans || !n_in) return ALOAM_E_ARG;
  if (!(c->stages & ALOAM_STAGE_REGISTRATION)) { c->err = "this context was created without ALOAM_STAGE_REGISTRATION"; return ALOAM_E_STATE; }
  if (stride_bytes < 12 || (stride_bytes & 3) || seq_stride_bytes < 0) { c->err = "bad stride"; return ALOAM_E_ARG; }
  int nmax = 0;
  for (int b = 0; b < c->B; ++b) {
    if (n_in[b] < 0) { c->err = "negative point count"; return ALOAM_E_ARG; }
    if (n_in[b] > c->max_points) { c->err = "scan exceeds max_points"; return ALOAM_E_CAPACITY; }
    nmax = std::max(nmax, n_in[b]);
  }
  const size_t d_seq_stride = (size_t)c->cap * stride_bytes;
  int slot = 0;
  int rc = acquire_slab(c, d_seq_stride * c->B, &slot);
  if (rc) return rc;
  const size_t row = (size_t)nmax * stride_bytes;
  if (c->B > 1 && (size_t)seq_stride_bytes < row) { c->err = "seq_stride_bytes smaller than a scan"; return ALOAM_E_ARG; }
  if (row > 0) {
    // rows 0 .. B-2 as one strided copy of the batch-wide maximum (every row but the last is followed by the next one, so the
    // extra bytes are readable); the last row with its own length, so that a buffer that ends with the last sweep is never over-read
    if (c->B > 1) HIP_TRY(c, hipMemcpy2DAsync(c->d_in[slot], d_seq_stride, h_scans, (size_t)seq_stride_bytes, row, (size_t)c->B - 1, hipMemcpyHostToDevice, c->copy_stream));
    const size_t last = (size_t)n_in[c->B - 1] * stride_bytes;
    if (last > 0) HIP_TRY(c, hipMemcpyAsync(c->d_in[slot] + (size_t)(c->B - 1) * d_seq_stride, (const char*)h_scans + (size_t)(c->B - 1) * (size_t)seq_stride_bytes, last, hipMemcpyHostToDevice, c->copy_stream));
  }
  HIP_TRY(c, hipEventRecord(c->in_copied[slot], c->copy_stream));
  HIP_TRY(c, hipStreamWaitEvent(c->stream, c->in_copied[slot], 0));
  return register_launch(c, c->d_in[slot], (long long)d_seq_stride, n_in, stride_bytes, slot, debug_arrays);
}

int aloam_scan_register_host(aloam_ctx* c, const void* h_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes) {
  DeviceScope device_scope(c);
  return stage_and_register(c, h_scans, seq_stride_bytes, n_in, stride_bytes, true);
}

int aloam_process_host(aloam_ctx* c, const void* h_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes) {
  DeviceScope device_scope(c);
  const int rc = stage_and_register(c, h_scans, seq_stride_bytes, n_in, stride_bytes, false);
  if (rc) return rc;
  return aloam_odometry_step(c);
}

int aloam_input_consumed(aloam_ctx* c) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  for (int s = 0; s < 2; ++s) if (c->in_used[s]) HIP_TRY(c, hipEventSynchronize(c->in_consumed[s]));
  return ALOAM_OK;
}

int aloam_odometry_step(aloam_ctx* c) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  if (!(c->stages & ALOAM_STAGE_ODOMETRY)) { c->err = "this context was created without ALOAM_STAGE_ODOMETRY"; return ALOAM_E_STATE; }
  if (!c->have_features) { c->err = "aloam_odometry_step before any features were registered / set"; return ALOAM_E_STATE; }
  auto launch_all = [&]() {
    OdomArgs a = odom_args(c);
    { ProfScope p(c, K_BUILD_GRIDS); launch_build_grids(a, c->stream); }          // kd-tree stand-in over the last clouds
    for (int outer = 0; outer < c->cfg.outer_iterations; ++outer) {
      a.outer = outer;
      a.last_outer = outer == c->cfg.outer_iterations - 1;
      { ProfScope p(c, K_TRANSFORM); launch_transform_queries(a, c->stream); }    // TransformToStart of the features (:300, :388)
      { ProfScope p(c, K_ASSOC_CORNER); launch_associate(a, false, c->stream); }
      { ProfScope p(c, K_ASSOC_PLANE); launch_associate(a, true, c->stream); }
      { ProfScope p(c, K_SOLVE); launch_solve(a, c->stream); }
    }
    { ProfScope p(c, K_ADVANCE); launch_advance(c->d_meta, c->B, c->stream); }   // swap (src/laserOdometry.cpp:554-563)
  };
  if (!c->system_inited) {
    c->system_inited = true;                       // first frame: no solve (src/laserOdometry.cpp:267-271)
    { ProfScope p(c, K_ADVANCE); launch_advance(c->d_meta, c->B, c->stream); }
  } else if (c->use_graph && !c->prof_on && !c->debug_sync) {
    // The kernel arguments of a step depend on the buffer parity only (pointer flip of the last clouds), so each parity is captured once
    // and replayed: one launch instead of ~15.  Measured at batch 1 (bench.py latency leg): 0.418 ms per step against 0.416 ms with separate
    // launches — the step is bound by the execution of its dependent kernels (one sequence fills a fraction of the chip), not by launching them,
    // so the path is kept (tested bit for bit) but off by default.
    hipGraphExec_t& ge = c->odom_graph[c->cur];
    if (!ge) {
      // A failed capture must not leave the stream in capture mode or leak the graph: the capture is always ended, the graph always
      // destroyed, and on any error this context goes back to separate launches for good (the step itself is then launched normally).
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        launch_all();
        e = hipStreamEndCapture(c->stream, &g);                  // launch errors inside the capture surface here
        if (e == hipSuccess) e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();                                 // clear the sticky capture error; the cause is not lost: the plain launches below report theirs
        if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
        c->use_graph = false;
      }
    }
    if (ge) HIP_TRY(c, hipGraphLaunch(ge, c->stream));
    else launch_all();
  } else {
    launch_all();
  }
  HIP_TRY(c, hipGetLastError());
  c->cur ^= 1;
  return ALOAM_OK;
}

int aloam_process_device(aloam_ctx* c, const void* d_scans, long long seq_stride_bytes, const int* n_in, int stride_bytes) {
  DeviceScope device_scope(c);
  if (!c || !d_scans || !n_in) return ALOAM_E_ARG;
  const int rc = register_launch(c, d_scans, seq_stride_bytes, n_in, stride_bytes, -1, /*debug_arrays=*/false);
  if (rc) return rc;
  return aloam_odometry_step(c);
}

// ---- results ---------------------------------------------------------------------------------------------
static int cloud_ref(aloam_ctx* c, int seq, int which, const SeqMeta& m, const float4** ptr, int* n) {
  const size_t b = seq;
  auto at = [](const float4* base, size_t off) -> const float4* { return base ? base + off : nullptr; };
  // aloam_odometry_step ends with the reference's pointer swap (src/laserOdometry.cpp:554-560): afterwards the sweep
  // just processed is read through CORNER_LAST / SURF_LAST, exactly like laserCloudCornerLast / laserCloudSurfLast.
  switch (which) {
    case ALOAM_CLOUD_FULL: *ptr = at(c->d_cloud, b * c->cap); *n = m.n_cloud; return 0;
    case ALOAM_CLOUD_SHARP: *ptr = at(c->d_sharp, b * c->R * 12); *n = m.n_sharp; return 0;
    case ALOAM_CLOUD_FLAT: *ptr = at(c->d_flat, b * c->R * 24); *n = m.n_flat; return 0;
    case ALOAM_CLOUD_LESS_SHARP: *ptr = at(c->d_less_sharp[c->cur], b * c->R * 120); *n = m.n_less_sharp; return 0;
    case ALOAM_CLOUD_LESS_FLAT: *ptr = at(c->d_less_flat[c->cur], b * c->cap); *n = m.n_less_flat; return 0;
    case ALOAM_CLOUD_CORNER_LAST: *ptr = at(c->d_less_sharp[1 - c->cur], b * c->R * 120); *n = m.n_corner_last; return 0;
    case ALOAM_CLOUD_SURF_LAST: *ptr = at(c->d_less_flat[1 - c->cur], b * c->cap); *n = m.n_surf_last; return 0;
  }
  return -1;
}

int aloam_cloud_size(aloam_ctx* c, int seq, int which) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (which == ALOAM_CLOUD_FULL && (rc = ensure_dense(c))) return rc;
  SeqMeta m;
  if ((rc = fetch_meta(c, seq, &m))) return rc;
  const float4* p; int n;
  if (cloud_ref(c, seq, which, m, &p, &n)) { c->err = "unknown cloud id"; return ALOAM_E_ARG; }
  if (!p) { c->err = "this context holds no such cloud (see aloam_create_stages)"; return ALOAM_E_STATE; }
  return n;
}

int aloam_get_cloud(aloam_ctx* c, int seq, int which, float* out, int cap_points) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (which == ALOAM_CLOUD_FULL && (rc = ensure_dense(c))) return rc;
  SeqMeta m;
  if ((rc = fetch_meta(c, seq, &m))) return rc;
  const float4* p; int n;
  if (cloud_ref(c, seq, which, m, &p, &n)) { c->err = "unknown cloud id"; return ALOAM_E_ARG; }
  if (!p) { c->err = "this context holds no such cloud (see aloam_create_stages)"; return ALOAM_E_STATE; }
  const int k = n < cap_points ? n : cap_points;
  if (k > 0) HIP_TRY(c, hipMemcpy(out, p, sizeof(float4) * k, hipMemcpyDeviceToHost));
  return n;
}

int aloam_get_pose(aloam_ctx* c, int seq, double q_w[4], double t_w[3], double q_lc[4], double t_lc[3]) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if ((rc = sync_and_check(c))) return rc;
  OdomState s;
  HIP_TRY(c, hipMemcpy(&s, c->d_state + seq, sizeof(OdomState), hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; ++k) { q_w[k] = s.q_w[k]; q_lc[k] = s.para_q[k]; }
  for (int k = 0; k < 3; ++k) { t_w[k] = s.t_w[k]; t_lc[k] = s.para_t[k]; }
  return ALOAM_OK;
}

int aloam_get_odom_stats(aloam_ctx* c, int seq, aloam_odom_stats* out) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if ((rc = sync_and_check(c))) return rc;
  OdomState s;
  HIP_TRY(c, hipMemcpy(&s, c->d_state + seq, sizeof(OdomState), hipMemcpyDeviceToHost));
  for (int k = 0; k < 2; ++k) {
    out->corner_corr[k] = s.corner_corr[k]; out->plane_corr[k] = s.plane_corr[k];
    out->lm_iterations[k] = s.lm_iterations[k]; out->lm_successful[k] = s.lm_successful[k];
    out->initial_cost[k] = s.initial_cost[k]; out->final_cost[k] = s.final_cost[k]; out->termination[k] = s.termination[k];
  }
  return ALOAM_OK;
}

// ---- state injection -----------------------------------------------------------------------------------------
int aloam_set_features(aloam_ctx* c, int seq, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp,
                       const float* flat, int n_flat, const float* less_flat, int n_less_flat) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (n_sharp < 0 || n_sharp > c->R * 12 || n_less_sharp < 0 || n_less_sharp > c->R * 120 || n_flat < 0 || n_flat > c->R * 24 ||
      n_less_flat < 0 || n_less_flat > c->max_points) { c->err = "feature cloud larger than the selection rules allow"; return ALOAM_E_CAPACITY; }
  if (!c->d_sharp || !c->d_less_sharp[c->cur]) { c->err = "this context has no feature buffers (created for the mapping stage only)"; return ALOAM_E_STATE; }
  const size_t b = seq;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n_sharp) HIP_TRY(c, hipMemcpy(c->d_sharp + b * c->R * 12, sharp, sizeof(float4) * n_sharp, hipMemcpyHostToDevice));
  if (n_less_sharp) HIP_TRY(c, hipMemcpy(c->d_less_sharp[c->cur] + b * c->R * 120, less_sharp, sizeof(float4) * n_less_sharp, hipMemcpyHostToDevice));
  if (n_flat) HIP_TRY(c, hipMemcpy(c->d_flat + b * c->R * 24, flat, sizeof(float4) * n_flat, hipMemcpyHostToDevice));
  if (n_less_flat) HIP_TRY(c, hipMemcpy(c->d_less_flat[c->cur] + b * c->cap, less_flat, sizeof(float4) * n_less_flat, hipMemcpyHostToDevice));
  SeqMeta m;
  HIP_TRY(c, hipMemcpy(&m, c->d_meta + seq, sizeof(SeqMeta), hipMemcpyDeviceToHost));
  m.n_sharp = n_sharp; m.n_less_sharp = n_less_sharp; m.n_flat = n_flat; m.n_less_flat = n_less_flat; m.err = 0;
  HIP_TRY(c, hipMemcpy(c->d_meta + seq, &m, sizeof(SeqMeta), hipMemcpyHostToDevice));
  c->have_features = true;
  return ALOAM_OK;
}

int aloam_set_last(aloam_ctx* c, int seq, const float* corner_last, int n_corner, const float* surf_last, int n_surf) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (n_corner < 0 || n_corner > c->R * 120 || n_surf < 0 || n_surf > c->max_points) { c->err = "last cloud too large"; return ALOAM_E_CAPACITY; }
  c->inject_max = std::max(c->inject_max, std::max(n_corner, n_surf));
  if (!c->d_less_sharp[1 - c->cur]) { c->err = "this context has no buffers for the last clouds (created for the registration stage only)"; return ALOAM_E_STATE; }
  const size_t b = seq;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n_corner) HIP_TRY(c, hipMemcpy(c->d_less_sharp[1 - c->cur] + b * c->R * 120, corner_last, sizeof(float4) * n_corner, hipMemcpyHostToDevice));
  if (n_surf) HIP_TRY(c, hipMemcpy(c->d_less_flat[1 - c->cur] + b * c->cap, surf_last, sizeof(float4) * n_surf, hipMemcpyHostToDevice));
  SeqMeta m;
  HIP_TRY(c, hipMemcpy(&m, c->d_meta + seq, sizeof(SeqMeta), hipMemcpyDeviceToHost));
  m.n_corner_last = n_corner; m.n_surf_last = n_surf;
  HIP_TRY(c, hipMemcpy(c->d_meta + seq, &m, sizeof(SeqMeta), hipMemcpyHostToDevice));
  return ALOAM_OK;
}

int aloam_set_state(aloam_ctx* c, int seq, const double para_q[4], const double para_t[3], const double q_w[4], const double t_w[3]) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  OdomState s;
  HIP_TRY(c, hipMemcpy(&s, c->d_state + seq, sizeof(OdomState), hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; ++k) { s.para_q[k] = para_q[k]; s.q_w[k] = q_w[k]; }
  for (int k = 0; k < 3; ++k) { s.para_t[k] = para_t[k]; s.t_w[k] = t_w[k]; }
  HIP_TRY(c, hipMemcpy(c->d_state + seq, &s, sizeof(OdomState), hipMemcpyHostToDevice));
  return ALOAM_OK;
}

int aloam_set_system_inited(aloam_ctx* c, int inited) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  c->system_inited = inited != 0;
  return ALOAM_OK;
}

// ---- intermediate arrays ---------------------------------------------------------------------------------------
int aloam_get_ring_ranges(aloam_ctx* c, int seq, int* start, int* count) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!(c->stages & ALOAM_STAGE_REGISTRATION)) { c->err = "this context was created without ALOAM_STAGE_REGISTRATION"; return ALOAM_E_STATE; }
  if ((rc = sync_and_check(c))) return rc;
  std::vector<int> rs(c->R + 1);
  HIP_TRY(c, hipMemcpy(rs.data(), c->d_ringstart + (size_t)seq * (c->R + 1), sizeof(int) * (c->R + 1), hipMemcpyDeviceToHost));
  for (int r = 0; r < c->R; ++r) { start[r] = rs[r]; count[r] = rs[r + 1] - rs[r]; }
  return c->R;
}

int aloam_get_curvature(aloam_ctx* c, int seq, float* out, int cap) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!(c->stages & ALOAM_STAGE_REGISTRATION)) { c->err = "this context was created without ALOAM_STAGE_REGISTRATION"; return ALOAM_E_STATE; }
  if (!c->debug_arrays) { c->err = "curvature is only kept by aloam_scan_register*; the throughput entries (aloam_process_*) skip it"; return ALOAM_E_STATE; }
  SeqMeta m;
  if ((rc = fetch_meta(c, seq, &m))) return rc;
  const int k = m.n_cloud < cap ? m.n_cloud : cap;
  if (k > 0) HIP_TRY(c, hipMemcpy(out, c->d_curv + (size_t)seq * c->cap, sizeof(float) * k, hipMemcpyDeviceToHost));
  return m.n_cloud;
}

int aloam_get_labels(aloam_ctx* c, int seq, int* out, int cap) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!(c->stages & ALOAM_STAGE_REGISTRATION)) { c->err = "this context was created without ALOAM_STAGE_REGISTRATION"; return ALOAM_E_STATE; }
  if (!c->debug_arrays) { c->err = "labels are only kept by aloam_scan_register*; the throughput entries (aloam_process_*) skip them"; return ALOAM_E_STATE; }
  SeqMeta m;
  if ((rc = fetch_meta(c, seq, &m))) return rc;
  const int k = m.n_cloud < cap ? m.n_cloud : cap;
  std::vector<int8_t> tmp(k > 0 ? k : 1);
  if (k > 0) HIP_TRY(c, hipMemcpy(tmp.data(), c->d_label + (size_t)seq * c->cap, k, hipMemcpyDeviceToHost));
  for (int i = 0; i < k; ++i) out[i] = tmp[i];
  return m.n_cloud;
}

// Which association kernels own the sequence's last clouds (k_build_grids_fused): per cloud 0 = ring-sorted keys (pair kernel), 1 = nearly
// ring-sorted (pair kernel with the index-range walk window), 2 = not sorted (literal walks), -1 = keys / coordinates out of range (literal search).
int aloam_get_last_cloud_order(aloam_ctx* c, int seq, int out[2]) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!c->d_grid_flags[0]) { c->err = "this context has no odometry stage"; return ALOAM_E_STATE; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 2; ++k) {
    int f[4];
    HIP_TRY(c, hipMemcpy(f, c->d_grid_flags[k] + (size_t)seq * 4, sizeof(f), hipMemcpyDeviceToHost));
    out[k] = f[0] ? -1 : f[1];
  }
  return ALOAM_OK;
}

int aloam_get_correspondences(aloam_ctx* c, int seq, float* edges, int cap_edges, int* n_edges, int* edge_query,
                              float* planes, int cap_planes, int* n_planes, int* plane_query) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  SeqMeta m;
  if ((rc = fetch_meta(c, seq, &m))) return rc;
  std::vector<EdgeRec> E(m.n_sharp > 0 ? m.n_sharp : 1);
  std::vector<PlaneRec> P(m.n_flat > 0 ? m.n_flat : 1);
  if (m.n_sharp > 0) HIP_TRY(c, hipMemcpy(E.data(), c->d_edges + (size_t)seq * c->R * 12, sizeof(EdgeRec) * m.n_sharp, hipMemcpyDeviceToHost));
  if (m.n_flat > 0) HIP_TRY(c, hipMemcpy(P.data(), c->d_planes + (size_t)seq * c->R * 24, sizeof(PlaneRec) * m.n_flat, hipMemcpyDeviceToHost));
  int ne = 0, np = 0;
  for (int i = 0; i < m.n_sharp; ++i) {
    if (!E[i].valid) continue;
    if (ne < cap_edges) {
      float* o = edges + (size_t)ne * 9;
      for (int k = 0; k < 3; ++k) { o[k] = E[i].cp[k]; o[3 + k] = E[i].a[k]; o[6 + k] = E[i].b[k]; }
      if (edge_query) edge_query[ne] = i;
    }
    ++ne;
  }
  for (int i = 0; i < m.n_flat; ++i) {
    if (!P[i].valid) continue;
    if (np < cap_planes) {
      float* o = planes + (size_t)np * 12;
      for (int k = 0; k < 3; ++k) { o[k] = P[i].cp[k]; o[3 + k] = P[i].j[k]; o[6 + k] = P[i].l[k]; o[9 + k] = P[i].m[k]; }
      if (plane_query) plane_query[np] = i;
    }
    ++np;
  }
  *n_edges = ne;
  *n_planes = np;
  return ALOAM_OK;
}

// ---- profiling -----------------------------------------------------------------------------------------------
int aloam_profile_enable(aloam_ctx* c, int on) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_resolve(c);
  if (on) { for (int k = 0; k < K_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_launches[k] = 0; } }
  c->prof_on = on != 0;
  return ALOAM_OK;
}
int aloam_profile_kernel_count(void) { return K_COUNT; }
const char* aloam_profile_kernel_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }

int aloam_profile_get(aloam_ctx* c, int kernel, double* total_ms, long long* launches, double* algorithmic_bytes) {
  DeviceScope device_scope(c);
  if (!c || kernel < 0 || kernel >= K_COUNT) return ALOAM_E_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_resolve(c);
  if (total_ms) *total_ms = c->prof_ms[kernel];
  if (launches) *launches = c->prof_launches[kernel];
  if (algorithmic_bytes) {
    // per-launch algorithmic traffic from the sizes of the LAST sweep (DESIGN.md "Algorithmic bytes")
    std::vector<SeqMeta> m(c->B);
    HIP_TRY(c, hipMemcpy(m.data(), c->d_meta, sizeof(SeqMeta) * c->B, hipMemcpyDeviceToHost));
    std::vector<MapSeq> ms(c->map_on ? c->B : 0);
    if (c->map_on) HIP_TRY(c, hipMemcpy(ms.data(), c->d_mapseq, sizeof(MapSeq) * c->B, hipMemcpyDeviceToHost));
    double bytes = 0;
    for (int b = 0; b < c->B; ++b) {
      const double Nin = m[b].n_in, N = m[b].n_cloud, Fc = m[b].n_sharp, Lc = m[b].n_less_sharp, Fs = m[b].n_flat, Ls = m[b].n_less_flat;
      const double Lcl = m[b].n_corner_last, Lsl = m[b].n_surf_last;
      switch (kernel) {
        case K_FIND_ENDS: bytes += 2 * 256 * 16; break;
        case K_CLASSIFY: bytes += 16 * Nin + 16 * N + 16.0 * (c->R + 1) * ((Nin + kBlockPts - 1) / kBlockPts); break;   // k_front: the sweep in, the slabs out, two granules per ring and block
        case K_RING_OFFSETS: bytes += 12.0 * c->R; break;                                                                  // k_ring_starts
        case K_SCATTER: bytes += 32 * N; break;                                                                            // k_dense_cloud (on demand)
        case K_RING_FEATURES: bytes += 16 * N + (c->debug_arrays ? 5 * N : 0) + 16 * (Fc + Lc + Fs + Ls); break;   // ring-ordered cloud in, the four feature clouds out (+ curvature / labels for the parity entry points)
        case K_BUILD_GRIDS: bytes += 16 * (Lcl + Lsl) + 48 * (Lcl + Lsl) + 12.0 * (c->grid_H[0] + c->grid_H[1]); break;   // read once, three sorted copies + three bucket tables out
        case K_TRANSFORM: bytes += 32 * (Fc + Fs); break;
        case K_ASSOC_CORNER: bytes += 16 * (Fc + Lcl) + 48 * Fc; break;
        case K_ASSOC_PLANE: bytes += 16 * (Fs + Lsl) + 64 * Fs; break;
        case K_SOLVE: bytes += 9.0 * (48 * Fc + 64 * Fs); break;
        // mapping: incoming clouds read + stacks written; submap read + grid written; queries + 5 neighbours + records;
        // 9 evaluations of the records; stacks -> cubes; valid cubes read + written; full cloud in + out
        default: break;
      }
      if (c->map_on && kernel >= K_MAP_BEGIN) {
        // scan-to-map stages, from the sizes of the last frame: S = down-sampled stacks, M = submap, F = factors
        const double Sc = ms[b].n_stack[0], Ss = ms[b].n_stack[1], Mc = ms[b].from_total[0], Ms = ms[b].from_total[1];
        const double Fc = ms[b].factor_num[1][0], Fs = ms[b].factor_num[1][1];
        switch (kernel) {
          case K_MAP_BEGIN: bytes += 2.0 * kMapValidMax * sizeof(CubeDesc) + sizeof(MapSeq); break;           // window descriptors + state
          case K_MAP_VOXEL_STACK: bytes += 16 * (Lcl + Lsl) + 16 * (Sc + Ss); break;                          // incoming clouds in, stacks out
          case K_MAP_GRID: bytes += 32 * (Mc + Ms) + 16.0 * c->map_H; break;               // submap in, bucketed copy + tables out
          case K_MAP_ASSOC: bytes += 16 * (Sc + Ss) + 80 * (Sc + Ss) + sizeof(MapEdgeRec) * Sc + sizeof(MapNormRec) * Ss; break;   // query + 5 neighbours in, record out
          case K_MAP_SOLVE: bytes += 9.0 * (sizeof(MapEdgeRec) * Fc + sizeof(MapNormRec) * Fs); break;        // <= 5 Jacobian + 4 cost evaluations
          case K_MAP_INSERT: bytes += 32 * (Sc + Ss); break;                                                  // stacks in, cube appends out
          case K_MAP_VOXEL_CUBES: bytes += 32 * (Mc + Ms + Sc + Ss); break;                                   // valid cubes re-filtered in place
          case K_MAP_REGISTER: bytes += 32 * N; break;                                                        // full cloud in + out
          default: break;
        }
      }
    }
    *algorithmic_bytes = bytes;
  }
  return ALOAM_OK;
}

// ---- stage 3: scan-to-map refinement --------------------------------------------------------------------------------
static MapArgs map_args(aloam_ctx* c) {
  MapArgs a{};
  a.B = c->B; a.cap = c->cap; a.R = c->R;
  a.meta = c->d_meta; a.odom = c->d_state; a.seq = c->d_mapseq;
  a.line_res = c->map_line_res; a.plane_res = c->map_plane_res;
  // after aloam_odometry_step's swap the sweep just processed is the "last" one: exactly what the odometry node publishes
  // as /laser_cloud_corner_last, /laser_cloud_surf_last and /velodyne_cloud_3 (reference src/laserOdometry.cpp:570-591)
  a.corner_last = c->d_less_sharp[1 - c->cur]; a.surf_last = c->d_less_flat[1 - c->cur]; a.full = c->d_cloud;
  if (!c->dense_valid) { a.slabs = c->d_slabs; a.slab = c->slab; a.ringstart = c->d_ringstart; }   // the sweep just registered lives in its ring slabs; the dense copy is made only for who asks
  a.registered = c->d_registered;
  a.cubes = c->d_cubes; a.pool_cap = c->map_pool; a.tab = c->d_maptab;
  for (int k = 0; k < 2; ++k) {
    a.pool[k] = c->d_pool[k]; a.stack[k] = c->d_stack[k]; a.stack_world[k] = c->d_stack_world[k]; a.stack_cube[k] = c->d_stack_cube[k];
    a.grid_sorted[k] = c->d_mgrid_sorted[k]; a.grid_start[k] = c->d_mgrid_start[k];
  }
  a.grid_H = c->map_H; a.live = c->d_map_live; a.report_dev = c->d_map_report; a.report_host = c->d_map_report_host;
  a.addcnt = c->d_addcnt; a.cursor = c->d_cursor; a.compact_flag = c->d_compact_flag;
  a.edges = c->d_medges; a.norms = c->d_mnorms; a.knn = c->d_knn;
  a.lm_max_iterations = c->cfg.lm_max_iterations;
  a.vox_counters = c->d_vox_counters;
  a.rec_tiles = c->d_rec_tiles; a.rec_tiles_per_seq = c->rec_tiles_per_seq; a.rec_tiles_corner = c->rec_tiles_corner;
  return a;
}
static VoxArgs vox_args(aloam_ctx* c, int n_segs, int levels) {
  VoxArgs v{};
  v.segs = c->d_segs; v.n_segs = n_segs; v.tile_seg = c->d_tile_seg; v.tile_heads = c->d_tile_heads; v.tile_pref = c->d_tile_pref;
  v.counters = c->d_vox_counters; v.keys[0] = c->d_keys[0]; v.keys[1] = c->d_keys[1]; v.tmp = c->d_voxtmp; v.bbox = c->d_bbox;
  v.tile_cap = c->map_tile_cap; v.key_cap = c->map_key_cap; v.levels = levels; v.lists = c->d_vox_lists;
  return v;
}

// Everything whose size follows the pool: the two class pools (contents kept when growing), the bucketed copy of the submap, the scratch of
// the general voxel path (keys, staging = 2 pools per sequence, tile lists) and the bucket tables.  New buffers are allocated first and
// the old ones released only when every allocation has succeeded, so a failed growth leaves the context as it was.
static int map_alloc_pool(aloam_ctx* c, int pool_points) {
  const size_t B = c->B, cap = c->cap, R = c->R, T = kVoxTile, pool = pool_points, old_pool = c->map_pool;
  int H = 4096;
  while (H < (int)(pool / 16) && H < kMapGridMaxH) H <<= 1;            // ~ submap size
  const long long key_cap = (long long)(B * std::max(cap + R * 120, 2 * pool));
  const int tile_bound1 = (int)(B * (2 * pool / T + 2 * kMapValidMax));
  const int tile_cap = std::max(c->map_tile_bound[0], tile_bound1);
  float4 *n_pool[2] = {nullptr, nullptr}, *n_sorted[2] = {nullptr, nullptr}, *n_tmp = nullptr;
  int *n_start[2] = {nullptr, nullptr}, *n_tseg = nullptr, *n_theads = nullptr, *n_tpref = nullptr;
  unsigned long long* n_keys[2] = {nullptr, nullptr};
  bool ok = true;
  auto grab = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) { *p = nullptr; ok = false; (void)hipGetLastError(); } };
  for (int k = 0; k < 2; ++k) {
    grab((void**)&n_pool[k], sizeof(float4) * B * pool);
    grab((void**)&n_sorted[k], sizeof(float4) * B * pool);
    grab((void**)&n_start[k], sizeof(int) * B * ((size_t)H + 1));
    grab((void**)&n_keys[k], sizeof(unsigned long long) * (size_t)key_cap);
  }
  grab((void**)&n_tmp, sizeof(float4) * (size_t)key_cap);
  grab((void**)&n_tseg, sizeof(int) * (size_t)tile_cap);
  grab((void**)&n_theads, sizeof(int) * (size_t)tile_cap);
  grab((void**)&n_tpref, sizeof(int) * ((size_t)tile_cap + 1));
  if (ok && prepare_map_grid(H)) ok = false;
  if (!ok) {
    void* fresh[] = {n_pool[0], n_pool[1], n_sorted[0], n_sorted[1], n_start[0], n_start[1], n_keys[0], n_keys[1], n_tmp, n_tseg, n_theads, n_tpref};
    for (void* q : fresh) if (q) (void)hipFree(q);
    c->err = "map pool of " + std::to_string(pool_points) + " points per sequence and class: allocation failed";
    return ALOAM_E_HIP;
  }
  for (int k = 0; k < 2; ++k) {
    if (old_pool) HIP_TRY(c, hipMemcpy2DAsync(n_pool[k], sizeof(float4) * pool, c->d_pool[k], sizeof(float4) * old_pool, sizeof(float4) * old_pool, B, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(c, hipMemsetAsync(n_pool[k], 0, sizeof(float4) * B * pool, c->stream));
    HIP_TRY(c, hipMemsetAsync(n_start[k], 0, sizeof(int) * B * ((size_t)H + 1), c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  void* old[] = {c->d_pool[0], c->d_pool[1], c->d_mgrid_sorted[0], c->d_mgrid_sorted[1], c->d_mgrid_start[0], c->d_mgrid_start[1], c->d_keys[0], c->d_keys[1],
                 c->d_voxtmp, c->d_tile_seg, c->d_tile_heads, c->d_tile_pref};
  for (void* q : old) if (q) (void)hipFree(q);
  for (int k = 0; k < 2; ++k) { c->d_pool[k] = n_pool[k]; c->d_mgrid_sorted[k] = n_sorted[k]; c->d_mgrid_start[k] = n_start[k]; c->d_keys[k] = n_keys[k]; }
  c->d_voxtmp = n_tmp; c->d_tile_seg = n_tseg; c->d_tile_heads = n_theads; c->d_tile_pref = n_tpref;
  c->map_pool = pool_points; c->map_H = H; c->map_key_cap = key_cap; c->map_tile_bound[1] = tile_bound1; c->map_tile_cap = tile_cap;
  c->map_cube_levels = 0;                                  // one 50 m cube may hold up to the whole pool: enough merge levels for that
  while (((size_t)kVoxTile << c->map_cube_levels) < pool) ++c->map_cube_levels;   // (levels a cube does not need cost one skipped tile loop each)
  return ALOAM_OK;
}

// The reference's cubes are std::vectors: a map grows as long as the sensor travels (src/laserMapping.cpp:737-783).  Here a (sequence,
// class) pool must hold the live points of its cubes plus what the step adds, and the steps are queued asynchronously, so the host sizes
// the pools AHEAD of the device from what k_map_report wrote after the last step that has finished: live points + (steps in flight + 1) x
// the most a step can add.  "The most": the stack sizes of the step are not known before its voxel filter has run, so it is the scan size
// (a stack is a filtered subset of one sweep) until a step has reported, then twice the largest stack any step has produced so far - a
// step that breaks that bound AND fills the pool drops points and raises ALOAM_E_CAPACITY like a full pool at the ceiling does.  When the
// bound exceeds the pool: wait for the device (the report is then exact), double the pool until it holds the bound, move the contents.
static int map_ensure_capacity(aloam_ctx* c) {
  if (c->map_pool >= c->map_pool_limit) return ALOAM_OK;     // at the ceiling: nothing to decide (the device counts what does not fit)
  const int hard[2] = {std::min(c->R * 120, c->nin_max ? c->nin_max : c->cap), std::min(c->cap, c->nin_max ? c->nin_max : c->cap)};
  auto bound = [&](long long lag) {
    const int done = c->h_map_report[0];
    long long worst = 0;
    for (int k = 0; k < 2; ++k) {
      const int inc = done > 0 ? std::min(hard[k], 2 * (int)c->h_map_report[3 + k] + 1024) : hard[k];
      worst = std::max(worst, (long long)c->h_map_report[1 + k] + (lag + 1) * inc);
    }
    return worst;
  };
  const long long lag = c->map_steps - c->h_map_report[0];
  if (bound(lag) <= c->map_pool) return ALOAM_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));               // now the report is that of the last queued step
  const long long need = bound(std::min<long long>(lag, 3));   // keep room for the run-ahead this caller has shown
  if (need <= c->map_pool || c->map_pool >= c->map_pool_limit) return ALOAM_OK;
  long long np = c->map_pool;
  while (np < need) np *= 2;
  np = std::min<long long>(np, c->map_pool_limit);
  const int rc = map_alloc_pool(c, (int)np);
  if (rc) { c->map_pool_limit = c->map_pool; return ALOAM_OK; }   // out of device memory: this pool is the ceiling from now on
  c->map_growths += 1;
  return ALOAM_OK;
}

int aloam_mapping_enable(aloam_ctx* c, float line_res, float plane_res, int pool_points) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  if (!(c->stages & ALOAM_STAGE_MAPPING)) { c->err = "this context was created without ALOAM_STAGE_MAPPING"; return ALOAM_E_STATE; }
  if (c->map_on) { c->err = "mapping already enabled"; return ALOAM_E_STATE; }
  if (!(line_res > 0.f) || !(plane_res > 0.f) || pool_points < 4096 || pool_points > (1 << 26)) { c->err = "bad mapping parameters (4096 <= pool_points <= 2^26)"; return ALOAM_E_ARG; }
  const size_t B = c->B, cap = c->cap, R = c->R;
  c->map_line_res = line_res; c->map_plane_res = plane_res;
  c->map_levels = 0;                                       // incoming clouds: up to max_points
  while (((size_t)kVoxTile << c->map_levels) < cap) ++c->map_levels;
  const size_t T = kVoxTile;
  c->map_tile_bound[0] = (int)(B * ((cap + T - 1) / T + (R * 120 + T - 1) / T));
  c->map_nsegs_max = (int)(B * 2 * kMapValidMax);
  int rc = 0;
  const int pool0 = (pool_points + 1023) / 1024 * 1024;
  if (c->map_pool_limit < pool0) c->map_pool_limit = pool0;
  if ((rc = map_alloc_pool(c, pool0))) return rc;
  if ((rc = dmalloc(c, &c->d_mapseq, B))) return rc;
  if ((rc = dmalloc(c, &c->d_cubes, B * 2 * kMapCubes))) return rc;
  if ((rc = dmalloc(c, &c->d_maptab, B * kTabInts))) return rc;
  if ((rc = dmalloc(c, &c->d_addcnt, B * 2 * kMapCubes))) return rc;
  if ((rc = dmalloc(c, &c->d_cursor, B * 2 * kMapCubes))) return rc;
  if ((rc = dmalloc(c, &c->d_compact_flag, B * 2))) return rc;
  if ((rc = dmalloc(c, &c->d_map_live, B * 2))) return rc;
  if ((rc = dmalloc(c, &c->d_map_report, 4))) return rc;
  HIP_TRY(c, hipHostMalloc((void**)&c->h_map_report, sizeof(int) * 8, hipHostMallocMapped));
  for (int k = 0; k < 8; ++k) c->h_map_report[k] = 0;
  HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_map_report_host, (void*)c->h_map_report, 0));
  for (hipEvent_t& e : c->map_step_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int k = 0; k < 2; ++k) {
    const size_t per = k == 0 ? R * 120 : cap;
    if ((rc = dmalloc(c, &c->d_stack[k], B * per))) return rc;
    if ((rc = dmalloc(c, &c->d_stack_world[k], B * per))) return rc;
    if ((rc = dmalloc(c, &c->d_stack_cube[k], B * per))) return rc;
  }
  c->rec_tiles_corner = (int)((R * 120 + 255) / 256);
  c->rec_tiles_per_seq = c->rec_tiles_corner + (int)((cap + 255) / 256);
  if ((rc = dmalloc(c, &c->d_rec_tiles, B * (size_t)c->rec_tiles_per_seq))) return rc;
  if ((rc = dmalloc(c, &c->d_medges, B * R * 120))) return rc;
  if ((rc = dmalloc(c, &c->d_mnorms, B * cap))) return rc;
  if ((rc = dmalloc(c, &c->d_registered, B * cap))) return rc;
  if ((rc = dmalloc(c, &c->d_knn, B * cap * 4))) return rc;
  if ((rc = dmalloc(c, &c->d_segs, (size_t)c->map_nsegs_max))) return rc;
  if ((rc = dmalloc(c, &c->d_vox_counters, 8))) return rc;
  if ((rc = dmalloc(c, &c->d_vox_lists, 3 * (size_t)c->map_nsegs_max))) return rc;
  if (prepare_voxel_filter()) { c->err = "k_vox_lds: dynamic LDS size rejected"; return ALOAM_E_HIP; }
  if ((rc = dmalloc(c, &c->d_bbox, (size_t)c->map_nsegs_max * 6))) return rc;
  std::vector<MapSeq> init(B);
  std::memset(init.data(), 0, sizeof(MapSeq) * B);
  for (size_t b = 0; b < B; ++b) {                       // reference src/laserMapping.cpp:72-74,109,115
    init[b].par[3] = 1.0; init[b].q_wmap_wodom[3] = 1.0;
    init[b].cen[0] = 10; init[b].cen[1] = 10; init[b].cen[2] = 5;
  }
  HIP_TRY(c, hipMemcpyAsync(c->d_mapseq, init.data(), sizeof(MapSeq) * B, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->map_on = true;
  return ALOAM_OK;
}

int aloam_set_voxel_sum_order(aloam_ctx* c, int order) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  if (order != ALOAM_SUM_INPUT_ORDER && order != ALOAM_SUM_REFERENCE_ORDER) { c->err = "unknown summation order"; return ALOAM_E_ARG; }
  if (order == ALOAM_SUM_REFERENCE_ORDER && prepare_reference_order()) { c->err = "k_vox_reference_order: dynamic LDS size rejected"; return ALOAM_E_HIP; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->sum_order = order;
  return ALOAM_OK;
}

int aloam_mapping_set_pool_limit(aloam_ctx* c, int max_pool_points) {
  if (!c) return ALOAM_E_ARG;
  if (max_pool_points < 4096 || max_pool_points > (1 << 26)) { c->err = "bad pool limit (4096 .. 2^26 points)"; return ALOAM_E_ARG; }
  c->map_pool_limit = std::max((max_pool_points + 1023) / 1024 * 1024, c->map_pool);
  return ALOAM_OK;
}

int aloam_get_map_pool_info(aloam_ctx* c, int out[4]) {
  DeviceScope device_scope(c);
  if (!c || !out) return ALOAM_E_ARG;
  if (!c->map_on) { c->err = "mapping not enabled"; return ALOAM_E_STATE; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out[0] = c->map_pool; out[1] = c->map_growths; out[2] = c->map_pool_limit;
  out[3] = std::max((int)c->h_map_report[1], (int)c->h_map_report[2]);
  return ALOAM_OK;
}

int aloam_mapping_step(aloam_ctx* c) {
  DeviceScope device_scope(c);
  if (!c) return ALOAM_E_ARG;
  if (!c->map_on) { c->err = "aloam_mapping_step before aloam_mapping_enable"; return ALOAM_E_STATE; }
  // at most four steps queued ahead of the device: the occupancy report the pools are sized from is never older than that
  hipEvent_t done = c->map_step_done[c->map_steps & 3];
  if (c->map_steps >= 4) HIP_TRY(c, hipEventSynchronize(done));
  if (c->inject_max > 0) { c->nin_max = c->inject_max; c->inject_max = 0; }   // this step's clouds came through aloam_set_last
  int rc = map_ensure_capacity(c);
  if (rc) return rc;
  const MapArgs a = map_args(c);
  { ProfScope p(c, K_MAP_BEGIN); launch_map_begin(a, c->stream); }
  { ProfScope p(c, K_MAP_VOXEL_STACK);                                      // downSizeFilterCorner / Surf on the incoming clouds (:542-550)
    const VoxArgs v = vox_args(c, c->B * 2, c->map_levels);
    HIP_TRY(c, hipMemsetAsync(c->d_vox_counters + 4, 0, 4 * sizeof(int), c->stream));   // general-path count, the two LDS-filter lists
    launch_map_stack_segments(a, v, c->stream);
    if (c->sum_order) launch_voxel_filter_reference_order(v, a, true, c->stream);
    else launch_voxel_filter(v, c->map_tile_bound[0], c->stream); }
  { ProfScope p(c, K_MAP_GRID); launch_map_grid(a, c->stream); }            // kdtree*FromMap->setInputCloud (:558-559)
  for (int iter = 0; iter < 2; ++iter) {                                    // :562
    { ProfScope p(c, K_MAP_ASSOC); launch_map_associate(a, iter, c->stream); }
    { ProfScope p(c, K_MAP_SOLVE); launch_map_solve(a, iter, iter == 1, c->stream); }
  }
  { ProfScope p(c, K_MAP_INSERT); launch_map_insert(a, c->d_voxtmp, c->stream); }        // :737-783
  { ProfScope p(c, K_MAP_VOXEL_CUBES);                                      // per-cube re-filter (:788-801)
    const VoxArgs v = vox_args(c, c->B * 2 * kMapValidMax, c->map_cube_levels);
    HIP_TRY(c, hipMemsetAsync(c->d_vox_counters + 4, 0, 4 * sizeof(int), c->stream));
    launch_map_cube_segments(a, v, c->stream);
    if (c->sum_order) launch_voxel_filter_reference_order(v, a, false, c->stream);
    else launch_voxel_filter(v, c->map_tile_bound[1], c->stream); }
  { ProfScope p(c, K_MAP_REGISTER); launch_map_register(a, c->stream);      // :836-846
    c->map_steps += 1;
    launch_map_report(a, (int)c->map_steps, c->stream); }
  HIP_TRY(c, hipEventRecord(done, c->stream));
  HIP_TRY(c, hipGetLastError());
  return ALOAM_OK;
}

int aloam_set_full_cloud(aloam_ctx* c, int seq, const float* cloud, int n) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (n < 0 || n > c->max_points) { c->err = "cloud too large"; return ALOAM_E_CAPACITY; }
  if ((rc = ensure_dense(c))) return rc;                // the other sequences' clouds of the last registration, before this one is replaced
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n) HIP_TRY(c, hipMemcpy(c->d_cloud + (size_t)seq * c->cap, cloud, sizeof(float4) * n, hipMemcpyHostToDevice));
  SeqMeta m;
  HIP_TRY(c, hipMemcpy(&m, c->d_meta + seq, sizeof(SeqMeta), hipMemcpyDeviceToHost));
  m.n_cloud = n;
  HIP_TRY(c, hipMemcpy(c->d_meta + seq, &m, sizeof(SeqMeta), hipMemcpyHostToDevice));
  return ALOAM_OK;
}

// The mapping node's globals for one sequence (reference src/laserMapping.cpp:72-74,84-91,115-116): what a test or a restarted node
// injects to continue from a known map.
int aloam_set_map(aloam_ctx* c, int seq, int cls, const int* cube_ids, const int* counts, int n_cubes, const float* points_xyzw) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!c->map_on) { c->err = "mapping not enabled"; return ALOAM_E_STATE; }
  if (cls < 0 || cls > 1 || n_cubes < 0 || (n_cubes && (!cube_ids || !counts))) { c->err = "bad class / cube list"; return ALOAM_E_ARG; }
  long long total = 0;
  std::vector<CubeDesc> d(kMapCubes, CubeDesc{0, 0, 0, 0});
  for (int i = 0; i < n_cubes; ++i) {
    if (cube_ids[i] < 0 || cube_ids[i] >= kMapCubes || counts[i] < 0 || d[cube_ids[i]].cap) { c->err = "bad or repeated cube index"; return ALOAM_E_ARG; }
    d[cube_ids[i]] = CubeDesc{(int)total, counts[i], counts[i], 0};
    total += counts[i];
  }
  if (total && !points_xyzw) return ALOAM_E_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (total > c->map_pool) {
    long long np = c->map_pool;
    while (np < total) np *= 2;
    if (np > c->map_pool_limit) { c->err = "the injected map exceeds the pool limit"; return ALOAM_E_CAPACITY; }
    if ((rc = map_alloc_pool(c, (int)np))) return rc;
    c->map_growths += 1;
  }
  HIP_TRY(c, hipMemcpy(c->d_cubes + ((size_t)seq * 2 + cls) * kMapCubes, d.data(), sizeof(CubeDesc) * kMapCubes, hipMemcpyHostToDevice));
  if (total) HIP_TRY(c, hipMemcpy(c->d_pool[cls] + (size_t)seq * c->map_pool, points_xyzw, sizeof(float4) * (size_t)total, hipMemcpyHostToDevice));
  MapSeq ms;
  HIP_TRY(c, hipMemcpy(&ms, c->d_mapseq + seq, sizeof(MapSeq), hipMemcpyDeviceToHost));
  ms.pool_used[cls] = (int)total;
  HIP_TRY(c, hipMemcpy(c->d_mapseq + seq, &ms, sizeof(MapSeq), hipMemcpyHostToDevice));
  c->h_map_report[1 + cls] = std::max((int)c->h_map_report[1 + cls], (int)total);   // the pools are sized from this until the next step reports
  return ALOAM_OK;
}

int aloam_set_map_frame(aloam_ctx* c, int seq, const int cen[3], const double q_wmap_wodom[4], const double t_wmap_wodom[3], int frame_count) {
  DeviceScope device_scope(c);
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!c->map_on) { c->err = "mapping not enabled"; return ALOAM_E_STATE; }
  if (!cen || !q_wmap_wodom || !t_wmap_wodom) return ALOAM_E_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  MapSeq ms;
  HIP_TRY(c, hipMemcpy(&ms, c->d_mapseq + seq, sizeof(MapSeq), hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; ++k) { ms.cen[k] = cen[k]; ms.t_wmap_wodom[k] = t_wmap_wodom[k]; }
  for (int k = 0; k < 4; ++k) ms.q_wmap_wodom[k] = q_wmap_wodom[k];
  ms.frame_count = frame_count;
  HIP_TRY(c, hipMemcpy(c->d_mapseq + seq, &ms, sizeof(MapSeq), hipMemcpyHostToDevice));
  return ALOAM_OK;
}

static int fetch_mapseq(aloam_ctx* c, int seq, MapSeq* ms) {
  int rc = check_seq(c, seq);
  if (rc) return rc;
  if (!c->map_on) { c->err = "mapping not enabled"; return ALOAM_E_STATE; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(ms, c->d_mapseq + seq, sizeof(MapSeq), hipMemcpyDeviceToHost));
  return ALOAM_OK;
}

int aloam_get_map_pose(aloam_ctx* c, int seq, double q_w_curr[4], double t_w_curr[3], double q_wmap_wodom[4], double t_wmap_wodom[3]) {
  DeviceScope device_scope(c);
  MapSeq ms;
  const int rc = fetch_mapseq(c, seq, &ms);
  if (rc) return rc;
  for (int k = 0; k < 4; ++k) { q_w_curr[k] = ms.par[k]; q_wmap_wodom[k] = ms.q_wmap_wodom[k]; }
  for (int k = 0; k < 3; ++k) { t_w_curr[k] = ms.par[4 + k]; t_wmap_wodom[k] = ms.t_wmap_wodom[k]; }
  return ALOAM_OK;
}

int aloam_get_map_info(aloam_ctx* c, int seq, int out[16]) {
  DeviceScope device_scope(c);
  MapSeq ms;
  const int rc = fetch_mapseq(c, seq, &ms);
  if (rc) return rc;
  const int v[16] = {ms.cen[0], ms.cen[1], ms.cen[2], ms.frame_count, ms.from_total[0], ms.from_total[1], ms.n_stack[0], ms.n_stack[1],
                     ms.factor_num[0][0], ms.factor_num[1][0], ms.factor_num[0][1], ms.factor_num[1][1], ms.lm_iterations[0], ms.lm_iterations[1],
                     ms.lm_termination[0], ms.compactions};
  std::memcpy(out, v, sizeof(v));
  return ALOAM_OK;
}

int aloam_map_cube_counts(aloam_ctx* c, int seq, int cls, int* out) {
  DeviceScope device_scope(c);
  MapSeq ms;
  const int rc = fetch_mapseq(c, seq, &ms);
  if (rc) return rc;
  if (cls < 0 || cls > 1) { c->err = "class must be 0 (corner) or 1 (surf)"; return ALOAM_E_ARG; }
  std::vector<CubeDesc> d(kMapCubes);
  HIP_TRY(c, hipMemcpy(d.data(), c->d_cubes + ((size_t)seq * 2 + cls) * kMapCubes, sizeof(CubeDesc) * kMapCubes, hipMemcpyDeviceToHost));
  for (int i = 0; i < kMapCubes; ++i) out[i] = d[i].cnt;
  return kMapCubes;
}

int aloam_get_map_cube(aloam_ctx* c, int seq, int cls, int cube, float* out, int cap_points) {
  DeviceScope device_scope(c);
  MapSeq ms;
  const int rc = fetch_mapseq(c, seq, &ms);
  if (rc) return rc;
  if (cls < 0 || cls > 1 || cube < 0 || cube >= kMapCubes) { c->err = "bad class / cube index"; return ALOAM_E_ARG; }
  CubeDesc d;
  HIP_TRY(c, hipMemcpy(&d, c->d_cubes + ((size_t)seq * 2 + cls) * kMapCubes + cube, sizeof(CubeDesc), hipMemcpyDeviceToHost));
  const int k = d.cnt < cap_points ? d.cnt : cap_points;
  if (k > 0) HIP_TRY(c, hipMemcpy(out, c->d_pool[cls] + (size_t)seq * c->map_pool + d.off, sizeof(float4) * k, hipMemcpyDeviceToHost));
  return d.cnt;
}

int aloam_get_map_cloud(aloam_ctx* c, int seq, int which, float* out, int cap_points) {
  DeviceScope device_scope(c);
  MapSeq ms;
  const int rc = fetch_mapseq(c, seq, &ms);
  if (rc) return rc;
  const float4* p = nullptr;
  int n = 0;
  if (which == ALOAM_MAP_REGISTERED) {
    SeqMeta m;
    HIP_TRY(c, hipMemcpy(&m, c->d_meta + seq, sizeof(SeqMeta), hipMemcpyDeviceToHost));
    p = c->d_registered + (size_t)seq * c->cap; n = m.n_cloud;
  } else if (which == ALOAM_MAP_CORNER_STACK) { p = c->d_stack[0] + (size_t)seq * c->R * 120; n = ms.n_stack[0]; }
  else if (which == ALOAM_MAP_SURF_STACK) { p = c->d_stack[1] + (size_t)seq * c->cap; n = ms.n_stack[1]; }
  else { c->err = "unknown map cloud id"; return ALOAM_E_ARG; }
  const int k = n < cap_points ? n : cap_points;
  if (k > 0) HIP_TRY(c, hipMemcpy(out, p, sizeof(float4) * k, hipMemcpyDeviceToHost));
  return n;
}

}  // extern "C"
