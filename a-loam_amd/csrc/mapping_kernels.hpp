// a-loam_amd/csrc/mapping_kernels.hpp — layouts and launchers of the scan-to-map refinement (reference src/laserMapping.cpp).
#pragma once
#include "aloam_device.hpp"

namespace aloam {

constexpr int kMapW = 21, kMapH = 21, kMapD = 11, kMapCubes = kMapW * kMapH * kMapD;   // reference src/laserMapping.cpp:75-80
constexpr int kMapValidMax = 75;                                                       // 5 x 5 x 3 window (:512-529)
constexpr int kMapGridMaxH = 32768;                                                    // bucket table of k_mapgrid_build (LDS)
constexpr int kVoxTile = 2048;                                                         // keys per sort tile (general path)
constexpr int kVoxTinyN = 2048, kVoxSmallN = 8192, kVoxBigN = 65536;                                    // segment sizes the single-workgroup LDS filter takes (256 / 1024 threads)

enum MapErrBits { kMapErrPool = 1, kMapErrKeys = 2, kMapErrSegment = 4 };

struct CubeDesc { int off, cnt, cap, pad; };             // one map cube of one class: segment [off, off + cap) of the class pool

struct alignas(16) MapSeq {                              // one per sequence
  double par[7];                                         // `parameters`: q_w_curr (x,y,z,w), t_w_curr      (:109-111)
  double q_wmap_wodom[4], t_wmap_wodom[3];               // (:115-116)
  double q_wodom[4], t_wodom[3];                         // odometry pose this frame was started from        (:290-296)
  int cen[3];                                            // laserCloudCenWidth / Height / Depth               (:72-74)
  int center[3];                                         // centerCubeI / J / K after the shifts
  int frame_count;
  int n_valid;
  int from_total[2];                                     // laserCloudCornerFromMapNum / SurfFromMapNum
  int gate;                                              // from_total[0] > 10 && from_total[1] > 50          (:554)
  int n_stack[2];                                        // laserCloudCornerStackNum / SurfStackNum
  int factor_num[2][2];                                  // [iteration][class]
  int lm_iterations[2], lm_termination[2];
  int pool_used[2];
  int err;
  int compactions;                                        // pool compactions so far (k_map_compact_*)
  int err_steps;                                          // earlier steps of this sequence that ended with err != 0 (folded in by k_map_begin)
  int pad;
};

// Factor records of the valid stack points (k_map_fit -> k_map_solve; every LM evaluation streams them again, and k_map_solve runs at HBM
// speed: bytes are its time).  curr_point is a float point of the stack, so it is kept as three floats (the same doubles come back when it
// is read): 64 / 48 bytes instead of 80 / 64.  pad: the stack index of the point.
struct MapEdgeRec { double a[3], b[3]; float cp[3]; int pad; };         // LidarEdgeFactor(curr_point, point_a, point_b, 1.0)      (:618)
struct MapNormRec { double n[3], d; float cp[3]; int pad; };            // LidarPlaneNormFactor(curr_point, norm, negative_OA_dot_norm) (:683)
static_assert(sizeof(MapEdgeRec) == 64 && sizeof(MapNormRec) == 48, "record sizes");

struct VoxSeg {                                          // one pcl::VoxelGrid::filter call
  const float4* in;
  float4* out;                                           // ascending-voxel centroids
  int* out_count;
  float4* final_out;                                     // optional: copy the result here afterwards (in-place cube re-filter)
  int* final_count;
  int n;
  float leaf;
  int key_off, tile0, ntiles, pad;
};

struct VoxArgs {
  VoxSeg* segs;
  int n_segs;
  int* tile_seg;           // [tile_cap] owning segment of every tile
  int* tile_heads;         // [tile_cap]
  int* tile_pref;          // [tile_cap + 1]
  int* counters;           // [0] total tiles, [1] error of this step, [2] merge levels needed, [3] earlier steps with an error,
                           // [4] segments left to the general path, [5] / [6] / [7] lengths of the three LDS-filter lists
  int* lists;              // [3][n_segs] segment ids for k_vox_lds (small, big, tiny)
  unsigned long long* keys[2];
  float4* tmp;             // [key_cap]
  int* bbox;               // [n_segs][6] order-preserving ints
  int tile_cap;
  long long key_cap;
  int levels;
};

struct MapArgs {
  int B, cap, R;
  SeqMeta* meta;
  OdomState* odom;
  MapSeq* seq;
  float line_res, plane_res;
  const float4* corner_last;     // [B][R*120]  /laser_cloud_corner_last
  const float4* surf_last;       // [B][cap]    /laser_cloud_surf_last
  const float4* full;            // [B][cap]    /velodyne_cloud_3, dense (set from outside, or made by k_dense_cloud)
  const float4* slabs; int slab; const int* ringstart;   // ... or, straight from scan registration: one slab per ring + the dense start of every ring (slabs == nullptr: use `full`)
  float4* registered;            // [B][cap]    /velodyne_cloud_registered
  CubeDesc* cubes;               // [B][2][kMapCubes]
  float4* pool[2];               // [B][pool_cap]
  int pool_cap;
  int* tab;                      // [B][kTabInts]  valid cubes + submap prefixes
  float4* stack[2];              // [B][R*120] / [B][cap]   laserCloudCornerStack / SurfStack
  float4* stack_world[2];        // same shapes: the stacks transformed with the refined pose (:739, :762)
  int* stack_cube[2];            // cube index of every stack point, -1 outside the window
  int* addcnt;                   // [B][2][kMapCubes]
  int* cursor;                   // [B][2][kMapCubes]   append cursors (insertion) / new offsets (compaction)
  int* compact_flag;             // [B][2]
  float4* grid_sorted[2];        // [B][pool_cap]
  int* grid_start[2];            // [B][H + 1]
  int grid_H;                    // buckets of the submap hash, the same for both classes
  int* live;                     // [B][2] live points of every (sequence, class) after the step (k_map_report)
  int* report_dev;               // [3] ticket of k_map_report, largest corner / surf stack so far
  int* report_host;              // pinned host memory: step, largest live corner / surf, largest stack corner / surf
  float4* knn;                   // [B][cap][4]  the five neighbours of every stack point (search -> fit)
  MapEdgeRec* edges;             // [B][R*120]
  MapNormRec* norms;             // [B][cap]
  int lm_max_iterations;
  int* rec_tiles;                // [B][rec_tiles_per_seq] valid factor records per tile of 256 stack points: corner tiles, then surf tiles
  int rec_tiles_per_seq, rec_tiles_corner;
  int* vox_counters;             // VoxArgs::counters: [1] capacity flag of this step's voxel filters, [3] earlier steps that raised it
};
constexpr int kTabInts = 256;   // [0..74] valid cube ids, [80..155] corner prefix, [160..235] surf prefix

void launch_map_begin(const MapArgs& a, hipStream_t s);
void launch_map_stack_segments(const MapArgs& a, const VoxArgs& v, hipStream_t s);
void launch_map_cube_segments(const MapArgs& a, const VoxArgs& v, hipStream_t s);
void launch_voxel_filter(const VoxArgs& v, int tile_bound, hipStream_t s);
int prepare_voxel_filter();
int prepare_map_grid(int H);
void launch_map_grid(const MapArgs& a, hipStream_t s);
void launch_map_associate(const MapArgs& a, int iter, hipStream_t s);
void launch_map_solve(const MapArgs& a, int iter, bool last, hipStream_t s);
void launch_map_insert(const MapArgs& a, float4* staging, hipStream_t s);   // staging: 2 pools per sequence
void launch_map_register(const MapArgs& a, hipStream_t s);   // reads the slabs when a.slabs is set, the dense cloud otherwise
void launch_map_report(const MapArgs& a, int step, hipStream_t s);
int prepare_reference_order();                                                                  // reference_order_kernels.hip
void launch_voxel_filter_reference_order(const VoxArgs& v, const MapArgs& a, bool stacks, hipStream_t s);

}  // namespace aloam
