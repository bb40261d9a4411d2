// a-loam_amd/csrc/aloam_device.hpp — device-side layouts and helpers shared by the gfx950 kernels.
//
// Compiled with -ffp-contract=off: every decision quantity (range test, ring id, relative time, curvature,
// gap test, voxel index, squared distances) is evaluated with the individually rounded IEEE operations the
// reference's x86-64 build performs, in the same order, so discrete decisions match bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "aloam_atan.hpp"

namespace aloam {

constexpr int kFrontSlots = 128 + 8;   // look-back granules per k_front block: one per ring (kMaxRings), slot R = halfPassed index
constexpr int kBlockPts = 1024;        // points per k_front workgroup (256 threads x 4)
constexpr int kMaxRings = 128;
constexpr int kSectors = 6;            // reference src/scanRegistration.cpp:282
constexpr int kSharpPerSector = 2;     // :301
constexpr int kLessSharpPerSector = 20;// :307
constexpr int kFlatPerSector = 4;      // :359
constexpr int kNnTile = 1024;          // targets staged in LDS per NN workgroup

enum ErrBits { kErrEmpty = 1, kErrRingCap = 2, kErrPointCap = 4, kErrInternal = 8 };   // kErrInternal: a look-back wait of k_ring_features timed out

struct alignas(16) SeqMeta {           // one per sequence, device resident
  int n_in;
  int first_kept, last_kept;
  int half_idx;                        // index of the point that flips halfPassed (src/scanRegistration.cpp:220-223)
  float start_ori, end_ori;            // :141-153
  int n_cloud;
  int err;
  int n_sharp, n_less_sharp, n_flat, n_less_flat;   // current sweep
  int n_corner_last, n_surf_last;                   // previous sweep (laserCloudCornerLast / SurfLast)
  int pad0, pad1;
};

struct alignas(16) OdomState {         // one per sequence, device resident
  double para_q[4];                    // src/laserOdometry.cpp:97  (x,y,z,w)
  double para_t[3];                    // :98
  double q_w[4];                       // :93       (x,y,z,w)
  double t_w[3];                       // :94
  int corner_corr[2], plane_corr[2];
  int lm_iterations[2], lm_successful[2];
  double initial_cost[2], final_cost[2];
  int termination[2];
  int pad[2];
};

struct EdgeRec { float cp[3], a[3], b[3]; int valid; int pad[2]; };          // 48 B : LidarEdgeFactor ctor args
struct PlaneRec { float cp[3], j[3], l[3], m[3]; int valid; int pad[3]; };   // 64 B : LidarPlaneFactor ctor args

// Everything the registration kernels need; passed by value.
struct RegArgs {
  const char* in;            // batch of input scans
  long long seq_stride;      // bytes between sequences
  int pt_stride;             // bytes between points
  int B, cap, R, NB;         // batch, max_points, rings, blocks per scan
  int ring_from_field;
  float min_range;
  SeqMeta* meta;             // [B]
  float4* slabs;             // [B][R][slab] ring-ordered points, ring r of a sweep in its own slab (k_front)
  int slab;                  // points per slab (>= the longest ring k_ring_features accepts)
  unsigned long long* front_lb;   // [B][NB][kFrontSlots] {epoch, state, value} granules of k_front's look-back over the blocks of a sweep
  int* front_ticket;         // [B] blocks of the sweep handed out so far (k_front takes, k_ring_starts resets)
  int* ringstart;            // [B][R+1] start of every ring in the DENSE numbering (scanStartInd - 5)
  float4* cloud;             // [B][cap] ring-ordered, dense: written by k_dense_cloud when a consumer of the full cloud asks
  float* curv;               // [B][cap]
  int8_t* label;             // [B][cap]
  unsigned long long* lookback;   // [B][4][R] {launch epoch, count} granules: points per ring and output class (k_ring_features)
  unsigned epoch;            // this launch; never 0 (the buffer starts zeroed)
  int* ring_ticket;          // [B] rings of the sweep handed out so far (k_ring_features takes, k_cloud_sizes resets)
  int store_debug;           // 1: also write curv / label (parity tests); the throughput entries leave them out
  float4* sharp;             // [B][R*12]
  float4* less_sharp;        // [B][R*120]   (current buffer)
  float4* flat;              // [B][R*24]
  float4* less_flat;         // [B][cap]     (current buffer)
};

struct OdomArgs {
  int B, cap, R;
  SeqMeta* meta;
  OdomState* state;
  const float4* sharp;       // [B][R*12]
  const float4* flat;        // [B][R*24]
  const float4* corner_last; // [B][R*120]
  const float4* surf_last;   // [B][cap]
  // spatial hash grids over the last clouds (k_build_grids): index 0 = corner_last, 1 = surf_last
  float4* grid_sorted3[2];   // [B][R*120] / [B][cap]   entries bucketed by (ix,iy,iz)
  float4* grid_sorted2[2];   //                          entries bucketed by (ix,iy,ring key)
  int* grid_start3[2];       // [B][H+1]
  int* grid_start2[2];       // [B][H+1]
  float4* grid_sorted3c[2];  // coarse levels of the same two grids: they bound the search of far queries
  int* grid_start3c[2];      // [B][H+1]
  int* grid_flags[2];        // [B][4]   flags[0] != 0: keys / coordinates out of range -> literal brute-force path; flags[1]: 0 = ring-sorted, 1 = NEARLY ring-sorted
                             //          (no key more than 2 below an earlier one: the reference's walks still visit one index range, grid_walk holds its ends),
                             //          2 = not sorted -> literal walks; flags[2] != 0: the coarse level holds 16-bit positions into the fine copy (k_build_grids_fused)
  int* grid_walk[2];         // [B][2][R + 8]   per ring key k: first index with key >= k, last index with key <= k (written for nearly-sorted clouds only)
  int grid_H_corner, grid_H_surf;   // buckets (power of two, multiple of 1024)
  float4* sel_sharp;         // [B][R*12]  features moved to the start of the sweep with the current pose (k_transform_queries)
  float4* sel_flat;          // [B][R*24]
  EdgeRec* edges;            // [B][R*12]
  PlaneRec* planes;          // [B][R*24]
  int outer;                 // which outer iteration (0/1)
  int last_outer;            // 1: integrate the pose after solving (src/laserOdometry.cpp:504-505)
  int lm_max_iterations;
  int distortion;            // 1: per-point interpolation ratio (reference DISTORTION 1); 0: s = 1
};

struct GridView {
  int H;
  float4 *sorted3, *sorted2, *sorted3c;
  int *start3, *start2, *start3c, *flags, *walk;
};
__device__ __forceinline__ GridView grid_view(const OdomArgs& a, int b, int which) {
  GridView g;
  g.H = which == 0 ? a.grid_H_corner : a.grid_H_surf;
  const long long per = which == 0 ? (long long)a.R * 120 : (long long)a.cap;
  g.sorted3 = a.grid_sorted3[which] + b * per;
  g.sorted2 = a.grid_sorted2[which] + b * per;
  g.start3 = a.grid_start3[which] + (long long)b * (g.H + 1);
  g.start2 = a.grid_start2[which] + (long long)b * (g.H + 1);
  g.sorted3c = a.grid_sorted3c[which] + b * per;
  g.start3c = a.grid_start3c[which] + (long long)b * (g.H + 1);
  g.flags = a.grid_flags[which] + b * 4;
  g.walk = a.grid_walk[which] + (long long)b * 2 * (a.R + 8);
  return g;
}

__device__ __forceinline__ float4 load_point(const char* base, long long i, int stride) {
  const float* p = reinterpret_cast<const float*>(base + i * (long long)stride);
  if ((stride & 15) == 0) return *reinterpret_cast<const float4*>(p);
  if (stride == 12) return make_float4(p[0], p[1], p[2], 0.f);                // x, y, z only: the reference never reads the 4th input float
  return make_float4(p[0], p[1], p[2], p[3]);
}

// NaN removal + removeClosedPointCloud (reference src/scanRegistration.cpp:136-137, :99): f32 arithmetic.
__device__ __forceinline__ bool point_kept(const float4& p, float thres) {
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return false;
  const float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
  return !(r2 < thres * thres);
}

// 64-bit wave shuffles
__device__ __forceinline__ double shfl_down_f64(double v, int d) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_down(lo, d, 64);
  hi = __shfl_down(hi, d, 64);
  return __hiloint2double(hi, lo);
}

// ---- wave-wide max / min of a 64-bit key without touching the LDS crossbar ---------------------------------------
// value of lane ^ D inside a row of 16 lanes: D = 1, 2 one DPP quad permute; D = 4, 8 two DPP row shifts + a select.
template <int D>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned v, int lane) {
  if constexpr (D == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
  else if constexpr (D == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
  else {
    const unsigned up = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x100 + D, 0xF, 0xF, true);            // row_shl:D  -> lane + D
    const unsigned dn = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x110 + D, 0xF, 0xF, true);            // row_shr:D  -> lane - D
    return (lane & D) ? dn : up;
  }
}
template <int D, bool MAX>
__device__ __forceinline__ unsigned long long row_step_u64(unsigned long long v, int lane) {
  const unsigned lo = xor_lane_u32<D>((unsigned)v, lane), hi = xor_lane_u32<D>((unsigned)(v >> 32), lane);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return MAX ? (o > v ? o : v) : (o < v ? o : v);
}
template <bool MAX>
__device__ __forceinline__ unsigned long long wave_extreme_u64(unsigned long long v, int lane) {
  v = row_step_u64<1, MAX>(v, lane);
  v = row_step_u64<2, MAX>(v, lane);
  v = row_step_u64<4, MAX>(v, lane);
  v = row_step_u64<8, MAX>(v, lane);                                          // every lane holds the extreme of its row of 16
  unsigned long long best = MAX ? 0ull : ~0ull;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * q), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * q);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    best = MAX ? (o > best ? o : best) : (o < best ? o : best);
  }
  return best;                                                                // wave-uniform
}

// Wave-wide max / min of a 32-bit value: the same DPP ladder as the scans below (row_shr 1, 2, 4, 8, then row_bcast 15 / 31 carry the
// row results upward); the compiler folds every step into one v_max/min_u32_dpp, lane 63 ends up with the result, one v_readlane
// makes it wave-uniform.  7 VALU instructions instead of the ~35 of the xor-butterfly + four readlanes.
template <bool MAX>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned v) {
  constexpr int identity = MAX ? 0 : -1;
  auto op = [](unsigned a, unsigned b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(identity, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(identity, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(identity, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(identity, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(identity, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp(identity, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Minimum of packed (high word = f32 distance bits, low word = tie-break) keys: reduce the high words; only when several lanes
// hold the minimal high word (exactly equal f32 distances, or nobody holds a candidate at all) a second reduction settles the
// low word.  Same value as the 64-bit reduction, a third of its instructions in the common case.
__device__ __forceinline__ unsigned long long wave_min_packed(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_reduce_u32<false>(hi);
  const unsigned long long tie = __ballot(hi == mh);
  unsigned ml;
  if (__popcll(tie) == 1) ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)tie) - 1);
  else ml = wave_reduce_u32<false>(hi == mh ? lo : 0xffffffffu);
  return ((unsigned long long)mh << 32) | ml;
}

// Inclusive wave64 scans on the DPP network (no LDS crossbar): Hillis-Steele inside each row of 16 lanes, then the row
// totals travel with row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3).  `identity` fills the lanes a shift
// leaves without a source.
template <bool MAX>
__device__ __forceinline__ int wave_scan_i32(int v) {
  constexpr int identity = MAX ? (int)0x80000000 : 0;
  auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : a + b; };
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xF, 0xF, false));   // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xF, 0xF, false));   // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xF, 0xF, false));   // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xF, 0xF, false));   // row_shr:8
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return v;
}


// LDS bitonic sort of n 32- or 64-bit keys, ascending, 256 threads.  The network is the all-ascending form (a "flip" stage
// i <-> block_end - i opens every merge, half-cleaners follow), so the slots n .. pow2ceil(n)-1 can stay imaginary +inf:
// a pair whose upper partner lies beyond n is simply skipped, which removes a third of the LDS traffic at the typical
// n ~ 0.7 * pow2ceil(n).  The stages that stay inside aligned groups of eight keys (the merges k = 2, 4 and 8, and the last
// three half-cleaners of every later merge) run in registers: one read and one write of the group instead of one per stage.
// Starts and ends with the data visible to the whole workgroup.
__device__ __forceinline__ int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
template <typename K>
__device__ __forceinline__ void bitonic_cx(K& a, K& b) {
  const K lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo; b = hi;
}
template <bool FIRST, typename K>
__device__ __forceinline__ void bitonic_groups_of_eight(K* keys, int n, int tid) {
  constexpr K kInf = (K)~(K)0;
  for (int g = tid * 8; g < n; g += 2048) {
    K v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = g + u < n ? keys[g + u] : kInf;
    if (FIRST) {                                                             // merges k = 2, 4 and 8
#pragma unroll
      for (int b = 0; b < 8; b += 2) bitonic_cx(v[b], v[b + 1]);
#pragma unroll
      for (int b = 0; b < 8; b += 4) { bitonic_cx(v[b], v[b + 3]); bitonic_cx(v[b + 1], v[b + 2]); }
#pragma unroll
      for (int b = 0; b < 8; b += 2) bitonic_cx(v[b], v[b + 1]);
#pragma unroll
      for (int o = 0; o < 4; ++o) bitonic_cx(v[o], v[7 - o]);
    } else {                                                                 // half-cleaner j = 4
#pragma unroll
      for (int o = 0; o < 4; ++o) bitonic_cx(v[o], v[o + 4]);
    }
#pragma unroll
    for (int b = 0; b < 8; b += 4) { bitonic_cx(v[b], v[b + 2]); bitonic_cx(v[b + 1], v[b + 3]); }   // j = 2
#pragma unroll
    for (int b = 0; b < 8; b += 2) bitonic_cx(v[b], v[b + 1]);                                          // j = 1
#pragma unroll
    for (int u = 0; u < 8; ++u) if (g + u < n) keys[g + u] = v[u];
  }
  __syncthreads();
}
template <typename K>
__device__ __forceinline__ void bitonic_sort_keys(K* keys, int n, int tid) {
  const int npad = pow2ceil(n);
  bitonic_groups_of_eight<true, K>(keys, n, tid);
  for (int k = 16; k <= npad; k <<= 1) {
    const int hk = k >> 1;
    for (int t = tid; t < (npad >> 1); t += 256) {                         // flip stage
      const int base = (t / hk) * k, off = t & (hk - 1);
      const int i = base + off, l = base + (k - 1 - off);
      if (l < n) {
        const K x = keys[i], y = keys[l];
        if (x > y) { keys[i] = y; keys[l] = x; }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 4; j >>= 1) {
      for (int t = tid; t < (npad >> 1); t += 256) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i + j;
        if (l < n) {
          const K x = keys[i], y = keys[l];
          if (x > y) { keys[i] = y; keys[l] = x; }
        }
      }
      __syncthreads();
    }
    bitonic_groups_of_eight<false, K>(keys, n, tid);
  }
}
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int n, int tid) { bitonic_sort_keys<unsigned long long>(keys, n, tid); }


}  // namespace aloam
