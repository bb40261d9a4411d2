// a-loam_amd/csrc/mapping_kernels.hip — gfx950 kernels for A-LOAM scan-to-map refinement.
//
// Replaces, for a BATCH of independent sequences, the body of process() in the reference's src/laserMapping.cpp:231-893
// (one frame per call, no frame dropping) and the third-party calls inside it.  Kernel <-> reference map:
//   k_map_compact_*    (no counterpart: the reference's cubes are std::vectors) packs the class pools when a frame's growth
//                      no longer fits behind the bump pointer
//   k_map_begin        :142-146 transformAssociateToMap, :311-321 centre cube, :323-507 window shifts (the 21 x 21 x 11
//                      pointer grid becomes a table of cube descriptors), :509-539 valid cubes + submap prefixes
//   k_vox_lds          pcl::VoxelGrid::filter (:542-550 incoming clouds, :788-801 per-cube re-filter) of one segment by one workgroup:
//                      run heads -> (voxel, first point) keys -> stable radix sort in registers / LDS -> centroids in input order
//   k_vox_*            the same through global memory for segments that do not fit (tile sort + rank-merge levels)
//   k_mapgrid_build    pcl::KdTreeFLANN::setInputCloud (:558-559): 2 m cell hash over the submap, LDS counting sort per (sequence, class)
//   k_map_search/_fit  :576-706  pointAssociateToMap, nearestKSearch(k = 5) as an exact fixed-radius search (the reference
//                      only uses the result when the 5th neighbour is closer than 1 m), line fit (3x3 symmetric
//                      eigen-decomposition) / plane fit (5x3 least squares), valid factor records compacted per tile of 256 points
//   k_map_solve        :565-572,712-720 ceres::Solve over LidarEdgeFactor + LidarPlaneNormFactor blocks (shared LM loop,
//                      lm_device.hpp), then :148-152 transformUpdate
//   k_map_cubeid / k_map_reserve / k_map_scatter   :737-783 map insertion (stable append per cube)
//   k_map_register     :836-846 /velodyne_cloud_registered
// Everything is integer / f32 / f64 scalar work on 16-byte point records: HBM- and latency-bound, no MFMA.
#include "mapping_kernels.hpp"

#include "lm_device.hpp"

namespace aloam {

namespace {

__device__ __forceinline__ int f2o(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }   // order-preserving
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ __forceinline__ unsigned hash_cell(int a, int b, int c) {
  return ((unsigned)a * 73856093u) ^ ((unsigned)b * 19349663u) ^ ((unsigned)c * 83492791u);
}
// Bucket of a 2 m map cell: the low kMapLocalBits bits of each cell coordinate, interleaved (x, y, z, x, y, z, ...), are the low bits of
// the bucket, the hash of the super-cell of 2^kMapLocalBits cells per axis the rest.  Neighbouring cells differ in a parity, so the
// eight cells of any 2x2x2 block land in eight DIFFERENT buckets (k_map_search walks them without a duplicate test), and the cells of
// one super-cell are neighbours in the bucket table and therefore in the sorted copy of the submap: the 64 queries of a wave, close
// together in space, read close together in memory.
constexpr int kMapLocalBits = 1;
static_assert(kMapLocalBits >= 1 && kMapLocalBits <= 4, "k_map_search walks the eight cells of a 2x2x2 block without a duplicate test: they must land in eight different buckets");
__device__ __forceinline__ unsigned map_local_bits(int v, int axis) {      // bit i of v -> bit 3 i + axis
  unsigned l = 0;
#pragma unroll
  for (int i = 0; i < kMapLocalBits; ++i) l |= (unsigned)((v >> i) & 1) << (3 * i + axis);
  return l;
}
__device__ __forceinline__ unsigned map_bucket(int x, int y, int z, int H) {
  return ((hash_cell(x >> kMapLocalBits, y >> kMapLocalBits, z >> kMapLocalBits) << (3 * kMapLocalBits)) | map_local_bits(x, 0) | map_local_bits(y, 1) | map_local_bits(z, 2)) & (unsigned)(H - 1);
}

// Hamilton product a * b (x,y,z,w storage), as Eigen evaluates it.
__device__ __forceinline__ void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// pointAssociateToMap (reference src/laserMapping.cpp:157-166): f64 rotation + translation, stored back to f32.
__device__ __forceinline__ float4 associate_to_map(const float4& p, const double par[7]) {
  double o[3];
  quat_rotate(par, (double)p.x, (double)p.y, (double)p.z, o);
  return make_float4((float)(o[0] + par[4]), (float)(o[1] + par[5]), (float)(o[2] + par[6]), p.w);
}

// int((v + 25.0) / 50.0) + cen, minus one when v + 25 < 0  (:312-321, :741-750)
__device__ __forceinline__ int cube_coord(double v, int cen) {
  int c = (int)((v + 25.0) / 50.0) + cen;
  if (v + 25.0 < 0) c--;
  return c;
}

// kd-tree stand-in of the submap: 2 m cells.  Every point closer than 1 m to a query lies in the 2x2x2 block of cells made of the
// query's own cell and, per axis, the neighbour on the side of the cell the query sits in (8 bucket look-ups instead of the 27 a
// 1 m grid needs).  Power-of-two cell size: p * 0.5f is exact, so cell membership is decided without rounding.
constexpr float kMapCellInv = 0.5f;

__device__ __forceinline__ CubeDesc* cube_table(const MapArgs& a, int b, int cls) { return a.cubes + ((long long)b * 2 + cls) * kMapCubes; }

// submap lookup: g-th point of the concatenated valid cubes of class cls (laserCloudCornerFromMap / SurfFromMap order)
__device__ __forceinline__ float4 submap_point(const MapArgs& a, int b, int cls, const int* tab, int n_valid, int g) {
  const int* pref = tab + 80 + cls * 80;
  int lo = 0, hi = n_valid - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pref[mid] <= g) lo = mid; else hi = mid - 1; }
  const CubeDesc d = cube_table(a, b, cls)[tab[lo]];
  return a.pool[cls][(long long)b * a.pool_cap + d.off + (g - pref[lo])];
}

}  // namespace

// =======================================================================================================
// pool compaction
// =======================================================================================================
// Cubes that outgrow their segment move to a fresh one and abandon the old one (k_map_reserve), so a growing map hands out
// the class pool faster than it fills it.  When the moves of this frame no longer fit behind the bump pointer, the live cubes
// are packed to the front instead (through a staging buffer, order inside every cube untouched), each with room for this
// frame's points and half as much again - or, when even that is too much, with exactly the room this frame needs.
__device__ __forceinline__ int compact_cap(int n, int mode) { return n <= 0 ? 0 : (mode == 1 ? n + n / 2 + 64 : n); }

__global__ __launch_bounds__(256) void k_map_compact_plan(MapArgs a) {
  const int b = blockIdx.x, cls = blockIdx.y, tid = threadIdx.x;
  MapSeq& ms = a.seq[b];
  int* flag = a.compact_flag + b * 2 + cls;
  const CubeDesc* T = cube_table(a, b, cls);
  const int* add = a.addcnt + ((long long)b * 2 + cls) * kMapCubes;
  int* newoff = a.cursor + ((long long)b * 2 + cls) * kMapCubes;
  __shared__ int s_part[256], s_sum[3];
  constexpr int PER = (kMapCubes + 255) / 256;
  if (tid < 3) s_sum[tid] = 0;
  __syncthreads();
  // First only what this frame asks for: the descriptors of the dozen cubes that receive points (the append counts of all 4851
  // cubes are 19 KB per class, their descriptors 78 KB: reading those every frame although a compaction is rare was 1.7 GB per
  // launch).  The sums over ALL cubes are formed only when the reservations do not fit behind the bump pointer.
  int need = 0;
  for (int k = 0; k < PER; ++k) {
    const int c = tid * PER + k;
    if (c >= kMapCubes) break;
    const int ad = add[c];
    if (ad > 0) {
      const CubeDesc d = T[c];
      const int n = d.cnt + ad;
      if (n > d.cap) need += 2 * n < 256 ? 256 : 2 * n;                      // what k_map_reserve would take
    }
  }
  if (need) atomicAdd(&s_sum[0], need);
  __syncthreads();
  if (ms.pool_used[cls] + s_sum[0] <= a.pool_cap) { if (tid == 0) *flag = 0; return; }   // fits as it is: the normal frame
  int roomy = 0, tight = 0;
  for (int k = 0; k < PER; ++k) {
    const int c = tid * PER + k;
    if (c >= kMapCubes) break;
    const int ad = add[c], n = T[c].cnt + (ad > 0 ? ad : 0);
    roomy += compact_cap(n, 1);
    tight += compact_cap(n, 2);
  }
  if (roomy) { atomicAdd(&s_sum[1], roomy); atomicAdd(&s_sum[2], tight); }
  __syncthreads();
  const int mode = s_sum[1] <= a.pool_cap ? 1 : (s_sum[2] <= a.pool_cap ? 2 : 0);
  if (mode == 0) { if (tid == 0) *flag = 0; return; }                       // really full (k_map_reserve reports that)
  int local = 0;
  for (int k = 0; k < PER; ++k) { const int c = tid * PER + k; if (c < kMapCubes) { const int ad = add[c]; local += compact_cap(T[c].cnt + (ad > 0 ? ad : 0), mode); } }
  s_part[tid] = local;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int x = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += x;
    __syncthreads();
  }
  int run = s_part[tid] - local;
  for (int k = 0; k < PER; ++k) { const int c = tid * PER + k; if (c < kMapCubes) { const int ad = add[c]; newoff[c] = run; run += compact_cap(T[c].cnt + (ad > 0 ? ad : 0), mode); } }
  if (tid == 0) { *flag = mode; ms.pool_used[cls] = s_part[255]; ms.compactions += 1; }
}

template <int PHASE>   // 0: cubes -> staging at their new offsets; 1: staging -> pool, descriptors updated
__global__ __launch_bounds__(256) void k_map_compact_move(MapArgs a, float4* staging) {
  const int b = blockIdx.y, cls = blockIdx.z, tid = threadIdx.x;
  const int mode = a.compact_flag[b * 2 + cls];
  if (!mode) return;
  CubeDesc* T = cube_table(a, b, cls);
  const int* add = a.addcnt + ((long long)b * 2 + cls) * kMapCubes;
  const int* newoff = a.cursor + ((long long)b * 2 + cls) * kMapCubes;
  float4* pool = a.pool[cls] + (long long)b * a.pool_cap;
  float4* stage = staging + ((long long)b * 2 + cls) * a.pool_cap;
  for (int c = blockIdx.x; c < kMapCubes; c += gridDim.x) {
    const CubeDesc d = T[c];
    const int no = newoff[c];
    if (PHASE == 0) { for (int i = tid; i < d.cnt; i += 256) stage[no + i] = pool[d.off + i]; }
    else {
      for (int i = tid; i < d.cnt; i += 256) pool[no + i] = stage[no + i];
      if (tid == 0) { const int ad = add[c]; CubeDesc nd = d; nd.off = no; nd.cap = compact_cap(d.cnt + (ad > 0 ? ad : 0), mode); T[c] = nd; }
    }
  }
}

// =======================================================================================================
// frame set-up
// =======================================================================================================
__global__ __launch_bounds__(256) void k_map_begin(MapArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  MapSeq& ms = a.seq[b];
  __shared__ int s_c[3], s_cen[3];
  if (tid == 0) {
    if (ms.err) ms.err_steps += 1;                                           // capacity flags describe one step; what the previous step
    ms.err = 0;                                                              // raised stays countable for a host that synchronises later
    if (b == 0) { if (a.vox_counters[1]) a.vox_counters[3] += 1; a.vox_counters[1] = 0; }
    const OdomState& od = a.odom[b];
    double qo[4], to[3], qm[4], q[4], rt[3];
    for (int k = 0; k < 4; ++k) { qo[k] = od.q_w[k]; qm[k] = ms.q_wmap_wodom[k]; ms.q_wodom[k] = qo[k]; }
    for (int k = 0; k < 3; ++k) { to[k] = od.t_w[k]; ms.t_wodom[k] = to[k]; }
    quat_mul(qm, qo, q);                                                   // transformAssociateToMap (:142-146)
    quat_rotate(qm, to[0], to[1], to[2], rt);
    for (int k = 0; k < 4; ++k) ms.par[k] = q[k];
    for (int k = 0; k < 3; ++k) ms.par[4 + k] = rt[k] + ms.t_wmap_wodom[k];
    for (int k = 0; k < 3; ++k) { s_cen[k] = ms.cen[k]; s_c[k] = cube_coord(ms.par[4 + k], ms.cen[k]); }
  }
  __syncthreads();
  // window shifts (:323-507): the descriptors move, the slab that falls off re-enters emptied at the other end
  const int dim[3] = {kMapW, kMapH, kMapD};
  for (int axis = 0; axis < 3; ++axis) {
    for (int guard = 0; guard < 64; ++guard) {
      const int c = s_c[axis], n = dim[axis];
      const int dir = c < 3 ? 1 : (c >= n - 3 ? -1 : 0);
      if (dir == 0) break;
      const int nu = dim[(axis + 1) % 3], nv = dim[(axis + 2) % 3];
      for (int line = tid; line < nu * nv * 2; line += 256) {
        const int cls = line / (nu * nv), u = (line % (nu * nv)) / nv, v = line % nv;
        CubeDesc* T = cube_table(a, b, cls);
        auto at = [&](int x) {
          int ijk[3];
          ijk[axis] = x; ijk[(axis + 1) % 3] = u; ijk[(axis + 2) % 3] = v;
          return ijk[0] + kMapW * ijk[1] + kMapW * kMapH * ijk[2];
        };
        if (dir > 0) {
          CubeDesc keep = T[at(n - 1)];
          for (int x = n - 1; x >= 1; --x) T[at(x)] = T[at(x - 1)];
          keep.cnt = 0;
          T[at(0)] = keep;
        } else {
          CubeDesc keep = T[at(0)];
          for (int x = 0; x < n - 1; ++x) T[at(x)] = T[at(x + 1)];
          keep.cnt = 0;
          T[at(n - 1)] = keep;
        }
      }
      __syncthreads();
      if (tid == 0) { s_c[axis] += dir; s_cen[axis] += dir; }
      __syncthreads();
    }
  }
  if (tid == 0) {
    int* tab = a.tab + (long long)b * kTabInts;
    const CubeDesc* Tc = cube_table(a, b, 0);
    const CubeDesc* Ts = cube_table(a, b, 1);
    int nvld = 0, pc = 0, ps = 0;
    for (int i = s_c[0] - 2; i <= s_c[0] + 2; i++)
      for (int j = s_c[1] - 2; j <= s_c[1] + 2; j++)
        for (int k = s_c[2] - 1; k <= s_c[2] + 1; k++)
          if (i >= 0 && i < kMapW && j >= 0 && j < kMapH && k >= 0 && k < kMapD) {
            const int ind = i + kMapW * j + kMapW * kMapH * k;
            tab[nvld] = ind;
            tab[80 + nvld] = pc;
            tab[160 + nvld] = ps;
            pc += Tc[ind].cnt;
            ps += Ts[ind].cnt;
            ++nvld;
          }
    tab[80 + nvld] = pc;
    tab[160 + nvld] = ps;
    ms.n_valid = nvld;
    ms.from_total[0] = pc;
    ms.from_total[1] = ps;
    ms.gate = (pc > 10 && ps > 50) ? 1 : 0;                                 // :554
    for (int k = 0; k < 3; ++k) { ms.cen[k] = s_cen[k]; ms.center[k] = s_c[k]; }
    for (int it = 0; it < 2; ++it) { ms.factor_num[it][0] = ms.factor_num[it][1] = 0; ms.lm_iterations[it] = 0; ms.lm_termination[it] = 4; }
  }
}

// =======================================================================================================
// pcl::VoxelGrid for any number of independent segments
// =======================================================================================================
// Which filter takes a segment: the single-workgroup LDS filter (k_vox_lds; list 2: <= kVoxTinyN points, one wave; list 0: <= kVoxSmallN
// points, 256 threads; list 1: up to kVoxBigN points, 1024 threads) or, for anything larger, the general tile-sort / rank-merge path
// through global memory.
__device__ __forceinline__ void vox_enlist(const VoxArgs& v, int seg, int n) {
  if (n <= 0) return;
  if (n <= kVoxTinyN) v.lists[2 * (long long)v.n_segs + atomicAdd(&v.counters[7], 1)] = seg;
  else if (n <= kVoxSmallN) v.lists[atomicAdd(&v.counters[5], 1)] = seg;
  else if (n <= kVoxBigN) v.lists[v.n_segs + atomicAdd(&v.counters[6], 1)] = seg;
  else atomicAdd(&v.counters[4], 1);
}

__global__ void k_map_stack_segments(MapArgs a, VoxArgs v) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.B * 2) return;
  const int b = g >> 1, cls = g & 1;
  VoxSeg s{};
  s.in = cls == 0 ? a.corner_last + (long long)b * a.R * 120 : a.surf_last + (long long)b * a.cap;
  s.n = cls == 0 ? a.meta[b].n_corner_last : a.meta[b].n_surf_last;
  s.out = a.stack[cls] + (long long)b * (cls == 0 ? a.R * 120 : a.cap);
  s.out_count = &a.seq[b].n_stack[cls];
  s.final_out = nullptr;
  s.final_count = nullptr;
  s.leaf = cls == 0 ? a.line_res : a.plane_res;                             // downSizeFilterCorner / Surf (:904-905)
  v.segs[g] = s;
  vox_enlist(v, g, s.n);
  if (s.n == 0) *s.out_count = 0;
}

__global__ void k_map_cube_segments(MapArgs a, VoxArgs v) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.B * 2 * kMapValidMax) return;
  const int j = g % kMapValidMax, cls = (g / kMapValidMax) & 1, b = g / (2 * kMapValidMax);
  VoxSeg s{};
  s.leaf = cls == 0 ? a.line_res : a.plane_res;
  if (j < a.seq[b].n_valid) {
    CubeDesc* d = cube_table(a, b, cls) + a.tab[(long long)b * kTabInts + j];
    s.in = a.pool[cls] + (long long)b * a.pool_cap + d->off;
    s.n = d->cnt;
    s.final_out = a.pool[cls] + (long long)b * a.pool_cap + d->off;
    s.final_count = &d->cnt;
    s.out = v.tmp + ((long long)b * 2 + cls) * a.pool_cap + d->off;         // staging of the in-place filter: the cube's own range of a pool-sized scratch
  }
  v.segs[g] = s;
  vox_enlist(v, g, s.n);
}

// key offsets, tile work list, bounding-box reset.  One 1024-thread workgroup.
__global__ __launch_bounds__(1024) void k_vox_setup(VoxArgs v) {
  const int tid = threadIdx.x;
  if (v.counters[4] == 0) { if (tid == 0) { v.counters[0] = 0; v.counters[2] = 0; } return; }   // the LDS filter took every segment
  __shared__ long long s_keys[1024];
  __shared__ int s_tiles[1024];
  const int per = (v.n_segs + 1023) / 1024;
  const int s0 = tid * per, s1 = min(v.n_segs, s0 + per);
  long long kk = 0;
  int tt = 0;
  for (int s = s0; s < s1; ++s) { const int n = v.segs[s].n; kk += n; tt += (n + kVoxTile - 1) / kVoxTile; }
  s_keys[tid] = kk;
  s_tiles[tid] = tt;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const long long ak = tid >= d ? s_keys[tid - d] : 0;
    const int at = tid >= d ? s_tiles[tid - d] : 0;
    __syncthreads();
    s_keys[tid] += ak;
    s_tiles[tid] += at;
    __syncthreads();
  }
  const long long total_keys = s_keys[1023];
  const int total_tiles = s_tiles[1023];
  const bool bad = total_keys > v.key_cap || total_tiles > v.tile_cap;
  long long ko = s_keys[tid] - kk;
  int to = s_tiles[tid] - tt;
  for (int s = s0; s < s1; ++s) {
    VoxSeg& sg = v.segs[s];
    if (bad || sg.n > (kVoxTile << v.levels)) {                              // skipped, reported through counters[1]
      if (!bad) atomicOr(&v.counters[1], kMapErrSegment);
      sg.n = 0; sg.final_out = nullptr; sg.final_count = nullptr;
    }
    const int n = sg.n, nt = (n + kVoxTile - 1) / kVoxTile;
    sg.key_off = (int)ko;
    sg.tile0 = to;
    sg.ntiles = nt;
    if (sg.final_out) sg.out = v.tmp + ko;
    if (!bad) for (int t = 0; t < nt; ++t) v.tile_seg[to + t] = s;
    for (int q = 0; q < 3; ++q) { v.bbox[s * 6 + q] = 0x7fffffff; v.bbox[s * 6 + 3 + q] = (int)0x80000000; }
    ko += n;
    to += nt;
  }
  // the largest number of merge levels any segment of this call needs: the merge launches behind it return at once
  __shared__ int s_maxt;
  if (tid == 0) s_maxt = 0;
  __syncthreads();
  int mt = 0;
  for (int s = s0; s < s1; ++s) mt = max(mt, v.segs[s].ntiles);
  if (mt > 0) atomicMax(&s_maxt, mt);
  __syncthreads();
  if (tid == 0) {
    int need = 0;
    while ((1 << need) < s_maxt) ++need;
    v.counters[2] = need;
    v.counters[0] = bad ? 0 : total_tiles;
    if (bad) atomicOr(&v.counters[1], kMapErrKeys);
  }
}

__global__ __launch_bounds__(256) void k_vox_bbox(VoxArgs v) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int gt = blockIdx.x; gt < v.counters[0]; gt += gridDim.x) {   // grid-stride over the tile work list
  const int s = v.tile_seg[gt];
  const VoxSeg sg = v.segs[s];
  const int t = gt - sg.tile0;
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int e = tid; e < kVoxTile; e += 256) {
    const int i = t * kVoxTile + e;
    if (i < sg.n) {
      const float4 p = sg.in[i];
      mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
      mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
      mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    for (int d = 32; d > 0; d >>= 1) { mn[q] = fminf(mn[q], __shfl_down(mn[q], d, 64)); mx[q] = fmaxf(mx[q], __shfl_down(mx[q], d, 64)); }
    if (lane == 0) { atomicMin(&v.bbox[s * 6 + q], f2o(mn[q])); atomicMax(&v.bbox[s * 6 + 3 + q], f2o(mx[q])); }
  }
  }
}

// (voxel index, point index) keys of one tile, sorted in LDS.
__global__ __launch_bounds__(256) void k_vox_keys_sort(VoxArgs v) {
  const int tid = threadIdx.x;
  for (int gt = blockIdx.x; gt < v.counters[0]; gt += gridDim.x) {   // grid-stride over the tile work list
  const int s = v.tile_seg[gt];
  const VoxSeg sg = v.segs[s];
  const int t = gt - sg.tile0;
  __shared__ unsigned long long keys[kVoxTile];
  const float inv = 1.0f / sg.leaf;
  float gmn[3], gmx[3], fminb[3];
  int divb[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { gmn[q] = o2f(v.bbox[s * 6 + q]); gmx[q] = o2f(v.bbox[s * 6 + 3 + q]); }
  const long long dx = (long long)((gmx[0] - gmn[0]) * inv) + 1, dy = (long long)((gmx[1] - gmn[1]) * inv) + 1, dz = (long long)((gmx[2] - gmn[2]) * inv) + 1;
  const bool overflow = dx * dy * dz > 2147483647ll;                      // PCL then returns the input unfiltered
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int minb = (int)floorf(gmn[q] * inv);
    divb[q] = (int)floorf(gmx[q] * inv) - minb + 1;
    fminb[q] = (float)minb;
  }
  for (int e = tid; e < kVoxTile; e += 256) {
    const int i = t * kVoxTile + e;
    unsigned long long key = ~0ull;
    if (i < sg.n) {
      unsigned vi;
      if (overflow) vi = (unsigned)i;
      else {
        const float4 p = sg.in[i];
        const int i0 = (int)(floorf(p.x * inv) - fminb[0]);
        const int i1 = (int)(floorf(p.y * inv) - fminb[1]);
        const int i2 = (int)(floorf(p.z * inv) - fminb[2]);
        vi = (unsigned)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]);
      }
      key = ((unsigned long long)vi << 32) | (unsigned)i;
    }
    keys[e] = key;
  }
  __syncthreads();
  const int in_tile = min(kVoxTile, sg.n - t * kVoxTile);
  bitonic_sort_u64(keys, in_tile, tid);
  unsigned long long* dst = v.keys[0] + sg.key_off;
  for (int e = tid; e < kVoxTile; e += 256) { const int i = t * kVoxTile + e; if (i < sg.n) dst[i] = keys[e]; }
  __syncthreads();
  }
}

// One rank-merge level: runs of `run` keys are merged pairwise; every key finds its place by a binary search in the sibling
// run (keys are unique, so no tie rule is needed).  src = keys[level & 1], dst = the other buffer.
__global__ __launch_bounds__(256) void k_vox_merge(VoxArgs v, int level) {
  const int tid = threadIdx.x;
  if (level > v.counters[2]) return;                                       // no segment of this call is that large (k_vox_setup)
  for (int gt = blockIdx.x; gt < v.counters[0]; gt += gridDim.x) {   // grid-stride over the tile work list
  const VoxSeg sg = v.segs[v.tile_seg[gt]];
  const int t = gt - sg.tile0;
  // a segment of ntiles tiles is sorted after ceil(log2(ntiles)) levels; after that it only has to end up in the buffer the
  // later stages read (keys[levels & 1]): at most one copy, then nothing
  int need = 0;
  while ((1 << need) < sg.ntiles) ++need;
  if (level > need || (level == need && ((v.levels - need) & 1) == 0)) continue;
  const unsigned long long* src = v.keys[level & 1] + sg.key_off;
  unsigned long long* dst = v.keys[(level & 1) ^ 1] + sg.key_off;
  const int run = kVoxTile << level;
  constexpr int PER = kVoxTile / 256;
  // all keys of a tile belong to the same run, so they search the same sibling range with the same number of steps: the
  // PER searches of a thread are advanced in lock-step, PER independent loads in flight per step
  const int p_first = t * kVoxTile;
  const int r = p_first / run, sb = (r ^ 1) * run;
  const bool has_sib = sb < sg.n;
  const int s_end = has_sib ? min(sg.n, sb + run) : sb;
  unsigned long long key[PER];
  int lo[PER], hi[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int p = p_first + tid + u * 256;
    key[u] = p < sg.n ? src[p] : ~0ull;
    lo[u] = sb; hi[u] = s_end;
  }
  if (has_sib) {
    for (int len = s_end - sb; len > 0; len >>= 1) {                    // ceil(log2(len + 1)) steps
      unsigned long long probe[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) { const int mid = (lo[u] + hi[u]) >> 1; probe[u] = lo[u] < hi[u] ? src[mid] : 0ull; }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        if (lo[u] < hi[u]) { const int mid = (lo[u] + hi[u]) >> 1; if (probe[u] < key[u]) lo[u] = mid + 1; else hi[u] = mid; }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int p = p_first + tid + u * 256;
    if (p >= sg.n) continue;
    const int dest = has_sib ? (r >> 1) * 2 * run + (p - r * run) + (lo[u] - sb) : p;
    dst[dest] = key[u];
  }
  }
}

__global__ __launch_bounds__(256) void k_vox_heads(VoxArgs v) {
  const int tid = threadIdx.x;
  for (int gt = blockIdx.x; gt < v.counters[0]; gt += gridDim.x) {   // grid-stride over the tile work list
  const VoxSeg sg = v.segs[v.tile_seg[gt]];
  const int t = gt - sg.tile0;
  const unsigned long long* K = v.keys[v.levels & 1] + sg.key_off;
  int heads = 0;
  for (int e = tid; e < kVoxTile; e += 256) {
    const int p = t * kVoxTile + e;
    if (p < sg.n && (p == 0 || (unsigned)(K[p - 1] >> 32) != (unsigned)(K[p] >> 32))) ++heads;
  }
  __shared__ int s_sum;
  if (tid == 0) s_sum = 0;
  __syncthreads();
  for (int d = 32; d > 0; d >>= 1) heads += __shfl_down(heads, d, 64);
  if ((tid & 63) == 0) atomicAdd(&s_sum, heads);
  __syncthreads();
  if (tid == 0) v.tile_heads[gt] = s_sum;
  __syncthreads();
  }
}

// exclusive prefix of the head counts over all tiles (one 1024-thread workgroup), per-segment output counts
__global__ __launch_bounds__(1024) void k_vox_scan(VoxArgs v) {
  const int tid = threadIdx.x;
  if (v.counters[4] == 0) return;
  const int n = v.counters[0];
  __shared__ int s_part[1024];
  const int per = (n + 1023) / 1024;
  const int t0 = tid * per, t1 = min(n, t0 + per);
  int local = 0;
  for (int t = t0; t < t1; ++t) local += v.tile_heads[t];
  s_part[tid] = local;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int x = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += x;
    __syncthreads();
  }
  int run = s_part[tid] - local;
  for (int t = t0; t < t1; ++t) { v.tile_pref[t] = run; run += v.tile_heads[t]; }
  if (tid == 1023 || t1 == n) v.tile_pref[n] = s_part[1023];
  __threadfence();
  __syncthreads();
  for (int s = tid; s < v.n_segs; s += 1024) {
    const VoxSeg& sg = v.segs[s];
    int m = 0;
    if (sg.ntiles > 0) {
      int hi = 0;                                                           // prefix at the end of the segment's last tile
      const int last = sg.tile0 + sg.ntiles;
      // tile_pref of other threads' ranges is visible after the barrier (same workgroup, global memory)
      hi = (last >= n) ? s_part[1023] : ((volatile int*)v.tile_pref)[last];
      m = hi - ((volatile int*)v.tile_pref)[sg.tile0];
    }
    if (sg.out_count && !sg.final_count) *sg.out_count = m;
    if (sg.final_count) *sg.final_count = m;
  }
}

// centroids: the head of every voxel run sums its members in key order (= input order) and writes output number `rank`
__global__ __launch_bounds__(256) void k_vox_emit(VoxArgs v) {
  const int tid = threadIdx.x;
  for (int gt = blockIdx.x; gt < v.counters[0]; gt += gridDim.x) {   // grid-stride over the tile work list
  const VoxSeg sg = v.segs[v.tile_seg[gt]];
  const int t = gt - sg.tile0;
  const unsigned long long* K = v.keys[v.levels & 1] + sg.key_off;
  constexpr int PER = kVoxTile / 256;
  __shared__ int s_scan[256];
  __shared__ float4 s_pts[kVoxTile];                                         // the tile's points in key order
  __shared__ unsigned s_vi[kVoxTile];                                         // and their voxel indices
  // stage the tile: every thread fetches its own members (independent loads); the serial per-voxel sums then run out of LDS
  for (int e = tid; e < kVoxTile; e += 256) {
    const int p = t * kVoxTile + e;
    if (p < sg.n) { const unsigned long long k = K[p]; s_vi[e] = (unsigned)(k >> 32); s_pts[e] = sg.in[(unsigned)k]; }
  }
  __syncthreads();
  const int e0 = tid * PER, p0 = t * kVoxTile + e0;
  const int in_tile = min(kVoxTile, sg.n - t * kVoxTile);
  int heads = 0;
  unsigned flags = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int q = e0 + e;
    if (q < in_tile && (q == 0 ? (p0 + e == 0 || (unsigned)(K[p0 + e - 1] >> 32) != s_vi[q]) : s_vi[q - 1] != s_vi[q])) { ++heads; flags |= 1u << e; }
  }
  s_scan[tid] = heads;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int x = tid >= d ? s_scan[tid - d] : 0;
    __syncthreads();
    s_scan[tid] += x;
    __syncthreads();
  }
  int rank = v.tile_pref[gt] - v.tile_pref[sg.tile0] + s_scan[tid] - heads;
#pragma unroll 1
  for (int e = 0; e < PER; ++e) {
    if (!(flags & (1u << e))) continue;
    const int q0 = e0 + e;
    const unsigned vi = s_vi[q0];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    int cnt = 0;
    int q = q0;
    for (; q < in_tile && s_vi[q] == vi; ++q) { const float4 pt = s_pts[q]; sx += pt.x; sy += pt.y; sz += pt.z; si += pt.w; ++cnt; }
    if (q == in_tile) {                                                       // the voxel continues in the next tile(s): finish from global memory
      for (int p = t * kVoxTile + q; p < sg.n; ++p) {
        const unsigned long long kq = K[p];
        if ((unsigned)(kq >> 32) != vi) break;
        const float4 pt = sg.in[(unsigned)kq];
        sx += pt.x; sy += pt.y; sz += pt.z; si += pt.w;
        ++cnt;
      }
    }
    const float fc = (float)cnt;
    sg.out[rank++] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
  }
  __syncthreads();
  }
}

// in-place re-filter of a map cube: copy the centroids from the scratch area back over the cube's segment
__global__ __launch_bounds__(256) void k_vox_copyback(VoxArgs v) {
  const int tid = threadIdx.x;
  for (int gt = blockIdx.x; gt < v.counters[0]; gt += gridDim.x) {   // grid-stride over the tile work list
  const VoxSeg sg = v.segs[v.tile_seg[gt]];
  if (!sg.final_out) continue;
  const int t = gt - sg.tile0;
  const int m = v.tile_pref[min(sg.tile0 + sg.ntiles, v.counters[0])] - v.tile_pref[sg.tile0];
  for (int e = tid; e < kVoxTile; e += 256) { const int i = t * kVoxTile + e; if (i < m) sg.final_out[i] = sg.out[i]; }
  }
}

// =======================================================================================================
// pcl::VoxelGrid of ONE segment by ONE workgroup, sorted in LDS
// =======================================================================================================
// The general path above sorts (voxel, point) keys through global memory: tile sort + up to log2(tiles) rank-merge launches +
// heads + scan + emit.  The clouds the mapping stage filters are small enough for one workgroup each — the incoming less-sharp /
// less-flat clouds (6 k / 40 k points) and the map cubes (a few hundred to a few thousand points) — so this kernel does the whole
// filter of a segment in one launch:
//   pass 1  bounding box (f32 min / max, as PCL computes it) and RUN heads: a point starts a run when its 0.8 / 0.4 m cell differs
//           from its predecessor's.  The cell test uses floor(p * inv) itself, which does not depend on the box.  The incoming
//           clouds are ring-ordered, so consecutive points often share a cell: 40 k less-flat points are ~18-23 k runs.
//   pass 2  one key per run: (voxel index [32 bits] , first point [16 bits]) in two LDS arrays, in input order
//   sort    stable radix sort on the voxel index (radix_sort_pairs) -> runs of one voxel adjacent and in ascending point order
//   heads   first run of every voxel -> output rank = ascending voxel index
//   sums    the head of a voxel walks its runs: members are added in input order, exactly what the general path does (the
//           "canonical order" of DESIGN.md section 5); centroid = sums / count
// A segment whose runs do not fit (kVox*Runs), whose coordinates exceed the range where floor(p * inv) is an exact f32 integer
// below 2^23, or that is larger than the list limits falls through to the general path untouched (counters[4]).
// Measured per 40 k-point segment (1024 threads, device timers of an instrumented round-3 build): pass 1 36 us, pass 2 30, sort 70 (a
// bitonic network on the same pairs: 385), heads 3, sums 190 before / ~70 after the batched loads.
// Stable LSD radix sort of n (voxel index, first point) pairs by the voxel index, 7 bits per pass.  The runs were generated in
// input order, so a STABLE sort on the voxel index alone leaves the runs of one voxel in ascending point order — the order the
// sums need.  Every thread keeps its CAPR / NT elements in registers (wave w owns the contiguous stretch [w * EPT * 64, ...),
// element = row k, lane l, so rows are 64 consecutive elements); the two LDS arrays are only the scatter target of a pass and
// are read back in the same striped layout for the next one: one buffer instead of a ping-pong pair.  Per pass: per-(digit, wave)
// counts from wave-level digit matching (7 ballots), one exclusive scan in digit-major order, scatter with the rank among the
// equal digits of the row.  A bitonic network on the same keys took 385 us for 23 k runs (120 barrier-separated LDS stages);
// three radix passes take a fraction of that.
template <int NT, int CAPR>
__device__ __forceinline__ void radix_sort_pairs(unsigned* hi, unsigned short* lo, unsigned short* cntw, int* s_w, int n, int key_bits, int tid) {
  constexpr int NW = NT / 64, EPT = CAPR / NT, RB = 7, NB = 1 << RB, PER = NB * NW / NT;
  static_assert(NB * NW % NT == 0 && PER == 2, "two counters per thread in the scan");
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int wbase = wave * EPT * 64;
  const int rows = __builtin_amdgcn_readfirstlane(n > wbase ? min(EPT, (n - wbase + 63) >> 6) : 0);   // rows of this wave that hold anything (wave-uniform: scalar branches)
  unsigned rv[EPT], rf[EPT];
  auto load = [&]() {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int p = wbase + k * 64 + lane;
      const bool valid = k < rows && p < n;
      rv[k] = valid ? hi[p] : 0xffffffffu;                                   // the tail of the last row sorts behind everything
      rf[k] = valid ? (unsigned)lo[p] : 0u;
    }
  };
  auto match = [&](unsigned d) {                                             // lanes of this wave holding the same digit
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit < RB; ++bit) { const bool one = (d >> bit) & 1u; const unsigned long long bal = __ballot(one); m &= one ? bal : ~bal; }
    return m;
  };
  load();
  __syncthreads();                                                           // every element sits in a register: the LDS arrays are free
  for (int shift = 0; shift < key_bits; shift += RB) {
    for (int c = tid; c < NB * NW; c += NT) cntw[c] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (k < rows) {
        const unsigned d = (rv[k] >> shift) & (NB - 1);
        const unsigned long long m = match(d);
        if ((m & lt) == 0ull) cntw[d * NW + wave] = (unsigned short)(cntw[d * NW + wave] + __popcll(m));   // the lowest lane of every group
      }
      __builtin_amdgcn_sched_barrier(0);                                     // one row at a time: interleaving the rows costs hundreds of registers
    }
    __syncthreads();
    {                                                                        // exclusive scan of the NB * NW counts, digit-major
      const int v0 = cntw[tid * 2], v1 = cntw[tid * 2 + 1], sum = v0 + v1;
      const int inc = wave_scan_i32<false>(sum);
      if (lane == 63) s_w[wave] = inc;
      __syncthreads();
      int off = 0;
#pragma unroll 1
      for (int w = 0; w < wave; ++w) off += s_w[w];
      const int base = off + inc - sum;
      cntw[tid * 2] = (unsigned short)base;
      cntw[tid * 2 + 1] = (unsigned short)(base + v0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (k < rows) {
        unsigned key = rv[k];
        asm volatile("" : "+v"(key));                                        // (keeps the compiler from carrying the histogram loop's 14 bit tests per row over in registers)
        const unsigned d = (key >> shift) & (NB - 1);
        const unsigned long long m = match(d);
        const int base = cntw[d * NW + wave], pos = base + __popcll(m & lt);
        hi[pos] = rv[k];
        lo[pos] = (unsigned short)rf[k];
        if ((m & lt) == 0ull) cntw[d * NW + wave] = (unsigned short)(base + __popcll(m));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (shift + RB < key_bits) { load(); __syncthreads(); }
  }
}

template <int NT, int CAPR, int CAPN>
__global__ __launch_bounds__(NT) void k_vox_lds(VoxArgs v, int which) {
  constexpr int NW = NT / 64, ITS = CAPR / NT, U = 4;
  static_assert(CAPR % NT == 0 && CAPN % 64 == 0 && CAPN <= 65536 && NW <= 16, "geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char vl_smem[];
  unsigned* khi = reinterpret_cast<unsigned*>(vl_smem);                      // [CAPR] voxel index of a run
  unsigned short* klo = reinterpret_cast<unsigned short*>(khi + CAPR);       // [CAPR] first point of the run
  unsigned* cont = reinterpret_cast<unsigned*>(klo + CAPR);                  // [CAPN / 32] bit i: point i continues the run of i - 1
  unsigned short* cntw = reinterpret_cast<unsigned short*>(cont + CAPN / 32); // [128 * NW] radix counters of the sort
  int* s_tab = reinterpret_cast<int*>(cntw + 128 * NW);                      // [ITS * NW + 1] voxel heads per (round, wave) -> offsets
  int* s_i = s_tab + ITS * NW + 1;                                           // [48] per-wave run counts / flags
  float* s_f = reinterpret_cast<float*>(s_i + 48);                           // [6][NW] bounding-box partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int* list = v.lists + (long long)which * v.n_segs;
  const int count = v.counters[5 + which];
  for (int li = blockIdx.x; li < count; li += gridDim.x) {
    __syncthreads();                                                         // LDS is reused from segment to segment
    const int s = list[li];
    const VoxSeg sg = v.segs[s];
    const int n = sg.n;
    const float4* in = sg.in;
    const float inv = 1.0f / sg.leaf;
    const int chunk = ((n + NW - 1) / NW + 63) & ~63;                        // every wave owns a contiguous, 64-aligned stretch of the segment
    const int w0 = wave * chunk, w1 = min(n, w0 + chunk);
    // ---- pass 1: bounding box, run heads ------------------------------------------------------------------------------------
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    int heads = 0;
    bool big = false;
    float cfx = 0.f, cfy = 0.f, cfz = 0.f;                                   // cell of the point in front of this row (lane 0's predecessor)
    if (w0 < w1 && w0 > 0) { const float4 p = in[w0 - 1]; cfx = floorf(p.x * inv); cfy = floorf(p.y * inv); cfz = floorf(p.z * inv); }
    for (int base = w0; base < w1; base += 64 * U) {
      float4 p[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = base + u * 64 + lane; p[u] = in[i < w1 ? i : w1 - 1]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rb = base + u * 64, i = rb + lane;
        if (rb >= w1) break;                                                 // uniform
        const bool act = i < w1;
        const float fx = floorf(p[u].x * inv), fy = floorf(p[u].y * inv), fz = floorf(p[u].z * inv);
        if (act) {
          mn[0] = fminf(mn[0], p[u].x); mx[0] = fmaxf(mx[0], p[u].x);
          mn[1] = fminf(mn[1], p[u].y); mx[1] = fmaxf(mx[1], p[u].y);
          mn[2] = fminf(mn[2], p[u].z); mx[2] = fmaxf(mx[2], p[u].z);
          if (!(fabsf(fx) < 8388608.f && fabsf(fy) < 8388608.f && fabsf(fz) < 8388608.f)) big = true;
        }
        // predecessor's cell: wave_shr:1 on the DPP network, lane 0 keeps `old` = the carried cell
        const float px = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cfx), __float_as_int(fx), 0x138, 0xF, 0xF, false));
        const float py = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cfy), __float_as_int(fy), 0x138, 0xF, 0xF, false));
        const float pz = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cfz), __float_as_int(fz), 0x138, 0xF, 0xF, false));
        const bool head = act && (i == 0 || fx != px || fy != py || fz != pz);
        const unsigned long long hm = __ballot(head), am = __ballot(act), cm = am & ~hm;
        heads += __popcll(hm);
        if (lane == 0) cont[rb >> 5] = (unsigned)cm;
        if (lane == 1) cont[(rb >> 5) + 1] = (unsigned)(cm >> 32);
        cfx = __shfl(fx, 63, 64); cfy = __shfl(fy, 63, 64); cfz = __shfl(fz, 63, 64);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      for (int d = 32; d > 0; d >>= 1) { mn[q] = fminf(mn[q], __shfl_down(mn[q], d, 64)); mx[q] = fmaxf(mx[q], __shfl_down(mx[q], d, 64)); }
      if (lane == 0) { s_f[q * NW + wave] = mn[q]; s_f[(3 + q) * NW + wave] = mx[q]; }
    }
    const bool anybig = __ballot(big) != 0ull;
    if (lane == 0) { s_i[wave] = heads; s_i[16 + wave] = anybig ? 1 : 0; }
    __syncthreads();
    float gmn[3], gmx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      gmn[q] = s_f[q * NW]; gmx[q] = s_f[(3 + q) * NW];
#pragma unroll 1
      for (int w = 1; w < NW; ++w) { gmn[q] = fminf(gmn[q], s_f[q * NW + w]); gmx[q] = fmaxf(gmx[q], s_f[(3 + q) * NW + w]); }
    }
    int n_runs = 0, rank = 0;
    bool unfit = false;
#pragma unroll 1
    for (int w = 0; w < NW; ++w) { if (w == wave) rank = n_runs; n_runs += s_i[w]; unfit = unfit || s_i[16 + w] != 0; }
    unfit = unfit || n_runs > CAPR || n > CAPN;
    if (unfit) {                                                             // general path (uniform decision; the segment stays as it is)
      if (tid == 0) atomicAdd(&v.counters[4], 1);
      continue;
    }
    float4* const stage = sg.out;                                            // final place (stacks) or the staging range (in-place cube filter)
    const long long dx = (long long)((gmx[0] - gmn[0]) * inv) + 1, dy = (long long)((gmx[1] - gmn[1]) * inv) + 1, dz = (long long)((gmx[2] - gmn[2]) * inv) + 1;
    int n_vox = 0;
    if (dx * dy * dz > 2147483647ll) {                                       // PCL's overflow guard: the input comes back unfiltered
      n_vox = n;
      if (!sg.final_out) for (int i = tid; i < n; i += NT) stage[i] = in[i];
    } else {
      int divb[3];
      float fminb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int minb = (int)floorf(gmn[q] * inv);
        divb[q] = (int)floorf(gmx[q] * inv) - minb + 1;
        fminb[q] = (float)minb;
      }
      // ---- pass 2: one (voxel index, first point) key per run, in input order ------------------------------------------------
      for (int base = w0; base < w1; base += 64 * U) {
        float4 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + u * 64 + lane; p[u] = in[i < w1 ? i : w1 - 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rb = base + u * 64, i = rb + lane;
          if (rb >= w1) break;
          const bool head = i < w1 && !((cont[i >> 5] >> (i & 31)) & 1u);
          const unsigned long long hm = __ballot(head);
          if (head) {
            const int i0 = (int)(floorf(p[u].x * inv) - fminb[0]), i1 = (int)(floorf(p[u].y * inv) - fminb[1]), i2 = (int)(floorf(p[u].z * inv) - fminb[2]);
            const int r = rank + __popcll(hm & lt);
            khi[r] = (unsigned)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]);
            klo[r] = (unsigned short)i;
          }
          rank += __popcll(hm);
        }
      }
      __syncthreads();
      {
        const long long cells = (long long)divb[0] * divb[1] * divb[2];      // every voxel index is below this (<= 2^31 - 1)
        const int key_bits = cells > 1 ? 64 - __clzll(cells - 1) : 1;
        radix_sort_pairs<NT, CAPR>(khi, klo, cntw, s_i + 32, n_runs, key_bits, tid);
      }
      // ---- voxel heads among the sorted runs -> output rank ---------------------------------------------------------------------
      const int rounds = (n_runs + NT - 1) / NT;                             // rounds of NT sorted runs that hold any
      for (int it = 0; it < rounds; ++it) {
        const int p = it * NT + tid;
        const bool h = p < n_runs && (p == 0 || khi[p - 1] != khi[p]);
        const unsigned long long m = __ballot(h);
        if (lane == 0) s_tab[it * NW + wave] = __popcll(m);
      }
      for (int e = rounds * NW + tid; e < ITS * NW; e += NT) s_tab[e] = 0;
      __syncthreads();
      if (wave == 0) {                                                       // exclusive scan of the ITS * NW counts by one wave
        constexpr int PER = (ITS * NW + 63) / 64;
        int loc[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int e = lane * PER + k; loc[k] = e < ITS * NW ? s_tab[e] : 0; sum += loc[k]; }
        const int inc = wave_scan_i32<false>(sum);
        int run = inc - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int e = lane * PER + k; if (e < ITS * NW) s_tab[e] = run; run += loc[k]; }
        if (lane == 63) s_tab[ITS * NW] = inc;
      }
      __syncthreads();
      n_vox = s_tab[ITS * NW];
      // ---- centroids: the members of a voxel in input order (runs ascending, points of a run consecutive) ----------------------
#pragma unroll 1
      for (int it = 0; it < rounds; ++it) {
        const int p = it * NT + tid;
        const bool h = p < n_runs && (p == 0 || khi[p - 1] != khi[p]);
        const int vrank = __popcll(__ballot(h) & lt);                        // (the ballot again instead of 20 registers carried across the scan)
        if (!h) continue;
        const unsigned vi = khi[p];
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        int cnt = 0;
        // the members in order: runs q = p, p + 1, ... of this voxel, the points of a run while the continuation bit is set.  Up to
        // eight member indices are collected from LDS first, their loads issued together, then added in order: one memory round trip
        // per eight members instead of one per run.
        int q = p, e = klo[p];
        bool have = true;
        while (have) {
          int idx[8], m = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            idx[k] = e;
            if (have) {
              ++m;
              if (e + 1 < n && ((cont[(e + 1) >> 5] >> ((e + 1) & 31)) & 1u)) ++e;
              else { ++q; if (q < n_runs && khi[q] == vi) e = klo[q]; else have = false; }
            }
          }
          float4 pt[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) pt[k] = in[idx[k]];
#pragma unroll
          for (int k = 0; k < 8; ++k) if (k < m) { sx += pt[k].x; sy += pt[k].y; sz += pt[k].z; si += pt[k].w; }
          cnt += m;
        }
        const float fc = (float)cnt;
        stage[s_tab[it * NW + wave] + vrank] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
      }
    }
    if (sg.final_out) {                                                      // in-place filter: back over the cube once every member has been read
      __syncthreads();
      if (stage != sg.final_out && !(dx * dy * dz > 2147483647ll)) for (int i = tid; i < n_vox; i += NT) sg.final_out[i] = stage[i];
    }
    if (tid == 0) {
      if (sg.final_count) *sg.final_count = n_vox; else if (sg.out_count) *sg.out_count = n_vox;
      VoxSeg& g = v.segs[s];                                                 // consumed: the general path sees an empty segment
      g.n = 0; g.out_count = nullptr; g.final_out = nullptr; g.final_count = nullptr;
    }
  }
}

// =======================================================================================================
// kd-tree stand-in over the submap: 2 m cell hash (map_bucket)
// =======================================================================================================
// Built by ONE 1024-thread workgroup per (sequence, class) with an LDS counting sort (count -> scan -> fill), the way k_build_grids
// builds the odometry grids: no global atomics, no zero-fill of a count table, one launch.  32-bit counters, so the submap may be of
// any size; the bucket table is capped at 32768 entries (132 KiB of the CU's 160 KiB) however far the pool has grown.
__global__ __launch_bounds__(1024) void k_mapgrid_build(MapArgs a) {
  const int b = blockIdx.x, cls = blockIdx.y, tid = threadIdx.x;
  const MapSeq& ms = a.seq[b];
  const int n = ms.from_total[cls], nv = ms.n_valid, H = a.grid_H;
  const int* tab = a.tab + (long long)b * kTabInts;
  extern __shared__ __attribute__((aligned(16))) int mg_lds[];
  int* cnt = mg_lds;                       // [H]
  int* part = cnt + H;                     // [1024]
  int* s_pref = part + 1024;               // [80] start of every valid cube in the concatenated submap
  int* s_off = s_pref + 80;                // [80] its offset in the class pool
  int* start = a.grid_start[cls] + (long long)b * (H + 1);
  float4* sorted = a.grid_sorted[cls] + (long long)b * a.pool_cap;
  const float4* pool = a.pool[cls] + (long long)b * a.pool_cap;
  for (int h = tid; h < H; h += 1024) cnt[h] = 0;
  if (tid <= nv) s_pref[tid] = tab[80 + cls * 80 + tid];
  if (tid < nv) s_off[tid] = cube_table(a, b, cls)[tab[tid]].off;
  __syncthreads();
  constexpr int U = 4;
  auto fetch = [&](int base, float4* p) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int g = base + u * 1024 + tid;
      g = g < n ? g : n - 1;
      int lo = 0, hi = nv - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= g) lo = mid; else hi = mid - 1; }
      p[u] = pool[s_off[lo] + (g - s_pref[lo])];
    }
  };
  auto bucket = [&](const float4& p) { return (int)map_bucket((int)floorf(p.x * kMapCellInv), (int)floorf(p.y * kMapCellInv), (int)floorf(p.z * kMapCellInv), H); };
  for (int base = 0; base < n; base += U * 1024) {
    float4 p[U];
    fetch(base, p);
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 1024 + tid < n) atomicAdd(&cnt[bucket(p[u])], 1);
  }
  __syncthreads();
  const int per = H / 1024;
  int local = 0;
  for (int k = 0; k < per; ++k) local += cnt[tid * per + k];
  part[tid] = local;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int x = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += x;
    __syncthreads();
  }
  int run = part[tid] - local;
  for (int k = 0; k < per; ++k) { const int c = cnt[tid * per + k]; cnt[tid * per + k] = run; start[tid * per + k] = run; run += c; }
  if (tid == 1023) start[H] = run;
  __syncthreads();
  for (int base = 0; base < n; base += U * 1024) {
    float4 p[U];
    fetch(base, p);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = base + u * 1024 + tid;
      if (g < n) { const int pos = atomicAdd(&cnt[bucket(p[u])], 1); sorted[pos] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(g)); }
    }
  }
}

// =======================================================================================================
// data association: exact 5-NN within 1 m, line / plane fit, factor records
// =======================================================================================================
namespace {

struct __attribute__((packed, aligned(4))) MapIntPair { int a, b; };      // start[h], start[h + 1] by one 8-byte load
typedef float mfloat2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dist_to_map(const float4& p, mfloat2 sxy, float sz) {
  const mfloat2 pxy = {p.x, p.y};
  const mfloat2 dxy = pxy - sxy, qxy = dxy * dxy;                             // packed f32: the same two subtractions and products
  const float ddz = p.z - sz;
  return (qxy.x + qxy.y) + ddz * ddz;                                         // FLANN L2_Simple, f32: (dx^2 + dy^2) + dz^2
}

// The five nearest as packed keys (f32 distance bits << 32 | submap index: one 64-bit compare orders by (distance, index)) + the position
// of the entry in the bucketed copy; coordinates are fetched through the position at the end: 15 registers instead of the 25 five
// neighbours with coordinates need, which is what takes k_map_search from 72 - 74 to 63 / 65 registers.
struct Top5P {
  unsigned long long k[5]; int pos[5];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < 5; ++s) { k[s] = 0x3f800000ull << 32; pos[s] = 0; }    // (1.0f, 0): a key is below it exactly when its distance is < 1.0f
  }
  __device__ __forceinline__ void insert(float dd, int ii, int pp) {          // branch-free; k stays ascending
    const unsigned long long key = (unsigned long long)__float_as_uint(dd) << 32 | (unsigned)ii;
    const bool c0 = key < k[0], c1 = key < k[1], c2 = key < k[2], c3 = key < k[3], c4 = key < k[4];
    k[4] = c3 ? k[3] : (c4 ? key : k[4]); pos[4] = c3 ? pos[3] : (c4 ? pp : pos[4]);
    k[3] = c2 ? k[2] : (c3 ? key : k[3]); pos[3] = c2 ? pos[2] : (c3 ? pp : pos[3]);
    k[2] = c1 ? k[1] : (c2 ? key : k[2]); pos[2] = c1 ? pos[1] : (c2 ? pp : pos[2]);
    k[1] = c0 ? k[0] : (c1 ? key : k[1]); pos[1] = c0 ? pos[0] : (c1 ? pp : pos[1]);
    k[0] = c0 ? key : k[0];               pos[0] = c0 ? pp : pos[0];
  }
};

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi, standing in for Eigen::SelfAdjointEigenSolver<Matrix3d>
// (reference src/laserMapping.cpp:605).  The operation sequence is fixed (DESIGN.md "Mapping") so that CPU restatements of it
// agree bit-for-bit; it is not Eigen's tridiagonal-QL algorithm, results differ from Eigen at the 1e-15 level.
// Returns ascending eigenvalues in vals and the eigenvector of the LARGEST one in dir (the only one the reference uses, :609).
__device__ __forceinline__ void sym_eigen3(const double A0[3][3], double vals[3], double dir[3]) {
  double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i][j] = 0.5 * (A0[i][j] + A0[j][i]);
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off <= 1e-22 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk; }
        a[p][q] = 0.0; a[q][p] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = cs * vkp - sn * vkq; v[k][q] = sn * vkp + cs * vkq; }
      }
  }
  // ascending order of the diagonal, same exchange sequence as the stand-in (order[] = {0,1,2}; swap when a later one is smaller)
  double dg[3] = {a[0][0], a[1][1], a[2][2]};
  double c0[3] = {v[0][0], v[1][0], v[2][0]}, c1[3] = {v[0][1], v[1][1], v[2][1]}, c2[3] = {v[0][2], v[1][2], v[2][2]};
  auto swp = [&](double& x, double& y, double* cx, double* cy) {
    const double tv = x; x = y; y = tv;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double tc = cx[k]; cx[k] = cy[k]; cy[k] = tc; }
  };
  if (dg[1] < dg[0]) swp(dg[0], dg[1], c0, c1);
  if (dg[2] < dg[0]) swp(dg[0], dg[2], c0, c2);
  if (dg[2] < dg[1]) swp(dg[1], dg[2], c1, c2);
  vals[0] = dg[0]; vals[1] = dg[1]; vals[2] = dg[2];
  dir[0] = c2[0]; dir[1] = c2[1]; dir[2] = c2[2];
}

// min |A x - b| for the 5 x 3 plane fit: Householder QR with column pivoting, standing in for colPivHouseholderQr().solve()
// (reference src/laserMapping.cpp:663); fixed operation sequence, see above.
__device__ __forceinline__ void lstsq_5x3(double a[5][3], double b[5], double x[3]) {
  int perm[3] = {0, 1, 2};
  int rank = 0;
  double maxnorm0 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (rank != k) break;
    int piv = k;
    double best = -1.0;
#pragma unroll
    for (int j = k; j < 3; ++j) {
      double s = 0.0;
#pragma unroll
      for (int i = k; i < 5; ++i) s += a[i][j] * a[i][j];
      if (s > best) { best = s; piv = j; }
    }
    if (k == 0) maxnorm0 = best;
    if (!(best > maxnorm0 * 1e-30)) break;
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      if (piv == j) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { const double tv = a[i][k]; a[i][k] = a[i][j]; a[i][j] = tv; }
        const int tp = perm[k]; perm[k] = perm[j]; perm[j] = tp;
      }
    }
    const double alpha = (a[k][k] > 0.0 ? -1.0 : 1.0) * sqrt(best);
    double v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = i < k ? 0.0 : a[i][k];
    v[k] -= alpha;
    double vv = 0.0;
#pragma unroll
    for (int i = k; i < 5; ++i) vv += v[i] * v[i];
    if (vv > 0.0) {
#pragma unroll
      for (int j = k; j < 3; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = k; i < 5; ++i) s += v[i] * a[i][j];
        s = 2.0 * s / vv;
#pragma unroll
        for (int i = k; i < 5; ++i) a[i][j] -= s * v[i];
      }
      double s = 0.0;
#pragma unroll
      for (int i = k; i < 5; ++i) s += v[i] * b[i];
      s = 2.0 * s / vv;
#pragma unroll
      for (int i = k; i < 5; ++i) b[i] -= s * v[i];
    }
    ++rank;
  }
  double y[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    if (k < rank) {
      double s = b[k];
#pragma unroll
      for (int j = k + 1; j < 3; ++j) if (j < rank) s -= a[k][j] * y[j];
      y[k] = s / a[k][k];
    }
  }
  x[0] = x[1] = x[2] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j) if (perm[k] == j) x[j] = y[k];
}

}  // namespace

constexpr int kMapSearchThreads = 256;
constexpr int kMapSearchU = 4;          // loads in flight per lane (measured, map_associate per step: 2: 7.76 ms, 4: 6.91, 6: 7.09, 8: 7.01)
constexpr int kMapSearchBlocks[2] = {16, 48};   // workgroups per sequence, corner / surf class
// Search half: lane per query, few registers, so that many waves hide the latency of the bucket walks.  Writes the five
// neighbours (x, y, z each; ascending (distance, index)) or a "not found" mark to a.knn[query].
// Rejected forms (measured slower in round 5, HISTORY.md): G lanes per query with LDS candidate lists ranked by counting, exact bucket
// tails under exec masks, the five neighbours with their coordinates in registers, a plain (block, sequence) grid.
template <int CLS>
__global__ __launch_bounds__(kMapSearchThreads) void k_map_search(MapArgs a, int nblk) {
  // XCD-aware work mapping (as in k_associate): workgroups are dealt round-robin over the 8 XCDs by linear id, and every XCD has its
  // own L2.  The bucketed submap of a sequence is read by all of that sequence's workgroups, so the 1-D grid is decoded
  // such that XCD x works through sequences x, x + 8, ...: one L2 fetches a sequence's submap instead of eight.
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int b = (slot / nblk) * 8 + xcd, blk = slot % nblk;
  if (b >= a.B) return;
  const MapSeq& ms = a.seq[b];
  const int n = ms.n_stack[CLS];
  const long long sb = (long long)b * (CLS == 0 ? a.R * 120 : a.cap);
  if (!ms.gate) return;
  double par[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) par[k] = ms.par[k];
  const int H = a.grid_H;
  const int* __restrict__ start = a.grid_start[CLS] + (long long)b * (H + 1);
  const float4* __restrict__ sorted = a.grid_sorted[CLS] + (long long)b * a.pool_cap;
  for (int i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) {          // grid-stride over the stack
    const float4 sel = associate_to_map(a.stack[CLS][sb + i], par);        // pointSel (:580, :646)
    const float gx = sel.x * kMapCellInv, gy = sel.y * kMapCellInv, gz = sel.z * kMapCellInv;
    const int cx = (int)floorf(gx), cy = (int)floorf(gy), cz = (int)floorf(gz);
    const int nx = gx - (float)cx >= 0.5f ? cx + 1 : cx - 1, ny = gy - (float)cy >= 0.5f ? cy + 1 : cy - 1, nz = gz - (float)cz >= 0.5f ? cz + 1 : cz - 1;
    Top5P top;
    top.init();
    // the reference discards the 5-NN result unless the 5th neighbour is closer than 1 m (:582, :650): collecting every point
    // with d < 1 from the 2x2x2 block and keeping the five smallest (distance, index) is an exact stand-in
    int s0[8], s1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {                                            // the 8 bucket heads first: independent 8-byte loads
      const unsigned h = map_bucket((c & 1) ? nx : cx, (c & 2) ? ny : cy, (c & 4) ? nz : cz, H);
      const MapIntPair v = *reinterpret_cast<const MapIntPair*>(reinterpret_cast<const char*>(start) + (h << 2));
      s0[c] = v.a; s1[c] = v.b;
    }
    // The eight cells sit in eight different buckets (map_bucket).  Other cells hashed into a bucket need no test of their own: every
    // point closer than 1 m lies in one of the eight cells, so a foreign point always fails d < 1.
    auto visit = [&](const float4& p, int at) {
      const float ddx = p.x - sel.x, ddy = p.y - sel.y, ddz = p.z - sel.z;
      const float d = (ddx * ddx + ddy * ddy) + ddz * ddz;                   // FLANN L2_Simple, f32
      if (d < 1.0f) top.insert(d, __float_as_int(p.w), at);
    };
    constexpr int U = kMapSearchU;                                           // independent loads in flight per lane
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      for (int k = s0[c]; k < s1[c]; k += U) {
        const int m = s1[c] - k;
        float4 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = sorted[u < m ? k + u : k];        // a lane past the end of its bucket re-reads its own entry k
#pragma unroll
        for (int u = 0; u < U; ++u) if (u < m) visit(p[u], k + u);
      }
    }
    float4* out = a.knn + ((long long)b * a.cap + i) * 4;
    if ((unsigned)(top.k[4] >> 32) < 0x3f800000u) {                          // pointSearchSqDis[4] < 1.0
      float4 q[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) q[s] = sorted[top.pos[s]];
      out[0] = make_float4(q[0].x, q[0].y, q[0].z, 1.f);
      out[1] = make_float4(q[1].x, q[1].y, q[1].z, q[2].x);
      out[2] = make_float4(q[2].y, q[2].z, q[3].x, q[3].y);
      out[3] = make_float4(q[3].z, q[4].x, q[4].y, q[4].z);
    } else {
      out[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Fit half: line fit (corner) / plane fit (surf) in f64 on the five neighbours, validity tests, factor record.
template <int CLS>
__global__ __launch_bounds__(256) void k_map_fit(MapArgs a) {
  const int b = blockIdx.y;
  const MapSeq& ms = a.seq[b];
  const int n = ms.n_stack[CLS];
  const long long sb = (long long)b * (CLS == 0 ? a.R * 120 : a.cap);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_wcnt[4];
  int* tile_cnt = a.rec_tiles + (long long)b * a.rec_tiles_per_seq + (CLS == 0 ? 0 : a.rec_tiles_corner);
  // Only the valid factors are stored, compacted per tile of 256 consecutive stack points (record (tile, k) = the k-th valid
  // point of the tile, in stack order; the number per tile in rec_tiles): k_map_solve then reads nothing but live records, densely.
  // Corner points pass the line test a third of the time; their 80-byte records were re-streamed by every LM evaluation before.
  for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
    const int i = i0 + tid;
    const bool live = i < n;
    const float4 ori = a.stack[CLS][sb + (live ? i : i0)];                 // pointOri (:578, :644)
    bool valid = false;
    double ra[3] = {0, 0, 0}, rb[3] = {0, 0, 0}, rd = 0.0;
    float nxs[5], nys[5], nzs[5];
    bool found = false;
    if (ms.gate && live) {
      const float4* in = a.knn + ((long long)b * a.cap + i) * 4;
      const float4 q0 = in[0], q1 = in[1], q2 = in[2], q3 = in[3];
      found = q0.w != 0.f;
      nxs[0] = q0.x; nys[0] = q0.y; nzs[0] = q0.z;
      nxs[1] = q1.x; nys[1] = q1.y; nzs[1] = q1.z;
      nxs[2] = q1.w; nys[2] = q2.x; nzs[2] = q2.y;
      nxs[3] = q2.z; nys[3] = q2.w; nzs[3] = q3.x;
      nxs[4] = q3.y; nys[4] = q3.z; nzs[4] = q3.w;
    }
    if (found) {
      if (CLS == 0) {
        double cxs = 0.0, cys = 0.0, czs = 0.0;
#pragma unroll
        for (int j = 0; j < 5; ++j) { cxs = cxs + (double)nxs[j]; cys = cys + (double)nys[j]; czs = czs + (double)nzs[j]; }
        const double ctr[3] = {cxs / 5.0, cys / 5.0, czs / 5.0};
        double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const double zm[3] = {(double)nxs[j] - ctr[0], (double)nys[j] - ctr[1], (double)nzs[j] - ctr[2]};
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) cov[r][c] = cov[r][c] + zm[r] * zm[c];
        }
        double vals[3], dir[3];
        sym_eigen3(cov, vals, dir);
        if (vals[2] > 3 * vals[1]) {                                       // :611
          valid = true;
#pragma unroll
          for (int k = 0; k < 3; ++k) { ra[k] = 0.1 * dir[k] + ctr[k]; rb[k] = -0.1 * dir[k] + ctr[k]; }
        }
      } else {
        double A[5][3], B[5] = {-1, -1, -1, -1, -1}, x[3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { A[j][0] = nxs[j]; A[j][1] = nys[j]; A[j][2] = nzs[j]; }
        lstsq_5x3(A, B, x);
        const double len = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const double d = 1 / len;                                          // negative_OA_dot_norm (:664)
        const double nx = x[0] / len, ny = x[1] / len, nz = x[2] / len;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 5; ++j)
          if (fabs(nx * (double)nxs[j] + ny * (double)nys[j] + nz * (double)nzs[j] + d) > 0.2) ok = false;   // :672-678
        if (ok) { valid = true; ra[0] = nx; ra[1] = ny; ra[2] = nz; rd = d; }
      }
    }
    const unsigned long long vm = __ballot(valid);
    if (lane == 0) s_wcnt[wave] = __popcll(vm);
    __syncthreads();
    int rank = __popcll(vm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += s_wcnt[w];
    if (tid == 0) tile_cnt[i0 >> 8] = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (valid) {
      if (CLS == 0) {
        MapEdgeRec e;
        e.cp[0] = ori.x; e.cp[1] = ori.y; e.cp[2] = ori.z;
#pragma unroll
        for (int k = 0; k < 3; ++k) { e.a[k] = ra[k]; e.b[k] = rb[k]; }
        e.pad = i;
        a.edges[(long long)b * a.R * 120 + i0 + rank] = e;
      } else {
        MapNormRec e;
        e.cp[0] = ori.x; e.cp[1] = ori.y; e.cp[2] = ori.z;
#pragma unroll
        for (int k = 0; k < 3; ++k) e.n[k] = ra[k];
        e.d = rd; e.pad = i;
        a.norms[(long long)b * a.cap + i0 + rank] = e;
      }
    }
    __syncthreads();                                                       // s_wcnt is reused by the next tile
  }
}

// =======================================================================================================
// solve
// =======================================================================================================
constexpr int kMapSolveThreads = 256;    // more waves do not pay: the kernel needs 256 VGPRs per lane for the f64 sums
template <bool WITH_JAC>
__device__ void map_evaluate(const MapArgs& a, int b, const int* s_pref, const double q[4], const double t[3], double* acc, int* n_edge, int* n_norm) {
  const int tid = threadIdx.x;
  const MapSeq& ms = a.seq[b];
  const MapEdgeRec* E = a.edges + (long long)b * a.R * 120;
  const MapNormRec* P = a.norms + (long long)b * a.cap;
  int ne = 0, np = 0;
  // records are fetched a few at a time ahead of the f64 work (same per-thread order as a plain strided loop)
  auto edge_term = [&](const MapEdgeRec& e) {
    ++ne;
    double rcp[3];
    quat_rotate(q, e.cp[0], e.cp[1], e.cp[2], rcp);
    const double lp[3] = {rcp[0] + t[0], rcp[1] + t[1], rcp[2] + t[2]};
    const double dex = e.a[0] - e.b[0], dey = e.a[1] - e.b[1], dez = e.a[2] - e.b[2];
    const double inv = 1.0 / sqrt(dex * dex + dey * dey + dez * dez);
    const double ux = lp[0] - e.a[0], uy = lp[1] - e.a[1], uz = lp[2] - e.a[2], vx = lp[0] - e.b[0], vy = lp[1] - e.b[1], vz = lp[2] - e.b[2];
    const double r0 = (uy * vz - uz * vy) * inv, r1 = (uz * vx - ux * vz) * inv, r2 = (ux * vy - uy * vx) * inv;
    double rho0, rho1;
    huber(r0 * r0 + r1 * r1 + r2 * r2, &rho0, &rho1);
    acc[27] += 0.5 * rho0;
    if (WITH_JAC) {
      const double wx = -dex * inv, wy = -dey * inv, wz = -dez * inv;
      const double A[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
      const double Bm[3][3] = {{0, 2 * rcp[2], -2 * rcp[1]}, {-2 * rcp[2], 0, 2 * rcp[0]}, {2 * rcp[1], -2 * rcp[0], 0}};
      const double rr[3] = {r0, r1, r2};
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        double J[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          J[c] = A[row][0] * Bm[0][c] + A[row][1] * Bm[1][c] + A[row][2] * Bm[2][c];
          J[3 + c] = A[row][c];
        }
        add_row(acc, J, rr[row], rho1);
      }
    }
  };
  constexpr int U = 4;
  // dense index d over the valid records -> tile by binary search in the tile prefixes (LDS, built once per launch), then the
  // record at tile * 256 + (d - prefix[tile])
  const int nt0 = (ms.n_stack[0] + 255) >> 8, nt1 = (ms.n_stack[1] + 255) >> 8;
  const int* pre0 = s_pref, *pre1 = s_pref + nt0 + 1;
  const int n0 = pre0[nt0], n1 = pre1[nt1];
  auto locate = [](const int* pre, int nt, int d) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= d) lo = mid; else hi = mid - 1; }
    return (lo << 8) + (d - pre[lo]);
  };
  for (int d0 = tid; d0 < n0; d0 += U * kMapSolveThreads) {
    MapEdgeRec e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int d = d0 + u * kMapSolveThreads; e[u] = E[locate(pre0, nt0, d < n0 ? d : d0)]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (d0 + u * kMapSolveThreads < n0) edge_term(e[u]);
  }
  auto norm_term = [&](const MapNormRec& p) {
    ++np;
    double rcp[3];
    quat_rotate(q, p.cp[0], p.cp[1], p.cp[2], rcp);
    // LidarPlaneNormFactor (reference src/lidarFactor.hpp:116-123): r = n . (q cp + t) + d
    const double r = (p.n[0] * (rcp[0] + t[0]) + p.n[1] * (rcp[1] + t[1]) + p.n[2] * (rcp[2] + t[2])) + p.d;
    double rho0, rho1;
    huber(r * r, &rho0, &rho1);
    acc[27] += 0.5 * rho0;
    if (WITH_JAC) {
      const double J[6] = {2.0 * (p.n[2] * rcp[1] - p.n[1] * rcp[2]), 2.0 * (p.n[0] * rcp[2] - p.n[2] * rcp[0]), 2.0 * (p.n[1] * rcp[0] - p.n[0] * rcp[1]),
                           p.n[0], p.n[1], p.n[2]};
      add_row(acc, J, r, rho1);
    }
  };
  for (int d0 = tid; d0 < n1; d0 += U * kMapSolveThreads) {
    MapNormRec pr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int d = d0 + u * kMapSolveThreads; pr[u] = P[locate(pre1, nt1, d < n1 ? d : d0)]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (d0 + u * kMapSolveThreads < n1) norm_term(pr[u]);
  }
  *n_edge = ne;
  *n_norm = np;
}

__global__ __launch_bounds__(kMapSolveThreads) void k_map_solve(MapArgs a, int iter, int last) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double s_red[(kMapSolveThreads / 64) * 28];
  extern __shared__ int s_pref[];                                            // exclusive prefixes of the valid records per tile: corner, then surf
  MapSeq& ms = a.seq[b];
  {
    const int nt0 = (ms.n_stack[0] + 255) >> 8, nt1 = (ms.n_stack[1] + 255) >> 8;
    const int* tc = a.rec_tiles + (long long)b * a.rec_tiles_per_seq;
    if (tid == 0) {                                                          // a few hundred tiles at most: one thread
      int run = 0;
      for (int k = 0; k < nt0; ++k) { s_pref[k] = run; run += ms.gate ? tc[k] : 0; }
      s_pref[nt0] = run;
      run = 0;
      for (int k = 0; k < nt1; ++k) { s_pref[nt0 + 1 + k] = run; run += ms.gate ? tc[a.rec_tiles_corner + k] : 0; }
      s_pref[nt0 + 1 + nt1] = run;
    }
    __syncthreads();
  }
  double q[4] = {ms.par[0], ms.par[1], ms.par[2], ms.par[3]};
  double t[3] = {ms.par[4], ms.par[5], ms.par[6]};
  const LmResult lm = lm_solve_block<kMapSolveThreads / 64>([&](bool with_jac, const double* qq, const double* tt, double* acc, int* ne, int* np) {
    if (with_jac) map_evaluate<true>(a, b, s_pref, qq, tt, acc, ne, np); else map_evaluate<false>(a, b, s_pref, qq, tt, acc, ne, np);
  }, q, t, a.lm_max_iterations, s_red);
  if (tid == 0) {
    if (lm.termination == 5) {                                               // FAILURE: `parameters` stay what they were (ceres::Solve restores them)
      for (int k = 0; k < 4; ++k) q[k] = ms.par[k];
      for (int k = 0; k < 3; ++k) t[k] = ms.par[4 + k];
    }
    for (int k = 0; k < 4; ++k) ms.par[k] = q[k];
    for (int k = 0; k < 3; ++k) ms.par[4 + k] = t[k];
    ms.factor_num[iter][0] = lm.n_a;
    ms.factor_num[iter][1] = lm.n_b;
    ms.lm_iterations[iter] = lm.iterations;
    ms.lm_termination[iter] = lm.termination;
    if (last) {                                                              // transformUpdate (:148-152)
      const double qo[4] = {ms.q_wodom[0], ms.q_wodom[1], ms.q_wodom[2], ms.q_wodom[3]};
      const double n2 = qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3];
      const double inv[4] = {-qo[0] / n2, -qo[1] / n2, -qo[2] / n2, qo[3] / n2};        // Eigen inverse(): conjugate / squaredNorm
      double qm[4], rt[3];
      quat_mul(q, inv, qm);
      quat_rotate(qm, ms.t_wodom[0], ms.t_wodom[1], ms.t_wodom[2], rt);
      for (int k = 0; k < 4; ++k) ms.q_wmap_wodom[k] = qm[k];
      for (int k = 0; k < 3; ++k) ms.t_wmap_wodom[k] = t[k] - rt[k];
      ms.frame_count += 1;
    }
  }
}

// =======================================================================================================
// map insertion (:737-783)
// =======================================================================================================
__global__ __launch_bounds__(256) void k_map_cubeid(MapArgs a) {
  const int b = blockIdx.y, cls = blockIdx.z;
  const MapSeq& ms = a.seq[b];
  const long long sb = (long long)b * (cls == 0 ? a.R * 120 : a.cap);
  double par[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) par[k] = ms.par[k];
  for (int i0 = blockIdx.x * 256; i0 < ms.n_stack[cls]; i0 += gridDim.x * 256) {
  const int i = i0 + threadIdx.x;
  const bool live = i < ms.n_stack[cls];
  const float4 w = associate_to_map(a.stack[cls][sb + (live ? i : 0)], par);
  const int ci = cube_coord((double)w.x, ms.cen[0]), cj = cube_coord((double)w.y, ms.cen[1]), ck = cube_coord((double)w.z, ms.cen[2]);
  int id = -1;
  if (live && ci >= 0 && ci < kMapW && cj >= 0 && cj < kMapH && ck >= 0 && ck < kMapD) id = ci + kMapW * cj + kMapW * kMapH * ck;
  // the points of one sweep fall into a dozen cubes: one atomic per distinct cube in the wave instead of one per point
  unsigned long long todo = __ballot(id >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int c = __shfl(id, leader, 64);
    const unsigned long long same = __ballot(id == c);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&a.addcnt[((long long)b * 2 + cls) * kMapCubes + c], __popcll(same));
    todo &= ~same;
  }
  if (live) { a.stack_world[cls][sb + i] = w; a.stack_cube[cls][sb + i] = id; }
  }
}

// capacity: a cube that would overflow its segment moves to a fresh one of twice the needed size (bump allocation from the
// class pool; the old segment is abandoned).  Then the append cursor of every touched cube is published.
__global__ __launch_bounds__(256) void k_map_reserve(MapArgs a) {
  const int b = blockIdx.x, cls = blockIdx.y, tid = threadIdx.x;
  CubeDesc* T = cube_table(a, b, cls);
  int* add = a.addcnt + ((long long)b * 2 + cls) * kMapCubes;
  int* cur = a.cursor + ((long long)b * 2 + cls) * kMapCubes;
  float4* pool = a.pool[cls] + (long long)b * a.pool_cap;
  MapSeq& ms = a.seq[b];
  __shared__ int s_n;
  __shared__ int s_list[256][4];   // cube, old offset, new offset, count   (moves of this round)
  for (int base = 0; base < kMapCubes; base += 256) {
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int c = base + tid;
    if (c < kMapCubes) {
      const int ad = add[c];
      if (ad > 0) {
        CubeDesc d = T[c];
        if (d.cnt + ad > d.cap) {
          int want = 2 * (d.cnt + ad);
          if (want < 256) want = 256;
          // reserve only what fits (compare-and-swap: a blind add followed by a subtraction on failure would let another thread's successful
          // reservation start inside the range handed back, and a later one overlap it)
          int off = atomicAdd(&ms.pool_used[cls], 0);
          for (;;) {
            if (off + want > a.pool_cap) { off = -1; break; }
            const int seen = atomicCAS(&ms.pool_used[cls], off, off + want);
            if (seen == off) break;
            off = seen;
          }
          if (off < 0) {
            atomicOr(&ms.err, kMapErrPool);
            add[c] = -1;                                                     // the scatter pass skips this cube
          } else {
            const int k = atomicAdd(&s_n, 1);
            s_list[k][0] = c; s_list[k][1] = d.off; s_list[k][2] = off; s_list[k][3] = d.cnt;
            d.off = off; d.cap = want;
            T[c] = d;
          }
        }
      }
    }
    __syncthreads();
    const int nmove = s_n;
    for (int k = 0; k < nmove; ++k)
      for (int i = tid; i < s_list[k][3]; i += 256) pool[s_list[k][2] + i] = pool[s_list[k][1] + i];
    __syncthreads();
    if (c < kMapCubes) {
      const int ad = add[c];
      cur[c] = T[c].cnt;
      if (ad > 0) T[c].cnt += ad;
    }
    __syncthreads();
  }
}

// stable append: one wave per (sequence, class) walks the stack in order, 64 points a step
__global__ __launch_bounds__(64) void k_map_scatter(MapArgs a) {
  const int b = blockIdx.x, cls = blockIdx.y, lane = threadIdx.x;
  const MapSeq& ms = a.seq[b];
  const int n = ms.n_stack[cls];
  const CubeDesc* T = cube_table(a, b, cls);
  int* add = a.addcnt + ((long long)b * 2 + cls) * kMapCubes;
  const int* cur = a.cursor + ((long long)b * 2 + cls) * kMapCubes;
  float4* pool = a.pool[cls] + (long long)b * a.pool_cap;
  const long long sb = (long long)b * (cls == 0 ? a.R * 120 : a.cap);
  __shared__ int s_cur[kMapCubes];                                           // absolute append position of every touched cube (-1: no room, skipped)
  for (int c = lane; c < kMapCubes; c += 64) {
    const int ad = add[c];
    s_cur[c] = ad > 0 ? T[c].off + cur[c] : -1;                              // (descriptor and cursor only of the dozen cubes that receive points)
  }
  __syncthreads();
  constexpr int U = 4;                                                       // rows of 64 points fetched ahead: one wave per (sequence, class) lives
  for (int base0 = 0; base0 < n; base0 += 64 * U) {                          // off memory latency, so the loads of four rows are in flight together
    int ids[U];
    float4 ws[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base0 + u * 64 + lane;
      ids[u] = -1;
      ws[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n) { ids[u] = a.stack_cube[cls][sb + i]; ws[u] = a.stack_world[cls][sb + i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = ids[u];
      const float4 w = ws[u];
      unsigned long long todo = __ballot(id >= 0);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __shfl(id, leader, 64);
        const unsigned long long same = __ballot(id == c);
        const int start = s_cur[c];
        if (id == c && start >= 0) pool[start + __popcll(same & ((1ull << lane) - 1ull))] = w;
        __syncthreads();
        if (lane == leader && start >= 0) s_cur[c] = start + __popcll(same);
        __syncthreads();
        todo &= ~same;
      }
    }
  }
  __syncthreads();
  for (int c = lane; c < kMapCubes; c += 64) add[c] = 0;                     // ready for the next frame
}

__global__ __launch_bounds__(256) void k_map_register(MapArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.meta[b].n_cloud) return;
  const MapSeq& ms = a.seq[b];
  double par[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) par[k] = ms.par[k];
  a.registered[(long long)b * a.cap + i] = associate_to_map(a.full[(long long)b * a.cap + i], par);
}

// the same from the ring slabs of scan registration (k_front): ring r of the sweep, written at the ring's place in the dense numbering
__global__ __launch_bounds__(256) void k_map_register_slabs(MapArgs a) {
  const int r = blockIdx.x, b = blockIdx.y;
  const int start = a.ringstart[b * (a.R + 1) + r], n = a.ringstart[b * (a.R + 1) + r + 1] - start;
  const MapSeq& ms = a.seq[b];
  double par[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) par[k] = ms.par[k];
  const float4* src = a.slabs + ((long long)b * a.R + r) * a.slab;
  float4* dst = a.registered + (long long)b * a.cap + start;
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = associate_to_map(src[i], par);
}

// =======================================================================================================
// pool occupancy report (host-side pool growth, aloam_capi.hip map_ensure_capacity)
// =======================================================================================================
// After every step: live points per (sequence, class) = sum of the cube populations (what the pool must hold at least), the largest of
// them over the batch and the largest incoming stack so far, written to pinned host memory with the step number LAST, so that the host
// can size the pools for the steps it is about to queue without waiting for the device.
__global__ __launch_bounds__(256) void k_map_report(MapArgs a, int step) {
  const int b = blockIdx.x, cls = blockIdx.y, tid = threadIdx.x;
  const CubeDesc* T = cube_table(a, b, cls);
  __shared__ int s_red[4][4];
  __shared__ int s_last;
  int live = 0;
  for (int c = tid; c < kMapCubes; c += 256) live += T[c].cnt;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) live += __shfl_xor(live, d, 64);
  if ((tid & 63) == 0) s_red[0][tid >> 6] = live;
  __syncthreads();
  if (tid == 0) {
    live = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    atomicExch(&a.live[b * 2 + cls], live);
    __threadfence();
    s_last = atomicAdd(&a.report_dev[0], 1) == (int)(gridDim.x * gridDim.y) - 1;
  }
  __syncthreads();
  if (!s_last) return;                                                       // the last workgroup to finish folds the batch
  int m[4] = {0, 0, 0, 0};                                                   // live corner / surf, stack corner / surf
  for (int i = tid; i < a.B; i += 256) {
    m[0] = max(m[0], atomicAdd(&a.live[i * 2], 0)); m[1] = max(m[1], atomicAdd(&a.live[i * 2 + 1], 0));
    m[2] = max(m[2], a.seq[i].n_stack[0]); m[3] = max(m[3], a.seq[i].n_stack[1]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m[k] = max(m[k], __shfl_xor(m[k], d, 64));
    if ((tid & 63) == 0) s_red[k][tid >> 6] = m[k];
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 4; ++k) m[k] = max(max(s_red[k][0], s_red[k][1]), max(s_red[k][2], s_red[k][3]));
    a.report_dev[0] = 0;
    a.report_dev[1] = max(a.report_dev[1], m[2]);                            // largest stacks of any step so far
    a.report_dev[2] = max(a.report_dev[2], m[3]);
    volatile int* h = a.report_host;
    h[1] = m[0]; h[2] = m[1]; h[3] = a.report_dev[1]; h[4] = a.report_dev[2];
    __threadfence_system();
    h[0] = step;
  }
}

// =======================================================================================================
// launchers
// =======================================================================================================
void launch_map_begin(const MapArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_map_begin, dim3(a.B), dim3(256), 0, s, a); }
void launch_map_stack_segments(const MapArgs& a, const VoxArgs& v, hipStream_t s) {
  hipLaunchKernelGGL(k_map_stack_segments, dim3((a.B * 2 + 255) / 256), dim3(256), 0, s, a, v);
}
void launch_map_cube_segments(const MapArgs& a, const VoxArgs& v, hipStream_t s) {
  hipLaunchKernelGGL(k_map_cube_segments, dim3((a.B * 2 * kMapValidMax + 255) / 256), dim3(256), 0, s, a, v);
}
constexpr int kVoxSmallRuns = 8192, kVoxBigRuns = 24576;
constexpr int kVoxBigThreads = 1024;   // the 5 VGPRs this instance spills are values computed before the segment loop and reloaded once per segment (two scratch loads
                                       // per 40 k-point segment); 768 threads (170 registers) and 512 threads (255) spill the same five: the allocator's choice, not the budget
constexpr size_t vox_lds_bytes(int nt, int capr, int capn) { return (size_t)capr * 6 + (size_t)capn / 8 + 2 * 128 * (size_t)(nt / 64) + sizeof(int) * ((size_t)(capr / nt) * (nt / 64) + 1 + 48) + sizeof(float) * 6 * (nt / 64) + 64; }
static_assert(vox_lds_bytes(kVoxBigThreads, kVoxBigRuns, kVoxBigN) <= 163840, "one workgroup may use the whole 160 KiB of a CU, not more");
int prepare_voxel_filter() {     // the 1024-thread instance needs > 64 KiB of dynamic LDS (attribute of the function on the current device)
  return hipFuncSetAttribute((const void*)k_vox_lds<kVoxBigThreads, kVoxBigRuns, kVoxBigN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)vox_lds_bytes(kVoxBigThreads, kVoxBigRuns, kVoxBigN)) == hipSuccess ? 0 : -1;
}
void launch_voxel_filter(const VoxArgs& v, int tile_bound, hipStream_t s) {
  // single-workgroup LDS filter first (grid-stride over the two device-built lists) ...
  const int nsmall = v.n_segs < 4096 ? v.n_segs : 4096, nbig = v.n_segs < 1024 ? v.n_segs : 1024;
  hipLaunchKernelGGL((k_vox_lds<kVoxBigThreads, kVoxBigRuns, kVoxBigN>), dim3(nbig), dim3(kVoxBigThreads), vox_lds_bytes(kVoxBigThreads, kVoxBigRuns, kVoxBigN), s, v, 1);
  hipLaunchKernelGGL((k_vox_lds<256, kVoxSmallRuns, kVoxSmallN>), dim3(nsmall), dim3(256), vox_lds_bytes(256, kVoxSmallRuns, kVoxSmallN), s, v, 0);
  hipLaunchKernelGGL((k_vox_lds<64, kVoxTinyN, kVoxTinyN>), dim3(v.n_segs < 16384 ? v.n_segs : 16384), dim3(64), vox_lds_bytes(64, kVoxTinyN, kVoxTinyN), s, v, 2);
  // ... then the general path for whatever did not fit: every kernel returns at once when counters[4] == 0 (no tiles)
  if (tile_bound > 1024) tile_bound = 1024;            // the kernels stride over the device-side tile list
  hipLaunchKernelGGL(k_vox_setup, dim3(1), dim3(1024), 0, s, v);
  hipLaunchKernelGGL(k_vox_bbox, dim3(tile_bound), dim3(256), 0, s, v);
  hipLaunchKernelGGL(k_vox_keys_sort, dim3(tile_bound), dim3(256), 0, s, v);
  for (int level = 0; level < v.levels; ++level) hipLaunchKernelGGL(k_vox_merge, dim3(tile_bound), dim3(256), 0, s, v, level);
  hipLaunchKernelGGL(k_vox_heads, dim3(tile_bound), dim3(256), 0, s, v);
  hipLaunchKernelGGL(k_vox_scan, dim3(1), dim3(1024), 0, s, v);
  hipLaunchKernelGGL(k_vox_emit, dim3(tile_bound), dim3(256), 0, s, v);
  hipLaunchKernelGGL(k_vox_copyback, dim3(tile_bound), dim3(256), 0, s, v);
}
static size_t mapgrid_lds_bytes(int H) { return sizeof(int) * ((size_t)H + 1024 + 160); }
static_assert(sizeof(int) * ((size_t)kMapGridMaxH + 1024 + 160) <= 163840, "the bucket table of k_mapgrid_build must fit one CU's LDS");
int prepare_map_grid(int H) {
  if (H > kMapGridMaxH || H < 1024 || (H & (H - 1))) return -1;
  return hipFuncSetAttribute((const void*)k_mapgrid_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mapgrid_lds_bytes(H)) == hipSuccess ? 0 : -1;
}
void launch_map_grid(const MapArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_mapgrid_build, dim3(a.B, 2), dim3(1024), mapgrid_lds_bytes(a.grid_H), s, a);
}
void launch_map_associate(const MapArgs& a, int iter, hipStream_t s) {
  const int by = (a.B + 7) / 8 * 8;                      // padded so that every (XCD, sequence slot) pair exists
  (void)iter;
  hipLaunchKernelGGL(k_map_search<0>, dim3(kMapSearchBlocks[0] * by), dim3(kMapSearchThreads), 0, s, a, kMapSearchBlocks[0]);
  hipLaunchKernelGGL(k_map_fit<0>, dim3(16, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_map_search<1>, dim3(kMapSearchBlocks[1] * by), dim3(kMapSearchThreads), 0, s, a, kMapSearchBlocks[1]);
  hipLaunchKernelGGL(k_map_fit<1>, dim3(48, a.B), dim3(256), 0, s, a);
}
void launch_map_solve(const MapArgs& a, int iter, bool last, hipStream_t s) {
  hipLaunchKernelGGL(k_map_solve, dim3(a.B), dim3(kMapSolveThreads), sizeof(int) * (size_t)(a.rec_tiles_per_seq + 2), s, a, iter, last ? 1 : 0);
}
void launch_map_insert(const MapArgs& a, float4* staging, hipStream_t s) {
  hipLaunchKernelGGL(k_map_cubeid, dim3(32, a.B, 2), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_map_compact_plan, dim3(a.B, 2), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_map_compact_move<0>, dim3(64, a.B, 2), dim3(256), 0, s, a, staging);
  hipLaunchKernelGGL(k_map_compact_move<1>, dim3(64, a.B, 2), dim3(256), 0, s, a, staging);
  hipLaunchKernelGGL(k_map_reserve, dim3(a.B, 2), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_map_scatter, dim3(a.B, 2), dim3(64), 0, s, a);
}
void launch_map_register(const MapArgs& a, hipStream_t s) {
  if (a.slabs) hipLaunchKernelGGL(k_map_register_slabs, dim3(a.R, a.B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_map_register, dim3((a.cap + 255) / 256, a.B), dim3(256), 0, s, a);
}
void launch_map_report(const MapArgs& a, int step, hipStream_t s) { hipLaunchKernelGGL(k_map_report, dim3(a.B, 2), dim3(256), 0, s, a, step); }

}  // namespace aloam
