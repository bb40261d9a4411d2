// a-loam_amd/csrc/registration_kernels.hip — gfx950 kernels for A-LOAM scan registration.
//
// Replaces the body of laserCloudHandler (reference src/scanRegistration.cpp:127-411) for a BATCH of
// independent sweeps.  Kernel <-> reference map:
//   k_find_ends       :136-153   first / last kept point, startOri / endOri
//   k_front           :85-112,156-241  the front end in one pass: NaN + range filter, ring id, azimuth, halfPassed, relTime -> intensity and
//                                the STABLE per-ring compaction into one slab per ring (decoupled look-back over the blocks of a sweep)
//   k_ring_starts     :246-252   ring lengths -> start of every ring in the reference's dense numbering, cloud size
//   k_dense_cloud     :246-252   the dense ring-by-ring cloud from the slabs, only when a consumer of the full cloud asks
//   k_ring_features   :256-407   one workgroup per (sweep, ring): 11-tap curvature from alternating 266-point LDS
//                                tiles; std::sort + greedy corner / flat picking with neighbour suppression evaluated as
//                                an iterative arg-max per sector (no sort); less-flat gather + 0.2 m voxel centroids
//                                (pcl::VoxelGrid stand-in: voxel cells packed into LDS during the curvature pass, runs of
//                                same-voxel points sorted by an LDS bitonic); the ring workgroups publish their counts to
//                                each other, so every pick and every less-flat centroid is written once, at its final place
//                                in the reference's output order (:304-310,356,407)
//   k_cloud_sizes                sizes of the four feature clouds = sums of the published ring counts
//
// All of it is HBM/latency-bound integer + f32 work: coalesced 16-B loads, LDS staging, wave64 ballots; no MFMA.
#include "aloam_device.hpp"
#include "registration_kernels.hpp"

namespace aloam {

// -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_find_ends(RegArgs a, const int* __restrict__ n_in) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_in[b];
  const char* in = a.in + (long long)b * a.seq_stride;
  __shared__ int s_first, s_last;
  if (tid == 0) { s_first = 0x7fffffff; s_last = -1; }
  __syncthreads();
  // 4096 points per iteration: KITTI-like sweeps end with ~18k points inside minimum_range (lowest rings)
  for (int base = 0; base < n; base += 4096) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + k * 1024 + tid;
      if (i < n && point_kept(load_point(in, i, a.pt_stride), a.min_range)) { atomicMin(&s_first, i); any = true; }
    }
    if (__syncthreads_or(any)) break;
  }
  for (int base = 0; base < n; base += 4096) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = n - 1 - (base + k * 1024 + tid);
      if (i >= 0 && point_kept(load_point(in, i, a.pt_stride), a.min_range)) { atomicMax(&s_last, i); any = true; }
    }
    if (__syncthreads_or(any)) break;
  }
  __syncthreads();
  if (tid == 0) {
    SeqMeta m = a.meta[b];
    m.n_in = n;
    m.first_kept = s_first;
    m.last_kept = s_last;
    m.half_idx = 0x7fffffff;
    m.err = 0;
    m.n_cloud = 0;
    m.start_ori = 0.f;
    m.end_ori = 0.f;
    if (s_last < 0) {
      m.err = kErrEmpty;
    } else {
      const float4 p0 = load_point(in, s_first, a.pt_stride);
      const float4 p1 = load_point(in, s_last, a.pt_stride);
      const float startOri = -atan2f_port(p0.y, p0.x);                                  // :141
      float endOri = (float)((double)(-atan2f_port(p1.y, p1.x)) + 2 * M_PI);             // :142-144
      if ((double)(endOri - startOri) > 3 * M_PI) endOri = (float)((double)endOri - 2 * M_PI);      // :146-149
      else if ((double)(endOri - startOri) < M_PI) endOri = (float)((double)endOri + 2 * M_PI);     // :150-153
      m.start_ori = startOri;
      m.end_ori = endOri;
    }
    if (n > a.cap) m.err |= kErrPointCap;
    a.meta[b] = m;
  }
}

// Ring id from the elevation angle in degrees, exactly the expressions of reference src/scanRegistration.cpp:169-205; -1 = rejected.
// Monotone (non-increasing ring id) in the angle over one contiguous accepted interval, which is what the fast path below relies on.
__device__ __forceinline__ int ring_from_angle(float angle, int R) {
  int scanID;
  if (R == 16) {
    scanID = (int)((double)((angle + 15) / 2) + 0.5);
    return (scanID > R - 1 || scanID < 0) ? -1 : scanID;
  }
  if (R == 32) {
    scanID = (int)(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);
    return (scanID > R - 1 || scanID < 0) ? -1 : scanID;
  }
  if ((double)angle >= -8.83) scanID = (int)((double)(2 - angle) * 3.0 + 0.5);
  else scanID = R / 2 + (int)((-8.83 - (double)angle) * 2.0 + 0.5);
  if ((double)angle > 2 || (double)angle < -24.33 || scanID > 50 || scanID < 0) return -1;
  return scanID;
}
// Ring id of a kept point (reference src/scanRegistration.cpp:166-205); -1 = rejected.
// `atan` / `sqrt` are unqualified at :166 -> double overloads on the pinned toolchain (sqrt's argument is an f32 sum), the result is
// rounded to float.  The f64 atan is the most expensive thing this kernel does, and its 16 extra digits only matter within ~1e-6
// degrees of a ring boundary: the angle is first evaluated in f32 (z * rsq(x^2 + y^2) through a degree-6 minimax polynomial of
// atan(t) / t on |t| <= 0.65, i.e. elevations up to 33 degrees; total error < 2e-5 degrees), and if the ring decision is the same
// 2e-4 degrees below and above it — the decision is monotone in the angle — that IS the decision of the exact angle.  Otherwise
// (0.1 % of the points, and anything steeper than 33 degrees) the exact f64 expression decides.
__device__ __forceinline__ int ring_of(const float4& p, int R, int ring_from_field) {
  if (ring_from_field) {
    const int scanID = (int)p.w;
    return (scanID > R - 1 || scanID < 0) ? -1 : scanID;
  }
  const float s2 = p.x * p.x + p.y * p.y;
  {
    const float t = p.z * __frsqrt_rn(s2), u = t * t;
    const float poly = 1.0f + u * (-0.3333319425582886f + u * (0.19994600117206573f + u * (-0.1420508623123169f + u * (0.10522426664829254f + u * (-0.06763934344053268f + u * 0.025188861414790154f)))));
    const float fast = t * poly * 57.29577951f;
    const int lo = ring_from_angle(fast - 2e-4f, R), hi = ring_from_angle(fast + 2e-4f, R);
    if (lo == hi && fabsf(t) <= 0.65f) return lo;                             // NaN / inf fail the second test
  }
  const float angle = (float)(atan((double)p.z / sqrt((double)s2)) * 180 / M_PI);
  return ring_from_angle(angle, R);
}

// -------------------------------------------------------------------------------------------------------
// The front end in ONE pass over the sweep (reference src/scanRegistration.cpp:156-252): ring id, azimuth, halfPassed, relTime -> intensity and
// the stable per-ring compaction, each point read once and written once.  (Rounds 1 - 5 used two passes - classify into ring id / azimuth arrays
// and block histograms, a scan, then a scatter that read the sweep again: 1.67 ms per 1024 sweeps; the arithmetic and the traffic of this
// kernel alone were measured at 1.04 ms.)  What makes one pass possible:
//   * ring r of a sweep owns a SLAB of `slab` points (slab >= the longest ring k_ring_features accepts), so where a point goes depends only on
//     how many points of its ring came before it, not on how long the other rings turn out to be; the dense concatenation the reference
//     publishes (/velodyne_cloud_2) is made from the slabs only when somebody asks for it (k_dense_cloud);
//   * "how many came before" = rank inside the workgroup's 1024 points (wave ballots, the 16 (round, wave) counts scanned in LDS) + the sum over
//     the blocks in front, by decoupled look-back: every block publishes per ring ONE 8-byte granule {epoch, state, value} - first its own count
//     (state A), then, once it knows what lies in front of it, the inclusive prefix (state P) - and a block walks back over the granules of its
//     predecessors, adding A's until it meets a P.  The halfPassed flip index (:220-223), a running minimum, travels the same way in slot R;
//   * WHICH block a workgroup takes is a ticket drawn when it starts, so a block only ever waits for workgroups that are already running
//     (the same argument as for the ring tickets of k_ring_features).  The launch is sweep-major (blockIdx.x = sweep), so the predecessor of
//     a block was dispatched a whole batch earlier and has normally published long ago; at batch 1 the blocks of the sweep run side by side
//     and the walk is a chain of up to NB / 2 granule reads.
constexpr int kGatherSpinLimit = 1 << 20;                                     // polls of a look-back wait before it gives up (~1 s): a fault must not hang the stream
__device__ __forceinline__ void front_publish(unsigned long long* g, unsigned epoch, unsigned prefix, int value) {
  __hip_atomic_store(g, ((unsigned long long)((epoch << 1) | prefix) << 32) | (unsigned)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exclusive prefix (sum or minimum) over the blocks in front of block `blk`; publishes this block's aggregate and inclusive prefix
template <bool MIN>
__device__ __forceinline__ int front_lookback(unsigned long long* slot0, int blk, unsigned epoch, int local, int* err) {   // slot0: granule of block 0 for this slot
  constexpr int identity = MIN ? 0x7fffffff : 0;
  unsigned long long* mine = slot0 + (long long)blk * kFrontSlots;
  if (blk == 0) { front_publish(mine, epoch, 1u, local); return identity; }
  front_publish(mine, epoch, 0u, local);
  int acc = identity;
  for (int t = blk - 1; t >= 0; --t) {
    unsigned long long g;
    int spins = 0;
    while ((unsigned)((g = __hip_atomic_load(slot0 + (long long)t * kFrontSlots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 33) != (epoch & 0x7fffffffu)) {
      if (++spins > kGatherSpinLimit) { atomicOr(err, kErrInternal); g = (1ull << 32) | (unsigned)identity; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    const int v = (int)(unsigned)g;
    acc = MIN ? (v < acc ? v : acc) : acc + v;
    if ((g >> 32) & 1ull) break;
  }
  front_publish(mine, epoch, 1u, MIN ? (local < acc ? local : acc) : acc + local);
  return acc;
}

// Measured at batch 1024 (A/B on one box): 1.13 ms (1.15 without the eight-waves hint, which costs six spilled registers); block-major launch order
// 1.59 ms (the look-back chains); with the waits compiled out 1.07 ms: the look-back costs 0.06 ms.  Before: k_classify 0.89 + k_ring_offsets 0.02 + k_scatter 0.76.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void k_front(RegArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_cnt[16][kMaxRings];        // [round * 4 + wave][ring]: points of that ring in that wave's round -> offsets inside the ring's slab
  __shared__ int s_half, s_ticket;
  if (tid == 0) { s_ticket = atomicAdd(a.front_ticket + b, 1); s_half = 0x7fffffff; }
  for (int q = tid; q < 16 * kMaxRings; q += 256) (&s_cnt[0][0])[q] = 0;
  __syncthreads();
  const int blk = s_ticket;
  const SeqMeta m = a.meta[b];
  const int n = m.n_in < a.cap ? m.n_in : a.cap;
  if (blk * kBlockPts >= n) return;            // (every later ticket of the sweep returns here too: nobody waits for this block)
  const char* in = a.in + (long long)b * a.seq_stride;
  const float startOri = m.start_ori, endOri = m.end_ori;
  int ring[4], rank[4];
  float4 p[4];
  float ori[4];
  int myhalf = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = blk * kBlockPts + k * 256 + tid;
    ring[k] = -1;
    p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    ori[k] = 0.f;
    if (i < n) {
      const float4 q = load_point(in, i, a.pt_stride);
      if (point_kept(q, a.min_range)) {
        ring[k] = ring_of(q, a.R, a.ring_from_field);
        if (ring[k] >= 0) {
          p[k] = q;
          ori[k] = -atan2f_port(q.y, q.x);                                                  // :208
          float o1 = ori[k];                                                                // branch taken while !halfPassed
          if ((double)o1 < (double)startOri - M_PI / 2) o1 = (float)((double)o1 + 2 * M_PI);          // :211-214
          else if ((double)o1 > (double)startOri + M_PI * 3 / 2) o1 = (float)((double)o1 - 2 * M_PI); // :215-218
          if ((double)(o1 - startOri) > M_PI && i < myhalf) myhalf = i;                      // :220-223
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // rank among the same-ring lanes of this wave (stable: lower lane = earlier point)
    rank[k] = 0;
    unsigned long long todo = __ballot(ring[k] >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int r = __shfl(ring[k], leader, 64);
      const unsigned long long same = __ballot(ring[k] == r);
      if (ring[k] == r) rank[k] = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) s_cnt[k * 4 + wave][r] = __popcll(same);
      todo &= ~same;
    }
  }
  for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_down(myhalf, d, 64); myhalf = o < myhalf ? o : myhalf; }
  if (lane == 0 && myhalf != 0x7fffffff) atomicMin(&s_half, myhalf);
  __syncthreads();
  unsigned long long* lb = a.front_lb + (long long)b * a.NB * kFrontSlots;     // granule (block 0, slot 0) of this sweep
  if (tid < a.R) {                             // offsets of the 16 (round, wave) slots inside the ring's slab, on top of what the blocks in front hold
    int total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) total += s_cnt[q][tid];
    int run = front_lookback<false>(lb + tid, blk, a.epoch, total, &a.meta[b].err);
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = s_cnt[q][tid]; s_cnt[q][tid] = run; run += c; }
  } else if (tid == 192) {                     // (wave 3: the rings occupy at most waves 0 and 1)
    const int before = front_lookback<true>(lb + a.R, blk, a.epoch, s_half, &a.meta[b].err);
    s_half = before < s_half ? before : s_half;
  }
  __syncthreads();
  const int half_idx = s_half;
  float4* slabs = a.slabs + (long long)b * a.R * a.slab;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (ring[k] < 0) continue;
    const int i = blk * kBlockPts + k * 256 + tid;
    const int pos = s_cnt[k * 4 + wave][ring[k]] + rank[k];
    float o = ori[k];
    if (i <= half_idx) {                                                                      // !halfPassed when visited
      if ((double)o < (double)startOri - M_PI / 2) o = (float)((double)o + 2 * M_PI);
      else if ((double)o > (double)startOri + M_PI * 3 / 2) o = (float)((double)o - 2 * M_PI);
    } else {                                                                                  // :225-236
      o = (float)((double)o + 2 * M_PI);
      if ((double)o < (double)endOri - M_PI * 3 / 2) o = (float)((double)o + 2 * M_PI);
      else if ((double)o > (double)endOri + M_PI / 2) o = (float)((double)o - 2 * M_PI);
    }
    const float relTime = (o - startOri) / (endOri - startOri);                               // :238
    const float inten = (float)((double)ring[k] + 0.1 * (double)relTime);                     // :239, scanPeriod 0.1
    if (pos < a.slab) slabs[(long long)ring[k] * a.slab + pos] = make_float4(p[k].x, p[k].y, p[k].z, inten);   // (a ring longer than its slab: kErrRingCap, k_ring_starts)
  }
}

// After the pass: ring lengths = the inclusive prefixes of the sweep's last block; scanStartInd-like dense starts (ringstart, :246-252 in the
// reference's dense numbering), the cloud size, the halfPassed index; the block tickets of the next launch.  One wave per sweep.
__global__ __launch_bounds__(64) void k_ring_starts(RegArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = a.meta[b].n_in < a.cap ? a.meta[b].n_in : a.cap;
  const int nb = (n + kBlockPts - 1) / kBlockPts;
  const unsigned long long* last = a.front_lb + ((long long)b * a.NB + (nb > 0 ? nb - 1 : 0)) * kFrontSlots;
  int run = 0, err = 0;
  for (int base = 0; base < a.R; base += 64) {
    const int r = base + lane;
    int c = 0;
    if (r < a.R && nb > 0) {
      const unsigned long long g = last[r];
      if ((unsigned)(g >> 32) != (((a.epoch & 0x7fffffffu) << 1) | 1u)) err |= kErrInternal; else c = (int)(unsigned)g;
      if (c > a.slab) { c = a.slab; err |= kErrRingCap; }
    }
    const int inc = wave_scan_i32<false>(c);
    if (r < a.R) a.ringstart[b * (a.R + 1) + r] = run + inc - c;
    run += __builtin_amdgcn_readlane(inc, 63);
  }
  if (err) atomicOr(&a.meta[b].err, err);
  if (lane == 0) {
    a.ringstart[b * (a.R + 1) + a.R] = run;
    a.meta[b].n_cloud = run;
    int half = 0x7fffffff;
    if (nb > 0) { const unsigned long long g = last[a.R]; if ((unsigned)(g >> 32) == (((a.epoch & 0x7fffffffu) << 1) | 1u)) half = (int)(unsigned)g; }
    a.meta[b].half_idx = half;
    a.front_ticket[b] = 0;
  }
}

// The reference's dense, ring-by-ring cloud (laserCloud, :246-252) from the slabs: only for the consumers of the FULL cloud (the getters, the
// /velodyne_cloud_2 publisher, laserMapping's full-resolution input) - the feature kernels read the slabs.
__global__ __launch_bounds__(256) void k_dense_cloud(RegArgs a) {
  const int r = blockIdx.x, b = blockIdx.y;
  const int start = a.ringstart[b * (a.R + 1) + r], n = a.ringstart[b * (a.R + 1) + r + 1] - start;
  const float4* src = a.slabs + ((long long)b * a.R + r) * a.slab;
  float4* dst = a.cloud + (long long)b * a.cap + start;
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// -------------------------------------------------------------------------------------------------------
// wave-wide max / min of a 32-bit value: DPP inside the rows of 16, v_readlane across the four rows (wave-uniform result)
template <bool MAX>
__device__ __forceinline__ unsigned wave_extreme_u32(unsigned v, int lane) { (void)lane; return wave_reduce_u32<MAX>(v); }

// ---- output offsets across the rings of a sweep --------------------------------------------------------------------------------
// The four feature clouds are the ring-by-ring concatenation of what the ring workgroups produce (reference
// src/scanRegistration.cpp:304-310,356,407), so a workgroup needs the counts of the rings in front of it.  Every (sweep, ring)
// workgroup publishes its own count per class as ONE 8-byte granule {launch epoch, count}: the value is its own flag, nothing has
// to be reset between launches, and a reader needs no other data of the writer (MI355X_MICROARCH.md, inter-workgroup visibility:
// agent-scope loads of 8-byte granules).  A reader gathers the granules of the rings in front of it with one wave (lane = ring)
// and sums them: no serial ripple from ring to ring.  WHICH ring a workgroup works on is decided when it starts executing: it takes
// the next ticket of its sweep (k_ring_features), so ring r of a sweep is always in the hands of a workgroup that started before the
// one holding ring r + 1 — a workgroup only ever waits for workgroups that are already running, whatever order the dispatcher
// launches them in (round 4 used blockIdx.y and relied on in-order dispatch).  No deadlock: the unfinished workgroup with the smallest
// ticket of a sweep waits for nobody; the wait is still bounded, as a trap for faults.
__device__ __forceinline__ void publish_count(unsigned long long* slot, unsigned epoch, int count) {
  __hip_atomic_store(slot, ((unsigned long long)epoch << 32) | (unsigned)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The wait is bounded: a fault in an earlier ring's workgroup must not hang the stream.  After kGatherSpinLimit polls (~1 s) the lane gives up, the count reads 0 and
// kErrInternal is raised in the sequence's SeqMeta.err (aloam_synchronize reports it).
__device__ __forceinline__ int gather_counts(const unsigned long long* slots, int upto, unsigned epoch, int lane, int* err) {   // whole wave
  int sum = 0;
  for (int base = 0; base < upto; base += 64) {
    const int q = base + lane;
    if (q < upto) {
      unsigned long long g;
      int spins = 0;
      while ((unsigned)((g = __hip_atomic_load(slots + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
        if (++spins > kGatherSpinLimit) { atomicOr(err, kErrInternal); g = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      sum += (int)(unsigned)g;
    }
  }
  sum = wave_scan_i32<false>(sum);
  return __builtin_amdgcn_readlane(sum, 63);
}
// Greedy corner / flat selection of ONE sector by ONE wave, with the whole sector in registers (reference
// src/scanRegistration.cpp:284-390).  std::sort + "walk from the top, skip picked points" is evaluated as an iterative arg-max
// over the still-unpicked points (arg-min for the flat points): same picks in the same order — ties in curvature go to the
// larger index for corners and to the smaller one for flats, exactly what an ascending sort by (curvature, index) gives — no
// sort, no LDS traffic between two picks.  Point `pos` of the sector lives in register pos / 64 of lane pos % 64.
//   init_marks : bit p set = point p (p < 5) of this sector was already marked by picks of the sectors before it
//   returns (through LDS) the picks, their counts and the marks this sector leaves on the five points behind it
//
// The loop body is what k_ring_features spends most of its VALU issue slots on (up to 20 + 4 picks per sector, 6 sectors per
// ring), so it is written for instruction count:
//   * a picked or marked point never comes back (cloudNeighborPicked is shared by the corner and the flat walk), so "dead" is
//     folded into the key itself: the corner walk keeps curvature bits + 1 with 0 = dead (the per-lane maximum is a v_max3 chain,
//     no mask tests), and one subtraction turns that into the key of the flat walk, curvature bits with 0xffffffff = dead;
//   * `(double)c > 0.1` for a non-negative float c is `bits(c) >= bits(0.1f)` (0.1f is the float above 0.1, the float below it is
//     below 0.1) and `bits(c) <= bits(inf)` for the NaNs; likewise `(double)c < 0.1` is `bits(c) < bits(0.1f)`: scalar integer
//     compares on the wave-uniform extreme;
//   * the register of the extreme inside its lane is computed per lane next to the DPP ladder (it fills the ladder's wait states);
//   * a pick at register kr of lane f can only touch registers kr - 1, kr, kr + 1, and kr is wave-uniform: the keys sit in one
//     register tuple that is indexed through the VGPR index mode (s_set_gpr_idx_on), one compare and one select per touched register
//     (the neighbours only when the reach leaves lanes 0 .. 63, one pick of six);
//   * the picks collect in one VGPR (lane = pick number) and go to LDS once.
// Measured (A/B on one box, batch 1024, profiles/r03_ab_ring_features.md): ~58 VALU + ~55 SALU per pick before, ~32 + ~36 now; together
// with the bit-mask reach and the packed curvature tiles below the kernel went from 2.95 to 2.43 ms, SQ_INSTS_VALU from 1322 M to 981 M.
constexpr unsigned kCurvThresholdBits = 0x3DCCCCCDu;                          // bits(0.1f); tests/test_selection_model.py checks the claim above
constexpr unsigned kInfBits = 0x7f800000u;
template <int K6>
__device__ __forceinline__ void pick_sector(int j, unsigned init_marks, int L, int lane, const float* curv_l, const unsigned char* flags,
                                            short* s_pick, int* s_misc) {
  constexpr int kSlots = kSharpPerSector + kLessSharpPerSector + kFlatPerSector;
  constexpr int NRP = (K6 + 3) / 4;                                           // reach bytes packed 4 x 8 bit per register
  const int sp = (L * j) / 6, len = (L * (j + 1)) / 6 - sp;
  const int first = sp + 5;                                                   // local index of the sector's first point
  typedef unsigned keys_t __attribute__((ext_vector_type(16)));               // > 8 elements: cc[kr] with a wave-uniform kr is one indexed v_mov
  static_assert(K6 <= 16, "sector registers");
  keys_t cc;                                                                  // corner walk: curvature bits + 1, 0 = dead
  unsigned rp[NRP];
#pragma unroll
  for (int q = 0; q < NRP; ++q) rp[q] = 0;
#pragma unroll
  for (int r = 0; r < K6; ++r) {
    const int pos = r * 64 + lane;
    cc[r] = 0u;
    if (pos < len) {
      const int i = first + pos;
      rp[r / 4] |= (unsigned)(flags[i] >> 2) << (8 * (r % 4));
      if (!(pos < 5 && ((init_marks >> pos) & 1u))) cc[r] = __float_as_uint(curv_l[i]) + 1u;
    }
  }
  unsigned spill = 0;
  // the pick at register kr of lane f (both wave-uniform) marks itself and its reach dead; returns its position in the sector.
  // The scalar side is kept as short as the vector side (the SALU issues one instruction per SIMD slot, like the VALU): the common
  // case — reach inside lanes 0 .. 63, pick not within five points of the sector end — costs two compares and two branches not taken.
  auto kill = [&](int kr, int f, const unsigned dead) {
    unsigned rb;
    if constexpr (NRP <= 2) {
      const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)rp[0], f), r1 = (unsigned)__builtin_amdgcn_readlane((int)rp[NRP - 1], f);
      rb = (unsigned)((((unsigned long long)r1 << 32) | r0) >> (8 * kr));
    } else {
      unsigned rpk = (unsigned)__builtin_amdgcn_readlane((int)rp[0], f);
#pragma unroll
      for (int q = 1; q < NRP; ++q) { const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)rp[q], f); if (q == (kr >> 2)) rpk = x; }
      rb = rpk >> (8 * (kr & 3));
    }
    const int fw = (int)(rb & 7u), bk = (int)((rb >> 3) & 7u);
    const int P = kr * 64 + f;
    const int lo = f - bk, w = fw + bk;                                       // lanes lo .. lo + w of register kr (may stick out of 0 .. 63)
    {
      const bool in = (unsigned)(lane - lo) <= (unsigned)w;
      const unsigned old = cc[kr];                                            // VGPR index mode (kr is wave-uniform)
      cc[kr] = in ? dead : old;
    }
    if ((unsigned)lo > (unsigned)(63 - w)) {                                  // lo < 0 or lo + w > 63 (one pick of six): a neighbouring register too
      asm volatile("");                                                       // the barriers keep these scalar branches (if-converted they become selects over the whole tuple)
      if (lo < 0 && kr > 0) { asm volatile(""); const bool in = lane >= 64 + lo; const unsigned old = cc[kr - 1]; cc[kr - 1] = in ? dead : old; }
      if (lo >= 0 && kr + 1 < K6) { asm volatile(""); const bool in = lane <= lo + w - 64; const unsigned old = cc[kr + 1]; cc[kr + 1] = in ? dead : old; }
    }
    if (P + 5 >= len) {                                                       // marks behind the sector: positions len .. P + fw
      asm volatile("");
      const int e = P + fw - (len - 1);
      if (e > 0) spill |= (1u << e) - 1u;
    }
    return P;
  };
  // corners: largest curvature first (:291-344).  The reference's walk breaks at the 21st candidate before marking it (:312-315):
  // nothing of that iteration survives, so the loop simply ends after the 20th pick.
  int picks = 0;                                                              // lane q: local index of pick q
  int count = 0;
  for (; count < kLessSharpPerSector; ++count) {
    unsigned m = cc[0];
#pragma unroll
    for (int r = 1; r < K6; ++r) m = cc[r] > m ? cc[r] : m;
    int krl = 0;
#pragma unroll
    for (int r = 1; r < K6; ++r) krl = cc[r] == m ? r : krl;                  // the largest register holding the lane's maximum
    const unsigned cmax = wave_reduce_u32<true>(m);
    if (!(cmax - 1u >= kCurvThresholdBits && cmax - 1u <= kInfBits)) break;   // nothing alive (0 wraps) or nothing above 0.1 left
    const unsigned long long tie = __ballot(m == cmax);
    int f = __builtin_ctzll(tie);                                             // tie != 0: some lane holds the extreme
    int kr = __builtin_amdgcn_readlane(krl, f);
    if (__popcll(tie) != 1) {                                                 // equal curvatures in several lanes: the largest index
      asm volatile("");
      const unsigned w = wave_reduce_u32<true>(m == cmax ? (unsigned)(krl * 64 + lane) : 0u);
      f = (int)(w & 63u); kr = (int)(w >> 6);
    }
    const int P = kill(kr, f, 0u);
    picks = lane == count ? first + P : picks;
  }
  const int ncorner = count;
  if (lane < ncorner) s_pick[j * kSlots + kSharpPerSector + lane] = (short)picks;
  // flats: smallest curvature first (:346-390)
#pragma unroll
  for (int r = 0; r < K6; ++r) cc[r] -= 1u;                                   // curvature bits, 0xffffffff = dead
  count = 0;
  while (true) {
    unsigned m = cc[0];
#pragma unroll
    for (int r = 1; r < K6; ++r) m = cc[r] < m ? cc[r] : m;
    int krl = K6 - 1;
#pragma unroll
    for (int r = K6 - 2; r >= 0; --r) krl = cc[r] == m ? r : krl;             // the smallest register holding the lane's minimum
    const unsigned cmin = wave_reduce_u32<false>(m);
    if (!(cmin < kCurvThresholdBits)) break;                                  // nothing alive (0xffffffff) or nothing below 0.1 left
    const unsigned long long tie = __ballot(m == cmin);
    int f = __builtin_ctzll(tie);                                             // tie != 0: some lane holds the extreme
    int kr = __builtin_amdgcn_readlane(krl, f);
    if (__popcll(tie) != 1) {                                                 // ... the smallest index
      asm volatile("");
      const unsigned w = wave_reduce_u32<false>(m == cmin ? (unsigned)(krl * 64 + lane) : 0xffffffffu);
      f = (int)(w & 63u); kr = (int)(w >> 6);
    }
    picks = lane == count ? first + kr * 64 + f : picks;
    ++count;
    if (count >= kFlatPerSector) break;                                       // 4th: break before marking (:359-362)
    kill(kr, f, 0xffffffffu);
  }
  if (lane < count) s_pick[j * kSlots + kSharpPerSector + kLessSharpPerSector + lane] = (short)picks;
  if (lane == 0) { s_misc[1 + j] = ncorner | (count << 8); s_misc[8 + j] = (int)spill; }
}


// markers in the generated code for tools/isa_phases.py (comments only: no instruction is emitted).  Debug builds (-DALOAM_PHASE_CLOCK): wave 0 of every
// workgroup also adds the shader clock at the marker to a per-marker sum, so the mean time BETWEEN two markers over a run is a difference of two sums
// divided by the count (tools/ab_check.py prints it); the product build emits nothing.
#ifdef ALOAM_PHASE_CLOCK
__device__ unsigned long long g_phase_clock[2][32];
constexpr int phase_slot(const char* s) { unsigned h = 0; while (*s) h = h * 33u + (unsigned char)*s++; return (int)(h % 32u); }   // collision-free over the markers of this file (tools/ab_check.py checks)
#define ALOAM_PHASE(name) do { asm volatile("; ##PHASE " name); if (threadIdx.x == 0) { constexpr int slot_ = phase_slot(name); \
    atomicAdd(&g_phase_clock[0][slot_], (unsigned long long)__builtin_readcyclecounter()); atomicAdd(&g_phase_clock[1][slot_], 1ull); } } while (0)
#else
#define ALOAM_PHASE(name) asm volatile("; ##PHASE " name)
#endif
// cloudLabel (2 sharp, 1 less sharp, 0, -1 flat) shares the per-point flag byte with the reach of the neighbour suppression: bits 0-1 hold the
// label code (2, 1, 0, 3 = -1), bits 2-7 the reach during the selection and, afterwards, bit 2 the "continues the run of its predecessor" mark of
// the voxel filter.  One byte array less per ring is what lets EIGHT ring workgroups share a CU's LDS (20.2 KB each) instead of seven.
__device__ __forceinline__ bool label_is_member(unsigned char f) { return ((f + 1u) & 2u) == 0u; }        // label <= 0: codes 0 and 3
__device__ __forceinline__ int label_of(unsigned char f) { const int c = f & 3; return c == 3 ? -1 : c; }
// ---- second half of pcl::VoxelGrid for one ring: run heads -> sort of the run keys -> voxel heads ->
// centroids in input order.  K = key type (voxel index << SHIFT | first element of the run), see the call site.
template <int NPAD, typename K, int SHIFT>
__device__ __forceinline__ int voxel_runs_tail(unsigned char* smem, unsigned char* flags, int* s_scan, int* s_misc,
                                               const float4* cloud, float4* out, int L, int tid, int lane, int wave,
                                               unsigned long long* lb_lf, int ring, int nrings, unsigned epoch, int* err) {
  const unsigned* vis = reinterpret_cast<const unsigned*>(smem);               // region A: voxel index per element [NPAD] ...
  K* rkeys = reinterpret_cast<K*>(smem);                                      // ... replaced by the run keys once the heads are known
  constexpr unsigned kEMask = SHIFT >= 32 ? 0xffffffffu : ((1u << (SHIFT & 31)) - 1u);
  // run heads: a member whose predecessor is no member or sits in another voxel (members have label <= 0, never 0xffffffff).
  // Elements are taken 256 at a time (element = it * 256 + tid: conflict-free LDS reads); the exclusive rank of a head in
  // element order comes from wave ballots + a 4 x EIT table of wave counts — two barriers instead of a 16-barrier scan.
  constexpr int EIT = NPAD / 256;
  unsigned hmask = 0;                                                        // bit it: element it * 256 + tid starts a run
  int hrank[EIT];
  unsigned myvi[EIT];
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int e = it * 256 + tid;
    bool h = false;
    myvi[it] = 0xffffffffu;
    if (e < L) {
      const unsigned vi = vis[e];
      const unsigned char fe = flags[e + 5];
      const bool member = label_is_member(fe);
      h = member && (e == 0 || !label_is_member(flags[e + 4]) || vis[e - 1] != vi);
      myvi[it] = vi;
      flags[e + 5] = (unsigned char)((fe & 3u) | (member && !h ? 4u : 0u));   // bit 2: the element continues the run of its predecessor (label bits stay: the
                                                                             // neighbour thread may still be reading them)
    }
    const unsigned long long m = __ballot(h);
    hrank[it] = __popcll(m & ((1ull << lane) - 1ull));
    if (h) hmask |= 1u << it;
    if (lane == 0) s_scan[it * 4 + wave] = __popcll(m);
  }
  __syncthreads();
  // exclusive prefix of the 4 x EIT (round, wave) counts by one wave (DPP scan), instead of every thread summing the whole table
  static_assert(4 * EIT <= 64, "one wave scans the table");
  if (wave == 0) {
    const int c = lane < 4 * EIT ? s_scan[lane] : 0;
    const int inc = wave_scan_i32<false>(c);
    if (lane < 4 * EIT) s_scan[lane] = inc - c;
    if (lane == 63) s_scan[64] = inc;
  }
  __syncthreads();
  const int n_runs = s_scan[64];
#pragma unroll
  for (int it = 0; it < EIT; ++it) hrank[it] += s_scan[it * 4 + wave];
#pragma unroll
  for (int it = 0; it < EIT; ++it)
    if ((hmask >> it) & 1u) { const int e = it * 256 + tid; rkeys[hrank[it]] = (K)(((K)myvi[it] << SHIFT) | (K)e); }   // vis is dead: every thread read its share before the barrier
  __syncthreads();
  ALOAM_PHASE("after_run_heads");   // run heads + keys
  bitonic_sort_keys<K>(rkeys, n_runs, tid);      // (a radix sort of the run keys - 9 instead of ~40 barriers, half the VALU work - was measured three times and lost, last at eight waves per SIMD: 2.25 against 2.15 ms: HISTORY.md)

  ALOAM_PHASE("after_sort");   // sort
  // voxel heads among the sorted runs -> output rank (ascending voxel index), centroid = f32 sums in input order / count
  int n_vox = 0;
  unsigned vmask = 0;
  int vrank[EIT];
  __syncthreads();                                                           // s_scan is reused
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int p = it * 256 + tid;
    const bool h = p < n_runs && (p == 0 || (unsigned)(rkeys[p - 1] >> SHIFT) != (unsigned)(rkeys[p] >> SHIFT));
    const unsigned long long m = __ballot(h);
    vrank[it] = __popcll(m & ((1ull << lane) - 1ull));
    if (h) vmask |= 1u << it;
    if (lane == 0) s_scan[it * 4 + wave] = __popcll(m);
  }
  __syncthreads();
  if (wave == 0) {
    const int c = lane < 4 * EIT ? s_scan[lane] : 0;
    const int inc = wave_scan_i32<false>(c);
    if (lane < 4 * EIT) s_scan[lane] = inc - c;
    if (lane == 63) { s_scan[64] = inc; publish_count(lb_lf + ring, epoch, inc); }   // number of occupied voxels = less-flat points of this ring
  }
  __syncthreads();
  n_vox = s_scan[64];
#pragma unroll
  for (int it = 0; it < EIT; ++it) vrank[it] += s_scan[it * 4 + wave];
  // offsets of this ring in the four clouds = what the rings in front produced.  The launch orders the workgroups ring-major
  // over the sweeps of the batch (blockIdx.x = sweep), so the rings in front of this one were dispatched a whole batch row
  // earlier and have normally published long ago: the gather does not wait.
  if (wave == 0) {
    const unsigned long long* lb0 = lb_lf - 3 * nrings;
    const int b0 = gather_counts(lb0 + 0 * nrings, ring, epoch, lane, err), b1 = gather_counts(lb0 + 1 * nrings, ring, epoch, lane, err);
    const int b2 = gather_counts(lb0 + 2 * nrings, ring, epoch, lane, err), b3 = gather_counts(lb_lf, ring, epoch, lane, err);
    if (lane == 0) { s_misc[40] = b0; s_misc[41] = b1; s_misc[42] = b2; s_misc[43] = b3; }
  }
  __syncthreads();
  ALOAM_PHASE("after_vox_heads_gather");   // voxel heads + gather of the counts in front
  out += s_misc[43];
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    if (!((vmask >> it) & 1u)) continue;
    const int p = it * 256 + tid;
    const unsigned vi = (unsigned)(rkeys[p] >> SHIFT);
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    int cnt = 0;
    for (int q = p; q < n_runs; ++q) {                                       // the runs of this voxel, in element order
      const K kq = rkeys[q];
      if ((unsigned)(kq >> SHIFT) != vi) break;
      int e = (int)((unsigned)kq & kEMask);
      // The additions are a sequential chain (f32, input order) but the loads are not: a run is fetched four elements at a time, speculatively (the
      // elements behind a run's end are loaded and dropped; L >= 6 and e + 8 <= L + 7 < n keep the addresses inside the ring), so a run of k elements
      // costs ceil(k / 4) memory round trips instead of k.  The ring left the L2 while its selection ran: these are HBM / MALL latencies.
      bool more;
      do {
        const float4 p0 = cloud[e + 5], p1 = cloud[e + 6], p2 = cloud[e + 7], p3 = cloud[e + 8];
        const bool c1 = e + 1 < L && (flags[e + 6] & 4), c2 = e + 2 < L && (flags[e + 7] & 4), c3 = e + 3 < L && (flags[e + 8] & 4);
        more = e + 4 < L && (flags[e + 9] & 4);                               // the run stops at the next head or non-member
        sx += p0.x; sy += p0.y; sz += p0.z; si += p0.w; ++cnt;
        if (!c1) break;
        sx += p1.x; sy += p1.y; sz += p1.z; si += p1.w; ++cnt;
        if (!c2) break;
        sx += p2.x; sy += p2.y; sz += p2.z; si += p2.w; ++cnt;
        if (!c3) break;
        sx += p3.x; sy += p3.y; sz += p3.z; si += p3.w; ++cnt;
        e += 4;
      } while (more);
    }
    const float fc = (float)cnt;
    out[vrank[it]] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
  }
  ALOAM_PHASE("after_centroids");   // centroids
  return n_vox;
}

// Eight ring workgroups share a CU's LDS (20.2 KB each), i.e. eight waves per SIMD - if the registers allow it, and that includes the SCALAR ones: a gfx9 SIMD
// has 800 SGPRs, a wave with more than 100 of them (the kernel used 106) is one of seven.  Telling the allocator the target (amdgpu_waves_per_eu) costs no spill
// (62 VGPRs / 78 SGPRs) and took the kernel from 2.43 to 2.07 ms at batch 1024; fetching the ring further ahead (2, 3, 5, 8 chunks in registers) on top of it
// changed nothing (2.06 - 2.08 ms).  The <4096> instance is limited to four workgroups by its LDS.
template <int NPAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NPAD <= 2048 ? 8 : 4))) void k_ring_features(RegArgs a, float leaf) {
  constexpr int MAXN = NPAD + 11;
  constexpr int ITEMS = (MAXN + 255) / 256;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // ring-major over the batch, see voxel_runs_tail
  // the ring of this workgroup = the next ticket of its sweep (see "output offsets across the rings of a sweep" above)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ticket[];
  int* s_ticket = reinterpret_cast<int*>(smem_ticket);                       // first word of the dynamic LDS, free again after the barrier
  typedef __attribute__((address_space(1))) int global_int;                 // a GLOBAL atomic (the generic pointer of the argument struct would make it a flat one)
  if (tid == 0) *s_ticket = __hip_atomic_fetch_add((global_int*)(a.ring_ticket + b), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int r_ = __builtin_amdgcn_readfirstlane(*s_ticket);
  asm volatile("" : "+s"(r_));                                               // a scalar register from here on, like blockIdx.y was
  const int r = r_;
  __syncthreads();
  if (r >= a.R) {                                                           // more tickets than rings: the counter was not reset (k_cloud_sizes of an earlier launch never ran)
    if (tid == 0) atomicOr(&a.meta[b].err, kErrInternal);
    return;
  }
  const int start = a.ringstart[b * (a.R + 1) + r];
  const int n = a.ringstart[b * (a.R + 1) + r + 1] - start;
  unsigned long long* lb = a.lookback + (long long)b * 4 * a.R;             // [class][ring] count granules of this sweep
  if (n - 11 < 6 || n > MAXN) {                                             // :279: nothing is selected from this ring
    if (n - 11 >= 6 && tid == 0) atomicOr(&a.meta[b].err, kErrRingCap);
    if (tid < 4) publish_count(lb + tid * a.R + r, a.epoch, 0);
    return;
  }

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ALOAM_PHASE("ticket_taken");
  // region A (aliased over time): two 266-point xyz tiles during the curvature pass, then the curvature per point, then the voxel
  // index per element, then the run keys.  Keeping it at 8 * NPAD bytes is what lets seven workgroups share a CU's LDS.
  constexpr int A_BYTES = 8 * NPAD + 128;                                     // + room for the packed voxel cells behind the curvatures
  constexpr int TW = 256 + 10;                                                // a chunk of 256 points + 5 on either side
  static_assert(2 * 3 * TW * 4 <= A_BYTES && 4 * MAXN <= A_BYTES, "region A holds the curvature tiles and the curvature array");
  // a tile keeps (x, y) as pairs and z apart: the 11-point sums run as packed f32 adds on the pairs a 64-bit LDS read delivers
  // (one v_pk_add_f32 for x and y, the same IEEE additions in the same order) plus scalar adds for z
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 (*txy)[TW] = reinterpret_cast<f2 (*)[TW]>(smem);
  float (*tz)[TW] = reinterpret_cast<float (*)[TW]>(smem + 2 * TW * 8);
  // all scratch lives in the dynamic region so its base stays 16-byte aligned (no static __shared__ in front)
  constexpr int FLAG_BYTES = (MAXN + 15) & ~15;
  unsigned char* flags = smem + ((A_BYTES + 15) & ~15);
  int* s_scan = reinterpret_cast<int*>(smem + ((A_BYTES + 15) & ~15) + FLAG_BYTES);
  float (*s_red)[4] = reinterpret_cast<float (*)[4]>(s_scan + 256);
  int* s_misc = reinterpret_cast<int*>(s_scan + 256 + 24);
  // picks of the 6 sectors (local indices), staged in LDS so that the serial picking loop issues no global store:
  // [j][0..1] sharp, [j][2..21] less sharp, [j][22..25] flat
  constexpr int kSlots = kSharpPerSector + kLessSharpPerSector + kFlatPerSector;
  short* s_pick = reinterpret_cast<short*>(s_misc + 48);      // s_misc: [0] scratch, [1..6] pick counts, [8..13] spill marks, [16..], [24..], [32..] sector offsets

  // 0.2 m voxel cell of every point, packed 11 + 11 + 10 bits (+-204 m, +-102 m in z), written while the point is in the
  // curvature tile: the voxel filter of the less-flat points then needs no second and third pass over the ring in global memory
  // (its bounding box in cells and the voxel indices are integer work on this array).  A point outside that range (s_misc[44])
  // or a box of more than 2^31 cells sends the ring down the float path, which is pcl::VoxelGrid's arithmetic as written.
  constexpr int CELLS_OFF = (4 * MAXN + 15) & ~15;
  static_assert(CELLS_OFF + 4 * MAXN <= A_BYTES, "the packed cells sit behind the curvature array in region A");
  unsigned* cells = reinterpret_cast<unsigned*>(smem + CELLS_OFF);
  const float inv = 1.0f / leaf;
  // "the step s -> s + 1 is longer than the 0.05 threshold" as ONE BIT per step (bit s + 64 of gapw; steps that do not exist, s < 0
  // and s >= n - 1, read 1): a wave ballots the 64 steps it has just computed, and the reach of the neighbour suppression around a
  // point is two count-trailing-zeros on a 10-bit window of this array instead of up to ten dependent LDS byte reads.
  unsigned long long* gapw = reinterpret_cast<unsigned long long*>(s_scan);
  static_assert((ITEMS * 4 + 2) * 8 <= 256 * 4, "the gap bits fit the scan scratch");
  if (tid == 0) { s_misc[44] = 0; gapw[0] = ~0ull; gapw[ITEMS * 4 + 1] = ~0ull; }
  const float4* cloud = a.slabs + ((long long)b * a.R + r) * a.slab;       // the ring's slab (k_front); `start` is its place in the dense numbering
  // ---- curvature (:256-266) + gap flags, kept in registers until the tiles are retired.  The ring goes through LDS in chunks
  // of 256 points (thread tid owns point it * 256 + tid = tile column tid + 5); the next chunk is fetched while this one is used.
  const int L = n - 11;                          // E - S
  float cv[ITEMS];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pa = tid - 5 >= 0 && tid - 5 < n ? cloud[tid - 5] : zero4, pb = tid < 10 && tid + 251 < n ? cloud[tid + 251] : zero4;
  // byte offsets of this thread's column in the two (x, y) tiles and the two z tiles, opaque to the compiler: every LDS read of the
  // stencil is then base register + small immediate (folded into one constant, the second tile lies beyond the 8-bit offset range
  // of ds_read2 and costs an address add per pair of reads)
  unsigned oxy[2] = {(unsigned)tid * 8u, (unsigned)tid * 8u + TW * 8u}, oz[2] = {2u * TW * 8u + (unsigned)tid * 4u, 2u * TW * 8u + TW * 4u + (unsigned)tid * 4u};
  asm volatile("" : "+v"(oxy[1]), "+v"(oz[0]), "+v"(oz[1]));
  txy[0][tid] = f2{pa.x, pa.y}; tz[0][tid] = pa.z;
  if (tid < 10) { txy[0][tid + 256] = f2{pb.x, pb.y}; tz[0][tid + 256] = pb.z; }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int i = tid + it * 256;
    const bool more = (it + 1) * 256 < n;                                      // uniform
    if (more) {
      const int ia = i + 251, ib = i + 507;                                    // columns tid and tid + 256 of the next chunk
      pa = ia < n ? cloud[ia] : zero4;
      pb = tid < 10 && ib < n ? cloud[ib] : zero4;
    }
    cv[it] = 0.f;
    bool gapf = true;
    if (i < n) {
      const f2* q = reinterpret_cast<const f2*>(smem + oxy[it & 1]) + 5;
      const float* zs = reinterpret_cast<const float*>(smem + oz[it & 1]) + 5;
      const f2 q0 = q[0];
      const float z0 = zs[0];
      if (i < n - 1) {
        const f2 d = q[1] - q0, dd = d * d;
        const float dz = zs[1] - z0;
        gapf = (double)(dd.x + dd.y + dz * dz) > 0.05;                         // :324 etc.
      }
      if (i >= 5 && i < n - 5) {
        f2 sxy = q[-5] + q[-4]; sxy = sxy + q[-3]; sxy = sxy + q[-2]; sxy = sxy + q[-1]; sxy = sxy - 10.f * q0;
        sxy = sxy + q[1]; sxy = sxy + q[2]; sxy = sxy + q[3]; sxy = sxy + q[4]; sxy = sxy + q[5];
        const float dZ = zs[-5] + zs[-4] + zs[-3] + zs[-2] + zs[-1] - 10 * z0 + zs[1] + zs[2] + zs[3] + zs[4] + zs[5];
        const f2 ss = sxy * sxy;
        cv[it] = ss.x + ss.y + dZ * dZ;
        if (a.store_debug) a.curv[(long long)b * a.cap + start + i] = cv[it];
      }
      const f2 cxy = q0 * inv;
      const float fx = floorf(cxy.x), fy = floorf(cxy.y), fz = floorf(z0 * inv);
      const bool okc = fabsf(fx) < 1024.f && fabsf(fy) < 1024.f && fabsf(fz) < 512.f;
      if (!okc) s_misc[44] = 1;
      cells[i] = okc ? (unsigned)((int)fx + 1024) | ((unsigned)((int)fy + 1024) << 11) | ((unsigned)((int)fz + 512) << 22) : 0u;
    }
    {
      const unsigned long long gb = __ballot(gapf);                            // steps it * 256 + wave * 64 .. + 63
      if (lane == 0) gapw[1 + it * 4 + wave] = gb;
    }
    if (more) {
      const int nb = (it + 1) & 1;
      txy[nb][tid] = f2{pa.x, pa.y}; tz[nb][tid] = pa.z;
      if (tid < 10) { txy[nb][tid + 256] = f2{pb.x, pb.y}; tz[nb][tid + 256] = pb.z; }
    }
    __syncthreads();
  }

  ALOAM_PHASE("after_curvature");   // curvature
  // reach of the neighbour suppression around every point: a pick of i marks i+1 .. i+fw and i-1 .. i-bk (runs of consecutive
  // gap-free steps, at most 5; :316-341).  Packed into the flag byte: bits 2-4 fw, bits 5-7 bk.  The curvature tiles are retired
  // (the loop above ends with a barrier), so region A takes the curvature per local point in the same pass.
  float* curv_l = reinterpret_cast<float*>(smem);
  {
    const unsigned* gw32 = reinterpret_cast<const unsigned*>(gapw);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int i = tid + it * 256;
      if (i < n) {
        const unsigned bit = (unsigned)i + 59u;                                // bit of step i - 5
        const unsigned win = __builtin_amdgcn_alignbit(gw32[(bit >> 5) + 1], gw32[bit >> 5], bit & 31u);   // bits 0 .. 9: steps i - 5 .. i + 4
        const int fw = __builtin_ctz((win >> 5) | 32u);                        // steps i, i + 1, ...
        const int bk = __builtin_ctz(__builtin_bitreverse32(win << 27) | 32u); // steps i - 1, i - 2, ...
        curv_l[i] = cv[it];
        flags[i] = (unsigned char)((fw << 2) | (bk << 5));
      }
    }
  }
  __syncthreads();

  ALOAM_PHASE("after_reach");   // reach + curvature array
  // ---- corner / flat selection (:284-390): every sector by its own wave, speculatively without the marks the previous
  // sectors leave on its first five points; those marks only matter if the sector picked one of the marked points, which
  // the second pass detects (and then redoes that sector with the marks) in sector order
  constexpr int K6 = (NPAD / 6 + 2 + 63) / 64;
#pragma unroll 1
  for (int j = __builtin_amdgcn_readfirstlane(wave); j < kSectors; j += 4) pick_sector<K6>(j, 0u, L, lane, curv_l, flags, s_pick, s_misc);   // j in an SGPR: sector bounds, pick positions and spill marks are scalar work
  __syncthreads();
  ALOAM_PHASE("after_select1");   // first-pass selection
  if (wave == 0) {
    unsigned carry = (unsigned)s_misc[8];                                    // marks on the (up to 5) points after sector 0
#pragma unroll 1
    for (int j = 1; j < kSectors; ++j) {
      const int sp = (L * j) / 6, len = (L * (j + 1)) / 6 - sp;
      const unsigned m = len >= 5 ? carry : (carry & ((1u << len) - 1u));    // marks that fall inside this sector
      if (m) {
        const int nc = s_misc[1 + j] & 0xff, nfl = s_misc[1 + j] >> 8;
        bool hit = false;
        if (lane < kSlots) {
          const int slot = lane;
          const bool used = slot >= kSharpPerSector && (slot < kSharpPerSector + kLessSharpPerSector ? slot - kSharpPerSector < nc : slot - kSharpPerSector - kLessSharpPerSector < nfl);
          if (used) { const int pos = s_pick[j * kSlots + slot] - 5 - sp; hit = pos < 5 && ((m >> pos) & 1u); }
        }
        if (__ballot(hit)) pick_sector<K6>(j, m, L, lane, curv_l, flags, s_pick, s_misc);
      }
      // a sector shorter than 5 points passes the rest of the incoming marks on to the sectors behind it
      carry = (len >= 5 ? 0u : (carry >> len)) | (unsigned)s_misc[8 + j];
    }
  }
  __syncthreads();
  // labels of the picked points (cloudLabel: 2 sharp, 1 less sharp, -1 flat)
  if (tid < kSectors * kSlots) {
    const int j = tid / kSlots, slot = tid % kSlots;
    const int nc = s_misc[1 + j] & 0xff, nfl = s_misc[1 + j] >> 8;
    if (slot >= kSharpPerSector) {
      if (slot < kSharpPerSector + kLessSharpPerSector) { const int q = slot - kSharpPerSector; if (q < nc) flags[s_pick[j * kSlots + slot]] = q < kSharpPerSector ? 2 : 1; }
      else { const int q = slot - kSharpPerSector - kLessSharpPerSector; if (q < nfl) flags[s_pick[j * kSlots + slot]] = 3; }
    }
  }
  __syncthreads();
  // ---- counts of this ring per class -> published right away (no waiting); the picks themselves are written at the very end,
  // together with the less-flat centroids, once the counts of the rings in front are in (they are dispatched earlier and are
  // about to finish by then)
  if (wave == 0) {
    int nc = 0, nf = 0;
    if (lane < kSectors) { nc = s_misc[1 + lane] & 0xff; nf = s_misc[1 + lane] >> 8; }
    const int ns = nc < kSharpPerSector ? nc : kSharpPerSector;
    const int is = wave_scan_i32<false>(ns), il = wave_scan_i32<false>(nc), ifl = wave_scan_i32<false>(nf);
    if (lane < kSectors) { s_misc[16 + lane] = is - ns; s_misc[24 + lane] = il - nc; s_misc[32 + lane] = ifl - nf; }   // exclusive over the sectors
    if (lane == 0) {
      publish_count(lb + 0 * a.R + r, a.epoch, __builtin_amdgcn_readlane(is, 63));
      publish_count(lb + 1 * a.R + r, a.epoch, __builtin_amdgcn_readlane(il, 63));
      publish_count(lb + 2 * a.R + r, a.epoch, __builtin_amdgcn_readlane(ifl, 63));
    }
  }

  ALOAM_PHASE("after_redo_labels_counts");   // redo, labels, counts
  // ---- labels out (parity tests only) and less-flat membership: local 5 .. n-7 with label <= 0 (:392-398)
  if (a.store_debug) for (int i = tid; i < n; i += 256) a.label[(long long)b * a.cap + start + i] = (int8_t)label_of(flags[i]);

  // ---- pcl::VoxelGrid (leaf 0.2) over the less-flat points of this ring (:401-405)
  // Points follow the ring, so consecutive less-flat points mostly share a voxel: the sort works on RUNS of consecutive
  // same-voxel members, keyed (voxel index, first element), typically a third of the points.  Runs of one voxel end up adjacent
  // and in ascending element order, i.e. the members of a voxel are still summed in input order.
  unsigned* vis = reinterpret_cast<unsigned*>(smem);                          // region A: voxel index per element [NPAD], later the run keys
  static_assert(8 * NPAD <= A_BYTES, "region A holds the voxel indices, then the run keys");
  bool overflow = false;
  long long cells_in_box = 0;                                                // every voxel index is below this
  bool voxels_done = false;
  int (*s_redi)[4] = reinterpret_cast<int (*)[4]>(s_red);
  if (s_misc[44] == 0) {
    // integer path: min_b = floor(min * inv) = min of floor(p * inv) (floor is monotone), likewise max_b; PCL's own overflow
    // guard multiplies int((max - min) * inv) + 1 <= div_b + 1 per axis, so a product of (div_b + 1) below 2^31 settles it
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int e = tid; e < L; e += 256) {
      const int i = e + 5;
      if (label_is_member(flags[i])) {
        const unsigned c = cells[i];
        const int cx = (int)(c & 2047u), cy = (int)((c >> 11) & 2047u), cz = (int)(c >> 22);
        mn[0] = min(mn[0], cx); mx[0] = max(mx[0], cx); mn[1] = min(mn[1], cy); mx[1] = max(mx[1], cy); mn[2] = min(mn[2], cz); mx[2] = max(mx[2], cz);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {                                            // DPP ladders (signed order = unsigned order of x ^ 0x80000000)
      mn[q] = (int)(wave_reduce_u32<false>((unsigned)mn[q] ^ 0x80000000u) ^ 0x80000000u);
      mx[q] = (int)(wave_reduce_u32<true>((unsigned)mx[q] ^ 0x80000000u) ^ 0x80000000u);
      if (lane == 0) { s_redi[q][wave] = mn[q]; s_redi[3 + q][wave] = mx[q]; }
    }
    __syncthreads();
    int minc[3], divc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      minc[q] = min(min(s_redi[q][0], s_redi[q][1]), min(s_redi[q][2], s_redi[q][3]));
      divc[q] = max(max(s_redi[3 + q][0], s_redi[3 + q][1]), max(s_redi[3 + q][2], s_redi[3 + q][3])) - minc[q] + 1;
    }
    __syncthreads();                                                         // s_red may be needed by the float path below
    if (divc[0] > 0 && (long long)(divc[0] + 1) * (divc[1] + 1) * (divc[2] + 1) <= 2147483647ll) {
      cells_in_box = (long long)divc[0] * divc[1] * divc[2];
      for (int e = tid; e < L; e += 256) {
        const int i = e + 5;
        unsigned vi = 0xffffffffu;                                            // not a member (corner-labelled)
        if (label_is_member(flags[i])) {
          const unsigned c = cells[i];
          const int i0 = (int)(c & 2047u) - minc[0], i1 = (int)((c >> 11) & 2047u) - minc[1], i2 = (int)(c >> 22) - minc[2];
          vi = (unsigned)(i0 + i1 * divc[0] + i2 * divc[0] * divc[1]);
        }
        vis[e] = vi;
      }
      voxels_done = true;
    }
  }
  if (!voxels_done) {                                                        // float path: the bounding box and the indices from the points themselves
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int e = tid; e < L; e += 256) {
      const int i = e + 5;
      if (label_is_member(flags[i])) {
        const float4 p = cloud[i];
        mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
        mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
        mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      for (int d = 32; d > 0; d >>= 1) { mn[q] = fminf(mn[q], __shfl_down(mn[q], d, 64)); mx[q] = fmaxf(mx[q], __shfl_down(mx[q], d, 64)); }
      if (lane == 0) { s_red[q][wave] = mn[q]; s_red[3 + q][wave] = mx[q]; }
    }
    __syncthreads();
    int minb[3], divb[3];
    float fminb[3], gmn[3], gmx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      gmn[q] = fminf(fminf(s_red[q][0], s_red[q][1]), fminf(s_red[q][2], s_red[q][3]));
      gmx[q] = fmaxf(fmaxf(s_red[3 + q][0], s_red[3 + q][1]), fmaxf(s_red[3 + q][2], s_red[3 + q][3]));
    }
    const long long dx = (long long)((gmx[0] - gmn[0]) * inv) + 1, dy = (long long)((gmx[1] - gmn[1]) * inv) + 1, dz = (long long)((gmx[2] - gmn[2]) * inv) + 1;
    overflow = dx * dy * dz > 2147483647ll;   // PCL returns the input unfiltered in this case
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      minb[q] = (int)floorf(gmn[q] * inv);
      divb[q] = (int)floorf(gmx[q] * inv) - minb[q] + 1;
      fminb[q] = (float)minb[q];
    }
    cells_in_box = (long long)divb[0] * divb[1] * divb[2];
    for (int e = tid; e < L; e += 256) {
      const int i = e + 5;
      unsigned vi = 0xffffffffu;                                              // not a member (corner-labelled)
      if (label_is_member(flags[i])) {
        if (overflow) vi = (unsigned)e;         // every point its own cell -> output = input, in order
        else {
          const float4 p = cloud[i];
          const int i0 = (int)(floorf(p.x * inv) - fminb[0]);
          const int i1 = (int)(floorf(p.y * inv) - fminb[1]);
          const int i2 = (int)(floorf(p.z * inv) - fminb[2]);
          vi = (unsigned)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]);
        }
      }
      vis[e] = vi;
    }
  }
  __syncthreads();
  ALOAM_PHASE("after_bbox_voxidx");   // bounding box + voxel indices
  // 32-bit run keys (voxel index << EB | first element) whenever the voxel box is small enough — most rings: half the LDS
  // traffic and a third fewer VALU instructions in the sort; the 64-bit keys remain for rings whose box has more cells.
  constexpr int EB = NPAD <= 2048 ? 11 : 12;                                  // bits of an element index
  float4* out = a.less_flat + (long long)b * a.cap;                          // final place: offset = less-flat points of the rings in front
  if (overflow || cells_in_box <= (1ll << (32 - EB))) {
    // bits of a voxel index: every index is below cells_in_box (or, in PCL's overflow case, the element number itself)
    voxel_runs_tail<NPAD, unsigned, EB>(smem, flags, s_scan, s_misc, cloud, out, L, tid, lane, wave, lb + 3 * a.R, r, a.R, a.epoch, &a.meta[b].err);
  } else
    voxel_runs_tail<NPAD, unsigned long long, 32>(smem, flags, s_scan, s_misc, cloud, out, L, tid, lane, wave, lb + 3 * a.R, r, a.R, a.epoch, &a.meta[b].err);
  // the picked points go straight to their final place in the three clouds, in the reference's order (ring, sector, pick order;
  // sharp = the first two less-sharp picks, :301-311)
  {                                                                          // one slot per thread: a single memory round trip for the whole ring
    const int base_sharp = s_misc[40], base_less = s_misc[41], base_flat = s_misc[42];
    static_assert(kSectors * kSlots <= 256, "one thread per pick slot");
    if (const int q = tid; q < kSectors * kSlots) {
      const int j = q / kSlots, slot = q % kSlots;
      const int ncorner = s_misc[1 + j] & 0xff, nflat = s_misc[1 + j] >> 8;
      if (slot < kSharpPerSector) {
        if (slot < ncorner) a.sharp[(long long)b * a.R * 12 + base_sharp + s_misc[16 + j] + slot] = cloud[s_pick[j * kSlots + kSharpPerSector + slot]];
      } else if (slot < kSharpPerSector + kLessSharpPerSector) {
        const int k = slot - kSharpPerSector;
        if (k < ncorner) a.less_sharp[(long long)b * a.R * 120 + base_less + s_misc[24 + j] + k] = cloud[s_pick[j * kSlots + slot]];
      } else {
        const int k = slot - kSharpPerSector - kLessSharpPerSector;
        if (k < nflat) a.flat[(long long)b * a.R * 24 + base_flat + s_misc[32 + j] + k] = cloud[s_pick[j * kSlots + slot]];
      }
    }
  }
  ALOAM_PHASE("picks_out");
}

// Sizes of the four feature clouds of every sweep = sums of the published ring counts (one wave per sweep).
__global__ __launch_bounds__(64) void k_cloud_sizes(RegArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const unsigned long long* lb = a.lookback + (long long)b * 4 * a.R;
  int* err = &a.meta[b].err;
  const int n0 = gather_counts(lb + 0 * a.R, a.R, a.epoch, lane, err), n1 = gather_counts(lb + 1 * a.R, a.R, a.epoch, lane, err);
  const int n2 = gather_counts(lb + 2 * a.R, a.R, a.epoch, lane, err), n3 = gather_counts(lb + 3 * a.R, a.R, a.epoch, lane, err);
  if (lane == 0) { a.meta[b].n_sharp = n0; a.meta[b].n_less_sharp = n1; a.meta[b].n_flat = n2; a.meta[b].n_less_flat = n3; a.ring_ticket[b] = 0; }   // tickets for the next launch
}

// -------------------------------------------------------------------------------------------------------
size_t ring_features_lds_bytes(int npad) {
  const int maxn = npad + 11;
  const int a_bytes = 8 * npad + 128;
  const int flag_bytes = (maxn + 15) & ~15;
  return (size_t)((a_bytes + 15) & ~15) + (size_t)flag_bytes + (256 + 24 + 48) * sizeof(int) + 6 * 26 * sizeof(short) + 8;
}

void launch_find_ends(const RegArgs& a, const int* d_nin, hipStream_t s) { hipLaunchKernelGGL(k_find_ends, dim3(a.B), dim3(1024), 0, s, a, d_nin); }
void launch_front(const RegArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_front, dim3(a.B, a.NB), dim3(256), 0, s, a); }   // sweep-major, see k_front   // sweep-major, see k_front
void launch_ring_starts(const RegArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_ring_starts, dim3(a.B), dim3(64), 0, s, a); }
void launch_dense_cloud(const RegArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_dense_cloud, dim3(a.R, a.B), dim3(256), 0, s, a); }
void launch_ring_features(const RegArgs& a, int npad, float leaf, hipStream_t s) {
  const size_t lds = ring_features_lds_bytes(npad);
  if (npad <= 2048) hipLaunchKernelGGL(k_ring_features<2048>, dim3(a.B, a.R), dim3(256), lds, s, a, leaf);
  else hipLaunchKernelGGL(k_ring_features<4096>, dim3(a.B, a.R), dim3(256), lds, s, a, leaf);
  hipLaunchKernelGGL(k_cloud_sizes, dim3(a.B), dim3(64), 0, s, a);
}

}  // namespace aloam

#ifdef ALOAM_PHASE_CLOCK
extern "C" int aloam_debug_phase_clock(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(aloam::g_phase_clock), sizeof(unsigned long long) * 64); }
#endif
