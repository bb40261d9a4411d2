// oracle/oracle_registration.cpp — CPU restatement of A-LOAM scan registration.
// TEST INFRASTRUCTURE ONLY (see aloam_oracle.h).  Follows reference src/scanRegistration.cpp:
//   range / NaN filter           :85-112,136-137
//   sweep start / end azimuth    :140-153
//   ring id + relative time      :156-241
//   ring concatenation           :246-252
//   11-tap curvature             :256-266
//   per-ring 6-sector selection  :277-399
//   per-ring 0.2 m voxel filter  :401-407  (pcl::VoxelGrid, PCL 1.8.0 — SURVEY.md Appendix B)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "oracle_internal.hpp"

namespace orc {

// ------------------------------------------------------------------------------------------------
// atan2f restated after the classic FDLIBM float algorithm (argument reduction to 4 break points +
// odd/even polynomial).  glibc 2.35's atan2f — what `atan2` resolves to at scanRegistration.cpp:141,
// 142,208 through `using std::atan2` (:56) on float arguments — produces the same bits (tests check
// 1e7 samples); the HIP kernels carry the same code so ring-relative times are bit-exact.
// ------------------------------------------------------------------------------------------------
namespace {
inline int32_t f2i(float x) { int32_t i; std::memcpy(&i, &x, 4); return i; }
inline float i2f(int32_t i) { float x; std::memcpy(&x, &i, 4); return x; }
const float kAtanHi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
const float kAtanLo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
const float kAT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                       9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                       4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
float atanf_port(float x) {
  const int32_t hx = f2i(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? kAtanHi[3] + kAtanLo[3] : -kAtanHi[3] - kAtanLo[3];
  }
  if (ix < 0x3ee00000) {   // |x| < 7/16
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = std::fabs(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else                 { id = 3; x = -1.0f / x; }
    }
  }
  const float z = x * x;
  const float w = z * z;
  const float s1 = z * (kAT[0] + w * (kAT[2] + w * (kAT[4] + w * (kAT[6] + w * (kAT[8] + w * kAT[10])))));
  const float s2 = w * (kAT[1] + w * (kAT[3] + w * (kAT[5] + w * (kAT[7] + w * kAT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = kAtanHi[id] - ((x * (s1 + s2) - kAtanLo[id]) - x);
  return hx < 0 ? -r : r;
}
}  // namespace

float atan2f_port(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  const int32_t hx = f2i(x), hy = f2i(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return atanf_port(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    if (m < 2) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
    }
    switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = atanf_port(std::fabs(y / x));
  switch (m) {
    case 0: return z;
    case 1: return i2f(f2i(z) ^ (int32_t)0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// ------------------------------------------------------------------------------------------------
// pcl::VoxelGrid<PointXYZI>::applyFilter, defaults (downsample_all_data = true, min_points 0).
// ------------------------------------------------------------------------------------------------
void voxel_filter(const std::vector<P4>& in, float leaf, bool canonical, std::vector<P4>* out) {
  out->clear();
  if (in.empty()) return;
  const float inv = 1.0f / leaf;                       // inverse_leaf_size_ (f32)
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (const P4& p : in) {                            // getMinMax3D over the (dense) input
    mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
    mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
    mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
  }
  // overflow guard of PCL: too many voxels -> input returned unfiltered (with a warning there)
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
  const int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
  const int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) { *out = in; return; }
  int minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)std::floor(mn[a] * inv);
    maxb[a] = (int)std::floor(mx[a] * inv);
    divb[a] = maxb[a] - minb[a] + 1;
  }
  const int mul[3] = {1, divb[0], divb[0] * divb[1]};
  struct Cell { unsigned idx; unsigned pt; };
  std::vector<Cell> cells(in.size());
  for (size_t n = 0; n < in.size(); ++n) {
    const int i0 = (int)(std::floor(in[n].x * inv) - (float)minb[0]);
    const int i1 = (int)(std::floor(in[n].y * inv) - (float)minb[1]);
    const int i2 = (int)(std::floor(in[n].z * inv) - (float)minb[2]);
    cells[n].idx = (unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]);
    cells[n].pt = (unsigned)n;
  }
  if (canonical)
    std::stable_sort(cells.begin(), cells.end(), [](const Cell& a, const Cell& b) { return a.idx < b.idx; });
  else
    std::sort(cells.begin(), cells.end(), [](const Cell& a, const Cell& b) { return a.idx < b.idx; });
  size_t first = 0;
  while (first < cells.size()) {
    size_t last = first + 1;
    while (last < cells.size() && cells[last].idx == cells[first].idx) ++last;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;      // f32 accumulation of all four fields
    for (size_t k = first; k < last; ++k) {
      const P4& p = in[cells[k].pt];
      sx += p.x; sy += p.y; sz += p.z; si += p.i;
    }
    const float cnt = (float)(last - first);
    out->push_back(P4{sx / cnt, sy / cnt, sz / cnt, si / cnt});
    first = last;
  }
}

// ------------------------------------------------------------------------------------------------
// laserCloudHandler body.
// ------------------------------------------------------------------------------------------------
int register_scan(const orc_config& cfg, const void* pts, int n_in, int stride_bytes, RegistrationResult* out, std::string* err) {
  const int R = cfg.n_scans;
  if (!cfg.ring_from_field && R != 16 && R != 32 && R != 64) { *err = "only 16/32/64 scan lines have an elevation formula"; return -2; }
  if (R <= 0 || stride_bytes < 16 || n_in < 0) { *err = "bad arguments"; return -1; }

  // -- decode + NaN removal + removeClosedPointCloud (:132-137, :85-112); order preserved.
  std::vector<P4> in;
  in.reserve(n_in);
  const float thres = cfg.min_range;
  for (int n = 0; n < n_in; ++n) {
    P4 p;
    std::memcpy(&p, (const char*)pts + (size_t)n * stride_bytes, 16);
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    log_decision(kDecRange, p.x * p.x + p.y * p.y + p.z * p.z, thres * thres);
    if (p.x * p.x + p.y * p.y + p.z * p.z < thres * thres) continue;
    in.push_back(p);
  }
  int cloudSize = (int)in.size();
  if (cloudSize == 0) { *err = "no point survives the NaN / minimum-range filter"; return -3; }

  // -- sweep start / end azimuth (:141-153): atan2f on float arguments, mixed float/double arithmetic.
  float startOri = -std::atan2(in[0].y, in[0].x);
  float endOri = (float)(-std::atan2(in[cloudSize - 1].y, in[cloudSize - 1].x) + 2 * M_PI);
  if (endOri - startOri > 3 * M_PI) endOri = (float)(endOri - 2 * M_PI);
  else if (endOri - startOri < M_PI) endOri = (float)(endOri + 2 * M_PI);

  // -- ring id + relative time (:156-241)
  bool halfPassed = false;
  int count = cloudSize;
  std::vector<std::vector<P4>> rings(R);
  for (int n = 0; n < cloudSize; ++n) {
    P4 point{in[n].x, in[n].y, in[n].z, 0.f};
    int scanID = 0;
    if (cfg.ring_from_field) {
      // config 4 (no reference formula for 128 lines, scanRegistration.cpp:472-476 would exit):
      // the ring is carried by the 4th input float.
      scanID = (int)in[n].i;
      if (scanID > R - 1 || scanID < 0) { count--; continue; }
    } else {
      // `atan` / `sqrt` are unqualified at :166 -> the C double overloads (GCC 5, the pinned image);
      // the argument of sqrt is an f32 expression.
      const float angle = (float)(std::atan((double)point.z / std::sqrt((double)(point.x * point.x + point.y * point.y))) * 180 / M_PI);
      if (R == 16) {
        scanID = (int)((angle + 15) / 2 + 0.5);
        if (scanID > R - 1 || scanID < 0) { count--; continue; }
      } else if (R == 32) {
        scanID = (int)((angle + 92.0 / 3.0) * 3.0 / 4.0);
        if (scanID > R - 1 || scanID < 0) { count--; continue; }
      } else {
        if (angle >= -8.83) scanID = (int)((2 - angle) * 3.0 + 0.5);
        else scanID = R / 2 + (int)((-8.83 - angle) * 2.0 + 0.5);
        if (angle > 2 || angle < -24.33 || scanID > 50 || scanID < 0) { count--; continue; }
      }
    }
    float ori = -std::atan2(point.y, point.x);
    if (!halfPassed) {
      if (ori < startOri - M_PI / 2) ori = (float)(ori + 2 * M_PI);
      else if (ori > startOri + M_PI * 3 / 2) ori = (float)(ori - 2 * M_PI);
      if (ori - startOri > M_PI) halfPassed = true;
    } else {
      ori = (float)(ori + 2 * M_PI);
      if (ori < endOri - M_PI * 3 / 2) ori = (float)(ori + 2 * M_PI);
      else if (ori > endOri + M_PI / 2) ori = (float)(ori - 2 * M_PI);
    }
    const float relTime = (ori - startOri) / (endOri - startOri);
    point.i = (float)(scanID + 0.1 * relTime);          // scanPeriod = 0.1 (:60,239)
    rings[scanID].push_back(point);
  }
  cloudSize = count;

  // -- concatenation, selectable range per ring (:246-252)
  out->cloud.clear();
  out->ring_start.assign(R, 0);
  out->ring_count.assign(R, 0);
  std::vector<int> S(R), E(R);
  for (int r = 0; r < R; ++r) {
    out->ring_start[r] = (int)out->cloud.size();
    out->ring_count[r] = (int)rings[r].size();
    S[r] = (int)out->cloud.size() + 5;
    out->cloud.insert(out->cloud.end(), rings[r].begin(), rings[r].end());
    E[r] = (int)out->cloud.size() - 6;
  }
  const std::vector<P4>& c = out->cloud;

  // -- curvature (:256-266)
  out->curvature.assign(cloudSize, 0.f);
  out->label.assign(cloudSize, 0);
  out->picked.assign(cloudSize, 0);
  std::vector<int> sortInd(cloudSize, 0);
  for (int i = 5; i < cloudSize - 5; ++i) {
    const float dX = c[i - 5].x + c[i - 4].x + c[i - 3].x + c[i - 2].x + c[i - 1].x - 10 * c[i].x + c[i + 1].x + c[i + 2].x + c[i + 3].x + c[i + 4].x + c[i + 5].x;
    const float dY = c[i - 5].y + c[i - 4].y + c[i - 3].y + c[i - 2].y + c[i - 1].y - 10 * c[i].y + c[i + 1].y + c[i + 2].y + c[i + 3].y + c[i + 4].y + c[i + 5].y;
    const float dZ = c[i - 5].z + c[i - 4].z + c[i - 3].z + c[i - 2].z + c[i - 1].z - 10 * c[i].z + c[i + 1].z + c[i + 2].z + c[i + 3].z + c[i + 4].z + c[i + 5].z;
    out->curvature[i] = dX * dX + dY * dY + dZ * dZ;
    sortInd[i] = i;
  }
  const std::vector<float>& curv = out->curvature;
  std::vector<int>& picked = out->picked;
  std::vector<int>& label = out->label;

  auto gap2 = [&](int a, int b) {   // squared step between consecutive points (:321-324 and siblings)
    const float dX = c[a].x - c[b].x, dY = c[a].y - c[b].y, dZ = c[a].z - c[b].z;
    return dX * dX + dY * dY + dZ * dZ;
  };
  auto suppress = [&](int ind) {    // (:317-342, :364-388)
    picked[ind] = 1;
    for (int l = 1; l <= 5; ++l) { log_decision(kDecGap, gap2(ind + l, ind + l - 1), 0.05); if (gap2(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
    for (int l = -1; l >= -5; --l) { log_decision(kDecGap, gap2(ind + l, ind + l + 1), 0.05); if (gap2(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
  };

  out->sharp.clear(); out->less_sharp.clear(); out->flat.clear(); out->less_flat.clear();
  for (int r = 0; r < R; ++r) {
    if (E[r] - S[r] < 6) continue;                                  // :279
    std::vector<P4> lessFlatScan;
    for (int j = 0; j < 6; ++j) {
      const int sp = S[r] + (E[r] - S[r]) * j / 6;                  // :284
      const int ep = S[r] + (E[r] - S[r]) * (j + 1) / 6 - 1;        // :285
      if (cfg.canonical_order)
        std::sort(sortInd.begin() + sp, sortInd.begin() + ep + 1,
                  [&](int a, int b) { return curv[a] < curv[b] || (curv[a] == curv[b] && a < b); });
      else
        std::sort(sortInd.begin() + sp, sortInd.begin() + ep + 1, [&](int a, int b) { return curv[a] < curv[b]; });  // :71,288

      int largestPickedNum = 0;                                     // :291-344
      for (int k = ep; k >= sp; --k) {
        const int ind = sortInd[k];
        if (picked[ind] == 0) log_decision(kDecCurvCorner, curv[ind], 0.1);
        if (picked[ind] == 0 && curv[ind] > 0.1) {
          largestPickedNum++;
          if (largestPickedNum <= 2) { label[ind] = 2; out->sharp.push_back(c[ind]); out->less_sharp.push_back(c[ind]); }
          else if (largestPickedNum <= 20) { label[ind] = 1; out->less_sharp.push_back(c[ind]); }
          else break;
          suppress(ind);
        }
      }
      int smallestPickedNum = 0;                                    // :346-390
      for (int k = sp; k <= ep; ++k) {
        const int ind = sortInd[k];
        if (picked[ind] == 0) log_decision(kDecCurvFlat, curv[ind], 0.1);
        if (picked[ind] == 0 && curv[ind] < 0.1) {
          label[ind] = -1;
          out->flat.push_back(c[ind]);
          smallestPickedNum++;
          if (smallestPickedNum >= 4) break;                         // 4th flat point is never marked / suppressed
          suppress(ind);
        }
      }
      for (int k = sp; k <= ep; ++k)                                // :392-398
        if (label[k] <= 0) lessFlatScan.push_back(c[k]);
    }
    std::vector<P4> ds;
    voxel_filter(lessFlatScan, 0.2f, cfg.canonical_order != 0, &ds);  // :401-405
    out->less_flat.insert(out->less_flat.end(), ds.begin(), ds.end());  // :407
  }
  return 0;
}

}  // namespace orc
