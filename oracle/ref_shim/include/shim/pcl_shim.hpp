// oracle/ref_shim/include/shim/pcl_shim.hpp — stand-in for the slice of PCL 1.8 the A-LOAM nodes touch.
//
// TEST INFRASTRUCTURE ONLY (see shim/ros_shim.hpp).  PCL is a third-party dependency that is neither vendored in
// /root/reference nor installed here (pinned to 1.8.0 only by reference docker/Dockerfile:4), so its behaviour is
// restated from the published implementation:
//   VoxelGrid<PointT>::applyFilter   pcl/filters/impl/voxel_grid.hpp (min/max -> integer cell index -> std::sort by cell
//                                    -> per-cell centroid of ALL fields, downsample_all_data = true, min_points 0)
//   KdTreeFLANN<PointT>              exact (eps = 0) k-NN over x,y,z with FLANN's L2_Simple f32 accumulation; written
//                                    here as a median-split tree that returns what a brute-force scan ordered by
//                                    (distance, index) returns - result-equivalent to FLANN except on exact ties
//   fromROSMsg / toROSMsg            field lookup by name / the PointXYZI wire layout (x@0 y@4 z@8 intensity@16, step 32)
//   removeNaNFromPointCloud          pcl/filters/impl/filter.hpp
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "shim/ros_shim.hpp"

namespace pcl {

struct PCLHeader { uint32_t seq = 0; uint64_t stamp = 0; std::string frame_id; };

struct PointXYZ { float x = 0, y = 0, z = 0; PointXYZ() {} PointXYZ(float a, float b, float c) : x(a), y(b), z(c) {} };
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };

template <class PointT>
struct PointCloud {
  PCLHeader header;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = 0; height = 0; }
  void push_back(const PointT& p) { points.push_back(p); width = static_cast<uint32_t>(points.size()); height = 1; }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  PointCloud& operator+=(const PointCloud& rhs) {
    points.insert(points.end(), rhs.points.begin(), rhs.points.end());
    width = static_cast<uint32_t>(points.size());
    height = 1;
    is_dense = is_dense && rhs.is_dense;
    return *this;
  }
};

// ---- conversions ------------------------------------------------------------------------------------------
namespace detail {
inline int field_offset(const sensor_msgs::PointCloud2& m, const char* name) {
  for (const auto& f : m.fields) if (f.name == name) return static_cast<int>(f.offset);
  return -1;
}
inline void set_fields(sensor_msgs::PointCloud2& m, bool with_intensity) {
  const char* names[4] = {"x", "y", "z", "intensity"};
  const uint32_t offs[4] = {0, 4, 8, 16};
  m.fields.clear();
  for (int k = 0; k < (with_intensity ? 4 : 3); ++k) {
    sensor_msgs::PointField f; f.name = names[k]; f.offset = offs[k]; f.datatype = sensor_msgs::PointField::FLOAT32; f.count = 1;
    m.fields.push_back(f);
  }
}
}  // namespace detail

inline void fromROSMsg(const sensor_msgs::PointCloud2& m, PointCloud<PointXYZ>& c) {
  const int ox = detail::field_offset(m, "x"), oy = detail::field_offset(m, "y"), oz = detail::field_offset(m, "z");
  const size_t n = static_cast<size_t>(m.width) * m.height;
  c.points.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* p = m.data.data() + i * m.point_step;
    std::memcpy(&c.points[i].x, p + ox, 4); std::memcpy(&c.points[i].y, p + oy, 4); std::memcpy(&c.points[i].z, p + oz, 4);
  }
  c.width = m.width; c.height = m.height; c.is_dense = m.is_dense;
}
inline void fromROSMsg(const sensor_msgs::PointCloud2& m, PointCloud<PointXYZI>& c) {
  const int ox = detail::field_offset(m, "x"), oy = detail::field_offset(m, "y"), oz = detail::field_offset(m, "z"), oi = detail::field_offset(m, "intensity");
  const size_t n = static_cast<size_t>(m.width) * m.height;
  c.points.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* p = m.data.data() + i * m.point_step;
    std::memcpy(&c.points[i].x, p + ox, 4); std::memcpy(&c.points[i].y, p + oy, 4); std::memcpy(&c.points[i].z, p + oz, 4);
    if (oi >= 0) std::memcpy(&c.points[i].intensity, p + oi, 4); else c.points[i].intensity = 0.f;
  }
  c.width = m.width; c.height = m.height; c.is_dense = m.is_dense;
}
inline void toROSMsg(const PointCloud<PointXYZI>& c, sensor_msgs::PointCloud2& m) {
  const size_t n = c.points.size();
  m.height = 1; m.width = static_cast<uint32_t>(n); m.point_step = 32; m.row_step = static_cast<uint32_t>(32 * n);
  m.is_bigendian = false; m.is_dense = c.is_dense;
  detail::set_fields(m, true);
  m.data.assign(32 * n, 0);
  for (size_t i = 0; i < n; ++i) {
    uint8_t* p = m.data.data() + 32 * i;
    std::memcpy(p, &c.points[i].x, 4); std::memcpy(p + 4, &c.points[i].y, 4); std::memcpy(p + 8, &c.points[i].z, 4);
    std::memcpy(p + 16, &c.points[i].intensity, 4);
  }
}

// pcl::removeNaNFromPointCloud (filter.hpp): order-preserving compaction of the finite points.
template <class PointT>
void removeNaNFromPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, std::vector<int>& index) {
  if (&in != &out) { out.header = in.header; out.points.resize(in.points.size()); }
  index.resize(in.points.size());
  size_t j = 0;
  if (in.is_dense) {
    if (&in != &out) out.points = in.points;
    for (j = 0; j < out.points.size(); ++j) index[j] = static_cast<int>(j);
  } else {
    for (size_t i = 0; i < in.points.size(); ++i) {
      if (!std::isfinite(in.points[i].x) || !std::isfinite(in.points[i].y) || !std::isfinite(in.points[i].z)) continue;
      out.points[j] = in.points[i];
      index[j] = static_cast<int>(i);
      j++;
    }
    if (j != in.points.size()) { out.points.resize(j); index.resize(j); }
    out.height = 1; out.width = static_cast<uint32_t>(j);
    out.is_dense = true;
  }
}

// ---- VoxelGrid ----------------------------------------------------------------------------------------------
template <class PointT>
class VoxelGrid {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
  void setLeafSize(float lx, float ly, float lz) { leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz; for (int k = 0; k < 3; ++k) inv_[k] = 1.0f / leaf_[k]; }
  void filter(PointCloud<PointT>& output) {
    const PointCloud<PointT>& in = *input_;
    output.points.clear(); output.height = 1; output.is_dense = true;
    if (in.points.empty()) { output.width = 0; return; }
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[0], -mn[0]};
    for (const PointT& p : in.points) {
      if (!in.is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
      const float v[3] = {p.x, p.y, p.z};
      for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); }
    }
    const int64_t dx = static_cast<int64_t>((mx[0] - mn[0]) * inv_[0]) + 1, dy = static_cast<int64_t>((mx[1] - mn[1]) * inv_[1]) + 1,
                  dz = static_cast<int64_t>((mx[2] - mn[2]) * inv_[2]) + 1;
    if (dx * dy * dz > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) { output = in; return; }   // PCL warns and copies
    int minb[3], divb[3];
    for (int k = 0; k < 3; ++k) {
      minb[k] = static_cast<int>(std::floor(mn[k] * inv_[k]));
      divb[k] = static_cast<int>(std::floor(mx[k] * inv_[k])) - minb[k] + 1;
    }
    const int mul[3] = {1, divb[0], divb[0] * divb[1]};
    struct Entry { unsigned idx; unsigned pt; bool operator<(const Entry& o) const { return idx < o.idx; } };
    std::vector<Entry> ev;
    ev.reserve(in.points.size());
    for (size_t i = 0; i < in.points.size(); ++i) {
      const PointT& p = in.points[i];
      if (!in.is_dense && (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))) continue;
      const int i0 = static_cast<int>(std::floor(p.x * inv_[0]) - static_cast<float>(minb[0]));
      const int i1 = static_cast<int>(std::floor(p.y * inv_[1]) - static_cast<float>(minb[1]));
      const int i2 = static_cast<int>(std::floor(p.z * inv_[2]) - static_cast<float>(minb[2]));
      ev.push_back(Entry{static_cast<unsigned>(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), static_cast<unsigned>(i)});
    }
    std::sort(ev.begin(), ev.end(), std::less<Entry>());
    for (size_t a = 0; a < ev.size();) {
      size_t b = a + 1;
      while (b < ev.size() && ev[b].idx == ev[a].idx) ++b;
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
      for (size_t q = a; q < b; ++q) { const PointT& p = in.points[ev[q].pt]; sx += p.x; sy += p.y; sz += p.z; si += p.intensity; }
      const float n = static_cast<float>(b - a);
      PointT o;
#ifdef SHIM_CENTROID_BY_RECIPROCAL   // tools/third_party_sensitivity.py: "what if the real PCL / Eigen pair scales the sums by 1 / n instead of dividing them"
      { const float rn = 1.0f / n; o.x = sx * rn; o.y = sy * rn; o.z = sz * rn; o.intensity = si * rn; }
#else
      o.x = sx / n; o.y = sy / n; o.z = sz / n; o.intensity = si / n;
#endif
      output.points.push_back(o);
      a = b;
    }
    output.width = static_cast<uint32_t>(output.points.size());
  }
 private:
  typename PointCloud<PointT>::ConstPtr input_;
  float leaf_[3] = {0, 0, 0}, inv_[3] = {0, 0, 0};
};

// ---- KdTreeFLANN ----------------------------------------------------------------------------------------------
// Exact k-NN with FLANN's L2_Simple arithmetic (f32, the three squared differences added x -> y -> z) and results in ascending
// (distance, index) order - the order a brute-force scan with that key produces; a median-split tree over the finite points only prunes
// sub-trees that cannot hold a smaller key.  The pruning bound is the squared f32 distance to the splitting plane: rounding is monotone,
// so every point beyond the plane has a distance that is no smaller, and `<=` keeps equal distances with a lower index reachable.
// (The true FLANN breaks exact ties in traversal order, which is not restated here; SURVEY.md Appendix C.)
template <class PointT>
class KdTreeFLANN {
 public:
  typedef std::shared_ptr<KdTreeFLANN<PointT>> Ptr;
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) {
    cloud_ = c; idx_.clear(); nodes_.clear();
    for (size_t i = 0; i < c->points.size(); ++i) {
      const PointT& p = c->points[i];
      if (std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)) idx_.push_back(static_cast<int>(i));
    }
    if (!idx_.empty()) { nodes_.reserve(idx_.size() / 4 + 8); build(0, static_cast<int>(idx_.size())); }
  }
  int nearestKSearch(const PointT& q, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
    if (k > static_cast<int>(idx_.size())) k = static_cast<int>(idx_.size());
    k_indices.resize(k); k_sqr_distances.resize(k);
    if (k == 0) return 0;
    std::vector<std::pair<float, int>> best;    // ascending (distance, index), at most k entries
    best.reserve(k + 1);
    const float qv[3] = {q.x, q.y, q.z};
    search(0, qv, k, best);
#ifdef SHIM_KNN_TIES_HIGHEST          // tools/third_party_sensitivity.py: equal distances resolved the other way round (FLANN's own order is its traversal order)
    for (int j = 0; j < k; ++j) { k_indices[j] = -best[j].second; k_sqr_distances[j] = best[j].first; }
#else
    for (int j = 0; j < k; ++j) { k_indices[j] = best[j].second; k_sqr_distances[j] = best[j].first; }
#endif
    return k;
  }
 private:
  struct Node { int lo, hi, left, right, axis; float split; };   // leaf: left < 0, points idx_[lo, hi)
  static float coord(const PointT& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }
  int build(int lo, int hi) {
    const int id = static_cast<int>(nodes_.size());
    nodes_.push_back(Node{lo, hi, -1, -1, 0, 0.f});
    if (hi - lo <= 12) return id;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()}, mx[3] = {-mn[0], -mn[0], -mn[0]};
    for (int i = lo; i < hi; ++i) for (int a = 0; a < 3; ++a) { const float v = coord(cloud_->points[idx_[i]], a); mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v); }
    int axis = 0;
    for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    if (!(mx[axis] > mn[axis])) return id;                        // all points coincide: one leaf
    const int mid = lo + (hi - lo) / 2;
    std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi,
                     [&](int x, int y) { return coord(cloud_->points[x], axis) < coord(cloud_->points[y], axis); });
    const float split = coord(cloud_->points[idx_[mid]], axis);   // left: <= split, right: >= split
    const int l = build(lo, mid), r = build(mid, hi);
    nodes_[id].left = l; nodes_[id].right = r; nodes_[id].axis = axis; nodes_[id].split = split;
    return id;
  }
  void search(int id, const float q[3], int k, std::vector<std::pair<float, int>>& best) const {
    const Node& n = nodes_[id];
    if (n.left < 0) {
      for (int j = n.lo; j < n.hi; ++j) {
        const int i = idx_[j];
        const PointT& p = cloud_->points[i];
        float d = 0.f;
        float diff = p.x - q[0]; d += diff * diff;
        diff = p.y - q[1]; d += diff * diff;
        diff = p.z - q[2]; d += diff * diff;
#ifdef SHIM_KNN_TIES_HIGHEST
        const std::pair<float, int> key(d, -i);
#else
        const std::pair<float, int> key(d, i);
#endif
        if (static_cast<int>(best.size()) == k && !(key < best.back())) continue;
        best.insert(std::upper_bound(best.begin(), best.end(), key), key);
        if (static_cast<int>(best.size()) > k) best.pop_back();
      }
      return;
    }
    const float diff = n.split - q[n.axis];
    const int near = diff >= 0.f ? n.left : n.right, far = diff >= 0.f ? n.right : n.left;
    search(near, q, k, best);
    if (static_cast<int>(best.size()) < k || diff * diff <= best.back().first) search(far, q, k, best);
  }
  typename PointCloud<PointT>::ConstPtr cloud_;
  std::vector<int> idx_;
  std::vector<Node> nodes_;
};

}  // namespace pcl
