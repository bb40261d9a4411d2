// oracle/ref_shim/ref_io.hpp — tiny binary I/O shared by the _ref drivers (TEST INFRASTRUCTURE ONLY).
// File = sequence of records; a cloud record is int32 n followed by n x 4 float32 (x, y, z, intensity).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "shim/pcl_shim.hpp"

namespace ref_io {
inline void must(bool ok, const char* what) { if (!ok) { std::fprintf(stderr, "ref driver: %s\n", what); std::exit(2); } }
inline int32_t read_i32(FILE* f) { int32_t v; must(std::fread(&v, 4, 1, f) == 1, "short read"); return v; }
inline void write_i32(FILE* f, int32_t v) { std::fwrite(&v, 4, 1, f); }
inline void write_f64(FILE* f, const double* v, int n) { std::fwrite(v, 8, n, f); }
inline std::vector<float> read_cloud(FILE* f) {
  const int32_t n = read_i32(f);
  std::vector<float> v(static_cast<size_t>(n) * 4);
  if (n) must(std::fread(v.data(), 16, n, f) == static_cast<size_t>(n), "short read");
  return v;
}
inline void write_cloud(FILE* f, const sensor_msgs::PointCloud2& m) {   // PointXYZI wire layout -> 16-byte records
  const int32_t n = static_cast<int32_t>(m.width * m.height);
  write_i32(f, n);
  for (int32_t i = 0; i < n; ++i) {
    float rec[4];
    const uint8_t* p = m.data.data() + static_cast<size_t>(i) * m.point_step;
    std::memcpy(&rec[0], p, 4); std::memcpy(&rec[1], p + 4, 4); std::memcpy(&rec[2], p + 8, 4); std::memcpy(&rec[3], p + 16, 4);
    std::fwrite(rec, 4, 4, f);
  }
}
// 16-byte records -> the message pcl::toROSMsg<PointXYZI> would have produced (x@0 y@4 z@8 intensity@16, step 32)
inline sensor_msgs::PointCloud2 make_msg(const std::vector<float>& v, double stamp, bool dense = true) {
  pcl::PointCloud<pcl::PointXYZI> c;
  c.points.resize(v.size() / 4);
  for (size_t i = 0; i < c.points.size(); ++i) { c.points[i].x = v[4 * i]; c.points[i].y = v[4 * i + 1]; c.points[i].z = v[4 * i + 2]; c.points[i].intensity = v[4 * i + 3]; }
  c.is_dense = dense;
  sensor_msgs::PointCloud2 m;
  pcl::toROSMsg(c, m);
  m.header.stamp.fromSec(stamp);
  m.header.frame_id = "/camera_init";
  return m;
}
}  // namespace ref_io
