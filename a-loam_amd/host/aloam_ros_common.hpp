// a-loam_amd/host/aloam_ros_common.hpp — message <-> 16-byte point record helpers shared by the three node shims.
//
// The nodes in this directory keep the aloam_velodyne ROS node / topic surface (names, message types, frame ids, stamps,
// queue depths; reference src/scanRegistration.cpp:461-503, src/laserOdometry.cpp:186-263,508-599, src/laserMapping.cpp:
// 175-229,803-938) and replace the per-scan work by calls into libaloam_mi355x.so.  They are written against the roscpp /
// sensor_msgs / nav_msgs / tf headers only; nothing here depends on PCL, Eigen or Ceres.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "aloam_mi355x.h"

namespace aloam_host {

inline int field_offset(const sensor_msgs::PointCloud2& m, const char* name) {
  for (const auto& f : m.fields) if (f.name == name) return static_cast<int>(f.offset);
  return -1;
}

// What pcl::fromROSMsg<PointXYZI> hands the reference: x, y, z (and intensity when present) looked up by field name.
inline int msg_to_xyzi(const sensor_msgs::PointCloud2& m, std::vector<float>* out) {
  const int n = static_cast<int>(m.width * m.height);
  const int ox = field_offset(m, "x"), oy = field_offset(m, "y"), oz = field_offset(m, "z"), oi = field_offset(m, "intensity");
  out->assign(4 * static_cast<size_t>(n), 0.f);
  if (ox < 0 || oy < 0 || oz < 0) return -1;
  for (int i = 0; i < n; ++i) {
    const uint8_t* p = m.data.data() + static_cast<size_t>(i) * m.point_step;
    float* o = out->data() + 4 * static_cast<size_t>(i);
    std::memcpy(o, p + ox, 4); std::memcpy(o + 1, p + oy, 4); std::memcpy(o + 2, p + oz, 4);
    if (oi >= 0) std::memcpy(o + 3, p + oi, 4);
  }
  return n;
}

// The message pcl::toROSMsg<PointXYZI> produces: x@0 y@4 z@8 intensity@16 FLOAT32, point_step 32, height 1, little endian.
inline sensor_msgs::PointCloud2 xyzi_to_msg(const float* xyzi, int n, const ros::Time& stamp, const std::string& frame) {
  sensor_msgs::PointCloud2 m;
  const char* names[4] = {"x", "y", "z", "intensity"};
  const uint32_t offs[4] = {0, 4, 8, 16};
  for (int k = 0; k < 4; ++k) {
    sensor_msgs::PointField f;
    f.name = names[k]; f.offset = offs[k]; f.datatype = sensor_msgs::PointField::FLOAT32; f.count = 1;
    m.fields.push_back(f);
  }
  m.height = 1; m.width = static_cast<uint32_t>(n); m.point_step = 32; m.row_step = static_cast<uint32_t>(32 * n);
  m.is_bigendian = false; m.is_dense = true;
  m.data.assign(32 * static_cast<size_t>(n), 0);
  for (int i = 0; i < n; ++i) {
    uint8_t* p = m.data.data() + 32 * static_cast<size_t>(i);
    std::memcpy(p, xyzi + 4 * static_cast<size_t>(i), 12);
    std::memcpy(p + 16, xyzi + 4 * static_cast<size_t>(i) + 3, 4);
  }
  m.header.stamp = stamp;
  m.header.frame_id = frame;
  return m;
}

inline sensor_msgs::PointCloud2 cloud_msg(aloam_ctx* ctx, int which, const ros::Time& stamp, const std::string& frame) {
  const int n = aloam_cloud_size(ctx, 0, which);
  std::vector<float> v(4 * static_cast<size_t>(n > 0 ? n : 0));
  if (n > 0) aloam_get_cloud(ctx, 0, which, v.data(), n);
  return xyzi_to_msg(v.data(), n > 0 ? n : 0, stamp, frame);
}

}  // namespace aloam_host
