// a-loam_amd/host/scan_registration_node.cpp — the `ascanRegistration` node on top of libaloam_mi355x.so.
// Same node name, parameters, topics, frame ids and stamps as the reference (src/scanRegistration.cpp:461-503); the body
// of laserCloudHandler (:127-411) is one call into the C ABI.
#include <chrono>
#include <cstdio>

#include "aloam_ros_common.hpp"

namespace {
aloam_ctx* g_ctx = nullptr;
ros::Publisher pubLaserCloud, pubCornerPointsSharp, pubCornerPointsLessSharp, pubSurfPointsFlat, pubSurfPointsLessFlat, pubRemovePoints;
std::vector<ros::Publisher> pubEachScan;
std::vector<float> g_repack;
const bool PUB_EACH_LINE = false;                          // reference src/scanRegistration.cpp:81
int g_n_scans = 16;
}  // namespace
aloam_ctx* aloam_node_context() { return g_ctx; }

void laserCloudHandler(const sensor_msgs::PointCloud2ConstPtr& msg) {
  const auto t_whole = std::chrono::steady_clock::now();
  const int n = static_cast<int>(msg->width * msg->height);
  const int ox = aloam_host::field_offset(*msg, "x"), oy = aloam_host::field_offset(*msg, "y"), oz = aloam_host::field_offset(*msg, "z");
  const void* scans[1];
  int n_in[1] = {n};
  int stride = static_cast<int>(msg->point_step);
  if (ox == 0 && oy == 4 && oz == 8 && stride >= 16 && stride % 4 == 0) {
    scans[0] = msg->data.data();                           // Velodyne drivers and kittiHelper both lay points out like this
  } else {
    if (aloam_host::msg_to_xyzi(*msg, &g_repack) < 0) { ROS_WARN("point cloud without x / y / z fields"); return; }
    scans[0] = g_repack.data();
    stride = 16;
  }
  if (aloam_scan_register(g_ctx, scans, n_in, stride) != ALOAM_OK || aloam_synchronize(g_ctx) != ALOAM_OK) {
    ROS_WARN("scan registration failed: %s", aloam_last_error(g_ctx));
    return;
  }
  const ros::Time stamp = msg->header.stamp;               // every output carries the input stamp (:415)
  pubLaserCloud.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_FULL, stamp, "/camera_init"));
  pubCornerPointsSharp.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_SHARP, stamp, "/camera_init"));
  pubCornerPointsLessSharp.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_LESS_SHARP, stamp, "/camera_init"));
  pubSurfPointsFlat.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_FLAT, stamp, "/camera_init"));
  pubSurfPointsLessFlat.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_LESS_FLAT, stamp, "/camera_init"));
  if (PUB_EACH_LINE) {                                     // one topic per ring (:444-454): slices of the ring-ordered cloud
    std::vector<int> start(g_n_scans), count(g_n_scans);
    const int nc = aloam_cloud_size(g_ctx, 0, ALOAM_CLOUD_FULL);
    std::vector<float> full(4 * static_cast<size_t>(nc > 0 ? nc : 1));
    if (nc > 0 && aloam_get_cloud(g_ctx, 0, ALOAM_CLOUD_FULL, full.data(), nc) == nc && aloam_get_ring_ranges(g_ctx, 0, start.data(), count.data()) == g_n_scans)
      for (int i = 0; i < g_n_scans; i++) pubEachScan[i].publish(aloam_host::xyzi_to_msg(full.data() + 4 * static_cast<size_t>(start[i]), count[i], stamp, "/camera_init"));
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_whole).count();
  printf("scan registration time %f ms *************\n", ms);
  if (ms > 100) ROS_WARN("scan registration process over 100ms");   // :456-458
}

int main(int argc, char** argv) {
  ros::init(argc, argv, "scanRegistration");
  ros::NodeHandle nh;
  int n_scans = 16;
  double minimum_range = 0.1;
  nh.param<int>("scan_line", n_scans, 16);
  nh.param<double>("minimum_range", minimum_range, 0.1);
  printf("scan line number %d \n", n_scans);
  if (n_scans != 16 && n_scans != 32 && n_scans != 64) {
    printf("only support velodyne with 16, 32 or 64 scan line!");
    return 0;
  }
  aloam_config cfg;
  aloam_default_config(&cfg);
  cfg.n_scans = n_scans;
  cfg.min_range = static_cast<float>(minimum_range);
  cfg.batch = 1;
  cfg.max_points = 400000;                                 // the reference's global arrays (:66-69)
  g_n_scans = n_scans;
  if (aloam_create_stages(&cfg, ALOAM_STAGE_REGISTRATION, &g_ctx) != ALOAM_OK) {   // this node hosts stage 1 only
    ROS_ERROR("aloam_create: %s", g_ctx ? aloam_last_error(g_ctx) : "out of memory");
    return 1;
  }
  int reference_sum_order = 0;                             // 1: pcl::VoxelGrid's own summation order (the reference's bits in the less-flat cloud; validation, slower)
  nh.param<int>("reference_sum_order", reference_sum_order, 0);
  if (reference_sum_order) aloam_set_voxel_sum_order(g_ctx, ALOAM_SUM_REFERENCE_ORDER);
  ros::Subscriber subLaserCloud = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_points", 100, laserCloudHandler);
  pubLaserCloud = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_2", 100);
  pubCornerPointsSharp = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 100);
  pubCornerPointsLessSharp = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 100);
  pubSurfPointsFlat = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_flat", 100);
  pubSurfPointsLessFlat = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 100);
  pubRemovePoints = nh.advertise<sensor_msgs::PointCloud2>("/laser_remove_points", 100);   // advertised, never published (:490)
  if (PUB_EACH_LINE)
    for (int i = 0; i < n_scans; i++) pubEachScan.push_back(nh.advertise<sensor_msgs::PointCloud2>("/laser_scanid_" + std::to_string(i), 100));   // :492-499
  ros::spin();
  aloam_destroy(g_ctx);
  g_ctx = nullptr;
  return 0;
}
