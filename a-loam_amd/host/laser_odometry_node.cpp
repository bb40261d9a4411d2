// a-loam_amd/host/laser_odometry_node.cpp — the `alaserOdometry` node on top of libaloam_mi355x.so.
// Same subscriptions, queues, stamp check, publications and frame ids as the reference (src/laserOdometry.cpp:186-263,
// 508-599); the solve (:265-506) and the cloud swap / kd-tree rebuild (:554-568) are calls into the C ABI.
#include <chrono>
#include <cstdio>
#include <mutex>
#include <queue>

#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>

#include "aloam_ros_common.hpp"

namespace {
aloam_ctx* g_ctx = nullptr;
int skipFrameNum = 5;
std::queue<sensor_msgs::PointCloud2ConstPtr> cornerSharpBuf, cornerLessSharpBuf, surfFlatBuf, surfLessFlatBuf, fullPointsBuf;
std::mutex mBuf;
}  // namespace
aloam_ctx* aloam_node_context() { return g_ctx; }

void laserCloudSharpHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); cornerSharpBuf.push(m); mBuf.unlock(); }
void laserCloudLessSharpHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); cornerLessSharpBuf.push(m); mBuf.unlock(); }
void laserCloudFlatHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); surfFlatBuf.push(m); mBuf.unlock(); }
void laserCloudLessFlatHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); surfLessFlatBuf.push(m); mBuf.unlock(); }
void laserCloudFullResHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); fullPointsBuf.push(m); mBuf.unlock(); }

int main(int argc, char** argv) {
  ros::init(argc, argv, "laserOdometry");
  ros::NodeHandle nh;
  nh.param<int>("mapping_skip_frame", skipFrameNum, 2);
  int n_scans = 64;
  nh.param<int>("scan_line", n_scans, 64);                 // only sizes the device buffers of this node
  printf("Mapping %d Hz \n", 10 / skipFrameNum);

  aloam_config cfg;
  aloam_default_config(&cfg);
  cfg.n_scans = n_scans;
  cfg.batch = 1;
  cfg.max_points = 400000;
  if (aloam_create_stages(&cfg, ALOAM_STAGE_ODOMETRY, &g_ctx) != ALOAM_OK) {   // this node hosts stage 2 only
    ROS_ERROR("aloam_create: %s", g_ctx ? aloam_last_error(g_ctx) : "out of memory");
    return 1;
  }

  ros::Subscriber subCornerPointsSharp = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_sharp", 100, laserCloudSharpHandler);
  ros::Subscriber subCornerPointsLessSharp = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_sharp", 100, laserCloudLessSharpHandler);
  ros::Subscriber subSurfPointsFlat = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_flat", 100, laserCloudFlatHandler);
  ros::Subscriber subSurfPointsLessFlat = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_less_flat", 100, laserCloudLessFlatHandler);
  ros::Subscriber subLaserCloudFullRes = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_2", 100, laserCloudFullResHandler);
  ros::Publisher pubLaserCloudCornerLast = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 100);
  ros::Publisher pubLaserCloudSurfLast = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 100);
  ros::Publisher pubLaserCloudFullRes = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_3", 100);
  ros::Publisher pubLaserOdometry = nh.advertise<nav_msgs::Odometry>("/laser_odom_to_init", 100);
  ros::Publisher pubLaserPath = nh.advertise<nav_msgs::Path>("/laser_odom_path", 100);

  nav_msgs::Path laserPath;
  int frameCount = 0, frameTotal = 0;
  ros::Rate rate(100);
  std::vector<float> sharp, lessSharp, flat, lessFlat, full;

  while (ros::ok()) {
    ros::spinOnce();
    if (!cornerSharpBuf.empty() && !cornerLessSharpBuf.empty() && !surfFlatBuf.empty() && !surfLessFlatBuf.empty() && !fullPointsBuf.empty()) {
      const double tSharp = cornerSharpBuf.front()->header.stamp.toSec(), tLessSharp = cornerLessSharpBuf.front()->header.stamp.toSec();
      const double tFlat = surfFlatBuf.front()->header.stamp.toSec(), tLessFlat = surfLessFlatBuf.front()->header.stamp.toSec();
      const double tFull = fullPointsBuf.front()->header.stamp.toSec();
      if (tSharp != tFull || tLessSharp != tFull || tFlat != tFull || tLessFlat != tFull) {
        printf("unsync messeage!");
        ROS_BREAK();
      }
      const auto t_whole = std::chrono::steady_clock::now();
      mBuf.lock();
      const int nSharp = aloam_host::msg_to_xyzi(*cornerSharpBuf.front(), &sharp); cornerSharpBuf.pop();
      const int nLessSharp = aloam_host::msg_to_xyzi(*cornerLessSharpBuf.front(), &lessSharp); cornerLessSharpBuf.pop();
      const int nFlat = aloam_host::msg_to_xyzi(*surfFlatBuf.front(), &flat); surfFlatBuf.pop();
      const int nLessFlat = aloam_host::msg_to_xyzi(*surfLessFlatBuf.front(), &lessFlat); surfLessFlatBuf.pop();
      const int nFull = aloam_host::msg_to_xyzi(*fullPointsBuf.front(), &full); fullPointsBuf.pop();
      mBuf.unlock();

      // first frame: initialisation only (systemInited, :267-271); afterwards two association + solve passes (:278-501)
      if (aloam_set_features(g_ctx, 0, sharp.data(), nSharp, lessSharp.data(), nLessSharp, flat.data(), nFlat, lessFlat.data(), nLessFlat) != ALOAM_OK ||
          aloam_odometry_step(g_ctx) != ALOAM_OK || aloam_synchronize(g_ctx) != ALOAM_OK) {
        ROS_WARN("odometry step failed: %s", aloam_last_error(g_ctx));
        continue;
      }
      double q_w[4], t_w[3], q_lc[4], t_lc[3];
      aloam_get_pose(g_ctx, 0, q_w, t_w, q_lc, t_lc);
      aloam_odom_stats st;
      if (frameTotal > 0 && aloam_get_odom_stats(g_ctx, 0, &st) == ALOAM_OK)
        for (int it = 0; it < 2; ++it)
          if (st.corner_corr[it] + st.plane_corr[it] < 10) printf("less correspondence! *************************************************\n");   // :488-491
      ++frameTotal;

      nav_msgs::Odometry laserOdometry;                    // :511-522
      laserOdometry.header.frame_id = "/camera_init";
      laserOdometry.child_frame_id = "/laser_odom";
      laserOdometry.header.stamp = ros::Time().fromSec(tLessFlat);
      laserOdometry.pose.pose.orientation.x = q_w[0]; laserOdometry.pose.pose.orientation.y = q_w[1];
      laserOdometry.pose.pose.orientation.z = q_w[2]; laserOdometry.pose.pose.orientation.w = q_w[3];
      laserOdometry.pose.pose.position.x = t_w[0]; laserOdometry.pose.pose.position.y = t_w[1]; laserOdometry.pose.pose.position.z = t_w[2];
      pubLaserOdometry.publish(laserOdometry);

      geometry_msgs::PoseStamped laserPose;                // :524-530
      laserPose.header = laserOdometry.header;
      laserPose.pose = laserOdometry.pose.pose;
      laserPath.header.stamp = laserOdometry.header.stamp;
      laserPath.poses.push_back(laserPose);
      laserPath.header.frame_id = "/camera_init";
      pubLaserPath.publish(laserPath);

      if (frameCount % skipFrameNum == 0) {                // :570-591
        frameCount = 0;
        const ros::Time stamp = ros::Time().fromSec(tLessFlat);
        pubLaserCloudCornerLast.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_CORNER_LAST, stamp, "/camera"));
        pubLaserCloudSurfLast.publish(aloam_host::cloud_msg(g_ctx, ALOAM_CLOUD_SURF_LAST, stamp, "/camera"));
        pubLaserCloudFullRes.publish(aloam_host::xyzi_to_msg(full.data(), nFull, stamp, "/camera"));
      }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_whole).count();
      printf("whole laserOdometry time %f ms \n \n", ms);
      if (ms > 100) ROS_WARN("odometry process over 100ms");   // :593-595
      frameCount++;
    }
    rate.sleep();
  }
  aloam_destroy(g_ctx);
  g_ctx = nullptr;
  return 0;
}
