// a-loam_amd/host/laser_mapping_node.cpp — the `alaserMapping` node on top of libaloam_mi355x.so.
// Same subscriptions, queue alignment, publications, frame ids and tf as the reference (src/laserMapping.cpp:175-304,
// 803-938); the body of process() between the queue handling and the publishers (:307-802) is calls into the C ABI.  The
// frames are processed in the spin loop instead of a second thread: with the work on the GPU the node is never behind, so
// the reference's "drop queued frames" branch (:299-303) is kept but does not fire in practice.
#include <cstdio>
#include <mutex>
#include <queue>

#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_datatypes.h>

#include "aloam_ros_common.hpp"

namespace {
aloam_ctx* g_ctx = nullptr;
std::queue<sensor_msgs::PointCloud2ConstPtr> cornerLastBuf, surfLastBuf, fullResBuf;
std::queue<nav_msgs::Odometry::ConstPtr> odometryBuf;
std::mutex mBuf;
ros::Publisher pubLaserCloudSurround, pubLaserCloudMap, pubLaserCloudFullRes, pubOdomAftMapped, pubOdomAftMappedHighFrec, pubLaserAfterMappedPath;
nav_msgs::Path laserAfterMappedPath;
double q_wmap_wodom[4] = {0, 0, 0, 1}, t_wmap_wodom[3] = {0, 0, 0};
int frameCount = 0;

void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void quat_rot(const double q[4], const double v[3], double o[3]) {
  double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
  ux += ux; uy += uy; uz += uz;
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
nav_msgs::Odometry make_odom(const double q[4], const double t[3], const ros::Time& stamp) {
  nav_msgs::Odometry o;
  o.header.frame_id = "/camera_init";
  o.child_frame_id = "/aft_mapped";
  o.header.stamp = stamp;
  o.pose.pose.orientation.x = q[0]; o.pose.pose.orientation.y = q[1]; o.pose.pose.orientation.z = q[2]; o.pose.pose.orientation.w = q[3];
  o.pose.pose.position.x = t[0]; o.pose.pose.position.y = t[1]; o.pose.pose.position.z = t[2];
  return o;
}
sensor_msgs::PointCloud2 cubes_msg(const int* cubes, int n_cubes, const ros::Time& stamp) {   // corner then surf points per cube (:808-813, :825-829)
  std::vector<float> all;
  std::vector<int> cnt(21 * 21 * 11);
  for (int i = 0; i < n_cubes; ++i)
    for (int cls = 0; cls < 2; ++cls) {
      std::vector<float> v(4);
      const int n = aloam_get_map_cube(g_ctx, 0, cls, cubes[i], v.data(), 0);
      if (n <= 0) continue;
      v.resize(4 * static_cast<size_t>(n));
      aloam_get_map_cube(g_ctx, 0, cls, cubes[i], v.data(), n);
      all.insert(all.end(), v.begin(), v.end());
    }
  return aloam_host::xyzi_to_msg(all.data(), static_cast<int>(all.size() / 4), stamp, "/camera_init");
}
}  // namespace
aloam_ctx* aloam_node_context() { return g_ctx; }

void laserCloudCornerLastHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); cornerLastBuf.push(m); mBuf.unlock(); }
void laserCloudSurfLastHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); surfLastBuf.push(m); mBuf.unlock(); }
void laserCloudFullResHandler(const sensor_msgs::PointCloud2ConstPtr& m) { mBuf.lock(); fullResBuf.push(m); mBuf.unlock(); }
void laserOdometryHandler(const nav_msgs::Odometry::ConstPtr& laserOdometry) {   // :197-228: queue + high-frequency corrected pose
  mBuf.lock();
  odometryBuf.push(laserOdometry);
  mBuf.unlock();
  const double qo[4] = {laserOdometry->pose.pose.orientation.x, laserOdometry->pose.pose.orientation.y, laserOdometry->pose.pose.orientation.z,
                        laserOdometry->pose.pose.orientation.w};
  const double to[3] = {laserOdometry->pose.pose.position.x, laserOdometry->pose.pose.position.y, laserOdometry->pose.pose.position.z};
  double q[4], t[3];
  quat_mul(q_wmap_wodom, qo, q);
  quat_rot(q_wmap_wodom, to, t);
  for (int k = 0; k < 3; ++k) t[k] += t_wmap_wodom[k];
  pubOdomAftMappedHighFrec.publish(make_odom(q, t, laserOdometry->header.stamp));
}

static void process() {
  std::vector<float> corner, surf, full;
  while (!cornerLastBuf.empty() && !surfLastBuf.empty() && !fullResBuf.empty() && !odometryBuf.empty()) {
    mBuf.lock();
    while (!odometryBuf.empty() && odometryBuf.front()->header.stamp.toSec() < cornerLastBuf.front()->header.stamp.toSec()) odometryBuf.pop();
    if (odometryBuf.empty()) { mBuf.unlock(); break; }
    while (!surfLastBuf.empty() && surfLastBuf.front()->header.stamp.toSec() < cornerLastBuf.front()->header.stamp.toSec()) surfLastBuf.pop();
    if (surfLastBuf.empty()) { mBuf.unlock(); break; }
    while (!fullResBuf.empty() && fullResBuf.front()->header.stamp.toSec() < cornerLastBuf.front()->header.stamp.toSec()) fullResBuf.pop();
    if (fullResBuf.empty()) { mBuf.unlock(); break; }
    const double tCorner = cornerLastBuf.front()->header.stamp.toSec(), tSurf = surfLastBuf.front()->header.stamp.toSec();
    const double tFull = fullResBuf.front()->header.stamp.toSec(), tOdom = odometryBuf.front()->header.stamp.toSec();
    if (tCorner != tOdom || tSurf != tOdom || tFull != tOdom) {
      printf("time corner %f surf %f full %f odom %f \n", tCorner, tSurf, tFull, tOdom);
      printf("unsync messeage!");
      mBuf.unlock();
      break;
    }
    const int nCorner = aloam_host::msg_to_xyzi(*cornerLastBuf.front(), &corner); cornerLastBuf.pop();
    const int nSurf = aloam_host::msg_to_xyzi(*surfLastBuf.front(), &surf); surfLastBuf.pop();
    const int nFull = aloam_host::msg_to_xyzi(*fullResBuf.front(), &full); fullResBuf.pop();
    const nav_msgs::Odometry::ConstPtr od = odometryBuf.front();
    odometryBuf.pop();
    while (!cornerLastBuf.empty()) { cornerLastBuf.pop(); printf("drop lidar frame in mapping for real time performance \n"); }   // :299-303
    mBuf.unlock();

    const double qo[4] = {od->pose.pose.orientation.x, od->pose.pose.orientation.y, od->pose.pose.orientation.z, od->pose.pose.orientation.w};
    const double to[3] = {od->pose.pose.position.x, od->pose.pose.position.y, od->pose.pose.position.z};
    const double id_q[4] = {0, 0, 0, 1}, id_t[3] = {0, 0, 0};
    double t_guess[3];                                     // t_w_curr of transformAssociateToMap (:142-146): picks the centre cube
    quat_rot(q_wmap_wodom, to, t_guess);
    for (int k = 0; k < 3; ++k) t_guess[k] += t_wmap_wodom[k];
    if (aloam_set_last(g_ctx, 0, corner.data(), nCorner, surf.data(), nSurf) != ALOAM_OK || aloam_set_full_cloud(g_ctx, 0, full.data(), nFull) != ALOAM_OK ||
        aloam_set_state(g_ctx, 0, id_q, id_t, qo, to) != ALOAM_OK || aloam_mapping_step(g_ctx) != ALOAM_OK) {
      ROS_WARN("mapping step failed: %s", aloam_last_error(g_ctx));
      continue;
    }
    const int rc_sync = aloam_synchronize(g_ctx);
    if (rc_sync == ALOAM_E_CAPACITY) ROS_WARN("mapping: %s", aloam_last_error(g_ctx));   // the step has run; only points that did not fit are missing from the map
    else if (rc_sync != ALOAM_OK) { ROS_WARN("mapping step failed: %s", aloam_last_error(g_ctx)); continue; }
    {
      int info[16];
      if (aloam_get_map_info(g_ctx, 0, info) == ALOAM_OK && !(info[4] > 10 && info[5] > 50)) ROS_WARN("time Map corner and surf num are not enough");   // :554,730-733
    }
    double q_w[4], t_w[3];
    aloam_get_map_pose(g_ctx, 0, q_w, t_w, q_wmap_wodom, t_wmap_wodom);
    const ros::Time stamp = ros::Time().fromSec(tOdom);

    if (frameCount % 5 == 0 || frameCount % 20 == 0) {     // :803-834
      int info[16];
      aloam_get_map_info(g_ctx, 0, info);
      std::vector<int> counts(21 * 21 * 11);
      if (frameCount % 5 == 0) {
        // laserCloudSurroundInd = the valid 5 x 5 x 3 window around the centre cube (:512-529)
        const int cI = static_cast<int>((t_guess[0] + 25.0) / 50.0) + info[0] - (t_guess[0] + 25.0 < 0),
                  cJ = static_cast<int>((t_guess[1] + 25.0) / 50.0) + info[1] - (t_guess[1] + 25.0 < 0),
                  cK = static_cast<int>((t_guess[2] + 25.0) / 50.0) + info[2] - (t_guess[2] + 25.0 < 0);
        std::vector<int> cubes;
        for (int i = cI - 2; i <= cI + 2; i++) for (int j = cJ - 2; j <= cJ + 2; j++) for (int k = cK - 1; k <= cK + 1; k++)
          if (i >= 0 && i < 21 && j >= 0 && j < 21 && k >= 0 && k < 11) cubes.push_back(i + 21 * j + 21 * 21 * k);
        pubLaserCloudSurround.publish(cubes_msg(cubes.data(), static_cast<int>(cubes.size()), stamp));
      }
      if (frameCount % 20 == 0) {
        std::vector<int> cubes(21 * 21 * 11);
        for (int i = 0; i < 21 * 21 * 11; ++i) cubes[i] = i;
        pubLaserCloudMap.publish(cubes_msg(cubes.data(), static_cast<int>(cubes.size()), stamp));
      }
    }
    {  // /velodyne_cloud_registered (:836-846)
      std::vector<float> reg(4 * static_cast<size_t>(nFull > 0 ? nFull : 1));
      const int n = aloam_get_map_cloud(g_ctx, 0, ALOAM_MAP_REGISTERED, reg.data(), nFull);
      pubLaserCloudFullRes.publish(aloam_host::xyzi_to_msg(reg.data(), n, stamp, "/camera_init"));
    }
    const nav_msgs::Odometry odomAftMapped = make_odom(q_w, t_w, stamp);                  // :851-863
    pubOdomAftMapped.publish(odomAftMapped);
    geometry_msgs::PoseStamped pose;                                                      // :865-871
    pose.header = odomAftMapped.header;
    pose.pose = odomAftMapped.pose.pose;
    laserAfterMappedPath.header.stamp = odomAftMapped.header.stamp;
    laserAfterMappedPath.header.frame_id = "/camera_init";
    laserAfterMappedPath.poses.push_back(pose);
    pubLaserAfterMappedPath.publish(laserAfterMappedPath);
    static tf::TransformBroadcaster br;                                                   // :873-887
    tf::Transform transform;
    tf::Quaternion q;
    transform.setOrigin(tf::Vector3(t_w[0], t_w[1], t_w[2]));
    q.setW(q_w[3]); q.setX(q_w[0]); q.setY(q_w[1]); q.setZ(q_w[2]);
    transform.setRotation(q);
    br.sendTransform(tf::StampedTransform(transform, odomAftMapped.header.stamp, "/camera_init", "/aft_mapped"));
    frameCount++;
  }
}

int main(int argc, char** argv) {
  ros::init(argc, argv, "laserMapping");
  ros::NodeHandle nh;
  float lineRes = 0, planeRes = 0;
  nh.param<float>("mapping_line_resolution", lineRes, 0.4);
  nh.param<float>("mapping_plane_resolution", planeRes, 0.8);
  int n_scans = 64, pool_points = 1 << 17;
  nh.param<int>("scan_line", n_scans, 64);
  nh.param<int>("map_pool_points", pool_points, 1 << 17);  // where the device-resident map starts (points per feature class); it doubles as the map grows, like the reference's std::vector cubes
  printf("line resolution %f plane resolution %f \n", lineRes, planeRes);

  aloam_config cfg;
  aloam_default_config(&cfg);
  cfg.n_scans = n_scans;
  cfg.batch = 1;
  cfg.max_points = 400000;
  if (aloam_create_stages(&cfg, ALOAM_STAGE_MAPPING, &g_ctx) != ALOAM_OK || aloam_mapping_enable(g_ctx, lineRes, planeRes, pool_points) != ALOAM_OK) {   // this node hosts stage 3 only
    ROS_ERROR("aloam set-up: %s", g_ctx ? aloam_last_error(g_ctx) : "out of memory");
    return 1;
  }
  int reference_sum_order = 0;                             // 1: pcl::VoxelGrid's own summation order in the stack / cube filters (validation: ~80 ms per HDL-64 frame)
  nh.param<int>("reference_sum_order", reference_sum_order, 0);
  if (reference_sum_order) aloam_set_voxel_sum_order(g_ctx, ALOAM_SUM_REFERENCE_ORDER);
  ros::Subscriber subLaserCloudCornerLast = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_corner_last", 100, laserCloudCornerLastHandler);
  ros::Subscriber subLaserCloudSurfLast = nh.subscribe<sensor_msgs::PointCloud2>("/laser_cloud_surf_last", 100, laserCloudSurfLastHandler);
  ros::Subscriber subLaserOdometry = nh.subscribe<nav_msgs::Odometry>("/laser_odom_to_init", 100, laserOdometryHandler);
  ros::Subscriber subLaserCloudFullRes = nh.subscribe<sensor_msgs::PointCloud2>("/velodyne_cloud_3", 100, laserCloudFullResHandler);
  pubLaserCloudSurround = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surround", 100);
  pubLaserCloudMap = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_map", 100);
  pubLaserCloudFullRes = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_registered", 100);
  pubOdomAftMapped = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init", 100);
  pubOdomAftMappedHighFrec = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init_high_frec", 100);
  pubLaserAfterMappedPath = nh.advertise<nav_msgs::Path>("/aft_mapped_path", 100);

  ros::Rate rate(500);                                     // the reference's worker polls every 2 ms (:890-891)
  while (ros::ok()) {
    ros::spinOnce();
    process();
    rate.sleep();
  }
  aloam_destroy(g_ctx);
  g_ctx = nullptr;
  return 0;
}
