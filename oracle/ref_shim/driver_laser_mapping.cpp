// oracle/ref_shim/driver_laser_mapping.cpp — runs the reference's OWN mapping translation unit
// (/root/reference/src/laserMapping.cpp + src/lidarFactor.hpp, compiled in place against the stand-in headers under
// ref_shim/include, with mapping_prefix.hpp force-included) frame by frame.  TEST INFRASTRUCTURE ONLY.
// The node's main() is not called (it parks the work in a std::thread and blocks in ros::spin()); its few set-up lines
// (reference src/laserMapping.cpp:895-931) are reproduced here on the node's own globals, then every frame is pushed
// through the node's subscriber callbacks and process() is called directly, so no frame is ever dropped (quirk 10 of
// the live node drops queued frames; offline parity runs no-drop).
//   usage: ref_laser_mapping <mapping_line_resolution> <mapping_plane_resolution> <in.bin> <out.bin>
//   in : int32 n_frames, then per frame: 7 float64 odometry pose (q xyzw, t) + three cloud records
//        (/laser_cloud_corner_last, /laser_cloud_surf_last, /velodyne_cloud_3)
//   out: per frame 14 float64 (q_w_curr xyzw, t_w_curr, q_wmap_wodom xyzw, t_wmap_wodom), the published
//        /velodyne_cloud_registered record, 3 int32 (laserCloudCenWidth/Height/Depth), then the whole cube map:
//        int32 n_c, n_c x (int32 cube, cloud record) for the non-empty corner cubes, same for the surf cubes
#include "ref_io.hpp"
#include "shim/eigen_shim.hpp"

typedef pcl::PointXYZI PointType;
void process();                                                                                   // reference src/laserMapping.cpp:231
void laserCloudCornerLastHandler(const sensor_msgs::PointCloud2ConstPtr&);                        // :175
void laserCloudSurfLastHandler(const sensor_msgs::PointCloud2ConstPtr&);                          // :182
void laserCloudFullResHandler(const sensor_msgs::PointCloud2ConstPtr&);                           // :189
void laserOdometryHandler(const nav_msgs::Odometry::ConstPtr&);                                   // :197
extern pcl::VoxelGrid<PointType> downSizeFilterCorner, downSizeFilterSurf;                        // :125-126
extern ros::Publisher pubLaserCloudSurround, pubLaserCloudMap, pubLaserCloudFullRes, pubOdomAftMapped, pubOdomAftMappedHighFrec, pubLaserAfterMappedPath;
extern pcl::PointCloud<PointType>::Ptr laserCloudCornerArray[], laserCloudSurfArray[];            // :102-103
extern int laserCloudCenWidth, laserCloudCenHeight, laserCloudCenDepth;                           // :72-74
extern Eigen::Quaterniond q_wmap_wodom;                                                           // :116-117
extern Eigen::Vector3d t_wmap_wodom;
static const int kCubes = 21 * 21 * 11;

int main(int argc, char** argv) {
  ref_io::must(argc == 5, "usage: ref_laser_mapping <line_res> <plane_res> <in.bin> <out.bin>");
  const float lineRes = static_cast<float>(std::atof(argv[1])), planeRes = static_cast<float>(std::atof(argv[2]));
  FILE* fin = std::fopen(argv[3], "rb");
  FILE* fout = std::fopen(argv[4], "wb");
  ref_io::must(fin && fout, "cannot open files");
  // set-up lines of the node's main()
  downSizeFilterCorner.setLeafSize(lineRes, lineRes, lineRes);
  downSizeFilterSurf.setLeafSize(planeRes, planeRes, planeRes);
  ros::NodeHandle nh;
  pubLaserCloudSurround = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_surround", 100);
  pubLaserCloudMap = nh.advertise<sensor_msgs::PointCloud2>("/laser_cloud_map", 100);
  pubLaserCloudFullRes = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_cloud_registered", 100);
  pubOdomAftMapped = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init", 100);
  pubOdomAftMappedHighFrec = nh.advertise<nav_msgs::Odometry>("/aft_mapped_to_init_high_frec", 100);
  pubLaserAfterMappedPath = nh.advertise<nav_msgs::Path>("/aft_mapped_path", 100);
  for (int i = 0; i < kCubes; i++) { laserCloudCornerArray[i].reset(new pcl::PointCloud<PointType>()); laserCloudSurfArray[i].reset(new pcl::PointCloud<PointType>()); }

  const int n_frames = ref_io::read_i32(fin);
  for (int k = 0; k < n_frames; ++k) {
    const double stamp = 0.1 * k;
    double pose[7];
    ref_io::must(std::fread(pose, 8, 7, fin) == 7, "short read");
    auto odom = std::make_shared<nav_msgs::Odometry>();
    odom->header.stamp.fromSec(stamp);
    odom->pose.pose.orientation.x = pose[0]; odom->pose.pose.orientation.y = pose[1]; odom->pose.pose.orientation.z = pose[2]; odom->pose.pose.orientation.w = pose[3];
    odom->pose.pose.position.x = pose[4]; odom->pose.pose.position.y = pose[5]; odom->pose.pose.position.z = pose[6];
    auto mk = [&](const std::vector<float>& v) { return std::make_shared<const sensor_msgs::PointCloud2>(ref_io::make_msg(v, stamp)); };
    laserCloudCornerLastHandler(mk(ref_io::read_cloud(fin)));
    laserCloudSurfLastHandler(mk(ref_io::read_cloud(fin)));
    laserCloudFullResHandler(mk(ref_io::read_cloud(fin)));
    laserOdometryHandler(odom);
    process();
    auto& aft = ref_shim::published<nav_msgs::Odometry>()["/aft_mapped_to_init"];
    ref_io::must(aft.size() == static_cast<size_t>(k) + 1, "process() did not handle the frame");
    const nav_msgs::Odometry& o = aft.back();
    const double rec[14] = {o.pose.pose.orientation.x, o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.pose.pose.orientation.w,
                            o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z,
                            q_wmap_wodom.x(), q_wmap_wodom.y(), q_wmap_wodom.z(), q_wmap_wodom.w(), t_wmap_wodom.x(), t_wmap_wodom.y(), t_wmap_wodom.z()};
    ref_io::write_f64(fout, rec, 14);
    ref_io::write_cloud(fout, ref_shim::published<sensor_msgs::PointCloud2>()["/velodyne_cloud_registered"].back());
    ref_io::write_i32(fout, laserCloudCenWidth); ref_io::write_i32(fout, laserCloudCenHeight); ref_io::write_i32(fout, laserCloudCenDepth);
    for (int cls = 0; cls < 2; ++cls) {
      pcl::PointCloud<PointType>::Ptr* arr = cls == 0 ? laserCloudCornerArray : laserCloudSurfArray;
      int n = 0;
      for (int i = 0; i < kCubes; ++i) n += !arr[i]->points.empty();
      ref_io::write_i32(fout, n);
      for (int i = 0; i < kCubes; ++i) {
        if (arr[i]->points.empty()) continue;
        ref_io::write_i32(fout, i);
        sensor_msgs::PointCloud2 m;
        pcl::toROSMsg(*arr[i], m);
        ref_io::write_cloud(fout, m);
      }
    }
    // keep the capture lists short
    for (auto& kv : ref_shim::published<sensor_msgs::PointCloud2>()) if (kv.second.size() > 1) kv.second.erase(kv.second.begin(), kv.second.end() - 1);
  }
  std::fclose(fout);
  return 0;
}
