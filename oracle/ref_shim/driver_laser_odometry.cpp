// oracle/ref_shim/driver_laser_odometry.cpp — runs the reference's OWN odometry translation unit
// (/root/reference/src/laserOdometry.cpp + src/lidarFactor.hpp, compiled in place with -Dmain=ref_node_main against
// the stand-in headers under ref_shim/include) on feature clouds read from a file.  TEST INFRASTRUCTURE ONLY.
//   usage: ref_laser_odometry <in.bin> <out.bin>
//   in : int32 n_frames, then per frame five cloud records: sharp, less_sharp, flat, less_flat, full (/velodyne_cloud_2)
//   out: per frame 14 float64 (q_w_curr xyzw, t_w_curr, para_q xyzw, para_t), 2 int32 (corner_correspondence,
//        plane_correspondence of the last outer iteration), then the published /laser_cloud_corner_last and
//        /laser_cloud_surf_last records (mapping_skip_frame is set to 1 so that every frame is published)
#include "ref_io.hpp"

int ref_node_main(int argc, char** argv);                                     // = main() of the reference node
extern double para_q[4];                                                      // reference src/laserOdometry.cpp:97-98
extern double para_t[3];
extern int corner_correspondence, plane_correspondence;                       // :62

int main(int argc, char** argv) {
  ref_io::must(argc == 3, "usage: ref_laser_odometry <in.bin> <out.bin>");
  ref_shim::params()["mapping_skip_frame"] = 1;
  FILE* fin = std::fopen(argv[1], "rb");
  FILE* fout = std::fopen(argv[2], "wb");
  ref_io::must(fin && fout, "cannot open files");
  const int n_frames = ref_io::read_i32(fin);
  int delivered = 0, flushed = 0;
  auto flush = [&]() {                                                        // results of the frames processed so far
    auto& odom = ref_shim::published<nav_msgs::Odometry>()["/laser_odom_to_init"];
    auto& pc = ref_shim::published<sensor_msgs::PointCloud2>();
    while (flushed < static_cast<int>(odom.size())) {
      const nav_msgs::Odometry& o = odom[flushed];
      const double rec[14] = {o.pose.pose.orientation.x, o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.pose.pose.orientation.w,
                              o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z,
                              para_q[0], para_q[1], para_q[2], para_q[3], para_t[0], para_t[1], para_t[2]};
      ref_io::write_f64(fout, rec, 14);
      ref_io::write_i32(fout, corner_correspondence); ref_io::write_i32(fout, plane_correspondence);
      ref_io::must(pc["/laser_cloud_corner_last"].size() == static_cast<size_t>(flushed) + 1, "corner_last not published every frame");
      ref_io::write_cloud(fout, pc["/laser_cloud_corner_last"][flushed]);
      ref_io::write_cloud(fout, pc["/laser_cloud_surf_last"][flushed]);
      ++flushed;
    }
  };
  ref_shim::ok_hook() = [&]() { flush(); return flushed < n_frames; };
  ref_shim::spin_hook() = [&]() {                                             // ros::spinOnce(): one synchronised frame per turn
    if (delivered >= n_frames) return;
    const double stamp = 0.1 * delivered;
    const char* topics[5] = {"/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat", "/velodyne_cloud_2"};
    for (const char* t : topics) ref_shim::deliver(t, ref_io::make_msg(ref_io::read_cloud(fin), stamp));
    ++delivered;
  };
  ref_node_main(argc, argv);
  flush();
  std::fclose(fout);
  return 0;
}
