// tests/host/drive_laser_odometry.cpp — TEST DRIVER for a-loam_amd/host/laser_odometry_node.cpp (protocol of
// oracle/ref_shim/driver_laser_odometry.cpp; para_q / para_t / correspondence counts are read through the C ABI).
#include "aloam_mi355x.h"
#include "ref_io.hpp"
#include <vector>

int node_main(int argc, char** argv);
aloam_ctx* aloam_node_context();

int main(int argc, char** argv) {
  ref_io::must(argc == 4, "usage: drive_laser_odometry <scan_line> <in.bin> <out.bin>");
  ref_shim::params()["mapping_skip_frame"] = 1;
  ref_shim::params()["scan_line"] = std::atof(argv[1]);
  FILE* fin = std::fopen(argv[2], "rb");
  FILE* fout = std::fopen(argv[3], "wb");
  ref_io::must(fin && fout, "cannot open files");
  const int n_frames = ref_io::read_i32(fin);
  int delivered = 0, flushed = 0;
  auto flush = [&]() {
    auto& odom = ref_shim::published<nav_msgs::Odometry>()["/laser_odom_to_init"];
    auto& pc = ref_shim::published<sensor_msgs::PointCloud2>();
    while (flushed < static_cast<int>(odom.size())) {
      const nav_msgs::Odometry& o = odom[flushed];
      ref_io::must(o.header.frame_id == "/camera_init" && o.child_frame_id == "/laser_odom" && o.header.stamp.toSec() == 0.1 * flushed, "odometry header");
      double q_w[4], t_w[3], q_lc[4], t_lc[3];
      aloam_odom_stats st;
      aloam_get_pose(aloam_node_context(), 0, q_w, t_w, q_lc, t_lc);
      aloam_get_odom_stats(aloam_node_context(), 0, &st);
      const double rec[14] = {o.pose.pose.orientation.x, o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.pose.pose.orientation.w,
                              o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z,
                              q_lc[0], q_lc[1], q_lc[2], q_lc[3], t_lc[0], t_lc[1], t_lc[2]};
      ref_io::write_f64(fout, rec, 14);
      ref_io::write_i32(fout, flushed == 0 ? 0 : st.corner_corr[1]); ref_io::write_i32(fout, flushed == 0 ? 0 : st.plane_corr[1]);
      ref_io::must(pc["/laser_cloud_corner_last"].size() == static_cast<size_t>(flushed) + 1 && pc["/velodyne_cloud_3"].size() == static_cast<size_t>(flushed) + 1,
                   "corner_last / cloud_3 not published every frame");
      ref_io::must(pc["/laser_cloud_corner_last"][flushed].header.frame_id == "/camera", "frame id of the last clouds");
      ref_io::write_cloud(fout, pc["/laser_cloud_corner_last"][flushed]);
      ref_io::write_cloud(fout, pc["/laser_cloud_surf_last"][flushed]);
      // the factors of the frame's last solve, in the format the reference's driver dumps them (constructor arguments as float64)
      const int cap_e = 128 * 12, cap_p = 128 * 24;
      std::vector<float> e(9 * cap_e), p(12 * cap_p);
      std::vector<int> eq(cap_e), pq(cap_p);
      int ne = 0, np = 0;
      if (flushed > 0) ref_io::must(aloam_get_correspondences(aloam_node_context(), 0, e.data(), cap_e, &ne, eq.data(), p.data(), cap_p, &np, pq.data()) == ALOAM_OK, "aloam_get_correspondences");
      ref_io::write_i32(fout, ne); ref_io::write_i32(fout, np);
      std::vector<double> de(e.begin(), e.begin() + 9 * ne), dp(p.begin(), p.begin() + 12 * np);
      if (ne) ref_io::write_f64(fout, de.data(), 9 * ne);
      if (np) ref_io::write_f64(fout, dp.data(), 12 * np);
      ++flushed;
    }
  };
  ref_shim::ok_hook() = [&]() { flush(); return flushed < n_frames; };
  ref_shim::spin_hook() = [&]() {
    if (delivered >= n_frames) return;
    const double stamp = 0.1 * delivered;
    const char* topics[5] = {"/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat", "/velodyne_cloud_2"};
    for (const char* t : topics) ref_shim::deliver(t, ref_io::make_msg(ref_io::read_cloud(fin), stamp));
    ++delivered;
  };
  const int rc = node_main(argc, argv);
  std::fclose(fout);
  return rc;
}
