// tests/host/drive_laser_mapping.cpp — TEST DRIVER for a-loam_amd/host/laser_mapping_node.cpp: same input / output files as
// oracle/ref_shim/driver_laser_mapping.cpp (which drives the reference's own node); the cube map is dumped through the C ABI.
#include "aloam_mi355x.h"
#include "ref_io.hpp"

int node_main(int argc, char** argv);
aloam_ctx* aloam_node_context();

int main(int argc, char** argv) {
  ref_io::must(argc == 6, "usage: drive_laser_mapping <scan_line> <line_res> <plane_res> <in.bin> <out.bin>");
  ref_shim::params()["scan_line"] = std::atof(argv[1]);
  ref_shim::params()["mapping_line_resolution"] = std::atof(argv[2]);
  ref_shim::params()["mapping_plane_resolution"] = std::atof(argv[3]);
  ref_shim::params()["map_pool_points"] = 262144;
  FILE* fin = std::fopen(argv[4], "rb");
  FILE* fout = std::fopen(argv[5], "wb");
  ref_io::must(fin && fout, "cannot open files");
  const int n_frames = ref_io::read_i32(fin);
  int delivered = 0, flushed = 0;
  auto flush = [&]() {
    auto& aft = ref_shim::published<nav_msgs::Odometry>()["/aft_mapped_to_init"];
    while (flushed < static_cast<int>(aft.size())) {
      ref_io::must(flushed == delivered - 1, "one frame at a time");
      const nav_msgs::Odometry& o = aft[flushed];
      ref_io::must(o.header.frame_id == "/camera_init" && o.child_frame_id == "/aft_mapped" && o.header.stamp.toSec() == 0.1 * flushed, "aft_mapped header");
      double q_w[4], t_w[3], q_mo[4], t_mo[3];
      aloam_ctx* ctx = aloam_node_context();
      aloam_get_map_pose(ctx, 0, q_w, t_w, q_mo, t_mo);
      const double rec[14] = {o.pose.pose.orientation.x, o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.pose.pose.orientation.w,
                              o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z, q_mo[0], q_mo[1], q_mo[2], q_mo[3], t_mo[0], t_mo[1], t_mo[2]};
      ref_io::write_f64(fout, rec, 14);
      ref_io::write_cloud(fout, ref_shim::published<sensor_msgs::PointCloud2>()["/velodyne_cloud_registered"].back());
      int info[16];
      aloam_get_map_info(ctx, 0, info);
      ref_io::write_i32(fout, info[0]); ref_io::write_i32(fout, info[1]); ref_io::write_i32(fout, info[2]);
      std::vector<int> cnt(21 * 21 * 11);
      for (int cls = 0; cls < 2; ++cls) {
        aloam_map_cube_counts(ctx, 0, cls, cnt.data());
        int n = 0;
        for (int c : cnt) n += c > 0;
        ref_io::write_i32(fout, n);
        for (int i = 0; i < 21 * 21 * 11; ++i) {
          if (cnt[i] <= 0) continue;
          ref_io::write_i32(fout, i);
          std::vector<float> v(4 * static_cast<size_t>(cnt[i]));
          aloam_get_map_cube(ctx, 0, cls, i, v.data(), cnt[i]);
          ref_io::write_i32(fout, cnt[i]);
          std::fwrite(v.data(), 16, cnt[i], fout);
        }
      }
      ++flushed;
    }
  };
  ref_shim::ok_hook() = [&]() { flush(); return flushed < n_frames; };
  ref_shim::spin_hook() = [&]() {
    if (delivered >= n_frames || delivered > flushed) return;
    const double stamp = 0.1 * delivered;
    double pose[7];
    ref_io::must(std::fread(pose, 8, 7, fin) == 7, "short read");
    nav_msgs::Odometry odom;
    odom.header.stamp.fromSec(stamp);
    odom.pose.pose.orientation.x = pose[0]; odom.pose.pose.orientation.y = pose[1]; odom.pose.pose.orientation.z = pose[2]; odom.pose.pose.orientation.w = pose[3];
    odom.pose.pose.position.x = pose[4]; odom.pose.pose.position.y = pose[5]; odom.pose.pose.position.z = pose[6];
    ref_shim::deliver("/laser_cloud_corner_last", ref_io::make_msg(ref_io::read_cloud(fin), stamp));
    ref_shim::deliver("/laser_cloud_surf_last", ref_io::make_msg(ref_io::read_cloud(fin), stamp));
    ref_shim::deliver("/velodyne_cloud_3", ref_io::make_msg(ref_io::read_cloud(fin), stamp));
    ref_shim::deliver("/laser_odom_to_init", odom);
    ++delivered;
  };
  const int rc = node_main(argc, argv);
  flush();
  std::fclose(fout);
  return rc;
}
