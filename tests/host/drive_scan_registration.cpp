// tests/host/drive_scan_registration.cpp — TEST DRIVER for a-loam_amd/host/scan_registration_node.cpp.  Same file protocol
// as oracle/ref_shim/driver_scan_registration.cpp (which drives the reference's own node), so tests can run both on the same
// sweeps and compare what they publish.  The ROS API is the message-capturing stand-in under oracle/ref_shim/include.
#include <cmath>

#include "ref_io.hpp"

int node_main(int argc, char** argv);

int main(int argc, char** argv) {
  ref_io::must(argc == 5, "usage: drive_scan_registration <scan_line> <minimum_range> <in.bin> <out.bin>");
  ref_shim::params()["scan_line"] = std::atof(argv[1]);
  ref_shim::params()["minimum_range"] = std::atof(argv[2]);
  FILE* fin = std::fopen(argv[3], "rb");
  FILE* fout = std::fopen(argv[4], "wb");
  ref_io::must(fin && fout, "cannot open files");
  const int n_frames = ref_io::read_i32(fin);
  int frame = 0;
  const char* topics[5] = {"/velodyne_cloud_2", "/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat"};
  ref_shim::ok_hook() = [&]() { return frame < n_frames; };
  ref_shim::spin_hook() = [&]() {
    const std::vector<float> v = ref_io::read_cloud(fin);
    bool dense = true;
    for (float x : v) dense = dense && std::isfinite(x);
    ref_shim::deliver("/velodyne_points", ref_io::make_msg(v, 0.1 * frame, dense));
    auto& pub = ref_shim::published<sensor_msgs::PointCloud2>();
    for (const char* t : topics) {
      ref_io::must(pub[t].size() == static_cast<size_t>(frame) + 1, "the node did not publish one message per sweep");
      ref_io::must(pub[t].back().header.stamp.toSec() == 0.1 * frame && pub[t].back().header.frame_id == "/camera_init", "stamp / frame id");
      ref_io::write_cloud(fout, pub[t].back());
    }
    ref_io::must(pub.count("/laser_remove_points") == 1 && pub["/laser_remove_points"].empty(), "/laser_remove_points must be advertised and silent");
    ref_io::write_i32(fout, 0);                                // no per-point arrays from this node
    ++frame;
  };
  const int rc = node_main(argc, argv);
  std::fclose(fout);
  return rc;
}
