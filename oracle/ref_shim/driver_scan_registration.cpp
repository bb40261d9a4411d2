// oracle/ref_shim/driver_scan_registration.cpp — runs the reference's OWN scan-registration translation unit
// (/root/reference/src/scanRegistration.cpp, compiled in place with -Dmain=ref_node_main against the stand-in
// headers under ref_shim/include) on sweeps read from a file.  TEST INFRASTRUCTURE ONLY.
//   usage: ref_scan_registration <scan_line> <minimum_range> <in.bin> <out.bin>
//   in : int32 n_frames, then per frame one cloud record (x, y, z, ignored)
//   out: per frame the five published clouds (/velodyne_cloud_2, /laser_cloud_sharp, /laser_cloud_less_sharp,
//        /laser_cloud_flat, /laser_cloud_less_flat) followed by int32 n + n x (float curvature, int32 label, int32 picked)
#include <chrono>
#include <cmath>

#include "ref_io.hpp"

int ref_node_main(int argc, char** argv);                                     // = main() of the reference node
extern float cloudCurvature[400000];                                          // reference src/scanRegistration.cpp:66-69
extern int cloudNeighborPicked[400000];
extern int cloudLabel[400000];

int main(int argc, char** argv) {
  ref_io::must(argc == 5, "usage: ref_scan_registration <scan_line> <minimum_range> <in.bin> <out.bin>");
  ref_shim::params()["scan_line"] = std::atof(argv[1]);
  ref_shim::params()["minimum_range"] = std::atof(argv[2]);
  FILE* fin = std::fopen(argv[3], "rb");
  FILE* fout = std::fopen(argv[4], "wb");
  ref_io::must(fin && fout, "cannot open files");
  const int n_frames = ref_io::read_i32(fin);
  int frame = 0;
  double handler_s = 0.0;                                                     // time inside the reference's callback (REF_TIMING=1 prints it)
  const char* topics[5] = {"/velodyne_cloud_2", "/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat"};
  ref_shim::ok_hook() = [&]() { return frame < n_frames; };
  ref_shim::spin_hook() = [&]() {                                             // ros::spin() -> one /velodyne_points message per turn
    const std::vector<float> v = ref_io::read_cloud(fin);
    bool dense = true;
    for (float x : v) dense = dense && std::isfinite(x);
    sensor_msgs::PointCloud2 msg = ref_io::make_msg(v, 0.1 * frame, dense);
    const auto t0 = std::chrono::steady_clock::now();
    ref_shim::deliver("/velodyne_points", msg);                              // laserCloudHandler runs inside: conversion, selection, the five toROSMsg + publish
    handler_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto& pub = ref_shim::published<sensor_msgs::PointCloud2>();
    for (const char* t : topics) {
      ref_io::must(pub[t].size() == static_cast<size_t>(frame) + 1, "the node did not publish one message per sweep");
      ref_io::write_cloud(fout, pub[t].back());
    }
    const int n = static_cast<int>(pub[topics[0]].back().width);
    ref_io::write_i32(fout, n);
    for (int i = 0; i < n; ++i) {
      const bool written = i >= 5 && i < n - 5;                               // the reference fills [5, cloudSize-5) only (:256-266)
      const float c = written ? cloudCurvature[i] : 0.f;
      const int32_t lab = written ? cloudLabel[i] : 0, pk = written ? cloudNeighborPicked[i] : 0;
      std::fwrite(&c, 4, 1, fout); ref_io::write_i32(fout, lab); ref_io::write_i32(fout, pk);
    }
    ++frame;
  };
  ref_node_main(argc, argv);
  std::fclose(fout);
  if (std::getenv("REF_TIMING")) std::fprintf(stderr, "REF_TIMING scan_registration frames %d seconds %.6f\n", frame, handler_s);
  return 0;
}
