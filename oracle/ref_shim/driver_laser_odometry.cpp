// oracle/ref_shim/driver_laser_odometry.cpp — runs the reference's OWN odometry translation unit
// (/root/reference/src/laserOdometry.cpp + src/lidarFactor.hpp, compiled in place with -Dmain=ref_node_main against
// the stand-in headers under ref_shim/include) on feature clouds read from a file.  TEST INFRASTRUCTURE ONLY.
//   usage: ref_laser_odometry <in.bin> <out.bin>
//   in : int32 n_frames, then per frame five cloud records: sharp, less_sharp, flat, less_flat, full (/velodyne_cloud_2)
//   out: per frame 14 float64 (q_w_curr xyzw, t_w_curr, para_q xyzw, para_t), 2 int32 (corner_correspondence,
//        plane_correspondence of the last outer iteration), then the published /laser_cloud_corner_last and
//        /laser_cloud_surf_last records (mapping_skip_frame is set to 1 so that every frame is published), then the
//        correspondences of the frame's LAST ceres::Solve: int32 n_edge, n_plane, n_edge x 9 float64 (curr_point, last_point_a,
//        last_point_b of every LidarEdgeFactor the reference created, :365-381), n_plane x 12 float64 (curr_point, last_point_j, l, m
//        of every LidarPlaneFactor, :460-479) — read off the functors inside the residual blocks, the reference's source untouched
#include <chrono>

#include "ref_io.hpp"
#include "lidarFactor.hpp"                                                    // the reference's own header (struct members are public)

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
  std::vector<double> edges, planes;                                          // of the last Solve
  ceres::shim_solve_hook() = [&](const ceres::Problem& pb) {
    edges.clear(); planes.clear();
    auto put = [](std::vector<double>& v, const Eigen::Vector3d& p) { v.push_back(p.x()); v.push_back(p.y()); v.push_back(p.z()); };
    for (const auto& rb : pb.residuals_) {
      const std::type_info* ti = rb.cost->shim_functor_type();
      if (ti && *ti == typeid(LidarEdgeFactor)) {
        const LidarEdgeFactor* f = static_cast<const LidarEdgeFactor*>(rb.cost->shim_functor());
        put(edges, f->curr_point); put(edges, f->last_point_a); put(edges, f->last_point_b);
      } else if (ti && *ti == typeid(LidarPlaneFactor)) {
        const LidarPlaneFactor* f = static_cast<const LidarPlaneFactor*>(rb.cost->shim_functor());
        put(planes, f->curr_point); put(planes, f->last_point_j); put(planes, f->last_point_l); put(planes, f->last_point_m);
      }
    }
  };
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
      ref_io::write_i32(fout, static_cast<int>(edges.size() / 9)); ref_io::write_i32(fout, static_cast<int>(planes.size() / 12));
      if (!edges.empty()) ref_io::write_f64(fout, edges.data(), static_cast<int>(edges.size()));
      if (!planes.empty()) ref_io::write_f64(fout, planes.data(), static_cast<int>(planes.size()));
      edges.clear(); planes.clear();
      ++flushed;
    }
  };
  // REF_TIMING=1: time the node spends between the delivery of a frame's five messages and its next ros::ok(): the body of its main loop
  // (association, ceres::Solve, the publishers), without this driver's file reads and result write-out
  double loop_s = 0.0;
  bool running = false;
  std::chrono::steady_clock::time_point t_frame;
  ref_shim::ok_hook() = [&]() {
    if (running) { loop_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_frame).count(); running = false; }
    flush();
    return flushed < n_frames;
  };
  ref_shim::spin_hook() = [&]() {                                             // ros::spinOnce(): one synchronised frame per turn
    if (delivered >= n_frames) return;
    const double stamp = 0.1 * delivered;
    const char* topics[5] = {"/laser_cloud_sharp", "/laser_cloud_less_sharp", "/laser_cloud_flat", "/laser_cloud_less_flat", "/velodyne_cloud_2"};
    sensor_msgs::PointCloud2 msgs[5];
    for (int k = 0; k < 5; ++k) msgs[k] = ref_io::make_msg(ref_io::read_cloud(fin), stamp);
    t_frame = std::chrono::steady_clock::now();
    running = true;
    for (int k = 0; k < 5; ++k) ref_shim::deliver(topics[k], msgs[k]);
    ++delivered;
  };
  ref_node_main(argc, argv);
  flush();
  std::fclose(fout);
  if (std::getenv("REF_TIMING")) std::fprintf(stderr, "REF_TIMING laser_odometry frames %d seconds %.6f\n", delivered, loop_s);
  return 0;
}
