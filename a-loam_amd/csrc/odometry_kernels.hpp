// a-loam_amd/csrc/odometry_kernels.hpp — host-callable launchers of the odometry kernels.
#pragma once
#include "aloam_device.hpp"

namespace aloam {
void launch_nn_search(const OdomArgs& a, int which, int max_queries, int max_targets, hipStream_t s);
void launch_walk_corner(const OdomArgs& a, int max_queries, hipStream_t s);
void launch_walk_plane(const OdomArgs& a, int max_queries, hipStream_t s);
void launch_solve(const OdomArgs& a, hipStream_t s);
void launch_advance(SeqMeta* meta, int B, hipStream_t s);
}  // namespace aloam
