// a-loam_amd/csrc/odometry_kernels.hpp — host-callable launchers of the odometry kernels.
#pragma once
#include "aloam_device.hpp"

namespace aloam {
size_t build_grids_lds_bytes(int H, int R);
int prepare_build_grids(int H_surf);
void launch_build_grids(const OdomArgs& a, hipStream_t s);
void launch_transform_queries(const OdomArgs& a, hipStream_t s);
void launch_associate(const OdomArgs& a, bool plane, hipStream_t s);
void launch_solve(const OdomArgs& a, hipStream_t s);
void launch_advance(SeqMeta* meta, int B, hipStream_t s);
}  // namespace aloam
