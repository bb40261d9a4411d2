// a-loam_amd/csrc/registration_kernels.hpp — host-callable launchers of the scan-registration kernels.
#pragma once
#include "aloam_device.hpp"

namespace aloam {
size_t ring_features_lds_bytes(int npad);
void launch_find_ends(const RegArgs& a, const int* d_nin, hipStream_t s);
void launch_front(const RegArgs& a, hipStream_t s);
void launch_ring_starts(const RegArgs& a, hipStream_t s);
void launch_dense_cloud(const RegArgs& a, hipStream_t s);
void launch_ring_features(const RegArgs& a, int npad, float leaf, hipStream_t s);
void launch_less_flat_reference_order(const RegArgs& a, int npad, float leaf, hipStream_t s);   // reference_order_kernels.hip
}  // namespace aloam
