#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...> — an A/B build of the library next to the product one
# (a-loam_amd/lib/variants/lib<name>.so, git-ignored, travels to the GPU box); select it with ALOAM_MI355X_LIB=<path>.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../a-loam_amd/csrc"
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -shared -o ../lib/variants/lib$NAME.so registration_kernels.hip odometry_kernels.hip mapping_kernels.hip reference_order_kernels.hip aloam_capi.hip
echo a-loam_amd/lib/variants/lib$NAME.so
