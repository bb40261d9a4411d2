"""a-loam_amd — what the MI355X A-LOAM hot path needs, nothing else.

csrc/       HIP kernels (gfx950) + the C-ABI host side -> lib/libaloam_mi355x.so (include/aloam_mi355x.h)
host/       the three ROS node shims that keep the aloam_velodyne topic surface and call the C ABI
binding.py  ctypes stub of the C ABI (tests, bench.py); no second implementation
synthetic.py  seeded synthetic sweeps (regular and KITTI-shaped rough ones); shard.py  sequence -> rank assignment
KITTI-layout input and ground-truth evaluation live in tools/run_kitti.py.
"""
