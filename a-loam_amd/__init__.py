"""a-loam_amd — MI355X-native A-LOAM hot path (scan registration + scan-to-scan odometry).

The directory name follows the project layout contract and is not a Python identifier; import it with
    aloam = importlib.import_module("a-loam_amd")
The product is the C-ABI shared library built from csrc/ (include/aloam_mi355x.h); this package only holds
the ctypes binding used by tests / bench (binding.py), the synthetic input generator (synthetic.py) and the
KITTI file readers (kitti_io.py).
"""
