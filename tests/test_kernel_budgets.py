"""Register / LDS / scratch budgets of the hot kernels, read from the code-object metadata hipcc emits for gfx950 (no GPU needed).

Several measured decisions of DESIGN.md §4a hinge on occupancy: a pick loop that needed 79 VGPRs lost against one with 59 although
it issued a third fewer instructions, `k_associate` lives off eight waves per SIMD (<= 64 VGPRs), `k_ring_features` off eight
workgroups per CU (<= 64 VGPRs AND <= 100 SGPRs, 20.2 KB of LDS each).  A source change that silently crosses one of these lines shows up
here, on the CPU box, instead of as an unexplained slowdown on the GPU."""
import os
import re
import shutil
import subprocess

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "a-loam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "--offload-device-only", "-S"]   # a-loam_amd/csrc/Makefile


def _kernels(tu, tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / (tu + ".s")
    r = subprocess.run([HIPCC, *FLAGS, "-o", str(out), os.path.join(CSRC, tu + ".hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    md = yaml.safe_load(re.search(r"\.amdgpu_metadata\n(.*?)\n\s*\.end_amdgpu_metadata", out.read_text(), re.S).group(1))
    names = [k[".name"] for k in md["amdhsa.kernels"]]
    dem = subprocess.run(["c++filt", *names], capture_output=True, text=True).stdout.splitlines() if shutil.which("c++filt") else names
    return {d.split("(")[0].replace("void ", "").replace("aloam::", ""): k for d, k in zip(dem, md["amdhsa.kernels"])}


@pytest.fixture(scope="module")
def registration(tmp_path_factory):
    return _kernels("registration_kernels", tmp_path_factory)


@pytest.fixture(scope="module")
def odometry(tmp_path_factory):
    return _kernels("odometry_kernels", tmp_path_factory)


@pytest.fixture(scope="module")
def mapping(tmp_path_factory):
    return _kernels("mapping_kernels", tmp_path_factory)


pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def occupancy_waves(k):
    """waves per SIMD the register file of a gfx9-family CU allows: 512 VGPRs per lane in steps of 8, and 800 SGPRs per SIMD (LLVM's
    getOccupancyWithNumSGPRs: <= 80 -> 10, <= 88 -> 9, <= 100 -> 8, else 7).  The scalar limit is the one nobody looks at."""
    v, s = k[".vgpr_count"], k[".sgpr_count"]
    by_v = min(8, 512 // (-(-max(v, 1) // 8) * 8))
    by_s = 10 if s <= 80 else 9 if s <= 88 else 8 if s <= 100 else 7
    return min(by_v, by_s)


def test_ring_features_keeps_eight_workgroups_per_cu(registration):
    k = registration["k_ring_features<2048>"]
    # eight workgroups of four waves share a CU's LDS (20.2 KB each) = eight waves per SIMD: <= 64 VECTOR registers and <= 100 SCALAR ones.
    # Until round 6 the kernel had 62 / 106 - seven waves, by the scalar file - and ran at 2.43 ms; amdgpu_waves_per_eu(8) on the kernel
    # gives 62 / 78 and 2.07 ms.  Pinned so that a regression of either file shows here.
    assert k[".vgpr_count"] <= 64 and k[".sgpr_count"] <= 100 and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, k
    assert occupancy_waves(k) == 8, k
    k = registration["k_ring_features<4096>"]                                  # the long-ring class: 38 KB of LDS, four workgroups per CU = four waves per SIMD; 86 measured
    assert k[".vgpr_count"] <= 88 and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, k
    assert occupancy_waves(k) >= 4, k
    for name in ("k_find_ends", "k_ring_starts", "k_dense_cloud"):
        k = registration[name]
        assert k[".vgpr_count"] <= 64 and k[".vgpr_spill_count"] == 0 and k[".sgpr_spill_count"] == 0, (name, k)
    # the one-pass front end holds four points per thread (coordinates, ring, rank, azimuth): 68 registers unforced; at eight waves per SIMD it
    # spills six of them and is still the faster build (1.13 against 1.15 ms at batch 1024)
    k = registration["k_front"]
    assert k[".vgpr_count"] <= 64 and k[".vgpr_spill_count"] <= 8 and k[".private_segment_fixed_size"] <= 40 and occupancy_waves(k) == 8, k


def test_association_waves_fit_eight_per_simd(odometry):
    """Round 4: two queries per wave (k_associate_pair).  The corner class fits eight waves per SIMD (<= 64 VGPRs), the planar class seven
    (<= 72: six kept rows of 32 candidates per half; forcing 64 spills 8 registers); 128-ring sensors keep eight rows (<= 80).  The nearly-sorted
    form (k_associate_nearly, round 6) and the flagged-sequence fallback only must not spill: they serve a handful of sequences per step."""
    limits = {"k_associate_pair<false, false>": 64, "k_associate_pair<true, false>": 72, "k_associate_pair<false, true>": 72, "k_associate_pair<true, true>": 80}
    for name, lim in limits.items():
        k = odometry[name]
        assert k[".vgpr_count"] <= lim and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, (name, k)
        assert k[".group_segment_fixed_size"] <= 1024, (name, k)               # mark slots + rank table: half a KiB per wave
    for name, k in odometry.items():
        if name.startswith("k_associate"):
            assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, (name, k)
    assert odometry["k_build_grids_fused"][".vgpr_count"] <= 64


def test_solvers_hold_their_state_in_registers(odometry, mapping):
    """The LM kernels run one wave per SIMD on purpose (the whole 6 x 6 system, the trust-region state and the accumulators of a
    pass in registers): up to 512 architected + accumulation registers, nothing in scratch."""
    for k in (odometry["k_solve<true>"], odometry["k_solve<false>"], mapping["k_map_solve"]):
        assert k[".vgpr_count"] <= 512 and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, k


def test_mapping_search_and_filters(mapping):
    for name in ("k_map_search<0>", "k_map_search<1>"):
        k = mapping[name]
        assert k[".vgpr_count"] <= 66 and k[".vgpr_spill_count"] == 0, (name, k)   # round 5: packed (distance, index) keys + positions: 63 / 65 registers (72 / 74 before)
    # the LDS voxel filter keeps its keys in registers; the 1024-thread instance is capped at 128 VGPRs by its workgroup size and is
    # allowed the handful of spilled registers it has today, not more
    big = mapping["k_vox_lds<1024, 24576, 65536>"]
    assert big[".vgpr_count"] <= 128 and big[".vgpr_spill_count"] <= 8 and big[".private_segment_fixed_size"] <= 64, big
    for name in ("k_vox_lds<256, 8192, 8192>", "k_vox_lds<64, 2048, 2048>"):
        assert mapping[name][".vgpr_spill_count"] == 0 and mapping[name][".private_segment_fixed_size"] == 0, (name, mapping[name])
