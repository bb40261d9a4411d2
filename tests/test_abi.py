"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/aloam_mi355x.h
declares, validates its configuration like the reference nodes do, and refuses to run without a HIP device
(no CPU fallback).  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import pytest


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol(binding):
    L = binding.lib()
    declared = binding.declared_symbols()
    assert len(declared) >= 24
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", binding.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (aloam_[a-z_0-9]+)", out))
    assert set(declared) <= exported
    # the boundary is plain C: nothing but aloam_* is part of the public surface, and the header compiles as C
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", binding.HEADER_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_product_library_does_not_link_the_oracle(binding):
    out = subprocess.run(["ldd", binding.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out
    syms = subprocess.run(["nm", "-D", binding.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in syms
    src = os.path.join(os.path.dirname(binding.LIB_PATH), "..", "csrc")
    for f in os.listdir(src):
        txt = open(os.path.join(src, f)).read()
        assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


def test_default_config_matches_hdl64_launch_file(binding):
    cfg = binding.AloamConfig()
    binding.lib().aloam_default_config(C.byref(cfg))
    assert cfg.n_scans == 64 and abs(cfg.min_range - 5.0) < 1e-7          # launch/aloam_velodyne_HDL_64.launch:3-4
    assert cfg.lm_max_iterations == 4 and cfg.outer_iterations == 2        # laserOdometry.cpp:278,496
    assert cfg.batch == 1 and cfg.max_points <= 400000                     # scanRegistration.cpp:66-69


def test_unsupported_scan_line_is_rejected_like_the_reference(binding):
    """scanRegistration.cpp:472-476: only 16 / 32 / 64 scan lines."""
    with pytest.raises(binding.AloamError) as e:
        binding.Aloam(n_scans=48)
    assert e.value.code == binding.E_SCAN_LINES
    with pytest.raises(binding.AloamError) as e:
        binding.Aloam(n_scans=64, batch=0)
    assert e.value.code == binding.E_ARG


@pytest.mark.skipif(_has_gpu(), reason="this box has a GPU")
def test_no_cpu_fallback_without_a_device(binding):
    with pytest.raises(binding.AloamError) as e:
        binding.Aloam(n_scans=16, min_range=0.3, max_points=30000)
    assert e.value.code == binding.E_HIP


def test_ctypes_mirrors_match_the_header_layout(binding, tmp_path):
    """The structures of a-loam_amd/binding.py against what a C compiler makes of include/aloam_mi355x.h: same size, same
    field names in the same order at the same offsets (catches a field added on one side only), distortion defaulting to 0."""
    fields = {"aloam_config": [n for n, _ in binding.AloamConfig._fields_], "aloam_odom_stats": [n for n, _ in binding.AloamOdomStats._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{binding.HEADER_PATH}"', "int main(void) {"]
    for st, names in fields.items():
        src.append(f'  printf("{st} %zu", sizeof({st}));')
        for n in names:
            src.append(f'  printf(" {n}:%zu", offsetof({st}, {n}));')
        src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for line, (st, cls) in zip(out, (("aloam_config", binding.AloamConfig), ("aloam_odom_stats", binding.AloamOdomStats))):
        parts = line.split()
        assert parts[0] == st and int(parts[1]) == C.sizeof(cls), line
        for tok, (name, _) in zip(parts[2:], cls._fields_):
            n, off = tok.split(":")
            assert n == name and int(off) == getattr(cls, name).offset, (st, tok)
    hdr = re.sub(r"/\*.*?\*/", "", open(binding.HEADER_PATH).read(), flags=re.S)
    body = re.search(r"typedef struct aloam_config \{(.*?)\} aloam_config;", hdr, flags=re.S).group(1)
    declared = re.findall(r"\b(\w+)\s*;", body)
    assert declared == fields["aloam_config"], (declared, fields["aloam_config"])
    cfg = binding.AloamConfig()
    binding.lib().aloam_default_config(C.byref(cfg))
    assert cfg.distortion == 0                                             # #define DISTORTION 0 (laserOdometry.cpp:59)
