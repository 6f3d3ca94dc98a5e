"""CPU: the C-ABI library loads, exports every symbol include/gloc3d.h declares, and refuses to
compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gloc3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gloc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(capi):
    L = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/gloc3d.h but not exported"
    assert sorted(capi.EXPORTED_SYMBOLS) == declared


def test_abi_version_and_defaults(capi):
    L = capi.lib()
    assert L.gloc_abi_version() == 6
    p = capi.default_reg_params()
    assert p.max_final_step == 0.0 and p.max_rmse == 0.0      # both plausibility checks off by default, as the reference (round 5)
    # constants mirrored from the reference: loop_detector.cpp:257, global_registration.cpp:242
    assert p.ransac_iters == 3000 and abs(p.inlier_thresh - 0.6) < 1e-7 and p.icp_iters == 30
    assert capi.reg_select_first_ok([0, 0, 1, 1]) == 2
    assert capi.reg_select_first_ok([0, 0]) == -1


def test_bench_passes_the_suggested_convergence_threshold():
    """bench.py opts into the convergence check explicitly (the library default is off): the value it passes is the one
    include/gloc3d.h names."""
    import os, re, sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    hdr = open(os.path.join(root, "include", "gloc3d.h")).read()
    m = re.search(r"#define\s+GLOC_REG_FINAL_STEP_SUGGESTED\s+([0-9.]+)f", hdr)
    assert m, "GLOC_REG_FINAL_STEP_SUGGESTED is declared"
    sys.path.insert(0, root)
    import bench
    assert abs(float(m.group(1)) - bench.MAX_FINAL_STEP) < 1e-9


def test_no_cpu_fallback_without_gpu(capi):
    if capi.lib().gloc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.GlocError) as e:
        capi.KnnIndex(512)
    assert e.value.code == 4  # GLOC_ERR_NODEVICE
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(capi.GlocError):
        capi.Registrar()
    with pytest.raises(capi.GlocError):
        capi.ScanStore()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gloc3d_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
                assert "oracle/" not in re.sub(r"oracle/(reg|knn|bev|ground|coarse)_oracle\.c", "", txt), f
