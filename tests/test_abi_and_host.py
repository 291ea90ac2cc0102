"""CPU-side checks (no GPU): the C-ABI library builds, loads and exports every symbol that include/swz_gpu.h
declares; it refuses to work without a device instead of falling back to the CPU; host-side bookkeeping
(node lists) that needs no device behaves; the product never reaches into oracle/."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "schwarzwald_amd", "csrc"), "-j", "4", "-s"], check=True)
    import schwarzwald_amd as swz
    return swz.load_library()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "swz_gpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(swz_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), "libswz_gpu.so does not export %s" % name
    import schwarzwald_amd.api as api
    assert lib.swz_abi_version() == 3 == api.ABI_VERSION


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import schwarzwald_amd as swz
    with pytest.raises(swz.SwzError) as e:
        swz.Context(0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_use_the_oracle():
    pkg = os.path.join(ROOT, "schwarzwald_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "jitter_tables.inc", os.path.join(dirpath, f)


def test_jitter_tables_are_identical_in_product_and_checker():
    a = open(os.path.join(ROOT, "schwarzwald_amd", "csrc", "jitter_tables.inc")).read()
    b = open(os.path.join(ROOT, "oracle", "jitter_tables.inc")).read()
    assert a == b
    nums = [int(x) for x in re.findall(r"\b\d+\b", a.split("SWZ_JITTER_TABLE(64)")[1])]
    assert len(nums) == 16 * 64
    for r in range(16):
        assert sorted(nums[r * 64:(r + 1) * 64]) == list(range(1, 65))


def test_spacing_from_diagonal_matches_reference_value():
    import schwarzwald_amd as swz
    s = swz.spacing_from_diagonal([0, 0, 0], [1, 1, 1], 250)
    assert np.float32(s).view(np.uint32) == 0x3BE305FB
