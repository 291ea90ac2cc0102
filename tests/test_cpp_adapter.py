"""Builds and runs the C++ host adapter test (tests/cpp/test_adapter.cpp): the reference-shaped C++ interface
over the C ABI, checked with the invariants of the reference's own (disabled) tiler tests and against the
oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmpdir):
    exe = os.path.join(tmpdir, "test_adapter")
    lib_dir = os.path.join(ROOT, "schwarzwald_amd", "lib")
    orc_dir = os.path.join(ROOT, "oracle")
    subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp"), "-o", exe,
                    "-L" + lib_dir, "-lswz_gpu", "-L" + orc_dir, "-loracle",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + orc_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_adapter_compiles_against_the_abi(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "schwarzwald_amd", "csrc"), "-j", "4", "-s"], check=True)
    assert os.path.exists(_build(str(tmp_path)))


@pytest.mark.gpu
def test_adapter_tiles_like_the_oracle(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok: ") == 8
