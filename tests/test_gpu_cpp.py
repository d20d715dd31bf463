"""The C++ mirror of the reference API (include/phnsw.hpp): tests/cpp/test_hnsw.cpp restates the
reference's own unit tests in C++; it is compiled with g++ against libphnsw.so and run here."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "parallel_hnsw_amd")


def _compile(tmp_path):
    exe = str(tmp_path / "test_hnsw")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_hnsw.cpp"), "-o", exe, "-L", LIBDIR, "-lphnsw",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_header_compiles_and_links(tmp_path):
    _compile(tmp_path)


@pytest.mark.gpu
def test_cpp_reference_tests(tmp_path):
    exe = _compile(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout
