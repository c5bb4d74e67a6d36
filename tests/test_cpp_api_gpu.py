"""GPU: the header-only C++ surface (include/cuvs_amd/neighbors.hpp: cuvs::neighbors::{brute_force, ivf_flat, ivf_pq,
cagra}::build / search over the C ABI) - compiled with g++ against libcuvs_c.so and run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_surface_builds_and_searches(tmp_path):
    exe = tmp_path / "cpp_api_demo"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "cpp_api_demo.cpp"), "-L",
                           os.path.join(ROOT, "cuvs_amd"), "-lcuvs_c", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "cuvs_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp api OK" in out.stdout
