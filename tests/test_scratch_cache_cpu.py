"""CPU: the bookkeeping of the handle's scratch cache (cuvs_amd/csrc/scratch_cache.hpp - exact-size re-use, stream bypass,
size limit, allocation-failure retry, stream change) against a counting stand-in for the runtime's allocator."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scratch_cache_bookkeeping(tmp_path):
    exe = tmp_path / "scratch_cache_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-I", os.path.join(ROOT, "cuvs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "scratch_cache_test.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "scratch cache OK" in out.stdout
