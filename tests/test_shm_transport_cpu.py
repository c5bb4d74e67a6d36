"""CPU: the host-staged transport of the shard communicator (cuvs_amd/csrc/shm_transport.hpp) between three forked
PROCESSES - all-gathers across two growths of the mapped file, min all-reduces, and the two failure modes that must raise
instead of hang (ranks in different collectives; a rank that never arrives) - under ASan / UBSan."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shm_transport_between_processes(tmp_path):
    exe = tmp_path / "shm_transport_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-I", os.path.join(ROOT, "cuvs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "shm_transport_test.cpp"), "-o", str(exe), "-lpthread"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shm transport OK" in out.stdout
