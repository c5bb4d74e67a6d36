"""CPU: the container of a multi-GPU index file (snmg.cuh:735-757: dtype prefix, mode, number of ranks, then the index
streams back to back). The per-index writers/readers stay filename based; mg.hip opens a thread-local window
(npy_io.hpp: append / read_offset / end_offset). This drives exactly that mechanism with stand-in index streams."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROGRAM = r"""
#include "npy_io.hpp"
#include <cstdio>
using namespace cuvs_amd;
static void write_one(const char* f, int tag, int n)
{
  npy_writer w(f);
  w.scalar<int32_t>(tag);
  std::vector<float> v(n, (float)tag);
  w.host_array<float>(v.data(), {n});
  w.close();
}
static int read_one(const char* f, int n)
{
  npy_reader r(f);
  int tag = r.scalar<int32_t>();
  auto v  = r.host_array<float>(n);
  for (float x : v) if (x != (float)tag) return -1;
  return tag;
}
int main(int argc, char** argv)
{
  const char* f = argv[1];
  { npy_writer w(f); char p[4]; elem_prefix(elem_t::f16, p); w.raw(p, 4); w.scalar<int32_t>(1); w.scalar<int32_t>(3); w.close(); }
  for (int r = 0; r < 3; ++r) { g_npy_io = npy_io_window{}; g_npy_io.append = true; write_one(f, 10 + r, 5 + r); g_npy_io = npy_io_window{}; }
  long off; int mode, ranks; elem_t et;
  { npy_reader r(f); char p[4]; r.raw(p, 4); if (!parse_elem_prefix(p, &et)) return 2; mode = r.scalar<int32_t>(); ranks = r.scalar<int32_t>(); off = r.tell(); }
  if (et != elem_t::f16 || mode != 1 || ranks != 3) return 3;
  for (int r = 0; r < ranks; ++r) {
    g_npy_io = npy_io_window{}; g_npy_io.read_offset = off;
    int tag = read_one(f, 5 + r);
    off = g_npy_io.end_offset; g_npy_io = npy_io_window{};
    if (tag != 10 + r) return 4;
  }
  FILE* fp = fopen(f, "rb"); fseek(fp, 0, SEEK_END); long sz = ftell(fp); fclose(fp);
  if (sz != off) return 5;              // the last stream ends exactly at the end of the file
  write_one(f, 99, 2);                  // outside a window a writer truncates and a reader starts at 0 again
  if (read_one(f, 2) != 99) return 6;
  printf("ok %ld\n", sz);
  return 0;
}
"""


def test_index_streams_append_and_read_back_at_their_offsets(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    src, exe = tmp_path / "w.cpp", tmp_path / "w"
    src.write_text(_PROGRAM)
    subprocess.check_call(["hipcc", "-std=c++17", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cuvs_amd", "csrc"), str(src),
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", str(exe)])
    out = subprocess.run([str(exe), str(tmp_path / "mg.bin")], capture_output=True)
    assert out.returncode == 0, (out.returncode, out.stderr.decode()[-300:])
    assert out.stdout.decode().startswith("ok ")
