"""CPU: the C-ABI library loads and exports every symbol that include/cuvs/**.h declares; struct layouts that
callers mutate directly have the reference's sizes/offsets (c/include/cuvs/neighbors/*.h)."""
import ctypes as C
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cuvs_amd", "libcuvs_c.so")


def _declared_symbols():
    names = set()
    hdrs = glob.glob(os.path.join(ROOT, "include", "cuvs", "**", "*.h"), recursive=True)
    hdrs += glob.glob(os.path.join(ROOT, "include", "cuvs_amd", "*.h"))  # extensions (list-sharded multi-GPU)
    for h in hdrs:
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"CUVS_EXPORT\s+[\w\s\*]+?\b(cuvs\w+)\s*\(", text):
            names.add(m.group(1))
    return sorted(names)


def test_library_exists_and_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run `make` first"
    syms = _declared_symbols()
    assert len(syms) > 80
    lib = C.CDLL(LIB)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert "cuvsAmdShardAllGatherTopK" in syms and "cuvsAmdIvfPqSetListShard" in syms


def test_shard_header_is_valid_c99(tmp_path):
    src = tmp_path / "s.c"
    src.write_text("#include <cuvs_amd/shard.h>\nint main(void) { cuvsAmdShardComm_t c = 0; (void)c; return 0; }\n")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def test_error_text_convention_without_gpu():
    lib = C.CDLL(LIB)
    lib.cuvsGetLastErrorText.restype = C.c_char_p
    lib.cuvsSetLastErrorText(b"boom")
    assert lib.cuvsGetLastErrorText() == b"boom"
    lib.cuvsSetLastErrorText(b"")
    assert lib.cuvsGetLastErrorText() is None
    major, minor, patch = C.c_uint16(), C.c_uint16(), C.c_uint16()
    assert lib.cuvsVersionGet(C.byref(major), C.byref(minor), C.byref(patch)) == 1  # CUVS_SUCCESS
    assert (major.value, minor.value) == (26, 8)
    assert lib.cuvsVersionGet(None, None, None) == 0  # CUVS_ERROR, text set
    assert lib.cuvsGetLastErrorText() is not None


def test_param_struct_layouts_match_the_reference_headers(tmp_path):
    # tests/golden/abi_layout.txt was produced by compiling tests/golden/abi_probe.c against the REFERENCE's
    # c/include (tests/golden/gen_abi_layout.sh); the same probe against our include/ must print the same.
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "golden", "abi_probe.c"),
                           "-o", str(exe)])
    ours = subprocess.check_output([str(exe)]).decode()
    want = open(os.path.join(ROOT, "tests", "golden", "abi_layout.txt")).read()
    assert ours == want


def test_reference_c_sources_compile_and_link_against_our_headers(tmp_path):
    """The reference's pure-C API drivers and its include-guard test compile unchanged against include/ and link
    against libcuvs_c.so with no unresolved symbol (run on the GPU by tests/test_reference_c_drivers_gpu.py).
    Needs the reference tree, which only exists in the build container."""
    ref = "/root/reference/c/tests"
    if not os.path.isdir(ref):
        pytest.skip("no reference tree on this machine")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c11", "-fsyntax-only", "-I", inc, os.path.join(ref, "core", "headers.c")])
    so = tmp_path / "drivers.so"
    srcs = [os.path.join(ref, "neighbors", f) for f in ("run_brute_force_c.c", "run_ivf_flat_c.c", "run_ivf_pq_c.c", "run_mg_c.c")]
    srcs.append(os.path.join(ref, "distance", "run_pairwise_distance_c.c"))
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror=implicit-function-declaration", "-fPIC", "-shared", "-I", inc,
                           *srcs, "-L", os.path.join(ROOT, "cuvs_amd"), "-lcuvs_c", "-Wl,--no-undefined", "-o", str(so)])
    assert so.exists()
    # c/tests/core/c_api.c: a program; its one CUDA-runtime call (stream creation) is mapped by the recipe
    exe = tmp_path / "core_c_api"
    subprocess.check_call(["gcc", "-std=c11", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-include", "hip/hip_runtime_api.h", "-DcudaStreamCreate=hipStreamCreate",
                           "-I", inc, os.path.join(ref, "core", "c_api.c"), "-L", os.path.join(ROOT, "cuvs_amd"), "-lcuvs_c",
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", str(exe)])
    assert exe.exists()


def test_headers_are_valid_c99_and_cxx17(tmp_path):
    """The umbrella header (twice, as c/tests/core/headers.c does) and the generated version header compile as strict
    C99 and as C++17: what cgo/bindgen and C++ callers respectively need."""
    src = tmp_path / "h.c"
    src.write_text("#include <cuvs/core/all.h>\n#include <cuvs/core/all.h>\n#include <cuvs/version_config.h>\n"
                   "int main(void) { cuvsResources_t r = 0; (void)r; return CUVS_VERSION_MAJOR == 26 ? 0 : 1; }\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])


def test_cpp_surface_header_compiles(tmp_path):
    """include/cuvs_amd/neighbors.hpp (cuvs::neighbors::* over the C ABI) is valid C++17 without HIP or RAFT headers."""
    src = tmp_path / "n.cpp"
    src.write_text("#include <cuvs_amd/neighbors.hpp>\nint main() { cuvs::neighbors::ivf_pq::search_params p; return p.n_probes == 20 ? 0 : 1; }\n")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
