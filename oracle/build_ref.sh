#!/bin/bash
# Compiles the REFERENCE's own C-API test drivers (pure C, they only call cuvs*() entry points) from the sources
# where they lie under /root/reference against THIS repo's headers and library. Output: oracle/_ref/libref_c_drivers.so
# (git-ignored, travels to the GPU box). tests/test_reference_c_drivers_gpu.py runs them: the code the reference
# uses to test its C ABI then drives cuvs_amd/libcuvs_c.so unchanged - source-level proof of the drop-in boundary.
# These files hold no search algorithm (the reference's hot path is CUDA + un-vendored RAFT and cannot be built
# here); nothing is copied into the repository.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/.." && pwd)
REF=/root/reference/c/tests/neighbors
[ -d "$REF" ] || { echo "no reference tree: skipping"; exit 0; }
mkdir -p "$HERE/_ref"
# the reference's include-guard smoke test (c/tests/core/headers.c includes <cuvs/core/all.h> twice): compile only
gcc -std=c11 -fsyntax-only -I"$ROOT/include" /root/reference/c/tests/core/headers.c
gcc -O1 -fPIC -shared -std=c11 -I"$ROOT/include" \
    "$REF/run_brute_force_c.c" "$REF/run_ivf_flat_c.c" "$REF/run_ivf_pq_c.c" "$REF/run_mg_c.c" \
    /root/reference/c/tests/distance/run_pairwise_distance_c.c \
    -L"$ROOT/cuvs_amd" -lcuvs_c -Wl,-rpath,'$ORIGIN/../../cuvs_amd' \
    -o "$HERE/_ref/libref_c_drivers.so"
echo "built $HERE/_ref/libref_c_drivers.so"
# c/tests/core/c_api.c is a program of its own (resources, stream, RMM alloc/free/pool, pinned host memory, version).
# It creates its stream with the CUDA runtime; the recipe maps that one call to the HIP runtime, the way a maintainer
# retargeting the test would (nothing in the library or its headers aliases runtime functions).
gcc -O1 -std=c11 -Werror=implicit-function-declaration -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    -include hip/hip_runtime_api.h -DcudaStreamCreate=hipStreamCreate -I"$ROOT/include" \
    /root/reference/c/tests/core/c_api.c -L"$ROOT/cuvs_amd" -lcuvs_c -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,'$ORIGIN/../../cuvs_amd' -Wl,-rpath,/opt/rocm/lib -o "$HERE/_ref/ref_core_c_api"
echo "built $HERE/_ref/ref_core_c_api"
# The reference's C example programs (examples/c/src) call six CUDA runtime names directly (cudaMemcpy and its kinds,
# cudaError_t, cudaSuccess, cudaGetErrorString). A recipe-local <cuda_runtime.h> maps exactly those to the HIP runtime -
# the retargeting a maintainer of the examples would do; nothing of it ships in include/ or in the library.
# Best effort (they are demos, no test depends on them): oracle/_ref/examples/<name>, run them on a GPU box by hand.
STUB=$(mktemp -d)
cat > "$STUB/cuda_runtime.h" <<'EOS'
#pragma once
#include <hip/hip_runtime_api.h>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDefault hipMemcpyDefault
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
EOS
mkdir -p "$HERE/_ref/examples"
for ex in L2_c_example bruteforce_c_example ivf_flat_c_example ivf_pq_c_example cagra_c_example; do
  gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -I"$STUB" -include cuda_runtime.h -I/opt/rocm/include -I"$ROOT/include" \
      -I/root/reference/examples/c/src "/root/reference/examples/c/src/$ex.c" -L"$ROOT/cuvs_amd" -lcuvs_c \
      -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../../../cuvs_amd' -Wl,-rpath,/opt/rocm/lib \
      -o "$HERE/_ref/examples/$ex" 2>"$STUB/$ex.log" && echo "built examples/$ex" || { echo "examples/$ex not built:"; head -5 "$STUB/$ex.log"; }
done
rm -rf "$STUB"

