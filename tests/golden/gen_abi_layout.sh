#!/bin/bash
# Generates tests/golden/abi_layout.txt from the REFERENCE's C headers (run in the build container, where
# /root/reference exists). cuda_runtime.h is stubbed the way the reference's own Rust binding stubs it
# (rust/cuvs-sys/bindgen-stubs/cuda_runtime.h); dlpack.h is this repo's copy of the public spec.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
STUB=$(mktemp -d)
cat > $STUB/cuda_runtime.h <<'EOS'
typedef struct CUstream_st* cudaStream_t;
typedef enum { CUDA_R_16F = 2, CUDA_R_32F = 0, CUDA_R_8I = 3, CUDA_R_8U = 8 } cudaDataType_t;
EOS
mkdir -p $STUB/dlpack && cp $ROOT/include/dlpack/dlpack.h $STUB/dlpack/
gcc -I$STUB -I/root/reference/c/include $HERE/abi_probe.c -o $STUB/probe_ref
$STUB/probe_ref > $HERE/abi_layout.txt
echo "wrote $HERE/abi_layout.txt"
