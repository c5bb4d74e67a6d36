#!/bin/bash
# duration of the threshold-append tile kernel alone (10k x 2M x 128), with parts of the main loop switched off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/tb.py <<'PY'
import sys, os, ctypes
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, cuvs_amd
from cuvs_amd import _lib
lib = _lib.lib() if callable(getattr(_lib, "lib", None)) else _lib.load()
res = cuvs_amd.common.Resources()
lib.cuvsAmdTileBench.argtypes = [ctypes.c_size_t, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
for spec in sys.argv[1:]:
    m, n, dim, dbg = (int(v) for v in spec.split(","))
    ms = ctypes.c_float(0)
    rc = lib.cuvsAmdTileBench(res.handle if hasattr(res, "handle") else res.get_c_obj(), m, n, dim, dbg, 3, ctypes.byref(ms))
    print(spec, "rc", rc, "ms %.3f" % ms.value, "TF %.1f" % (2.0 * m * n * dim / ms.value / 1e9))
PY
timeout 600 python /tmp/tb.py "$@" 2>&1 | grep -v amdgpu.ids | tail -12
