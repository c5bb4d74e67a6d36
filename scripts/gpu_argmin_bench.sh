#!/bin/bash
# fused argmin (k-means E-step / list assignment): 2M x 16384 x 128
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/am.py <<'PY'
import sys, os, ctypes, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import cuvs_amd
from cuvs_amd import _lib
lib = _lib.lib(); res = cuvs_amd.common.Resources()
m, n, d = 2_000_000, 16384, 128
q = torch.randn((m, d), device="cuda"); c = torch.randn((n, d), device="cuda"); lab = torch.empty((m,), dtype=torch.int32, device="cuda")
lib.cuvsAmdFusedArgmin.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
def run(): lib.cuvsAmdFusedArgmin(res.get_c_obj(), q.data_ptr(), m, c.data_ptr(), n, d, lab.data_ptr())
run(); res.sync(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): run()
res.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("argmin ms %.2f TF %.1f checksum %d" % (dt * 1e3, 2.0 * m * n * d / dt / 1e12, int(lab.long().sum())))
PY
timeout 300 python /tmp/am.py 2>&1 | grep -v amdgpu.ids | tail -1
