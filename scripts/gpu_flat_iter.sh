#!/bin/bash
# IVF-Flat iteration on the GPU: parity tests + int8 vs fp32 scan timing (1M x 128, n_lists 1024, n_probes 32, 10k queries)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/iter
timeout 900 python -m pytest tests/test_ivf_flat_gpu.py -x -q > gpurun_out/iter/flat_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/iter/flat_tests.log
cat > /tmp/flat_i8.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import cuvs_amd
from cuvs_amd.neighbors import ivf_flat
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
n, nq, dim = 2_000_000, 10000, 128
xf = torch.randn((n, dim), device=dev, generator=g) * 30
qf = torch.randn((nq, dim), device=dev, generator=g) * 30
for name, x, q in (("int8", xf.clamp(-127, 127).to(torch.int8), qf.clamp(-127, 127).to(torch.int8)), ("fp32", xf, qf)):
    idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=1024, kmeans_trainset_fraction=0.2), x, resources=res)
    sp = ivf_flat.SearchParams(n_probes=32)
    for _ in range(2): ivf_flat.search(sp, idx, q, 10, resources=res)
    res.sync(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ivf_flat.search(sp, idx, q, 10, resources=res)
    res.sync(); torch.cuda.synchronize()
    print(name, "ms/search %.3f" % ((time.perf_counter() - t) / 5 * 1e3))
    del idx
PY
timeout 600 python /tmp/flat_i8.py 2>&1 | grep -v amdgpu.ids | tail -4
