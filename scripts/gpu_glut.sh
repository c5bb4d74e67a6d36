#!/bin/bash
# the LUT-in-global-memory fallback: parity tests + search time at 1M x 768 with the default pq_dim (384)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/iter
timeout 900 python -m pytest tests/test_ivf_pq_gpu.py -x -q > gpurun_out/iter/pq_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/iter/pq_tests.log
cat > /tmp/glut.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import ivf_pq, brute_force
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
n, dim, nq = 1_000_000, 768, 2000
x = bench.gen_rows(n, dim, 1234, dev, latent=24, n_modes=1); q = bench.gen_rows(nq, dim, 4321, dev, latent=24, n_modes=1)
bf = brute_force.build(x, resources=res); _, gt = brute_force.search(bf, q, 10, resources=res); res.sync(); gt = gt.cpu().numpy()
for pq_dim in (0, 192, 64):
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=1024, pq_dim=pq_dim, pq_bits=8, kmeans_trainset_fraction=0.2), x, resources=res); res.sync()
    sp = ivf_pq.SearchParams(n_probes=32, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f16"])
    for _ in range(2): d, i = ivf_pq.search(sp, idx, q, 10, resources=res)
    res.sync(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): d, i = ivf_pq.search(sp, idx, q, 10, resources=res)
    res.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("pq_dim %d (index %d): %.2f ms / %d queries, recall@10 %.3f" % (pq_dim, idx.pq_dim, dt * 1e3, nq, bench.recall_of(i.cpu().numpy(), gt)), flush=True)
    del idx
PY
timeout 600 python /tmp/glut.py 2>&1 | grep -v amdgpu.ids | tail -4
