#!/bin/bash
# graph of a CAGRA build with two builds of the library: identical? + build time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/pr.py <<'PY'
import sys, os, time, hashlib, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import cagra
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
for n, dim, K, deg in ((300_000, 64, 128, 64), (150_000, 32, 48, 24)):
    x = bench.gen_rows(n, dim, 7, dev, latent=16, n_modes=1)
    t0 = time.time()
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=K, graph_degree=deg), x, resources=res); res.sync()
    g = idx.graph.cpu().numpy()
    print(n, dim, K, deg, "build %.2fs" % (time.time() - t0), hashlib.sha1(g.tobytes()).hexdigest()[:16], flush=True)
PY
timeout 600 python /tmp/pr.py 2>&1 | grep -v amdgpu.ids | tail -2
