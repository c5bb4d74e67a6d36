#!/bin/bash
# one rank's share of C5 (1B x 96 int8 over 8 GPUs): 125M x 96 int8, IVF-PQ n_lists 16384, pq_dim 64, 10k queries
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/c5.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import ivf_pq, refine
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
n, dim, nq = 125_000_000, 96, 10000
x = torch.empty((n, dim), dtype=torch.int8, device=dev)
step = 5_000_000
for r0 in range(0, n, step):
    f = bench.gen_rows(min(step, n - r0), dim, 1234 + r0, dev)
    x[r0:r0 + f.shape[0]] = (f * 40).clamp(-127, 127).to(torch.int8)
q = (bench.gen_rows(nq, dim, 1234, dev) * 40).clamp(-127, 127).to(torch.int8)  # same modes as the first chunk
torch.cuda.synchronize(); t0 = time.time()
idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=16384, pq_dim=64, pq_bits=8, kmeans_trainset_fraction=0.02), x, resources=res)
res.sync(); print("build %.1f s, %d rows" % (time.time() - t0, len(idx)), flush=True)
sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f16"])
for _ in range(2): d, i = ivf_pq.search(sp, idx, q, 20, resources=res)
res.sync(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): d, i = ivf_pq.search(sp, idx, q, 20, resources=res)
res.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
# exact top-10 of 200 queries by chunked matmul in fp32
qs = q[:200].float(); best_d = torch.full((200, 10), float("inf"), device=dev); best_i = torch.zeros((200, 10), dtype=torch.int64, device=dev)
for r0 in range(0, n, step):
    xc = x[r0:r0 + step].float()
    dd = (xc * xc).sum(1)[None, :] - 2.0 * (qs @ xc.T)
    v, ii = torch.topk(dd, 10, dim=1, largest=False)
    cat_d = torch.cat([best_d, v], 1); cat_i = torch.cat([best_i, ii + r0], 1)
    o = torch.argsort(cat_d, dim=1)[:, :10]
    best_d = torch.gather(cat_d, 1, o); best_i = torch.gather(cat_i, 1, o)
_, ri = refine(x, q[:200], i[:200], k=10, metric="sqeuclidean", resources=res); res.sync()
print("search %.2f ms / %d queries = %.0f q/s; recall@10 (k=20 + refine) %.4f" % (dt * 1e3, nq, nq / dt, bench.recall_of(ri.cpu().numpy(), best_i.cpu().numpy())), flush=True)
PY
timeout 900 python /tmp/c5.py 2>&1 | grep -v amdgpu.ids | tail -4
