"""Diagnose large-scale mismatch: compare IVF-PQ ids, our brute-force GT and a torch reference for a few queries."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import cuvs_amd
from cuvs_amd.neighbors import brute_force, ivf_pq

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
n_lists = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda", 0)
res = cuvs_amd.common.Resources()
data = bench.gen_rows(rows, 128, 1234, dev)
q = bench.gen_rows(64, 128, 4321, dev)
# torch reference top-10 for the queries
best_d = torch.full((64, 10), float("inf"), device=dev)
best_i = torch.full((64, 10), -1, dtype=torch.int64, device=dev)
qn = (q * q).sum(1, keepdim=True)
for r0 in range(0, rows, 1 << 22):
    x = data[r0:r0 + (1 << 22)]
    d = qn + (x * x).sum(1)[None, :] - 2.0 * (q @ x.T)
    dd, ii = torch.topk(d, 10, dim=1, largest=False)
    cat_d = torch.cat([best_d, dd], 1); cat_i = torch.cat([best_i, ii + r0], 1)
    o = torch.argsort(cat_d, dim=1)[:, :10]
    best_d = torch.gather(cat_d, 1, o); best_i = torch.gather(cat_i, 1, o)
bf = brute_force.build(data, resources=res)
gd, gi = brute_force.search(bf, q, 10, resources=res); res.sync()
print("torch ids[0]", best_i[0].tolist()); print("bf    ids[0]", gi[0].tolist())
print("bf vs torch recall", np.mean([len(np.intersect1d(a, b)) for a, b in zip(gi.cpu().numpy(), best_i.cpu().numpy())]) / 10)
ip = ivf_pq.IndexParams(n_lists=n_lists, pq_dim=64, kmeans_trainset_fraction=0.02)
t0 = time.time(); index = ivf_pq.build(ip, data, resources=res); res.sync(); print("build s", time.time() - t0)
sizes = index.list_sizes.cpu().numpy().astype(np.int64)
print("list sizes: sum", sizes.sum(), "min", sizes.min(), "max", sizes.max(), "len", len(index))
sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=np.float16, internal_distance_dtype=np.float16, max_internal_batch_size=10000)
d, i = ivf_pq.search(sp, index, q, 10, resources=res); res.sync()
print("ivf   ids[0]", i[0].tolist()); print("ivf d[0]", d[0].tolist()); print("true d[0]", best_d[0].tolist())
print("ivf vs torch recall", np.mean([len(np.intersect1d(a, b)) for a, b in zip(i.cpu().numpy(), best_i.cpu().numpy())]) / 10)
ids0 = index.list_indices(0).cpu().numpy(); print("list0 ids head", ids0[:8], "n", len(ids0))
sp32 = ivf_pq.SearchParams(n_probes=128, max_internal_batch_size=10000)
d, i = ivf_pq.search(sp32, index, q, 10, resources=res); res.sync()
print("ivf32 vs torch recall", np.mean([len(np.intersect1d(a, b)) for a, b in zip(i.cpu().numpy(), best_i.cpu().numpy())]) / 10)
