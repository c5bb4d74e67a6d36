"""IVF-Flat at 768 dimensions (1M x 768 fp16 rows, 1024 lists, 32 probes, 10 k queries, k = 10): ms per search, recall, against the scan
kernel alone (CUVS_AMD_FLAT_SCAN3=0). usage: python scripts/flat768_bench.py [dim] [dtype f16|f32] [metric]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench, cuvs_amd
from cuvs_amd.neighbors import ivf_flat

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dt = torch.float16 if (len(sys.argv) < 3 or sys.argv[2] == "f16") else torch.float32
metric = sys.argv[3] if len(sys.argv) > 3 else "sqeuclidean"
rows, nq, k = 1_000_000, 10000, 10
dev = torch.device("cuda:0")
x = torch.empty((rows, dim), dtype=dt, device=dev)
bench.gen_rows(rows, dim, 1234, dev, latent=32, n_modes=4096, out=x, spread=0.7)
q = torch.empty((nq, dim), dtype=dt, device=dev)
bench.gen_rows(nq, dim, 4321, dev, latent=32, n_modes=4096, out=q, spread=0.7)
idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=1024, metric=metric, kmeans_trainset_fraction=0.5), x)
sp = ivf_flat.SearchParams(n_probes=32)

def run(tag, **env):
    res = bench.comparator_handle(**env)
    d, i = ivf_flat.search(sp, idx, q, k, resources=res); res.sync(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.time(); ivf_flat.search(sp, idx, q, k, resources=res); res.sync(); torch.cuda.synchronize(); ts.append(time.time() - t0)
    print(f"{tag}: {min(ts) * 1e3:.2f} ms per {nq} queries", flush=True)
    return d, i

print(f"IVF-Flat {rows} x {dim} {dt}, 1024 lists, 32 probes, k {k}, {metric}")
b = run("scan kernel alone", CUVS_AMD_FLAT_SCAN3=0)
g = run("default path")
print("   ids and distances equal:", bool((g[1] == b[1]).all() and (g[0] == b[0]).all()))
gt = bench.exact_topk_fp64(x, q[:500], k).cpu().numpy()
print("   recall@10:", bench.recall_of(g[1][:500].cpu().numpy(), gt))
