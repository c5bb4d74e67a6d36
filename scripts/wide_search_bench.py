"""IVF-PQ search at 768 dimensions on the wide matrix-core path (ivf_pq_wide.hip) against the LUT scan kernels: ms per batch, equality
of the results, the filter's counters. usage: python scripts/wide_search_bench.py rows n_lists pq_dim n_probes nq k [lut acc metric dim]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench, cuvs_amd
from cuvs_amd.neighbors import ivf_pq
from cuvs_amd._lib import lib

rows, n_lists, pq_dim, n_probes, nq, k = [int(v) for v in sys.argv[1:7]]
lut = {"f16": np.float16, "f32": np.float32, "fp8": np.uint8}[sys.argv[7] if len(sys.argv) > 7 else "f16"]
acc = {"f16": np.float16, "f32": np.float32}[sys.argv[8] if len(sys.argv) > 8 else "f32"]
metric = sys.argv[9] if len(sys.argv) > 9 else "sqeuclidean"
dim = int(sys.argv[10]) if len(sys.argv) > 10 else 768
dev = torch.device("cuda:0")
x = torch.empty((rows, dim), dtype=torch.float16, device=dev)
bench.gen_rows(rows, dim, 1234, dev, latent=32, n_modes=4096, out=x, spread=0.7)
q = torch.empty((nq, dim), dtype=torch.float16, device=dev)
bench.gen_rows(nq, dim, 4321, dev, latent=32, n_modes=4096, out=q, spread=0.7)
index = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, pq_dim=pq_dim, pq_bits=8, metric=metric, kmeans_n_iters=10, kmeans_trainset_fraction=min(1.0, 5e5 / rows)), x)
torch.cuda.synchronize()

def run(tag, **env):
    res = bench.comparator_handle(**env)
    sp = ivf_pq.SearchParams(n_probes=n_probes, lut_dtype=lut, internal_distance_dtype=acc, max_internal_batch_size=nq)
    d, i = ivf_pq.search(sp, index, q, k, resources=res); res.sync(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.time(); ivf_pq.search(sp, index, q, k, resources=res); res.sync(); torch.cuda.synchronize(); ts.append(time.time() - t0)
    st = (C.c_uint64 * 6)(); lib().cuvsAmdIvfPqLastFilterStats6(st)
    print(f"{tag}: {min(ts) * 1e3:.2f} ms per {nq} queries; counters {[int(v) for v in st]}", flush=True)
    return d, i

print(f"{rows} x {dim}, {n_lists} lists, pq_dim {pq_dim}, {n_probes} probes, k {k}, lut {sys.argv[7] if len(sys.argv) > 7 else 'f16'} acc {sys.argv[8] if len(sys.argv) > 8 else 'f32'}, {metric}")
b = run("LUT scan kernels", CUVS_AMD_PQ_WIDE=0)
run("wide path (counters on)", CUVS_AMD_SCAN_DEBUG=1024)
g = run("wide path")
print("   ids and distances equal:", bool((g[1] == b[1]).all() and (g[0] == b[0]).all()), flush=True)
