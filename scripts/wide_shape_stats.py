"""The IVF-PQ search a CAGRA build runs (rows x 768 fp16, n_lists = sqrt(rows), pq_dim 64, k = 256, n_probes = n_lists / 50, fp16 LUT and
scores), one batch of 16384 of the dataset's own rows: ms per batch and the filter's counters, wide path vs LUT scan.
usage: python scripts/wide_shape_stats.py [rows]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench, cuvs_amd
from cuvs_amd.neighbors import ivf_pq
from cuvs_amd._lib import lib

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda:0")
x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
bench.gen_rows(rows, 768, 1234, dev, latent=24, n_modes=1, out=x, spread=0.35)
n_lists = int(rows ** 0.5)
t0 = time.time()
index = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, pq_dim=64, pq_bits=8, kmeans_n_iters=10, kmeans_trainset_fraction=min(1.0, max(0.02, 2e6 / rows))), x)
torch.cuda.synchronize()
print(f"build {time.time() - t0:.1f} s, n_lists {n_lists}", flush=True)
q = x[:16384].contiguous()
k = 256
n_probes = max(8, n_lists // 50)

def run(tag, acc=np.float16, **env):
    res = bench.comparator_handle(**env)  # (the switches are read when the handle is created)
    sp = ivf_pq.SearchParams(n_probes=n_probes, lut_dtype=np.float16, internal_distance_dtype=acc, max_internal_batch_size=16384)
    d, i = ivf_pq.search(sp, index, q, k, resources=res); res.sync(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.time(); ivf_pq.search(sp, index, q, k, resources=res); res.sync(); torch.cuda.synchronize(); ts.append(time.time() - t0)
    st = (C.c_uint64 * 6)(); lib().cuvsAmdIvfPqLastFilterStats6(st)
    print(f"{tag}: {min(ts) * 1e3:.1f} ms per batch; stats {[int(v) for v in st]}", flush=True)
    return d, i

base = run("LUT scan", CUVS_AMD_PQ_WIDE=0)
for h in (0, 2, 4, 8):
    got = run(f"wide heads={h} (0: rule) acc f16", CUVS_AMD_PQ_WIDE_HEADS=h, CUVS_AMD_SCAN_DEBUG=1024)
    print("   equal to the LUT scan:", bool((got[1] == base[1]).all() and (got[0] == base[0]).all()))
b32 = run("LUT scan acc f32", acc=np.float32, CUVS_AMD_PQ_WIDE=0)
for h in (0, 3, 8, 12):
    g32 = run(f"wide heads={h} (0: rule) acc f32", acc=np.float32, CUVS_AMD_PQ_WIDE_HEADS=h, CUVS_AMD_SCAN_DEBUG=1024)
    print("   equal:", bool((g32[1] == b32[1]).all() and (g32[0] == b32[0]).all()))
