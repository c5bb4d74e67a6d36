"""Kernel times of the IVF-PQ coarse search per coarse_search_dtype (run under rocprofv3 --kernel-trace --stats):
10k queries x n_lists centres, 128-d. Usage: python scripts/coarse_dtype_profile.py [n_lists]"""
import os
import sys

os.environ.setdefault("CUVS_AMD_DEBUG_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cuvs_amd.neighbors import ivf_pq

n_lists = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((n_lists * 32, 128)).astype(np.float32) * 0.3).cuda()
q = torch.from_numpy(rng.standard_normal((10000, 128)).astype(np.float32) * 0.3).cuda()
index = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, pq_dim=64, kmeans_n_iters=2, kmeans_trainset_fraction=0.25), x)
for name, dt in (("f32", np.float32), ("f16", np.float16), ("i8", np.int8)):
    sp = ivf_pq.SearchParams(n_probes=32, coarse_search_dtype=dt)
    for _ in range(3):
        ivf_pq.search(sp, index, q, 10)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        ivf_pq.search(sp, index, q, 10)
    t1.record(); torch.cuda.synchronize()
    print(f"coarse {name}: {t0.elapsed_time(t1) / 10:.3f} ms per search", flush=True)
