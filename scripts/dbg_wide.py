import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
os.environ["CUVS_AMD_DEBUG_SWITCHES"] = "1"
import test_ivf_pq_wide_gpu as T
x, q = T._mixture(60_000, 768, 400, seed=9)
index = T._pq_build(x, n_lists=32, pq_dim=64, pq_bits=8, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
kw = dict(n_probes=12, lut_dtype=np.float16, internal_distance_dtype=np.float32)
os.environ["CUVS_AMD_PQ_WIDE"] = "0"
sd, si = T._pq_search(index, q, 20, **kw)
del os.environ["CUVS_AMD_PQ_WIDE"]
for cap in ("2000", "20000", "200000"):
    os.environ["CUVS_AMD_PQ3_SURV_CAP"] = cap
    os.environ["CUVS_AMD_SCAN_DEBUG"] = "1024"
    hd, hi = T._pq_search(index, q, 20, **kw)
    st = T._filter_stats()
    bad = np.where((hi != si).any(axis=1))[0]
    print(cap, st, "bad queries", len(bad), bad[:10])
    if len(bad):
        b = bad[0]
        print(hi[b], si[b]); print(hd[b], sd[b])
