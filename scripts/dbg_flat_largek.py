import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from cuvs_amd.neighbors import ivf_flat
rng = np.random.default_rng(5)
x = (rng.random((6000, 24), dtype=np.float32) * 1.9 + 0.1); q = (rng.random((40, 24), dtype=np.float32) * 1.9 + 0.1)
for metric in ("inner_product",):
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=16, metric=metric, kmeans_n_iters=10), torch.from_numpy(x).cuda())
    ex = ivf_flat.export_for_oracle(index, np.float32)
    for k, npb in ((512, 6), (256, 6), (200, 6)):
        gd, gi = ivf_flat.search(ivf_flat.SearchParams(n_probes=npb), index, torch.from_numpy(q).cuda(), k)
        torch.cuda.synchronize(); gd, gi = gd.cpu().numpy(), gi.cpu().numpy()
        od, oi = oracle.ivf_flat_search(ex, q, k, npb, metric=metric)
        bad = np.argwhere((gi != oi) | (gd != od))
        print(metric, k, npb, "mismatches", len(bad), bad[:5].tolist())
        for a, b in bad[:3]:
            print("  gpu", gi[a, max(0,b-1):b+2], gd[a, max(0,b-1):b+2], "oracle", oi[a, max(0,b-1):b+2], od[a, max(0,b-1):b+2])
