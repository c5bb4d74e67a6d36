import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from cuvs_amd.neighbors import ivf_pq
rng = np.random.default_rng(31)
n, d, nq = 30000, 128, 400
x = (rng.random((n, d), dtype=np.float32) * 1.9 + 0.1); q = (rng.random((nq, d), dtype=np.float32) * 1.9 + 0.1)
index = ivf_pq.build(ivf_pq.IndexParams(n_lists=64, pq_dim=64, pq_bits=8, kmeans_n_iters=10), torch.from_numpy(x).cuda())
ex = ivf_pq.export_for_oracle(index)
L = {"f32": np.float32, "f16": np.float16, "fp8": np.uint8}
for lut, acc in (("f16", "f32"), ("f16", "f16"), ("fp8", "f32"), ("fp8", "f16")):
    od, oi = oracle.ivf_pq_search(ex, q, 20, 16, lut=lut, acc=acc)
    for env in ({}, {"CUVS_AMD_PQ_SCAN2": "0"}, {"CUVS_AMD_PQ_HEAD_PROBES": "0"}, {"CUVS_AMD_PQ_HEAD_PROBES": "0", "CUVS_AMD_SCAN_DEBUG": "8"}):
        for k_ in ("CUVS_AMD_PQ_SCAN2", "CUVS_AMD_PQ_HEAD_PROBES", "CUVS_AMD_SCAN_DEBUG"):
            os.environ.pop(k_, None)
        os.environ.update(env)
        gd, gi = ivf_pq.search(ivf_pq.SearchParams(n_probes=16, lut_dtype=L[lut], internal_distance_dtype=L[acc]), index, torch.from_numpy(q).cuda(), 20)
        torch.cuda.synchronize()
        gd, gi = gd.cpu().numpy(), gi.cpu().numpy()
        bad = np.argwhere(gd != od)
        print(lut, acc, env, "id mismatches", int((gi != oi).sum()), "dist mismatches", len(bad),
              [(float(gd[a, b]), float(od[a, b])) for a, b in bad[:3]])
