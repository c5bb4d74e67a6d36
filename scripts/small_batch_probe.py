#!/usr/bin/env python3
"""Small batches on the headline index (100M x 128, n_probes 128, k 20): one-phase LUT scan (the default below 256 queries) against
the two-phase schedule forced by CUVS_AMD_PQ_HEAD_PROBES=1 - where should the threshold sit? Search only (no refine)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from cuvs_amd.neighbors import ivf_pq  # noqa: E402
import cuvs_amd  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
res = cuvs_amd.common.Resources()
res_h = bench.comparator_handle(CUVS_AMD_PQ_HEAD_PROBES=1)
data = bench.gen_rows(rows, 128, seed=1234, device=dev)
queries = bench.gen_rows(1024, 128, seed=4321, device=dev)
ip = ivf_pq.IndexParams(n_lists=16384 if rows >= 50_000_000 else 4096, metric="sqeuclidean", pq_dim=64, pq_bits=8, kmeans_n_iters=20,
                        kmeans_trainset_fraction=0.02)
index = ivf_pq.build(ip, data, resources=res)
res.sync()
out = []
for nb in (8, 16, 32, 64, 100, 128, 192, 255, 256, 512):
    qs = queries[:nb].contiguous()
    sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f32"],
                             max_internal_batch_size=nb)
    line = {"batch": nb}
    keep = {}
    for name, r in (("one_phase_default" if nb < 256 else "default", res), ("two_phase_forced", res_h)):
        b_i = torch.empty((nb, 20), dtype=torch.int64, device=dev)
        b_d = torch.empty((nb, 20), dtype=torch.float32, device=dev)
        dt = bench.timeit(lambda: ivf_pq.search(sp, index, qs, 20, neighbors=b_i, distances=b_d, resources=r), 40, 8)
        line[name + "_ms"] = round(dt * 1e3, 3)
        keep[name] = (b_i.clone(), b_d.clone())
    a, b = list(keep.values())
    line["ids_equal"] = bool(torch.equal(a[0], b[0]))
    line["distances_equal"] = bool(torch.equal(a[1], b[1]))
    out.append(line)
    print(json.dumps(line), flush=True)
