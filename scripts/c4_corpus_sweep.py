"""CAGRA recall on multi-modal corpora: which mixtures of the generator family are navigable from random seeds? For every
(modes, spread) the graph is built once (degree 64, intermediate 128) and searched at itopk 64 / 128 / 256 (k = 10, 10k queries);
recall@10 against fp64 ground truth. Usage: python scripts/c4_corpus_sweep.py [rows] [latent]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cuvs_amd  # noqa: E402
from cuvs_amd.neighbors import cagra  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
latent = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda", 0)
res = cuvs_amd.common.Resources()
nq = 10000
for modes, spread in ((1, 0.35), (64, 1.0), (256, 1.0), (1024, 0.7), (1024, 1.0), (4096, 0.7), (4096, 1.0)):
    x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
    bench.gen_rows(rows, 768, 1234, dev, latent=latent, n_modes=modes, out=x, spread=spread)
    q = torch.empty((nq, 768), dtype=torch.float16, device=dev)
    bench.gen_rows(nq, 768, 4321, dev, latent=latent, n_modes=modes, out=q, spread=spread)
    gt = bench.exact_topk_fp64(x, q[:1000], 10, chunk=250_000).cpu().numpy()
    t0 = time.time()
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res)
    res.sync()
    line = {"rows": rows, "latent": latent, "modes": modes, "spread": spread, "build_seconds": round(time.time() - t0, 1)}
    for itopk in (64, 128, 256):
        sp = cagra.SearchParams(itopk_size=itopk, algo="auto")
        nb = torch.empty((nq, 10), dtype=torch.int32, device=dev)
        dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        dt = bench.timeit(lambda: cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res), 3, 1)
        rec = bench.recall_of(nb[:1000].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt)
        line[f"itopk_{itopk}"] = {"ms": round(dt * 1e3, 3), "qps": round(nq / dt, 1), "recall_at_10": round(rec, 4)}
    print(json.dumps(line), flush=True)
    del idx, x, q
    torch.cuda.empty_cache()
