"""Operating points of the headline search on SURVEY 8d's OWN generator (4 n_lists centres ~ U[-1,1)^d, points = centre +
N(0, 0.1^2 I)) and on an isotropic Gaussian: for every (n_probes, refine_ratio) the step time (search of k * ratio candidates +
cuvsRefine), QPS, recall@10 against fp64 ground truth, the candidates' own recall before the re-ranking, the scan kernels' times.
Usage: python scripts/contract_corpus_sweep.py [rows] [corpus ...]   (corpus: survey8d gaussian)"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cuvs_amd  # noqa: E402
from cuvs_amd._lib import lib  # noqa: E402
from cuvs_amd.neighbors import ivf_pq, refine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
corpora = sys.argv[2:] or ["survey8d", "gaussian"]
n_lists, nq, k, ngt = 16384, 10000, 10, 500
dev = torch.device("cuda", 0)
res = cuvs_amd.common.Resources()
out = []
for corpus in corpora:
    gen = (lambda n, s: bench.gen_rows_survey8d(n, 128, s, dev, n_lists)) if corpus == "survey8d" else (lambda n, s: bench.gen_rows_gaussian(n, 128, s, dev))
    data, queries = gen(rows, 1234), gen(nq, 4321)
    t0 = time.time()
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, pq_dim=64, pq_bits=8, kmeans_n_iters=20, kmeans_trainset_fraction=0.02), data, resources=res)
    res.sync()
    print(f"{corpus}: built in {time.time() - t0:.1f} s", flush=True)
    truth = bench.exact_topk_fp64(data, queries[:ngt], k).cpu().numpy()
    oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    od = torch.empty((nq, k), dtype=torch.float32, device=dev)
    grid = [(p, r) for p in (128, 256) for r in (2, 4, 8, 16, 32)] if corpus == "survey8d" else [(128, 8), (128, 32), (256, 32)]
    for n_probes, ratio in grid:
        kk = k * ratio
        ci = torch.empty((nq, kk), dtype=torch.int64, device=dev)
        cd = torch.empty((nq, kk), dtype=torch.float32, device=dev)
        sp = ivf_pq.SearchParams(n_probes=n_probes, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f32"], max_internal_batch_size=nq)

        def step():
            ivf_pq.search(sp, index, queries, kk, neighbors=ci, distances=cd, resources=res)
            refine(data, queries, ci, indices=oi, distances=od, metric="sqeuclidean", resources=res)

        try:
            step(); step()
            lib().cuvsAmdProfileEnable(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            lib().cuvsAmdProfileEnable(0)
            ph = {}
            for nm in (b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel", b"pq_scan_kernel"):
                v = C.c_double(0)
                lib().cuvsAmdProfileCollect(nm, C.byref(v))
                ph[nm.decode()] = round(v.value / 3, 3)
            rec = bench.recall_of(oi[:ngt].cpu().numpy(), truth)
            # how many of the true top-10 are among the kk candidates at all (what no re-ranking can repair)
            cand = ci[:ngt].cpu().numpy()
            cand_rec = float(sum(len(set(t) & set(c)) for t, c in zip(truth, cand))) / truth.size
            line = {"corpus": corpus, "n_probes": n_probes, "refine_ratio": ratio, "ms_per_step": round(dt * 1e3, 3), "qps": round(nq / dt, 1),
                    "recall_at_10": round(rec, 4), "true_top10_among_candidates": round(cand_rec, 4), "phase_ms": ph}
        except Exception as e:  # a shape the search refuses is a line, not the end of the sweep
            line = {"corpus": corpus, "n_probes": n_probes, "refine_ratio": ratio, "error": repr(e)[:200]}
        out.append(line)
        print(json.dumps(line), flush=True)
        del ci, cd
    # the ceiling of the index itself: the true top-10's lists among the probed ones (coarse recall)
    del index, data, queries
    torch.cuda.empty_cache()
print(json.dumps({"rows": rows, "sweep": out}))
