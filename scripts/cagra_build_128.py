"""CAGRA build (IVF-PQ kNN graph + refine + optimize) on rows x dim fp32: usage python scripts/cagra_build_128.py [rows] [dim] [lut]
(lut: the kNN-graph search on the LUT scan kernels, CUVS_AMD_PQ_WIDE=0 - the builds of rounds 1-5)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, cuvs_amd
from cuvs_amd.neighbors import cagra
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
res = bench.comparator_handle(CUVS_AMD_PQ_WIDE=0) if (len(sys.argv) > 3 and sys.argv[3] == "lut") else cuvs_amd.common.Resources()
dev = torch.device("cuda:0")
x = bench.gen_rows(rows, dim, 1234, dev, latent=24, n_modes=4096, spread=0.7)
torch.cuda.synchronize()
t0 = time.time()
idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res)
res.sync(); torch.cuda.synchronize()
print(f"build {rows} x {dim}{' (kNN-graph search on the LUT scan kernels)' if len(sys.argv) > 3 else ''}: {time.time() - t0:.2f} s", flush=True)
