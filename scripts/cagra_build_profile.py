"""Kernel-time split of a CAGRA build (IVF-PQ kNN graph + refine + optimize) on rows x 768 fp16, for `rocprofv3 --kernel-trace --stats`.
usage: python scripts/cagra_build_profile.py [rows]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, cuvs_amd
from cuvs_amd.neighbors import cagra

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
bench.gen_rows(rows, 768, 1234, dev, latent=24, n_modes=1, out=x, spread=0.35)
torch.cuda.synchronize()
t0 = time.time()
idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res)
res.sync(); torch.cuda.synchronize()
print(f"build {rows} x 768: {time.time() - t0:.2f} s", flush=True)
