"""Graph recall of NN-descent for inner product on the reference's table shape (4000 x 1024, degree 32) against iterations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuvs_amd.neighbors import nn_descent
import cuvs_amd
res = cuvs_amd.common.Resources()
g = torch.Generator(device="cuda"); g.manual_seed(1234)
for (n, dim, deg) in ((4000, 1024, 32), (4000, 256, 32), (4000, 1024, 64)):
    x = torch.randn((n, dim), generator=g, device="cuda") * 2.0 + 0.1
    d = x.double() @ x.double().T
    ti = torch.topk(d, deg, dim=1).indices
    for iters, thr, inter in ((20, 1e-4, 2 * deg), (100, 1e-4, 2 * deg), (100, 1e-9, 2 * deg), (300, 1e-9, 2 * deg), (100, 1e-4, 4 * deg)):
        idx = nn_descent.build(nn_descent.IndexParams(metric="inner_product", graph_degree=deg, intermediate_graph_degree=inter,
                                                      max_iterations=iters, termination_threshold=thr), x, resources=res)
        gr = idx.graph.to(torch.int64) & 0xFFFFFFFF
        hit = (gr[:, :, None] == ti[:, None, :]).any(2).float().mean().item()
        print(f"n {n} dim {dim} deg {deg}: iters {iters} thr {thr:g} intermediate {inter}: recall {hit:.4f}", flush=True)
