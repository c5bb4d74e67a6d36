"""graph::optimize on the GPU (cuvsAmdCagraOptimize / cuvsCagraBuild with guarantee_connectivity) against the CPU
restatement oracle/oracle_cagra_optimize.c: prune + reverse edges + merge bit-identical, the connectivity pass
bit-identical and one component (graph_core.cuh:206-470, 1186-1581, 1706-1809; cagra.hpp:193)."""
import numpy as np
import pytest
import torch

import oracle
from tests.test_oracle_cagra_optimize import clustered_knn, components

pytestmark = pytest.mark.gpu


def gpu_optimize(knn, degree, guarantee, host=False):
    from cuvs_amd.neighbors import cagra

    g = torch.from_numpy(knn.view(np.int32))
    out = cagra.optimize(g if host else g.cuda(), degree, guarantee_connectivity=guarantee)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n_clusters,per,K,degree", [(8, 30, 16, 8), (40, 12, 8, 6), (5, 64, 32, 16), (1, 500, 64, 32),
                                                       (300, 20, 16, 10)])
@pytest.mark.parametrize("guarantee", [False, True])
def test_optimize_equals_the_restatement(n_clusters, per, K, degree, guarantee):
    knn, _ = clustered_knn(n_clusters, per, K, 17 + n_clusters)
    want, left = oracle.cagra_optimize(knn, degree, guarantee)
    got = gpu_optimize(knn, degree, guarantee)
    assert (got == want).all()
    if guarantee:
        assert left == 1 and components(got) == (1, 1)
    elif n_clusters > 1:
        assert components(got)[0] == n_clusters


def test_optimize_takes_a_host_graph_and_odd_lists():
    knn, _ = clustered_knn(6, 25, 12, 3)
    knn[5, 3] = knn[5, 1]          # a duplicate id
    knn[7, 11] = 0xFFFFFFFF        # an invalid id
    knn[9, 0] = 9                  # a self edge
    for guarantee in (False, True):
        want, _ = oracle.cagra_optimize(knn, 8, guarantee)
        assert (gpu_optimize(knn, 8, guarantee, host=True) == want).all()


def test_build_with_guarantee_connectivity_searches_across_clusters():
    """Tight, far-apart clusters: without the pass a walk never leaves the clusters its random seeds fell into."""
    from cuvs_amd.neighbors import brute_force, cagra

    knn, x = clustered_knn(200, 40, 8, 11, dim=16)
    xd = torch.from_numpy(x).cuda()
    q = xd[::7].contiguous()
    _, gt = brute_force.search(brute_force.build(xd), q, 5)
    rec = {}
    for guarantee in (False, True):
        idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=16, graph_degree=8, build_algo="ivf_pq",
                                            guarantee_connectivity=guarantee), xd)
        g = idx.graph.cpu().numpy().view(np.uint32)
        weak, strong = components(g)
        if guarantee:
            assert (weak, strong) == (1, 1)
        else:
            assert weak > 1
        _, nb = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), idx, q, 5)
        torch.cuda.synchronize()
        rec[guarantee] = oracle.recall(nb.cpu().numpy(), gt.cpu().numpy())
    assert rec[True] >= rec[False]
