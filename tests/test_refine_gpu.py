"""GPU: cuvsRefine vs its CPU twin (bit-exact) and vs exact kNN (refining ALL rows must equal exact search)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "inner_product", "cosine"])
@pytest.mark.parametrize("n,d,m,n_cand,k", [(2000, 64, 50, 40, 10), (500, 7, 20, 100, 100), (3000, 200, 10, 33, 5)])
def test_refine_matches_oracle(metric, n, d, m, n_cand, k):
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((m, d)).astype(np.float32)
    cand = rng.integers(0, n, size=(m, n_cand)).astype(np.int64)
    cand[0, 0] = -1  # invalid ids are skipped
    cand[1, 1] = n + 5
    gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=k,
                    metric=metric)
    torch.cuda.synchronize()
    od, oi = oracle.refine(x, q, cand, k, metric=metric)
    assert (gi.cpu().numpy() == oi).all()
    assert (gd.cpu().numpy() == od).all()


def test_refine_all_rows_is_exact_search():
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(0)
    x = rng.standard_normal((512, 32)).astype(np.float32)
    q = rng.standard_normal((16, 32)).astype(np.float32)
    cand = np.tile(np.arange(512, dtype=np.int64), (16, 1))
    gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=10)
    torch.cuda.synchronize()
    td, ti = oracle.exact_knn(q, x, 10)
    assert oracle.recall(gi.cpu().numpy(), ti) == 1.0
    np.testing.assert_allclose(gd.cpu().numpy(), td, rtol=1e-4, atol=1e-4)
