"""GPU: cuvsPairwiseDistance through the Python mirror of cuvs.distance.pairwise_distance — row-major bit for bit
against the oracle, column-major inputs/outputs (c/src/distance/pairwise_distance.cpp:93-121) against the row-major
result, scipy cdist as the outside reference (python/cuvs/cuvs/tests/test_distance.py:39-70 uses the same)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _col_major(t):
    return t.t().contiguous().t()


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "inner_product", "cosine"])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("m,n,k", [(300, 200, 33), (1000, 70, 128), (65, 1, 7)])
def test_row_and_column_major(metric, dtype, m, n, k):
    import torch
    from scipy.spatial.distance import cdist
    from cuvs_amd.distance import pairwise_distance

    rng = np.random.default_rng(m + n + k)
    x = rng.random((m, k)).astype(dtype)
    y = rng.random((n, k)).astype(dtype)
    tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    row = pairwise_distance(tx, ty, metric=metric)
    assert row.is_contiguous() and row.shape == (m, n)
    if dtype == np.float32:
        assert (row.cpu().numpy() == oracle.pairwise(x, y, metric=metric)).all()
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    want = x64 @ y64.T if metric == "inner_product" else cdist(x64, y64, metric)
    assert np.allclose(row.cpu().numpy(), want, rtol=1e-3, atol=2e-3)
    col = pairwise_distance(_col_major(tx), _col_major(ty), metric=metric)
    assert col.shape == (m, n) and col.stride() == (1, m)  # column-major result, like the inputs
    # same dot products (the MFMA fma chain commutes), the two norms are added in the other order: 1-2 ulp
    assert np.allclose(col.cpu().numpy(), row.cpu().numpy(), rtol=1e-5, atol=1e-5)
    out = torch.empty((n, m), dtype=torch.float32, device="cuda").t()
    assert pairwise_distance(_col_major(tx), _col_major(ty), out=out, metric=metric) is out
    assert (out == col).all()


def test_mixed_layouts_and_dtypes_are_rejected():
    import torch
    from cuvs_amd._lib import CuvsError
    from cuvs_amd.distance import pairwise_distance

    x = torch.rand(50, 8, device="cuda")
    y = torch.rand(40, 8, device="cuda")
    with pytest.raises(CuvsError, match="same layout"):
        pairwise_distance(x, _col_major(y))
    with pytest.raises(ValueError, match="same dtypes"):
        pairwise_distance(x, y.half())
    with pytest.raises(ValueError, match="same number of columns"):
        pairwise_distance(x, torch.rand(40, 9, device="cuda"))
    with pytest.raises(CuvsError, match="unsupported metric"):
        pairwise_distance(x, y, metric="canberra")
