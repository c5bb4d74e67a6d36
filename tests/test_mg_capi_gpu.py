"""GPU: cuvsMultiGpu{IvfFlat,IvfPq,Cagra}* (single process, host tensors).

* the reference's own pure-C driver c/tests/neighbors/run_mg_c.c (compiled unchanged by oracle/build_ref.sh) with the
  configurations and pass criteria of its gtest wrapper c/tests/neighbors/ann_mg_c.cu:300-314 (1000 queries, 5000 x 8
  uniform[0.1, 2) rows, k 16, n_probes 40 of 256 lists, eval_neighbours eps 1e-3, min_recall n_probes / n_lists for the
  IVF types and 0.9 for CAGRA): build, extend, serialize, deserialize, search on every visible GPU;
* two ranks on ONE GPU (device ids [0, 0]) so that the shard split, the id translation, the host merge, the batch
  dealing and the multi-rank file layout run on a single-GPU box.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from tests.test_reference_c_drivers_gpu import _drivers, _eval_neighbours

pytestmark = pytest.mark.gpu


class _MgTestParams(C.Structure):  # run_mg_c.c:17-27
    _fields_ = [("num_queries", C.c_int64), ("num_db_vecs", C.c_int64), ("dim", C.c_int64), ("k", C.c_int64),
                ("mode", C.c_int), ("algo", C.c_int), ("nprobe", C.c_int64), ("nlist", C.c_int64), ("metric", C.c_int)]


@pytest.mark.parametrize("mode", [0, 1], ids=["replicated", "sharded"])
@pytest.mark.parametrize("algo", ["ivf_flat", "ivf_pq", "cagra"])
def test_reference_mg_driver(algo, mode):
    lib = _drivers()
    if not hasattr(lib, "run_mg_ivf_pq_test"):
        pytest.skip("oracle/_ref/libref_c_drivers.so was built without run_mg_c.c")
    rng = np.random.default_rng(1234)
    x = (rng.random((5000, 8), dtype=np.float32) * 1.9 + 0.1).astype(np.float32)
    q = (rng.random((1000, 8), dtype=np.float32) * 1.9 + 0.1).astype(np.float32)
    nb = np.full((1000, 16), -7, dtype=np.int64)
    ds = np.full((1000, 16), -7, dtype=np.float32)
    params = _MgTestParams(1000, 5000, 8, 16, mode, {"ivf_flat": 0, "ivf_pq": 1, "cagra": 2}[algo], 40, 256, 0)
    fn = getattr(lib, f"run_mg_{algo}_test")
    fn.argtypes = [_MgTestParams] + [C.c_void_p] * 6
    fn.restype = C.c_int
    assert fn(params, x.ctypes.data, q.ctypes.data, ds.ctypes.data, nb.ctypes.data, None, None) == 0
    td, ti = oracle.exact_knn(q, x, 16)
    _eval_neighbours(nb, ds, ti, td, 1e-3, 0.9 if algo == "cagra" else 40 / 256)
    assert ((nb >= 0) & (nb < 5000)).mean() > 0.99


def _two_ranks_on_gpu0():
    from cuvs_amd.neighbors import mg

    return mg.MultiGpuResources(device_ids=[0, 0])


def _data(n, d, nq, seed):
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((32, d)).astype(np.float32) * 2
    x = (centres[rng.integers(0, 32, n)] + rng.standard_normal((n, d))).astype(np.float32)
    q = (centres[rng.integers(0, 32, nq)] + rng.standard_normal((nq, d))).astype(np.float32)
    return x, q


def test_sharded_ivf_flat_is_exact_when_every_list_is_probed(tmp_path):
    """Two shards of 3001 + 3000 rows: with n_probes = n_lists each shard search is exact, so the merged answer must be
    the exact kNN of the whole set - shard split, translation of shard-local ids and the host merge all have to be
    right. 1000 queries in batches of 300 (last one partial)."""
    from cuvs_amd.neighbors import ivf_flat, mg

    x, q = _data(6001, 16, 1000, 3)
    res = _two_ranks_on_gpu0()
    index = mg.build("ivf_flat", ivf_flat.IndexParams(n_lists=16), x, mode="sharded", resources=res)
    sp = ivf_flat.SearchParams(n_probes=16)
    d, i = mg.search(sp, index, q, 10, n_rows_per_batch=300)
    td, ti = oracle.exact_knn(q, x, 10)
    assert oracle.recall(i, ti) >= 0.999, oracle.recall(i, ti)
    np.testing.assert_allclose(d, td, rtol=1e-4, atol=1e-4)
    assert (np.diff(d, axis=1) >= 0).all()
    # one file, two index streams; a fresh handle reads it back
    path = str(tmp_path / "mg_flat.bin")
    mg.save(path, index)
    again = mg.load("ivf_flat", path, res)
    d2, i2 = mg.search(sp, again, q, 10, merge_mode="merge_on_root_rank")
    assert (i2 == i).all() and (d2 == d).all()
    # the file of a 2-rank index does not load on a 1-GPU handle (snmg.cuh:72-76)
    from cuvs_amd._lib import CuvsError

    with pytest.raises(CuvsError, match="ranks"):
        mg.load("ivf_flat", path, mg.MultiGpuResources(device_ids=[0]))


def test_replicated_ivf_flat_extend_with_ids():
    """Extend on a non-empty index needs ids (ivf_flat_build.cuh: "You must pass data indices when the index is
    non-empty"); every replica takes all new rows (snmg.cuh:178-206), so whichever replica answers finds them."""
    from cuvs_amd.neighbors import ivf_flat, mg

    x, _ = _data(4000, 16, 1, 9)
    extra, _ = _data(501, 16, 1, 10)
    res = _two_ranks_on_gpu0()
    index = mg.build("ivf_flat", ivf_flat.IndexParams(n_lists=16), x, mode="replicated", resources=res)
    mg.extend(index, extra, 100000 + np.arange(501))
    d, i = mg.search(ivf_flat.SearchParams(n_probes=16), index, extra, 1, n_rows_per_batch=100)
    assert (d[:, 0] < 1e-3).all() and (i[:, 0] == 100000 + np.arange(501)).all()
    from cuvs_amd._lib import CuvsError

    with pytest.raises(CuvsError, match="indices"):
        mg.extend(index, extra)


def test_replicated_ivf_pq_matches_the_single_gpu_index():
    """Both replicas are built from the same rows with the same (deterministic) build, so whichever replica serves a
    batch the answer is that of a single-GPU index: checks how batches are dealt out and written back."""
    import torch
    from cuvs_amd.neighbors import ivf_pq, mg

    x, q = _data(8000, 32, 1000, 5)
    ip = ivf_pq.IndexParams(n_lists=32, pq_dim=16, kmeans_trainset_fraction=1.0)
    sp = ivf_pq.SearchParams(n_probes=8)
    single = ivf_pq.build(ip, torch.from_numpy(x).cuda())
    sd, si = ivf_pq.search(sp, single, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    sd, si = sd.cpu().numpy(), si.cpu().numpy()
    res = _two_ranks_on_gpu0()
    index = mg.build("ivf_pq", ip, x, mode="replicated", resources=res)
    d, i = mg.search(sp, index, q, 10, n_rows_per_batch=300)  # 4 batches of <= 300 dealt to rank 0, 1, 0, 1
    assert (i == si).mean() > 0.999 and np.allclose(d, sd, rtol=1e-5, atol=1e-5)
    for _ in range(2):  # round robin: the whole call on rank 0, then on rank 1
        d2, i2 = mg.search(sp, index, q[:200], 10, search_mode="round_robin", n_rows_per_batch=300)
        assert (i2 == i[:200]).all() and (d2 == d[:200]).all()
    from cuvs_amd._lib import CuvsError

    with pytest.raises(CuvsError, match="round-robin"):
        mg.search(sp, index, q, 10, search_mode="round_robin", n_rows_per_batch=300)


def test_sharded_and_distributed_cagra(tmp_path):
    import torch
    from cuvs_amd.neighbors import cagra, mg

    x, q = _data(4000, 16, 300, 7)
    res = _two_ranks_on_gpu0()
    index = mg.build("cagra", cagra.IndexParams(intermediate_graph_degree=64, graph_degree=32), x, mode="sharded",
                     resources=res)
    sp = cagra.SearchParams(itopk_size=64)
    d, i = mg.search(sp, index, q, 10)
    _, ti = oracle.exact_knn(q, x, 10)
    assert oracle.recall(i, ti) >= 0.9, oracle.recall(i, ti)
    assert ((i >= 0) & (i < 4000)).all() and (i[:, 0] >= 2000).any() and (i[:, 0] < 2000).any()
    true = ((x[i] - q[:, None, :]) ** 2).sum(-1)
    np.testing.assert_allclose(d, true, rtol=1e-3, atol=1e-3)  # translated ids name the rows the distances belong to
    # a single-GPU index file spread over both ranks
    single = cagra.build(cagra.IndexParams(intermediate_graph_degree=64, graph_degree=32), torch.from_numpy(x).cuda())
    path = str(tmp_path / "cagra_single.bin")
    cagra.save(path, single)
    spread = mg.distribute("cagra", path, res)
    d2, i2 = mg.search(sp, spread, q, 10, n_rows_per_batch=100)
    assert oracle.recall(i2, ti) >= 0.9
