"""CPU (gloo, world_size 2): the sharded-search collective path — shard bounds, id translation, all_gather + merge
must reproduce the exact global kNN when every shard answers exactly (reference: cpp/tests/neighbors/mg.cuh:59-69
checks sharded search against naive_knn)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from cuvs_amd import mg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, x, q, k, metric, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, end = mg.shard_rows(x.shape[0], rank, world)

        def build_fn(rows):
            return rows  # the "index" of this CPU test is the shard itself

        def search_fn(index, queries, kk):
            d, i = oracle.exact_knn(queries.numpy(), index, kk, metric=metric)
            return torch.from_numpy(d), torch.from_numpy(i)

        sidx = mg.ShardedIndex.build(None, build_fn, x[start:end], x.shape[0])
        sidx.select_min = metric != "inner_product"
        d, i = sidx.search(search_fn, torch.from_numpy(q), k)
        if rank == 0:
            out["d"], out["i"] = d.numpy().copy(), i.numpy().copy()
        # results are replicated on every rank
        chk = [torch.empty_like(i) for _ in range(world)]
        dist.all_gather(chk, i)
        assert all(torch.equal(c, i) for c in chk)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
def test_sharded_search_equals_global_exact_knn(metric):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1001, 24)).astype(np.float32)  # odd size: uneven shards
    q = rng.standard_normal((37, 24)).astype(np.float32)
    k = 10
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), x, q, k, metric, out), nprocs=2, join=True)
    td, ti = oracle.exact_knn(q, x, k, metric=metric)
    assert (out["i"] == ti).all()
    np.testing.assert_allclose(out["d"], td, rtol=1e-6)


def test_shard_rows_and_translation():
    assert mg.shard_rows(10, 0, 3) == (0, 4) and mg.shard_rows(10, 2, 3) == (8, 10) and mg.shard_rows(2, 2, 3) == (2, 2)
    ids = torch.tensor([[0, 5, -1, torch.iinfo(torch.int64).max]])
    t = mg.translate_ids(ids, 100)
    assert t.tolist() == [[100, 105, -1, torch.iinfo(torch.int64).max]]


def test_merge_parts_ties_and_padding():
    d = torch.tensor([[1.0, 2.0, 2.0, 3.4e38, 0.5, 2.0]])
    i = torch.tensor([[7, 9, 3, -1, 11, 1]])
    md, mi = mg.merge_parts(d, i, 4)
    assert mi.tolist() == [[11, 7, 1, 3]] and md.tolist() == [[0.5, 1.0, 2.0, 2.0]]
