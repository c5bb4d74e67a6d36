"""CPU (gloo, world size 2): twins of the row-range sharded searches that the GPU path runs over the native communicator
(cuvs_amd/neighbors/row_sharded.py; reference SHARDED mode, cpp/src/neighbors/mg/snmg.cuh:248-375): every rank searches
its shard with the CPU oracle - an IVF-Flat index / a CAGRA graph built on the CPU - translates the ids by its row
offset, ONE all_gather of the [Q, k] blocks, and the merge rule of shard_comm.hip (merge_gathered). Checked against the
single-process computation of the same thing and against exact kNN of the whole corpus."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from cuvs_amd import mg
from cuvs_amd.neighbors.ivf_pq_sharded import merge_gathered

BIG = np.iinfo(np.int64).max


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_index(rows, n_lists):
    """An IVF-Flat index in the oracle's exported form, built on the CPU (balanced k-means centres, L2 argmin lists)."""
    centers = oracle.kmeans_balanced_fit(rows, n_lists, n_iters=5, hierarchical=False)[0]
    lab = np.argmin(oracle.pairwise(rows, centers), axis=1)
    ids = [np.nonzero(lab == L)[0].astype(np.int64) for L in range(n_lists)]
    return dict(centers=centers, list_sizes=np.array([len(i) for i in ids], np.uint32), rows=[rows[i] for i in ids], ids=ids)


def _knn_graph(rows, degree):
    _, nb = oracle.exact_knn(rows, rows, degree + 1)
    g = np.empty((len(rows), degree), np.uint32)
    for r in range(len(rows)):
        g[r] = [v for v in nb[r] if v != r][:degree]
    return g


def _local_search(kind, rows, q, k):
    if kind == "ivf_flat":
        return oracle.ivf_flat_search(_flat_index(rows, 6), q, k, 6)   # every list probed: exact on the shard
    return oracle.cagra_search(rows, _knn_graph(rows, 16), q, k, itopk_size=64)


def _worker(rank, world, port, kind, x, q, k, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r0, r1 = mg.shard_rows(len(x), rank, world)
        d, i = _local_search(kind, x[r0:r1], q, k)
        gi = mg.translate_ids(torch.from_numpy(i), r0).numpy()       # local row -> global row; empty slots stay INT64_MAX
        dt, it = torch.from_numpy(d), torch.from_numpy(gi)
        gd = [torch.empty_like(dt) for _ in range(world)]
        gj = [torch.empty_like(it) for _ in range(world)]
        dist.all_gather(gd, dt)                                       # the one collective of the data path
        dist.all_gather(gj, it)
        md, mi = merge_gathered([t.numpy() for t in gd], [t.numpy() for t in gj], k, True)
        out[rank] = (md.copy(), mi.copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["ivf_flat", "cagra"])
def test_two_row_shards_over_gloo(kind):
    rng = np.random.default_rng(21)
    x = rng.standard_normal((1501, 16)).astype(np.float32)   # odd size: uneven shards
    q = rng.standard_normal((40, 16)).astype(np.float32)
    k, world = 8, 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, x, q, k, out), nprocs=world, join=True)
    (d0, i0), (d1, i1) = out[0], out[1]
    assert (i0 == i1).all() and (d0 == d1).all()                      # replicated on every rank
    # the same computation in one process
    parts = []
    for rank in range(world):
        r0, r1 = mg.shard_rows(len(x), rank, world)
        d, i = _local_search(kind, x[r0:r1], q, k)
        parts.append((d, np.where(i == BIG, BIG, i + r0)))
    md, mi = merge_gathered([p[0] for p in parts], [p[1] for p in parts], k, True)
    assert (mi == i0).all() and (md == d0).all()
    td, ti = oracle.exact_knn(q, x, k)
    if kind == "ivf_flat":
        assert (i0 == ti).all()
    else:
        assert oracle.recall(i0, ti) >= 0.9
