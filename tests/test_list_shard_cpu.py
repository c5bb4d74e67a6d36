"""CPU (gloo, world_size 2): the list-sharded IVF-PQ search - list L on rank L % world, every rank ranks all centres and
scans the probes it owns, ONE all-gather of the [Q, k] blocks, R-way merge - must equal the single-index search.
Per-rank search = the oracle on an index whose foreign lists are empty (exactly what cuvsIvfPqExtend leaves on a
sharded index); the merge rule is the CPU twin of shard_comm.hip (`merge_gathered`)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from tests.test_mg_cpu import _free_port
from tests.test_oracle_ivf_cagra import _toy_pq_index


def merge_gathered(d_parts, i_parts, k, select_min=True):
    # imported lazily: cuvs_amd needs libcuvs_c.so, which exists wherever the CPU suite runs (build() makes it)
    from cuvs_amd.neighbors.ivf_pq_sharded import merge_gathered as m

    return m(d_parts, i_parts, k, select_min)


def shard_of(ex, rank, world):
    sh = dict(ex)
    sizes = ex["list_sizes"].copy()
    codes, ids = [], []
    for L in range(len(sizes)):
        mine = L % world == rank
        if not mine:
            sizes[L] = 0
        codes.append(ex["codes"][L] if mine else ex["codes"][L][:0])
        ids.append(ex["ids"][L] if mine else ex["ids"][L][:0])
    sh.update(list_sizes=sizes, codes=codes, ids=ids)
    return sh


def _worker(rank, world, port, ex, q, k, n_probes, metric, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, i = oracle.ivf_pq_search(shard_of(ex, rank, world), q, k, n_probes, metric=metric)
        if metric == "inner_product":  # the search pads similarity results with FLT_MAX; the merge needs them to lose
            d = np.where(i == np.iinfo(np.int64).max, -np.float32(3.4028235e38), d)
        gd = [torch.empty(d.shape, dtype=torch.float32) for _ in range(world)]
        gi = [torch.empty(i.shape, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gd, torch.from_numpy(d))
        dist.all_gather(gi, torch.from_numpy(i))
        md, mi = merge_gathered([t.numpy() for t in gd], [t.numpy() for t in gi], k, metric != "inner_product")
        out[rank] = (md, mi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
def test_list_sharded_search_equals_single_index(metric):
    rng = np.random.default_rng(3)
    x, ex, _, _ = _toy_pq_index(rng, n=900, n_lists=9)
    q = rng.standard_normal((25, x.shape[1])).astype(np.float32)
    k, n_probes = 8, 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ex, q, k, n_probes, metric, out), nprocs=2, join=True)
    td, ti = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric)
    for rank in (0, 1):  # replicated result
        md, mi = out[rank]
        assert (md == td).all()
        for qi in range(len(q)):  # ids agree as sets per distance value (tie order: (distance, id) on both sides)
            assert sorted(zip(md[qi].tolist(), mi[qi].tolist())) == sorted(zip(td[qi].tolist(), ti[qi].tolist()))


def test_merge_rule():
    d = [np.array([[1.0, 2.0, 3.4e38]], np.float32), np.array([[0.5, 2.0, 2.0]], np.float32)]
    i = [np.array([[7, 9, np.iinfo(np.int64).max]]), np.array([[11, 1, 3]])]
    md, mi = merge_gathered(d, i, 3)
    assert md.tolist() == [[0.5, 1.0, 2.0]] and mi.tolist() == [[11, 7, 9]]  # the rank-0 candidate wins the tie at k


def _refine_worker(rank, world, port, ex, x, q, k, ratio, n_probes, out):
    """One rank of bench.py --config c5: its shard's k x ratio candidates, re-ranked exactly against the rows of its OWN lists
    (local row numbers -> global ids only after the refinement), then the all-gather + merge of the [Q, k] blocks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = shard_of(ex, rank, world)
        own = np.sort(np.concatenate([ids for ids in shard["ids"]]).astype(np.int64))   # global ids of the rank's rows
        local_of = {int(g): j for j, g in enumerate(own)}
        shard["ids"] = [np.array([local_of[int(g)] for g in ids], dtype=ids.dtype) for ids in shard["ids"]]   # codes under LOCAL ids
        own_rows = x[own]                                                                                    # the only rows the rank keeps
        _, ci = oracle.ivf_pq_search(shard, q, k * ratio, n_probes)
        invalid = np.iinfo(np.int64).max
        assert (ci != invalid).all()
        rd, ri = oracle.refine(own_rows, q, ci, k)
        gi_local = own[ri]
        gd = [torch.empty(rd.shape, dtype=torch.float32) for _ in range(world)]
        gi = [torch.empty(gi_local.shape, dtype=torch.int64) for _ in range(world)]
        gc = [torch.empty(ci.shape, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gd, torch.from_numpy(rd))
        dist.all_gather(gi, torch.from_numpy(gi_local))
        dist.all_gather(gc, torch.from_numpy(own[ci]))   # (for the check only: every rank's candidates under global ids)
        md, mi = merge_gathered([t.numpy() for t in gd], [t.numpy() for t in gi], k, True)
        out[rank] = (md, mi, np.concatenate([t.numpy() for t in gc], axis=1))
    finally:
        dist.destroy_process_group()


def test_shard_local_refinement_equals_refining_the_union():
    """world_size 2 over gloo: no row crosses a rank, yet the merged [Q, k] block is the exact re-ranking of the union of
    the ranks' candidates (reference recipe refine_ratio 2 - 4, cuvs_ivf_pq.yaml:17, refine_device.cuh)."""
    rng = np.random.default_rng(8)
    x, ex, _, _ = _toy_pq_index(rng, n=1200, n_lists=10)
    q = rng.standard_normal((30, x.shape[1])).astype(np.float32)
    k, ratio, n_probes = 5, 3, 6
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_refine_worker, args=(2, _free_port(), ex, x, q, k, ratio, n_probes, out), nprocs=2, join=True)
    md0, mi0, cand = out[0]
    td, ti = oracle.refine(x, q, cand, k)   # the union of both ranks' candidates, re-ranked in one go over the whole corpus
    for rank in (0, 1):  # replicated result
        md, mi, _ = out[rank]
        assert (md == td).all()
        for qi in range(len(q)):
            assert sorted(zip(md[qi].tolist(), mi[qi].tolist())) == sorted(zip(td[qi].tolist(), ti[qi].tolist()))
