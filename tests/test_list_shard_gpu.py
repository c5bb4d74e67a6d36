"""GPU: list-sharded IVF-PQ through the C ABI (include/cuvs_amd/shard.h). One GPU per box, so: (a) the native RCCL
communicator with world size 1 (ncclCommInitRank + ncclAllGather + the merge kernel really run); (b) two list shards
built and searched one after the other on the same device, their [Q, k] blocks merged by the CPU twin of the merge -
together they must answer exactly like the unsharded index; (c) extend() on a shard keeps only owned lists."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _data(n=20000, d=32, nq=200, seed=5):
    rng = np.random.default_rng(seed)
    return (rng.random((n, d), dtype=np.float32) * 1.9 + 0.1), (rng.random((nq, d), dtype=np.float32) * 1.9 + 0.1)


def test_world_size_one_all_gather_is_identity():
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh

    res = cuvs_amd.common.Resources()
    x, q = _data()
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    params = ivf_pq.IndexParams(n_lists=32, pq_dim=16, kmeans_n_iters=10, add_data_on_build=False)
    comm = sh.ShardComm(0, 1, sh.ShardComm.unique_id(), res)
    index = sh.build(params, xt, 0, 1, resources=res)
    sh.extend(index, xt, torch.arange(len(x), dtype=torch.int64, device="cuda"), resources=res)
    sp = ivf_pq.SearchParams(n_probes=8)
    d0, i0 = ivf_pq.search(sp, index, qt, 10, resources=res)
    d1, i1 = sh.search(sp, index, qt, 10, comm, resources=res)
    res.sync()
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
    # with the communicator attached, a batch large enough for the two-phase schedule all-reduces its bounds in between
    sh.attach_comm(index, comm)
    qb = torch.from_numpy(np.tile(q, (2, 1))).cuda()  # 400 queries: head phase + ncclAllReduce + tail phase
    sp2 = ivf_pq.SearchParams(n_probes=12)
    d2, i2 = sh.search(sp2, index, qb, 10, comm, resources=res)
    sh.attach_comm(index, None)
    d3, i3 = ivf_pq.search(sp2, index, qb, 10, resources=res)
    res.sync()
    assert torch.equal(i2, i3) and torch.equal(d2, d3)
    comm.close()


@pytest.mark.parametrize("metric,shape", [("sqeuclidean", "small"), ("inner_product", "small"), ("sqeuclidean", "c3")])
def test_two_list_shards_equal_the_unsharded_index(metric, shape):
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh

    res = cuvs_amd.common.Resources()
    # "c3": the bench kernels' shape (128-d, pq_dim 64, two-phase schedule: head scan, matrix-core filter, pool merge;
    # a shard searched alone has no bound for queries whose nearest probe is foreign: handed back to the LUT scan)
    x, q = _data(seed=6) if shape == "small" else _data(n=40000, d=128, nq=300, seed=7)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    ids = torch.arange(len(x), dtype=torch.int64, device="cuda")
    k, n_probes, world = (10, 6, 2) if shape == "small" else (10, 12, 2)
    pq_dim = 16 if shape == "small" else 64

    def params():
        return ivf_pq.IndexParams(n_lists=24, pq_dim=pq_dim, kmeans_n_iters=10, metric=metric, add_data_on_build=False)

    full = ivf_pq.build(params(), xt, resources=res)
    ivf_pq.extend(full, xt, ids, resources=res)
    sp = ivf_pq.SearchParams(n_probes=n_probes)
    fd, fi = ivf_pq.search(sp, full, qt, k, resources=res)
    res.sync()
    parts_d, parts_i, kept = [], [], 0
    for rank in range(world):
        shard = sh.build(params(), xt, rank, world, resources=res)
        sh.extend(shard, xt[:12000], ids[:12000], resources=res)   # two chunks: extend() with explicit global ids
        sh.extend(shard, xt[12000:], ids[12000:], resources=res)
        sizes = shard.list_sizes.cpu().numpy()
        assert (sizes[np.arange(24) % world != rank] == 0).all()   # foreign lists stay empty
        assert (sizes == np.where(np.arange(24) % world == rank, full.list_sizes.cpu().numpy(), 0)).all()
        kept += len(shard)
        d, i = ivf_pq.search(sp, shard, qt, k, resources=res)
        res.sync()
        d, i = d.cpu().numpy(), i.cpu().numpy()
        if metric == "inner_product":
            d = np.where(i == np.iinfo(np.int64).max, -np.float32(3.4028235e38), d)
        parts_d.append(d); parts_i.append(i)
    assert kept == len(x)
    md, mi = sh.merge_gathered(parts_d, parts_i, k, metric != "inner_product")
    fd, fi = fd.cpu().numpy(), fi.cpu().numpy()
    assert (md == fd).all()
    for qi in range(len(q)):
        assert sorted(zip(md[qi].tolist(), mi[qi].tolist())) == sorted(zip(fd[qi].tolist(), fi[qi].tolist()))


def test_lists_dealt_by_size_equal_the_unsharded_index():
    """Lists dealt to the ranks by size (cuvsAmdIvfPqListHistogram per row slice, summed; cuvsAmdShardDealLists = greedy
    LPT; cuvsAmdIvfPqSetListOwners) instead of L % world: the rows per rank are balanced and the merged shards answer
    like the unsharded index."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh

    res = cuvs_amd.common.Resources()
    rng = np.random.default_rng(11)
    # very uneven lists: a third of the rows in one tight cluster
    x = (rng.random((24000, 32), dtype=np.float32) * 1.9 + 0.1)
    x[:8000] = 1.0 + 0.02 * rng.standard_normal((8000, 32)).astype(np.float32)
    q = (rng.random((150, 32), dtype=np.float32) * 1.9 + 0.1)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    ids = torch.arange(len(x), dtype=torch.int64, device="cuda")
    k, n_probes, world, n_lists = 10, 8, 3, 30

    def params():
        return ivf_pq.IndexParams(n_lists=n_lists, pq_dim=16, kmeans_n_iters=10, add_data_on_build=False)

    full = ivf_pq.build(params(), xt, resources=res)
    # every "rank" counts the lists of its slice of the rows; the launcher sums the histograms
    counts = np.zeros(n_lists, np.uint64)
    for r in range(world):
        counts += sh.list_histogram(full, xt[r * 8000:(r + 1) * 8000], resources=res)
    ivf_pq.extend(full, xt, ids, resources=res)
    assert (counts == full.list_sizes.cpu().numpy().astype(np.uint64)).all()
    owners = sh.deal_lists(counts, world)
    loads = np.array([counts[owners == r].sum() for r in range(world)], dtype=np.float64)
    modulo = np.array([counts[np.arange(n_lists) % world == r].sum() for r in range(world)], dtype=np.float64)
    assert loads.max() <= modulo.max()            # never worse than L % world ...
    assert loads.max() <= max(counts.max(), 1.34 * loads.mean())   # ... and within the LPT bound (4/3 - 1/(3 world))
    sp = ivf_pq.SearchParams(n_probes=n_probes)
    fd, fi = ivf_pq.search(sp, full, qt, k, resources=res)
    res.sync()
    parts_d, parts_i = [], []
    for rank in range(world):
        shard = sh.build(params(), xt, rank, world, owners=owners, resources=res)
        sh.extend(shard, xt, ids, resources=res)
        sizes = shard.list_sizes.cpu().numpy()
        assert (sizes == np.where(owners == rank, counts, 0)).all()
        d, i = ivf_pq.search(sp, shard, qt, k, resources=res)
        res.sync()
        parts_d.append(d.cpu().numpy()); parts_i.append(i.cpu().numpy())
    md, mi = sh.merge_gathered(parts_d, parts_i, k, True)
    fd, fi = fd.cpu().numpy(), fi.cpu().numpy()
    assert (md == fd).all()
    for qi in range(len(q)):
        assert sorted(zip(md[qi].tolist(), mi[qi].tolist())) == sorted(zip(fd[qi].tolist(), fi[qi].tolist()))


def test_shard_local_refinement_equals_refining_the_union():
    """bench.py --config c5 (SURVEY 8e / refine_device.cuh): every rank keeps the int8 rows of its own lists
    (cuvsAmdIvfPqRowLabels tells it which), stores their codes under LOCAL ids, re-ranks its k x refine_ratio candidates
    exactly against its own rows (cuvsRefine) and hands global ids to the all-gather - no row moves between ranks, and the
    merged result is the exact re-ranking of the union of the ranks' candidates."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh, refine

    res = cuvs_amd.common.Resources()
    rng = np.random.default_rng(21)
    n, d, nq, k, ratio, world, n_lists = 30000, 96, 300, 10, 2, 2, 24
    x = rng.integers(1, 20, size=(n, d), dtype=np.int8)   # the reference's int8 generator (ann_ivf_pq.cuh:150-168)
    q = rng.integers(1, 20, size=(nq, d), dtype=np.int8)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    kk = k * ratio

    def params():
        return ivf_pq.IndexParams(n_lists=n_lists, pq_dim=64, kmeans_n_iters=10, add_data_on_build=False)

    sp = ivf_pq.SearchParams(n_probes=8)
    invalid = np.iinfo(np.int64).max
    parts_d, parts_i, union = [], [], []
    for rank in range(world):
        shard = sh.build(params(), xt, rank, world, resources=res)   # same rows -> same model on every rank
        labels = sh.row_labels(shard, xt, resources=res)
        own = torch.nonzero(labels % world == rank).flatten()
        own_rows = xt[own].contiguous()
        sh.extend(shard, own_rows, torch.arange(len(own), dtype=torch.int64, device="cuda"), resources=res)
        res.sync()
        assert len(shard) == len(own)   # every kept row belongs to an owned list
        counts = np.bincount(labels.cpu().numpy()[own.cpu().numpy()], minlength=n_lists)
        assert (shard.list_sizes.cpu().numpy() == counts).all()
        _, ci = ivf_pq.search(sp, shard, qt, kk, resources=res)                       # local ids
        rd, ri = refine(own_rows, qt, ci, k=k, metric="sqeuclidean", resources=res)   # exact, against the rank's own rows
        res.sync()
        gmap = own.cpu().numpy()
        ci, ri, rd = ci.cpu().numpy(), ri.cpu().numpy(), rd.cpu().numpy()
        parts_i.append(np.where(ri != invalid, gmap[np.where(ri != invalid, ri, 0)], invalid))
        parts_d.append(rd)
        union.append(np.where(ci != invalid, gmap[np.where(ci != invalid, ci, 0)], invalid))
    md, mi = sh.merge_gathered(parts_d, parts_i, k, True)
    # the same candidates - the union over the ranks - re-ranked in one go by the oracle against the whole corpus
    cand = np.concatenate(union, axis=1)
    assert (cand != invalid).all()
    od, oi = oracle.refine(x, q, cand, k)
    assert (md == od).all()
    xf, qf = x.astype(np.float32), q.astype(np.float32)
    for qi in range(nq):
        # integer distances tie: below the k-th distance the ids are the oracle's, at it any candidate of that distance serves
        kth = od[qi, -1]
        assert set(mi[qi][md[qi] < kth].tolist()) == set(oi[qi][od[qi] < kth].tolist())
        assert len(set(mi[qi].tolist())) == k and set(mi[qi].tolist()) <= set(cand[qi].tolist())
        assert (((xf[mi[qi]] - qf[qi]) ** 2).sum(1) == md[qi]).all()
