"""GPU: row-range shards (the reference's SHARDED mode, snmg.cuh:128-166,248-375) of IVF-Flat and CAGRA through the native
communicator: device-side id translation (cuvsAmdShardTranslateIds) + cuvsAmdShardAllGatherTopK. One GPU per box, so
the communicator has one rank (ncclAllGather + merge really run) and the two-shard case merges the per-shard blocks
with the CPU twin of the merge."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_translate_ids_on_the_device():
    import torch
    from cuvs_amd.neighbors import row_sharded as rs

    u32 = torch.tensor([[0, 5, -1], [7, 1, 2]], dtype=torch.int32, device="cuda")   # -1 = 0xffffffff: empty slot
    g = rs.translate_ids(u32, 1000)
    big = np.iinfo(np.int64).max
    assert g.cpu().tolist() == [[1000, 1005, big], [1007, 1001, 1002]]
    i64 = torch.tensor([[3, big, -1]], dtype=torch.int64, device="cuda")
    assert rs.translate_ids(i64, 10).cpu().tolist() == [[13, big, big]]


@pytest.mark.parametrize("kind", ["ivf_flat", "cagra"])
def test_one_rank_communicator_and_two_row_shards(kind):
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import cagra, ivf_flat, ivf_pq_sharded as sh, row_sharded as rs

    res = cuvs_amd.common.Resources()
    rng = np.random.default_rng(12)
    n, d, nq, k, world = 6001, 32, 120, 10, 2
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    qt = torch.from_numpy(q).cuda()
    if kind == "ivf_flat":
        def build(rows):
            return ivf_flat.build(ivf_flat.IndexParams(n_lists=8, kmeans_n_iters=10), rows, resources=res)

        def search(index, queries, kk):
            return ivf_flat.search(ivf_flat.SearchParams(n_probes=8), index, queries, kk, resources=res)  # every list: exact
        module = ivf_flat
    else:
        def build(rows):
            return cagra.build(cagra.IndexParams(intermediate_graph_degree=64, graph_degree=32), rows, resources=res)

        def search(index, queries, kk):
            return cagra.search(cagra.SearchParams(itopk_size=128), index, queries, kk, resources=res)
        module = cagra
    comm = sh.ShardComm(0, 1, sh.ShardComm.unique_id(), res)
    # (a) one rank owning every row: translate + ncclAllGather + merge must be the identity on the local answer
    full = build(torch.from_numpy(x).cuda())
    d0, i0 = search(full, qt, k)
    d1, i1 = rs.RowShard(module, full, 0, comm).search(search, qt, k, resources=res)
    res.sync()
    i0g = i0.to(torch.int64) & 0xFFFFFFFF if i0.dtype == torch.int32 else i0
    for a in range(nq):   # same (distance, id) pairs; the merge orders ties by id
        assert sorted(zip(d0[a].tolist(), i0g[a].tolist())) == sorted(zip(d1[a].tolist(), i1[a].tolist()))
    # (b) two row-range shards searched one after the other, merged by the CPU twin
    parts_d, parts_i = [], []
    for rank in range(world):
        r0, r1 = rs.shard_rows(n, rank, world)
        shard = build(torch.from_numpy(x[r0:r1]).cuda())
        dl, il = search(shard, qt, k)
        gi = rs.translate_ids(il, r0, resources=res)
        res.sync()
        parts_d.append(dl.cpu().numpy()); parts_i.append(gi.cpu().numpy())
    md, mi = sh.merge_gathered(parts_d, parts_i, k, True)
    td, ti = oracle.exact_knn(q, x, k)
    if kind == "ivf_flat":
        assert (mi == ti).all()          # every list probed on every shard: the merged answer is the exact kNN
    else:
        assert oracle.recall(mi, ti) >= 0.95
    comm.close()
