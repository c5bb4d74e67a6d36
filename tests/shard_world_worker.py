"""One RANK of the world > 1 GPU tests (tests/test_list_shard_world2_gpu.py starts `world` of these as separate processes
that SHARE device 0; the communicator uses the host-staged transport because RCCL refuses two ranks on one device).

Everything a rank of an 8-GPU job does through the C ABI runs here: the same model on every rank (deterministic k-means),
list histogram + LPT dealing + cuvsAmdIvfPqSetListOwners, extend() keeping the owned lists, the attached communicator
(probe all-gather by query slice, head-bound all-reduce, the batch-size all-reduce of the non-fused path),
cuvsAmdShardAllGatherTopK + merge, shard-local refine, and the row-range shards of IVF-Flat. Every case writes the merged
result of THIS rank to <out>/<case>_rank<r>.npz; the parent compares all ranks with the unsharded index."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case_data(name):
    """(rows, queries, index params kwargs, n_probes, k, metric) of a case: shared with the parent process."""
    if name == "small_l2":
        rng = np.random.default_rng(6)
        x = rng.random((20000, 32), dtype=np.float32) * 1.9 + 0.1
        q = rng.random((200, 32), dtype=np.float32) * 1.9 + 0.1
        return x, q, dict(n_lists=24, pq_dim=16, kmeans_n_iters=10), 6, 10, "sqeuclidean"
    if name == "small_ip_uneven":
        # inner product + very uneven lists (a third of the rows in one tight cluster): the shards' list sizes differ
        # widely, so any decision a rank derives from ITS sizes would differ between the ranks (scan3:1386)
        rng = np.random.default_rng(11)
        x = rng.random((24000, 32), dtype=np.float32) * 1.9 + 0.1
        x[:8000] = 1.0 + 0.02 * rng.standard_normal((8000, 32)).astype(np.float32)
        q = rng.random((300, 32), dtype=np.float32) * 1.9 + 0.1
        return x, q, dict(n_lists=30, pq_dim=16, kmeans_n_iters=10), 8, 30, "inner_product"
    if name == "c3_two_phase":
        # the bench kernels' shape: 128-d, pq_dim 64, >= 256 queries: head phase, bound all-reduce, matrix-core filter
        rng = np.random.default_rng(7)
        x = rng.random((40000, 128), dtype=np.float32) * 1.9 + 0.1
        q = rng.random((600, 128), dtype=np.float32) * 1.9 + 0.1
        return x, q, dict(n_lists=24, pq_dim=64, kmeans_n_iters=10), 12, 10, "sqeuclidean"
    if name == "c3_two_phase_uneven_k40":
        # k = 40 over uneven lists (mean list 1250 rows: the two-phase rule k <= 4 % of a list holds for the whole index, while
        # the ranks' own means lie on both sides of it): the decision must be the same on every rank
        rng = np.random.default_rng(8)
        x = rng.random((40000, 128), dtype=np.float32) * 1.9 + 0.1
        x[:15000] = 1.0 + 0.05 * rng.standard_normal((15000, 128)).astype(np.float32)
        q = rng.random((400, 128), dtype=np.float32) * 1.9 + 0.1
        return x, q, dict(n_lists=32, pq_dim=64, kmeans_n_iters=10), 12, 40, "sqeuclidean"
    if name == "large_k":
        # k > 256: the non-fused path, whose batch size comes from an all-reduce over the ranks
        rng = np.random.default_rng(9)
        x = rng.random((12000, 32), dtype=np.float32) * 1.9 + 0.1
        q = rng.random((64, 32), dtype=np.float32) * 1.9 + 0.1
        return x, q, dict(n_lists=16, pq_dim=16, kmeans_n_iters=10), 5, 300, "sqeuclidean"
    if name == "two_batches":
        # max_internal_batch_size below the batch: two passes, each with its own probe all-gather and bound all-reduce
        rng = np.random.default_rng(10)
        x = rng.random((30000, 64), dtype=np.float32) * 1.9 + 0.1
        q = rng.random((700, 64), dtype=np.float32) * 1.9 + 0.1
        return x, q, dict(n_lists=24, pq_dim=32, kmeans_n_iters=10), 10, 10, "sqeuclidean"
    raise KeyError(name)


LIST_CASES = ["small_l2", "small_ip_uneven", "c3_two_phase", "c3_two_phase_uneven_k40", "large_k", "two_batches"]


def main():
    rank, world, id_hex, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch

    import cuvs_amd
    from cuvs_amd.neighbors import ivf_flat, ivf_pq, ivf_pq_sharded as sh, refine, row_sharded as rs

    # ranks share device 0 (host-staged transport) unless the launcher gives every rank a device of its own (RCCL)
    torch.cuda.set_device(rank if os.environ.get("CUVS_AMD_WORLD_OWN_DEVICES") == "1" else 0)
    res = cuvs_amd.common.Resources()
    comm = sh.ShardComm(rank, world, bytes.fromhex(id_hex), res)
    for name in LIST_CASES:
        x, q, ipk, n_probes, k, metric = case_data(name)
        xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
        ids = torch.arange(len(x), dtype=torch.int64, device="cuda")
        ip = ivf_pq.IndexParams(metric=metric, add_data_on_build=False, **ipk)
        shard = sh.build(ip, xt, rank, world, resources=res)
        # every rank counts the lists of ITS slice of the corpus; the slices' histograms are summed with the communicator's
        # own all-gather (uint64 counts as two uint32 words would need another collective: the launcher's job - here every
        # rank simply counts all slices, as bench.py does on its replicated synthetic corpus)
        counts = np.zeros(ipk["n_lists"], np.uint64)
        per = (len(x) + world - 1) // world
        for r in range(world):
            sh.list_histogram(shard, xt[r * per:(r + 1) * per], counts, resources=res)
        owners = sh.deal_lists(counts, world)
        sh.set_list_owners(shard, owners, rank, world)
        half = len(x) // 2
        sh.extend(shard, xt[:half], ids[:half], resources=res)
        sh.extend(shard, xt[half:], ids[half:], resources=res)
        sizes = shard.list_sizes.cpu().numpy()
        assert (sizes == np.where(owners == rank, counts, 0)).all(), "a shard holds exactly its own lists"
        sh.attach_comm(shard, comm)
        sp = ivf_pq.SearchParams(n_probes=n_probes, max_internal_batch_size=400 if name == "two_batches" else 32768)
        d, i = sh.search(sp, shard, qt, k, comm, select_min=metric != "inner_product", resources=res)
        res.sync()
        out = dict(d=d.cpu().numpy(), i=i.cpu().numpy(), owners=owners, counts=counts)
        if name == "c3_two_phase":
            # shard-local refinement (bench.py --config c5): candidates re-ranked exactly against the rank's OWN rows before
            # the all-gather; local ids in the shard, global ids on the wire
            labels = sh.row_labels(shard, xt, resources=res).cpu().numpy()
            own = np.nonzero(owners[labels] == rank)[0]
            own_t = torch.from_numpy(own).cuda()
            own_rows = xt[own_t].contiguous()
            lshard = sh.build(ip, xt, rank, world, owners=owners, resources=res)
            sh.extend(lshard, own_rows, torch.arange(len(own), dtype=torch.int64, device="cuda"), resources=res)
            sh.attach_comm(lshard, comm)
            _, ci = ivf_pq.search(sp, lshard, qt, 2 * k, resources=res)
            rd, ri = refine(own_rows, qt, ci, k=k, metric="sqeuclidean", resources=res)
            invalid = np.iinfo(np.int64).max
            gi = torch.where(ri != invalid, own_t[torch.where(ri != invalid, ri, torch.zeros_like(ri))], ri)
            md, mi = comm.all_gather_topk(rd.contiguous(), gi.contiguous(), resources=res)
            res.sync()
            ci = ci.cpu().numpy()
            out.update(refined_d=md.cpu().numpy(), refined_i=mi.cpu().numpy(),
                       cand=np.where(ci != invalid, own[np.where(ci != invalid, ci, 0)], invalid))
            sh.attach_comm(lshard, None)
        np.savez(os.path.join(out_dir, f"{name}_rank{rank}.npz"), **out)
        sh.attach_comm(shard, None)
        del shard
    # row-range shards (the reference's SHARDED mode): IVF-Flat over the rank's rows, every list probed -> exact kNN
    rng = np.random.default_rng(12)
    n, dim, nq, k = 6001, 32, 120, 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    r0, r1 = rs.shard_rows(n, rank, world)
    idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=8, kmeans_n_iters=10), torch.from_numpy(x[r0:r1]).cuda(), resources=res)
    d, i = rs.RowShard(ivf_flat, idx, r0, comm).search(
        lambda ix, qq, kk: ivf_flat.search(ivf_flat.SearchParams(n_probes=8), ix, qq, kk, resources=res),
        torch.from_numpy(q).cuda(), k, resources=res)
    res.sync()
    np.savez(os.path.join(out_dir, f"row_shards_rank{rank}.npz"), d=d.cpu().numpy(), i=i.cpu().numpy())
    comm.close()
    print(f"rank {rank} of {world}: done", flush=True)


if __name__ == "__main__":
    main()
