"""GPU: the world > 1 code paths of the list-sharded search, run by SEPARATE PROCESSES that share device 0.

RCCL refuses two ranks on one device, so the ranks talk through the communicator's host-staged transport
(cuvsAmdShardCommGetUniqueIdHostStaged, cuvs_amd/csrc/shm_transport.hpp) - everything else is what an 8-GPU job runs:
the query-sliced coarse search + probe all-gather, the head-bound all-reduce between the scan phases, the batch-size
all-reduce of the non-fused path, LPT list owners, cuvsAmdShardAllGatherTopK + the R-way merge, shard-local refine, and
row-range shards. Every rank's merged answer must equal the unsharded index bit for bit (reference: the sharded search of
cpp/src/neighbors/mg/snmg.cuh:248-375, whose tests need real devices: cpp/tests/neighbors/mg.cuh:647).

A rank that waits for a peer longer than CUVS_AMD_SHM_TIMEOUT_S raises (no hang); the worker processes run under a hard
time limit on top of that."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _run_world(world, out_dir, timeout=420):
    from cuvs_amd.neighbors import ivf_pq_sharded as sh

    # CUVS_AMD_WORLD_OWN_DEVICES=1 (scripts/gpu_first_8gpu.sh, a multi-GPU node): rank r on device r, the communicator over RCCL
    own = os.environ.get("CUVS_AMD_WORLD_OWN_DEVICES") == "1"
    if own:
        import torch

        if torch.cuda.device_count() < world:
            pytest.skip(f"{world} ranks on their own devices need {world} devices")
    comm_id = sh.ShardComm.unique_id(host_staged=not own).hex()
    env = dict(os.environ, CUVS_AMD_SHM_TIMEOUT_S="90")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_world_worker.py"), str(r), str(world), comm_id, str(out_dir)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"


def _same_pairs(d0, i0, d1, i1, select_min=True):
    """The same distances, and the same ids for every distance better than a query's k-th: rows with equal codes have equal
    scores, and WHICH of several rows tied at the k-th score is kept depends on the order the candidates are merged in
    ((score, probe rank, row) on one GPU, (score, rank, position) across ranks) - as in the reference's knn_merge_parts."""
    assert d0.shape == d1.shape
    assert (np.sort(d0, axis=1) == np.sort(d1, axis=1)).all()
    for q in range(d0.shape[0]):
        kth = d0[q].max() if select_min else d0[q].min()
        a = sorted((d, i) for d, i in zip(d0[q].tolist(), i0[q].tolist()) if d != kth)
        b = sorted((d, i) for d, i in zip(d1[q].tolist(), i1[q].tolist()) if d != kth)
        assert a == b, f"query {q}"
        assert len(set(i0[q].tolist())) == len(set(i1[q].tolist()))


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_device_equal_the_unsharded_index(world, tmp_path):
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq, refine
    from shard_world_worker import LIST_CASES, case_data

    _run_world(world, tmp_path)
    res = cuvs_amd.common.Resources()
    for name in LIST_CASES:
        x, q, ipk, n_probes, k, metric = case_data(name)
        xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
        full = ivf_pq.build(ivf_pq.IndexParams(metric=metric, add_data_on_build=False, **ipk), xt, resources=res)
        ivf_pq.extend(full, xt, torch.arange(len(x), dtype=torch.int64, device="cuda"), resources=res)
        sp = ivf_pq.SearchParams(n_probes=n_probes, max_internal_batch_size=32768)
        fd, fi = ivf_pq.search(sp, full, qt, k, resources=res)
        res.sync()
        fd, fi = fd.cpu().numpy(), fi.cpu().numpy()
        parts = [np.load(tmp_path / f"{name}_rank{r}.npz") for r in range(world)]
        counts = full.list_sizes.cpu().numpy().astype(np.uint64)
        for r, part in enumerate(parts):
            assert (part["counts"] == counts).all()               # the slices' histograms add up to the index's lists
            assert (part["owners"] == parts[0]["owners"]).all()   # every rank dealt the same table
            assert (part["d"] == parts[0]["d"]).all() and (part["i"] == parts[0]["i"]).all()   # replicated result
            _same_pairs(part["d"], part["i"], fd, fi, metric != "inner_product")
        loads = np.array([counts[parts[0]["owners"] == r].sum() for r in range(world)], dtype=np.float64)
        assert loads.max() <= max(float(counts.max()), 1.34 * loads.mean())   # LPT bound
        if name == "c3_two_phase":
            # shard-local refinement: the merged result is the exact re-ranking of the union of the ranks' candidates
            cand = np.concatenate([p["cand"] for p in parts], axis=1)
            od, oi = oracle.refine(x, q, cand, k)
            for part in parts:
                assert (part["refined_d"] == od).all()
                _same_pairs(part["refined_d"], part["refined_i"], od, oi)
            # ... and at least as good as refining the unsharded index's own 2k candidates
            _, ci = ivf_pq.search(sp, full, qt, 2 * k, resources=res)
            rd, _ = refine(xt, qt, ci, k=k, metric="sqeuclidean", resources=res)
            res.sync()
            assert (parts[0]["refined_d"] <= rd.cpu().numpy()).all()
        del full
    # row-range shards of IVF-Flat with every list probed: the merged answer is the exact kNN of the whole set
    rng = np.random.default_rng(12)
    x = rng.standard_normal((6001, 32)).astype(np.float32)
    q = rng.standard_normal((120, 32)).astype(np.float32)
    td, ti = oracle.exact_knn(q, x, 10)
    for r in range(world):
        part = np.load(tmp_path / f"row_shards_rank{r}.npz")
        assert (part["i"] == ti).all()


def test_a_missing_rank_raises_instead_of_hanging():
    """One rank of a two-rank communicator never shows up: creating the communicator fails after the time limit."""
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq_sharded as sh

    os.environ["CUVS_AMD_SHM_TIMEOUT_S"] = "2"
    try:
        with pytest.raises(Exception, match="shm transport"):
            sh.ShardComm(0, 2, sh.ShardComm.unique_id(host_staged=True), cuvs_amd.common.Resources())
    finally:
        del os.environ["CUVS_AMD_SHM_TIMEOUT_S"]
