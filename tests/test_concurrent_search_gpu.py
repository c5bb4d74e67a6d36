"""GPU: the reference's threading contract for a built index (SURVEY 8b "Threading"; cpp/bench/ann/src/common/benchmark.hpp:296-307:
every bench thread searches the SAME index through its own copy of the algo object / its own cuvsResources_t).

A freshly built (or extended) IVF-PQ / IVF-Flat / CAGRA / brute-force index is searched by four host threads at once, each with its
own cuvsResources_t (own stream, own scratch cache). The first search of every thread races the index's lazily built caches (decode
tables and row terms of the matrix-core tail phase, the reduced-precision coarse centres, IVF-Flat's fp16 residual copy): every
thread's every answer - ids AND distances - must equal the answer of one thread searching alone afterwards.
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_THREADS = 4
ROUNDS = 3


def _clustered(n, d, q, seed, modes=64):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((modes, d)).astype(np.float32)
    x = c[rng.integers(0, modes, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    qq = c[rng.integers(0, modes, q)] + 0.3 * rng.standard_normal((q, d)).astype(np.float32)
    return x.astype(np.float32), qq.astype(np.float32)


def _race(search_one, own_streams=True):
    """search_one(resources) -> (distances, neighbors) as numpy arrays. Runs it ROUNDS times on N_THREADS threads released together,
    then once more alone; returns the list of all concurrent answers and the single-thread answer. own_streams: every thread's handle
    wraps a stream of its own (the bench's setup); otherwise all handles share torch's default stream (what the Python layer does when
    no stream is given) - legal too, the calls then simply queue behind each other."""
    import torch

    import cuvs_amd

    start = threading.Barrier(N_THREADS)
    results, errors = [[] for _ in range(N_THREADS)], []

    def worker(t):
        try:
            stream = torch.cuda.Stream() if own_streams else None
            res = cuvs_amd.common.Resources(stream=stream)  # this thread's own handle
            start.wait()
            for _ in range(ROUNDS):
                results[t].append(search_one(res))
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(N_THREADS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    alone = search_one(cuvs_amd.common.Resources())
    return [r for per_thread in results for r in per_thread], alone


def _check(all_answers, alone):
    assert len(all_answers) == N_THREADS * ROUNDS
    for d, i in all_answers:
        assert (i == alone[1]).all(), f"ids differ from the single-thread answer in {(i != alone[1]).mean():.5f} of the slots"
        assert (d == alone[0]).all(), "distances differ from the single-thread answer"


@pytest.mark.parametrize("pq_bits,coarse", [(8, "f32"), (5, "f32"), (8, "f16")])
def test_ivf_pq_index_shared_by_threads(pq_bits, coarse):
    """two-phase path (1000 queries, 32 probes): the first searches build the decode tables / row terms (and, for 5-bit codes, the
    byte-per-code copy of the lists; for the fp16 coarse search the packed centres) while the other threads ask for them"""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, q = _clustered(120_000, 64, 1000, seed=11)
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=128, pq_dim=32, pq_bits=pq_bits, kmeans_n_iters=10), torch.from_numpy(x).cuda())
    qd = torch.from_numpy(q).cuda()
    sp = ivf_pq.SearchParams(n_probes=32, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=1000,
                             **({"coarse_search_dtype": np.float16} if coarse == "f16" else {}))

    def search_one(res):
        d, i = ivf_pq.search(sp, index, qd, 10, resources=res)
        res.sync()
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    _check(*_race(search_one))
    if pq_bits == 8 and coarse == "f32":
        _check(*_race(search_one, own_streams=False))


def test_ivf_pq_extended_index_shared_by_threads():
    """extend() invalidates the derived tables: the race is on their REbuild"""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, q = _clustered(100_000, 64, 600, seed=12)
    xd = torch.from_numpy(x).cuda()
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=64, pq_dim=32, kmeans_n_iters=10), xd[:60_000])
    qd = torch.from_numpy(q).cuda()
    sp = ivf_pq.SearchParams(n_probes=16, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=600)
    ivf_pq.search(sp, index, qd, 10)  # tables of the 60k-row index exist
    ivf_pq.extend(index, xd[60_000:], torch.arange(60_000, 100_000, dtype=torch.int64, device="cuda"))

    def search_one(res):
        d, i = ivf_pq.search(sp, index, qd, 10, resources=res)
        res.sync()
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    all_answers, alone = _race(search_one)
    _check(all_answers, alone)
    assert (alone[1] >= 60_000).any(), "the extension's rows are found"


def test_ivf_pq_wide_index_shared_by_threads():
    """768 dimensions, pq_dim 64 x pq_len 12: the wide matrix-core path (ivf_pq_wide.hip) - the first searches race the build of the
    index's DECODED fp16 rows and its entry-major codebook (scan3_cache::rows16w / cbt: made under the cache's lock, published after
    the fill kernels have run)"""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, q = _clustered(60_000, 768, 600, seed=21)
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=32, pq_dim=64, kmeans_n_iters=8), torch.from_numpy(x).cuda())
    qd = torch.from_numpy(q).cuda()
    sp = ivf_pq.SearchParams(n_probes=12, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=600)

    def search_one(res):
        d, i = ivf_pq.search(sp, index, qd, 10, resources=res)
        res.sync()
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    _check(*_race(search_one))


@pytest.mark.parametrize("dtype", [np.float32, np.int8])
def test_ivf_flat_index_shared_by_threads(dtype):
    import torch
    from cuvs_amd.neighbors import ivf_flat

    x, q = _clustered(100_000, 64, 1000, seed=13)
    if dtype == np.int8:
        x, q = np.clip(np.rint(x * 30), -127, 127).astype(np.int8), np.clip(np.rint(q * 30), -127, 127).astype(np.int8)
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=128, kmeans_n_iters=10), torch.from_numpy(x).cuda())
    qd = torch.from_numpy(q).cuda()
    sp = ivf_flat.SearchParams(n_probes=32)

    def search_one(res):
        d, i = ivf_flat.search(sp, index, qd, 10, resources=res)
        res.sync()
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    _check(*_race(search_one))


def test_cagra_index_shared_by_threads():
    """single-CTA walks are deterministic (one wave per query): every thread's answer is the single-thread answer"""
    import torch
    from cuvs_amd.neighbors import cagra

    x, q = _clustered(40_000, 32, 500, seed=14, modes=1)
    index = cagra.build(cagra.IndexParams(intermediate_graph_degree=64, graph_degree=32), torch.from_numpy(x).cuda())
    qd = torch.from_numpy(q).cuda()
    sp = cagra.SearchParams(itopk_size=64, algo="single_cta")

    def search_one(res):
        d, i = cagra.search(sp, index, qd, 10, resources=res)
        res.sync()
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    _check(*_race(search_one))


def test_brute_force_index_shared_by_threads():
    import torch
    from cuvs_amd.neighbors import brute_force

    x, q = _clustered(50_000, 48, 700, seed=15)
    xd = torch.from_numpy(x).cuda()
    index = brute_force.build(xd)
    qd = torch.from_numpy(q).cuda()

    def search_one(res):
        d, i = brute_force.search(index, qd, 10, resources=res)
        res.sync()
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    _check(*_race(search_one))
