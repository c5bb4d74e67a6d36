"""GPU: fixed parity cases at the shapes bench.py times ("next round" item 1).

* C3 shape - IVF-PQ with lists of thousands of rows (pq_dim 64 / 8 bit / pq_len 2: pq_scan_kernel head phase + the
  matrix-core filter / re-score / pool merge of ivf_pq_scan3.hip in the tail phase), every LUT / score type the bench
  runs, against the oracle; against the same search with the tail phase on pq_scan2_kernel (CUVS_AMD_PQ_SCAN3=0: LDS
  filter LUT, multi-block tickets, survivor queues) and on pq_scan_kernel (+ CUVS_AMD_PQ_SCAN2=0); with pq_scan2's
  survivor queues shrunk until they overflow (CUVS_AMD_PQ_QCAP) and with the matrix-core filter's survivor list shrunk
  until queries are handed back to the LUT scan (CUVS_AMD_PQ3_SURV_CAP);
* C4 shape - CAGRA 768-d fp16, graph degree 64, itopk 64: single_cta bit-exact against oracle.cagra_search, multi_cta
  and auto by recall against exact kNN;
* C5 shape - IVF-PQ on 96-d int8 rows (pq_dim 64 -> rot_dim 128, the kDivisor scaling of ann_utils.cuh:134-160) with
  the reference's int8 generator (uniformInt[1, 20), ann_ivf_pq.cuh:150-168): parity + the reference's recall bound.
"""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

_LUTS = {"f32": np.float32, "f16": np.float16, "fp8": np.uint8}


def _mixture(n, d, q, seed, modes=64, latent=16, sigma=0.35):
    """Gaussian mixture in a latent subspace embedded in R^d (what bench.py draws): neighbours are much closer than the
    bulk of a probed list, so the k-th bounds prune as they do at the bench shape."""
    rng = np.random.default_rng(seed)
    basis = rng.standard_normal((latent, d)).astype(np.float32) / math.sqrt(latent)
    centres = rng.standard_normal((modes, latent)).astype(np.float32) * 2.0

    def draw(m):
        z = centres[rng.integers(0, modes, size=m)] + sigma * rng.standard_normal((m, latent)).astype(np.float32)
        return (z @ basis + 0.02 * rng.standard_normal((m, d)).astype(np.float32)).astype(np.float32)

    return draw(n), draw(q)


def _pq_build(x, **kw):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    return ivf_pq.build(ivf_pq.IndexParams(**kw), torch.from_numpy(x).cuda())


def _pq_search(index, q, k, **kw):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    d, i = ivf_pq.search(ivf_pq.SearchParams(**kw), index, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy()


@pytest.fixture(scope="module")
def big_lists():
    """200k x 128, 16 lists of ~12.5k rows (the bench's lists hold ~6.1k), 512 queries: two-phase schedule, 8-pair items,
    several 4-tile filter blocks per wave."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(200_000, 128, 512, seed=2024)
    index = _pq_build(x, n_lists=16, pq_dim=64, pq_bits=8, kmeans_n_iters=10, kmeans_trainset_fraction=0.2)
    return x, q, index, ivf_pq.export_for_oracle(index)


@pytest.mark.parametrize("lut,acc", [("f16", "f16"), ("f16", "f32"), ("fp8", "f16"), ("f32", "f32")])
def test_c3_shape_lists_of_thousands_of_rows(big_lists, lut, acc, monkeypatch):
    x, q, index, ex = big_lists
    k, n_probes = 20, 12  # more than 8 probes and 256+ queries: two-phase schedule
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi = _pq_search(index, q, k, **kw)  # tail phase: matrix-core filter + re-score
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    # a survivor list of 1000 entries: most queries are handed back to the LUT scan, pair by pair
    monkeypatch.setenv("CUVS_AMD_PQ3_SURV_CAP", "1000")
    hd, hi = _pq_search(index, q, k, **kw)
    assert (gi == hi).all() and (gd == hd).all()
    monkeypatch.delenv("CUVS_AMD_PQ3_SURV_CAP")
    # the tail phase on pq_scan2_kernel (filter LUT in LDS)
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (gi == si).all() and (gd == sd).all()
    # survivor queues of 64 rows: every group overflows and scores every row of the list
    monkeypatch.setenv("CUVS_AMD_PQ_QCAP", "64")
    qd, qi = _pq_search(index, q, k, **kw)
    assert (gi == qi).all() and (gd == qd).all()
    monkeypatch.delenv("CUVS_AMD_PQ_QCAP")
    # the tail phase on pq_scan_kernel (no filter stage at all)
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN2", "0")
    pd, pi = _pq_search(index, q, k, **kw)
    assert (gi == pi).all() and (gd == pd).all()


@pytest.mark.parametrize("d,pq_dim,metric,lut,acc", [
    (64, 64, "sqeuclidean", "f16", "f32"),     # pq_len 1, 4 K steps
    (96, 96, "sqeuclidean", "f16", "f16"),     # pq_len 1, 6 code chunks
    (128, 128, "sqeuclidean", "f32", "f32"),   # pq_len 1, 8 K steps
    (128, 32, "sqeuclidean", "f16", "f32"),    # pq_len 4, 8 K steps
    (128, 32, "inner_product", "f16", "f32"),  # pq_len 4, no row term, survivors by the thousand
    (192, 48, "cosine", "f16", "f32"),         # pq_len 4, 12 K steps: two query groups per unit
    (256, 64, "sqeuclidean", "fp8", "f16"),    # pq_len 4, 16 K steps
    (128, 16, "sqeuclidean", "f16", "f32"),    # pq_len 8, 8 K steps
    (256, 32, "sqeuclidean", "f32", "f32"),    # pq_len 8, 16 K steps
])
def test_c3_shape_every_pq_len(d, pq_dim, metric, lut, acc, monkeypatch):
    """pq_filter4_kernel<NCH, PL>: 8-bit codes with pq_len 1 / 4 / 8 (a code byte = 1 / 4 / 8 fp16 values of the decode
    table: 2 / 8 / 16-byte gathers, 8 / 2 / 1 per K step), head LUT and re-score for any pq_len - ids and distances
    identical to the oracle and to the LUT scan kernels (CUVS_AMD_PQ_SCAN3=0)."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(100_000, d, 400, seed=d + pq_dim, modes=48)
    index = _pq_build(x, n_lists=32, pq_dim=pq_dim, pq_bits=8, metric=metric, kmeans_n_iters=8, kmeans_trainset_fraction=0.2)
    ex = ivf_pq.export_for_oracle(index)
    k, n_probes = 20, 12
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (si == oi).all() and (sd == od).all()


@pytest.mark.parametrize("d,pq_dim,pq_bits,metric,lut,acc", [
    (128, 64, 5, "sqeuclidean", "f16", "f32"),
    (128, 64, 4, "sqeuclidean", "f16", "f16"),
    (96, 48, 6, "sqeuclidean", "f32", "f32"),
    (96, 96, 7, "sqeuclidean", "f16", "f32"),   # pq_len 1
    (128, 32, 5, "inner_product", "f16", "f32"),  # pq_len 4
    (128, 64, 6, "inner_product", "f16", "f32"),  # pq_len 2, inner product: pq_filter_kernel
])
def test_c3_shape_codes_of_fewer_than_8_bits(d, pq_dim, pq_bits, metric, lut, acc, monkeypatch):
    """pq_bits 4 .. 7 on the matrix-core tail phase: head kernel, filter, re-score and row terms read a one-byte-per-code copy
    of the bit-packed lists (pq3_codes), the decode table keeps 256 slots per subspace - ids and distances identical to the
    oracle and to the LUT scan kernels (generic bit extraction, CUVS_AMD_PQ_SCAN3=0)."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(100_000, d, 400, seed=d + pq_dim + pq_bits, modes=48)
    index = _pq_build(x, n_lists=32, pq_dim=pq_dim, pq_bits=pq_bits, metric=metric, kmeans_n_iters=8, kmeans_trainset_fraction=0.2)
    ex = ivf_pq.export_for_oracle(index)
    k, n_probes = 20, 12
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (si == oi).all() and (sd == od).all()
    # rows added after the copy was made: it is rebuilt
    monkeypatch.delenv("CUVS_AMD_PQ_SCAN3")
    import torch
    x2, _ = _mixture(5_000, d, 1, seed=7, modes=48)
    index = ivf_pq.extend(index, torch.from_numpy(x2).cuda(), torch.arange(100_000, 105_000, dtype=torch.int64).cuda())
    ex = ivf_pq.export_for_oracle(index)
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all() and (gd == od).all()


@pytest.mark.parametrize("d,pq_dim,pq_bits,metric,lut,acc", [
    (128, 64, 8, "sqeuclidean", "f16", "f32"),
    (128, 64, 8, "inner_product", "f16", "f32"),
    (128, 32, 8, "sqeuclidean", "f32", "f32"),   # pq_len 4
    (96, 96, 6, "sqeuclidean", "f16", "f16"),    # pq_len 1, 6-bit codes
    (128, 16, 8, "cosine", "f16", "f32"),        # pq_len 8
])
def test_c3_shape_per_cluster_codebooks(d, pq_dim, pq_bits, metric, lut, acc, monkeypatch):
    """codebook_gen::PER_CLUSTER on the matrix-core tail phase: pq_filter4_kernel<.., PC> keeps the decode table of the
    current unit's list in a per-wave LDS region; head LUT, row terms and re-score index the codebook by list - ids and
    distances identical to the oracle and to the LUT scan kernels."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(100_000, d, 400, seed=3 * d + pq_dim + pq_bits, modes=48)
    index = _pq_build(x, n_lists=32, pq_dim=pq_dim, pq_bits=pq_bits, metric=metric, kmeans_n_iters=8, kmeans_trainset_fraction=0.2,
                      codebook_kind="cluster")
    ex = ivf_pq.export_for_oracle(index, per_cluster=True)
    k, n_probes = 20, 12
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (si == oi).all() and (sd == od).all()


@pytest.mark.parametrize("k", [129, 200, 256])
def test_c3_shape_k_up_to_256(big_lists, k, monkeypatch):
    """k = 129 .. 256 (the IVF-PQ searches of a CAGRA build ask for 2 x intermediate_graph_degree = 256 candidates,
    cagra_build.cuh:120-190) on the two-phase path: head kernel with one group minimum per thread and 1024-entry candidate
    buffers, pool merge with three / four ranks per lane - identical to the oracle and to the LUT scan kernels."""
    x, q, index, ex = big_lists
    n_probes = 12
    kw = dict(n_probes=n_probes, lut_dtype=np.float16, internal_distance_dtype=np.float32)
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, lut="f16", acc="f32")
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (si == oi).all() and (sd == od).all()


@pytest.mark.parametrize("n_queries,n_lists", [(300, 96), (260, 32), (700, 48), (1500, 24)])
def test_c3_shape_query_groups_of_every_width(n_queries, n_lists, monkeypatch):
    """pq_filter4_kernel holds up to four groups of 32 queries per work unit and is instantiated per group count: batches
    and list counts chosen so that the lists are probed by ~35 / ~90 / ~160 / ~690 queries (one .. four groups, several
    units per list chunk, lists cut into row chunks) - ids and distances identical to the oracle, to the round-3 filter
    (CUVS_AMD_PQ_FILTER4=0) and to the LUT scan."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(120_000, 128, n_queries, seed=n_queries + n_lists, modes=48)
    index = _pq_build(x, n_lists=n_lists, pq_dim=64, pq_bits=8, kmeans_n_iters=8, kmeans_trainset_fraction=0.2)
    ex = ivf_pq.export_for_oracle(index)
    k, n_probes = 20, 12
    kw = dict(n_probes=n_probes, lut_dtype=np.float16, internal_distance_dtype=np.float32)
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, lut="f16", acc="f32")
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_FILTER4", "0")
    fd, fi = _pq_search(index, q, k, **kw)
    assert (fi == oi).all() and (fd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (si == oi).all() and (sd == od).all()


@pytest.mark.parametrize("k", [100, 128])
def test_c3_shape_k_beyond_64(big_lists, k, monkeypatch):
    """k = 100 (the usual second setting of ANN benchmarks) and 128: two ranks per lane in the pool merge, groups of two
    threads in the head kernel's threshold."""
    x, q, index, ex = big_lists
    kw = dict(n_probes=12, lut_dtype=np.float16, internal_distance_dtype=np.float32)
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, 12, lut="f16", acc="f32")
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (gi == si).all() and (gd == sd).all()


def test_c3_shape_cold_bounds_overflow_the_default_queues(big_lists, monkeypatch):
    """Uniform queries far from every mode: bounds stay loose, most rows survive the filter and the default queues
    (3072 rows per group for 12.5k-row lists) overflow without any test hook."""
    x, _, index, ex = big_lists
    rng = np.random.default_rng(5)
    q = (rng.random((300, 128), dtype=np.float32) * 4.0 - 2.0).astype(np.float32)
    od, oi = oracle.ivf_pq_search(ex, q, 64, 12, lut="f16", acc="f32")
    for scan3 in ("1", "0"):  # matrix-core filter (pools fill up: queries handed back) / pq_scan2 (queues overflow)
        monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", scan3)
        gd, gi = _pq_search(index, q, 64, n_probes=12, lut_dtype=np.float16, internal_distance_dtype=np.float32)
        assert (gi == oi).all() and (gd == od).all()


def test_c3_shape_recall_with_refine(big_lists):
    """bench.py's step: search(2k) + refine(k) at fp16 LUT / fp32 score reaches the headline's recall bar on the mixture."""
    import torch
    from cuvs_amd.neighbors import refine

    x, q, index, _ = big_lists
    k = 10
    _, gi = _pq_search(index, q, 2 * k, n_probes=8, lut_dtype=np.float16, internal_distance_dtype=np.float32)
    _, ri = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(gi).cuda(), k)
    torch.cuda.synchronize()
    _, ti = oracle.exact_knn(q, x, k)
    assert oracle.recall(ri.cpu().numpy(), ti) >= 0.9


@pytest.mark.parametrize("metric", ["inner_product", "cosine"])
@pytest.mark.parametrize("lut,acc", [("f16", "f32"), ("f16", "f16"), ("fp8", "f16"), ("f32", "f32")])
def test_c3_shape_inner_product_and_cosine(metric, lut, acc, monkeypatch):
    """Signed LUT entries have no early stop in the LUT scan; the matrix-core filter works on full-score bounds, so the
    two-phase schedule (head scan -> bounds -> filter / re-score / pool merge) serves inner product and cosine too."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(80_000, 128, 384, seed=77)
    x += 0.5  # off-centre: inner products of both signs and a wide spread of norms
    q += 0.5
    index = _pq_build(x, n_lists=16, pq_dim=64, pq_bits=8, kmeans_n_iters=10, kmeans_trainset_fraction=0.3, metric=metric)
    ex = ivf_pq.export_for_oracle(index)
    k, n_probes = 20, 12
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")  # single-phase LUT scan
    sd, si = _pq_search(index, q, k, **kw)
    assert (gi == si).all() and (gd == sd).all()


@pytest.mark.parametrize("dim,pq_dim,metric,lut,acc", [(32, 16, "sqeuclidean", "f16", "f32"), (64, 32, "sqeuclidean", "f16", "f32"),
                                                        (96, 48, "sqeuclidean", "f32", "f32"), (96, 48, "inner_product", "f16", "f16"),
                                                        (160, 80, "sqeuclidean", "f16", "f32"), (192, 96, "sqeuclidean", "fp8", "f16"),
                                                        (256, 128, "sqeuclidean", "f16", "f32"), (256, 128, "cosine", "f16", "f32")])
def test_matrix_core_tail_phase_at_other_pq_dims(dim, pq_dim, metric, lut, acc, monkeypatch):
    """pq_len 2 with 1 .. 8 code chunks per row (pq_dim 16 .. 128): the decode / MFMA filter is instantiated per chunk
    count (two query groups and double-buffered rows up to pq_dim 64, one group beyond), the head kernel and the
    re-scoring take pq_dim at run time."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(50_000, dim, 320, seed=dim + pq_dim)
    if metric != "sqeuclidean":
        x += 0.4
        q += 0.4
    index = _pq_build(x, n_lists=16, pq_dim=pq_dim, pq_bits=8, kmeans_n_iters=8, kmeans_trainset_fraction=0.3, metric=metric)
    assert index.pq_dim == pq_dim and index.pq_len == 2
    ex = ivf_pq.export_for_oracle(index)
    k, n_probes = 16, 10
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi = _pq_search(index, q, k, **kw)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_SCAN3", "0")
    sd, si = _pq_search(index, q, k, **kw)
    assert (gi == si).all() and (gd == sd).all()


# ---------------------------------------------------------------------------------------------------------- C2 shape
@pytest.fixture(scope="module")
def flat_big_lists():
    """IVF-Flat at the C2 shape's list length: 100k x 128 fp32, 32 lists of ~3.1k rows (the bench's hold 2.4k), 384 queries."""
    import torch
    from cuvs_amd.neighbors import ivf_flat

    x, q = _mixture(100_000, 128, 384, seed=4096)
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=32, kmeans_n_iters=10, kmeans_trainset_fraction=0.3), torch.from_numpy(x).cuda())
    return x, q, index, ivf_flat.export_for_oracle(index, np.float32)


def _flat_search(index, q, k, n_probes):
    import torch
    from cuvs_amd.neighbors import ivf_flat

    d, i = ivf_flat.search(ivf_flat.SearchParams(n_probes=n_probes), index, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy()


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean"])
def test_c2_shape_matrix_core_tail_phase(flat_big_lists, metric, monkeypatch):
    """Tail phase of IVF-Flat on the matrix cores (fp16 residual copy as MFMA operands, survivors re-scored with the scan
    kernel's fp32 fma chain): the oracle's ids / distances, the scan kernel's (CUVS_AMD_FLAT_SCAN3=0), and the re-run on
    the scan kernel when a buffer runs over (CUVS_AMD_PQ3_SURV_CAP)."""
    import torch
    from cuvs_amd.neighbors import ivf_flat

    x, q, index, ex = flat_big_lists
    if metric != "sqeuclidean":
        index = ivf_flat.build(ivf_flat.IndexParams(n_lists=32, kmeans_n_iters=10, kmeans_trainset_fraction=0.3, metric=metric),
                               torch.from_numpy(x).cuda())
        ex = ivf_flat.export_for_oracle(index, np.float32)
    k, n_probes = 10, 12
    gd, gi = _flat_search(index, q, k, n_probes)
    od, oi = oracle.ivf_flat_search(ex, q, k, n_probes, metric=metric)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ3_SURV_CAP", "2000")  # buffers run over: the tail phase is re-run on the scan kernel
    hd, hi = _flat_search(index, q, k, n_probes)
    assert (gi == hi).all() and (gd == hd).all()
    monkeypatch.delenv("CUVS_AMD_PQ3_SURV_CAP")
    monkeypatch.setenv("CUVS_AMD_FLAT_SCAN3", "0")
    sd, si = _flat_search(index, q, k, n_probes)
    assert (gi == si).all() and (gd == sd).all()


@pytest.mark.parametrize("k", [64, 100])
def test_c2_shape_far_queries_and_larger_k(flat_big_lists, k):
    """Uniform queries far from every mode (loose head bounds: pools run over into the overflow list), k = 64 and 100."""
    x, _, index, ex = flat_big_lists
    rng = np.random.default_rng(6)
    q = (rng.random((300, 128), dtype=np.float32) * 4.0 - 2.0).astype(np.float32)
    gd, gi = _flat_search(index, q, k, 12)
    od, oi = oracle.ivf_flat_search(ex, q, k, 12)
    assert (gi == oi).all() and (gd == od).all()


def test_c2_shape_fp16_rows(monkeypatch):
    """fp16 rows and queries through the matrix-core tail phase (rows widened exactly, the same fp32 fma chain): the
    oracle's and the scan kernel's ids / distances, at 128 and at 96 dimensions."""
    import torch
    from cuvs_amd.neighbors import ivf_flat

    for dim in (128, 96):
        monkeypatch.delenv("CUVS_AMD_FLAT_SCAN3", raising=False)
        x, q = _mixture(60_000, dim, 320, seed=1600 + dim)
        x, q = x.astype(np.float16), q.astype(np.float16)
        index = ivf_flat.build(ivf_flat.IndexParams(n_lists=24, kmeans_n_iters=10, kmeans_trainset_fraction=0.3),
                               torch.from_numpy(x).cuda())
        ex = ivf_flat.export_for_oracle(index, np.float16)
        for k in (10, 64):
            gd, gi = _flat_search(index, q, k, 12)
            od, oi = oracle.ivf_flat_search(ex, q, k, 12)
            assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
            assert (gd == od).all()
        monkeypatch.setenv("CUVS_AMD_FLAT_SCAN3", "0")
        sd, si = _flat_search(index, q, 64, 12)
        assert (gi == si).all() and (gd == sd).all()


@pytest.mark.parametrize("dtype,scale", [(np.int8, 1 / 128), (np.uint8, 1 / 256)])
def test_c2_shape_int8_rows(dtype, scale, monkeypatch):
    """int8 / uint8 rows through the matrix-core tail phase (round 5; reference: interleaved_scan_impl.cuh:71-206 with the dp4a
    metrics of metric_impl.cuh:12-49). The fp16 copy holds the residuals against the list centre, the survivors' distances are
    integers below 2^24 in an fp32 chain = the scan kernel's integer sums: ids and distances equal the integer oracle and the
    scan kernel (CUVS_AMD_FLAT_SCAN3=0), with the many exact ties integer data has, at 128 and 96 dimensions."""
    import torch
    from cuvs_amd.neighbors import ivf_flat

    for dim in (128, 96):
        monkeypatch.delenv("CUVS_AMD_FLAT_SCAN3", raising=False)
        x, q = _mixture(60_000, dim, 320, seed=1700 + dim)
        amp = 20.0 if dtype == np.int8 else 18.0
        off = 0.0 if dtype == np.int8 else 128.0
        lo, hi = (-128, 127) if dtype == np.int8 else (0, 255)
        x = np.clip(np.rint(x * amp + off), lo, hi).astype(dtype)
        q = np.clip(np.rint(q * amp + off), lo, hi).astype(dtype)
        x[100:140] = x[50_000:50_040]  # exact duplicates: ties at equal distance are kept by row order
        index = ivf_flat.build(ivf_flat.IndexParams(n_lists=24, kmeans_n_iters=10, kmeans_trainset_fraction=0.3),
                               torch.from_numpy(x).cuda())
        ex = ivf_flat.export_for_oracle(index, dtype)
        for k in (10, 64):
            gd, gi = _flat_search(index, q, k, 12)
            od, oi = oracle.ivf_flat_search(ex, q, k, 12, coarse_scale=scale)
            assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
            assert (gd == od).all()
        monkeypatch.setenv("CUVS_AMD_FLAT_SCAN3", "0")
        sd, si = _flat_search(index, q, 64, 12)
        assert (gi == si).all() and (gd == sd).all()


@pytest.mark.parametrize("metric", ["inner_product", "cosine"])
@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8])
def test_c2_shape_inner_product(dtype, metric, monkeypatch):
    """Inner product through the matrix-core tail phase (round 5): a head phase on the scan kernel leaves the k-th full score of the
    nearest probe, the filter screens -(q.c + q.(x - c)) against it (filter_threshold_ip), survivors are re-scored with the scan
    kernel's chain acc = fma(x, q, acc). Ids and distances equal the oracle's and the scan kernel's one-phase search
    (CUVS_AMD_FLAT_SCAN3=0), k 10 and 64; rows of very different norms (the large ones win most dot products). Cosine: the same
    filter on unit-length queries and rows (the fp16 copy holds x / |x| - c), the exact chain forms dot / (|q| |x|) as the scan kernel
    does."""
    import torch
    from cuvs_amd.neighbors import ivf_flat

    monkeypatch.delenv("CUVS_AMD_FLAT_SCAN3", raising=False)
    x, q = _mixture(60_000, 128, 320, seed=1800)
    rng = np.random.default_rng(18)
    x *= rng.uniform(0.5, 1.5, size=(len(x), 1)).astype(np.float32)
    scale = 1.0
    if dtype == np.int8:
        x, q = np.clip(np.rint(x * 20.0), -128, 127), np.clip(np.rint(q * 20.0), -128, 127)
        scale = 1 / 128
    x, q = x.astype(dtype), q.astype(dtype)
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=24, kmeans_n_iters=10, kmeans_trainset_fraction=0.3, metric=metric),
                           torch.from_numpy(x).cuda())
    ex = ivf_flat.export_for_oracle(index, dtype)
    for k in (10, 64):
        gd, gi = _flat_search(index, q, k, 12)
        od, oi = oracle.ivf_flat_search(ex, q, k, 12, metric=metric, coarse_scale=scale)
        assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
        assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_FLAT_SCAN3", "0")
    sd, si = _flat_search(index, q, 64, 12)
    assert (gi == si).all() and (gd == sd).all()


# ---------------------------------------------------------------------------------------------------------- C4 shape
@pytest.fixture(scope="module")
def cagra_768():
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(768)
    z = rng.standard_normal((20_000, 24)).astype(np.float32)
    basis = rng.standard_normal((24, 768)).astype(np.float32) / math.sqrt(24)
    x = (z @ basis + 0.02 * rng.standard_normal((20_000, 768)).astype(np.float32)).astype(np.float16)
    zq = rng.standard_normal((256, 24)).astype(np.float32)
    q = (zq @ basis + 0.02 * rng.standard_normal((256, 768)).astype(np.float32)).astype(np.float16)
    index = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), torch.from_numpy(x).cuda())
    return x, q, index


def test_c4_shape_single_cta_walk_is_the_oracle_walk(cagra_768):
    import torch
    from cuvs_amd.neighbors import cagra

    x, q, index = cagra_768
    assert index.graph_degree == 64
    graph = index.graph.cpu().numpy().view(np.uint32)
    d, i = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), index, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    gi = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    od, oi = oracle.cagra_search(x, graph, q, 10, itopk_size=64)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (d.cpu().numpy() == od).all()


@pytest.mark.parametrize("algo", ["multi_cta", "auto"])
def test_c4_shape_recall(cagra_768, algo):
    import torch
    from cuvs_amd.neighbors import cagra

    x, q, index = cagra_768
    _, i = cagra.search(cagra.SearchParams(itopk_size=64, algo=algo), index, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    _, ti = oracle.exact_knn(q.astype(np.float32), x.astype(np.float32), 10)
    assert oracle.recall(i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, ti) >= 0.95


# ---------------------------------------------------------------------------------------------------------- C5 shape
@pytest.mark.parametrize("lut,acc", [("f16", "f32"), ("f16", "f16"), ("f32", "f32")])
def test_c5_shape_int8_rows(lut, acc):
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(96)
    n, d, nq, k, n_lists, n_probes = 60_000, 96, 400, 10, 32, 16
    x = rng.integers(1, 20, size=(n, d)).astype(np.int8)   # the reference's int8 generator
    q = rng.integers(1, 20, size=(nq, d)).astype(np.int8)
    index = _pq_build(x, n_lists=n_lists, pq_dim=64, pq_bits=8, kmeans_n_iters=10)
    assert index.pq_dim == 64 and index.pq_len == 2   # rot_dim 128: the bench kernels
    gd, gi = _pq_search(index, q, k, n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    ex = ivf_pq.export_for_oracle(index)
    od, oi = oracle.ivf_pq_search(ex, q.astype(np.float32) / 128.0, k, n_probes, scale=128.0, lut=lut, acc=acc)  # mapping<float>(int8)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    _, ti = oracle.exact_knn(q.astype(np.float32), x.astype(np.float32), k)
    p = n_probes / n_lists
    compression = d * 8 * 1 / (64 * 8)
    assert oracle.recall(gi, ti) >= min(math.erfc(0.05 * compression / max(p, 0.5)), p) * 0.9  # ann_ivf_pq.cuh:639-655
