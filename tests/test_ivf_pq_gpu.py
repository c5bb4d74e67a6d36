"""GPU: cuvsIvfPq* through the C ABI.

* search parity: the GPU search of an index vs the CPU oracle searching the SAME index bytes (exported
  through the reference's own getters / UnpackContiguousListData) — identical ids and distances;
* build parity: PQ codes vs the oracle's encoder on the same centres/codebooks;
* recall vs exact kNN with the reference's thresholds (cpp/tests/neighbors/ann_ivf_pq.cuh:639-655,
  defaults :26-43: n=4096, q=1024, d=64, k=32, n_lists=32) and its structural invariants (:662-678,
  python test_ivf_pq.py:119-122).
"""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _gen(n, d, q, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    if dtype in (np.int8, np.uint8):
        x = rng.integers(1, 20, size=(n, d)).astype(dtype)  # reference: uniformInt[1,20)
        qq = rng.integers(1, 20, size=(q, d)).astype(dtype)
    else:
        x = (rng.random((n, d), dtype=np.float32) * 1.9 + 0.1).astype(dtype)
        qq = (rng.random((q, d), dtype=np.float32) * 1.9 + 0.1).astype(dtype)
    return x, qq


def _build(x, **kw):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    params = ivf_pq.IndexParams(**kw)
    return ivf_pq.build(params, torch.from_numpy(x).cuda())


def _search(index, q, k, **kw):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    sp = ivf_pq.SearchParams(**kw)
    d, i = ivf_pq.search(sp, index, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy()


def _min_recall(n_probes, n_lists, dim, pq_dim, pq_bits, elem_bytes=4):
    p = n_probes / n_lists
    compression = dim * 8 * elem_bytes / (pq_dim * pq_bits)
    return min(math.erfc(0.05 * compression / max(p, 0.5)), p)


@pytest.mark.parametrize("lut,acc,k", [("f32", "f32", 10), ("f16", "f16", 10), ("f16", "f32", 100)])
def test_lut_larger_than_lds(lut, acc, k):
    """768-d data with the reference's default pq_dim (dim / 2 = 384, 8 bits): a 384 x 256 LUT does not fit 160 KiB of
    LDS even for one query (ivf_pq_compute_similarity_impl.cuh:449-465 falls back to a LUT outside shared memory). The
    scan keeps a per-workgroup LUT in global memory: ids and distances identical to the oracle."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(3000, 768, 60, seed=77)
    index = _build(x, n_lists=8, pq_dim=0, pq_bits=8, kmeans_n_iters=10)
    assert index.pq_dim == 384
    gd, gi = _search(index, q, k, n_probes=4, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    ex = ivf_pq.export_for_oracle(index)
    od, oi = oracle.ivf_pq_search(ex, q, k, 4, metric="sqeuclidean", lut=lut, acc=acc)
    assert (gi == oi).all() and (gd == od).all()


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "inner_product"])
@pytest.mark.parametrize("n,d,n_lists,pq_dim,pq_bits,k,n_probes", [
    (4096, 64, 32, 32, 8, 32, 8),      # reference defaults
    (3000, 30, 20, 10, 8, 10, 20),     # pq_len 3, rot_dim == dim
    (5000, 33, 16, 8, 8, 7, 5),        # rot_dim 40 != dim 33 -> random rotation
    (4096, 64, 32, 64, 5, 16, 6),      # 5-bit codes (generic bit path), pq_len 1
    (2048, 128, 8, 64, 8, 100, 8),     # k > 64 (2 ranks per lane, workgroup merge of the wave lists)
    (3000, 64, 8, 16, 8, 200, 8),      # k > 128 (4 ranks per lane)
])
def test_search_parity_with_oracle_on_same_index(metric, n, d, n_lists, pq_dim, pq_bits, k, n_probes):
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(n, d, 200, seed=n + d)
    index = _build(x, n_lists=n_lists, metric=metric, pq_dim=pq_dim, pq_bits=pq_bits, kmeans_n_iters=10)
    gd, gi = _search(index, q, k, n_probes=n_probes)
    ex = ivf_pq.export_for_oracle(index)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.4f}"
    assert (gd == od).all(), f"max |d| diff {np.abs(gd - od).max()}"


def test_structure_and_codes_match_oracle_encoder():
    from cuvs_amd.neighbors import ivf_pq

    n, d = 6000, 48
    x, _ = _gen(n, d, 1, seed=7)
    index = _build(x, n_lists=24, pq_dim=16, pq_bits=8, kmeans_n_iters=10)
    ex = ivf_pq.export_for_oracle(index)
    sizes = ex["list_sizes"]
    assert sizes.sum() == n == len(index)
    all_ids = np.concatenate(ex["ids"])
    assert np.array_equal(np.sort(all_ids), np.arange(n))  # every source row exactly once
    # in-list order is ascending source id (deterministic layout)
    for ids in ex["ids"]:
        assert (np.diff(ids) > 0).all()
    # every row sits in the list of its nearest centre (L2 argmin, ties -> smaller list)
    cd = oracle.pairwise(x, ex["centers"])
    lab = np.argmin(cd, axis=1)
    got_lab = np.empty(n, np.int64)
    for L, ids in enumerate(ex["ids"]):
        got_lab[ids] = L
    assert (got_lab == lab).mean() > 0.999  # oracle.pairwise adds |x|^2 (different rounding than the argmin kernel)
    # codes: rotate with the canonical dot, subtract the rotated centre, encode
    rx = oracle.pairwise(x, ex["rotation"], metric="inner_product")
    resid = rx - ex["centers_rot"][got_lab]
    want = oracle.pq_encode(resid, ex["pq_centers"], 8)
    got = np.empty_like(want)
    for L, ids in enumerate(ex["ids"]):
        got[ids] = ex["codes"][L]
    assert (got == want).all(), f"code mismatch rate {(got != want).mean():.5f}"


@pytest.mark.parametrize("lut,acc", [(np.float32, np.float32), (np.float16, np.float32), (np.float16, np.float16),
                                     (np.uint8, np.float16)])
def test_recall_reference_defaults(lut, acc):
    # ann_ivf_pq.cuh defaults; threshold formula :639-646
    n, d, nq, k = 4096, 64, 1024, 32
    x, q = _gen(n, d, nq, seed=1234)
    index = _build(x, n_lists=32, pq_dim=32, pq_bits=8)
    _, gi = _search(index, q, k, n_probes=20, lut_dtype=lut, internal_distance_dtype=acc)
    _, ti = oracle.exact_knn(q, x, k)
    r = oracle.recall(gi, ti)
    assert r >= _min_recall(20, 32, d, 32, 8), r
    assert (gi >= 0).all() and (gi < n).all()  # no out-of-bounds / invalid records (:662-678)


def test_fp16_lut_close_to_fp32():
    x, q = _gen(8000, 64, 300, seed=5)
    index = _build(x, n_lists=32, pq_dim=32)
    d32, i32 = _search(index, q, 10, n_probes=16)
    d16, i16 = _search(index, q, 10, n_probes=16, lut_dtype=np.float16, internal_distance_dtype=np.float16)
    assert oracle.recall(i16, i32) > 0.9
    np.testing.assert_allclose(d16, d32, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [np.float16, np.int8, np.uint8])
def test_other_dtypes_recall(dtype):
    n, d, k = 5000, 32, 10
    x, q = _gen(n, d, 200, seed=3, dtype=dtype)
    index = _build(x, n_lists=16, pq_dim=32, pq_bits=8)
    gd, gi = _search(index, q, k, n_probes=16)  # all lists probed: only PQ error remains
    td, ti = oracle.exact_knn(q.astype(np.float32), x.astype(np.float32), k)
    assert oracle.recall(gi, ti) > 0.8
    # distances are reported in the units of the input (scaling undone, ivf_pq_search.cuh:1043)
    np.testing.assert_allclose(gd[:, 0], td[:, 0], rtol=0.2, atol=0.2 * td[:, 0].mean())


def test_extend_and_probe_all():
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(6000, 32, 100, seed=11)
    index = _build(x[:4000], n_lists=16, pq_dim=32)
    ids = torch.arange(4000, 6000, dtype=torch.int64, device="cuda")
    ivf_pq.extend(index, torch.from_numpy(x[4000:]).cuda(), ids)
    assert len(index) == 6000
    gd, gi = _search(index, q, 10, n_probes=1000)  # n_probes is clamped to n_lists
    _, ti = oracle.exact_knn(q, x, 10)
    assert oracle.recall(gi, ti) > 0.85
    ex = ivf_pq.export_for_oracle(index)
    assert np.array_equal(np.sort(np.concatenate(ex["ids"])), np.arange(6000))
    od, oi = oracle.ivf_pq_search(ex, q, 10, 16)
    assert (gi == oi).all() and (gd == od).all()


def test_small_lists_pad_with_out_of_bounds_record():
    # fewer candidates than k: the tail is kOutOfBoundsRecord / FLT_MAX (ivf_common.cuh:25-31)
    x, q = _gen(600, 16, 20, seed=2)
    index = _build(x, n_lists=64, pq_dim=8, kmeans_n_iters=5)
    gd, gi = _search(index, q, 50, n_probes=1)
    assert ((gi == np.iinfo(np.int64).max) == (gd == np.finfo(np.float32).max)).all()
    assert (gi == np.iinfo(np.int64).max).any()


def test_errors():
    import torch
    from cuvs_amd._lib import CuvsError
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(1000, 16, 5, seed=1)
    with pytest.raises(CuvsError):
        _build(x, n_lists=2000)  # rows < n_lists
    with pytest.raises(CuvsError):
        _build(x, n_lists=8, pq_bits=9)
    index = _build(x, n_lists=8, pq_dim=8)
    with pytest.raises(CuvsError):
        _search(index, q[:, :8].copy(), 3)
    with pytest.raises(CuvsError):
        ivf_pq.extend(index, torch.from_numpy(x).cuda(), None)  # ids required for a non-empty index


@pytest.mark.parametrize("n,d,n_lists,pq_dim,pq_bits,k,n_probes", [(4096, 64, 32, 32, 8, 16, 8), (3000, 30, 20, 10, 8, 10, 20),
                                                                   (4096, 64, 32, 64, 5, 16, 6)])
def test_cosine_parity_and_recall(n, d, n_lists, pq_dim, pq_bits, k, n_probes):
    """CosineExpanded = inner product of unit-length rows / centres / queries, reported as 1 - cos
    (ivf_pq_build.cuh:159-166,1336-1348; ivf_pq_search.cuh:101-107,1020-1025)."""
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.5, 3.0, (n, 1)).astype(np.float32)
    q = rng.standard_normal((150, d)).astype(np.float32)
    index = _build(x, n_lists=n_lists, metric="cosine", pq_dim=pq_dim, pq_bits=pq_bits, kmeans_n_iters=10)
    gd, gi = _search(index, q, k, n_probes=n_probes)
    ex = ivf_pq.export_for_oracle(index)
    np.testing.assert_allclose(np.linalg.norm(ex["centers"], axis=1), 1.0, rtol=1e-5)   # centres are unit vectors
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric="cosine")
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.4f}"
    assert (gd == od).all(), f"max |d| diff {np.abs(gd - od).max()}"
    # recall against exact cosine kNN, the reference's threshold formula
    xf, qf = x.astype(np.float64), q.astype(np.float64)
    cosd = 1.0 - (qf @ xf.T) / (np.linalg.norm(qf, axis=1)[:, None] * np.linalg.norm(xf, axis=1)[None, :])
    truth = np.argsort(cosd, axis=1, kind="stable")[:, :k]
    assert oracle.recall(gi, truth) >= _min_recall(n_probes, n_lists, d, pq_dim, pq_bits) * 0.9
    assert (gd > -1e-3).all() and (gd < 2.001).all()


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
@pytest.mark.parametrize("n,d,n_lists,pq_dim,pq_bits,k,n_probes", [(6000, 64, 8, 32, 8, 16, 4), (8000, 64, 4, 64, 8, 10, 4),
                                                                   (6000, 40, 6, 20, 5, 10, 6)])
def test_per_cluster_codebooks(metric, n, d, n_lists, pq_dim, pq_bits, k, n_probes):
    """codebook_gen::PER_CLUSTER (ivf_pq_build.cuh:410-497): one codebook per list shared by its subspaces; search
    parity with the oracle on the same index, the reference's recall threshold, pq_centers getter shape."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(n, d, 120, seed=n + pq_dim)
    index = _build(x, n_lists=n_lists, metric=metric, pq_dim=pq_dim, pq_bits=pq_bits, kmeans_n_iters=10,
                   codebook_kind="cluster")
    assert tuple(index.pq_centers.shape) == (n_lists, d // pq_dim if d % pq_dim == 0 else -(-d // pq_dim), 1 << pq_bits)
    gd, gi = _search(index, q, k, n_probes=n_probes)
    ex = ivf_pq.export_for_oracle(index, per_cluster=True)
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.4f}"
    assert (gd == od).all(), f"max |d| diff {np.abs(gd - od).max()}"
    td, ti = oracle.exact_knn(q, x, k, metric=metric)
    assert oracle.recall(gi, ti) >= _min_recall(n_probes, n_lists, d, pq_dim, pq_bits) * 0.9
    # every source id is stored exactly once
    ids = np.concatenate(ex["ids"])
    assert len(ids) == n and len(np.unique(ids)) == n


@pytest.mark.parametrize("lut", [np.float32, np.float16])
def test_two_phase_schedule_and_early_stop(lut, monkeypatch):
    """Batches of 256+ queries scan every query's nearest probe first (head launch) and prune the tail launch with
    the early stop (compute_score_impl.cuh:70-71). Neither may change a result: fp32 LUT = the oracle bit for bit,
    fp16 LUT = the same kernel with both switched off."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(30000, 128, 400, seed=77)
    index = _build(x, n_lists=64, pq_dim=64, pq_bits=8, kmeans_n_iters=10)   # pq_dim 64 / 8 bits: the FAST4 kernel
    gd, gi = _search(index, q, 20, n_probes=16, lut_dtype=lut, internal_distance_dtype=lut)
    if lut == np.float32:
        od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, 20, 16)
        assert (gi == oi).all() and (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ_HEAD_PROBES", "0")
    monkeypatch.setenv("CUVS_AMD_SCAN_DEBUG", "8")
    pd, pi = _search(index, q, 20, n_probes=16, lut_dtype=lut, internal_distance_dtype=lut)
    assert (gi == pi).all() and (gd == pd).all()


@pytest.mark.parametrize("pq_bits,codebook_kind", [(8, "subspace"), (5, "subspace"), (8, "cluster")])
def test_transform_matches_the_index_contents(pq_bits, codebook_kind):
    """cuvsIvfPqTransform (labels + contiguous codes) must reproduce, row by row, what build() stored in the lists."""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, _ = _gen(5000, 48, 1, seed=5)
    index = _build(x, n_lists=12, pq_dim=16, pq_bits=pq_bits, kmeans_n_iters=10, codebook_kind=codebook_kind)
    labels, codes = ivf_pq.transform(index, torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    labels, codes = labels.cpu().numpy().view(np.uint32), codes.cpu().numpy()
    ex = ivf_pq.export_for_oracle(index, per_cluster=codebook_kind == "cluster")
    seen = 0
    for L in range(12):
        ids = ex["ids"][L]
        assert (labels[ids] == L).all()
        assert (codes[ids] == ex["codes"][L]).all()
        seen += len(ids)
    assert seen == 5000


@pytest.mark.parametrize("d,pq_dim,pq_bits", [(64, 0, 8), (64, 32, 8), (64, 16, 6), (33, 8, 8)])
def test_reconstruction_error_bound(d, pq_dim, pq_bits):
    """cpp/tests/neighbors/ann_ivf_pq.cuh:98-124,317-350: decode every row from its PQ code (centre + codebook entries,
    rotated back) and compare with the original: rms error per dimension <= 1.2 * 0.06 * 2^compression_ratio,
    compression_ratio = dim * 8 / (pq_dim * pq_bits) (:597-598)."""
    x, _ = _gen(4096, d, 1, seed=21)
    index = _build(x, n_lists=32, pq_dim=pq_dim, pq_bits=pq_bits, kmeans_trainset_fraction=1.0)
    from cuvs_amd.neighbors import ivf_pq

    e = ivf_pq.export_for_oracle(index)
    R, cr, pq = e["rotation"], e["centers_rot"], e["pq_centers"]  # [rot_dim, dim], [n_lists, rot_dim], [pq_dim, pq_len, book]
    assert np.allclose(R.T @ R, np.eye(d), atol=1e-4)  # orthonormal columns: R^T undoes the rotation
    bound = 1.2 * 0.06 * 2.0 ** (d * 8 / (e["pq_dim"] * e["pq_bits"]))
    worst = 0.0
    for L in range(len(e["codes"])):
        codes = oracle.unpack_codes(e["codes"][L], e["pq_dim"], e["pq_bits"]) if hasattr(oracle, "unpack_codes") else None
        if codes is None:
            bits = np.unpackbits(e["codes"][L], axis=1, bitorder="little")[:, : e["pq_dim"] * e["pq_bits"]]
            w = (1 << np.arange(e["pq_bits"])).astype(np.int64)
            codes = (bits.reshape(len(bits), e["pq_dim"], e["pq_bits"]) * w).sum(2)
        if len(codes) == 0:
            continue
        rec_rot = cr[L][None, :].repeat(len(codes), 0).copy()
        for s in range(e["pq_dim"]):
            rec_rot[:, s * e["pq_len"]:(s + 1) * e["pq_len"]] += pq[s][:, codes[:, s]].T
        rec = rec_rot @ R
        err = np.sqrt(((rec.astype(np.float64) - x[e["ids"][L]]) ** 2).mean(1))
        worst = max(worst, float(err.max()))
    assert worst <= bound, (worst, bound)


def test_pack_unpack_round_trip_through_extend():
    """ann_ivf_pq.cuh:391-420 (check_packing): dump the codes of a list, write them into a fresh index through
    BuildPrecomputed + the packed-code view, and find identical contents and search results."""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, qq = _gen(3000, 32, 64, seed=22)
    index = _build(x, n_lists=16, pq_dim=16, pq_bits=8)
    e = ivf_pq.export_for_oracle(index)
    d0, i0 = _search(index, qq, 10, n_probes=16)
    # every source id appears exactly once over the lists, none is the invalid record (test_ivf_pq.py:119-122)
    all_ids = np.concatenate(e["ids"])
    assert len(all_ids) == 3000 and (np.sort(all_ids) == np.arange(3000)).all()
    # transform() of the original rows reproduces the list contents byte for byte: pack(unpack(list)) == list
    labels, codes = ivf_pq.transform(index, torch.from_numpy(x).cuda())
    labels, codes = labels.cpu().numpy(), codes.cpu().numpy()
    for L in range(16):
        rows = e["ids"][L]
        assert (labels[rows] == L).all()
        assert (codes[rows] == e["codes"][L]).all()
    # and the oracle searching the dumped bytes returns what the GPU index returns
    od, oi = oracle.ivf_pq_search(e, qq, 10, 16)
    assert (oi == i0).all() and (od == d0).all()


_LUTS = {"f32": np.float32, "f16": np.float16, "fp8": np.uint8}


@pytest.mark.parametrize("lut,acc", [("f16", "f32"), ("f16", "f16"), ("fp8", "f32"), ("fp8", "f16")])
@pytest.mark.parametrize("metric,n,d,n_lists,pq_dim,pq_bits,k,n_probes,nq", [
    ("sqeuclidean", 4096, 64, 32, 32, 8, 32, 8, 64),      # reference defaults
    ("sqeuclidean", 30000, 128, 64, 64, 8, 20, 16, 400),  # FAST4 kernels, two-phase schedule, pq_scan2 (fp16 LUT)
    ("inner_product", 4096, 64, 32, 32, 8, 16, 8, 64),    # signed fp8 (ivf_pq_search.cuh:711-728)
    ("sqeuclidean", 4096, 64, 32, 64, 5, 16, 6, 64),      # 5-bit codes
])
def test_reduced_precision_lut_parity(lut, acc, metric, n, d, n_lists, pq_dim, pq_bits, k, n_probes, nq):
    """lut_dtype / internal_distance_dtype (ivf_pq.hpp:167-205): fp16 and the reference's fp_8bit<5, signed> LUT
    entries, fp16 / fp32 scores - ids and distances bit-identical to the oracle's restatement of that arithmetic."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(n, d, nq, seed=31)
    index = _build(x, n_lists=n_lists, pq_dim=pq_dim, pq_bits=pq_bits, metric=metric, kmeans_n_iters=10)
    gd, gi = _search(index, q, k, n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()


def test_fp8_lut_recall_thresholds():
    """ann_ivf_pq.cuh:1032-1035 (enum_variety: defaults + lut_dtype = CUDA_R_8U -> min_recall 0.84), :1081-1086 (inner
    product: x 0.88). n_probes default 20 of 32 lists."""
    n, d, nq, k = 4096, 64, 1024, 32
    x, q = _gen(n, d, nq, seed=1234)
    for metric, thr in (("sqeuclidean", 0.84), ("inner_product", 0.84 * 0.88)):
        index = _build(x, n_lists=32, kmeans_trainset_fraction=1.0, metric=metric)
        _, gi = _search(index, q, k, n_probes=20, lut_dtype=np.uint8)
        _, ti = oracle.brute_force_knn(q, x, k, metric=metric)
        r = oracle.recall(gi, ti)
        assert r >= thr, (metric, r)


@pytest.mark.parametrize("lut,acc", [("f32", "f32"), ("f16", "f16")])
def test_large_k_non_fused_path(lut, acc):
    """k > 256 (ivf_pq_compute_similarity_impl.cuh:39-45 is_local_topk_feasible -> ivf_pq_search.cuh:620 select_k over
    every probed row): identical to the oracle, for the generic and the pq_dim 64 / 8-bit kernels."""
    from cuvs_amd.neighbors import ivf_pq

    for (n, d, n_lists, pq_dim) in ((6000, 32, 16, 16), (8000, 128, 16, 64)):
        x, q = _gen(n, d, 40, seed=6)
        index = _build(x, n_lists=n_lists, pq_dim=pq_dim, pq_bits=8, kmeans_n_iters=10)
        ex = ivf_pq.export_for_oracle(index)
        for k, n_probes in ((512, 6), (1000, 2)):
            gd, gi = _search(index, q, k, n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
            od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, lut=lut, acc=acc)
            assert (gi == oi).all() and (gd == od).all(), (n, k, n_probes)


_COARSE = {"f16": np.float16, "i8": np.int8}


@pytest.mark.parametrize("coarse", ["f16", "i8"])
@pytest.mark.parametrize("metric,n,d,n_lists,pq_dim", [("sqeuclidean", 4096, 64, 32, 32), ("inner_product", 4096, 64, 32, 32),
                                                       ("sqeuclidean", 6000, 30, 24, 10), ("cosine", 4096, 64, 32, 32)])
def test_coarse_search_dtype_parity(coarse, metric, n, d, n_lists, pq_dim):
    """search_params.coarse_search_dtype (ivf_pq_search.cuh:171-340,:995-1017): queries, centres and the rotation matrix
    rounded to half / int8 for the coarse GEMM and the rotation GEMM - identical to the oracle's restatement."""
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(41)
    x = rng.standard_normal((n, d)).astype(np.float32) * 0.4   # inside the int8 range after the x128 mapping
    q = rng.standard_normal((64, d)).astype(np.float32) * 0.4
    index = _build(x, n_lists=n_lists, pq_dim=pq_dim, metric=metric, kmeans_n_iters=10)
    gd, gi = _search(index, q, 16, n_probes=8, coarse_search_dtype=_COARSE[coarse])
    od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, 16, 8, metric=metric, coarse=coarse)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()


@pytest.mark.parametrize("coarse", ["f16", "i8"])
@pytest.mark.parametrize("n,d,n_lists,nq", [(20000, 128, 200, 333), (9000, 300, 50, 130), (12000, 96, 1000, 257)])
def test_coarse_search_dtype_matrix_core_kernel(coarse, n, d, n_lists, nq, monkeypatch):
    """coarse_lowp_kernel (fp16 / int8 matrix cores, output arithmetic in the epilogue): ragged query blocks and centre
    tiles, K steps in registers (d 96 / 128) and re-read (d 300) - identical to the oracle and to the same search with the
    rounded operands on the fp32 matrix cores (CUVS_AMD_COARSE_LOWP=0)."""
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32) * 0.3
    q = rng.standard_normal((nq, d)).astype(np.float32) * 0.3
    index = _build(x, n_lists=n_lists, pq_dim=32, kmeans_n_iters=6)
    gd, gi = _search(index, q, 10, n_probes=9, coarse_search_dtype=_COARSE[coarse])
    od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, 10, 9, coarse=coarse)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_COARSE_LOWP", "0")
    fd, fi = _search(index, q, 10, n_probes=9, coarse_search_dtype=_COARSE[coarse])
    assert (fi == gi).all() and (fd == gd).all()


def test_coarse_search_dtype_recall_thresholds():
    """ann_ivf_pq.cuh:1036-1046: defaults + coarse_search_dtype = CUDA_R_16F -> min_recall 0.86; CUDA_R_8I -> 0.1
    ("experimental ... no guarantee of any recall if the data is not normalized")."""
    n, d, nq, k = 4096, 64, 1024, 32
    x, q = _gen(n, d, nq, seed=1234)
    index = _build(x, n_lists=32, kmeans_trainset_fraction=1.0)
    _, ti = oracle.brute_force_knn(q, x, k)
    for dt, thr in ((np.float16, 0.86), (np.int8, 0.1)):
        _, gi = _search(index, q, k, n_probes=20, coarse_search_dtype=dt)
        assert oracle.recall(gi, ti) >= thr, (dt, oracle.recall(gi, ti))


def _bitset(keep):
    words = np.zeros((len(keep) + 31) // 32, np.uint32)
    idx = np.nonzero(keep)[0]
    np.bitwise_or.at(words, idx >> 5, (np.uint32(1) << (idx & 31).astype(np.uint32)))
    return words


@pytest.mark.parametrize("lut,acc,k,nq,n,pq_dim,n_lists", [
    ("f32", "f32", 10, 64, 6000, 16, 16),     # generic kernel, small batch (single phase)
    ("f16", "f32", 20, 400, 30000, 64, 64),   # FAST4 head phase + pq_scan2 tail phase
    ("f16", "f16", 100, 300, 20000, 64, 32),  # k > 64: workgroup merge of the wave lists
    ("f32", "f32", 300, 40, 8000, 64, 16),    # k > 256: non-fused path (every score written, select_k)
])
def test_bitset_prefilter_parity(lut, acc, k, nq, n, pq_dim, n_lists):
    """Pre-filtered search (cpp/include/cuvs/neighbors/ivf_pq.hpp:1818-1828 with a bitset_filter; applied per scanned row,
    compute_distances_impl.cuh:78-80, ivf_pq_search.cuh:1111-1134): ids and distances identical to the oracle with the same
    bitset, and no masked id ever comes back."""
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import ivf_pq

    x, q = _gen(n, 128 if pq_dim == 64 else 32, nq, seed=61)
    rng = np.random.default_rng(8)
    keep = rng.random(n) < 0.6
    words = _bitset(keep)
    index = _build(x, n_lists=n_lists, pq_dim=pq_dim, pq_bits=8, kmeans_n_iters=10)
    sp = ivf_pq.SearchParams(n_probes=8, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    tw = torch.from_numpy(words.view(np.int32)).cuda()
    d, i = ivf_pq.search(sp, index, torch.from_numpy(q).cuda(), k, filter=(tw, BITSET))
    torch.cuda.synchronize()
    gd, gi = d.cpu().numpy(), i.cpu().numpy()
    od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, k, 8, lut=lut, acc=acc, keep_bits=words)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    valid = gi != np.iinfo(np.int64).max
    assert keep[gi[valid]].all()


def test_bitset_prefilter_recall():
    """python/cuvs/cuvs/tests/ann_utils.py:131 style: recall of the filtered search against the exact kNN of the kept rows
    (all lists probed: only the PQ error remains)."""
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import ivf_pq

    n, d, nq, k = 8000, 32, 200, 10
    x, q = _gen(n, d, nq, seed=62)
    keep = np.arange(n) % 3 != 0
    index = _build(x, n_lists=16, pq_dim=32, pq_bits=8)
    tw = torch.from_numpy(_bitset(keep).view(np.int32)).cuda()
    _, i = ivf_pq.search(ivf_pq.SearchParams(n_probes=16), index, torch.from_numpy(q).cuda(), k, filter=(tw, BITSET))
    torch.cuda.synchronize()
    gi = i.cpu().numpy()
    kept_ids = np.nonzero(keep)[0]
    _, ti = oracle.exact_knn(q, x[keep], k)
    assert oracle.recall(gi, kept_ids[ti]) > 0.7
    assert keep[gi].all()

