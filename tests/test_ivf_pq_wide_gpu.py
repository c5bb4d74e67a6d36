"""GPU: the wide matrix-core path of the IVF-PQ search (ivf_pq_wide.hip) - rot_dim beyond pq_filter4_kernel's decode
table (the reference's default at 768 dimensions: pq_dim 384 x pq_len 2, ivf_pq_index.cu:350-364; a CAGRA build's kNN-graph search:
pq_dim 64 x pq_len 12), and k too large a fraction of ONE list for its k-th score to prune (the bound then comes from the union of
several head lists). The decoded fp16 copy of the index, the bound-only head phase through the filter's emit form, the filter, the
head pairs' survivors, the re-score and the pool merge against:
  * the CPU oracle (oracle.ivf_pq_search: compute_score_impl.cuh:52-79 / ivf_pq_search.cuh:421-669 restated) - ids and distances
    bit for bit;
  * the same search on the LUT scan kernels (CUVS_AMD_PQ_WIDE=0).
Every case asserts through the filter's counters that the wide path is what ran.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

_LUTS = {"f32": np.float32, "f16": np.float16, "fp8": np.uint8}


def _mixture(n, d, q, seed, modes=48, latent=16, sigma=0.35):
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


def _filter_stats():
    from cuvs_amd._lib import lib

    st = (C.c_uint64 * 6)()
    lib().cuvsAmdIvfPqLastFilterStats6(st)
    return [int(v) for v in st]


def _wide_and_lut(index, q, k, monkeypatch, capfd, **kw):
    """the search on the wide path (counters on: proves the path), then on the LUT scan kernels"""
    monkeypatch.setenv("CUVS_AMD_SCAN_DEBUG", "1024")
    gd, gi = _pq_search(index, q, k, **kw)
    st = _filter_stats()
    capfd.readouterr()  # (the counters' stderr lines)
    monkeypatch.delenv("CUVS_AMD_SCAN_DEBUG")
    monkeypatch.setenv("CUVS_AMD_PQ_WIDE", "0")
    sd, si = _pq_search(index, q, k, **kw)
    monkeypatch.delenv("CUVS_AMD_PQ_WIDE")
    return gd, gi, sd, si, st


@pytest.mark.parametrize("d,pq_dim,lut,acc,k,metric", [
    (768, 64, "f16", "f32", 10, "sqeuclidean"),    # pq_len 12: the CAGRA build's shape
    (768, 64, "f16", "f16", 100, "sqeuclidean"),   # k = a fifth of a list: bound from several head lists
    (768, 384, "f16", "f32", 10, "sqeuclidean"),   # pq_len 2: the reference's default pq_dim at 768 dimensions (LUT beyond the LDS: global LUT fallback)
    (768, 192, "f32", "f32", 32, "sqeuclidean"),   # pq_len 4: the reference's CAGRA default (dim / 4)
    (512, 64, "fp8", "f16", 20, "sqeuclidean"),    # pq_len 8, 32 K steps (4 chunks, ring of 2)
    (384, 96, "f16", "f32", 20, "sqeuclidean"),    # pq_len 4, 24 K steps (3 chunks)
    (256, 16, "f16", "f32", 20, "sqeuclidean"),    # pq_len 16, 16 K steps (2 chunks)
    (768, 384, "f16", "f32", 10, "inner_product"),  # score -(q.c + q.d): the query as the operand, no row terms, thresholds by filter_threshold_ip
    (768, 64, "f16", "f16", 40, "cosine"),          # pq_len 12, two head lists
    (512, 128, "fp8", "f32", 20, "inner_product"),  # signed fp8 LUT entries
    (384, 96, "f32", "f32", 20, "cosine"),
])
def test_wide_path_equals_oracle_and_lut_scan(d, pq_dim, lut, acc, k, metric, monkeypatch, capfd):
    from cuvs_amd.neighbors import ivf_pq

    # (modes as wide as they are apart: a query's neighbours sit in several of its probed lists - tail-phase survivors)
    x, q = _mixture(60_000, d, 400, seed=d + pq_dim, modes=200, sigma=1.0)
    index = _pq_build(x, n_lists=32, pq_dim=pq_dim, pq_bits=8, metric=metric, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
    ex = ivf_pq.export_for_oracle(index)
    n_probes = 12
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi, sd, si, st = _wide_and_lut(index, q, k, monkeypatch, capfd, **kw)
    assert st[0] > 0, "the wide filter did not run"
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    assert (si == oi).all() and (sd == od).all()
    # the tail phase has survivors, and most pairs are dropped by the screen (the point of the path; the margins of an unnormalised
    # inner product scale with |q| (|c| + |d|), not with the score: looser)
    assert 0 < st[1] < (0.1 if metric == "sqeuclidean" else 0.5) * st[0], st


def test_wide_path_large_k_of_short_lists(monkeypatch, capfd):
    """k = 256 of lists of ~470 rows at 128 probed of 256 lists (a CAGRA build's search in miniature): pq3_bound_useful says no, the
    wide path bounds with the union of its head lists; 256 dimensions with pq_len 8 - a shape pq_filter4_kernel serves at small k"""
    x, q = _mixture(120_000, 256, 600, seed=77, modes=64)
    index = _pq_build(x, n_lists=256, pq_dim=32, pq_bits=8, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
    kw = dict(n_probes=128, lut_dtype=np.float16, internal_distance_dtype=np.float16)
    gd, gi, sd, si, st = _wide_and_lut(index, q, 256, monkeypatch, capfd, **kw)
    assert st[0] > 0, "the wide filter did not run"
    assert (gi == si).all(), f"id mismatch rate {(gi != si).mean():.5f}"
    assert (gd == sd).all()


@pytest.mark.parametrize("d,pq_dim,lut,acc,metric", [
    (128, 64, "f16", "f32", "sqeuclidean"),    # pq_len 2: the kNN-graph search of a CAGRA build on 128-d rows
    (128, 32, "f16", "f16", "sqeuclidean"),    # pq_len 4
    (96, 48, "f16", "f32", "sqeuclidean"),     # 6 K steps (chunks of 2, ring of 3)
    (64, 32, "f32", "f32", "inner_product"),   # 4 K steps
    (128, 64, "fp8", "f16", "cosine"),
])
def test_wide_path_narrow_shapes_at_large_k(d, pq_dim, lut, acc, metric, monkeypatch, capfd):
    """64 / 96 / 128 dimensions - shapes pq_filter4_kernel serves while k is a small fraction of a list - at k = 200 of ~470-row lists:
    the bound of ONE list does not prune (pq3_bound_useful), the wide path bounds with the union of its head lists. Oracle and LUT scan."""
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(120_000, d, 500, seed=d + pq_dim, modes=64)
    index = _pq_build(x, n_lists=256, pq_dim=pq_dim, pq_bits=8, metric=metric, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
    ex = ivf_pq.export_for_oracle(index)
    k, n_probes = 200, 100
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc])
    gd, gi, sd, si, st = _wide_and_lut(index, q, k, monkeypatch, capfd, **kw)
    assert st[0] > 0, "the wide filter did not run"
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric, lut=lut, acc=acc)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    assert (si == oi).all() and (sd == od).all()


def test_wide_path_codes_of_fewer_than_8_bits(monkeypatch, capfd):
    from cuvs_amd.neighbors import ivf_pq

    x, q = _mixture(60_000, 384, 400, seed=5)
    index = _pq_build(x, n_lists=32, pq_dim=48, pq_bits=5, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
    ex = ivf_pq.export_for_oracle(index)
    kw = dict(n_probes=12, lut_dtype=np.float16, internal_distance_dtype=np.float32)
    gd, gi, sd, si, st = _wide_and_lut(index, q, 20, monkeypatch, capfd, **kw)
    assert st[0] > 0
    od, oi = oracle.ivf_pq_search(ex, q, 20, 12, lut="f16", acc="f32")
    assert (gi == oi).all() and (gd == od).all()
    assert (si == oi).all() and (sd == od).all()


def test_wide_path_hands_back_and_rebuilds(monkeypatch, capfd):
    """(1) a survivor buffer of 2000 entries: the regions run over, queries are handed back to the LUT scan pair by pair - head pairs
    included (the bound-only head phase left no candidates of theirs); (2) queries whose head lists hold fewer than k rows have no
    bound: handed back as well; (3) rows added after the decoded copy was made: it is rebuilt."""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x_all, q = _mixture(65_000, 768, 400, seed=9)
    x, x2 = x_all[:60_000], x_all[60_000:]
    index = _pq_build(x, n_lists=32, pq_dim=64, pq_bits=8, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
    kw = dict(n_probes=12, lut_dtype=np.float16, internal_distance_dtype=np.float32)
    gd, gi, sd, si, st = _wide_and_lut(index, q, 20, monkeypatch, capfd, **kw)
    assert st[0] > 0 and (gi == si).all() and (gd == sd).all()
    monkeypatch.setenv("CUVS_AMD_PQ3_SURV_CAP", "2000")
    monkeypatch.setenv("CUVS_AMD_SCAN_DEBUG", "1024")
    hd, hi = _pq_search(index, q, 20, **kw)
    st = _filter_stats()
    capfd.readouterr()
    monkeypatch.delenv("CUVS_AMD_SCAN_DEBUG")
    monkeypatch.delenv("CUVS_AMD_PQ3_SURV_CAP")
    assert st[4] > 0, "no pair was handed back"
    assert (hi == si).all() and (hd == sd).all()
    # one head list per query forced, k = 250 of lists that hold ~1900 rows on average but as few as a few hundred
    monkeypatch.setenv("CUVS_AMD_PQ_WIDE_HEADS", "1")
    g2d, g2i = _pq_search(index, q, 250, **kw)
    monkeypatch.delenv("CUVS_AMD_PQ_WIDE_HEADS")
    monkeypatch.setenv("CUVS_AMD_PQ_WIDE", "0")
    s2d, s2i = _pq_search(index, q, 250, **kw)
    monkeypatch.delenv("CUVS_AMD_PQ_WIDE")
    assert (g2i == s2i).all() and (g2d == s2d).all()
    # extend: the decoded copy follows the lists
    index = ivf_pq.extend(index, torch.from_numpy(x2).cuda(), torch.arange(60_000, 65_000, dtype=torch.int64).cuda())
    ed, ei, fd, fi, st = _wide_and_lut(index, q, 20, monkeypatch, capfd, **kw)
    assert st[0] > 0 and (ei == fi).all() and (ed == fd).all()
    assert (ei >= 60_000).any(), "the extension's rows are found"


@pytest.mark.parametrize("seed", list(range(16)))
def test_wide_path_fuzz_against_lut_scan(seed, monkeypatch, capfd):
    """random members of the wide class - dimension, pq_dim / pq_bits, row type, metric (L2 / L2Sqrt / inner product / cosine), list count (down to lists of a few
    rows and empty lists), probes, k (1 .. 256), LUT / score types, one or several internal batches - wide path == LUT scan kernels, ids
    and distances"""
    import torch
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(1000 + seed)
    d, pq_dims = [(768, (64, 192, 384, 48)), (512, (64, 128, 32)), (384, (96, 48, 192)), (256, (16, 128))][seed % 4]
    pq_dim = int(rng.choice(pq_dims))
    if d // pq_dim in (1, 2, 4, 8) and d <= 256:
        pq_dim = 16  # (a shape of pq_filter4_kernel otherwise)
    pq_bits = int(rng.choice([8, 8, 8, 5, 6]))
    n = int(rng.integers(20_000, 70_000))
    n_lists = int(rng.choice([8, 24, 100, 400]))
    nq = int(rng.integers(256, 700))
    n_probes = int(min(n_lists, rng.integers(9, 40)))
    k = int(rng.choice([1, 7, 32, 100, 256]))
    lut, acc = [("f16", "f32"), ("f16", "f16"), ("f32", "f32"), ("fp8", "f16"), ("fp8", "f32")][int(rng.integers(0, 5))]
    metric = str(rng.choice(["sqeuclidean", "euclidean", "inner_product", "cosine"]))
    x, q = _mixture(n, d, nq, seed=2000 + seed, modes=int(rng.choice([20, 200])), sigma=float(rng.choice([0.35, 1.0])))
    dtype = rng.choice(["f32", "f16", "i8"])
    if dtype == "f16":
        x, q = x.astype(np.float16), q.astype(np.float16)
    elif dtype == "i8":
        x, q = np.clip(np.rint(x * 40), -127, 127).astype(np.int8), np.clip(np.rint(q * 40), -127, 127).astype(np.int8)
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, pq_dim=pq_dim, pq_bits=pq_bits, metric=metric, kmeans_n_iters=6,
                                            kmeans_trainset_fraction=0.5), torch.from_numpy(x).cuda())
    kw = dict(n_probes=n_probes, lut_dtype=_LUTS[lut], internal_distance_dtype=_LUTS[acc],
              max_internal_batch_size=int(rng.choice([nq, 300, 32768])))
    gd, gi, sd, si, st = _wide_and_lut(index, q, k, monkeypatch, capfd, **kw)
    desc = f"d {d} pq_dim {pq_dim} bits {pq_bits} n {n} lists {n_lists} nq {nq} probes {n_probes} k {k} {lut}/{acc} {metric} {dtype} {kw['max_internal_batch_size']}"
    assert (gi == si).all(), f"{desc}: id mismatch rate {(gi != si).mean():.5f}"
    assert (gd == sd).all(), desc
    # (the wide path ran unless the rule found no head-list count: more than half of the probes)
    if st[0] == 0:
        pytest.skip(f"{desc}: the wide path's rule declined (k too large for the probed lists)")


@pytest.mark.parametrize("metric,keep_frac", [("sqeuclidean", 0.6), ("inner_product", 0.3), ("sqeuclidean", 0.02)])
def test_wide_path_with_a_bitset_prefilter(metric, keep_frac, monkeypatch, capfd):
    """pre-filtered search (ivf_pq.hpp:1818-1828 with a bitset_filter): the emit pass gives rejected rows the value -inf (the bound's k
    rows are admissible), the re-score drops rejected survivors - ids and distances equal to the oracle with the same bitset and to the
    LUT scan kernels; at 2 % kept most queries find fewer than k admissible rows in their head lists and are handed back"""
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import ivf_pq

    n, nq, k, n_probes = 60_000, 400, 20, 12
    x, q = _mixture(n, 768, nq, seed=31, modes=200, sigma=1.0)
    keep = np.random.default_rng(5).random(n) < keep_frac
    words = np.packbits(keep, bitorder="little")
    words = np.concatenate([words, np.zeros((-len(words)) % 4, dtype=np.uint8)]).view(np.uint32)
    index = _pq_build(x, n_lists=32, pq_dim=64, pq_bits=8, metric=metric, kmeans_n_iters=8, kmeans_trainset_fraction=0.3)
    tw = torch.from_numpy(words.view(np.int32)).cuda()
    sp = ivf_pq.SearchParams(n_probes=n_probes, lut_dtype=np.float16, internal_distance_dtype=np.float32)

    def run():
        d, i = ivf_pq.search(sp, index, torch.from_numpy(q).cuda(), k, filter=(tw, BITSET))
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    monkeypatch.setenv("CUVS_AMD_SCAN_DEBUG", "1024")
    gd, gi = run()
    st = _filter_stats()
    capfd.readouterr()
    monkeypatch.delenv("CUVS_AMD_SCAN_DEBUG")
    assert st[0] > 0, "the wide filter did not run"
    monkeypatch.setenv("CUVS_AMD_PQ_WIDE", "0")
    sd, si = run()
    monkeypatch.delenv("CUVS_AMD_PQ_WIDE")
    od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, k, n_probes, metric=metric, lut="f16", acc="f32", keep_bits=words)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    assert (si == oi).all() and (sd == od).all()
    found = gi[gi >= 0]
    assert keep[found[found < n]].all(), "a rejected row came back"
    if keep_frac < 0.1:
        assert st[4] > 0, "no query was handed back"
