"""GPU: *Serialize writes and *Deserialize reads the reference's on-disk container.

Both directions are checked against tests/refformat.py, an independent Python restatement of the container:
(1) files written by the library parse record by record and hold exactly the index (lists de-interleaved by the
reference's formulas); (2) files written by Python/numpy (different header spelling, FLAT and INTERLEAVED PQ layouts)
load into an index whose search results are bit-identical to the index they were exported from."""
import numpy as np
import pytest

from tests import refformat as rf

pytestmark = pytest.mark.gpu


def _data(n=3000, d=32, q=50, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        return rng.standard_normal((n, d)).astype(dtype), rng.standard_normal((q, d)).astype(dtype)
    lo, hi = (-100, 100) if np.dtype(dtype) == np.int8 else (0, 200)
    return rng.integers(lo, hi, (n, d)).astype(dtype), rng.integers(lo, hi, (q, d)).astype(dtype)


def _same(a, b):
    import torch
    torch.cuda.synchronize()
    return all(torch.equal(u, v) for u, v in zip(a, b))


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product", "cosine"])
def test_brute_force_file(tmp_path, dtype, metric):
    import torch
    from cuvs_amd.neighbors import brute_force

    x, q = _data(dtype=dtype)
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    idx = brute_force.build(tx, metric=metric)
    f = str(tmp_path / "bf.bin")
    brute_force.save(f, idx)
    p = rf.parse_brute_force(f)
    code = {"sqeuclidean": 0, "cosine": 2, "inner_product": 6}[metric]
    assert p["prefix"] == rf.PREFIX[np.dtype(dtype)] and p["version"] == 0
    assert (p["rows"], p["dim"], p["metric"]) == (3000, 32, code)
    assert p["dataset"].dtype == np.dtype(dtype) and (p["dataset"] == x).all()
    if metric != "inner_product":
        want = (x.astype(np.float64) ** 2).sum(1)
        np.testing.assert_allclose(p["norms"], np.sqrt(want) if metric == "cosine" else want, rtol=1e-5)
    # numpy-written file -> identical search results
    g = str(tmp_path / "bf_np.bin")
    rf.write_brute_force(g, x, metric=code)
    assert _same(brute_force.search(idx, tq, 10), brute_force.search(brute_force.load(g), tq, 10))
    assert _same(brute_force.search(idx, tq, 10), brute_force.search(brute_force.load(f), tq, 10))


@pytest.mark.parametrize("dtype,dim", [(np.float32, 32), (np.float32, 30), (np.float16, 24), (np.int8, 32), (np.uint8, 20)])
def test_ivf_flat_file(tmp_path, dtype, dim):
    import torch
    from cuvs_amd.neighbors import ivf_flat

    x, q = _data(d=dim, dtype=dtype)
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=24), tx)
    ex = ivf_flat.export_for_oracle(idx, dtype)
    f = str(tmp_path / "flat.bin")
    ivf_flat.save(f, idx)
    p = rf.parse_ivf_flat(f)
    assert p["prefix"] == rf.PREFIX[np.dtype(dtype)] and p["version"] == 4
    assert (p["size"], p["dim"], p["n_lists"], p["metric"]) == (3000, dim, 24, 0)
    assert (p["centers"] == ex["centers"]).all() and (p["list_sizes"] == ex["list_sizes"]).all()
    for L in range(24):
        if ex["list_sizes"][L]:
            assert (p["rows"][L] == ex["rows"][L]).all() and (p["ids"][L] == ex["ids"][L]).all()
    # python-built file from the exported lists -> identical search
    g = str(tmp_path / "flat_np.bin")
    rf.write_ivf_flat(g, ex["centers"], ex["rows"], ex["ids"], metric=0, dtype=dtype)
    sp = ivf_flat.SearchParams(n_probes=6)
    want = ivf_flat.search(sp, idx, tq, 10)
    assert _same(want, ivf_flat.search(sp, ivf_flat.load(g), tq, 10))
    idx2 = ivf_flat.load(f)
    assert _same(want, ivf_flat.search(sp, idx2, tq, 10))
    # a loaded index can be extended
    extra = torch.from_numpy(_data(n=100, d=dim, seed=5, dtype=dtype)[0]).cuda()
    ivf_flat.extend(idx2, extra, torch.arange(3000, 3100, dtype=torch.int64, device="cuda"))
    _, nn = ivf_flat.search(ivf_flat.SearchParams(n_probes=24), idx2, extra, 1)
    assert (nn.cpu().numpy()[:, 0] >= 3000).mean() > 0.9


@pytest.mark.parametrize("pq_bits,pq_dim", [(8, 16), (5, 16), (8, 32)])
@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
def test_ivf_pq_file(tmp_path, pq_bits, pq_dim, metric):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, q = _data()
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=24, pq_dim=pq_dim, pq_bits=pq_bits, metric=metric), tx)
    ex = ivf_pq.export_for_oracle(idx)
    codes = [rf.bitstream_to_codes(c, pq_dim, pq_bits) for c in ex["codes"]]
    f = str(tmp_path / "pq.bin")
    ivf_pq.save(f, idx)
    p = rf.parse_ivf_pq(f)
    code = {"sqeuclidean": 0, "inner_product": 6}[metric]
    assert p["version"] == 4 and p["codes_layout"] == 1 and p["codebook_kind"] == 0
    assert (p["size"], p["dim"], p["pq_bits"], p["pq_dim"], p["metric"], p["n_lists"]) == (3000, 32, pq_bits, pq_dim, code, 24)
    assert p["pq_centers"].shape == (pq_dim, 32 // pq_dim, 1 << pq_bits) and (p["pq_centers"] == ex["pq_centers"]).all()
    assert p["centers"].shape == (24, 40) and (p["centers"][:, :32] == ex["centers"]).all()
    np.testing.assert_allclose(p["centers"][:, 32], (ex["centers"].astype(np.float64) ** 2).sum(1), rtol=1e-5)
    assert (p["centers"][:, 33:] == 0).all()
    assert (p["centers_rot"] == ex["centers_rot"]).all() and (p["rotation"] == ex["rotation"]).all()
    for L in range(24):
        assert (p["codes"][L] == codes[L]).all() and (p["ids"][L] == ex["ids"][L]).all()
    sp = ivf_pq.SearchParams(n_probes=6)
    want = ivf_pq.search(sp, idx, tq, 10)
    for layout in (1, 0):  # INTERLEAVED and FLAT list records written by numpy
        g = str(tmp_path / ("pq_np%d.bin" % layout))
        rf.write_ivf_pq(g, 32, pq_bits, pq_dim, code, ex["pq_centers"], p["centers"], ex["centers_rot"], ex["rotation"],
                        codes, ex["ids"], layout=layout)
        back = ivf_pq.load(g)
        if layout == 1:
            assert back.codes_layout == "interleaved" and _same(want, ivf_pq.search(sp, back, tq, 10))
        else:
            # a FLAT file loads as a FLAT index: same lists (contiguous codes, ids), and the search is refused as in the
            # reference (ivf_pq_search.cuh:914-916)
            assert back.codes_layout == "flat"
            for L in range(24):
                if len(ex["ids"][L]):
                    assert torch.equal(back.list_data(L), idx.list_data(L)) and torch.equal(back.list_indices(L), idx.list_indices(L))
            with pytest.raises(Exception, match="INTERLEAVED codes layout"):
                ivf_pq.search(sp, back, tq, 10)
    idx2 = ivf_pq.load(f)
    assert _same(want, ivf_pq.search(sp, idx2, tq, 10))
    assert len(idx2) == 3000 and idx2.pq_dim == pq_dim and idx2.pq_bits == pq_bits
    # the untyped (reference-format) handle takes its dtype from the first extend
    extra = torch.from_numpy(_data(n=64, seed=7)[0]).cuda()
    ivf_pq.extend(idx2, extra, torch.arange(3000, 3064, dtype=torch.int64, device="cuda"))
    assert len(idx2) == 3064


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8])
def test_cagra_file(tmp_path, dtype):
    import torch
    from cuvs_amd.neighbors import cagra

    x, q = _data(n=2000, dtype=dtype)
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=32, graph_degree=16), tx)
    graph = idx.graph.cpu().numpy().view(np.uint32)
    f = str(tmp_path / "cagra.bin")
    cagra.save(f, idx)
    p = rf.parse_cagra(f)
    assert p["prefix"] == rf.PREFIX[np.dtype(dtype)] and p["version"] == 5
    assert (p["size"], p["dim"], p["graph_degree"], p["metric"], p["content_map"]) == (2000, 32, 16, 0, 1)
    assert (p["graph"] == graph).all()
    assert (p["tag"], p["cuda_dtype"], p["n_rows"], p["ds_dim"]) == (2, rf.CUDA_DTYPE[np.dtype(dtype)], 2000, 32)
    assert p["stride"] * np.dtype(dtype).itemsize % 16 == 0 and p["stride"] >= 32
    assert p["dataset"].dtype == np.dtype(dtype) and (p["dataset"] == x).all()
    sp = cagra.SearchParams(itopk_size=64, algo="single_cta")  # the multi-wave walk is not bit-reproducible
    want = cagra.search(sp, idx, tq, 10)
    g = str(tmp_path / "cagra_np.bin")
    rf.write_cagra(g, graph, x, metric=0, dtype=dtype)
    assert _same(want, cagra.search(sp, cagra.load(g), tq, 10))
    assert _same(want, cagra.search(sp, cagra.load(f), tq, 10))
    # without the dataset: content_map 0 and nothing after it
    h = str(tmp_path / "cagra_nodata.bin")
    cagra.save(h, idx, include_dataset=False)
    p = rf.parse_cagra(h)
    assert p["content_map"] == 0 and (p["graph"] == graph).all()


def test_cagra_file_with_source_indices(tmp_path):
    """content-map bit 1 (cagra_serialize.cuh:72-83, :314-321): an index file that carries the source id of every row. A search
    reports source_indices[row] (search_multi_cta.cuh:266-272), saving the loaded index writes the array back."""
    import torch
    from cuvs_amd.neighbors import cagra

    x, q = _data(n=2000, dtype=np.float32)
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=32, graph_degree=16), tx)
    graph = idx.graph.cpu().numpy().view(np.uint32)
    src = (np.arange(2000, dtype=np.uint32)[::-1] * 3 + 7).astype(np.uint32)  # an injective map that is not the identity
    g = str(tmp_path / "cagra_src.bin")
    rf.write_cagra(g, graph, x, metric=0, dtype=np.float32, source_indices=src)
    for algo in ("single_cta", "multi_cta"):
        sp = cagra.SearchParams(itopk_size=64, algo=algo)
        wd, wi = cagra.search(sp, idx, tq, 10)
        loaded = cagra.load(g)
        gd, gi = cagra.search(sp, loaded, tq, 10)
        torch.cuda.synchronize()
        wi_np, gi_np = wi.cpu().numpy().view(np.uint32), gi.cpu().numpy().view(np.uint32)
        if algo == "single_cta":  # (the multi-wave walk is not bit-reproducible: checked through the id map's image only)
            assert (gi_np == src[wi_np]).all() and (gd.cpu().numpy() == wd.cpu().numpy()).all()
        assert np.isin(gi_np, src).all()
    h = str(tmp_path / "cagra_src_again.bin")
    cagra.save(h, loaded)
    p = rf.parse_cagra(h)
    assert p["content_map"] == 3 and (p["source_indices"] == src).all() and (p["graph"] == graph).all()


def test_cagra_hnswlib_export(tmp_path):
    import ctypes as C
    import torch
    from cuvs_amd._lib import check, lib
    from cuvs_amd.common import Resources
    from cuvs_amd.neighbors import cagra

    x, _ = _data(n=1500)
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=32, graph_degree=16), torch.from_numpy(x).cuda())
    graph = idx.graph.cpu().numpy().view(np.uint32)
    f = str(tmp_path / "cagra.hnsw")
    res = Resources()
    check(lib().cuvsCagraSerializeToHnswlib(res.get_c_obj(), C.c_char_p(f.encode()), idx._p))
    res.sync()
    p = rf.parse_hnswlib(f, 32, np.float32)
    assert (p["off0"], p["max_elements"], p["count"], p["max_level"], p["entry"]) == (0, 1500, 1500, 1, 750)
    assert (p["max_m"], p["m"]) == (8, 8) and p["ef_construction"] == 500 and abs(p["mult"] - 0.42424242) < 1e-12
    assert (p["degrees"] == 16).all() and (p["graph"] == graph).all() and (p["data"] == x).all()
    assert (p["labels"] == np.arange(1500)).all() and p["tail"].shape == (1500,) and (p["tail"] == 0).all()


def test_native_container_still_loads(tmp_path, monkeypatch):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    x, q = _data()
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=16, pq_dim=16), tx)
    monkeypatch.setenv("CUVS_AMD_NATIVE_FORMAT", "1")
    f = str(tmp_path / "pq_native.bin")
    ivf_pq.save(f, idx)
    assert open(f, "rb").read(8) == b"CUVSAMD1"
    sp = ivf_pq.SearchParams(n_probes=4)
    assert _same(ivf_pq.search(sp, idx, tq, 10), ivf_pq.search(sp, ivf_pq.load(f), tq, 10))


def test_bad_files(tmp_path):
    from cuvs_amd._lib import CuvsError
    from cuvs_amd.neighbors import brute_force, cagra, ivf_flat, ivf_pq

    x = np.zeros((10, 4), np.float32)
    f = str(tmp_path / "bf.bin")
    rf.write_brute_force(f, x)
    b = open(f, "rb").read()
    trunc = tmp_path / "trunc.bin"
    trunc.write_bytes(b[: len(b) - 40])
    with pytest.raises(CuvsError):
        brute_force.load(str(trunc))
    wrong = tmp_path / "wrong.bin"
    wrong.write_bytes(b"<i8\0" + b[4:])
    with pytest.raises(CuvsError):
        brute_force.load(str(wrong))
    for mod in (ivf_flat, ivf_pq, cagra):
        with pytest.raises(CuvsError):
            mod.load(f)  # a brute-force file is not any of the other index types
