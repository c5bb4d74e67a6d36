"""GPU: save/load round trips of all four index types (search results identical before/after) and the IVF-Flat
bitset pre-filter (reference: python/cuvs/cuvs/tests/test_serialization / test_ivf_flat filter cases)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _data(n=4000, d=32, q=60, seed=0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d)).astype(np.float32), rng.standard_normal((q, d)).astype(np.float32)


def test_roundtrip_all_indexes(tmp_path):
    import torch
    from cuvs_amd.neighbors import brute_force, cagra, ivf_flat, ivf_pq

    x, q = _data()
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()

    def same(a, b):
        torch.cuda.synchronize()
        return all(torch.equal(u, v) for u, v in zip(a, b))

    bf = brute_force.build(tx)
    f = str(tmp_path / "bf.bin"); brute_force.save(f, bf)
    assert same(brute_force.search(bf, tq, 10), brute_force.search(brute_force.load(f), tq, 10))

    fl = ivf_flat.build(ivf_flat.IndexParams(n_lists=16), tx)
    f = str(tmp_path / "flat.bin"); ivf_flat.save(f, fl)
    sp = ivf_flat.SearchParams(n_probes=4)
    assert same(ivf_flat.search(sp, fl, tq, 10), ivf_flat.search(sp, ivf_flat.load(f), tq, 10))

    pq = ivf_pq.build(ivf_pq.IndexParams(n_lists=16, pq_dim=16), tx)
    f = str(tmp_path / "pq.bin"); ivf_pq.save(f, pq)
    pq2 = ivf_pq.load(f)
    sp = ivf_pq.SearchParams(n_probes=4)
    assert same(ivf_pq.search(sp, pq, tq, 10), ivf_pq.search(sp, pq2, tq, 10))
    assert len(pq2) == 4000 and pq2.pq_dim == 16

    cg = cagra.build(cagra.IndexParams(intermediate_graph_degree=32, graph_degree=16), tx)
    f = str(tmp_path / "cagra.bin"); cagra.save(f, cg)
    sp = cagra.SearchParams(itopk_size=64, algo="single_cta")  # the multi-wave walk is not bit-reproducible
    assert same(cagra.search(sp, cg, tq, 10), cagra.search(sp, cagra.load(f), tq, 10))


def test_load_errors(tmp_path):
    from cuvs_amd._lib import CuvsError
    from cuvs_amd.neighbors import ivf_pq

    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not an index")
    with pytest.raises(CuvsError):
        ivf_pq.load(str(bad))
    with pytest.raises(CuvsError):
        ivf_pq.load(str(tmp_path / "missing.bin"))


def test_ivf_flat_bitset_filter():
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import ivf_flat

    x, q = _data(n=5000, d=24, q=80, seed=3)
    rng = np.random.default_rng(1)
    keep = rng.random(5000) < 0.4
    pad = np.zeros((-keep.size) % 32, bool)
    words = np.packbits(np.concatenate([keep, pad]), bitorder="little").view(np.uint32)
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=16), torch.from_numpy(x).cuda())
    tw = torch.from_numpy(words.view(np.int32)).cuda()
    d, i = ivf_flat.search(ivf_flat.SearchParams(n_probes=16), index, torch.from_numpy(q).cuda(), 10, filter=(tw, BITSET))
    torch.cuda.synchronize()
    gi = i.cpu().numpy()
    assert keep[gi].all()  # only kept rows are returned
    # all lists probed + filter == exact search restricted to the kept rows
    kept_ids = np.nonzero(keep)[0]
    td, ti = oracle.exact_knn(q, x[kept_ids], 10)
    assert oracle.recall(gi, kept_ids[ti]) > 0.999
