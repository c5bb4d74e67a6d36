"""The handle's scratch cache (cuvs_amd/csrc/core.hip: device_alloc / device_free keep freed blocks by exact size, DESIGN
3.1d): same results with the cache, without it (CUVS_AMD_ALLOC_CACHE=0) and with a cache too small to keep anything
(CUVS_AMD_ALLOC_CACHE_MB=1: every free empties it), and the block re-use rule seen through cuvsRMMAlloc / cuvsRMMFree
(c_api.h:74-75; the reference hands these to rmm's stream-ordered pool)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rmm(handle):
    from cuvs_amd._lib import check, lib

    L = lib()
    L.cuvsRMMAlloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p), C.c_size_t]
    L.cuvsRMMFree.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t]

    def alloc(n):
        p = C.c_void_p()
        check(L.cuvsRMMAlloc(handle, C.byref(p), n))
        assert p.value
        return p

    def free(p, n):
        check(L.cuvsRMMFree(handle, p, n))

    return alloc, free


def test_a_freed_block_serves_the_next_request_of_its_size(monkeypatch):
    import cuvs_amd

    monkeypatch.delenv("CUVS_AMD_ALLOC_CACHE", raising=False)
    monkeypatch.delenv("CUVS_AMD_ALLOC_CACHE_MB", raising=False)
    res = cuvs_amd.common.Resources()
    alloc, free = _rmm(res.get_c_obj())
    n = 3 << 20
    a = alloc(n)
    free(a, n)
    b = alloc(n)          # exact size: the kept block
    assert b.value == a.value
    c = alloc(n)          # b is live: a different block
    assert c.value != b.value
    d = alloc(n + 256)    # another size never gets a kept block of this one
    assert d.value not in (b.value, c.value)
    free(c, n)
    free(b, n)
    e, f = alloc(n), alloc(n)
    assert {e.value, f.value} == {b.value, c.value}
    for p, sz in ((d, n + 256), (e, n), (f, n)):
        free(p, sz)
    res.sync()


def test_changing_the_stream_gives_the_kept_blocks_back(monkeypatch):
    import torch
    import cuvs_amd
    from cuvs_amd._lib import check, lib

    monkeypatch.delenv("CUVS_AMD_ALLOC_CACHE", raising=False)
    res = cuvs_amd.common.Resources()
    alloc, free = _rmm(res.get_c_obj())
    n = 1 << 20
    a = alloc(n)
    free(a, n)
    side = torch.cuda.Stream()
    check(lib().cuvsStreamSet(res.get_c_obj(), C.c_void_p(side.cuda_stream)))
    b = alloc(n)  # from the pool, on the new stream; kept for it afterwards
    free(b, n)
    c = alloc(n)
    assert c.value == b.value
    free(c, n)
    res.sync()
    del res
    torch.cuda.synchronize()


@pytest.mark.parametrize("env", [{}, {"CUVS_AMD_ALLOC_CACHE": "0"}, {"CUVS_AMD_ALLOC_CACHE_MB": "1"}])
def test_searches_do_not_depend_on_the_cache(monkeypatch, env):
    """IVF-Flat (matrix-core tail phase: ~30 scratch buffers, one flag read back) and brute force, three searches each
    on one handle, against a handle with the default cache."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import brute_force, ivf_flat

    rng = np.random.default_rng(77)
    centers = rng.standard_normal((64, 64)).astype(np.float32)
    x = (centers[rng.integers(0, 64, 30_000)] + 0.3 * rng.standard_normal((30_000, 64))).astype(np.float32)
    q = (centers[rng.integers(0, 64, 320)] + 0.3 * rng.standard_normal((320, 64))).astype(np.float32)
    xd, qd = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()

    for key in ("CUVS_AMD_ALLOC_CACHE", "CUVS_AMD_ALLOC_CACHE_MB"):
        monkeypatch.delenv(key, raising=False)
    base = cuvs_amd.common.Resources()
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=16, kmeans_n_iters=8), xd, resources=base)
    bf = brute_force.build(xd, metric="sqeuclidean", resources=base)
    base.sync()

    def run(res):  # a built index may be searched through any handle
        out = []
        for _ in range(3):
            d, i = ivf_flat.search(ivf_flat.SearchParams(n_probes=12), index, qd, 10, resources=res)
            bd, bi = brute_force.search(bf, qd, 10, resources=res)
            res.sync()
            out.append((d.cpu().numpy(), i.cpu().numpy(), bd.cpu().numpy(), bi.cpu().numpy()))
        return out

    ref = run(base)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    got = run(cuvs_amd.common.Resources())
    for r, g in zip(ref, got):
        for a, b in zip(r, g):
            assert (a == b).all()
    for a, b in zip(ref[0], ref[2]):
        assert (a == b).all()
