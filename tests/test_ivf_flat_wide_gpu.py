"""GPU: IVF-Flat's tail phase on the wide matrix-core filter (ivf_pq_wide.hip) - 256 / 384 / 512 / 768 dimensions, where rounds 3-6 had the
two-waves-per-SIMD filter (256) or the scan kernel alone (beyond). The fp16 residual copy in the natural K order, fp32 row terms, the
pre-pass in blocks of 32 pairs per list, the filter, the scan kernel's fp32 chain for the survivors (reference semantics:
interleaved_scan_impl.cuh:71-206): ids and distances equal to the CPU oracle and to the scan kernel alone (CUVS_AMD_FLAT_SCAN3=0)."""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _mixture(n, d, q, seed, modes=200, latent=16, sigma=1.0):
    rng = np.random.default_rng(seed)
    basis = rng.standard_normal((latent, d)).astype(np.float32) / math.sqrt(latent)
    centres = rng.standard_normal((modes, latent)).astype(np.float32) * 2.0

    def draw(m):
        z = centres[rng.integers(0, modes, size=m)] + sigma * rng.standard_normal((m, latent)).astype(np.float32)
        return (z @ basis + 0.02 * rng.standard_normal((m, d)).astype(np.float32)).astype(np.float32)

    return draw(n), draw(q)


def _search(index, q, k, n_probes):
    import torch
    from cuvs_amd.neighbors import ivf_flat

    d, i = ivf_flat.search(ivf_flat.SearchParams(n_probes=n_probes), index, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy()


@pytest.mark.parametrize("dim,dtype,metric,k", [
    (768, np.float16, "sqeuclidean", 10),
    (768, np.float32, "sqeuclidean", 64),
    (768, np.float32, "inner_product", 10),
    (512, np.float16, "cosine", 20),
    (384, np.float32, "euclidean", 10),
    (256, np.float32, "sqeuclidean", 100),
    (256, np.int8, "sqeuclidean", 10),     # integer rows: exact in the fp32 chain up to 256 dimensions
    (256, np.uint8, "inner_product", 10),
])
def test_flat_wide_filter_equals_oracle_and_scan_kernel(dim, dtype, metric, k, monkeypatch):
    import torch
    from cuvs_amd.neighbors import ivf_flat

    x, q = _mixture(50_000, dim, 400, seed=dim + k)
    if dtype == np.int8:
        x, q = np.clip(np.rint(x * 40), -127, 127).astype(np.int8), np.clip(np.rint(q * 40), -127, 127).astype(np.int8)
    elif dtype == np.uint8:
        x, q = np.clip(np.rint(x * 30 + 128), 0, 255).astype(np.uint8), np.clip(np.rint(q * 30 + 128), 0, 255).astype(np.uint8)
    else:
        x, q = x.astype(dtype), q.astype(dtype)
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=32, kmeans_n_iters=8, kmeans_trainset_fraction=0.3, metric=metric),
                           torch.from_numpy(x).cuda())
    ex = ivf_flat.export_for_oracle(index, dtype)
    n_probes = 12
    gd, gi = _search(index, q, k, n_probes)
    scale = 1 / 128 if dtype == np.int8 else 1 / 256 if dtype == np.uint8 else 1.0  # (the coarse quantizer sees mapped floats: ann_utils.cuh:134-196)
    od, oi = oracle.ivf_flat_search(ex, q, k, n_probes, metric=metric, coarse_scale=scale)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.5f}"
    assert (gd == od).all()
    monkeypatch.setenv("CUVS_AMD_PQ3_SURV_CAP", "2000")  # the survivor buffer runs over: the tail phase is re-run on the scan kernel
    hd, hi = _search(index, q, k, n_probes)
    assert (gi == hi).all() and (gd == hd).all()
    monkeypatch.delenv("CUVS_AMD_PQ3_SURV_CAP")
    monkeypatch.setenv("CUVS_AMD_FLAT_SCAN3", "0")
    sd, si = _search(index, q, k, n_probes)
    assert (gi == si).all() and (gd == sd).all()


def test_flat_wide_filter_after_extend_and_with_a_bitset(monkeypatch):
    """rows added after the fp16 copy was made (it is rebuilt); a bitset pre-filter (applied by the re-score): equal to the scan kernel alone under the same bitset"""
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import ivf_flat

    x_all, q = _mixture(45_000, 768, 300, seed=3)
    x, x2 = x_all[:40_000], x_all[40_000:]
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=32, kmeans_n_iters=8, kmeans_trainset_fraction=0.3), torch.from_numpy(x).cuda())
    gd, gi = _search(index, q, 10, 12)
    index = ivf_flat.extend(index, torch.from_numpy(x2).cuda(), torch.arange(40_000, 45_000, dtype=torch.int64).cuda())
    ex = ivf_flat.export_for_oracle(index, np.float32)
    gd, gi = _search(index, q, 10, 12)
    od, oi = oracle.ivf_flat_search(ex, q, 10, 12)
    assert (gi == oi).all() and (gd == od).all()
    assert (gi >= 40_000).any(), "the extension's rows are found"
    keep = np.random.default_rng(1).random(45_000) < 0.5
    words = np.packbits(keep, bitorder="little")
    words = np.concatenate([words, np.zeros((-len(words)) % 4, dtype=np.uint8)]).view(np.uint32)
    tw = torch.from_numpy(words.view(np.int32)).cuda()

    def filtered():
        d, i = ivf_flat.search(ivf_flat.SearchParams(n_probes=12), index, torch.from_numpy(q).cuda(), 10, filter=(tw, BITSET))
        torch.cuda.synchronize()
        return d.cpu().numpy(), i.cpu().numpy()

    fd, fi = filtered()
    monkeypatch.setenv("CUVS_AMD_FLAT_SCAN3", "0")  # the scan kernel alone, the same bitset
    sd, si = filtered()
    assert (fi == si).all() and (fd == sd).all()
    found = fi[fi >= 0]
    assert keep[found[found < 45_000]].all(), "a rejected row came back"
