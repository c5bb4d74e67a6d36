"""GPU: the reference's OWN C-API test drivers (c/tests/neighbors/run_{brute_force,ivf_flat,ivf_pq}_c.c), compiled
unchanged against this repo's headers by oracle/build_ref.sh (oracle/_ref/libref_c_drivers.so), drive
cuvs_amd/libcuvs_c.so. Configurations and pass criteria are those of the gtest wrappers that call them in the
reference (c/tests/neighbors/brute_force_c.cu:394-440, ann_ivf_flat_c.cu:86-131, ann_ivf_pq_c.cu:88-131):
uniform[0.1, 2.0) data, 8096 x 32, 128 queries, k = 8, eval_neighbours with eps 1e-3 (id OR distance match) and
min_recall 0.95 (brute force) / n_probes / n_lists (IVF)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = os.path.join(ROOT, "oracle", "_ref", "libref_c_drivers.so")


def _drivers():
    if not os.path.exists(DRIVERS):
        pytest.skip("oracle/_ref/libref_c_drivers.so not built (needs /root/reference at build time)")
    return C.CDLL(DRIVERS)


def _data():
    import torch

    rng = np.random.default_rng(1234)
    x = (rng.random((8096, 32), dtype=np.float32) * 1.9 + 0.1).astype(np.float32)
    q = (rng.random((128, 32), dtype=np.float32) * 1.9 + 0.1).astype(np.float32)
    nb = torch.empty((128, 8), dtype=torch.int64, device="cuda")
    ds = torch.empty((128, 8), dtype=torch.float32, device="cuda")
    return x, q, torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), nb, ds


def _eval_neighbours(ids, dist, ref_ids, ref_dist, eps, min_recall):
    """cpp/tests/neighbors/ann_utils.cuh:222-252: a result counts when its id OR its distance matches an expected one."""
    hits = 0
    for r in range(ids.shape[0]):
        for j in range(ids.shape[1]):
            if ids[r, j] in ref_ids[r] or np.any(np.abs(ref_dist[r] - dist[r, j]) < eps):
                hits += 1
    assert hits / ids.size >= min_recall, hits / ids.size


def test_run_brute_force_c():
    import torch

    lib = _drivers()
    x, q, tx, tq, nb, ds = _data()
    lib.run_brute_force(C.c_int64(8096), C.c_int64(128), C.c_int64(32), C.c_uint32(8), C.c_void_p(tx.data_ptr()),
                        C.c_void_p(tq.data_ptr()), None, C.c_int(0), C.c_void_p(ds.data_ptr()), C.c_void_p(nb.data_ptr()),
                        C.c_int(0))
    torch.cuda.synchronize()
    td, ti = oracle.exact_knn(q, x, 8)
    _eval_neighbours(nb.cpu().numpy(), ds.cpu().numpy(), ti, td, 1e-3, 0.95)
    # with a bitset prefilter (brute_force_c.cu run_test_with_filter): only rows whose bit is set may come back
    keep = np.random.default_rng(5).random(8096) < 0.5
    words = torch.from_numpy(np.packbits(keep, bitorder="little").view(np.int32).copy()).cuda()
    lib.run_brute_force(C.c_int64(8096), C.c_int64(128), C.c_int64(32), C.c_uint32(8), C.c_void_p(tx.data_ptr()),
                        C.c_void_p(tq.data_ptr()), C.c_void_p(words.data_ptr()), C.c_int(1), C.c_void_p(ds.data_ptr()),
                        C.c_void_p(nb.data_ptr()), C.c_int(0))
    torch.cuda.synchronize()
    got = nb.cpu().numpy()
    assert keep[got].all()
    td, ti = oracle.exact_knn(q, x[keep], 8)
    _eval_neighbours(got, ds.cpu().numpy(), np.nonzero(keep)[0][ti], td, 1e-3, 0.95)


@pytest.mark.parametrize("driver", ["run_ivf_flat", "run_ivf_pq"])
def test_run_ivf_c(driver):
    import torch

    lib = _drivers()
    x, q, tx, tq, nb, ds = _data()
    n_probes, n_lists = 20, 1024
    getattr(lib, driver)(C.c_int64(8096), C.c_int64(128), C.c_int64(32), C.c_uint32(8), C.c_void_p(tx.data_ptr()),
                         C.c_void_p(tq.data_ptr()), C.c_void_p(ds.data_ptr()), C.c_void_p(nb.data_ptr()), C.c_int(0),
                         C.c_size_t(n_probes), C.c_size_t(n_lists))
    torch.cuda.synchronize()
    td, ti = oracle.exact_knn(q, x, 8)
    _eval_neighbours(nb.cpu().numpy(), ds.cpu().numpy(), ti, td, 1e-3, n_probes / n_lists)


@pytest.mark.parametrize("metric,code", [("sqeuclidean", 0), ("inner_product", 6), ("cosine", 2)])
def test_run_pairwise_distance_c(metric, code):
    """c/tests/distance/run_pairwise_distance_c.c through cuvsPairwiseDistance (c/tests/distance/pairwise_distance_c.cu
    configuration: 8096 x 32 against 128 x 32); result must equal the canonical oracle bit for bit."""
    import torch

    lib = _drivers()
    x, q, tx, tq, _, _ = _data()
    out = torch.empty((8096, 128), dtype=torch.float32, device="cuda")
    lib.run_pairwise_distance(C.c_int64(8096), C.c_int64(128), C.c_int64(32), C.c_void_p(tx.data_ptr()),
                              C.c_void_p(tq.data_ptr()), C.c_void_p(out.data_ptr()), None, C.c_int(code))
    torch.cuda.synchronize()
    want = oracle.pairwise(x, q, metric=metric)
    assert (out.cpu().numpy() == want).all()


def test_core_c_api_program():
    """c/tests/core/c_api.c is a whole program (resources, stream set, device alloc/free with and without the pool,
    pinned host memory, version == <cuvs/version_config.h>); it exits 0 when every call returned CUVS_SUCCESS."""
    import subprocess

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_core_c_api")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_core_c_api not built (needs /root/reference at build time)")
    proc = subprocess.run([exe], capture_output=True, timeout=120)
    assert proc.returncode == 0, proc.stderr.decode()[-400:]
