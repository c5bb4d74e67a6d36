"""CPU: the host-side arithmetic of the multi-GPU layer (cuvs_amd/csrc/mg_host.hpp, plain C++): row ranges of the
ranks, batch plans and the merge of per-shard result lists, against numpy restatements of snmg.cuh / knn_merge_parts."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WRAPPER = r"""
#include "mg_host.hpp"
using namespace cuvs_amd;
extern "C" {
void t_rows(int sharded, int64_t n, int r, int R, int64_t* r0, int64_t* cnt) { rows_of_rank(sharded != 0, n, r, R, r0, cnt); }
void t_rep(int64_t nq, int64_t per, int R, int64_t* b, int64_t* nb) { replicated_batches(nq, per, R, b, nb); }
void t_shard(int64_t nq, int64_t per, int64_t* b, int64_t* nb) { sharded_batches(nq, per, b, nb); }
void t_merge(const int64_t* pi, const float* pd, int R, int64_t cnt, int64_t k, const int64_t* tr, int select_min,
             int64_t* oi, float* od) { merge_on_host(pi, pd, R, cnt, k, tr, select_min != 0, oi, od); }
}
"""


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("mg_host")
    src, so = d / "w.cpp", d / "w.so"
    src.write_text(_WRAPPER)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "cuvs_amd", "csrc"),
                           str(src), "-o", str(so)])
    return C.CDLL(str(so))


def test_row_ranges_cover_the_dataset_once(host):
    for n in (1, 7, 5000, 6001, 1 << 20):
        for R in (1, 2, 3, 8):
            covered, per = [], -(-n // R)
            for r in range(R):
                r0, cnt = C.c_int64(), C.c_int64()
                host.t_rows(1, C.c_int64(n), r, R, C.byref(r0), C.byref(cnt))
                assert cnt.value == max(0, min(per, n - r * per))  # snmg.cuh:131-150
                covered += list(range(r0.value, r0.value + cnt.value)) if n <= 6001 else []
                host.t_rows(0, C.c_int64(n), r, R, C.byref(r0), C.byref(cnt))
                assert (r0.value, cnt.value) == (0, n)  # replicated: everything everywhere
            if n <= 6001:
                assert covered == list(range(n))


def test_batch_plans(host):
    b, nb = C.c_int64(), C.c_int64()
    for nq, per, R in [(1000, 3000, 1), (1000, 3000, 2), (1000, 300, 2), (10, 3000, 8), (10000, 1 << 20, 8), (5, 1, 2)]:
        host.t_rep(C.c_int64(nq), C.c_int64(per), R, C.byref(b), C.byref(nb))
        want = min(per, -(-nq // R))  # snmg.cuh:596-600
        want_nb = -(-nq // want)
        if want_nb <= 1:
            want, want_nb = nq, 1
        assert (b.value, nb.value) == (want, want_nb)
        assert b.value * nb.value >= nq > b.value * (nb.value - 1)
        host.t_shard(C.c_int64(nq), C.c_int64(per), C.byref(b), C.byref(nb))
        want_nb = -(-nq // per)
        assert (b.value, nb.value) == ((nq, 1) if want_nb <= 1 else (per, want_nb))


@pytest.mark.parametrize("select_min", [True, False])
def test_merge_matches_a_numpy_restatement(host, select_min):
    rng = np.random.default_rng(5)
    R, cnt, k = 3, 40, 6
    pi = rng.integers(0, 50, size=(R, cnt, k)).astype(np.int64)
    pd = rng.integers(0, 8, size=(R, cnt, k)).astype(np.float32)  # few distinct values: plenty of ties
    pi[rng.random((R, cnt, k)) < 0.2] = np.iinfo(np.int64).max  # IVF "no neighbour"
    pi[rng.random((R, cnt, k)) < 0.1] = -1                      # CAGRA "no neighbour"
    pi[:, 0] = -1                                                # a query nobody answers
    pi[1:, 1] = -1                                               # a query with fewer than k answers in total
    pi[0, 1, 3:] = -1
    tr = np.array([0, 1000, 5000], dtype=np.int64)
    oi = np.empty((cnt, k), dtype=np.int64)
    od = np.empty((cnt, k), dtype=np.float32)
    host.t_merge(pi.ctypes.data_as(C.c_void_p), pd.ctypes.data_as(C.c_void_p), R, C.c_int64(cnt), C.c_int64(k),
                 tr.ctypes.data_as(C.c_void_p), int(select_min), oi.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p))
    big, worst = np.iinfo(np.int64).max, np.float32(np.finfo(np.float32).max if select_min else -np.finfo(np.float32).max)
    for q in range(cnt):
        cand = [(float(pd[r, q, j]), int(pi[r, q, j]) + int(tr[r])) for r in range(R) for j in range(k)
                if 0 <= pi[r, q, j] != big]
        cand.sort(key=lambda t: (t[0] if select_min else -t[0], t[1]))  # knn_merge_parts order; ties -> smaller id
        want = cand[:k] + [(float(worst), big)] * (k - len(cand[:k]))
        assert [int(v) for v in oi[q]] == [c[1] for c in want], q
        assert [float(v) for v in od[q]] == [c[0] for c in want], q
    assert (oi[0] == big).all() and (oi[1, 3:] == big).all() and (oi[1, :3] < 1000).all()


def test_deal_lists_is_lpt_and_deterministic():
    """cuvsAmdShardDealLists (no GPU needed): greedy longest-processing-time dealing - every list exactly one owner, the
    heaviest rank within the LPT bound of the mean, the same table for the same weights."""
    from cuvs_amd.neighbors import ivf_pq_sharded as sh

    rng = np.random.default_rng(3)
    w = (rng.pareto(1.5, size=4096) * 1000 + 10).astype(np.uint64)
    for world in (1, 2, 3, 8):
        o1, o2 = sh.deal_lists(w, world), sh.deal_lists(w, world)
        assert (o1 == o2).all() and o1.min() >= 0 and o1.max() < world
        loads = np.array([w[o1 == r].sum() for r in range(world)], dtype=np.float64)
        assert loads.max() <= max(float(w.max()), loads.mean() * (4 / 3))
        # reference python restatement of the rule
        order = np.argsort(-w.astype(np.int64), kind="stable")
        ld = np.zeros(world)
        want = np.empty(len(w), np.int32)
        for L in order:
            r = int(np.argmin(ld))
            want[L] = r
            ld[r] += float(w[L])
        assert (want == o1).all()
