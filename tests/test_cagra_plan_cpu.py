"""CPU: the CAGRA search plan (host arithmetic only, no GPU) pinned to the reference's rules -
cpp/src/neighbors/detail/cagra/search_plan.cuh:121-131 (AUTO), :199-245 (max_iterations, filter-rate itopk, rounding),
:248-340 (hash sizes) - at the BASELINE C4 shape and against an independent Python transcription."""
import ctypes as C
import math

import numpy as np
import pytest

SINGLE, MULTI, MULTI_KERNEL = 0, 1, 2


def gpu_plan(n_rows, degree, topk, nq, num_cus=256, filtering_rate=0.0, **kw):
    from cuvs_amd._lib import check, lib
    from cuvs_amd.neighbors import cagra

    sp = cagra.SearchParams(**kw)
    out = (C.c_uint32 * 10)()
    fn = lib().cuvsAmdCagraSearchPlan
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int, C.c_float, C.c_void_p]
    check(fn(sp._p, n_rows, degree, topk, nq, num_cus, filtering_rate, out))
    keys = ["itopk", "max_iterations", "search_width", "algo", "small_hash_bitlen", "hash_bitlen", "reset_interval",
            "mc_num_cta_per_query", "mc_max_iterations"]
    return dict(zip(keys, list(out)))


def ref_plan(n_rows, degree, topk, nq, num_sm=256, filtering_rate=0.0, itopk_size=64, search_width=1, max_iterations=0,
             min_iterations=0, algo="auto", hashmap_mode="auto", hashmap_min_bitlen=0, hashmap_max_fill_rate=0.5,
             max_queries=0):
    """Line-by-line Python transcription of search_plan.cuh (written from the .cuh, independently of cagra.hip)."""
    itopk = max(itopk_size, topk)
    mq = max_queries or nq
    if algo == "auto":
        algo = SINGLE if (itopk <= 512 and mq >= num_sm * 2) else MULTI
    else:
        algo = {"single_cta": SINGLE, "multi_cta": MULTI, "multi_kernel": MULTI_KERNEL}[algo]
    _max = max_iterations
    if max_iterations == 0:
        _max = (32 // 1) if algo == MULTI else itopk // search_width
        reach = 1
        while reach < n_rows:
            reach *= max(2, degree // 2)
            _max += 1
    if max_iterations < min_iterations:
        _max = min_iterations
    if max_iterations < _max:
        max_iterations = _max
    if algo == MULTI and 0.0 < filtering_rate < 1.0:
        adj = int(np.float32(topk) / (1.0 - filtering_rate) + np.float32(itopk - topk) / math.sqrt(1.0 - filtering_rate))
        if adj % 32:
            adj += 32 - adj % 32
        itopk = max(itopk, adj)
    if itopk % 32:
        itopk += 32 - itopk % 32
    out = dict(itopk=itopk, max_iterations=max_iterations, search_width=search_width, algo=algo, small_hash_bitlen=0,
               hash_bitlen=0, reset_interval=1024 * 1024, mc_num_cta_per_query=0)
    fill = hashmap_max_fill_rate
    if algo == MULTI:
        ncta = max(search_width, -(-itopk // 32))
        out["mc_num_cta_per_query"] = ncta
        sb = 8
        while 32 + degree * 2 > (1 << sb) * fill:
            sb += 1
        out["small_hash_bitlen"] = sb
        hb = max(11, hashmap_min_bitlen)
        while ncta * max(32, max_iterations) > (1 << hb) * fill:
            hb += 1
        out["hash_bitlen"] = hb
    else:
        hb = 0
        if hashmap_mode in ("auto", "small"):
            hb = max(8, hashmap_min_bitlen)
            while itopk + search_width * degree > (1 << hb) * fill:
                hb += 1
            if hb > 13:
                hb = 0
            else:
                out["small_hash_bitlen"] = hb
                r = 1
                while not (itopk + search_width * degree * (r + 1) > (1 << hb) * fill):
                    r += 1
                out["reset_interval"] = r
        if hb == 0:
            hb = max(11, hashmap_min_bitlen)
            while itopk + search_width * degree * max_iterations > (1 << hb) * fill:
                hb += 1
        out["hash_bitlen"] = hb
    return out


def test_c4_shape_pins():
    """10M rows, degree 64, itopk 64, k 10, 10k queries."""
    auto = gpu_plan(10_000_000, 64, 10, 10000)
    assert auto["algo"] == SINGLE                      # 10k queries >= 2 x 256 CUs
    assert auto["itopk"] == 64 and auto["max_iterations"] == 64 + 5  # reach x32 per iteration until >= 10M: 5 steps
    # 64 + 1 * 64 = 128 is NOT > 2^8 * 0.5 (strict inequality, search_plan.cuh:298): the small hash stays at 2^8 and is
    # reset every iteration (64 + 64 * 2 = 192 > 128). (SURVEY 8 a16 says 2^9; the rule says 2^8.)
    assert auto["small_hash_bitlen"] == 8
    assert auto["hash_bitlen"] == 8 and auto["reset_interval"] == 1
    multi = gpu_plan(10_000_000, 64, 10, 10000, algo="multi_cta")
    assert multi["algo"] == MULTI and multi["mc_num_cta_per_query"] == 2
    assert multi["max_iterations"] == 32 + 5 and multi["mc_max_iterations"] == 37
    assert multi["small_hash_bitlen"] == 9             # 32 + 128 = 160 > 128
    assert multi["hash_bitlen"] == 11                  # traversed hash >= 2^11
    small_batch = gpu_plan(10_000_000, 64, 10, 100)
    assert small_batch["algo"] == MULTI                # 100 queries < 2 x 256 CUs


@pytest.mark.parametrize("n_rows,degree,topk,nq,kw", [
    (10_000_000, 64, 10, 10000, {}),
    (10_000_000, 64, 10, 10000, dict(algo="multi_cta")),
    (1_000_000, 32, 100, 64, dict(itopk_size=100)),
    (5000, 16, 16, 1000, dict(itopk_size=512, search_width=4)),
    (200_000, 128, 10, 5000, dict(itopk_size=256, search_width=8)),        # small hash does not fit -> normal hash
    (3_000_000, 64, 10, 10, dict(itopk_size=64, max_iterations=20, min_iterations=30)),
    (3_000_000, 64, 10, 10000, dict(itopk_size=64, min_iterations=10)),   # max_iterations 0: the walk runs min_iterations
    (3_000_000, 64, 10, 10, dict(algo="multi_cta", min_iterations=12)),
    (1_000_000, 64, 10, 100, dict(algo="multi_cta", itopk_size=128, hashmap_min_bitlen=13)),
    (50_000, 32, 8, 10000, dict(algo="single_cta", hashmap_mode="hash")),
])
@pytest.mark.parametrize("rate", [0.0, 0.5, 0.9])
def test_plan_equals_the_transcribed_rules(n_rows, degree, topk, nq, kw, rate):
    got = gpu_plan(n_rows, degree, topk, nq, filtering_rate=rate, **kw)
    want = ref_plan(n_rows, degree, topk, nq, filtering_rate=rate, **kw)
    for key, val in want.items():
        assert got[key] == val, (key, got, want)
