"""The reference's own parameter tables for the IVF-PQ / IVF-Flat / CAGRA search tests, transcribed as DATA (test
infrastructure; nothing here is product code). Each table cites the reference lines it restates; a case keeps the reference's
defaults unless a field says otherwise, and carries the reference's explicit `min_recall` where the table gives one.

The reference holds NO golden output vectors for the IVF / CAGRA searches - these tables (parameters + recall thresholds +
the data generators' ranges) are the only pins it holds for them, so the GPU tests run every case id listed here
(tests/test_reference_tables_gpu.py) with the reference's own pass criterion (eval_neighbours: a result counts when its id OR
its distance matches an expected one, cpp/tests/neighbors/ann_utils.cuh:114-125,222-289)."""
import math

# --------------------------------------------------------------------------------------------------------------- IVF-PQ
# cpp/tests/neighbors/ann_ivf_pq.cuh:27-43: ivf_pq_inputs defaults
IVF_PQ_DEFAULTS = dict(num_db_vecs=4096, num_queries=1024, dim=64, k=32, min_recall=None,
                       # index_params (cpp/include/cuvs/neighbors/ivf_pq.hpp:40-140 defaults, n_lists / trainset fraction as set by
                       # the constructor at :38-42: n_lists = max(32, min(1024, num_db_vecs / 128)) = 32)
                       n_lists=32, kmeans_trainset_fraction=1.0, metric="sqeuclidean", pq_bits=8, pq_dim=0, codebook_kind="subspace",
                       force_random_rotation=False,
                       # search_params defaults (ivf_pq.hpp search_params: n_probes 20, fp32 LUT / score / coarse)
                       n_probes=20, lut_dtype="f32", internal_distance_dtype="f32", coarse_search_dtype="f32")


def _case(**kw):
    c = dict(IVF_PQ_DEFAULTS)
    c.update(kw)
    return c


def _div_up(a, b):
    return (a + b - 1) // b


def _round_up(a, b):
    return _div_up(a, b) * b


def ivf_pq_defaults():            # ann_ivf_pq.cuh:889
    return [_case()]


def ivf_pq_small_dims():          # :909-910  "These will surely trigger the fastest kernel available."
    return [_case(dim=d) for d in (1, 2, 3, 4, 5, 8, 15, 16, 17)]


def ivf_pq_small_dims_per_cluster():   # :912-919
    return [dict(c, codebook_kind="cluster") for c in ivf_pq_small_dims()]


def ivf_pq_big_dims():            # :921-933  pq_len 2, min_recall = 0.48 + 0.028 log2(dim)
    return [_case(dim=d, pq_dim=_div_up(d, 2), min_recall=0.48 + 0.028 * math.log2(d))
            for d in (512, 513, 1023, 1024, 1025, 2048, 2049, 2050, 2053, 6144)]


def ivf_pq_big_dims_moderate_lut():    # :935-947  "These will surely trigger no-smem-lut kernel."
    return [dict(c, pq_dim=_round_up(_div_up(c["dim"], 2), 4), pq_bits=6, lut_dtype="f16", min_recall=0.69) for c in ivf_pq_big_dims()]


def ivf_pq_big_dims_small_lut():       # :949-961  "Some of these should trigger no-basediff kernel."
    return [dict(c, pq_dim=_round_up(_div_up(c["dim"], 8), 4), pq_bits=6, lut_dtype="u8", min_recall=0.21) for c in ivf_pq_big_dims()]


def ivf_pq_enum_variety():        # :976-1062  "A minimal set of tests to check various enum-like parameters."
    return [
        _case(codebook_kind="cluster", min_recall=0.86),                                  # :980-983
        _case(codebook_kind="subspace", min_recall=0.86),                                 # :984-987
        _case(codebook_kind="cluster", pq_bits=4, min_recall=0.79),                       # :988-992
        _case(codebook_kind="cluster", pq_bits=5, min_recall=0.83),                       # :993-997
        _case(pq_bits=6, min_recall=0.84),                                                # :999-1002
        _case(pq_bits=7, min_recall=0.85),                                                # :1003-1006
        _case(pq_bits=8, min_recall=0.86),                                                # :1007-1010
        _case(force_random_rotation=True, min_recall=0.86),                               # :1012-1015
        _case(force_random_rotation=False, min_recall=0.86),                              # :1016-1019
        _case(lut_dtype="f32", min_recall=0.86),                                          # :1021-1024
        _case(lut_dtype="f16", min_recall=0.86),                                          # :1025-1028
        _case(lut_dtype="u8", min_recall=0.84),                                           # :1029-1032
        _case(coarse_search_dtype="f16", min_recall=0.86),                                # :1033-1036
        _case(coarse_search_dtype="i8", min_recall=0.1),                                  # :1037-1043 ("experimental ... no guarantee")
        _case(internal_distance_dtype="f32", min_recall=0.86),                            # :1045-1048
        _case(internal_distance_dtype="f16", lut_dtype="f16", min_recall=0.86),           # :1049-1053
        _case(internal_distance_dtype="f16", lut_dtype="f16", coarse_search_dtype="f16", min_recall=0.86),   # :1054-1059
    ]


def ivf_pq_enum_variety_l2():     # :1064-1071
    return [dict(c, metric="sqeuclidean") for c in ivf_pq_enum_variety()]


def ivf_pq_enum_variety_ip():     # :1073-1092  8-bit LUT: x 0.88 (signed entries lose a bit), otherwise x 0.94
    return [dict(c, metric="inner_product", min_recall=c["min_recall"] * (0.88 if c["lut_dtype"] == "u8" else 0.94))
            for c in ivf_pq_enum_variety()]


def ivf_pq_enum_variety_l2sqrt():  # :1094-1101
    return [dict(c, metric="euclidean") for c in ivf_pq_enum_variety()]


def ivf_pq_enum_variety_cosine():  # :1103-1122  8-bit LUT: x 0.70 (cuvs issue 390), otherwise x 0.94
    return [dict(c, metric="cosine", min_recall=c["min_recall"] * (0.70 if c["lut_dtype"] == "u8" else 0.94))
            for c in ivf_pq_enum_variety()]


def ivf_pq_var_n_probes():        # :1128-1140  n_lists, n_lists / 2, ..., 1
    out, x = [], IVF_PQ_DEFAULTS["n_lists"]
    while x >= 1:
        out.append(_case(n_probes=x))
        x //= 2
    return out


def ivf_pq_var_k():               # :1160-1171  n_probes = max(n_probes, min(n_lists, k))
    return [_case(k=k, n_probes=max(IVF_PQ_DEFAULTS["n_probes"], min(IVF_PQ_DEFAULTS["n_lists"], k)))
            for k in (1, 2, 3, 5, 8, 15, 16, 32, 63, 65, 127, 128, 256, 257, 1023, 2048, 2049)]


def ivf_pq_special_cases():       # :1176-1245 - the two that run anywhere: k 128 / 129 of 10000 x 16, every list probed
    # (the three 1183514- / 500000-row cases with up to 128 Mi queries of :1179-1222 are not instantiated by any reference
    # test file either: `special_cases()` appears in no INSTANTIATE list)
    return [_case(num_db_vecs=10000, dim=16, num_queries=500, k=128, n_lists=100, n_probes=100),
            _case(num_db_vecs=10000, dim=16, num_queries=500, k=129, n_lists=100, n_probes=100)]


def ivf_pq_flat_layout_tests():   # ann_ivf_pq.cuh:1293-1338  "Test cases for flat layout comparison."
    base = dict(num_db_vecs=1000, dim=64, n_lists=10, pq_dim=16)
    return [_case(pq_bits=8, **base), _case(pq_bits=6, **base), _case(pq_bits=8, codebook_kind="cluster", **base),
            _case(pq_bits=6, codebook_kind="cluster", **base)]


# what each reference test file instantiates (cpp/tests/neighbors/ann_ivf_pq/test_{float,int8_t,uint8_t}_int64_t.cu:19-21,17-19,17-19)
IVF_PQ_TABLES = {
    "f32": [("defaults", ivf_pq_defaults), ("small_dims", ivf_pq_small_dims), ("big_dims_moderate_lut", ivf_pq_big_dims_moderate_lut),
            ("enum_variety_l2", ivf_pq_enum_variety_l2), ("enum_variety_l2sqrt", ivf_pq_enum_variety_l2sqrt),
            ("enum_variety_ip", ivf_pq_enum_variety_ip), ("enum_variety_cosine", ivf_pq_enum_variety_cosine),
            # defined in the header, instantiated by no reference file - run here for float rows:
            ("var_n_probes", ivf_pq_var_n_probes), ("special_cases", ivf_pq_special_cases), ("big_dims_small_lut", ivf_pq_big_dims_small_lut)],
    "i8": [("defaults", ivf_pq_defaults), ("big_dims", ivf_pq_big_dims), ("var_k", ivf_pq_var_k), ("enum_variety_l2", ivf_pq_enum_variety_l2),
           ("enum_variety_ip", ivf_pq_enum_variety_ip), ("enum_variety_cosine", ivf_pq_enum_variety_cosine)],
    "u8": [("small_dims_per_cluster", ivf_pq_small_dims_per_cluster), ("enum_variety", ivf_pq_enum_variety),
           ("enum_variety_l2", ivf_pq_enum_variety_l2), ("enum_variety_l2sqrt", ivf_pq_enum_variety_l2sqrt),
           ("enum_variety_ip", ivf_pq_enum_variety_ip), ("enum_variety_cosine", ivf_pq_enum_variety_cosine)],
}


def ivf_pq_min_recall(case, pq_dim_effective):
    """ann_ivf_pq.cuh:596-597,639-646: compression_ratio = dim * 8 / (pq_dim * pq_bits); a 'very conservative lower bound':
    p = n_probes / n_lists, min(erfc(0.05 cr / max(p, 0.5)), p), unless the case gives one. Returns (min_recall, eps)."""
    cr = case["dim"] * 8.0 / (pq_dim_effective * case["pq_bits"])
    p = case["n_probes"] / case["n_lists"]
    mr = min(math.erfc(0.05 * cr / max(p, 0.5)), p)
    if case["min_recall"] is not None:
        mr = case["min_recall"]
    return mr, 0.0001 * cr


# ------------------------------------------------------------------------------------------------------------- IVF-Flat
# cpp/tests/neighbors/ann_ivf_flat.cuh:524-661: {num_queries, num_db_vecs, dim, k, nprobe, nlist, metric, adaptive_centers
# [, host_dataset [, kernel_copy_overlapping]]}; min_recall = nprobe / nlist (:102), eps 1e-3 (fp16 rows: 5e-3, :245)
L2, L2S, IP, COS = "sqeuclidean", "euclidean", "inner_product", "cosine"
KMAX = 256  # raft::matrix::detail::select::warpsort::kMaxCapacity
IVF_FLAT_CASES = [
    # test various dims (aligned and not aligned to vector sizes)  :525-540
    (1000, 10000, 1, 16, 40, 1024, L2, True), (1000, 10000, 2, 16, 40, 1024, L2, False), (1000, 10000, 2, 16, 40, 1024, COS, False),
    (1000, 10000, 3, 16, 40, 1024, L2, True), (1000, 10000, 3, 16, 40, 1024, COS, True), (1000, 10000, 4, 16, 40, 1024, L2, False),
    (1000, 10000, 4, 16, 40, 1024, COS, False), (1000, 10000, 5, 16, 40, 1024, IP, False), (1000, 10000, 5, 16, 40, 1024, COS, False),
    (1000, 10000, 8, 16, 40, 1024, IP, True), (1000, 10000, 8, 16, 40, 1024, COS, True), (1000, 10000, 5, 16, 40, 1024, L2S, False),
    (1000, 10000, 5, 16, 40, 1024, COS, False), (1000, 10000, 8, 16, 40, 1024, L2S, True), (1000, 10000, 8, 16, 40, 1024, COS, True),
    # test dims that do not fit into kernel shared memory limits  :542-558 (2051 / InnerProduct is disabled there: cuvs issue 1091)
    (1000, 10000, 2048, 16, 40, 1024, L2, False), (1000, 10000, 2048, 16, 40, 1024, COS, False), (1000, 10000, 2049, 16, 40, 1024, L2, False),
    (1000, 10000, 2049, 16, 40, 1024, COS, False), (1000, 10000, 2050, 16, 40, 1024, IP, False), (1000, 10000, 2050, 16, 40, 1024, COS, False),
    (1000, 10000, 2051, 16, 40, 1024, COS, True), (1000, 10000, 2052, 16, 40, 1024, IP, False), (1000, 10000, 2052, 16, 40, 1024, COS, False),
    (1000, 10000, 2053, 16, 40, 1024, L2, True), (1000, 10000, 2053, 16, 40, 1024, COS, True), (1000, 10000, 2056, 16, 40, 1024, L2, True),
    (1000, 10000, 2056, 16, 40, 1024, COS, True),
    # various random combinations  :560-574
    (1000, 10000, 16, 10, 40, 1024, L2, False), (1000, 10000, 16, 10, 40, 1024, COS, False), (1000, 10000, 16, 10, 50, 1024, L2, False),
    (1000, 10000, 16, 10, 50, 1024, COS, False), (1000, 10000, 16, 10, 70, 1024, L2, False), (1000, 10000, 16, 10, 70, 1024, COS, False),
    (100, 10000, 16, 10, 20, 512, L2, False), (100, 10000, 16, 10, 20, 512, COS, False), (20, 100000, 16, 10, 20, 1024, L2, True),
    (20, 100000, 16, 10, 20, 1024, COS, True), (1000, 100000, 16, 10, 20, 1024, L2, True), (1000, 100000, 16, 10, 20, 1024, COS, True),
    (10000, 131072, 8, 10, 20, 1024, L2, False), (10000, 131072, 8, 10, 20, 1024, COS, False),
    # host input data  :576-590
    (1000, 10000, 16, 10, 40, 1024, L2, False, True), (1000, 10000, 16, 10, 40, 1024, COS, False, True), (1000, 10000, 16, 10, 50, 1024, L2, False, True),
    (1000, 10000, 16, 10, 50, 1024, COS, False, True), (1000, 10000, 16, 10, 70, 1024, L2, False, True), (1000, 10000, 16, 10, 70, 1024, COS, False, True),
    (100, 10000, 16, 10, 20, 512, L2, False, True), (100, 10000, 16, 10, 20, 512, COS, False, True), (20, 100000, 16, 10, 20, 1024, L2, False, True),
    (20, 100000, 16, 10, 20, 1024, COS, False, True), (1000, 100000, 16, 10, 20, 1024, L2, False, True), (1000, 100000, 16, 10, 20, 1024, COS, False, True),
    (10000, 131072, 8, 10, 20, 1024, L2, False, True), (10000, 131072, 8, 10, 20, 1024, COS, False, True),
    # host input data with prefetching for kernel copy overlapping  :592-606 (the C ABI has one host-input path: same call)
    (1000, 10000, 16, 10, 40, 1024, L2, False, True, True), (1000, 10000, 16, 10, 40, 1024, COS, False, True, True),
    (1000, 10000, 16, 10, 50, 1024, L2, False, True, True), (1000, 10000, 16, 10, 50, 1024, COS, False, True, True),
    (1000, 10000, 16, 10, 70, 1024, L2, False, True, True), (1000, 10000, 16, 10, 70, 1024, COS, False, True, True),
    (100, 10000, 16, 10, 20, 512, L2, False, True, True), (100, 10000, 16, 10, 20, 512, COS, False, True, True),
    (20, 100000, 16, 10, 20, 1024, L2, False, True, True), (20, 100000, 16, 10, 20, 1024, COS, False, True, True),
    (1000, 100000, 16, 10, 20, 1024, L2, False, True, True), (1000, 100000, 16, 10, 20, 1024, COS, False, True, True),
    (10000, 131072, 8, 10, 20, 1024, L2, False, True, True), (10000, 131072, 8, 10, 20, 1024, COS, False, True, True),
    # :608-621
    (1000, 10000, 16, 10, 40, 1024, IP, True), (1000, 10000, 16, 10, 40, 1024, COS, True), (1000, 10000, 16, 10, 50, 1024, IP, True),
    (1000, 10000, 16, 10, 50, 1024, COS, True), (1000, 10000, 16, 10, 70, 1024, IP, False), (1000, 10000, 16, 10, 70, 1024, COS, False),
    (100, 10000, 16, 10, 20, 512, IP, True), (100, 10000, 16, 10, 20, 512, COS, True), (20, 100000, 16, 10, 20, 1024, IP, True),
    (20, 100000, 16, 10, 20, 1024, COS, True), (1000, 100000, 16, 10, 20, 1024, IP, False), (1000, 100000, 16, 10, 20, 1024, COS, False),
    (10000, 131072, 8, 10, 50, 1024, IP, True), (10000, 131072, 8, 10, 50, 1024, COS, True),
    # :623-624
    (1000, 10000, 4096, 20, 50, 1024, IP, False), (1000, 10000, 4096, 20, 50, 1024, COS, False),
    # test splitting the big query batches (> max gridDim.y) into smaller batches  :626-632
    (100000, 1024, 32, 10, 64, 64, IP, False), (100000, 1024, 32, 10, 64, 64, COS, False), (1000000, 1024, 32, 10, 256, 256, IP, False),
    (1000000, 1024, 32, 10, 256, 256, COS, False), (98306, 1024, 32, 10, 64, 64, IP, True), (98306, 1024, 32, 10, 64, 64, COS, True),
    # test radix_sort for getting the cluster selection  :634-655
    (1000, 10000, 16, 10, KMAX * 2, KMAX * 4, L2, False), (1000, 10000, 16, 10, KMAX * 4, KMAX * 4, IP, False),
    (1000, 10000, 16, 10, KMAX * 4, KMAX * 4, COS, False),
    # "The following two test cases should show very similar recall."  :657-660
    (20000, 8712, 3, 10, 51, 66, L2, False), (100000, 8712, 3, 10, 51, 66, L2, False),
]

# ---------------------------------------------------------------------------------------------------------------- CAGRA
# cpp/tests/neighbors/ann_cagra.cuh:1416-1470 generate_inputs(), first three products; AnnCagraInputs field order :255-278:
# n_queries, n_rows, dim, k, build_algo, algo, max_queries, team_size, itopk_size, search_width, metric, host_dataset,
# include_serialized_dataset, use_source_indices, min_recall. BitwiseHamming / L1 rows are listed and skipped: those metrics
# are outside this repo's scope.
CAGRA_METRICS_ALL = [L2, IP, "bitwise_hamming", COS, "l1"]


def cagra_cases():
    out = []
    # :1420-1443  product 1
    for dim in (1, 16):
        for build in ("ivf_pq", "nn_descent"):
            for algo in ("single_cta", "multi_cta", "multi_kernel"):
                for max_queries in (0, 10):
                    for metric in CAGRA_METRICS_ALL:
                        for use_src in (True, False):
                            out.append(dict(table="product1", n_queries=100, n_rows=1000, dim=dim, k=16, build_algo=build, algo=algo,
                                            max_queries=max_queries, team_size=0, itopk_size=256, search_width=1, metric=metric,
                                            host_dataset=False, include_serialized_dataset=True, use_source_indices=use_src,
                                            min_recall=0.995))
    # :1445-1466  product 2 (MERGE_STRATEGY_LOGICAL: a merge option, not a search parameter)
    for dim in (1, 16):
        for metric in CAGRA_METRICS_ALL:
            out.append(dict(table="product2", n_queries=100, n_rows=1000, dim=dim, k=16, build_algo="nn_descent", algo="multi_cta",
                            max_queries=10, team_size=0, itopk_size=256, search_width=1, metric=metric, host_dataset=False,
                            include_serialized_dataset=True, use_source_indices=False, min_recall=0.995))
    # :1468-1490  "Additional distances tested with a single search algo."
    for nq in (1, 100):
        for k in (1, 16):
            for metric in (IP, "bitwise_hamming", COS, "l1"):
                out.append(dict(table="product3", n_queries=nq, n_rows=1000, dim=8, k=k, build_algo="nn_descent", algo="single_cta",
                                max_queries=0, team_size=0, itopk_size=256, search_width=1, metric=metric, host_dataset=False,
                                include_serialized_dataset=True, use_source_indices=False, min_recall=0.995))
    return out


# ---------------------------------------------------------------------------------------------------------- brute force
# cpp/tests/neighbors/ann_brute_force.cuh:167-212: {num_queries, num_db_vecs, dim, k, metric}; pass = devArrMatchKnnPair with
# eps 0.001 after sorting both sides (knn_utils.cuh:19-70: position by position, id OR distance), before and after a
# serialize / deserialize round trip (:106-127); data uniform [0.1, 2.0) (:131-145)
L2U = "l2_unexpanded"
BRUTE_FORCE_CASES = [
    # test various dims (aligned and not aligned to vector sizes)
    (1000, 10000, 1, 16, L2), (1000, 10000, 2, 16, L2), (1000, 10000, 3, 16, L2), (1000, 10000, 4, 16, L2), (1000, 10000, 5, 16, IP),
    (1000, 10000, 8, 16, IP), (1000, 10000, 5, 16, L2S), (1000, 10000, 8, 16, L2S),
    # test dims that do not fit into kernel shared memory limits
    (1000, 10000, 2048, 16, L2), (1000, 10000, 2049, 16, L2), (1000, 10000, 2050, 16, IP), (1000, 10000, 2051, 16, IP),
    (1000, 10000, 2052, 16, IP), (1000, 10000, 2053, 16, L2), (1000, 10000, 2056, 16, L2),
    # test fused_l2_knn
    (100, 1000, 16, 10, L2), (256, 256, 30, 10, L2), (1000, 10000, 16, 10, L2), (100, 1000, 16, 50, L2), (20, 10000, 16, 10, L2),
    (1000, 10000, 16, 50, L2), (1000, 10000, 32, 50, L2), (10000, 40000, 32, 30, L2), (100, 1000, 16, 10, L2U), (1000, 10000, 16, 10, L2U),
    (100, 1000, 16, 50, L2U), (20, 10000, 16, 50, L2U), (1000, 10000, 16, 50, L2U), (1000, 10000, 32, 50, L2U), (10000, 40000, 32, 30, L2U),
    # test tile
    (256, 512, 16, 8, L2), (256, 512, 16, 8, L2U), (256, 512, 16, 8, IP), (256, 512, 16, 8, L2S), (10000, 40000, 32, 30, L2),
    (789, 20516, 64, 256, L2S), (4, 12, 32, 6, L2), (1, 40, 32, 30, L2), (1000, 500000, 128, 128, L2),
]

# --------------------------------------------------------------------------------------------------------------- refine
# cpp/tests/neighbors/refine.cu:93-101: product {n_queries 137} x {n_rows 1000} x {dim 16} x {k 1, 10, 33} x {k0 33} x {L2Expanded,
# InnerProduct} x {host_data false, true}, float and uint8 rows (:103-110); candidates = the k0 exact neighbours
# (refine_helper.cuh:60-90), min_recall 1 with eps 0.001 (refine.cu:72-81); float rows uniform [-10, 10), uint8 integers [1, 20)
REFINE_CASES = [(137, 1000, 16, k, 33, metric, host) for k in (1, 10, 33) for metric in (L2, IP) for host in (False, True)]

# ----------------------------------------------------------------------------------------------------------- NN-descent
# cpp/tests/neighbors/ann_nn_descent.cuh:466-477: product {n_rows 2000, 4000} x {dim 4, 16, 31, 64, 256, 1024} x {graph_degree 32,
# 64} x {BitwiseHamming, L2Expanded, L2SqrtExpanded, InnerProduct, CosineExpanded, L1} x {host_dataset false, true} x
# {min_recall 0.90}; float rows ~ N(0.1, 2.0) (:162-164); pass = eval_neighbours of the graph rows against the exact kNN graph
# (self included), eps 0.001 (:150-158). BitwiseHamming / L1 rows are listed and skipped (outside this repo's scope).
NN_DESCENT_CASES = [(n, d, deg, metric, host, 0.90) for n in (2000, 4000) for d in (4, 16, 31, 64, 256, 1024) for deg in (32, 64)
                    for metric in ("bitwise_hamming", L2, L2S, IP, COS, "l1") for host in (False, True)]
