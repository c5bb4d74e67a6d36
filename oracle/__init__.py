"""numpy front-end of oracle/liboracle.so — the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under cuvs_amd/ imports this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METRICS = {"sqeuclidean": 0, "euclidean": 1, "l2": 1, "cosine": 2, "inner_product": 6}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make oracle/liboracle.so`")
        _LIB = C.CDLL(path)
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _metric(m):
    return METRICS[m] if isinstance(m, str) else int(m)


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n=0):
    """OpenMP threads of the oracle's loops (0: every online core) - launchers such as torchrun pin OMP_NUM_THREADS=1."""
    lib().oracle_set_num_threads(int(n))


def row_norms(x, sqrt=False):
    x = _f32(x)
    out = np.empty(x.shape[0], np.float32)
    lib().oracle_row_norms(_p(x), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]), _p(out), C.c_int(int(sqrt)))
    return out


def pairwise(q, x, metric="sqeuclidean", clamp_eps=1e-6):
    q, x = _f32(q), _f32(x)
    out = np.empty((q.shape[0], x.shape[0]), np.float32)
    lib().oracle_pairwise(_p(q), C.c_int64(q.shape[0]), _p(x), C.c_int64(x.shape[0]), C.c_int64(q.shape[1]),
                          C.c_int(_metric(metric)), C.c_float(clamp_eps), _p(out))
    return out


def select_k(vals, k, select_min=True, in_idx=None, idx_offset=0):
    vals = _f32(vals)
    rows, ln = vals.shape
    ov = np.empty((rows, k), np.float32)
    oi = np.empty((rows, k), np.int64)
    ii = None if in_idx is None else np.ascontiguousarray(in_idx, dtype=np.int64)
    lib().oracle_select_k(_p(vals), _p(ii) if ii is not None else None, C.c_int64(rows), C.c_int64(ln), C.c_int(k),
                          C.c_int(int(select_min)), C.c_int64(idx_offset), _p(ov), _p(oi))
    return ov, oi


def brute_force_knn(q, x, k, metric="sqeuclidean", clamp_eps=1e-6, keep_bits=None, bitmap=False):
    """Bit-for-bit twin of cuvsBruteForceSearch. Returns (distances, indices)."""
    q, x = _f32(q), _f32(x)
    oi = np.empty((q.shape[0], k), np.int64)
    od = np.empty((q.shape[0], k), np.float32)
    kb = None if keep_bits is None else np.ascontiguousarray(keep_bits, dtype=np.uint32)
    lib().oracle_brute_force_knn(_p(q), C.c_int64(q.shape[0]), _p(x), C.c_int64(x.shape[0]), C.c_int64(q.shape[1]),
                                 C.c_int(k), C.c_int(_metric(metric)), C.c_float(clamp_eps),
                                 _p(kb) if kb is not None else None, C.c_int(int(bitmap)), _p(oi), _p(od))
    return od, oi


def exact_knn(q, x, k, metric="sqeuclidean"):
    """The reference's CPU path (refine_host with every row as candidate). Returns (distances, indices)."""
    q, x = _f32(q), _f32(x)
    oi = np.empty((q.shape[0], k), np.int64)
    od = np.empty((q.shape[0], k), np.float32)
    lib().oracle_exact_knn(_p(q), C.c_int64(q.shape[0]), _p(x), C.c_int64(x.shape[0]), C.c_int64(q.shape[1]),
                           C.c_int(k), C.c_int(_metric(metric)), _p(oi), _p(od))
    return od, oi


def recall(found, truth):
    """Fraction of true neighbours found (reference: python/cuvs/cuvs/tests/ann_utils.py:24-30)."""
    found, truth = np.asarray(found), np.asarray(truth)
    hits = 0
    for f, t in zip(found, truth):
        hits += len(np.intersect1d(f, t))
    return hits / truth.size


_LUT_MODES = {"f32": 0, "f16": 1, "fp8": 2}


_COARSE_MODES = {"f32": 0, "f16": 1, "i8": 2}


def ivf_pq_search(exported, queries, k, n_probes, metric="sqeuclidean", scale=1.0, lut="f32", acc="f32", coarse="f32",
                  keep_bits=None):
    """Search an index exported with cuvs_amd.neighbors.ivf_pq.export_for_oracle. lut: "f32" | "f16" | "fp8" (the
    reference's fp_8bit<5, signed-for-inner-product>), acc: "f32" | "f16" (search_params.lut_dtype /
    internal_distance_dtype), coarse: "f32" | "f16" | "i8" (coarse_search_dtype: coarse search and query rotation in
    that type), keep_bits: optional uint32 words of a bitset over source ids (1 keeps the row).
    Returns (distances, neighbors); bit-for-bit twin of cuvsIvfPqSearch."""
    q = _f32(queries)
    centers = _f32(exported["centers"])
    centers_rot = _f32(exported["centers_rot"])
    rotation = _f32(exported["rotation"])
    pqc = _f32(exported["pq_centers"])
    sizes = np.ascontiguousarray(exported["list_sizes"], dtype=np.uint32)
    start = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(sizes, out=start[1:])
    codes = np.ascontiguousarray(np.concatenate(exported["codes"], axis=0), dtype=np.uint8)
    ids = np.ascontiguousarray(np.concatenate(exported["ids"], axis=0), dtype=np.int64)
    nq = q.shape[0]
    nb = np.empty((nq, k), np.int64)
    ds = np.empty((nq, k), np.float32)
    kb = None if keep_bits is None else np.ascontiguousarray(keep_bits, dtype=np.uint32)
    lib().oracle_ivf_pq_search(
        _p(q), C.c_int64(nq), C.c_int(q.shape[1]), _p(centers), _p(centers_rot), _p(rotation), _p(pqc),
        C.c_int(len(sizes)), C.c_int(rotation.shape[0]), C.c_int(int(exported["pq_dim"])),
        C.c_int(int(exported["pq_len"])), C.c_int(int(exported["pq_bits"])), _p(sizes), _p(start), _p(codes), _p(ids),
        C.c_int(_metric(metric)), C.c_int(n_probes), C.c_int(k), C.c_float(scale), _p(nb), _p(ds),
        C.c_int(int(bool(exported.get("per_cluster", False)))), C.c_int(_LUT_MODES[lut]), C.c_int(_LUT_MODES[acc]),
        C.c_int(_COARSE_MODES[coarse]),
        _p(kb) if kb is not None else None)
    return ds, nb


def fp8_round_trip(values, signed=False, to_half=False):
    """decode(encode(v)) of the reference's fp_8bit<5, signed> (ivf_pq_fp_8bit.cuh:32-100), element-wise."""
    L = lib()
    L.oracle_fp8_encode.restype = C.c_uint8
    L.oracle_fp8_decode_f32.restype = C.c_float
    L.oracle_fp8_decode_f16.restype = C.c_float
    dec = L.oracle_fp8_decode_f16 if to_half else L.oracle_fp8_decode_f32
    v = np.asarray(values, np.float32)
    out = np.array([dec(C.c_uint8(L.oracle_fp8_encode(C.c_float(float(x)), C.c_int(int(signed)))), C.c_int(int(signed)))
                    for x in v.ravel()], np.float32)
    return out.reshape(v.shape)


def pq_encode(resid, pq_centers, pq_bits):
    resid = _f32(resid)
    pqc = _f32(pq_centers)
    pq_dim, pq_len, _ = pqc.shape
    out = np.empty((resid.shape[0], pq_dim), np.uint8)
    lib().oracle_pq_encode(_p(resid), C.c_int64(resid.shape[0]), C.c_int(resid.shape[1]), _p(pqc), C.c_int(pq_dim),
                           C.c_int(pq_len), C.c_int(pq_bits), _p(out))
    return out


def refine(dataset, queries, candidates, k, metric="sqeuclidean"):
    """CPU twin of cuvsRefine. Returns (distances, indices)."""
    x, q = _f32(dataset), _f32(queries)
    cand = np.ascontiguousarray(candidates, dtype=np.int64)
    oi = np.empty((q.shape[0], k), np.int64)
    od = np.empty((q.shape[0], k), np.float32)
    lib().oracle_refine(_p(x), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]), _p(q), C.c_int64(q.shape[0]), _p(cand),
                        C.c_int(cand.shape[1]), C.c_int(k), C.c_int(_metric(metric)), _p(oi), _p(od))
    return od, oi


def ivf_flat_search(exported, queries, k, n_probes, metric="sqeuclidean", coarse_scale=1.0):
    """Search an index exported with cuvs_amd.neighbors.ivf_flat.export_for_oracle. `queries` in the index dtype;
    coarse_scale = 1/128 (int8) or 1/256 (uint8): the coarse quantizer sees mapped floats (ann_utils.cuh:134-196)."""
    int_mode = int(np.asarray(queries).dtype in (np.int8, np.uint8) and metric != "cosine")
    q_raw = _f32(np.asarray(queries).astype(np.float32))
    q_coarse = _f32(q_raw * np.float32(coarse_scale))
    centers = _f32(exported["centers"])
    sizes = np.ascontiguousarray(exported["list_sizes"], dtype=np.uint32)
    start = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(sizes, out=start[1:])
    rows = _f32(np.concatenate([r.astype(np.float32) for r in exported["rows"]], axis=0))
    ids = np.ascontiguousarray(np.concatenate(exported["ids"], axis=0), dtype=np.int64)
    nq = q_raw.shape[0]
    nb = np.empty((nq, k), np.int64)
    ds = np.empty((nq, k), np.float32)
    lib().oracle_ivf_flat_search(_p(q_coarse), _p(q_raw), C.c_int64(nq), C.c_int(q_raw.shape[1]), _p(centers),
                                 C.c_int(len(sizes)), _p(sizes), _p(start), _p(rows), _p(ids), C.c_int(_metric(metric)),
                                 C.c_int(n_probes), C.c_int(k), C.c_int(int_mode), _p(nb), _p(ds))
    return ds, nb


def cagra_search(dataset, graph, queries, k, itopk_size=64, search_width=1, max_iterations=0, min_iterations=0,
                 hashmap_min_bitlen=0, rand_xor_mask=0x128394, metric="sqeuclidean", filter_words=None, num_random_samplings=1):
    """CPU twin of cuvsCagraSearch (same parameter derivation as cuvs_amd/csrc/cagra.hip: search_plan.cuh:199-245).
    dataset/queries in the index dtype; graph uint32 [n, degree]. Returns (distances, neighbors)."""
    raw = np.asarray(dataset)
    vl = 16 // raw.dtype.itemsize
    x, q = _f32(raw.astype(np.float32)), _f32(np.asarray(queries).astype(np.float32))
    g = np.ascontiguousarray(graph, dtype=np.uint32)
    n, degree = g.shape
    width = max(1, int(search_width))
    itopk = max(int(itopk_size) if itopk_size else 64, k)
    if itopk % 32:
        itopk += 32 - itopk % 32
    max_iter = int(max_iterations)
    if max_iter == 0:
        max_iter = itopk // width
        reach = 1
        while reach < n:
            reach *= max(2, degree // 2)
            max_iter += 1
    if int(max_iterations) < int(min_iterations):  # the reference tests the original field (search_plan.cuh:216)
        max_iter = int(min_iterations)
    bits = 11
    while (1 << bits) < 2 * (itopk + 2 * width * degree):
        bits += 1
    bits = max(bits, int(hashmap_min_bitlen))
    reset = max(1, ((1 << bits) // 2 - itopk) // (width * degree))
    nq = q.shape[0]
    oi = np.empty((nq, k), np.int64)
    od = np.empty((nq, k), np.float32)
    fw = None if filter_words is None else np.ascontiguousarray(filter_words, dtype=np.uint32)
    lib().oracle_cagra_search(_p(x), C.c_int64(n), C.c_int64(x.shape[1]), C.c_int(vl), _p(g), C.c_int(degree), _p(q),
                              C.c_int64(nq), C.c_int(k), C.c_int(itopk), C.c_int(width), C.c_int(max_iter),
                              C.c_int(int(min_iterations)), C.c_int(bits), C.c_int(reset), C.c_uint64(rand_xor_mask),
                              C.c_int({6: 1, 2: 2}.get(_metric(metric), 0)), _p(fw) if fw is not None else None, _p(oi), _p(od),
                              C.c_int(max(1, int(num_random_samplings))))
    return od, oi


def cagra_optimize(knn_graph, graph_degree, guarantee_connectivity=False):
    """CPU twin of graph::optimize (oracle_cagra_optimize.c). Returns (graph [n, degree] uint32, components left by the
    connectivity pass - 0 when it was not asked for)."""
    g = np.ascontiguousarray(knn_graph, dtype=np.uint32)
    n, K = g.shape
    out = np.empty((n, graph_degree), np.uint32)
    fn = lib().oracle_cagra_optimize
    fn.restype = C.c_int64
    left = fn(_p(g), C.c_int64(n), C.c_uint32(K), C.c_uint32(graph_degree), C.c_int(int(guarantee_connectivity)), _p(out))
    return out, int(left)


def cagra_mst(knn_graph, graph_degree):
    """The spanning forest alone: (rows [n, degree] front-packed with 0xffffffff fillers, counts [n], components left)."""
    g = np.ascontiguousarray(knn_graph, dtype=np.uint32)
    n, K = g.shape
    mst = np.empty((n, graph_degree), np.uint32)
    cnt = np.empty(n, np.uint32)
    fn = lib().oracle_cagra_mst
    fn.restype = C.c_int64
    left = fn(_p(g), C.c_int64(n), C.c_uint32(K), C.c_uint32(graph_degree), _p(mst), _p(cnt))
    return mst, cnt, int(left)


def kmeans_balanced_fit(x, n_clusters, n_iters=20, hierarchical=True):
    """CPU twin of the balanced k-means (cuvs_amd/csrc/kmeans_balanced.hip). Returns (centers, labels)."""
    x = _f32(x)
    n, dim = x.shape
    centers = np.empty((n_clusters, dim), np.float32)
    labels = np.empty(n, np.uint32)
    lib().oracle_kmeans_balanced_fit(_p(x), C.c_int64(n), C.c_int(dim), C.c_int(n_clusters), C.c_int(n_iters),
                                     C.c_int(int(hierarchical)), _p(centers), _p(labels))
    return centers, labels


def kmeans_lloyd(x, init_centroids, max_iter=300, tol=1e-4, sample_weights=None):
    """Lloyd iterations as the reference runs them (cpp/src/cluster/detail/kmeans.cuh:813-925): per iteration the
    cost against the CURRENT centroids, weighted means (a cluster with zero weight keeps its centroid,
    kmeans_common.cuh:585-600), squared centroid shift, then the stopping rule of kmeans_common.cuh:629-648 evaluated in
    float32 like the reference's DataT. Weights are rescaled to sum to n (kmeans.cuh:713-726). Arithmetic is float64
    here: the HIP path is compared within a stated tolerance, not bit for bit.
    Returns (centroids float32 [k, d], labels int32 [n], inertia, n_iter)."""
    x64 = np.asarray(x, dtype=np.float64)
    cur = np.asarray(init_centroids, dtype=np.float64).copy()
    n, k = x64.shape[0], cur.shape[0]
    w = np.ones(n) if sample_weights is None else np.asarray(sample_weights, dtype=np.float64) * n / np.sum(sample_weights)

    def assign(c):
        d = ((x64 * x64).sum(1)[:, None] - 2.0 * x64 @ c.T) + (c * c).sum(1)[None, :]
        lab = d.argmin(1)
        return lab, ((x64 - c[lab]) ** 2).sum(1)

    prior, ran = 0.0, 0
    for it in range(1, max_iter + 1):
        lab, dist = assign(cur)
        cost = float((w * dist).sum())
        nxt = cur.copy()
        for c in range(k):
            m = lab == c
            ws = w[m].sum()
            if ws > 0:
                nxt[c] = (w[m, None] * x64[m]).sum(0) / ws
        shift = float(((nxt - cur) ** 2).sum())
        cur, ran = nxt, it
        done = cost != 0.0 and it > 1 and np.float32(cost / prior) > np.float32(1.0) - np.float32(tol)
        done = done or np.float32(shift) < np.float32(tol)
        prior = cost
        if done:
            break
    lab, dist = assign(cur)
    return cur.astype(np.float32), lab.astype(np.int32), float((w * dist).sum()), ran
