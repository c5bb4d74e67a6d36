"""Known-answer vectors held by the reference's own tests (copied as DATA, with their location).

These pin the oracle (tests/test_oracle.py) and the HIP path (tests/test_*_gpu.py).
"""
import numpy as np

# c/tests/neighbors/ann_cagra_c.cu:31-50 — 4x2 dataset, 4 queries, k=1, L2Expanded (squared distances).
CAGRA_C_DATASET = np.array(
    [[0.74021935, 0.9209938], [0.03902049, 0.9689629], [0.92514056, 0.4463501], [0.6673192, 0.10993068]],
    dtype=np.float32,
)
CAGRA_C_QUERIES = np.array(
    [[0.48216683, 0.0428398], [0.5084142, 0.6545497], [0.51260436, 0.2643005], [0.05198065, 0.5789965]],
    dtype=np.float32,
)
CAGRA_C_NEIGHBORS = np.array([3, 0, 3, 1], dtype=np.int64)
CAGRA_C_DISTANCES = np.array([0.03878258, 0.12472608, 0.04776672, 0.15224178], dtype=np.float32)
# bitset 0b1001: rows 1 and 2 removed
CAGRA_C_FILTER_WORDS = np.array([0b1001], dtype=np.uint32)
CAGRA_C_NEIGHBORS_FILTERED = np.array([3, 0, 3, 0], dtype=np.int64)
CAGRA_C_DISTANCES_FILTERED = np.array([0.03878258, 0.12472608, 0.04776672, 0.59063464], dtype=np.float32)
CAGRA_C_TOL = 1e-3  # the reference compares with eps 0.001

# cpp/tests/neighbors/brute_force.cu:169-185 — 10 labelled 2-D points; the k=2 neighbours of every
# point (searching the set against itself) carry the point's own label.
BF_KAT_POINTS = np.array(
    [
        [2.7810836, 2.550537003],
        [1.465489372, 2.362125076],
        [3.396561688, 4.400293529],
        [1.38807019, 1.850220317],
        [3.06407232, 3.005305973],
        [7.627531214, 2.759262235],
        [5.332441248, 2.088626775],
        [6.922596716, 1.77106367],
        [8.675418651, -0.242068655],
        [7.673756466, 3.508563011],
    ],
    dtype=np.float32,
)
BF_KAT_K = 2
BF_KAT_LABELS = np.array([0, 0, 0, 0, 0, 1, 1, 1, 1, 1], dtype=np.int32)

# c/tests/cluster/kmeans_c.cu:24-48 — 8 points, 2 clusters, init = Array, max_iter 100, tol 1e-6; the reference
# compares centroids with 1e-4, labels exactly, inertia / predict inertia / cluster cost with 1e-4 (:157-167).
KMEANS_C_DATASET = np.array(
    [[1, 1], [1, 2], [2, 1], [2, 2], [10, 10], [10, 11], [11, 10], [11, 11]], dtype=np.float32
)
KMEANS_C_INIT_CENTROIDS = np.array([[0, 0], [12, 12]], dtype=np.float32)
KMEANS_C_CENTROIDS = np.array([[1.5, 1.5], [10.5, 10.5]], dtype=np.float32)
KMEANS_C_LABELS = np.array([0, 0, 0, 0, 1, 1, 1, 1], dtype=np.int32)
KMEANS_C_INERTIA = 4.0
KMEANS_C_TOL = 1e-4

# java/cuvs-java/src/test/java/com/nvidia/cuvs/BruteForceAndSearchIT.java:36-47 (same 4x2 data as the CAGRA C test),
# :93-100 — brute force L2Expanded, k = 3: {id: distance} per query (a Java Map: unordered), compared with
# |d - d_exp| < 1e-4 (CuVSTestCase.java:128).
BF_JAVA_K3 = [
    {3: 0.038782537, 2: 0.35904616, 0: 0.83774555},
    {0: 0.12472606, 2: 0.21700788, 1: 0.3191862},
    {3: 0.047766685, 2: 0.20332813, 0: 0.48305476},
    {1: 0.15224183, 0: 0.5906347, 3: 0.5986643},
]
# :109-122 — the same search with a bitset prefilter keeping rows {0, 1, 3} (one BitSet per query, all equal)
BF_JAVA_K3_FILTER_KEEP = [0, 1, 3]
BF_JAVA_K3_FILTERED = [
    {0: 0.83774555, 1: 1.0540828, 3: 0.038782537},
    {0: 0.12472606, 1: 0.3191862, 3: 0.32186073},
    {0: 0.48305476, 1: 0.7208309, 3: 0.047766685},
    {0: 0.5906347, 1: 0.15224195, 3: 0.5986643},
]
JAVA_TOL = 1e-4

# java/cuvs-java/src/test/java/com/nvidia/cuvs/CagraBuildAndSearchIT.java:104-109 — CAGRA on the same data, k = 3
CAGRA_JAVA_K3 = [
    {3: 0.038782578, 2: 0.3590463, 0: 0.83774555},
    {0: 0.12472608, 2: 0.21700792, 1: 0.31918612},
    {3: 0.047766715, 2: 0.20332818, 0: 0.48305473},
    {1: 0.15224178, 0: 0.59063464, 3: 0.5986642},
]

# Rust binding tests (property KATs on uniform [0,1) data; the queries are the first 4 dataset rows and every query
# must come back as its own nearest neighbour):
#   rust/cuvs/src/brute_force.rs:125-177   16 x 8,   k = 4, L2Expanded
#   rust/cuvs/src/ivf_pq/index.rs:101-151  1024 x 16, n_lists 64 (other params default), k = 10
#   rust/cuvs/src/ivf_flat/index.rs:110-...  1024 x 16, n_lists 64, k = 10
#   rust/cuvs/src/cagra/index.rs:302-312   default build params
RUST_SELF_NEIGHBOR_CASES = {
    "brute_force": dict(n=16, dim=8, k=4),
    "ivf_pq": dict(n=1024, dim=16, k=10, n_lists=64),
    "ivf_flat": dict(n=1024, dim=16, k=10, n_lists=64),
    "cagra": dict(n=256, dim=16, k=10),
}
