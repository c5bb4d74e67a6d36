// Internal (C++) interfaces between the hot-path translation units. Everything is asynchronous
// on res.stream; pointers are device pointers unless stated otherwise.
#pragma once
#include "common.hpp"

namespace cuvs_amd {

// ---------------------------------------------------------------- select_k.hip
// Batched exact top-k, one row per query (reference call sites: knn_brute_force.cuh:267,309;
// ivf_flat_search.cuh:180,283; ivf_pq_search.cuh:160,620 -> raft::matrix::select_k).
// in      [rows, len] with row pitch `in_ld` floats
// in_idx  optional [rows, len] (same pitch); nullptr => index = column + idx_offset
// out_val [rows, k], out_idx [rows, k]; sorted ascending by (value, index) for select_min,
// descending by value (ascending index among equals) for !select_min.
// Tie rule at the k-th boundary: the earliest columns win. Missing entries (len < k) are padded
// with +/-FLT_MAX and index -1.
template <typename InIdxT, typename OutIdxT>
void select_k(resources& res, const float* in, const InIdxT* in_idx, int64_t rows, int64_t len,
              int64_t in_ld, int k, float* out_val, OutIdxT* out_idx, bool select_min,
              int64_t idx_offset = 0, int64_t out_ld = -1, int64_t out_col_offset = 0, const uint32_t* run_if = nullptr);
// (run_if: optional device word - the kernels return at once when it is zero: a fallback pass decided on the device)

// ---------------------------------------------------------------- distance.hip
// Canonical squared row norms: 64 strided fmaf partials + fixed butterfly (oracle/oracle.c
// `canon_sqnorm`). sqrt_out => writes sqrt of that (cosine).
template <typename T>
void row_norms(resources& res, const T* x, int64_t n, int64_t dim, int64_t ld, float* out, bool sqrt_out);

// In-place row normalisation of a dense fp32 matrix with the canonical norm (cosine paths).
void normalize_rows(resources& res, float* x, int64_t n, int64_t dim);

// D[i, j] = epilogue(dot(Q_i, X_j)) for i < m, j < n; out pitch ldo.
//   metric L2*: max(0, fmaf(-2, dot, qn_i + xn_j)) with the reference's self-neighbour clamp
//               (distance_ops/l2_exp.cuh:36-50,113-125); sqrt variants take sqrt.
//   metric IP : dot ; metric cosine: 1 - dot / (qn_i * xn_j) (norms are sqrt norms).
// dot is one k-ordered fp32 fma chain (MFMA 16x16x4 f32), so results are reproducible bit for bit.
template <typename TQ, typename TX>
void pairwise_distance(resources& res, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n,
                       int64_t ldx, int64_t dim, const float* qn, const float* xn, int metric,
                       float* out, int64_t ldo, const uint32_t* run_if = nullptr);
// (run_if: a device word - the launch is a no-op while it is zero: fallback passes enqueued without a host round trip)

// Distance matrix in GROUPED layout + the best key of every 16-column group, and the top-k that reads them (the coarse
// search of the IVF indexes: k = n_probes of n_lists per query). pairwise_distance_grouped writes row i in a fixed
// permutation inside every 128-column tile (position p holds column grouped_col(p)) so that a lane's four values are one
// 16-byte store and 16 consecutive positions form a group; gkeys[i, p / 16] = the group's best order-preserving key.
// select_k_grouped = select_k on the same values: the k-th best of a thread's group keys bounds the row's k-th best value,
// only groups at or below the bound are read (~1.4 k of 16384 values at k = 128), sorted by (value, column). Results
// are those of pairwise_distance + select_k, ties included. pairwise_distance_grouped returns false (nothing launched) for
// shapes the tile kernel does not take; select_k_grouped handles every row (rows with masses of ties are re-laid in
// column order and go through the radix kernel).
bool pairwise_distance_grouped(resources& res, const float* q, int64_t m, int64_t ldq, const float* x, int64_t n, int64_t ldx,
                               int64_t dim, const float* qn, const float* xn, int metric, float* out, int64_t ldo, uint32_t* gkeys,
                               int64_t ldg);
bool select_k_grouped_ok(int64_t len, int k);  // the shapes select_k_grouped takes (len >= 4096, 8 <= k <= 256, 8 k <= len)
void select_k_grouped(resources& res, float* in, int64_t in_ld, const uint32_t* gkeys, int64_t ldg, int64_t rows, int64_t len, int k,
                      float* out_val, uint32_t* out_idx, bool select_min);
__host__ __device__ inline uint32_t grouped_col(uint32_t p)  // column held at position p of a grouped row
{
  return (p & ~63u) | ((p & 3u) << 4) | ((p >> 2) & 15u);
}

// The same distance tile, kept in registers: every element (i, j) that beats row i's current k-th value
// buf_v[i * (k + cap) + k - 1] (strictly) and passes the pre-filter is appended - value and source id col_off + j -
// at buf[i * (k + cap) + k + cnt[i]++] (atomic: the order within a row is arbitrary; cnt may run past cap).
template <typename TQ, typename TX>
void pairwise_threshold_append(resources& res, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n, int64_t ldx,
                               int64_t dim, const float* qn, const float* xn, int metric, float* buf_v, int64_t* buf_i,
                               int* cnt, int k, int cap, int64_t col_off, int64_t row_off, int64_t n_total,
                               const uint32_t* bits, int filter_type, int64_t col_stride = 1, const float* thr = nullptr);
// (col_stride: the source id of column j is col_off + j * col_stride - a strided view of a larger matrix; thr: [m] thresholds
// taken instead of the rows' current k-th values, e.g. the next float up so that ties at the k-th value are appended too)

// Reduced-precision coarse search (coarse_lowp.hip): rows packed as fp16 / int8 K steps of 32 bytes (zero-padded), the
// query x centre products on the fp16 / int8 matrix cores with the reference's output arithmetic in the epilogue
// (ivf_pq_search.cuh:171-340): half: out = half(alpha (dot - term / 2)) (term nullptr: half(alpha dot)); int8: alpha (dot + term)
int coarse_lowp_ksteps(bool i8, int64_t cols);  // 32-byte K steps of a packed row
void coarse_lowp_pack(resources& res, bool i8, const float* in, int64_t n, int64_t ld_in, int64_t cols, void* out);
void coarse_lowp_distances(resources& res, bool i8, const void* q_pack, int64_t nq, const void* c_pack, int64_t n, int64_t cols,
                           const float* term, float alpha, float* out, int64_t ldo);

// labels[i] = argmin_j ( xn_j - 2 dot(Q_i, X_j) ) (ties -> smallest j); optional min value out
// (= squared L2 distance minus |q|^2). The k-means E-step and IVF list assignment.
template <typename TQ>
void fused_l2_argmin(resources& res, const TQ* q, int64_t m, int64_t ldq, const float* centers,
                     int64_t n, int64_t dim, const float* center_norms, uint32_t* labels,
                     float* min_val);

// ---------------------------------------------------------------- fused_knn.hip
// Fused distance + top-k (k <= 64): same results as pairwise_distance + select_k, the tile never leaves the CU.
// Returns false when the shape is outside the fused path.
template <typename TQ, typename TX>
bool fused_knn(resources& res, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n, int64_t ldx,
               int64_t dim, const float* qn, const float* xn, int metric, int k, float* out_d, int64_t* out_i);

// ---------------------------------------------------------------- kmeans_balanced.hip
struct kmeans_params {
  int n_iters        = 20;
  bool hierarchical  = true;
  // false: L2 (argmin of |x - c|^2). true: inner product - rows go to the centre with the largest dot product and the
  // centres are normalised before every E-step so that they do not collapse towards zero
  // (kmeans_balanced.cuh:123-150 predict_core, :682-696)
  bool inner_product = false;
};
// Balanced k-means fit on float rows (reference: cluster/detail/kmeans_balanced.cuh:986-1148).
void kmeans_balanced_fit(resources& res, const float* x, int64_t n, int64_t dim, int n_clusters,
                         const kmeans_params& p, float* centers /*[n_clusters, dim]*/);
// Non-hierarchical balanced EM (kmeans_balanced.cuh:724-783); x has row pitch ld. Used for PQ codebooks.
void kmeans_build_clusters(resources& res, const float* x, int64_t n_rows, int64_t ld, int dim, int n_clusters,
                           int n_iters, float* centers, uint32_t* labels, uint32_t* sizes, bool inner_product = false);
// Stable grouping of rows by label: perm[n] = row ids ordered by (label, row); offsets[n_clusters + 1].
void group_by_label(resources& res, const uint32_t* labels, int64_t n, uint32_t n_clusters, uint32_t* perm,
                    uint32_t* offsets);
// labels for arbitrary element type
template <typename T>
void kmeans_predict(resources& res, const T* x, int64_t n, int64_t dim, const float* centers,
                    int n_clusters, uint32_t* labels);

// ---------------------------------------------------------------- nn_descent.hip
// kNN graph [n, K] (uint32 ids sorted by distance, self excluded, 0xffffffff = none) by NN-descent; `norms` = canonical
// |x| per row for the cosine metric (else unused).
// keys_out (optional, [n, K]): order-preserving keys of the distances (device_utils.hpp key_to_float; inner product
// negated). termination_threshold: stop when fewer than this share of the n*K slots changed in a round.
void knn_graph_nn_descent(resources& res, const void* data, elem_t et, int64_t n, int64_t dim, uint32_t K, int metric,
                          const float* norms, int n_iters, uint32_t* knn, uint32_t* keys_out = nullptr,
                          float termination_threshold = 1e-4f);

// ---------------------------------------------------------------- cagra_mst.hip
// graph::optimize's guarantee_connectivity pass: protected spanning-forest edges per node (front-packed rows of `mst`
// [n, degree], counts in mst_cnt [n]); returns the number of connected components left (1 on success).
int64_t cagra_mst_optimize(resources& res, const uint32_t* knn, int64_t n, uint32_t K, uint32_t degree, uint32_t* mst,
                           uint32_t* mst_cnt);

// ---------------------------------------------------------------- refine.hip
// exact re-ranking of candidate ids; out sorted by (distance, id). All pointers device.
void refine(resources& res, const void* data, elem_t et, int64_t n, int64_t dim, const void* queries, int64_t m,
            const int64_t* cand, int n_cand, int k, int metric, int64_t* out_i, float* out_d);

// ---------------------------------------------------------------- mg.hip helpers (defined next to the index structs)
void ivf_flat_index_info(uintptr_t addr, int64_t* size, int* metric);
void cagra_index_info(uintptr_t addr, int64_t* size, int* metric);

}  // namespace cuvs_amd
