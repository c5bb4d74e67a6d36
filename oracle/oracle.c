/*
 * oracle.c — CPU restatement of the reference's hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (cuvs_amd/libcuvs_c.so) never links or calls it.
 *
 * Parity pinning: the reference's kernels are CUDA + RAFT (not vendored; rapidsai/raft 26.08) and cannot
 * be compiled here (no nvcc, no GPU, RAFT absent), so this file restates the algorithms from the files
 * cited per function and is pinned against the reference's own fixtures:
 *   - CAGRA C golden vectors (c/tests/neighbors/ann_cagra_c.cu:31-50)          tests/test_oracle.py
 *   - brute-force label KAT (cpp/tests/neighbors/brute_force.cu:169-185)        tests/test_oracle.py
 *   - scipy cdist / sklearn brute kNN, the reference's Python oracles
 *     (python/cuvs/cuvs/tests/test_brute_force.py:88-103, ann_utils.py:24-30)   tests/test_oracle.py
 * RAFT's select_k tie order is "parity unpinned" (no in-tree test pins it); we fix the rule
 * "(value, position) lexicographic" here and in the HIP kernel.
 *
 * Arithmetic conventions shared bit-for-bit with the HIP kernels (compiled with -ffp-contract=off so
 * only the explicit fmaf calls fuse):
 *   canon_dot    : acc = fmaf(a[k], b[k], acc) for k = 0..d-1  (== v_mfma_f32_16x16x4_f32 k-ordered chain)
 *   canon_sqnorm : 64 strided fmaf partials, then butterfly p[i] += p[i+off], off = 32..1
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

enum { M_L2Expanded = 0, M_L2SqrtExpanded = 1, M_CosineExpanded = 2, M_L2Unexpanded = 4,
       M_L2SqrtUnexpanded = 5, M_InnerProduct = 6 };

/* ------------------------------------------------------------------ canonical arithmetic */
static float canon_dot(const float* a, const float* b, int64_t d)
{
  float acc = 0.f;
  for (int64_t k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
  return acc;
}

static float canon_sqnorm(const float* a, int64_t d)
{
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.f;
  for (int64_t j = 0; j < d; ++j) p[j & 63] = fmaf(a[j], a[j], p[j & 63]);
  for (int off = 32; off > 0; off >>= 1)
    for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
  return p[0];
}

EXPORT void oracle_row_norms(const float* x, int64_t n, int64_t d, float* out, int sqrt_out)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float v = canon_sqnorm(x + i * d, d);
    out[i]  = sqrt_out ? sqrtf(v) : v;
  }
}

/* epilogue of the expanded-form distances: cuvs_amd/csrc/distance.hip finish_distance;
 * reference knn_brute_force.cuh:204-232, distance_ops/l2_exp.cuh:36-50,113-125 */
static float finish_distance(float dot, float qn, float xn, int metric, float clamp_eps)
{
  if (metric == M_InnerProduct) return dot;
  if (metric == M_CosineExpanded) return 1.0f - dot / (qn * xn);
  float val = fmaf(-2.0f, dot, qn + xn);
  if (val * val < clamp_eps && qn == xn) val = 0.f;
  val = val > 0.f ? val : 0.f;
  if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) val = sqrtf(val);
  return val;
}

/* D[m,n] canonical distance matrix (small sizes) */
EXPORT void oracle_pairwise(const float* q, int64_t m, const float* x, int64_t n, int64_t d, int metric,
                            float clamp_eps, float* out)
{
  float* qn = (float*)malloc(sizeof(float) * (size_t)m);
  float* xn = (float*)malloc(sizeof(float) * (size_t)n);
  oracle_row_norms(q, m, d, qn, metric == M_CosineExpanded);
  oracle_row_norms(x, n, d, xn, metric == M_CosineExpanded);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j)
      out[i * n + j] = finish_distance(canon_dot(q + i * d, x + j * d, d), qn[i], xn[j], metric, clamp_eps);
  free(qn);
  free(xn);
}

/* ------------------------------------------------------------------ select_k */
static uint32_t float_to_key(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

typedef struct { uint32_t key; int64_t pos; int64_t idx; } cand_t;

static int cmp_key_pos(const void* a, const void* b)
{
  const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
  return 0;
}
static int cmp_key_idx(const void* a, const void* b)
{
  const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return 0;
}

/* Row-wise exact top-k. Winners = k smallest by (key, position); output sorted by (key, index).
 * Restates the call contract of raft::matrix::select_k as used at knn_brute_force.cuh:267,309,
 * ivf_flat_search.cuh:180,283, ivf_pq_search.cuh:160,620 (sorted output, optional input indices). */
EXPORT void oracle_select_k(const float* in, const int64_t* in_idx, int64_t rows, int64_t len, int k,
                            int select_min, int64_t idx_offset, float* out_val, int64_t* out_idx)
{
#pragma omp parallel
  {
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(len > 0 ? len : 1));
#pragma omp for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
      for (int64_t i = 0; i < len; ++i) {
        uint32_t key = float_to_key(in[r * len + i]);
        c[i].key = select_min ? key : ~key;
        c[i].pos = i;
        c[i].idx = in_idx ? in_idx[r * len + i] : i + idx_offset;
      }
      qsort(c, (size_t)len, sizeof(cand_t), cmp_key_pos);
      int64_t ke = k < len ? k : len;
      qsort(c, (size_t)ke, sizeof(cand_t), cmp_key_idx);
      for (int64_t j = 0; j < k; ++j) {
        if (j < ke) {
          out_val[r * k + j] = in[r * len + c[j].pos];
          out_idx[r * k + j] = c[j].idx;
        } else {
          out_val[r * k + j] = select_min ? FLT_MAX : -FLT_MAX;
          out_idx[r * k + j] = -1;
        }
      }
    }
    free(c);
  }
}

/* ------------------------------------------------------------------ brute force, canonical (GPU twin) */
/* Exact kNN with the expanded-form canonical arithmetic; bit-for-bit twin of cuvsBruteForceSearch.
 * keep_bits: optional bitset (n bits, 1 keeps) or bitmap (m*n bits), as c/include/cuvs/neighbors/common.h. */
EXPORT void oracle_brute_force_knn(const float* q, int64_t m, const float* x, int64_t n, int64_t d, int k,
                                   int metric, float clamp_eps, const uint32_t* keep_bits, int bitmap,
                                   int64_t* out_idx, float* out_dist)
{
  float* qn = (float*)malloc(sizeof(float) * (size_t)m);
  float* xn = (float*)malloc(sizeof(float) * (size_t)n);
  oracle_row_norms(q, m, d, qn, metric == M_CosineExpanded);
  oracle_row_norms(x, n, d, xn, metric == M_CosineExpanded);
  int select_min = metric != M_InnerProduct;
  float worst    = select_min ? FLT_MAX : -FLT_MAX;
#pragma omp parallel
  {
    float* row = (float*)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(dynamic, 4)
    for (int64_t i = 0; i < m; ++i) {
      for (int64_t j = 0; j < n; ++j) {
        row[j] = finish_distance(canon_dot(q + i * d, x + j * d, d), qn[i], xn[j], metric, clamp_eps);
        if (keep_bits) {
          int64_t bit = bitmap ? i * n + j : j;
          if (!((keep_bits[bit >> 5] >> (bit & 31)) & 1u)) row[j] = worst;
        }
      }
      /* per-row select on this thread (no nested parallel region) */
      cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)n);
      for (int64_t j = 0; j < n; ++j) {
        uint32_t key = float_to_key(row[j]);
        c[j].key = select_min ? key : ~key;
        c[j].pos = j;
        c[j].idx = j;
      }
      qsort(c, (size_t)n, sizeof(cand_t), cmp_key_pos);
      for (int64_t j = 0; j < k; ++j) {
        if (j < n) { out_dist[i * k + j] = row[c[j].pos]; out_idx[i * k + j] = c[j].idx; }
        else       { out_dist[i * k + j] = worst;         out_idx[i * k + j] = -1; }
      }
      free(c);
    }
    free(row);
  }
  free(qn);
  free(xn);
}

/* ------------------------------------------------------------------ exact kNN, the reference's CPU path */
/* Restates cpp/src/neighbors/refine/refine_host.hpp: euclidean_distance_squared_generic (:29-53, 16
 * strided fp32 accumulators, then serial lane sum), inner product (:465-505: eval = -a*b, postprocess
 * negates back), per-query std::sort of (distance, id) tuples (:430-460 => ties -> smaller id).
 * Used as (a) recall ground truth and (b) bench.py's cpu_baseline ("port": all rows are candidates). */
static float refine_l2(const float* a, const float* b, int64_t n)
{
  enum { V = 16 };
  float acc[V];
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  int64_t nr = n - (n % V);
  for (int64_t i = 0; i < nr; i += V)
    for (int j = 0; j < V; ++j) { float t = a[i + j] - b[i + j]; acc[j] += t * t; }
  for (int64_t i = nr; i < n; ++i) { float t = a[i] - b[i]; acc[i - nr] += t * t; }
  for (int j = 1; j < V; ++j) acc[0] += acc[j];
  return acc[0];
}
static float refine_ip(const float* a, const float* b, int64_t n)
{
  enum { V = 16 };
  float acc[V];
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  int64_t nr = n - (n % V);
  for (int64_t i = 0; i < nr; i += V)
    for (int j = 0; j < V; ++j) acc[j] += -(a[i + j] * b[i + j]);
  for (int64_t i = nr; i < n; ++i) acc[i - nr] += -(a[i] * b[i]);
  for (int j = 1; j < V; ++j) acc[0] += acc[j];
  return acc[0];
}
/* refine_host.hpp:333-350 — cosine accumulates in double */
static float refine_cos(const float* a, const float* b, int64_t n)
{
  double dot = 0, na = 0, nb = 0;
  for (int64_t i = 0; i < n; ++i) { dot += (double)a[i] * b[i]; na += (double)a[i] * a[i]; nb += (double)b[i] * b[i]; }
  double den = sqrt(na) * sqrt(nb);
  return (float)(den > 0 ? 1.0 - dot / den : 1.0);  /* refine_host.hpp:345: a zero denominator gives distance 1 */
}

typedef struct { float d; int64_t id; } pair_t;
static int cmp_pair(const void* a, const void* b)
{
  const pair_t* x = (const pair_t*)a; const pair_t* y = (const pair_t*)b;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}

/* bounded insertion of (d,id) into a sorted array of size k (ascending by (d,id)) */
static void topk_insert(pair_t* best, int k, float d, int64_t id)
{
  if (d > best[k - 1].d || (d == best[k - 1].d && id > best[k - 1].id)) return;
  int j = k - 1;
  while (j > 0 && (best[j - 1].d > d || (best[j - 1].d == d && best[j - 1].id > id))) { best[j] = best[j - 1]; --j; }
  best[j].d = d; best[j].id = id;
}

EXPORT void oracle_exact_knn(const float* q, int64_t m, const float* x, int64_t n, int64_t d, int k, int metric,
                             int64_t* out_idx, float* out_dist)
{
#pragma omp parallel
  {
    pair_t* best = (pair_t*)malloc(sizeof(pair_t) * (size_t)k);
    /* static: every query costs the same; a dynamic schedule made the CPU baseline swing 2x from box to box */
#pragma omp for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
      for (int j = 0; j < k; ++j) { best[j].d = FLT_MAX; best[j].id = INT64_MAX; }
      const float* qi = q + i * d;
      for (int64_t j = 0; j < n; ++j) {
        float dist;
        if (metric == M_InnerProduct) dist = refine_ip(qi, x + j * d, d);
        else if (metric == M_CosineExpanded) dist = refine_cos(qi, x + j * d, d);
        else dist = refine_l2(qi, x + j * d, d);
        topk_insert(best, k, dist, j);
      }
      for (int j = 0; j < k; ++j) {
        float v = best[j].d;
        if (metric == M_InnerProduct) v = -v;
        if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) v = sqrtf(v);
        out_dist[i * k + j] = v;
        out_idx[i * k + j]  = best[j].id == INT64_MAX ? -1 : best[j].id;
      }
    }
    free(best);
  }
  (void)cmp_pair;
}

/* test / bench infrastructure: a launcher may have pinned OMP_NUM_THREADS to 1 for its worker processes (torchrun does) -
 * the CPU baseline leg of bench.py asks for the host's cores explicitly; n <= 0: every online core */
EXPORT void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
  (void)n;
#endif
}

EXPORT int oracle_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ IVF-PQ search on an exported index */
/* ---- reduced-precision LUT entries and scores of the IVF-PQ search (search_params.lut_dtype / internal_distance_dtype,
 * ivf_pq.hpp:167-205). fp16: IEEE half, round to nearest even (F16C). fp8: the reference's own 8-bit storage type
 * fp_8bit<5, Signed> (ivf_pq_fp_8bit.cuh:32-100): 5 exponent bits (bias 15), 3 value bits, truncation on encode, half an
 * ulp added back on decode; the signed flavour (inner-product metrics, ivf_pq_search.cuh:711-728) keeps the sign in bit
 * 0. Its decoders to float (:75-88) and to half (:90-102) differ for the smallest exponent (the half has no implicit
 * one there), and which one runs depends on the score type - both are restated. */
#include <immintrin.h>
static inline uint16_t f32_to_f16(float f) { return _cvtss_sh(f, 0); }
static inline float f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }
static inline uint8_t fp8_encode_unsigned(float v)
{
  const float kMin = 1.0f / 32768.0f, kMax = 65536.0f * (2.0f - 0.125f);
  if (v < kMin) return 0;
  if (v >= kMax) return 0xff;
  uint32_t u; memcpy(&u, &v, 4);
  return (uint8_t)((u + (15u << 23) - 0x3f800000u) >> 20);
}
static inline uint8_t fp8_encode(float v, int is_signed)
{
  if (!is_signed) return fp8_encode_unsigned(v);
  uint8_t u = fp8_encode_unsigned(fabsf(v));
  return (uint8_t)((u & 0xfeu) | (v < 0.f ? 1u : 0u));
}
static inline float fp8_decode_f32(uint8_t b, int is_signed)
{
  uint32_t u = is_signed ? (b & ~1u) : b;
  uint32_t bits = ((0x3f800000u | (0x00400000u >> 3)) - (15u << 23)) + (u << 20);
  float r; memcpy(&r, &bits, 4);
  return (is_signed && (b & 1u)) ? -r : r;
}
static inline uint16_t fp8_decode_f16(uint8_t b, int is_signed)
{
  uint16_t u = is_signed ? (uint16_t)(b & ~1u) : b;
  uint16_t bits = (uint16_t)(((0x3c00u | (0x0200u >> 3)) - (15u << 10)) + (u << 7));
  return (is_signed && (b & 1u)) ? (uint16_t)(bits ^ 0x8000u) : bits;
}
static inline float sat_trunc_i8(float v) { return truncf(fmaxf(-128.0f, fminf(127.0f, v))); }
EXPORT uint8_t oracle_fp8_encode(float v, int is_signed) { return fp8_encode(v, is_signed); }
EXPORT float oracle_fp8_decode_f32(uint8_t b, int is_signed) { return fp8_decode_f32(b, is_signed); }
EXPORT float oracle_fp8_decode_f16(uint8_t b, int is_signed) { return f16_to_f32(fp8_decode_f16(b, is_signed)); }

/* Restates ivf_pq_search.cuh:60-168 (select_clusters), :1003-1017 (rotation), create_lut_impl.cuh:17-78 (LUT),
 * compute_distances_impl.cuh:73-91 / compute_score_impl.cuh:20-79 (score = sum of LUT entries in subspace
 * order), ivf_pq_search.cuh:646-674 (merge of n_probes*k candidates, distance/neighbour post-processing),
 * with fp32 LUT and fp32 accumulation. Codes are given in the externally visible contiguous bit-packed
 * format (ivf_pq_codepacking.cuh:22-52). Tie rules mirror the HIP path: per (query, probe) candidates are
 * ordered by (score, flat row); the merge picks by (score, buffer position) and orders by (score, flat row),
 * flat row = 64-padded list offset + in-list position. */
static uint32_t code_at(const uint8_t* row, int s, int pq_bits)
{
  int bit = s * pq_bits;
  uint32_t v = row[bit >> 3];
  if ((bit & 7) + pq_bits > 8) v |= (uint32_t)row[(bit >> 3) + 1] << 8;
  return (v >> (bit & 7)) & ((1u << pq_bits) - 1u);
}

EXPORT void oracle_ivf_pq_search(const float* queries, int64_t nq, int dim, const float* centers,
                                 const float* centers_rot, const float* rotation, const float* pq_centers,
                                 int n_lists, int rot_dim, int pq_dim, int pq_len, int pq_bits,
                                 const uint32_t* list_sizes, const int64_t* list_start, const uint8_t* codes,
                                 const int64_t* ids, int metric, int n_probes, int k, float scale,
                                 int64_t* neighbors, float* distances, int per_cluster, int lut_mode, int acc_mode, int coarse_mode,
                                 const uint32_t* keep_bits)
{ /* keep_bits: optional bitset over source ids, 1 keeps the row (sample_filter.cuh bitset_filter through
   * ivf_to_sample_filter, applied per scanned row: compute_distances_impl.cuh:78-80) */  /* coarse_mode 0 fp32 / 1 fp16 / 2 int8 coarse search and query rotation (search_params.coarse_search_dtype,
    * ivf_pq_search.cuh:171-340,:995-1017; centers_half / centers_int8 / rotation_matrix_* ivf_pq_index.cu:640-760).
    * lut_mode 0 fp32 / 1 fp16 / 2 fp8 LUT entries; acc_mode 0 fp32 / 1 fp16 scores (sums in subspace order).
    * per_cluster: pq_centers is [n_lists, pq_len, book] (codebook_gen::PER_CLUSTER), else [pq_dim, pq_len, book] */
  const int book = 1 << pq_bits;
  const int bpr  = (pq_dim * pq_bits + 7) / 8;
  const int is_cos = metric == M_CosineExpanded;         /* inner product of unit vectors, reported as 1 - cos */
  const int is_ip  = metric == M_InnerProduct || is_cos;
  if (n_probes > n_lists) n_probes = n_lists;
  float* cn = (float*)malloc(sizeof(float) * (size_t)n_lists);
  oracle_row_norms(centers, n_lists, dim, cn, 0);
  int64_t* pad_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_lists + 1));
  pad_off[0] = 0;
  for (int L = 0; L < n_lists; ++L) pad_off[L + 1] = pad_off[L] + (((int64_t)list_sizes[L] + 63) / 64) * 64;

#pragma omp parallel
  {
    float* cd     = (float*)malloc(sizeof(float) * (size_t)n_lists);
    cand_t* cc    = (cand_t*)malloc(sizeof(cand_t) * (size_t)n_lists);
    float* rq     = (float*)malloc(sizeof(float) * (size_t)rot_dim);
    float* qv     = (float*)malloc(sizeof(float) * (size_t)rot_dim);
    float* qh     = (float*)malloc(sizeof(float) * (size_t)dim);
    float* qcs    = (float*)malloc(sizeof(float) * (size_t)dim);  /* query in the coarse type */
    float* ccs    = (float*)malloc(sizeof(float) * (size_t)dim);  /* a centre / rotation row in the coarse type */
    float* lut    = (float*)malloc(sizeof(float) * (size_t)pq_dim * book);
    pair_t* best  = (pair_t*)malloc(sizeof(pair_t) * (size_t)k);
    float* buf_d  = (float*)malloc(sizeof(float) * (size_t)n_probes * k);
    int64_t* buf_i = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_probes * k);
    cand_t* mc    = (cand_t*)malloc(sizeof(cand_t) * (size_t)n_probes * k);
#pragma omp for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
      const float* q = queries + qi * dim;
      if (is_cos) { /* distance.hip normalize_rows_kernel: x * (1 / sqrt(canonical |x|^2)) */
        const float n2 = canon_sqnorm(q, dim), inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
        for (int d = 0; d < dim; ++d) qh[d] = q[d] * inv;
        q = qh;
      }
      /* coarse */
      float qn = canon_sqnorm(q, dim);
      if (coarse_mode != 0) {
        for (int d = 0; d < dim; ++d) qcs[d] = coarse_mode == 1 ? f16_to_f32(f32_to_f16(q[d])) : sat_trunc_i8(q[d] * 128.0f);
      }
      const int m8 = ((dim + 2 + 15) / 16) * 16 - dim;  /* dim_ext_int8 - dim */
      for (int j = 0; j < n_lists; ++j) {
        if (coarse_mode == 0) {
          float dot = canon_dot(q, centers + (int64_t)j * dim, dim);
          cd[j]     = is_ip ? dot : finish_distance(dot, qn, cn[j], M_L2Expanded, 1e-6f);
        } else {
          const float* c = centers + (int64_t)j * dim;
          for (int d = 0; d < dim; ++d) ccs[d] = coarse_mode == 1 ? f16_to_f32(f32_to_f16(c[d])) : sat_trunc_i8(c[d] * 128.0f);
          float v = canon_dot(qcs, ccs, dim);
          if (coarse_mode == 1) {
            if (!is_ip) v = v + -0.5f * f16_to_f32(f32_to_f16(cn[j]));
            cd[j] = f16_to_f32(f32_to_f16((is_ip ? -1.0f : -2.0f) * v));
          } else {
            const float cc8 = 64.0f / (float)(m8 - 1);
            const float y   = fmaxf(-128.0f, fminf(127.0f, cn[j] * cc8));
            const float z   = sat_trunc_i8((y - roundf(y)) * 128.0f);
            const float yr  = truncf(roundf(y));
            v     = v + (z * (float)(1 - m8) + (is_ip ? 0.0f : (float)(m8 - 1) * yr * -128.0f));
            cd[j] = (is_ip ? -1.0f : -2.0f) * v;
          }
        }
        uint32_t key = float_to_key(cd[j]);
        cc[j].key = (is_ip && coarse_mode == 0) ? ~key : key;  /* the reduced-precision GEMMs already minimise */
        cc[j].pos = j;
        cc[j].idx = j;
      }
      qsort(cc, (size_t)n_lists, sizeof(cand_t), cmp_key_pos);
      for (int r = 0; r < rot_dim; ++r) {
        const float* rr = rotation + (int64_t)r * dim;
        if (coarse_mode == 0) { rq[r] = canon_dot(q, rr, dim); continue; }
        for (int d = 0; d < dim; ++d) ccs[d] = coarse_mode == 1 ? f16_to_f32(f32_to_f16(rr[d])) : sat_trunc_i8(rr[d] * 128.0f);
        rq[r] = canon_dot(qcs, ccs, dim);
        if (coarse_mode == 2) rq[r] *= 1.0f / 128.0f / 128.0f;
      }
      if (coarse_mode != 0 && is_cos) { /* rotated queries are re-normalised (ivf_pq_search.cuh:1018-1023) */
        const float n2 = canon_sqnorm(rq, rot_dim), inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
        for (int r = 0; r < rot_dim; ++r) rq[r] = rq[r] * inv;
      }
      for (int p = 0; p < n_probes; ++p) {
        const int L = (int)cc[p].idx;
        const float* cr = centers_rot + (int64_t)L * rot_dim;
        for (int r = 0; r < rot_dim; ++r) qv[r] = is_ip ? rq[r] : rq[r] - cr[r];
        for (int s = 0; s < pq_dim; ++s)
          for (int c = 0; c < book; ++c) {
            float sc = 0.f;
            for (int l = 0; l < pq_len; ++l) {
              int dd  = s * pq_len + l;
              float pc = pq_centers[(int64_t)(per_cluster ? L * pq_len + l : dd) * book + c];
              if (!is_ip) { float diff = qv[dd] - pc; sc = fmaf(diff, diff, sc); }
              else        { sc = fmaf(-qv[dd], cr[dd], sc); sc = fmaf(-qv[dd], pc, sc); }
            }
            if (lut_mode == 1) sc = f16_to_f32(f32_to_f16(sc));
            else if (lut_mode == 2) {
              const uint8_t b = fp8_encode(sc, is_ip);
              sc = acc_mode == 1 ? f16_to_f32(fp8_decode_f16(b, is_ip)) : fp8_decode_f32(b, is_ip);
            }
            lut[s * book + c] = sc;
          }
        for (int j = 0; j < k; ++j) { best[j].d = FLT_MAX; best[j].id = INT64_MAX; }
        const uint32_t len = list_sizes[L];
        for (uint32_t v = 0; v < len; ++v) {
          const uint8_t* row = codes + (list_start[L] + v) * bpr;
          if (keep_bits != NULL) {
            const int64_t sid = ids[list_start[L] + v];
            if (!((keep_bits[sid >> 5] >> (sid & 31)) & 1u)) continue;
          }
          float acc = 0.f;
          if (acc_mode == 1) {  /* half-precision adds: every partial sum rounds to half */
            for (int s = 0; s < pq_dim; ++s) acc = f16_to_f32(f32_to_f16(acc + lut[s * book + code_at(row, s, pq_bits)]));
          } else {
            for (int s = 0; s < pq_dim; ++s) acc = acc + lut[s * book + code_at(row, s, pq_bits)];
          }
          topk_insert(best, k, acc, pad_off[L] + v);
        }
        for (int j = 0; j < k; ++j) {
          int valid = best[j].id != INT64_MAX;
          buf_d[p * k + j] = valid ? best[j].d : FLT_MAX;
          buf_i[p * k + j] = valid ? best[j].id : 0xffffffffLL;
        }
      }
      /* merge */
      int m = n_probes * k;
      for (int t = 0; t < m; ++t) { mc[t].key = float_to_key(buf_d[t]); mc[t].pos = t; mc[t].idx = buf_i[t]; }
      qsort(mc, (size_t)m, sizeof(cand_t), cmp_key_pos);
      int ke = k < m ? k : m;
      qsort(mc, (size_t)ke, sizeof(cand_t), cmp_key_idx);
      for (int j = 0; j < k; ++j) {
        float d = j < ke ? buf_d[mc[j].pos] : FLT_MAX;
        int64_t fr = j < ke ? mc[j].idx : 0xffffffffLL;
        if (fr == 0xffffffffLL) {
          neighbors[qi * k + j] = INT64_MAX;
          distances[qi * k + j] = FLT_MAX;
        } else {
          /* flat row -> (list, pos) -> id */
          int lo = 0, hi = n_lists;  /* last L with pad_off[L] <= fr */
          while (hi - lo > 1) { int mid = (lo + hi) / 2; if (pad_off[mid] <= fr) lo = mid; else hi = mid; }
          neighbors[qi * k + j] = ids[list_start[lo] + (fr - pad_off[lo])];
          float s2 = scale * scale;
          if (is_cos) d = 1.0f + d;
          else if (is_ip) d = -d * s2;
          else if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) d = sqrtf(d * s2);
          else d = d * s2;
          distances[qi * k + j] = d;
        }
      }
    }
    free(qcs); free(ccs);
    free(cd); free(cc); free(rq); free(qv); free(qh); free(lut); free(best); free(buf_d); free(buf_i); free(mc);
  }
  free(cn);
  free(pad_off);
}

/* PQ encoding of rotated residuals (encode_vectors, ivf_pq_process_and_fill_codes.cuh:65-112): per subspace
 * argmin over the codebook of sum_l (r - p)^2, ties -> smaller code. out: [n, pq_dim] one code per byte. */
EXPORT void oracle_pq_encode(const float* resid, int64_t n, int rot_dim, const float* pq_centers, int pq_dim,
                             int pq_len, int pq_bits, uint8_t* out)
{
  const int book = 1 << pq_bits;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    for (int s = 0; s < pq_dim; ++s) {
      float bestd = INFINITY; int code = 0;
      for (int c = 0; c < book; ++c) {
        float d = 0.f;
        for (int l = 0; l < pq_len; ++l) {
          float t = resid[i * rot_dim + s * pq_len + l] - pq_centers[((int64_t)s * pq_len + l) * book + c];
          d = fmaf(t, t, d);
        }
        if (d < bestd) { bestd = d; code = c; }
      }
      out[i * pq_dim + s] = (uint8_t)code;
    }
}
