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
  return (float)(den > 0 ? 1.0 - dot / den : 0.0);
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
#pragma omp for schedule(dynamic, 1)
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

EXPORT int oracle_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
