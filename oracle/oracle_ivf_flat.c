/*
 * oracle_ivf_flat.c — CPU twin of cuvsIvfFlatSearch on an exported index (TEST INFRASTRUCTURE ONLY).
 * Restates ivf_flat_search.cuh:104-187 (coarse search), interleaved_scan_impl.cuh:127-204 +
 * load_and_compute_dist_impl.cuh:690-738 + metric_impl.cuh:12-49 (per-row distance accumulated in
 * dimension order with fma; unexpanded L2 for both L2 variants; inner product; int8 / uint8 rows accumulate in
 * integers - metric_impl.cuh:12-49, dp4a - and the exact sum is converted to float once), the n_probes*k merge
 * (ivf_flat_search.cuh:273-295) and post-processing (post_process_impl.cuh:12-30).
 * Tie rules: per (query, probe) candidates ordered by (distance, flat row); merge by (distance, buffer
 * position) then ordered by (distance, flat row); flat row = 64-padded list offset + in-list position.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

typedef struct { uint32_t key; int64_t pos; int64_t idx; } fcand_t;
typedef struct { float d; int64_t id; } fpair_t;

static uint32_t f2k(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static int cmp_pos(const void* a, const void* b)
{
  const fcand_t* x = (const fcand_t*)a; const fcand_t* y = (const fcand_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
static int cmp_idx(const void* a, const void* b)
{
  const fcand_t* x = (const fcand_t*)a; const fcand_t* y = (const fcand_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static float sqnorm64(const float* a, int64_t d)
{
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.f;
  for (int64_t j = 0; j < d; ++j) p[j & 63] = fmaf(a[j], a[j], p[j & 63]);
  for (int off = 32; off > 0; off >>= 1)
    for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
  return p[0];
}
static void ins(fpair_t* best, int k, float d, int64_t id)
{
  if (d > best[k - 1].d || (d == best[k - 1].d && id > best[k - 1].id)) return;
  int j = k - 1;
  while (j > 0 && (best[j - 1].d > d || (best[j - 1].d == d && best[j - 1].id > id))) { best[j] = best[j - 1]; --j; }
  best[j].d = d; best[j].id = id;
}

/* queries_coarse: queries as the coarse search sees them (mapped floats); queries_raw / rows: raw values as float */
EXPORT void oracle_ivf_flat_search(const float* queries_coarse, const float* queries_raw, int64_t nq, int dim,
                                   const float* centers, int n_lists, const uint32_t* list_sizes,
                                   const int64_t* list_start, const float* rows, const int64_t* ids, int metric,
                                   int n_probes, int k, int int_mode, int64_t* neighbors, float* distances)
{
  const int is_ip = metric == 6, is_cos = metric == 2;
  if (n_probes > n_lists) n_probes = n_lists;
  float* cn = (float*)malloc(sizeof(float) * (size_t)n_lists);
  for (int j = 0; j < n_lists; ++j) cn[j] = sqnorm64(centers + (int64_t)j * dim, dim);
  int64_t* pad_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_lists + 1));
  pad_off[0] = 0;
  for (int L = 0; L < n_lists; ++L) pad_off[L + 1] = pad_off[L] + (((int64_t)list_sizes[L] + 63) / 64) * 64;
#pragma omp parallel
  {
    fcand_t* cc    = (fcand_t*)malloc(sizeof(fcand_t) * (size_t)n_lists);
    fpair_t* best  = (fpair_t*)malloc(sizeof(fpair_t) * (size_t)k);
    float* buf_d   = (float*)malloc(sizeof(float) * (size_t)n_probes * k);
    int64_t* buf_i = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_probes * k);
    fcand_t* mc    = (fcand_t*)malloc(sizeof(fcand_t) * (size_t)n_probes * k);
#pragma omp for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
      const float* qc = queries_coarse + qi * dim;
      const float* qr = queries_raw + qi * dim;
      float qn = sqnorm64(qc, dim);
      float qn_fine = 0.f;  /* |q| of the raw query, squares accumulated in dimension order */
      for (int d = 0; d < dim; ++d) qn_fine = fmaf(qr[d], qr[d], qn_fine);
      qn_fine = sqrtf(qn_fine);
      for (int j = 0; j < n_lists; ++j) {
        float dot = 0.f;
        for (int d = 0; d < dim; ++d) dot = fmaf(qc[d], centers[(int64_t)j * dim + d], dot);
        float v;
        if (is_ip) v = dot;
        else if (is_cos) v = 1.0f - dot / (sqrtf(qn) * sqrtf(cn[j]));  /* distance_tile.hpp finish_distance */
        else {
          v = fmaf(-2.0f, dot, qn + cn[j]);
          if (v * v < 1e-6f && qn == cn[j]) v = 0.f;
          v = v > 0.f ? v : 0.f;
        }
        uint32_t key = f2k(v);
        cc[j].key = is_ip ? ~key : key; cc[j].pos = j; cc[j].idx = j;
      }
      qsort(cc, (size_t)n_lists, sizeof(fcand_t), cmp_pos);
      for (int p = 0; p < n_probes; ++p) {
        const int L = (int)cc[p].idx;
        for (int j = 0; j < k; ++j) { best[j].d = FLT_MAX; best[j].id = INT64_MAX; }
        for (uint32_t v = 0; v < list_sizes[L]; ++v) {
          const float* x = rows + (list_start[L] + v) * dim;
          float acc = 0.f;
          if (is_cos) {
            /* ivf_flat.hip METRIC 2: dot and |x|^2 in dimension order, cos = dot / (|q| * |x|), key -cos */
            float xn2 = 0.f;
            for (int d = 0; d < dim; ++d) { xn2 = fmaf(x[d], x[d], xn2); acc = fmaf(x[d], qr[d], acc); }
            acc = acc / (qn_fine * sqrtf(xn2));
          } else if (int_mode) {  /* int8 / uint8: integer accumulator (exact), one conversion at the end */
            int64_t ia = 0;
            if (!is_ip) for (int d = 0; d < dim; ++d) { int64_t t = (int64_t)x[d] - (int64_t)qr[d]; ia += t * t; }
            else        for (int d = 0; d < dim; ++d) ia += (int64_t)x[d] * (int64_t)qr[d];
            acc = (float)ia;
          } else if (!is_ip) for (int d = 0; d < dim; ++d) { float t = x[d] - qr[d]; acc = fmaf(t, t, acc); }
          else               for (int d = 0; d < dim; ++d) acc = fmaf(x[d], qr[d], acc);
          ins(best, k, (is_ip || is_cos) ? -acc : acc, pad_off[L] + v);
        }
        for (int j = 0; j < k; ++j) {
          int valid = best[j].id != INT64_MAX;
          buf_d[p * k + j] = valid ? best[j].d : FLT_MAX;
          buf_i[p * k + j] = valid ? best[j].id : 0xffffffffLL;
        }
      }
      int m = n_probes * k;
      for (int t = 0; t < m; ++t) { mc[t].key = f2k(buf_d[t]); mc[t].pos = t; mc[t].idx = buf_i[t]; }
      qsort(mc, (size_t)m, sizeof(fcand_t), cmp_pos);
      int ke = k < m ? k : m;
      qsort(mc, (size_t)ke, sizeof(fcand_t), cmp_idx);
      for (int j = 0; j < k; ++j) {
        float d = j < ke ? buf_d[mc[j].pos] : FLT_MAX;
        int64_t fr = j < ke ? mc[j].idx : 0xffffffffLL;
        if (fr == 0xffffffffLL) { neighbors[qi * k + j] = INT64_MAX; distances[qi * k + j] = FLT_MAX; continue; }
        int lo = 0, hi = n_lists;
        while (hi - lo > 1) { int mid = (lo + hi) / 2; if (pad_off[mid] <= fr) lo = mid; else hi = mid; }
        neighbors[qi * k + j] = ids[list_start[lo] + (fr - pad_off[lo])];
        if (is_ip) d = -d;
        else if (is_cos) d = 1.0f + d;
        else if (metric == 1 || metric == 5) d = sqrtf(d);
        distances[qi * k + j] = d;
      }
    }
    free(cc); free(best); free(buf_d); free(buf_i); free(mc);
  }
  free(cn); free(pad_off);
}
