/*
 * oracle_refine.c — CPU twin of cuvsRefine (TEST INFRASTRUCTURE ONLY, see oracle.c header).
 * Restates cpp/src/neighbors/refine/refine_host.hpp:353-462: exact distance of every candidate, then a
 * per-query sort of (distance, id) tuples; a candidate id outside [0, n) STAYS in the list with
 * distance = max (:440-442) - it sorts behind every real row, among its like by id, and comes out with its
 * own id and postprocess(max) (sqrt for the L2Sqrt metrics, -max for inner product, :465-505).
 * Arithmetic is the HIP kernel's: 64 strided fmaf partial sums + butterfly.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define EXPORT __attribute__((visibility("default")))

typedef struct { float d; int64_t id; } rpair_t;

static int cmp_rpair(const void* a, const void* b)
{
  const rpair_t* x = (const rpair_t*)a; const rpair_t* y = (const rpair_t*)b;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}

static float lane_reduce(const float* a, const float* b, int64_t d, int ip)
{
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.f;
  for (int64_t j = 0; j < d; ++j) {
    if (ip) p[j & 63] = fmaf(a[j], b[j], p[j & 63]);
    else { float t = a[j] - b[j]; p[j & 63] = fmaf(t, t, p[j & 63]); }
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
  return p[0];
}

/* metric: 0/4 L2 squared, 1/5 L2 sqrt, 6 inner product, 2 cosine (1 - q.x / (|q| |x|), all three sums with the
 * strided partials of lane_reduce) */
EXPORT void oracle_refine(const float* data, int64_t n, int64_t dim, const float* queries, int64_t m,
                          const int64_t* cand, int n_cand, int k, int metric, int64_t* out_i, float* out_d)
{
  const int ip = metric == 6, cosm = metric == 2;
#pragma omp parallel
  {
    rpair_t* buf = (rpair_t*)malloc(sizeof(rpair_t) * (size_t)n_cand);
#pragma omp for schedule(static)
    for (int64_t q = 0; q < m; ++q) {
      int cnt = 0;
      for (int c = 0; c < n_cand; ++c) {
        int64_t id = cand[q * n_cand + c];
        if (id < 0 || id >= n) {  /* static_cast<size_t>(id) >= n_rows: distance = max, the id is kept */
          buf[cnt].d = FLT_MAX; buf[cnt].id = id; ++cnt;
          continue;
        }
        float v = lane_reduce(queries + q * dim, data + id * dim, dim, ip || cosm);
        if (cosm) {
          const float qn = sqrtf(lane_reduce(queries + q * dim, queries + q * dim, dim, 1));
          const float xn = lane_reduce(data + id * dim, data + id * dim, dim, 1);
          v = 1.0f - v / (qn * sqrtf(xn));
        }
        buf[cnt].d  = ip ? -v : v;  /* sort key: smaller is better */
        buf[cnt].id = id;
        ++cnt;
      }
      qsort(buf, (size_t)cnt, sizeof(rpair_t), cmp_rpair);
      for (int j = 0; j < k; ++j) {
        if (j < cnt) {
          float d = ip ? -buf[j].d : buf[j].d;
          if (metric == 1 || metric == 5) d = sqrtf(d);
          out_i[q * k + j] = buf[j].id;
          out_d[q * k + j] = d;
        } else {
          out_i[q * k + j] = INT64_MAX;
          out_d[q * k + j] = FLT_MAX;
        }
      }
    }
    free(buf);
  }
}
