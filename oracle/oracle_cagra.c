/*
 * oracle_cagra.c — CPU twin of cuvsCagraSearch on an exported index (TEST INFRASTRUCTURE ONLY, see oracle.c).
 *
 * Restates the reference's single-CTA search loop (search_single_cta_jit.cuh:105-452): seed every slot of the result
 * buffer (itopk + search_width * degree slots) with the nearest of num_random_samplings pseudo-random nodes
 * (device_common_jit.cuh:36-104: gid = slot + n_slots * j, xorshift64(gid ^ rand_xor_mask) % n, first wins ties), then
 * repeat { sort; pick the best entries that were not parents yet (search_single_cta_device_helpers.cuh:98-136,
 * MSB of the index marks "used"); expand their graph rows; drop children already seen (hashmap.hpp:37-134, a
 * SET here — the small-hash reset of the reference keeps only the current top list); compute child distances }
 * until no parent is left or max_iterations (search_plan.cuh:199-215) is reached.
 * Arithmetic is the HIP kernel's team distance (compute_distance_impl.cuh:23-64 restated for 8 lanes x 16 B):
 * lane t of a team accumulates the elements of its 16-byte pieces in order with fmaf; the 8 partial sums are
 * combined by the xor butterfly (1, 2, 4).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))
#define INVALID 0xffffffffu
#define PFLAG 0x80000000u

typedef struct { uint32_t key; uint32_t idx; } ent_t;

static int cmp_ent(const void* a, const void* b)
{
  const ent_t* x = (const ent_t*)a; const ent_t* y = (const ent_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return 0;
}
static uint32_t f2key(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static float key2f(uint32_t k)
{
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint64_t xorshift64(uint64_t u)
{
  u ^= u >> 12; u ^= u << 25; u ^= u >> 27;
  return u * 0x2545F4914F6CDD1DULL;
}
/* vl = elements per 16-byte piece of the dataset dtype (4 fp32, 8 fp16, 16 int8) */
static float team_distance(const float* x, const float* q, int64_t dim, int vl, int is_ip)
{
  float p[8];
  for (int t = 0; t < 8; ++t) {
    float acc = 0.f;
    for (int64_t d0 = (int64_t)t * vl; d0 < dim; d0 += 8 * vl)
      for (int e = 0; e < vl && d0 + e < dim; ++e) {
        if (is_ip) acc = fmaf(x[d0 + e], q[d0 + e], acc);
        else { float df = x[d0 + e] - q[d0 + e]; acc = fmaf(df, df, acc); }
      }
    p[t] = acc;
  }
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) a[i] = p[i] + p[i ^ 1];
  for (int i = 0; i < 8; ++i) b[i] = a[i] + a[i ^ 2];
  for (int i = 0; i < 8; ++i) c[i] = b[i] + b[i ^ 4];
  return c[0];
}

/* tiny open-addressing set */
typedef struct { uint32_t* t; uint32_t mask; } set_t;
static void set_clear(set_t* s) { memset(s->t, 0xff, sizeof(uint32_t) * (size_t)(s->mask + 1)); }
static int set_insert(set_t* s, uint32_t key, uint32_t bits)  /* 1 when new */
{
  uint32_t pos = (key ^ (key >> bits)) & s->mask;
  for (uint32_t probe = 0; probe <= s->mask; ++probe) {
    if (s->t[pos] == INVALID) { s->t[pos] = key; return 1; }
    if (s->t[pos] == key) return 0;
    pos = (pos + 1) & s->mask;
  }
  return 0;
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
/* is_ip: 0 L2, 1 inner product, 2 cosine = 1 - q.x / (|q| |x|) with canonical norms (cagra.hip team_distances) */
static float node_key_value(const float* x, const float* q, int64_t dim, int vl, int is_ip, float qn)
{
  float d = team_distance(x, q, dim, vl, is_ip != 0);
  if (is_ip == 2) d = 1.0f - d / (qn * sqrtf(sqnorm64(x, dim)));
  return is_ip == 1 ? -d : d;
}

EXPORT void oracle_cagra_search(const float* data, int64_t n, int64_t dim, int vl, const uint32_t* graph, int degree,
                                const float* queries, int64_t nq, int k, int itopk, int width, int max_iter,
                                int min_iter, int hash_bits, int reset_interval, uint64_t rand_xor_mask, int is_ip,
                                const uint32_t* filter_bits, int64_t* out_idx, float* out_dist, int n_distill)
{
  int np2 = 1;
  while (np2 < itopk + width * degree) np2 <<= 1;
#pragma omp parallel
  {
    ent_t* e = (ent_t*)malloc(sizeof(ent_t) * (size_t)np2);
    set_t set;
    set.mask = (1u << hash_bits) - 1u;
    set.t    = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(set.mask + 1));
    uint32_t* parents = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)width);
#pragma omp for schedule(dynamic, 4)
    for (int64_t qi = 0; qi < nq; ++qi) {
      const float* q = queries + qi * dim;
      const float qn = is_ip == 2 ? sqrtf(sqnorm64(q, dim)) : 1.f;
      for (int i = 0; i < np2; ++i) { e[i].key = 0xffffffffu; e[i].idx = INVALID; }
      set_clear(&set);
      const int n_seed = itopk + width * degree;  /* result_buffer_size (search_single_cta_jit.cuh:110) */
      for (int i = 0; i < n_seed; ++i) {
        uint32_t best_key = 0xffffffffu, best = INVALID;
        for (int j = 0; j < n_distill; ++j) {  /* device_common_jit.cuh:66-83 */
          uint64_t gid  = (uint64_t)i + (uint64_t)n_seed * (uint64_t)j;
          uint32_t node = (uint32_t)(xorshift64(gid ^ rand_xor_mask) % (uint64_t)n);
          uint32_t key  = f2key(node_key_value(data + (int64_t)node * dim, q, dim, vl, is_ip, qn));
          if (key < best_key) { best_key = key; best = node; }
        }
        if (best != INVALID && set_insert(&set, best, hash_bits)) { e[i].idx = best; e[i].key = best_key; }
      }
      int iter = 0;
      for (;;) {
        qsort(e, (size_t)np2, sizeof(ent_t), cmp_ent);
        if (iter >= max_iter) break;
        if (iter > 0 && reset_interval > 0 && (iter % reset_interval) == 0) {
          set_clear(&set);
          for (int i = 0; i < itopk; ++i)
            if (e[i].idx != INVALID) set_insert(&set, e[i].idx & ~PFLAG, hash_bits);
        }
        int np = 0;
        for (int i = 0; i < itopk && np < width; ++i)
          if (e[i].idx != INVALID && !(e[i].idx & PFLAG)) { parents[np++] = e[i].idx; e[i].idx |= PFLAG; }
        if (np == 0 && iter >= min_iter) break;
        for (int i = 0; i < width * degree; ++i) {
          int w = i / degree, c = i % degree;
          uint32_t child = INVALID;
          if (w < np) {
            child = graph[(int64_t)parents[w] * degree + c];
            if (child >= n || !set_insert(&set, child, hash_bits)) child = INVALID;
          }
          e[itopk + i].idx = child;
          e[itopk + i].key = 0xffffffffu;
          if (child != INVALID) {
            e[itopk + i].key = f2key(node_key_value(data + (int64_t)child * dim, q, dim, vl, is_ip, qn));
          }
        }
        ++iter;
      }
      int written = 0;
      for (int i = 0; i < itopk && written < k; ++i) {
        if (e[i].idx == INVALID) continue;
        uint32_t node = e[i].idx & ~PFLAG;
        if (filter_bits && !((filter_bits[node >> 5] >> (node & 31)) & 1u)) continue;
        float d = key2f(e[i].key);
        out_idx[qi * k + written]  = node;
        out_dist[qi * k + written] = is_ip == 1 ? -d : d;
        ++written;
      }
      for (; written < k; ++written) { out_idx[qi * k + written] = -1; out_dist[qi * k + written] = FLT_MAX; }
    }
    free(e); free(set.t); free(parents);
  }
}
