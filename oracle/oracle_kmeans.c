/*
 * oracle_kmeans.c — CPU twin of the balanced k-means in cuvs_amd/csrc/kmeans_balanced.hip
 * (TEST INFRASTRUCTURE ONLY, see oracle.c). Restates cpp/src/cluster/detail/kmeans_balanced.cuh:
 * build_clusters :724-783 (labels i mod k, calc_centers_and_sizes, balancing EM), balancing_em_iters :645-722,
 * adjust_centers :464-580 (kAdjustCentersWeight 7, prime offsets, "average" rule), calc_centers_and_sizes
 * :253-315 (mean per cluster, empty -> 0), arrange_fine_clusters :786-848, build_hierarchical :986-1148.
 * Two deliberate differences from the reference, shared with the HIP code so that results are reproducible:
 * the row that re-seeds a small cluster is chosen from a per-cluster sequence instead of a racing atomic
 * counter, and means are summed in a fixed order (S strided partial sums per cluster and dimension).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

static float k_dot(const float* a, const float* b, int d)
{
  float acc = 0.f;
  for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
  return acc;
}
static float k_sqnorm(const float* a, int d)
{
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.f;
  for (int j = 0; j < d; ++j) p[j & 63] = fmaf(a[j], a[j], p[j & 63]);
  for (int off = 32; off > 0; off >>= 1)
    for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
  return p[0];
}

/* E-step: labels[i] = argmin_j fma(-2, dot(x_i, c_j), |c_j|^2), ties -> smallest j */
static void predict(const float* x, int64_t n, int64_t ld, int dim, const float* centers, int k, uint32_t* labels)
{
  float* cn = (float*)malloc(sizeof(float) * (size_t)k);
  for (int j = 0; j < k; ++j) cn[j] = k_sqnorm(centers + (int64_t)j * dim, dim);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float best = 0.f; int bj = -1;
    for (int j = 0; j < k; ++j) {
      float v = fmaf(-2.0f, k_dot(x + i * ld, centers + (int64_t)j * dim, dim), cn[j]);
      if (bj < 0 || v < best) { best = v; bj = j; }
    }
    labels[i] = (uint32_t)bj;
  }
  free(cn);
}

/* M-step: rows of a cluster in ascending id; S strided partial sums combined ((p0+p1)+p2)+... with S = 4 for
 * dim > 32 and 4 * 64 / pow2ceil(dim) below (kmeans_balanced.hip cluster_means_kernel: rows per wave step) */
static void calc_centers_and_sizes(const float* x, int64_t n, int64_t ld, int dim, int k, const uint32_t* labels,
                                   float* centers, uint32_t* sizes)
{
  int64_t* off = (int64_t*)calloc((size_t)k + 1, sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) off[labels[i] + 1]++;
  for (int c = 0; c < k; ++c) off[c + 1] += off[c];
  int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  int64_t* cur  = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
  memcpy(cur, off, sizeof(int64_t) * (size_t)k);
  for (int64_t i = 0; i < n; ++i) perm[cur[labels[i]]++] = i;
#pragma omp parallel for schedule(dynamic, 8)
  for (int c = 0; c < k; ++c) {
    int64_t b = off[c], e = off[c + 1];
    sizes[c]  = (uint32_t)(e - b);
    int dp = 64, R = 1;
    if (dim <= 32) { dp = 1; while (dp < dim) dp <<= 1; R = 64 / dp; }
    const int S = 4 * R;
    for (int d = 0; d < dim; ++d) {
      float p[256];
      for (int w = 0; w < S; ++w) {
        p[w] = 0.f;
        for (int64_t j = b + w; j < e; j += S) p[w] += x[perm[j] * ld + d];
      }
      float s = p[0];
      for (int w = 1; w < S; ++w) s = s + p[w];
      centers[(int64_t)c * dim + d] = (e > b) ? s / (float)(e - b) : 0.f;
    }
  }
  free(off); free(perm); free(cur);
}

static const int kPrimes[] = {29,   71,   113,  173,  229,  281,  349,  409,  463,  541,  601,  659,  733,  809,
                              863,  941,  1013, 1069, 1151, 1223, 1291, 1373, 1451, 1511, 1583, 1657, 1733, 1811,
                              1889, 1987, 2053, 2129, 2213, 2287, 2357, 2423, 2531, 2617, 2687, 2741};

static int adjust_centers(float* centers, int k, int dim, const float* x, int64_t ld, int64_t n,
                          const uint32_t* labels, const uint32_t* sizes, float threshold, int* i_primes)
{
  if (k == 0) return 0;
  const int n_primes = (int)(sizeof(kPrimes) / sizeof(int));
  int64_t average = n / k, ofst;
  do { *i_primes = (*i_primes + 1) % n_primes; ofst = kPrimes[*i_primes]; } while (n % ofst == 0);
  int adjusted = 0;
  /* reads of centers[li] never alias a centre written in this pass (li is a large cluster) */
  for (int l = 0; l < k; ++l) {
    int64_t csz = sizes[l];
    if ((float)csz > (float)average * threshold) continue;
    int64_t i = 0, t = 0;
    do {
      i = (int64_t)(((unsigned long long)ofst * (unsigned long long)(l + 1 + t * (int64_t)k)) % (unsigned long long)n);
      ++t;
    } while ((int64_t)sizes[labels[i]] < average && t < 100000);
    adjusted = 1;
    int64_t li = labels[i];
    float wc = fminf((float)csz, 7.0f), wd = 1.0f;
    for (int j = 0; j < dim; ++j) {
      float val = 0.f;
      val += wc * centers[j + (int64_t)dim * li];
      val += wd * x[j + ld * i];
      val /= wc + wd;
      centers[j + (int64_t)dim * l] = val;
    }
  }
  return adjusted;
}

static void balancing_em_iters(uint32_t n_iters, int dim, const float* x, int64_t ld, int64_t n, int k, float* centers,
                               uint32_t* labels, uint32_t* sizes, uint32_t pullback, float threshold, int* i_primes)
{
  uint32_t counter = pullback;
  for (uint32_t iter = 0; iter < n_iters; iter++) {
    if (iter > 0 && adjust_centers(centers, k, dim, x, ld, n, labels, sizes, threshold, i_primes)) {
      if (counter++ >= pullback) { counter -= pullback; n_iters++; }
    }
    predict(x, n, ld, dim, centers, k, labels);
    calc_centers_and_sizes(x, n, ld, dim, k, labels, centers, sizes);
  }
}

EXPORT void oracle_kmeans_build_clusters(const float* x, int64_t n, int64_t ld, int dim, int k, int n_iters,
                                         float* centers, uint32_t* labels, uint32_t* sizes)
{
  int i_primes = 0;
  for (int64_t i = 0; i < n; ++i) labels[i] = (uint32_t)(i % k);
  calc_centers_and_sizes(x, n, ld, dim, k, labels, centers, sizes);
  balancing_em_iters((uint32_t)n_iters, dim, x, ld, n, k, centers, labels, sizes, 2, 0.25f, &i_primes);
}

EXPORT void oracle_kmeans_balanced_fit(const float* x, int64_t n, int dim, int k, int n_iters, int hierarchical,
                                       float* centers, uint32_t* out_labels)
{
  int n_meso = (int)(sqrt((double)k) + 0.5);
  if (n_meso > k) n_meso = k;
  uint32_t* labels = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  uint32_t* sizes  = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)k);
  if (!hierarchical || n_meso <= 1 || n_meso == k) {
    oracle_kmeans_build_clusters(x, n, dim, dim, k, n_iters, centers, labels, sizes);
  } else {
    int i_primes = 0;
    uint32_t* mlab   = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    uint32_t* msizes = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_meso);
    float* mcent     = (float*)malloc(sizeof(float) * (size_t)n_meso * dim);
    oracle_kmeans_build_clusters(x, n, dim, dim, n_meso, n_iters, mcent, mlab, msizes);
    /* arrange_fine_clusters */
    int64_t* fine_nums = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_meso);
    int64_t* fine_csum = (int64_t*)calloc((size_t)n_meso + 1, sizeof(int64_t));
    int64_t lists_rem = k, nonempty_rem = 0, rows_rem = n, msize_max = 0, fmax = 0;
    for (int i = 0; i < n_meso; i++) nonempty_rem += msizes[i] > 0 ? 1 : 0;
    for (int i = 0; i < n_meso; i++) {
      if (i < n_meso - 1) {
        if (msizes[i] == 0) fine_nums[i] = 0;
        else {
          nonempty_rem--;
          int64_t s = (int64_t)((double)(lists_rem * (int64_t)msizes[i]) / (double)rows_rem + .5);
          if (s > lists_rem - nonempty_rem) s = lists_rem - nonempty_rem;
          fine_nums[i] = s > 1 ? s : 1;
        }
      } else fine_nums[i] = lists_rem;
      lists_rem -= fine_nums[i];
      rows_rem -= msizes[i];
      if ((int64_t)msizes[i] > msize_max) msize_max = msizes[i];
      if (fine_nums[i] > fmax) fmax = fine_nums[i];
      fine_csum[i + 1] = fine_csum[i] + fine_nums[i];
    }
    int64_t balanced = (2 * n + n_meso - 1) / (n_meso > 1 ? n_meso : 1);
    if (msize_max > balanced) msize_max = balanced;
    /* rows of every mesocluster in ascending id */
    int64_t* off = (int64_t*)calloc((size_t)n_meso + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) off[mlab[i] + 1]++;
    for (int c = 0; c < n_meso; ++c) off[c + 1] += off[c];
    int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* cur  = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_meso);
    memcpy(cur, off, sizeof(int64_t) * (size_t)n_meso);
    for (int64_t i = 0; i < n; ++i) perm[cur[mlab[i]]++] = i;
    float* mc_train    = (float*)malloc(sizeof(float) * (size_t)(msize_max > 0 ? msize_max : 1) * dim);
    float* mc_centers  = (float*)malloc(sizeof(float) * (size_t)(fmax > 0 ? fmax : 1) * dim);
    uint32_t* mc_lab   = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(msize_max > 0 ? msize_max : 1));
    uint32_t* mc_sizes = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(fmax > 0 ? fmax : 1));
    for (int i = 0; i < n_meso; i++) {
      int64_t kk = off[i + 1] - off[i];
      if (kk > msize_max) kk = msize_max;
      if (kk == 0) continue;
      for (int64_t r = 0; r < kk; ++r) memcpy(mc_train + r * dim, x + perm[off[i] + r] * dim, sizeof(float) * (size_t)dim);
      if (kk >= fine_nums[i]) {
        oracle_kmeans_build_clusters(mc_train, kk, dim, dim, (int)fine_nums[i], n_iters, mc_centers, mc_lab, mc_sizes);
      } else {
        for (int64_t c = 0; c < fine_nums[i]; ++c)
          memcpy(mc_centers + c * dim, mc_train + (c % kk) * dim, sizeof(float) * (size_t)dim);
      }
      memcpy(centers + fine_csum[i] * dim, mc_centers, sizeof(float) * (size_t)fine_nums[i] * dim);
    }
    uint32_t fin = (uint32_t)(n_iters / 10 > 2 ? n_iters / 10 : 2);
    balancing_em_iters(fin, dim, x, dim, n, k, centers, labels, sizes, 5, 0.2f, &i_primes);
    free(mlab); free(msizes); free(mcent); free(fine_nums); free(fine_csum); free(off); free(perm); free(cur);
    free(mc_train); free(mc_centers); free(mc_lab); free(mc_sizes);
  }
  if (out_labels) predict(x, n, dim, dim, centers, k, out_labels);
  free(labels); free(sizes);
}
