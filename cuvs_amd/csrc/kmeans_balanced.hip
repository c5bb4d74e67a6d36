// Balanced hierarchical k-means (coarse quantizer + PQ codebooks).
// Restates cpp/src/cluster/detail/kmeans_balanced.cuh: build_clusters (:724-783: labels i mod k, EM with
// rebalancing), balancing_em_iters (:645-722), adjust_centers (:464-580, kAdjustCentersWeight = 7),
// calc_centers_and_sizes (:253-315), arrange_fine_clusters (:786-848), build_hierarchical (:986-1148).
// MI355X design: E-step = fp32-MFMA fused argmin (distance.hip); M-step = rows grouped by label with a
// stable radix sort, then one workgroup per cluster sums its rows in a fixed order (no float atomics:
// centroids are reproducible run to run, unlike the reference's reduce_rows_by_key/atomic kernels).
#include "ops.hpp"
#include "device_utils.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <numeric>

namespace cuvs_amd {

namespace {

__global__ void iota_mod_kernel(uint32_t* labels, int64_t n, uint32_t mod)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) labels[i] = (uint32_t)(i % mod);
}
__global__ void iota_kernel(uint32_t* v, int64_t n)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}
__global__ void histogram_kernel(const uint32_t* labels, int64_t n, uint32_t* counts)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&counts[labels[i]], 1u);
}
// exclusive scan of counts[n] -> offsets[n+1]; single workgroup (n <= a few 100k)
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const uint32_t* counts, int n, uint32_t* offsets)
{
  __shared__ int smem[17];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < n ? (int)counts[i] : 0;
    int total;
    int excl = block_exclusive_scan(v, smem, &total);
    if (i < n) offsets[i] = (uint32_t)(carry + excl);
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[n] = (uint32_t)carry;
}

// One workgroup per cluster: mean of the rows perm[offsets[c] .. offsets[c+1]) in a fixed order. Row j of the cluster
// goes to slot (j - b) % S; a slot adds its rows in ascending order and the S partial sums are combined
// (((p0 + p1) + p2) + ...) - oracle_kmeans.c restates exactly this. S = 4 waves x R rows per wave step, R = 64 / dp for
// dim <= 32 (dp = dim rounded up to a power of two): the 12-dimensional rows of PQ codebook training used 12 of 64
// lanes per step. Four row loads are in flight per lane (gathers of short rows are latency-bound).
// Empty cluster -> zeros (reference: div_checkzero_op).
__global__ __launch_bounds__(256) void cluster_means_kernel(const float* __restrict__ x, int64_t ld, int dim,
                                                            const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ offsets,
                                                            float* __restrict__ centers,
                                                            uint32_t* __restrict__ sizes)
{
  __shared__ float part[4][64];
  const int c      = blockIdx.x;
  const int lane   = threadIdx.x & 63;
  const int wave   = threadIdx.x >> 6;
  const uint32_t b = offsets[c], e = offsets[c + 1];
  const uint32_t cnt = e - b;
  if (threadIdx.x == 0) sizes[c] = cnt;
  int dp = 64, R = 1;
  if (dim <= 32) {
    dp = 1;
    while (dp < dim) dp <<= 1;
    R = 64 / dp;
  }
  const uint32_t S = 4u * (uint32_t)R;
  const int r = lane / dp, dl = lane % dp;
  for (int d0 = 0; d0 < dim; d0 += dp) {
    const int d = d0 + dl;
    float acc   = 0.f;
    if (d < dim) {
      uint32_t j = b + (uint32_t)(wave * R + r);
      for (; j + 3 * S < e; j += 4 * S) {
        const uint32_t p0 = perm[j], p1 = perm[j + S], p2 = perm[j + 2 * S], p3 = perm[j + 3 * S];
        const float v0 = x[(int64_t)p0 * ld + d], v1 = x[(int64_t)p1 * ld + d], v2 = x[(int64_t)p2 * ld + d],
                    v3 = x[(int64_t)p3 * ld + d];
        acc += v0; acc += v1; acc += v2; acc += v3;
      }
      for (; j < e; j += S) acc += x[(int64_t)perm[j] * ld + d];
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && r == 0 && d < dim) {
      float s = part[0][dl];
      for (uint32_t slot = 1; slot < S; ++slot) s = s + part[slot / R][(slot % R) * dp + dl];
      centers[(int64_t)c * dim + d] = cnt ? s / (float)cnt : 0.f;
    }
    __syncthreads();
  }
}

// kmeans_balanced.cuh:464-511 restated; the pseudo-random row is chosen from a per-cluster sequence
// (seed * (l + 1 + t * n_clusters)) % n_rows instead of a shared atomic counter, which keeps the choice
// reproducible. One wave per small cluster.
__global__ __launch_bounds__(256) void adjust_centers_kernel(float* centers, int n_clusters, int dim,
                                                             const float* __restrict__ x, int64_t ld,
                                                             int64_t n_rows, const uint32_t* __restrict__ labels,
                                                             const uint32_t* __restrict__ sizes, float threshold,
                                                             int64_t average, int64_t seed, int* adjusted)
{
  const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (l >= n_clusters) return;
  const int lane     = threadIdx.x & 63;
  const int64_t csz  = sizes[l];
  if ((float)csz > (float)average * threshold) return;
  int64_t i = 0;
  if (lane == 0) {
    int64_t t = 0;
    do {
      i = (int64_t)(((unsigned long long)seed * (unsigned long long)(l + 1 + t * (int64_t)n_clusters)) %
                    (unsigned long long)n_rows);
      ++t;
    } while ((int64_t)sizes[labels[i]] < average && t < 100000);
    *adjusted = 1;
  }
  i = __shfl(i, 0, kWave);
  const int64_t li = labels[i];
  const float wc   = fminf((float)csz, 7.0f);
  const float wd   = 1.0f;
  for (int j = lane; j < dim; j += kWave) {
    float val = 0.f;
    val += wc * centers[j + (int64_t)dim * li];
    val += wd * x[j + ld * i];
    val /= wc + wd;
    centers[j + (int64_t)dim * l] = val;
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ x, int64_t ld, int dim,
                                   const uint32_t* __restrict__ ids, int64_t n, float* __restrict__ out)
{
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* src = x + (int64_t)ids[row] * ld;
  for (int d = threadIdx.x & 63; d < dim; d += 64) out[row * dim + d] = src[d];
}

inline unsigned blocks_for(int64_t n, int per) { return grid_blocks(n, per); }

int bits_for(uint32_t n)
{
  int b = 1;
  while ((1ull << b) < n) ++b;
  return b;
}

}  // namespace

// stable grouping of rows by label: perm[n] (row ids ordered by (label, row)), offsets[n_clusters+1]
void group_by_label(resources& res, const uint32_t* labels, int64_t n, uint32_t n_clusters, uint32_t* perm,
                    uint32_t* offsets)
{
  CUVS_EXPECTS(n < (int64_t(1) << 32), "group_by_label: more than 2^32 rows");
  dev_buf<uint32_t> counts(res, n_clusters);
  HIP_TRY(hipMemsetAsync(counts.data(), 0, counts.bytes(), res.stream));
  hipLaunchKernelGGL(histogram_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, res.stream, labels, n, counts.data());
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, res.stream, counts.data(), (int)n_clusters,
                     offsets);
  dev_buf<uint32_t> keys_out(res, n), vals_in(res, n);
  hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, res.stream, vals_in.data(), n);
  size_t temp_bytes = 0;
  int end_bit       = bits_for(n_clusters);
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, labels, keys_out.data(), vals_in.data(), perm,
                                             (size_t)n, 0, end_bit, res.stream));
  dev_buf<char> temp(res, temp_bytes);
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.data(), temp_bytes, labels, keys_out.data(), vals_in.data(), perm,
                                             (size_t)n, 0, end_bit, res.stream));
}

namespace {

void calc_centers_and_sizes(resources& res, const float* x, int64_t n, int64_t ld, int dim, int n_clusters,
                            const uint32_t* labels, float* centers, uint32_t* sizes)
{
  dev_buf<uint32_t> perm(res, n), offsets(res, n_clusters + 1);
  group_by_label(res, labels, n, n_clusters, perm.data(), offsets.data());
  hipLaunchKernelGGL(cluster_means_kernel, dim3(n_clusters), dim3(256), 0, res.stream, x, ld, dim, perm.data(),
                     offsets.data(), centers, sizes);
}

void predict_f32(resources& res, const float* x, int64_t n, int64_t ld, int dim, const float* centers,
                 int n_clusters, uint32_t* labels, bool inner_product = false)
{
  dev_buf<float> cn(res, n_clusters);
  // argmin_c (|c|^2 - 2 x.c); with |c|^2 := 0 it is argmax_c x.c, the inner-product assignment
  if (inner_product) HIP_TRY(hipMemsetAsync(cn.data(), 0, cn.bytes(), res.stream));
  else               row_norms<float>(res, centers, n_clusters, dim, dim, cn.data(), false);
  fused_l2_argmin<float>(res, x, n, ld, centers, n_clusters, dim, cn.data(), labels, nullptr);
}

bool adjust_centers(resources& res, float* centers, int n_clusters, int dim, const float* x, int64_t ld,
                    int64_t n_rows, const uint32_t* labels, const uint32_t* sizes, float threshold, int* d_flag,
                    int& i_primes)
{
  if (n_clusters == 0) return false;
  static const int kPrimes[] = {29,   71,   113,  173,  229,  281,  349,  409,  463,  541,  601,  659,  733,  809,
                                863,  941,  1013, 1069, 1151, 1223, 1291, 1373, 1451, 1511, 1583, 1657, 1733, 1811,
                                1889, 1987, 2053, 2129, 2213, 2287, 2357, 2423, 2531, 2617, 2687, 2741};
  const int n_primes = sizeof(kPrimes) / sizeof(int);
  int64_t average    = n_rows / n_clusters;
  int64_t ofst;
  do {
    i_primes = (i_primes + 1) % n_primes;
    ofst     = kPrimes[i_primes];
  } while (n_rows % ofst == 0);
  HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), res.stream));
  hipLaunchKernelGGL(adjust_centers_kernel, dim3(blocks_for(n_clusters, 4)), dim3(256), 0, res.stream, centers,
                     n_clusters, dim, x, ld, n_rows, labels, sizes, threshold, average, ofst, d_flag);
  int h = 0;
  copy_async(res, &h, d_flag, sizeof(int));
  sync(res);
  return h != 0;
}

void balancing_em_iters(resources& res, uint32_t n_iters, int dim, const float* x, int64_t ld, int64_t n_rows,
                        int n_clusters, float* centers, uint32_t* labels, uint32_t* sizes,
                        uint32_t balancing_pullback, float balancing_threshold, int& i_primes, bool inner_product = false)
{
  dev_buf<int> flag(res, 1);
  uint32_t balancing_counter = balancing_pullback;
  for (uint32_t iter = 0; iter < n_iters; iter++) {
    if (iter > 0 && adjust_centers(res, centers, n_clusters, dim, x, ld, n_rows, labels, sizes,
                                   balancing_threshold, flag.data(), i_primes)) {
      if (balancing_counter++ >= balancing_pullback) {
        balancing_counter -= balancing_pullback;
        n_iters++;
      }
    }
    if (inner_product) normalize_rows(res, centers, n_clusters, dim);  // kmeans_balanced.cuh:682-696
    predict_f32(res, x, n_rows, ld, dim, centers, n_clusters, labels, inner_product);
    calc_centers_and_sizes(res, x, n_rows, ld, dim, n_clusters, labels, centers, sizes);
  }
}

}  // namespace

// kmeans_balanced.cuh:724-783
void kmeans_build_clusters(resources& res, const float* x, int64_t n_rows, int64_t ld, int dim, int n_clusters,
                           int n_iters, float* centers, uint32_t* labels, uint32_t* sizes, bool inner_product)
{
  int i_primes = 0;
  hipLaunchKernelGGL(iota_mod_kernel, dim3(blocks_for(n_rows, 256)), dim3(256), 0, res.stream, labels, n_rows,
                     (uint32_t)n_clusters);
  calc_centers_and_sizes(res, x, n_rows, ld, dim, n_clusters, labels, centers, sizes);
  balancing_em_iters(res, n_iters, dim, x, ld, n_rows, n_clusters, centers, labels, sizes, 2, 0.25f, i_primes, inner_product);
}

void kmeans_balanced_fit(resources& res, const float* x, int64_t n_rows, int64_t dim64, int n_clusters,
                         const kmeans_params& p, float* centers)
{
  const int dim = (int)dim64;
  CUVS_EXPECTS(n_rows >= n_clusters, "kmeans: number of rows (%ld) can't be less than n_clusters (%d)",
               (long)n_rows, n_clusters);
  const int n_meso = std::min<int>(n_clusters, (int)(std::sqrt((double)n_clusters) + 0.5));
  if (!p.hierarchical || n_meso <= 1 || n_meso == n_clusters) {
    dev_buf<uint32_t> labels(res, n_rows), sizes(res, n_clusters);
    kmeans_build_clusters(res, x, n_rows, dim, dim, n_clusters, p.n_iters, centers, labels.data(), sizes.data(), p.inner_product);
    return;
  }
  int i_primes = 0;
  // ---- mesoclusters
  dev_buf<uint32_t> meso_labels(res, n_rows), meso_sizes_d(res, n_meso);
  {
    dev_buf<float> meso_centers(res, (size_t)n_meso * dim);
    kmeans_build_clusters(res, x, n_rows, dim, dim, n_meso, p.n_iters, meso_centers.data(), meso_labels.data(),
                          meso_sizes_d.data(), p.inner_product);
  }
  std::vector<uint32_t> meso_sizes = to_host(res, meso_sizes_d.data(), n_meso);

  // ---- arrange_fine_clusters (kmeans_balanced.cuh:786-848)
  std::vector<int64_t> fine_nums(n_meso), fine_csum(n_meso + 1, 0);
  int64_t n_lists_rem = n_clusters, n_nonempty_rem = 0, n_rows_rem = n_rows;
  for (int i = 0; i < n_meso; i++) n_nonempty_rem += meso_sizes[i] > 0 ? 1 : 0;
  int64_t meso_size_max = 0, fine_nums_max = 0;
  for (int i = 0; i < n_meso; i++) {
    if (i < n_meso - 1) {
      if (meso_sizes[i] == 0) {
        fine_nums[i] = 0;
      } else {
        n_nonempty_rem--;
        auto s = (int64_t)((double)(n_lists_rem * (int64_t)meso_sizes[i]) / (double)n_rows_rem + .5);
        s            = std::min<int64_t>(s, n_lists_rem - n_nonempty_rem);
        fine_nums[i] = std::max<int64_t>(s, 1);
      }
    } else {
      fine_nums[i] = n_lists_rem;
    }
    n_lists_rem -= fine_nums[i];
    n_rows_rem -= meso_sizes[i];
    meso_size_max    = std::max<int64_t>(meso_size_max, meso_sizes[i]);
    fine_nums_max    = std::max<int64_t>(fine_nums_max, fine_nums[i]);
    fine_csum[i + 1] = fine_csum[i] + fine_nums[i];
  }
  CUVS_EXPECTS(fine_csum[n_meso] == n_clusters, "fine cluster numbers do not add up");
  const int64_t meso_size_max_balanced = (2 * n_rows + n_meso - 1) / std::max(n_meso, 1);
  meso_size_max = std::min(meso_size_max, meso_size_max_balanced);

  // ---- fine clusters inside every mesocluster (:851-984)
  {
    dev_buf<uint32_t> perm(res, n_rows), offsets_d(res, n_meso + 1);
    group_by_label(res, meso_labels.data(), n_rows, n_meso, perm.data(), offsets_d.data());
    std::vector<uint32_t> offsets = to_host(res, offsets_d.data(), n_meso + 1);
    dev_buf<float> mc_train(res, (size_t)meso_size_max * dim);
    dev_buf<float> mc_centers(res, (size_t)fine_nums_max * dim);
    dev_buf<uint32_t> mc_labels(res, meso_size_max), mc_sizes(res, fine_nums_max);
    for (int i = 0; i < n_meso; i++) {
      int64_t k = std::min<int64_t>(offsets[i + 1] - offsets[i], meso_size_max);
      if (k == 0) {
        CUVS_EXPECTS(fine_nums[i] == 0, "non-zero fine clusters for an empty mesocluster");
        continue;
      }
      CUVS_EXPECTS(fine_nums[i] > 0, "zero fine clusters for a non-empty mesocluster");
      hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(k, 4)), dim3(256), 0, res.stream, x, (int64_t)dim, dim,
                         perm.data() + offsets[i], k, mc_train.data());
      if (k >= fine_nums[i]) {
        kmeans_build_clusters(res, mc_train.data(), k, dim, dim, (int)fine_nums[i], p.n_iters, mc_centers.data(),
                              mc_labels.data(), mc_sizes.data(), p.inner_product);
      } else {
        // fewer training rows than fine clusters: seed the centres cyclically from the rows
        for (int64_t c = 0; c < fine_nums[i]; ++c)
          copy_async(res, mc_centers.data() + c * dim, mc_train.data() + (c % k) * dim, dim * sizeof(float));
      }
      copy_async(res, centers + fine_csum[i] * dim, mc_centers.data(), (size_t)fine_nums[i] * dim * sizeof(float));
    }
  }

  // ---- final balancing EM over all centres (:1113-1127)
  dev_buf<uint32_t> labels(res, n_rows), sizes(res, n_clusters);
  balancing_em_iters(res, std::max<uint32_t>(p.n_iters / 10, 2), dim, x, dim, n_rows, n_clusters, centers,
                     labels.data(), sizes.data(), 5, 0.2f, i_primes, p.inner_product);
}

template <typename T>
void kmeans_predict(resources& res, const T* x, int64_t n, int64_t dim, const float* centers, int n_clusters,
                    uint32_t* labels)
{
  dev_buf<float> cn(res, n_clusters);
  row_norms<float>(res, centers, n_clusters, dim, dim, cn.data(), false);
  fused_l2_argmin<T>(res, x, n, dim, centers, n_clusters, dim, cn.data(), labels, nullptr);
}
template void kmeans_predict<float>(resources&, const float*, int64_t, int64_t, const float*, int, uint32_t*);

}  // namespace cuvs_amd

// Test hooks (not part of the reference ABI): fit + predict on device float rows.
extern "C" __attribute__((visibility("default"))) int cuvsAmdKMeansBalancedFit(uintptr_t res, const float* x,
                                                                                int64_t n, int64_t dim,
                                                                                int n_clusters, int n_iters,
                                                                                int hierarchical, float* centers,
                                                                                uint32_t* labels)
{
  using namespace cuvs_amd;
  return translate_exceptions([=] {
    kmeans_params p;
    p.n_iters      = n_iters;
    p.hierarchical = hierarchical != 0;
    kmeans_balanced_fit(*as_res(res), x, n, dim, n_clusters, p, centers);
    if (labels) kmeans_predict<float>(*as_res(res), x, n, dim, centers, n_clusters, labels);
  });
}
