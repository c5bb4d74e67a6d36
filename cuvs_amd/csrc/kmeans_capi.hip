// cuvsKMeans* C entry points (include/cuvs/cluster/kmeans.h; reference wrapper c/src/cluster/kmeans.cpp).
//   hierarchical = true  -> kmeans_balanced_fit / kmeans_predict (kmeans_balanced.hip)
//   hierarchical = false -> Lloyd iterations below: E-step on the fp32-MFMA fused argmin (distance.hip), M-step =
//                           rows grouped by label (stable radix sort) and one workgroup per cluster summing its rows
//                           in a fixed order, so centroids are reproducible run to run. Stopping rule and the order
//                           cost -> new centroids -> shift follow cpp/src/cluster/detail/kmeans.cuh:813-925 and
//                           kmeans_common.cuh:629-648; empty clusters keep their centroid (kmeans_common.cuh:585-600).
// Seeding: `Array` copies the caller's centroids, `Random` draws distinct rows, `KMeansPlusPlus` is sequential
// k-means++ by an exponential race (argmax of w*d^2 / Exp(1) per round, one pass over the rows per centroid); the
// reference's scalable k-means|| (kmeans.cuh:305-470) consumes RAFT's generator and is not reproducible outside it.
#include "ops.hpp"
#include "device_utils.hpp"

#include <cuvs/cluster/kmeans.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <random>
#include <unordered_set>
#include <vector>

namespace cuvs_amd {
namespace {

__device__ inline double wave_sum_f64(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ inline uint64_t mix64(uint64_t z)
{
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// One wave per row: out[row] = sum_d (x[row][d] - c[d])^2 for the centroid the row is measured against
// (labels == nullptr: the single centroid `c`; else c + labels[row] * dim).
__device__ inline float row_sqdist(const float* __restrict__ xr, const float* __restrict__ cr, int dim, int lane)
{
  float acc = 0.f;
  for (int d = lane; d < dim; d += kWave) {
    float t = xr[d] - cr[d];
    acc     = __fmaf_rn(t, t, acc);
  }
  return wave_sum(acc);
}

// partial[b] = sum over the rows of block b of w * |x - c_label|^2 in double (fixed order: 4 waves x rows, then waves).
__global__ __launch_bounds__(256) void cost_partials_kernel(const float* __restrict__ x, int64_t n, int dim,
                                                            const float* __restrict__ centers,
                                                            const uint32_t* __restrict__ labels,
                                                            const float* __restrict__ w, double* __restrict__ partial)
{
  __shared__ double part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc     = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
    float d = row_sqdist(x + row * dim, centers + (int64_t)labels[row] * dim, dim, lane);
    acc += (double)(w ? w[row] * d : d);
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((part[0] + part[1]) + part[2]) + part[3];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ v, int64_t n,
                                                           double* __restrict__ partial)
{
  __shared__ double part[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)v[i];
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((part[0] + part[1]) + part[2]) + part[3];
}

__global__ void scale_kernel(const float* __restrict__ in, int64_t n, float s, float* __restrict__ out)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}

// M-step: one workgroup per cluster over perm[offsets[c] .. offsets[c+1]); weighted mean, the empty (zero weight)
// cluster keeps its centroid; shift[c] = |new - old|^2.
__global__ __launch_bounds__(256) void lloyd_means_kernel(const float* __restrict__ x, int dim,
                                                          const uint32_t* __restrict__ perm,
                                                          const uint32_t* __restrict__ offsets,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ cur, float* __restrict__ nxt,
                                                          float* __restrict__ shift)
{
  __shared__ float part[4][64];
  __shared__ float wpart[4];
  __shared__ float wsum_s;
  const int c      = blockIdx.x;
  const int lane   = threadIdx.x & 63;
  const int wave   = threadIdx.x >> 6;
  const uint32_t b = offsets[c], e = offsets[c + 1];
  {
    float ws = 0.f;
    for (uint32_t j = b + threadIdx.x; j < e; j += 256) ws += w ? w[perm[j]] : 1.f;
    ws = wave_sum(ws);
    if (lane == 0) wpart[wave] = ws;
    __syncthreads();
    if (threadIdx.x == 0) wsum_s = ((wpart[0] + wpart[1]) + wpart[2]) + wpart[3];
    __syncthreads();
  }
  const float wsum = wsum_s;
  float sh         = 0.f;
  for (int d0 = 0; d0 < dim; d0 += 64) {
    const int d = d0 + lane;
    float acc   = 0.f;
    if (d < dim) {
      for (uint32_t j = b + wave; j < e; j += 4) {
        const uint32_t r = perm[j];
        const float xv   = x[(int64_t)r * dim + d];
        acc += w ? w[r] * xv : xv;
      }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && d < dim) {
      const float s   = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
      const float old = cur[(int64_t)c * dim + d];
      const float nv  = wsum > 0.f ? s / wsum : old;
      nxt[(int64_t)c * dim + d] = nv;
      sh = __fmaf_rn(nv - old, nv - old, sh);
    }
    __syncthreads();
  }
  if (wave == 0) {
    sh = wave_sum(sh);
    if (lane == 0) shift[c] = sh;
  }
}

__global__ void gather_seed_rows_kernel(const float* __restrict__ x, int dim, const int64_t* __restrict__ ids, int k,
                                        float* __restrict__ out)
{
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= k) return;
  for (int d = threadIdx.x & 63; d < dim; d += 64) out[(int64_t)row * dim + d] = x[ids[row] * dim + d];
}

// k-means++ round `j`: mind[row] = min(mind[row], |x - centers[j]|^2) (round 0 initialises), then the block's best
// (w * mind / E, row) with E ~ Exp(1) from a counter hash of (seed, j, row): the arg max over all rows is a draw
// proportional to w * mind.
__global__ __launch_bounds__(256) void pp_round_kernel(const float* __restrict__ x, int64_t n, int dim,
                                                       const float* __restrict__ centers, int j,
                                                       const float* __restrict__ w, uint64_t seed,
                                                       float* __restrict__ mind, float* __restrict__ best_v,
                                                       int64_t* __restrict__ best_i)
{
  __shared__ float sv[4];
  __shared__ int64_t si[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float bv   = -1.f;
  int64_t bi = -1;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
    float d = row_sqdist(x + row * dim, centers + (int64_t)j * dim, dim, lane);
    if (j > 0) d = fminf(d, mind[row]);
    if (lane == 0) mind[row] = d;
    const uint64_t h = mix64(seed ^ mix64(((uint64_t)(j + 1) << 40) ^ (uint64_t)row));
    const float u    = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
    const float key  = (w ? w[row] * d : d) / -__logf(u);
    if (key > bv) { bv = key; bi = row; }  // rows visited in increasing order: ties keep the smaller row
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 4; ++t)
      if (sv[t] > bv || (sv[t] == bv && si[t] >= 0 && (bi < 0 || si[t] < bi))) { bv = sv[t]; bi = si[t]; }
    best_v[blockIdx.x] = bv;
    best_i[blockIdx.x] = bi;
  }
}

// picks the winner of the round and writes its row as centroid j + 1 (single workgroup)
__global__ __launch_bounds__(256) void pp_pick_kernel(const float* __restrict__ x, int dim,
                                                      const float* __restrict__ best_v,
                                                      const int64_t* __restrict__ best_i, int n_blocks, int j,
                                                      float* __restrict__ centers)
{
  __shared__ int64_t pick;
  if (threadIdx.x == 0) {
    float bv   = -1.f;
    int64_t bi = 0;
    for (int t = 0; t < n_blocks; ++t) {
      const int64_t i = best_i[t];
      if (i < 0) continue;
      if (best_v[t] > bv || (best_v[t] == bv && i < bi)) { bv = best_v[t]; bi = i; }
    }
    pick = bi;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < dim; d += 256) centers[(int64_t)(j + 1) * dim + d] = x[pick * dim + d];
}

double host_sum(resources& res, const double* d, size_t n)
{
  auto h   = to_host(res, d, n);
  double s = 0.0;
  for (double v : h) s += v;
  return s;
}

constexpr int kReduceBlocks = 1024;

// sum_i w_i * |x_i - centers[labels_i]|^2 (w == nullptr: unit weights)
double weighted_cost(resources& res, const float* x, int64_t n, int dim, const float* centers, const uint32_t* labels,
                     const float* w)
{
  if (n == 0) return 0.0;
  const int blocks = (int)std::min<int64_t>(kReduceBlocks, (n + 3) / 4);
  dev_buf<double> partial(res, blocks);
  hipLaunchKernelGGL(cost_partials_kernel, dim3(blocks), dim3(256), 0, res.stream, x, n, dim, centers, labels, w,
                     partial.data());
  return host_sum(res, partial.data(), blocks);
}

void assign(resources& res, const float* x, int64_t n, int dim, const float* centers, int k, uint32_t* labels)
{
  dev_buf<float> cn(res, k);
  row_norms<float>(res, centers, k, dim, dim, cn.data(), false);
  fused_l2_argmin<float>(res, x, n, dim, centers, k, dim, cn.data(), labels, nullptr);
}

// weights rescaled to sum to n (kmeans.cuh:713-726); returns nullptr for "no weights"
const float* normalized_weights(resources& res, const float* w, int64_t n, dev_buf<float>& storage)
{
  if (w == nullptr) return nullptr;
  const int blocks = (int)std::min<int64_t>(kReduceBlocks, (n + 255) / 256);
  dev_buf<double> partial(res, blocks);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks), dim3(256), 0, res.stream, w, n, partial.data());
  const double total = host_sum(res, partial.data(), blocks);
  CUVS_EXPECTS(total > 0.0 && std::isfinite(total), "kmeans: sample_weight must have a positive finite sum");
  storage = dev_buf<float>(res, n);
  hipLaunchKernelGGL(scale_kernel, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream, w, n,
                     (float)((double)n / total), storage.data());
  return storage.data();
}

struct lloyd_params {
  int n_clusters;
  int init;  // cuvsKMeansInitMethod
  int max_iter;
  double tol;
  int n_init;
  uint64_t seed = 0;  // the reference's default rng_state seed
};

void seed_centroids(resources& res, const float* x, int64_t n, int dim, const float* w, const lloyd_params& p,
                    uint64_t seed, float* centers)
{
  const int k = p.n_clusters;
  std::mt19937_64 gen(seed);
  if (p.init == Random) {
    // Floyd's algorithm: k distinct rows in O(k)
    std::unordered_set<int64_t> chosen;
    std::vector<int64_t> ids;
    for (int64_t j = n - k; j < n; ++j) {
      int64_t t = (int64_t)(gen() % (uint64_t)(j + 1));
      if (!chosen.insert(t).second) { chosen.insert(j); t = j; }
      ids.push_back(t);
    }
    std::sort(ids.begin(), ids.end());
    dev_buf<int64_t> d_ids(res, k);
    copy_async(res, d_ids.data(), ids.data(), k * sizeof(int64_t));
    hipLaunchKernelGGL(gather_seed_rows_kernel, dim3(grid_blocks(k, 4)), dim3(256), 0, res.stream, x, dim,
                       d_ids.data(), k, centers);
    sync(res);
    return;
  }
  // k-means++
  const int64_t first = (int64_t)(gen() % (uint64_t)n);
  copy_async(res, centers, x + first * dim, (size_t)dim * sizeof(float));
  const int blocks = (int)std::min<int64_t>(kReduceBlocks, (n + 3) / 4);
  dev_buf<float> mind(res, n), best_v(res, blocks);
  dev_buf<int64_t> best_i(res, blocks);
  const uint64_t round_seed = gen();
  for (int j = 0; j + 1 < k; ++j) {
    hipLaunchKernelGGL(pp_round_kernel, dim3(blocks), dim3(256), 0, res.stream, x, n, dim, centers, j, w, round_seed,
                       mind.data(), best_v.data(), best_i.data());
    hipLaunchKernelGGL(pp_pick_kernel, dim3(1), dim3(256), 0, res.stream, x, dim, best_v.data(), best_i.data(),
                       blocks, j, centers);
  }
  HIP_TRY(hipGetLastError());
  sync(res);
}

// x, centroids on device; w = normalized weights or nullptr
void lloyd_fit(resources& res, const float* x, int64_t n, int dim, const float* w, const lloyd_params& p,
               float* centroids, double* inertia, int* n_iter)
{
  const int k = p.n_clusters;
  CUVS_EXPECTS(k > 0, "invalid parameter (n_clusters<=0)");
  CUVS_EXPECTS(p.tol > 0, "invalid parameter (tol<=0)");
  CUVS_EXPECTS(p.max_iter >= 0, "invalid parameter (max_iter<0)");
  CUVS_EXPECTS(n >= k, "kmeans: number of samples (%ld) can't be less than n_clusters (%d)", (long)n, k);
  const int n_init = p.init == Array ? 1 : std::max(1, p.n_init);
  const size_t csz = (size_t)k * dim;
  dev_buf<float> buf_a(res, csz), buf_b(res, csz), shift(res, k);
  dev_buf<uint32_t> labels(res, n), perm(res, n), offsets(res, k + 1);
  std::mt19937_64 gen(p.seed);
  double best = std::numeric_limits<double>::max();
  for (int trial = 0; trial < n_init; ++trial) {
    float* cur = buf_a.data();
    float* nxt = buf_b.data();
    if (p.init == Array) {
      copy_async(res, cur, centroids, csz * sizeof(float));
    } else {
      seed_centroids(res, x, n, dim, w, p, gen(), cur);
    }
    double prior = 0.0;
    int iter     = 0;
    bool done    = false;
    for (iter = 1; iter <= p.max_iter && !done; ++iter) {
      assign(res, x, n, dim, cur, k, labels.data());
      const double cost = weighted_cost(res, x, n, dim, cur, labels.data(), w);
      group_by_label(res, labels.data(), n, (uint32_t)k, perm.data(), offsets.data());
      hipLaunchKernelGGL(lloyd_means_kernel, dim3(k), dim3(256), 0, res.stream, x, dim, perm.data(), offsets.data(),
                         w, cur, nxt, shift.data());
      HIP_TRY(hipGetLastError());
      auto h_shift = to_host(res, shift.data(), (size_t)k);
      double norm  = 0.0;
      for (float s : h_shift) norm += (double)s;
      std::swap(cur, nxt);
      // kmeans_common.cuh:629-648 (the reference evaluates this in the data type, float)
      if (cost != 0.0 && iter > 1 && (float)(cost / prior) > 1.0f - (float)p.tol) done = true;
      if ((float)norm < (float)p.tol) done = true;
      prior = cost;
    }
    const int ran = std::min(iter - 1, p.max_iter);
    assign(res, x, n, dim, cur, k, labels.data());
    const double trial_inertia = weighted_cost(res, x, n, dim, cur, labels.data(), w);
    if (trial_inertia < best) {
      best = trial_inertia;
      if (inertia) *inertia = trial_inertia;
      if (n_iter) *n_iter = ran;
      copy_async(res, centroids, cur, csz * sizeof(float));
      sync(res);
    }
  }
}

bool metric_ok(int m) { return metric_is_l2(m); }

void expect_f32_matrix(const DLTensor& t, const char* what)
{
  if (dtype_is(t.dtype, kDLFloat, 64)) CUVS_FAIL("float64 is an unsupported dtype for %s (float32 only)", what);
  CUVS_EXPECTS(dtype_is(t.dtype, kDLFloat, 32), "Unsupported %s DLtensor dtype: %d and bits: %d", what, (int)t.dtype.code,
               (int)t.dtype.bits);
  CUVS_EXPECTS(t.ndim == 2 && is_c_contiguous(t), "%s must be a row-major matrix", what);
}

const float* weight_ptr(DLManagedTensor* sw, int64_t n, bool need_device, resources& res, dev_buf<float>& staged)
{
  if (sw == nullptr) return nullptr;
  auto& t = sw->dl_tensor;
  CUVS_EXPECTS(dtype_is(t.dtype, kDLFloat, 32), "sample_weight must be float32");
  CUVS_EXPECTS(t.ndim == 1 && t.shape[0] == n, "sample_weight must have n_samples entries");
  if (is_device_accessible(t)) return static_cast<const float*>(dl_data(t));
  CUVS_EXPECTS(!need_device && is_host_accessible(t), "sample_weight must be host accessible when X is on host");
  staged = dev_buf<float>(res, n);
  copy_async(res, staged.data(), dl_data(t), n * sizeof(float));
  return staged.data();
}

template <typename P>
void fit_impl(cuvsResources_t res_h, const P& params, DLManagedTensor* X, DLManagedTensor* sample_weight,
              DLManagedTensor* centroids, double* inertia, int* n_iter)
{
  auto& res = *as_res(res_h);
  CUVS_EXPECTS(X && centroids && inertia && n_iter, "null argument");
  auto& x = X->dl_tensor;
  auto& c = centroids->dl_tensor;
  expect_f32_matrix(x, "dataset");
  expect_f32_matrix(c, "centroids");
  CUVS_EXPECTS(is_device_accessible(c), "centroids must be on device memory");
  const int64_t n = x.shape[0], dim = x.shape[1];
  CUVS_EXPECTS(c.shape[0] == params.n_clusters && c.shape[1] == dim,
               "centroids must be [n_clusters, n_features] = [%d, %ld]", params.n_clusters, (long)dim);
  CUVS_EXPECTS(metric_ok((int)params.metric), "kmeans: unsupported metric %d (L2 family only)", (int)params.metric);
  float* cptr         = static_cast<float*>(dl_data(c));
  const bool x_device = is_device_accessible(x);
  if (params.hierarchical) {
    if (!x_device) CUVS_FAIL("hierarchical kmeans is not supported with host data");
    if (sample_weight != nullptr) CUVS_FAIL("sample_weight cannot be used with hierarchical kmeans");
    const float* xp = static_cast<const float*>(dl_data(x));
    kmeans_params kp;
    kp.n_iters      = params.hierarchical_n_iters;
    kp.hierarchical = true;
    kmeans_balanced_fit(res, xp, n, dim, params.n_clusters, kp, cptr);
    dev_buf<uint32_t> labels(res, n);
    assign(res, xp, n, (int)dim, cptr, params.n_clusters, labels.data());
    *inertia = weighted_cost(res, xp, n, (int)dim, cptr, labels.data(), nullptr);
    *n_iter  = params.hierarchical_n_iters;
    return;
  }
  dev_buf<float> x_staged, w_staged, w_norm;
  const float* xp;
  if (x_device) {
    xp = static_cast<const float*>(dl_data(x));
  } else {
    CUVS_EXPECTS(is_host_accessible(x), "X must be host or device accessible");
    x_staged = dev_buf<float>(res, (size_t)n * dim);
    copy_async(res, x_staged.data(), dl_data(x), (size_t)n * dim * sizeof(float));
    xp = x_staged.data();
  }
  const float* w_raw = weight_ptr(sample_weight, n, x_device, res, w_staged);
  const float* w     = normalized_weights(res, w_raw, n, w_norm);
  lloyd_params lp{params.n_clusters, (int)params.init, params.max_iter, params.tol, params.n_init};
  lloyd_fit(res, xp, n, (int)dim, w, lp, cptr, inertia, n_iter);
}

template <typename P>
void predict_impl(cuvsResources_t res_h, const P& params, DLManagedTensor* X, DLManagedTensor* sample_weight,
                  DLManagedTensor* centroids, DLManagedTensor* labels, bool normalize_weight, double* inertia)
{
  auto& res = *as_res(res_h);
  CUVS_EXPECTS(X && centroids && labels && inertia, "null argument");
  auto& x = X->dl_tensor;
  auto& c = centroids->dl_tensor;
  auto& l = labels->dl_tensor;
  CUVS_EXPECTS(is_device_accessible(x), "X dataset must be accessible on device memory");
  expect_f32_matrix(x, "dataset");
  expect_f32_matrix(c, "centroids");
  CUVS_EXPECTS(is_device_accessible(c) && is_device_accessible(l), "centroids and labels must be on device memory");
  const int64_t n = x.shape[0], dim = x.shape[1];
  const int k     = (int)c.shape[0];
  CUVS_EXPECTS(c.shape[1] == dim && k > 0, "centroids must be [n_clusters, n_features]");
  CUVS_EXPECTS((l.dtype.code == kDLInt || l.dtype.code == kDLUInt) && l.dtype.bits == 32 && l.dtype.lanes == 1,
               "labels must be int32");
  CUVS_EXPECTS(l.ndim == 1 && l.shape[0] == n, "labels must have n_samples entries");
  CUVS_EXPECTS(metric_ok((int)params.metric), "kmeans: unsupported metric %d (L2 family only)", (int)params.metric);
  const float* xp = static_cast<const float*>(dl_data(x));
  const float* cp = static_cast<const float*>(dl_data(c));
  uint32_t* lp    = static_cast<uint32_t*>(dl_data(l));
  if (params.hierarchical) {
    if (sample_weight != nullptr) CUVS_FAIL("sample_weight cannot be used with hierarchical kmeans");
    kmeans_predict<float>(res, xp, n, dim, cp, k, lp);
    *inertia = 0;
    return;
  }
  dev_buf<float> w_staged, w_norm;
  const float* w = weight_ptr(sample_weight, n, true, res, w_staged);
  if (w != nullptr && normalize_weight) w = normalized_weights(res, w, n, w_norm);
  assign(res, xp, n, (int)dim, cp, k, lp);
  *inertia = weighted_cost(res, xp, n, (int)dim, cp, lp, w);
}

template <typename P>
P* default_params()
{
  P* p                    = new P{};
  p->metric               = L2Expanded;
  p->n_clusters           = 8;
  p->init                 = KMeansPlusPlus;
  p->max_iter             = 300;
  p->tol                  = 1e-4;
  p->n_init               = 1;
  p->oversampling_factor  = 2.0;
  p->batch_samples        = 1 << 15;
  p->batch_centroids      = 0;
  p->hierarchical         = false;
  p->hierarchical_n_iters = 20;
  p->streaming_batch_size = 0;
  p->init_size            = 0;
  return p;
}

}  // namespace
}  // namespace cuvs_amd

using namespace cuvs_amd;

extern "C" cuvsError_t cuvsKMeansParamsCreate(cuvsKMeansParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "null argument");
    *params = default_params<cuvsKMeansParams>();
  });
}
extern "C" cuvsError_t cuvsKMeansParamsDestroy(cuvsKMeansParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
extern "C" cuvsError_t cuvsKMeansParamsCreate_v2(cuvsKMeansParams_v2_t* params)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "null argument");
    *params = default_params<cuvsKMeansParams_v2>();
  });
}
extern "C" cuvsError_t cuvsKMeansParamsDestroy_v2(cuvsKMeansParams_v2_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}

extern "C" cuvsError_t cuvsKMeansFit(cuvsResources_t res, cuvsKMeansParams_t params, DLManagedTensor* X,
                                     DLManagedTensor* sample_weight, DLManagedTensor* centroids, double* inertia,
                                     int* n_iter)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "null argument");
    fit_impl(res, *params, X, sample_weight, centroids, inertia, n_iter);
  });
}
extern "C" cuvsError_t cuvsKMeansFit_v2(cuvsResources_t res, cuvsKMeansParams_v2_t params, DLManagedTensor* X,
                                        DLManagedTensor* sample_weight, DLManagedTensor* centroids, double* inertia,
                                        int* n_iter)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "null argument");
    fit_impl(res, *params, X, sample_weight, centroids, inertia, n_iter);
  });
}

extern "C" cuvsError_t cuvsKMeansPredict(cuvsResources_t res, cuvsKMeansParams_t params, DLManagedTensor* X,
                                         DLManagedTensor* sample_weight, DLManagedTensor* centroids,
                                         DLManagedTensor* labels, bool normalize_weight, double* inertia)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "null argument");
    predict_impl(res, *params, X, sample_weight, centroids, labels, normalize_weight, inertia);
  });
}
extern "C" cuvsError_t cuvsKMeansPredict_v2(cuvsResources_t res, cuvsKMeansParams_v2_t params, DLManagedTensor* X,
                                            DLManagedTensor* sample_weight, DLManagedTensor* centroids,
                                            DLManagedTensor* labels, bool normalize_weight, double* inertia)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "null argument");
    predict_impl(res, *params, X, sample_weight, centroids, labels, normalize_weight, inertia);
  });
}

extern "C" cuvsError_t cuvsKMeansClusterCost(cuvsResources_t res_h, DLManagedTensor* X, DLManagedTensor* centroids,
                                             double* cost)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(X && centroids && cost, "null argument");
    auto& x = X->dl_tensor;
    auto& c = centroids->dl_tensor;
    CUVS_EXPECTS(is_device_accessible(x), "X dataset must be accessible on device memory");
    expect_f32_matrix(x, "dataset");
    expect_f32_matrix(c, "centroids");
    CUVS_EXPECTS(is_device_accessible(c), "centroids must be on device memory");
    const int64_t n = x.shape[0], dim = x.shape[1];
    CUVS_EXPECTS(c.shape[1] == dim && c.shape[0] > 0, "centroids must be [n_clusters, n_features]");
    const float* xp = static_cast<const float*>(dl_data(x));
    const float* cp = static_cast<const float*>(dl_data(c));
    dev_buf<uint32_t> labels(res, n);
    assign(res, xp, n, (int)dim, cp, (int)c.shape[0], labels.data());
    *cost = weighted_cost(res, xp, n, (int)dim, cp, labels.data(), nullptr);
  });
}
