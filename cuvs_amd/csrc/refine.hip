// Exact re-ranking of candidate lists (reference: cpp/src/neighbors/refine/refine_device.cuh builds a
// throw-away IVF-Flat index to reuse its scan kernel; refine_host.hpp:353-462 is the CPU form).
// Here: one 256-thread workgroup per query; each wave takes candidates round-robin, its 64 lanes stride over
// the row (coalesced 256-byte reads) and butterfly-reduce; the n_cand exact distances are then bitonic-sorted
// in LDS by (distance, id) — the refine_host ordering (std::sort of (distance, id) tuples, :430-460).
#include "ops.hpp"
#include "device_utils.hpp"

#include <cuvs/neighbors/refine.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <thread>
#include <vector>

namespace cuvs_amd {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void refine_kernel(const T* __restrict__ data, int64_t n, int64_t dim,
                                                     const T* __restrict__ queries,
                                                     const int64_t* __restrict__ cand, int n_cand, int np2, int k,
                                                     int metric, int64_t* __restrict__ out_i,
                                                     float* __restrict__ out_d)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* s_idx  = reinterpret_cast<int64_t*>(smem);           // np2
  uint32_t* s_key = reinterpret_cast<uint32_t*>(s_idx + np2);   // np2
  const int64_t q = blockIdx.x;
  const int lane  = threadIdx.x & 63;
  const int wave  = threadIdx.x >> 6;
  const T* qv     = queries + q * dim;
  const bool ip   = metric == M_InnerProduct;
  const bool cosm = metric == M_CosineExpanded;
  for (int c = threadIdx.x; c < np2; c += 256) { s_key[c] = 0xffffffffu; s_idx[c] = INT64_MAX; }
  __syncthreads();
  float qn = 0.f;  // cosine: |q| with the canonical strided partial sums (every wave computes the same value)
  if (cosm) {
    for (int64_t j = lane; j < dim; j += kWave) { float a = to_float(qv[j]); qn = __fmaf_rn(a, a, qn); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) qn = qn + __shfl_xor(qn, off, kWave);
    qn = sqrtf(qn);
  }
  // Four candidates of a wave at a time: a candidate is ONE memory round trip (its row's loads are issued together) and a wave that
  // takes them one after the other spends its time waiting for HBM (a CAGRA build's refine of 256 candidates per row: 3.9 ms per
  // 16384 queries). The sums are the same strided partial sums per row, row by row.
  constexpr int U = 4;
  for (int c0 = wave; c0 < n_cand; c0 += 4 * U) {
    int64_t id[U];
    bool valid[U];
    float acc[U], xn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      id[u]    = c < n_cand ? cand[q * n_cand + c] : -1;
      valid[u] = c < n_cand && id[u] >= 0 && id[u] < n;  // wave-uniform
      acc[u] = 0.f; xn[u] = 0.f;
    }
    for (int64_t j = lane; j < dim; j += kWave) {
      const float a = to_float(qv[j]);
      float b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) b[u] = to_float(data[(valid[u] ? id[u] : 0) * dim + j]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (cosm) {
          acc[u] = __fmaf_rn(a, b[u], acc[u]);
          xn[u]  = __fmaf_rn(b[u], b[u], xn[u]);
        } else if (ip) {
          acc[u] = __fmaf_rn(a, b[u], acc[u]);
        } else {
          const float t = a - b[u];
          acc[u]        = __fmaf_rn(t, t, acc[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      if (c >= n_cand) continue;  // wave-uniform
      if (!valid[u]) {  // refine_host.hpp:440-442: the candidate stays, with distance = max and its own id
        if (lane == 0) {
          s_key[c] = float_to_key(FLT_MAX);  // (inner product: the sort key is -q.x there too - max sorts last under either rule)
          s_idx[c] = id[u];
        }
        continue;
      }
      float av = acc[u];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) av = av + __shfl_xor(av, off, kWave);
      if (cosm) {
        float xv = xn[u];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) xv = xv + __shfl_xor(xv, off, kWave);
        av = 1.0f - av / (qn * sqrtf(xv));  // the brute-force epilogue (distance_tile.hpp finish_distance)
      }
      if (lane == 0) {
        s_key[c] = ip ? ~float_to_key(av) : float_to_key(av);  // inner product: larger is better
        s_idx[c] = id[u];
      }
    }
  }
  __syncthreads();
  block_bitonic_sort<int64_t>(s_key, s_idx, np2);
  for (int j = threadIdx.x; j < k; j += 256) {  // (k <= n_cand: every slot read here holds a candidate)
    const bool bad = s_key[j] == float_to_key(FLT_MAX) && (s_idx[j] < 0 || s_idx[j] >= n);
    uint32_t key   = ip ? ~s_key[j] : s_key[j];
    float d        = bad ? (ip ? -FLT_MAX : FLT_MAX) : key_to_float(key);  // postprocess(max): -max for inner product (:465-505)
    if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) d = sqrtf(d);
    out_i[q * k + j] = s_idx[j];
    out_d[q * k + j] = d;
  }
}

template <typename T>
void refine_typed(resources& res, const void* data, int64_t n, int64_t dim, const void* queries, int64_t m,
                  const int64_t* cand, int n_cand, int k, int metric, int64_t* out_i, float* out_d)
{
  int np2     = next_pow2(n_cand);
  size_t smem = (size_t)np2 * 12;
  CUVS_EXPECTS(smem <= 64 * 1024, "refine: too many candidates per query (%d)", n_cand);
  CUVS_EXPECTS(m < (int64_t(1) << 24), "refine: too many queries in one call");
  hipLaunchKernelGGL((refine_kernel<T>), dim3((unsigned)m), dim3(256), smem, res.stream,
                     static_cast<const T*>(data), n, dim, static_cast<const T*>(queries), cand, n_cand, np2, k,
                     metric, out_i, out_d);
  HIP_TRY(hipGetLastError());
}

// ---- host tensors (reference: refine_host.hpp:353-462, dispatched from c/src/neighbors/refine.cpp when the tensors
// live in host memory): the same arithmetic as refine_kernel - 64 strided fma partial sums per row, summed as a
// butterfly - so that the host and device paths return identical results; queries are spread over host threads.
inline float host_value(float v) { return v; }
inline float host_value(__half v) { return __half2float(v); }
inline float host_value(int8_t v) { return (float)v; }
inline float host_value(uint8_t v) { return (float)v; }

template <typename T>
float host_strided(const T* a, const T* b, int64_t dim, int mode /*0 (a-b)^2, 1 a*b, 2 b*b*/)
{
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.f;
  for (int64_t j = 0; j < dim; ++j) {
    const float x = host_value(a[j]), y = host_value(b[j]);
    if (mode == 0) { const float t = x - y; p[j & 63] = fmaf(t, t, p[j & 63]); }
    else if (mode == 1) p[j & 63] = fmaf(x, y, p[j & 63]);
    else p[j & 63] = fmaf(y, y, p[j & 63]);
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
  return p[0];
}

template <typename T>
void refine_host_typed(const void* data_v, int64_t n, int64_t dim, const void* queries_v, int64_t m, const int64_t* cand,
                       int n_cand, int k, int metric, int64_t* out_i, float* out_d)
{
  const T* data    = static_cast<const T*>(data_v);
  const T* queries = static_cast<const T*>(queries_v);
  const bool ip = metric == M_InnerProduct, cosm = metric == M_CosineExpanded;
  const unsigned n_thr = (unsigned)std::max<int64_t>(1, std::min<int64_t>(m, std::max(1u, std::thread::hardware_concurrency())));
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < n_thr; ++t) {
    pool.emplace_back([=] {
      std::vector<std::pair<float, int64_t>> buf((size_t)n_cand);
      for (int64_t q = t; q < m; q += n_thr) {
        const T* qv = queries + q * dim;
        const float qn = cosm ? sqrtf(host_strided(qv, qv, dim, 2)) : 0.f;
        int cnt = 0;
        for (int c = 0; c < n_cand; ++c) {
          const int64_t id = cand[q * n_cand + c];
          if (id < 0 || id >= n) { buf[cnt++] = {FLT_MAX, id}; continue; }  // refine_host.hpp:440-442
          const T* row = data + id * dim;
          float v = host_strided(qv, row, dim, (ip || cosm) ? 1 : 0);
          if (cosm) v = 1.0f - v / (qn * sqrtf(host_strided(qv, row, dim, 2)));
          buf[cnt++] = {ip ? -v : v, id};  // sort key: smaller is better
        }
        std::sort(buf.begin(), buf.begin() + cnt);  // (distance, id) tuples, refine_host.hpp:430-460
        for (int j = 0; j < k; ++j) {
          if (j < cnt) {
            float d = ip ? -buf[j].first : buf[j].first;  // (an out-of-range candidate: postprocess(max) = -max for inner product)
            if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) d = sqrtf(d);
            out_i[q * k + j] = buf[j].second;
            out_d[q * k + j] = d;
          } else {
            out_i[q * k + j] = INT64_MAX;
            out_d[q * k + j] = FLT_MAX;
          }
        }
      }
    });
  }
  for (auto& th : pool) th.join();
}

}  // namespace

void refine_host(const void* data, elem_t et, int64_t n, int64_t dim, const void* queries, int64_t m, const int64_t* cand,
                 int n_cand, int k, int metric, int64_t* out_i, float* out_d)
{
  if (m == 0) return;
  CUVS_EXPECTS(k <= n_cand, "refine: k (%d) must not exceed the number of candidates (%d)", k, n_cand);
  CUVS_EXPECTS(metric_is_l2(metric) || metric == M_InnerProduct || metric == M_CosineExpanded,
               "refine: unsupported metric %d", metric);
  switch (et) {
    case elem_t::f32: refine_host_typed<float>(data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
    case elem_t::f16: refine_host_typed<__half>(data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
    case elem_t::i8: refine_host_typed<int8_t>(data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
    case elem_t::u8: refine_host_typed<uint8_t>(data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
  }
}

void refine(resources& res, const void* data, elem_t et, int64_t n, int64_t dim, const void* queries, int64_t m,
            const int64_t* cand, int n_cand, int k, int metric, int64_t* out_i, float* out_d)
{
  if (m == 0) return;
  CUVS_EXPECTS(k <= n_cand, "refine: k (%d) must not exceed the number of candidates (%d)", k, n_cand);
  CUVS_EXPECTS(metric_is_l2(metric) || metric == M_InnerProduct || metric == M_CosineExpanded,
               "refine: unsupported metric %d", metric);
  switch (et) {
    case elem_t::f32: refine_typed<float>(res, data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
    case elem_t::f16: refine_typed<__half>(res, data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
    case elem_t::i8: refine_typed<int8_t>(res, data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
    case elem_t::u8: refine_typed<uint8_t>(res, data, n, dim, queries, m, cand, n_cand, k, metric, out_i, out_d); break;
  }
}

}  // namespace cuvs_amd

using namespace cuvs_amd;

extern "C" cuvsError_t cuvsRefine(cuvsResources_t res_h, DLManagedTensor* dataset_tensor,
                                  DLManagedTensor* queries_tensor, DLManagedTensor* candidates_tensor,
                                  cuvsDistanceType metric, DLManagedTensor* indices_tensor,
                                  DLManagedTensor* distances_tensor)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(dataset_tensor && queries_tensor && candidates_tensor && indices_tensor && distances_tensor,
                 "null argument");
    auto& ds = dataset_tensor->dl_tensor;
    auto& qs = queries_tensor->dl_tensor;
    auto& cs = candidates_tensor->dl_tensor;
    auto& is = indices_tensor->dl_tensor;
    auto& dd = distances_tensor->dl_tensor;
    // c/src/neighbors/refine.cpp: all tensors in device memory -> the device path, all in host memory -> refine_host
    const bool all_dev  = is_device_accessible(ds) && is_device_accessible(qs) && is_device_accessible(cs) &&
                          is_device_accessible(is) && is_device_accessible(dd);
    const bool all_host = is_host_accessible(ds) && is_host_accessible(qs) && is_host_accessible(cs) &&
                          is_host_accessible(is) && is_host_accessible(dd);
    CUVS_EXPECTS(all_dev || all_host, "cuvsRefine: the tensors must be either all in device memory or all in host memory");
    CUVS_EXPECTS(ds.ndim == 2 && qs.ndim == 2 && cs.ndim == 2 && is.ndim == 2 && dd.ndim == 2, "tensors must be 2-D");
    CUVS_EXPECTS(is_c_contiguous(ds) && is_c_contiguous(qs) && is_c_contiguous(cs) && is_c_contiguous(is) &&
                   is_c_contiguous(dd),
                 "tensors must be C-contiguous");
    CUVS_EXPECTS(dtype_is(cs.dtype, kDLInt, 64) && dtype_is(is.dtype, kDLInt, 64), "candidates/indices must be int64");
    CUVS_EXPECTS(dtype_is(dd.dtype, kDLFloat, 32), "distances should be of type float32");
    CUVS_EXPECTS(ds.dtype.code == qs.dtype.code && ds.dtype.bits == qs.dtype.bits, "dataset/queries dtype mismatch");
    CUVS_EXPECTS(ds.shape[1] == qs.shape[1], "dataset/queries dim mismatch");
    CUVS_EXPECTS(cs.shape[0] == qs.shape[0] && is.shape[0] == qs.shape[0] && dd.shape[0] == qs.shape[0] &&
                   is.shape[1] == dd.shape[1],
                 "shape mismatch");
    if (!all_dev) {
      refine_host(dl_data(ds), elem_of(ds.dtype), ds.shape[0], ds.shape[1], dl_data(qs), qs.shape[0],
                  static_cast<const int64_t*>(dl_data(cs)), (int)cs.shape[1], (int)is.shape[1], (int)metric,
                  static_cast<int64_t*>(dl_data(is)), static_cast<float*>(dl_data(dd)));
      return;
    }
    refine(res, dl_data(ds), elem_of(ds.dtype), ds.shape[0], ds.shape[1], dl_data(qs), qs.shape[0],
           static_cast<const int64_t*>(dl_data(cs)), (int)cs.shape[1], (int)is.shape[1], (int)metric,
           static_cast<int64_t*>(dl_data(is)), static_cast<float*>(dl_data(dd)));
  });
}
