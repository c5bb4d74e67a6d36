// Brute-force kNN: index = (view of the dataset, canonical norms); search = MFMA distance tiles + exact
// radix select_k per tile + one merge select_k (reference: tiled_brute_force_knn,
// cpp/src/neighbors/detail/knn_brute_force.cuh:62-326, index build :779-817, C layer
// c/src/neighbors/brute_force.cpp:143-231). Pre-filters (bitset / bitmap) are applied to the distance tile.
#include "ops.hpp"
#include "device_utils.hpp"
#include "serialize.hpp"
#include "npy_io.hpp"

#include <cuvs/neighbors/brute_force.h>

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>

namespace cuvs_amd {

struct bf_index {
  int metric       = 0;
  float metric_arg = 2.0f;
  elem_t dtype     = elem_t::f32;
  int64_t n = 0, dim = 0, ld = 0;
  const void* data = nullptr;  // device pointer (view or owned)
  dev_buf<char> owned;         // set when the dataset had to be copied (F-contiguous / host input)
  dev_buf<float> norms;        // |x|^2 (L2) or |x| (cosine); empty for inner product
};

namespace {

// distances of filtered-out samples -> worst value. bits: 1 keeps the sample
// (reference: cuvs::core::bitset / bitmap_view semantics, sample_filter.cuh)
__global__ void apply_filter_kernel(float* d, int64_t m, int64_t n_tile, int64_t ldo, int64_t col0,
                                    int64_t row0, int64_t n_total, const uint32_t* bits, bool bitmap,
                                    float worst, const uint32_t* run_if = nullptr)
{
  if (run_if != nullptr && *run_if == 0u) return;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < m * n_tile;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx / n_tile, c = idx % n_tile;
    int64_t bit = bitmap ? (row0 + r) * n_total + (col0 + c) : (col0 + c);
    bool keep   = (bits[bit >> 5] >> (bit & 31)) & 1u;
    if (!keep) d[r * ldo + c] = worst;
  }
}

// ---- running-threshold path for datasets wider than one column tile (knn_brute_force.cuh:62-326 selects k per tile
// and merges; that re-reads every distance tile four times in select_k). After the first tile a row's current k-th
// value bounds its final k-th value, so a later tile only needs ONE pass: everything strictly better than the
// threshold is appended - in column order - behind the row's current top-k, and a small select_k over (k + appended)
// gives the new top-k. An element equal to the threshold cannot enter (ties go to the earlier column), so `<` is
// exact; appended elements all have larger column ids than the current winners, so position order == column order
// and the result is identical to the per-tile select + merge.
constexpr int kBfCap = 1024;  // appended candidates per row and tile; more (adversarial column order) -> select_k on the tile

// one wave per row: 64 columns per step, ordered compaction by ballot
__global__ __launch_bounds__(256) void bf_filter_append_kernel(const float* __restrict__ tile, int64_t m, int64_t nc,
                                                               int64_t ldo, int64_t col0, int k, bool select_min,
                                                               float* __restrict__ buf_v, int64_t* __restrict__ buf_i,
                                                               int* __restrict__ overflow)
{
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int lane     = threadIdx.x & 63;
  const int64_t ldb  = k + kBfCap;
  float* bv          = buf_v + row * ldb;
  int64_t* bi        = buf_i + row * ldb;
  const float thr    = bv[k - 1];  // current k-th value (worst value while fewer than k are known)
  const float* r     = tile + row * ldo;
  int cnt            = 0;
  for (int64_t c0 = 0; c0 < nc; c0 += 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t c = c0 + u * 64 + lane;
      v[u] = c < nc ? r[c] : (select_min ? FLT_MAX : -FLT_MAX);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool take = select_min ? v[u] < thr : v[u] > thr;
      const unsigned long long mk = __ballot(take);
      if (mk == 0ull) continue;
      const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      if (take && pos < kBfCap) { bv[k + pos] = v[u]; bi[k + pos] = col0 + c0 + u * 64 + lane; }
      cnt += (int)__popcll(mk);
    }
  }
  if (cnt > kBfCap) { if (lane == 0) *overflow = 1; cnt = kBfCap; }
  // unused slots lose: worst value, invalid id
  for (int j = cnt + lane; j < kBfCap; j += 64) { bv[k + j] = select_min ? FLT_MAX : -FLT_MAX; bi[k + j] = INT64_MAX; }
}

__global__ void bf_store_topk_kernel(const float* __restrict__ v, const int64_t* __restrict__ i, int64_t m, int k,
                                     float* __restrict__ buf_v, int64_t* __restrict__ buf_i)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * k) return;
  const int64_t row = t / k, j = t % k, ldb = k + kBfCap;
  buf_v[row * ldb + j] = v[t];
  buf_i[row * ldb + j] = i[t];
}

// ---- fused threshold path (round 2): beyond the first column tile the distance tile is never written. The MFMA kernel
// compares every element with the row's current k-th value in its epilogue and appends the few that beat it
// (pairwise_threshold_append); this kernel then sorts a row's k + appended entries by (value, source id) - the order
// of the tiled path: ties go to the smaller id - and keeps the first k. Column tiles grow geometrically: after `seen`
// columns a further `seen * cap / (4 k)` are expected to append cap / 4 per row.
__global__ __launch_bounds__(256) void bf_merge_appended_kernel(float* __restrict__ buf_v, int64_t* __restrict__ buf_i,
                                                                int* __restrict__ cnt, int k, bool select_min,
                                                                int* __restrict__ overflow)
{
  __shared__ float sv[2 * kBfCap];
  __shared__ int64_t si[2 * kBfCap];
  const int64_t row = blockIdx.x;
  int c             = cnt[row];
  if (c == 0) return;  // workgroup-uniform
  __syncthreads();
  if (threadIdx.x == 0) cnt[row] = 0;  // ready for the next tile's appends (no memset between the tiles)
  if (c > kBfCap) { if (threadIdx.x == 0) *overflow = 1; c = kBfCap; }
  const int total   = k + c;
  int P             = 1;
  while (P < total) P <<= 1;
  const int64_t ldb = k + kBfCap;
  const float worst = select_min ? FLT_MAX : -FLT_MAX;
  for (int t = threadIdx.x; t < P; t += 256) {
    sv[t] = t < total ? buf_v[row * ldb + t] : worst;
    si[t] = t < total ? buf_i[row * ldb + t] : INT64_MAX;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < P / 2; t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;  // ascending block: the better entry goes first
        const float va = sv[lo], vb = sv[hi];
        const int64_t ia = si[lo], ib = si[hi];
        const bool b_first = select_min ? (vb < va || (vb == va && ib < ia)) : (vb > va || (vb == va && ib < ia));
        const bool a_first = select_min ? (va < vb || (va == vb && ia < ib)) : (va > vb || (va == vb && ia < ib));
        if (up ? b_first : a_first) { sv[lo] = vb; sv[hi] = va; si[lo] = ib; si[hi] = ia; }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < k; t += 256) { buf_v[row * ldb + t] = sv[t]; buf_i[row * ldb + t] = si[t]; }
}

__global__ void bf_emit_topk_kernel(const float* __restrict__ buf_v, const int64_t* __restrict__ buf_i, int64_t m, int k,
                                    float* __restrict__ out_v, int64_t* __restrict__ out_i)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * k) return;
  const int64_t row = t / k, j = t % k, ldb = k + kBfCap;
  out_v[t] = buf_v[row * ldb + j];
  out_i[t] = buf_i[row * ldb + j];
}

template <typename T>
void bf_search_typed(resources& res, const bf_index& idx, const T* queries, int64_t m, int64_t ldq, int k,
                     int64_t* neighbors, float* distances, const uint32_t* filter_bits, int filter_type)
{
  const int metric      = idx.metric;
  const bool select_min = metric != M_InnerProduct;
  const int64_t n       = idx.n;
  const T* data         = static_cast<const T*>(idx.data);
  CUVS_EXPECTS(k >= 1, "brute_force::search: k must be positive");
  // (k > n is served, as in the reference - knn_brute_force.cuh has no such check: the per-tile select path pads the
  // missing slots with the worst value and an invalid id)

  // row tiles bounded by the workspace; column tiles as wide as the workspace allows. Large k (beyond the 2048
  // winners select_k keeps in LDS at full speed) shrinks the row tile so that the per-tile partial results stay small.
  const int64_t ws_floats = (int64_t)(res.workspace_limit / sizeof(float));
  int64_t m_tile          = std::min<int64_t>(m, k > 2048 ? 256 : 16384);
  int64_t n_tile          = std::max<int64_t>(128, (ws_floats / m_tile) / 128 * 128);
  n_tile                  = std::min<int64_t>(n_tile, round_up(n, 128));
  const int64_t n_ct      = (n + n_tile - 1) / n_tile;

  dev_buf<float> qn;
  if (metric != M_InnerProduct) {
    qn = dev_buf<float>(res, m);
    row_norms<T>(res, queries, m, idx.dim, ldq, qn.data(), metric == M_CosineExpanded);
  }
  // fused distance + top-k (the reference's fusedL2Knn path, knn_brute_force.cuh:452): k <= 64, no pre-filter.
  // Same results as the tiled path; in round 1 it is still slower than GEMM + select_k (244 VGPRs, cold-tile
  // cost), so it is opt-in: CUVS_AMD_BF_FUSED=1.
  if (k <= 64 && filter_type == NO_FILTER && res.tune.bf_fused) {
    bool done = true;
    for (int64_t r0 = 0; r0 < m && done; r0 += int64_t(65535) * 128) {
      const int64_t mr = std::min<int64_t>(int64_t(65535) * 128, m - r0);
      done = fused_knn<T, T>(res, queries + r0 * ldq, mr, ldq, data, n, idx.ld, idx.dim,
                             qn.data() ? qn.data() + r0 : nullptr, idx.norms.data(), metric, k, distances + r0 * k,
                             neighbors + r0 * k);
    }
    if (done) return;
  }
  dev_buf<float> tile(res, (size_t)m_tile * std::min<int64_t>(n_tile, n));
  const int64_t ldo = std::min<int64_t>(n_tile, n);
  // more than one column tile: the running-threshold path (above) unless switched off or k is too large for it
  // (a single wide tile - e.g. 1000 queries x 100k rows - is cut into four as well: three of its four quarters then
  // cost one pass instead of select_k's four, and the GEMM of a quarter overlaps the select of the previous one)
  const bool running = (n_ct > 1 || n >= 65536) && k <= 1024 && !res.tune.bf_no_threshold;
  dev_buf<float> part_v;
  dev_buf<int64_t> part_i;
  const float worst = select_min ? FLT_MAX : -FLT_MAX;

  // ---- running-threshold path, pipelined: the distance GEMM of tile ct + 1 (compute-bound, the handle's stream) runs
  // beside the filter + small select of tile ct (memory-bound, a helper stream); two half-width tiles alternate.
  std::vector<char> redo((size_t)((m + m_tile - 1) / m_tile), running ? 0 : 1);  // row tiles for the per-tile select path
  // first tile: n / 16 columns (4096 .. 32768, at least 32 k) - select_k reads it four times, and the thresholds it
  // gives let the next tile be 25 times wider at k = 10
  const int64_t n0_want = std::max<int64_t>(std::min<int64_t>(32768, std::max<int64_t>(4096, n / 16)), 32 * (int64_t)k);
  const int64_t n0      = std::min<int64_t>(std::min<int64_t>(n_tile, round_up(n, 128)), round_up(n0_want, 128));
  // (a workspace too small for a first tile of k columns: the tile path below)
  const bool fused_filter = running && std::min<int64_t>(n0, n) >= k && !res.tune.bf_no_fused_filter;
  // rows whose candidates were cut (below) are redone by the per-tile select path. With few column tiles that path is
  // ENQUEUED behind the fused one, every launch guarded on the device by the row tile's overflow flag (a handful of
  // no-op launches in the common case, no host round trip: the call stays asynchronous); with many, the flags are read
  // back once (a sync of tens of microseconds on a search of tens of milliseconds)
  const bool guarded = fused_filter && n_ct <= 4 && !res.tune.bf_host_flags;
  // overflow flags of the row tiles, then the rows' append counters (zeroed once: the merge kernel resets what it used)
  dev_buf<int> ovf(res, fused_filter ? redo.size() + (size_t)m_tile : 0);
  if (fused_filter) {
    const int64_t ld0 = std::min<int64_t>(n0, n);
    dev_buf<float> buf_v(res, (size_t)m_tile * (k + kBfCap));
    dev_buf<int64_t> buf_i(res, (size_t)m_tile * (k + kBfCap));
    const size_t n_row_tiles = redo.size();
    int* cnt = ovf.data() + n_row_tiles;
    HIP_TRY(hipMemsetAsync(ovf.data(), 0, ovf.bytes(), res.stream));
    for (int64_t r0 = 0; r0 < m; r0 += m_tile) {
      const int64_t mr = std::min(m_tile, m - r0);
      const T* qr      = queries + r0 * ldq;
      const float* qnr = qn.data() ? qn.data() + r0 : nullptr;
      // first tile: distances + select_k straight into the head of every row's buffer
      const int64_t nc0 = std::min(n0, n);
      pairwise_distance<T, T>(res, qr, mr, ldq, data, nc0, idx.ld, idx.dim, qnr, idx.norms.data(), metric, tile.data(), ld0);
      if (filter_type != NO_FILTER) {
        const int64_t total = mr * nc0;
        hipLaunchKernelGGL(apply_filter_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 22)), dim3(256), 0,
                           res.stream, tile.data(), mr, nc0, ld0, (int64_t)0, r0, n, filter_bits, filter_type == BITMAP, worst);
      }
      select_k<int64_t, int64_t>(res, tile.data(), nullptr, mr, nc0, ld0, k, buf_v.data(), buf_i.data(), select_min, 0,
                                 k + kBfCap, 0);
      for (int64_t seen = nc0; seen < n;) {
        int64_t nc = (int64_t)((double)seen * kBfCap / (4.0 * k));
        nc         = std::min<int64_t>(n - seen, std::max<int64_t>(16384, nc / 128 * 128));
        pairwise_threshold_append<T, T>(res, qr, mr, ldq, data + seen * idx.ld, nc, idx.ld, idx.dim, qnr,
                                        idx.norms.data() ? idx.norms.data() + seen : nullptr, metric, buf_v.data(),
                                        buf_i.data(), cnt, k, kBfCap, seen, r0, n, filter_bits,
                                        filter_type == NO_FILTER ? 0 : (filter_type == BITMAP ? 2 : 1));
        hipLaunchKernelGGL(bf_merge_appended_kernel, dim3((unsigned)mr), dim3(256), 0, res.stream, buf_v.data(), buf_i.data(),
                           cnt, k, select_min, ovf.data() + r0 / m_tile);
        seen += nc;
      }
      hipLaunchKernelGGL(bf_emit_topk_kernel, dim3(grid_blocks(mr * k, 256)), dim3(256), 0, res.stream, buf_v.data(),
                         buf_i.data(), mr, k, distances + r0 * k, neighbors + r0 * k);
    }
    // a row with more than kBfCap better elements in one tile (columns arriving in improving order, or a pre-filter
    // that leaves fewer than k of the first tile) had its candidates cut: such row tiles are redone by the per-tile
    // select path below. One host round trip per search, only to read the flags.
    if (guarded) {
      for (size_t t = 0; t < n_row_tiles; ++t) redo[t] = 1;
    } else {
      std::vector<int> h_ovf(n_row_tiles, 0);
      HIP_TRY(hipMemcpyAsync(h_ovf.data(), ovf.data(), n_row_tiles * sizeof(int), hipMemcpyDeviceToHost, res.stream));
      HIP_TRY(hipStreamSynchronize(res.stream));
      for (size_t t = 0; t < n_row_tiles; ++t) redo[t] = h_ovf[t] != 0;
    }
  } else if (running) {
    const int64_t nt2   = std::max<int64_t>(128, std::min<int64_t>((n_tile / 2) / 128 * 128, round_up((n + 3) / 4, 128)));
    const int64_t n_ct2 = (n + nt2 - 1) / nt2;
    const int64_t ld2   = std::min<int64_t>(nt2, n);
    float* tiles[2]     = {tile.data(), tile.data() + (size_t)m_tile * ld2};
    dev_buf<float> buf_v(res, (size_t)m_tile * (k + kBfCap)), cur_v(res, (size_t)m_tile * k);
    dev_buf<int64_t> buf_i(res, (size_t)m_tile * (k + kBfCap)), cur_i(res, (size_t)m_tile * k);
    const size_t n_row_tiles = redo.size();
    dev_buf<int> ovf2(res, n_row_tiles);  // one flag per row tile, read back once at the end (NOT the guard word of the fused path)
    HIP_TRY(hipMemsetAsync(ovf2.data(), 0, ovf2.bytes(), res.stream));
    ensure_aux_stream(res);  // the helper stream and its events live with the handle
    hipStream_t sb = res.aux_stream;
    hipEvent_t* eg = res.aux_events;
    hipEvent_t* ed = res.aux_events + 2;
    hipEvent_t e0  = res.aux_events[4];
    resources rb = res;
    rb.stream    = sb;
    HIP_TRY(hipEventRecord(e0, res.stream));  // the helper stream starts behind everything queued so far (norms, allocations)
    HIP_TRY(hipStreamWaitEvent(sb, e0, 0));
    for (int64_t r0 = 0; r0 < m; r0 += m_tile) {
      const int64_t mr = std::min(m_tile, m - r0);
      int* ovf_r       = ovf2.data() + r0 / m_tile;
      for (int64_t ct = 0; ct < n_ct2; ++ct) {
        const int b      = (int)(ct & 1);
        const int64_t c0 = ct * nt2, nc = std::min(nt2, n - c0);
        if (ct >= 2) HIP_TRY(hipStreamWaitEvent(res.stream, ed[b], 0));  // the consumer is done with this tile buffer
        pairwise_distance<T, T>(res, queries + r0 * ldq, mr, ldq, data + c0 * idx.ld, nc, idx.ld, idx.dim,
                                qn.data() ? qn.data() + r0 : nullptr, idx.norms.data() ? idx.norms.data() + c0 : nullptr,
                                metric, tiles[b], ld2);
        if (filter_type != NO_FILTER) {
          const int64_t total = mr * nc;
          hipLaunchKernelGGL(apply_filter_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 22)), dim3(256), 0,
                             res.stream, tiles[b], mr, nc, ld2, c0, r0, n, filter_bits, filter_type == BITMAP, worst);
        }
        HIP_TRY(hipEventRecord(eg[b], res.stream));
        HIP_TRY(hipStreamWaitEvent(sb, eg[b], 0));
        const bool last = ct == n_ct2 - 1;
        float* ov       = last ? distances + r0 * k : cur_v.data();
        int64_t* oi     = last ? neighbors + r0 * k : cur_i.data();
        if (ct == 0) {  // the first tile's top-k IS the current top-k: straight into the head of every row's buffer
          select_k<int64_t, int64_t>(rb, tiles[b], nullptr, mr, nc, ld2, k, buf_v.data(), buf_i.data(), select_min, c0,
                                     k + kBfCap, 0);
        } else {
          hipLaunchKernelGGL(bf_filter_append_kernel, dim3((unsigned)((mr + 3) / 4)), dim3(256), 0, sb, tiles[b], mr, nc, ld2,
                             c0, k, select_min, buf_v.data(), buf_i.data(), ovf_r);
          select_k<int64_t, int64_t>(rb, buf_v.data(), buf_i.data(), mr, k + kBfCap, k + kBfCap, k, ov, oi, select_min);
          if (!last)  // the new top-k becomes the head of every row's buffer for the next tile
            hipLaunchKernelGGL(bf_store_topk_kernel, dim3(grid_blocks(mr * k, 256)), dim3(256), 0, sb, cur_v.data(),
                               cur_i.data(), mr, k, buf_v.data(), buf_i.data());
        }
        HIP_TRY(hipEventRecord(ed[b], sb));
      }
      HIP_TRY(hipStreamWaitEvent(res.stream, ed[(n_ct2 - 1) & 1], 0));  // results are ordered on the handle's stream
      if (r0 + m_tile < m) {  // the next row tile reuses the buffers of the helper stream: let it drain first
        HIP_TRY(hipEventRecord(e0, sb));
        HIP_TRY(hipStreamWaitEvent(res.stream, e0, 0));
      }
    }
    // a row with more than kBfCap better elements in one tile (columns arriving in improving order) had its candidates
    // cut: such row tiles are redone by the per-tile select path below. One host round trip per search, only to read
    // the flags - the common case leaves the results where they are.
    // (`guarded` implies the fused path: this one always reads its flags back)
    std::vector<int> h_ovf(n_row_tiles, 0);
    HIP_TRY(hipMemcpyAsync(h_ovf.data(), ovf2.data(), n_row_tiles * sizeof(int), hipMemcpyDeviceToHost, res.stream));
    HIP_TRY(hipStreamSynchronize(res.stream));
    for (size_t t = 0; t < n_row_tiles; ++t) redo[t] = h_ovf[t] != 0;
  }
  bool any_redo = false;
  for (char c : redo) any_redo = any_redo || c;
  if (any_redo && n_ct > 1) {
    part_v = dev_buf<float>(res, (size_t)m_tile * n_ct * k);
    part_i = dev_buf<int64_t>(res, (size_t)m_tile * n_ct * k);
  }

  for (int64_t r0 = 0; r0 < m; r0 += m_tile) {
    if (!redo[(size_t)(r0 / m_tile)]) continue;
    const int64_t mr = std::min(m_tile, m - r0);
    const uint32_t* run_if = guarded ? reinterpret_cast<const uint32_t*>(ovf.data() + r0 / m_tile) : nullptr;
    for (int64_t ct = 0; ct < n_ct; ++ct) {
      const int64_t c0 = ct * n_tile;
      const int64_t nc = std::min(n_tile, n - c0);
      pairwise_distance<T, T>(res, queries + r0 * ldq, mr, ldq, data + c0 * idx.ld, nc, idx.ld, idx.dim,
                              qn.data() ? qn.data() + r0 : nullptr,
                              idx.norms.data() ? idx.norms.data() + c0 : nullptr, metric, tile.data(), ldo, run_if);
      if (filter_type != NO_FILTER) {
        int64_t total = mr * nc;
        hipLaunchKernelGGL(apply_filter_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 22)), dim3(256), 0, res.stream,
                           tile.data(), mr, nc, ldo, c0, r0, n, filter_bits, filter_type == BITMAP, worst, run_if);
      }
      if (n_ct == 1) {
        select_k<int64_t, int64_t>(res, tile.data(), nullptr, mr, nc, ldo, k, distances + r0 * k,
                                   neighbors + r0 * k, select_min, c0, -1, 0, run_if);
      } else {
        select_k<int64_t, int64_t>(res, tile.data(), nullptr, mr, nc, ldo, k, part_v.data(), part_i.data(),
                                   select_min, c0, n_ct * k, ct * k, run_if);
      }
    }
    if (n_ct > 1) {
      select_k<int64_t, int64_t>(res, part_v.data(), part_i.data(), mr, n_ct * k, n_ct * k, k,
                                 distances + r0 * k, neighbors + r0 * k, select_min, 0, -1, 0, run_if);
    }
  }
  HIP_TRY(hipGetLastError());
}

constexpr int kBfRefVersion = 0;  // brute_force_serialize.cu:19

template <typename T>
void bf_build_typed(resources& res, bf_index& idx)
{
  if (idx.metric != M_InnerProduct) {
    idx.norms = dev_buf<float>::persistent(idx.n);
    row_norms<T>(res, static_cast<const T*>(idx.data), idx.n, idx.dim, idx.ld, idx.norms.data(),
                 idx.metric == M_CosineExpanded);
  }
}

__global__ void transpose_copy_kernel(const char* src, char* dst, int64_t rows, int64_t cols, int esz)
{
  // src is column-major [rows, cols] (element (r,c) at c*rows + r); dst row-major
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / cols, c = i % cols;
    for (int b = 0; b < esz; ++b) dst[i * esz + b] = src[(c * rows + r) * esz + b];
  }
}

}  // namespace
}  // namespace cuvs_amd

using namespace cuvs_amd;

extern "C" {

cuvsError_t cuvsBruteForceIndexCreate(cuvsBruteForceIndex_t* index)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(index != nullptr, "index is null");
    *index = new cuvsBruteForceIndex{0, DLDataType{0, 0, 0}};
  });
}

cuvsError_t cuvsBruteForceIndexDestroy(cuvsBruteForceIndex_t index_c_ptr)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (index_c_ptr == nullptr) return;
    delete reinterpret_cast<bf_index*>(index_c_ptr->addr);
    delete index_c_ptr;
  });
}

cuvsError_t cuvsBruteForceBuild(cuvsResources_t res_h, DLManagedTensor* dataset_tensor, cuvsDistanceType metric,
                                float metric_arg, cuvsBruteForceIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(dataset_tensor != nullptr && index != nullptr, "null argument");
    auto& ds = dataset_tensor->dl_tensor;
    CUVS_EXPECTS(ds.ndim == 2, "dataset must be a matrix");
    CUVS_EXPECTS(metric_supported((int)metric), "brute_force: unsupported metric %d", (int)metric);
    elem_t et = elem_of(ds.dtype);
    CUVS_EXPECTS(et == elem_t::f32 || et == elem_t::f16, "Unsupported dataset DLtensor dtype: %d and bits: %d",
                 (int)ds.dtype.code, (int)ds.dtype.bits);
    auto idx        = std::make_unique<bf_index>();
    idx->metric     = (int)metric;
    idx->metric_arg = metric_arg;
    idx->dtype      = et;
    idx->n          = ds.shape[0];
    idx->dim        = ds.shape[1];
    idx->ld         = idx->dim;
    const size_t esz = elem_size(et);
    if (is_device_accessible(ds) && is_c_contiguous(ds)) {
      idx->data = dl_data(ds);  // non-owning view, as in the reference (brute_force.cu:66-79)
    } else if (is_device_accessible(ds) && is_f_contiguous(ds)) {
      idx->owned = dev_buf<char>::persistent((size_t)idx->n * idx->dim * esz);
      int64_t total = idx->n * idx->dim;
      hipLaunchKernelGGL(transpose_copy_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 22)), dim3(256), 0, res.stream,
                         static_cast<const char*>(dl_data(ds)), idx->owned.data(), idx->n, idx->dim, (int)esz);
      idx->data = idx->owned.data();
    } else if (is_c_contiguous(ds)) {
      idx->owned = dev_buf<char>::persistent((size_t)idx->n * idx->dim * esz);
      copy_async(res, idx->owned.data(), dl_data(ds), idx->owned.bytes());
      idx->data = idx->owned.data();
    } else {
      CUVS_FAIL("dataset input to cuvsBruteForceBuild must be contiguous (non-strided)");
    }
    if (et == elem_t::f32) bf_build_typed<float>(res, *idx); else bf_build_typed<__half>(res, *idx);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = ds.dtype;
  });
}

cuvsError_t cuvsBruteForceSearch(cuvsResources_t res_h, cuvsBruteForceIndex_t index_c_ptr,
                                 DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                                 DLManagedTensor* distances_tensor, cuvsFilter prefilter)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index_c_ptr && index_c_ptr->addr, "index is not built");
    auto& idx       = *reinterpret_cast<bf_index*>(index_c_ptr->addr);
    auto& queries   = queries_tensor->dl_tensor;
    auto& neighbors = neighbors_tensor->dl_tensor;
    auto& distances = distances_tensor->dl_tensor;
    CUVS_EXPECTS(is_device_accessible(queries), "queries should have device compatible memory");
    CUVS_EXPECTS(is_device_accessible(neighbors), "neighbors should have device compatible memory");
    CUVS_EXPECTS(is_device_accessible(distances), "distances should have device compatible memory");
    CUVS_EXPECTS(dtype_is(neighbors.dtype, kDLInt, 64), "neighbors should be of type int64_t");
    CUVS_EXPECTS(dtype_is(distances.dtype, kDLFloat, 32), "distances should be of type float32");
    CUVS_EXPECTS(queries.ndim == 2 && neighbors.ndim == 2 && distances.ndim == 2, "tensors must be 2-D");
    CUVS_EXPECTS(is_c_contiguous(queries) && is_c_contiguous(neighbors) && is_c_contiguous(distances),
                 "tensors must be C-contiguous");
    CUVS_EXPECTS(queries.dtype.code == index_c_ptr->dtype.code && queries.dtype.bits == index_c_ptr->dtype.bits,
                 "Unsupported queries DLtensor dtype: %d and bits: %d", (int)queries.dtype.code,
                 (int)queries.dtype.bits);
    CUVS_EXPECTS(queries.shape[1] == idx.dim, "queries dim %ld != index dim %ld", (long)queries.shape[1],
                 (long)idx.dim);
    int64_t m = queries.shape[0];
    int64_t k = neighbors.shape[1];
    CUVS_EXPECTS(neighbors.shape[0] == m && distances.shape[0] == m && distances.shape[1] == k,
                 "neighbors/distances shape mismatch");
    const uint32_t* bits = nullptr;
    if (prefilter.type != NO_FILTER) {
      CUVS_EXPECTS(prefilter.type == BITSET || prefilter.type == BITMAP, "unsupported prefilter type");
      CUVS_EXPECTS(prefilter.addr != 0, "prefilter tensor is null");
      auto& ft = reinterpret_cast<DLManagedTensor*>(prefilter.addr)->dl_tensor;
      CUVS_EXPECTS(dtype_is(ft.dtype, kDLUInt, 32) && is_device_accessible(ft),
                   "prefilter must be a device uint32 tensor");
      bits = static_cast<const uint32_t*>(dl_data(ft));
    }
    if (idx.dtype == elem_t::f32) {
      bf_search_typed<float>(res, idx, static_cast<const float*>(dl_data(queries)), m, idx.dim, (int)k,
                             static_cast<int64_t*>(dl_data(neighbors)), static_cast<float*>(dl_data(distances)),
                             bits, (int)prefilter.type);
    } else {
      bf_search_typed<__half>(res, idx, static_cast<const __half*>(dl_data(queries)), m, idx.dim, (int)k,
                              static_cast<int64_t*>(dl_data(neighbors)), static_cast<float*>(dl_data(distances)),
                              bits, (int)prefilter.type);
    }
  });
}

cuvsError_t cuvsBruteForceSerialize(cuvsResources_t res_h, const char* filename, cuvsBruteForceIndex_t index_c_ptr)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index_c_ptr && index_c_ptr->addr, "index is not built");
    auto& idx = *reinterpret_cast<bf_index*>(index_c_ptr->addr);
    if (write_native_container(res)) {
      file_writer w(filename, KIND_BRUTE_FORCE);
      w.scalar<int32_t>(idx.metric); w.scalar<float>(idx.metric_arg); w.scalar<int32_t>((int)idx.dtype);
      w.scalar<int64_t>(idx.n); w.scalar<int64_t>(idx.dim);
      w.scalar<uint8_t>(index_c_ptr->dtype.code); w.scalar<uint8_t>(index_c_ptr->dtype.bits);
      CUVS_EXPECTS(idx.ld == idx.dim, "strided dataset cannot be serialized");
      w.device_array(res, idx.data, (size_t)idx.n * idx.dim * elem_size(idx.dtype));
      w.device_array(res, idx.norms.data(), idx.norms.bytes());
      return;
    }
    // the reference's record sequence (brute_force_serialize.cu:30-45): dtype prefix, version, rows, dim,
    // metric, metric_arg, include_dataset, dataset [rows, dim], has_norms, norms [rows]
    npy_writer w(filename);
    char prefix[4];
    elem_prefix(idx.dtype, prefix);
    w.raw(prefix, 4);
    w.scalar<int32_t>(kBfRefVersion);
    w.scalar<uint64_t>((uint64_t)idx.n);
    w.scalar<uint64_t>((uint64_t)idx.dim);
    w.scalar<int32_t>(idx.metric);
    w.scalar<float>(idx.metric_arg);
    w.scalar<bool>(true);
    const size_t es = elem_size(idx.dtype);
    w.device_array(res, idx.dtype == elem_t::f32 ? 'f' : 'e', (uint32_t)es, {idx.n, idx.dim}, idx.data,
                   (size_t)idx.dim * es, (size_t)idx.ld * es);
    const bool has_norms = idx.norms.size() > 0;
    w.scalar<bool>(has_norms);
    if (has_norms) w.device_array(res, 'f', 4, {idx.n}, idx.norms.data());
    w.close();
  });
}

cuvsError_t cuvsBruteForceDeserialize(cuvsResources_t res_h, const char* filename, cuvsBruteForceIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr, "index is null");
    auto idx = std::make_unique<bf_index>();
    DLDataType dl;
    if (is_native_container(filename)) {
      file_reader r(filename, KIND_BRUTE_FORCE);
      idx->metric     = r.scalar<int32_t>();
      idx->metric_arg = r.scalar<float>();
      idx->dtype      = (elem_t)r.scalar<int32_t>();
      idx->n          = r.scalar<int64_t>();
      idx->dim        = r.scalar<int64_t>();
      idx->ld         = idx->dim;
      uint8_t code = r.scalar<uint8_t>(), bits = r.scalar<uint8_t>();
      idx->owned = r.device_array<char>(res);
      idx->norms = r.device_array<float>(res);
      idx->data  = idx->owned.data();
      dl         = DLDataType{code, bits, 1};
    } else {
      // reference format (brute_force_serialize.cu:82-140; dtype dispatch c/src/neighbors/brute_force.cpp:239-262)
      npy_reader r(filename);
      char prefix[4];
      r.raw(prefix, 4);
      CUVS_EXPECTS(parse_elem_prefix(prefix, &idx->dtype) && (idx->dtype == elem_t::f32 || idx->dtype == elem_t::f16),
                   "Unsupported index dtype in file %s", filename);
      int ver = r.scalar<int32_t>();
      CUVS_EXPECTS(ver == kBfRefVersion, "serialization version mismatch, expected %d, got %d ", kBfRefVersion, ver);
      idx->n   = (int64_t)r.scalar<uint64_t>();
      idx->dim = (int64_t)r.scalar<uint64_t>();
      idx->ld  = idx->dim;
      CUVS_EXPECTS(idx->n >= 0 && idx->dim > 0, "brute_force::deserialize: bad shape");
      idx->metric     = r.scalar<int32_t>();
      idx->metric_arg = r.scalar<float>();
      CUVS_EXPECTS(metric_supported(idx->metric), "brute_force::deserialize: invalid metric value %d", idx->metric);
      bool include_dataset = r.scalar<bool>();
      CUVS_EXPECTS(include_dataset, "%s holds no dataset: a brute-force index cannot be searched without one", filename);
      idx->owned = r.device_bytes(res, (uint32_t)elem_size(idx->dtype), idx->n * idx->dim);
      idx->data = idx->owned.data();
      // norms are recomputed canonically (distance.hip row_norms) instead of trusting the file's rounding, so a
      // loaded index answers bit-identically to one built here from the same rows
      bool has_norms = r.scalar<bool>();
      (void)has_norms;
      if (idx->dtype == elem_t::f32) bf_build_typed<float>(res, *idx); else bf_build_typed<__half>(res, *idx);
      sync(res);
      dl = dl_of(idx->dtype);
    }
    delete reinterpret_cast<bf_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = dl;
  });
}

}  // extern "C"
