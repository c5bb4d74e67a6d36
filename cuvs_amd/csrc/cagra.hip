// CAGRA on MI355X: graph build (kNN graph + rank-based prune + reverse-edge merge), graph search, C ABI
// (drop-in for c/src/neighbors/cagra.cpp).
//
// Reference: cpp/src/neighbors/detail/cagra/graph_core.cuh (kern_fused_prune :206-330 — 2-hop detour
// counting, kern_make_rev_graph_k :178-200, kern_merge_graph :375-470, optimize :1706-1809),
// cagra_build.cuh:1629-1760 (kNN graph by IVF-PQ + refine), search_plan.cuh:199-245 (max_iterations rule),
// search_single_cta_jit.cuh:105-452 (search loop), device_common_jit.cuh:36-181 (seeding, child distances),
// hashmap.hpp:23-148 (open addressing), compute_distance_impl.cuh:23-64 (team distance).
//
// MI355X design of the search: ONE wave64 per query (the reference's "single CTA" sized for 32-lane warps
// does not transfer): the wave keeps the sorted internal top list and the candidate list in LDS, dedups
// children with a small LDS hash (reset every few iterations like the reference's SMALL hash mode), computes
// child distances with 8 teams of 8 lanes (each team streams one dataset row in 128-byte pieces, 16 B per
// lane, so 8 rows are in flight per wave) and merges with a wave-level bitonic sort. ~12 KB of LDS per query
// lets a CU hold 13+ concurrent queries, which is what hides the dependent-gather latency of the walk.
#include "ivf_pq.hpp"
#include "ivf_pq_filter_common.hpp"
#include "ops.hpp"
#include "device_utils.hpp"
#include "serialize.hpp"
#include "npy_io.hpp"

#include <cuvs/neighbors/cagra.h>

#include <algorithm>
#include <type_traits>
#include <cfloat>
#include <cmath>

namespace cuvs_amd {

struct cagra_index {
  int metric   = 0;
  elem_t dtype = elem_t::f32;
  int64_t n = 0, dim = 0;
  uint32_t degree = 0;
  const void* data = nullptr;  // device rows [n, dim]
  dev_buf<char> owned;
  dev_buf<uint32_t> graph;     // [n, degree]
  dev_buf<float> norms;        // [n] canonical |x| (cosine only; the reference's dataset_norms)
  // optional: the source id of every row (cagra.hpp index::source_indices, written by the reference's serializer as content-map
  // bit 1, cagra_serialize.cuh:72-83): a search reports source_indices[row] instead of the row (search_multi_cta.cuh:266-272)
  dev_buf<uint32_t> source_indices;
};

// canonical row norms of the dataset for the cosine metric (no-op otherwise)
static void cagra_set_norms(resources& res, cagra_index& idx)
{
  if (idx.metric != M_CosineExpanded || idx.data == nullptr) return;
  idx.norms = dev_buf<float>::persistent(idx.n);
  switch (idx.dtype) {
    case elem_t::f32: row_norms<float>(res, static_cast<const float*>(idx.data), idx.n, idx.dim, idx.dim, idx.norms.data(), true); break;
    case elem_t::f16: row_norms<__half>(res, static_cast<const __half*>(idx.data), idx.n, idx.dim, idx.dim, idx.norms.data(), true); break;
    case elem_t::i8: row_norms<int8_t>(res, static_cast<const int8_t*>(idx.data), idx.n, idx.dim, idx.dim, idx.norms.data(), true); break;
    case elem_t::u8: row_norms<uint8_t>(res, static_cast<const uint8_t*>(idx.data), idx.n, idx.dim, idx.dim, idx.norms.data(), true); break;
  }
}

namespace {

constexpr uint32_t kInvalidNode = 0xffffffffu;
constexpr uint32_t kParentFlag  = 0x80000000u;

// ------------------------------------------------------------------ graph optimisation
// kern_fused_prune restated: one wave per node. For every neighbour D (rank kAD) of the node and every neighbour B of D,
// the FIRST rank kAB > kAD at which the node lists B gets a detour count. The node's list is looked up through a small
// open-addressing table (id -> lowest rank) instead of a linear scan of up to K LDS words per candidate; a hit at a
// rank <= kAD (duplicate ids in the list) falls back to the scan, so the counts are exactly those of the scan. The row
// of the next neighbour is loaded while the current one is processed (the 127 row gathers of a node were serial).
template <int PER>  // ids per lane and neighbour row: K <= 64 * PER; 16 / PER rows are in flight
__global__ __launch_bounds__(256) void prune_kernel(const uint32_t* __restrict__ knn, int64_t n, uint32_t K,
                                                    uint32_t out_degree, uint32_t* __restrict__ out, int64_t nid0, int dbg)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t tsize = 64;
  while (tsize < 2 * K) tsize <<= 1;  // <= 50 % load
  uint32_t* s_idx = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * (2 * K + 2 * tsize);
  uint32_t* s_det = s_idx + K;
  uint32_t* t_key = s_det + K;
  uint32_t* t_pos = t_key + tsize;
  const int64_t nid = nid0 + (int64_t)blockIdx.x * 4 + wave;
  if (nid >= n) return;
  for (uint32_t k = lane; k < K; k += 64) {
    uint32_t v = knn[nid * K + k];
    s_idx[k]   = v;
    s_det[k]   = (v == (uint32_t)nid) ? K : 0;
  }
  for (uint32_t t = lane; t < tsize; t += 64) { t_key[t] = kInvalidNode; t_pos[t] = 0xffffffffu; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // table: id -> lowest rank holding it (invalid ids are not entered). A list with duplicate or invalid ids (not what a
  // kNN graph holds, but legal input) keeps the scan as the fallback; a wave whose list has neither never scans - one
  // scanning lane would hold up the other 63
  bool odd = false;
  for (uint32_t k = lane; k < K; k += 64) {
    const uint32_t v = s_idx[k];
    if (v >= n) { odd = true; continue; }
    uint32_t pos = (v * 2654435761u) & (tsize - 1);
    for (;;) {
      const uint32_t old = atomicCAS(&t_key[pos], kInvalidNode, v);
      if (old == v) odd = true;  // a second rank with this id
      if (old == kInvalidNode || old == v) { atomicMin(&t_pos[pos], k); break; }
      pos = (pos + 1) & (tsize - 1);
    }
  }
  const bool has_odd = __ballot(odd) != 0ull;  // wave-uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int D = 16 / PER;  // neighbour rows loaded ahead as a block
  uint32_t nxt[D][PER];
  auto load_block = [&](const uint32_t kAD0, uint32_t (&blk)[D][PER]) {
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const uint32_t kAD = kAD0 + dd;
      const uint32_t iD  = kAD + 1 < K ? s_idx[kAD] : kInvalidNode;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t kDB = lane + 64 * u;
        blk[dd][u] = (kDB < K && iD < n) ? knn[(int64_t)iD * K + kDB] : kInvalidNode;
      }
    }
  };
  load_block(0, nxt);
  for (uint32_t kAD0 = 0; kAD0 + 1 < ((dbg & 1) ? 0u : K); kAD0 += D) {
    uint32_t cur[D][PER];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
      for (int u = 0; u < PER; ++u) cur[dd][u] = nxt[dd][u];
    load_block(kAD0 + D, nxt);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const uint32_t kAD = kAD0 + dd;
      if (kAD + 1 >= K) break;  // wave-uniform
      const uint32_t iD = s_idx[kAD];
      if (iD >= n) continue;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t cand = cur[dd][u];
        if (lane + 64 * u >= K) continue;
        if (cand >= n) {  // invalid candidate: the scan matches the first invalid entry of the node's list, if any
          if (has_odd)
            for (uint32_t kAB = kAD + 1; kAB < K; ++kAB)
              if (s_idx[kAB] == cand) { atomicAdd(&s_det[kAB], 1u); break; }
          continue;
        }
        uint32_t pos = (cand * 2654435761u) & (tsize - 1), hit = 0xffffffffu;
        for (;;) {
          const uint32_t key = t_key[pos];
          if (key == cand) { hit = t_pos[pos]; break; }
          if (key == kInvalidNode) break;
          pos = (pos + 1) & (tsize - 1);
        }
        if (hit == 0xffffffffu) continue;  // the node does not list cand
        if (hit > kAD) {
          atomicAdd(&s_det[hit], 1u);
        } else if (has_odd) {  // lowest rank is not behind kAD: a duplicate further down would still count
          for (uint32_t kAB = kAD + 1; kAB < K; ++kAB)
            if (s_idx[kAB] == cand) { atomicAdd(&s_det[kAB], 1u); break; }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (uint32_t k = lane; k < K; k += 64) {
    uint32_t d = min(s_det[k], 0xffffu);
    if (s_idx[k] >= n) d = 0xffffu;
    s_det[k] = d;
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = 0; i < ((dbg & 2) ? 0u : out_degree); ++i) {
    uint32_t best = 0xffffffffu;
    for (uint32_t k = lane; k < K; k += 64) {
      uint32_t tag = (s_det[k] << 16) | k;
      if (s_det[k] < 0xffffu && tag < best) best = tag;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, kWave));
    uint32_t sel = kInvalidNode;
    if (best != 0xffffffffu) {
      sel = s_idx[best & 0xffffu];
      for (uint32_t k = lane; k < K; k += 64)
        if (s_idx[k] == sel) s_det[k] = 0xffffu;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) out[nid * out_degree + i] = sel;
  }
}

// edge list of ranks [r0, r0 + nr) in (rank-major, source-ascending) order: dest[e], e = (k - r0) * n + src
__global__ void edge_dest_kernel(const uint32_t* __restrict__ g, int64_t n, uint32_t degree, uint32_t r0, uint32_t nr,
                                 uint32_t* dest)
{
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * (int64_t)nr) return;
  int64_t k = r0 + e / n, src = e % n;
  uint32_t d = g[src * degree + k];
  dest[e]    = d < n ? d : (uint32_t)(n);  // invalid edges go to the extra bucket n
}

// the first `degree` reverse edges of every node in (rank, source) order, collected rank chunk by rank chunk (a chunk
// holds fewer than 2^32 edges, so n * degree itself is not bounded)
__global__ void append_reverse_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ off, int64_t n,
                                      uint32_t degree, uint32_t* __restrict__ rev, uint32_t* __restrict__ rev_cnt)
{
  const int64_t nid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (nid >= n) return;
  const uint32_t b = off[nid], e = off[nid + 1];
  const uint32_t c = rev_cnt[nid];
  const uint32_t take = min(e - b, degree - c);
  for (uint32_t t = 0; t < take; ++t) rev[(size_t)nid * degree + c + t] = perm[b + t] % (uint32_t)n;  // source node
  rev_cnt[nid] = c + take;
}

// kern_merge_graph restated: the protected head of a row is its spanning-forest edges (guarantee_connectivity; mst ==
// nullptr otherwise) followed by the pruned edges not among them, at least degree/2 entries; reverse edges are inserted
// behind the protected head.
__global__ __launch_bounds__(256) void merge_graph_kernel(uint32_t* __restrict__ g, int64_t n, uint32_t degree,
                                                          const uint32_t* __restrict__ rev,
                                                          const uint32_t* __restrict__ rev_cnt, int64_t nid0,
                                                          const uint32_t* __restrict__ mst,
                                                          const uint32_t* __restrict__ mst_cnt)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* row  = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * degree;
  const int64_t nid = nid0 + (int64_t)blockIdx.x * 4 + wave;
  if (nid >= n) return;
  uint32_t n_mst = 0;
  if (mst != nullptr) {
    n_mst = min(mst_cnt[nid], degree);
    for (uint32_t i = lane; i < n_mst; i += 64) row[i] = mst[nid * degree + i];
    __builtin_amdgcn_wave_barrier();
    uint32_t out = n_mst;
    for (uint32_t pj = 0; pj < degree && out < degree; ++pj) {
      const uint32_t v = g[nid * degree + pj];
      bool dup = false;
      for (uint32_t m = lane; m < out; m += 64) dup |= row[m] == v;
      if (__ballot(dup) == 0) {
        if (lane == 0) row[out] = v;
        ++out;
      }
      __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t i = out + lane; i < degree; i += 64) row[i] = kInvalidNode;
    __builtin_amdgcn_wave_barrier();
  } else {
    for (uint32_t i = lane; i < degree; i += 64) row[i] = g[nid * degree + i];
    __builtin_amdgcn_wave_barrier();
  }
  const uint32_t prot = max(n_mst, degree / 2);
  if (prot < degree) {
    uint32_t kr = min(rev_cnt[nid], degree);
    while (kr) {
      kr -= 1;
      const uint32_t rv = rev[(size_t)nid * degree + kr];  // source node of the kr-th reverse edge
      // position of rv in the row (degree if absent)
      uint32_t pos = degree;
      for (uint32_t i = lane; i < degree; i += 64)
        if (row[i] == rv) pos = i;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) pos = min(pos, (uint32_t)__shfl_xor((int)pos, off, kWave));
      if (pos < prot) continue;
      uint32_t num_shift = pos - prot;
      if (pos >= degree) num_shift = degree - prot - 1;
      // shift row[prot .. prot + num_shift) one to the right
      uint32_t tmp[4];
      for (uint32_t c = 0, i = lane; c < 4; ++c, i += 64) tmp[c] = (i < num_shift) ? row[prot + i] : 0;
      __builtin_amdgcn_wave_barrier();
      for (uint32_t c = 0, i = lane; c < 4; ++c, i += 64)
        if (i < num_shift) row[prot + i + 1] = tmp[c];
      if (lane == 0) row[prot] = rv;
      __builtin_amdgcn_wave_barrier();
    }
  }
  for (uint32_t i = lane; i < degree; i += 64) g[nid * degree + i] = row[i];
}

// drop the self match from [n_rows, K + 1] kNN results (rows r0..): keep the first K ids != row
__global__ void strip_self_kernel(const int64_t* __restrict__ ids, int64_t n_rows, int64_t r0, uint32_t K,
                                  uint32_t kp1, uint32_t* __restrict__ knn)
{
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  uint32_t o = 0;
  for (uint32_t j = 0; j < kp1 && o < K; ++j) {
    int64_t v = ids[r * kp1 + j];
    if (v == r0 + r) continue;
    knn[(r0 + r) * K + o++] = (v < 0 || v == INT64_MAX) ? kInvalidNode : (uint32_t)v;
  }
  for (; o < K; ++o) knn[(r0 + r) * K + o] = kInvalidNode;
}

template <typename T>
__global__ void to_float_kernel(const T* __restrict__ in, int64_t n, float* __restrict__ out)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = to_float(in[i]);
}

// exact kNN graph by tiled brute force (small datasets)
template <typename T>
void knn_graph_bruteforce(resources& res, const T* data, int64_t n, int64_t dim, uint32_t K, int metric,
                          uint32_t* knn)
{
  const uint32_t kp1    = (uint32_t)std::min<int64_t>(K + 1, n);
  const bool select_min = metric != M_InnerProduct;
  dev_buf<float> norms;
  if (metric != M_InnerProduct) {
    norms = dev_buf<float>(res, n);
    row_norms<T>(res, data, n, dim, dim, norms.data(), metric == M_CosineExpanded);
  }
  const int64_t m_tile = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(res.workspace_limit / 4) / n));
  dev_buf<float> tile(res, (size_t)m_tile * n), dv(res, (size_t)m_tile * kp1);
  dev_buf<int64_t> di(res, (size_t)m_tile * kp1);
  for (int64_t r0 = 0; r0 < n; r0 += m_tile) {
    int64_t mr = std::min(m_tile, n - r0);
    pairwise_distance<T, T>(res, data + r0 * dim, mr, dim, data, n, dim, dim, norms.data() ? norms.data() + r0 : nullptr,
                            norms.data(),
                            metric == M_InnerProduct ? M_InnerProduct : (metric == M_CosineExpanded ? M_CosineExpanded : M_L2Expanded),
                            tile.data(), n);
    select_k<int64_t, int64_t>(res, tile.data(), nullptr, mr, n, n, (int)kp1, dv.data(), di.data(), select_min);
    hipLaunchKernelGGL(strip_self_kernel, dim3(grid_blocks(mr, 256)), dim3(256), 0, res.stream, di.data(), mr, r0, K,
                       kp1, knn);
  }
}

// approximate kNN graph: IVF-PQ search of the dataset against itself + exact refine (cagra_build.cuh:1629-1760)
void knn_graph_ivf_pq(resources& res, const void* data, elem_t et, int64_t n, int64_t dim, uint32_t K, int metric,
                      uint32_t* knn)
{
  ivf_pq_build_params bp;
  bp.metric                   = metric == M_InnerProduct ? M_InnerProduct : (metric == M_CosineExpanded ? M_CosineExpanded : M_L2Expanded);
  bp.n_lists                  = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(65536, (int64_t)std::sqrt((double)n)));
  bp.kmeans_n_iters           = 10;
  bp.kmeans_trainset_fraction = std::min(1.0, std::max(0.02, 2.0e6 / (double)n));
  bp.pq_bits                  = 8;
  bp.pq_dim                   = (uint32_t)std::max<int64_t>(8, std::min<int64_t>(64, round_up(dim / 2, 8)));
  if (res.tune.cagra_pq_lists > 0) bp.n_lists = (uint32_t)res.tune.cagra_pq_lists;
  auto pq = ivf_pq_build(res, bp, data, et, n, dim, false);
  ivf_pq_search_params sp;
  sp.n_probes                = std::max<uint32_t>(8, bp.n_lists / 50);
  if (res.tune.cagra_pq_probes > 0) sp.n_probes = (uint32_t)res.tune.cagra_pq_probes;
  sp.lut_dtype               = 2;
  sp.internal_distance_dtype = 2;
  // a shape of the wide matrix-core path (ivf_pq_wide.hip; 768 dimensions: pq_dim 64 x pq_len 12): scores summed in fp32 - the
  // screen's margin for fp16 sums (6 % of the bound) lets four times the rows through to the exact re-score (measured at 2M x 768:
  // 34 vs 19 ms per batch of 16384; the LUT scan of rounds 1-5: 60 ms)
  if (pqw_shape(pq->rot_dim)) sp.internal_distance_dtype = 0;
  sp.max_internal_batch_size = 16384;
  const int kp1   = (int)K + 1;
  int k_pq        = std::min(256, 2 * kp1);
  if (res.tune.cagra_kpq > 0) k_pq = std::max(kp1, std::min(256, res.tune.cagra_kpq));
  const int64_t b = 16384;
  dev_buf<int64_t> cand(res, (size_t)b * k_pq), ri(res, (size_t)b * kp1);
  dev_buf<float> cd(res, (size_t)b * k_pq), rd(res, (size_t)b * kp1);
  const size_t esz = elem_size(et);
  for (int64_t r0 = 0; r0 < n; r0 += b) {
    int64_t mr      = std::min(b, n - r0);
    const void* qry = static_cast<const char*>(data) + (size_t)r0 * dim * esz;
    ivf_pq_search(res, sp, *pq, qry, et, mr, k_pq, cand.data(), cd.data());
    refine(res, data, et, n, dim, qry, mr, cand.data(), k_pq, std::min(kp1, k_pq), bp.metric, ri.data(), rd.data());
    hipLaunchKernelGGL(strip_self_kernel, dim3(grid_blocks(mr, 256)), dim3(256), 0, res.stream, ri.data(), mr, r0, K,
                       (uint32_t)std::min(kp1, k_pq), knn);
  }
  sync(res);
}

// graph::optimize (graph_core.cuh:1706-1809); guarantee_connectivity adds the spanning-forest pass of cagra_mst.hip
void optimize_graph(resources& res, const uint32_t* knn, int64_t n, uint32_t K, uint32_t degree, uint32_t* graph,
                    bool guarantee_connectivity)
{
  CUVS_EXPECTS(degree <= K, "graph_degree (%u) must not exceed intermediate_graph_degree (%u)", degree, K);
  CUVS_EXPECTS(degree <= 256 && K <= 1024, "cagra: degree <= 256 and intermediate degree <= 1024 supported");
  CUVS_EXPECTS(n < (int64_t(1) << 32) - 1, "cagra: at most 2^32 - 2 rows (uint32 graph)");
  uint32_t tsize = 64;
  while (tsize < 2 * K) tsize <<= 1;
  size_t smem = (size_t)4 * (2 * K + 2 * tsize) * sizeof(uint32_t);
  auto launch_prune = [&](auto per_tag, int64_t r0, int64_t rows) {
    constexpr int PER = decltype(per_tag)::value;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(prune_kernel<PER>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    hipLaunchKernelGGL(prune_kernel<PER>, dim3(grid_blocks(rows, 4)), dim3(256), smem, res.stream, knn, n, K, degree, graph, r0,
                       res.tune.prune_dbg);
  };
  const int64_t slab = int64_t(1) << 25;  // rows per launch of the wave-per-row kernels (2^23 workgroups)
  for (int64_t r0 = 0; r0 < n; r0 += slab) {
    const int64_t rows = std::min(slab, n - r0);
    if (K <= 64)       launch_prune(std::integral_constant<int, 1>{}, r0, rows);
    else if (K <= 128) launch_prune(std::integral_constant<int, 2>{}, r0, rows);
    else if (K <= 256) launch_prune(std::integral_constant<int, 4>{}, r0, rows);
    else if (K <= 512) launch_prune(std::integral_constant<int, 8>{}, r0, rows);
    else               launch_prune(std::integral_constant<int, 16>{}, r0, rows);
  }
  // reverse edges grouped by destination, ordered by (rank, source): the edge list is sorted a chunk of ranks at a
  // time (a chunk stays below 2^32 edges; 100M rows x degree 64 takes two)
  uint32_t ranks = (uint32_t)std::min<int64_t>(degree, std::max<int64_t>(1, ((int64_t(1) << 32) - 1024) / n));
  if (res.tune.cagra_rank_chunk > 0) ranks = (uint32_t)std::min<int>(res.tune.cagra_rank_chunk, (int)degree);
  const int64_t chunk_edges = n * (int64_t)ranks;
  dev_buf<uint32_t> mst, mst_cnt;
  if (guarantee_connectivity) {
    mst     = dev_buf<uint32_t>(res, (size_t)n * degree);
    mst_cnt = dev_buf<uint32_t>(res, n);
    const int64_t left = cagra_mst_optimize(res, knn, n, K, degree, mst.data(), mst_cnt.data());
    if (left > 1)
      fprintf(stderr, "[cuvs_amd] cagra: guarantee_connectivity left %lld components (rows out of protected slots)\n", (long long)left);
  }
  dev_buf<uint32_t> dest(res, chunk_edges), perm(res, chunk_edges), off(res, n + 2), rev(res, (size_t)n * degree), rev_cnt(res, n);
  HIP_TRY(hipMemsetAsync(rev_cnt.data(), 0, rev_cnt.bytes(), res.stream));
  for (uint32_t r0 = 0; r0 < degree; r0 += ranks) {
    const uint32_t nr = std::min(ranks, degree - r0);
    hipLaunchKernelGGL(edge_dest_kernel, dim3(grid_blocks(n * (int64_t)nr, 256)), dim3(256), 0, res.stream, graph, n, degree,
                       r0, nr, dest.data());
    group_by_label(res, dest.data(), n * (int64_t)nr, (uint32_t)(n + 1), perm.data(), off.data());
    hipLaunchKernelGGL(append_reverse_kernel, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream, perm.data(), off.data(), n,
                       degree, rev.data(), rev_cnt.data());
  }
  for (int64_t r0 = 0; r0 < n; r0 += slab)
    hipLaunchKernelGGL(merge_graph_kernel, dim3(grid_blocks(std::min(slab, n - r0), 4)), dim3(256),
                       (size_t)4 * degree * sizeof(uint32_t), res.stream, graph, n, degree, rev.data(), rev_cnt.data(), r0,
                       (const uint32_t*)mst.data(), (const uint32_t*)mst_cnt.data());
  HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------ search
struct search_args {
  const void* data;
  const uint32_t* graph;
  const void* queries;
  const uint32_t* filter_bits;  // optional bitset (1 keeps)
  void* out_idx;                // uint32 or int64 [nq, k]
  float* out_dist;
  int64_t n, dim;
  uint32_t degree, itopk, width, max_iter, min_iter, k, np2, hash_bits, reset_interval;
  uint32_t n_distill;  // num_random_samplings: random candidates per seed slot, the nearest one is kept (device_common_jit.cuh:60-83)
  uint64_t rand_xor_mask;
  int is_ip, idx64;    // is_ip: 0 L2, 1 inner product, 2 cosine (1 - q.x / (|q| |x|), |x| from `norms`)
  const float* norms;  // [n] canonical |x| (cosine)
  // optional work counters (cuvsAmdCagraWorkCounters; SURVEY 8d: bytes(q) = n_dist * dim * sizeof(T) + n_iter * degree * 4
  // with n_dist, n_iter MEASURED): [0] rows scored, [1] walk iterations (graph rows read), [2] walkers (waves). One atomic
  // per wave and counter at the end of the walk; nullptr: not counted.
  unsigned long long* work = nullptr;
};

__device__ inline uint32_t hash_slot(uint32_t key, uint32_t bits) { return (key ^ (key >> bits)) & ((1u << bits) - 1u); }

// returns true when `key` was not in the table (and is now)
__device__ inline bool hash_insert(uint32_t* table, uint32_t bits, uint32_t key)
{
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t pos        = hash_slot(key, bits);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    uint32_t old = atomicCAS(&table[pos], kInvalidNode, key);
    if (old == kInvalidNode) return true;
    if (old == key) return false;
    pos = (pos + 1) & mask;
  }
  return false;  // table full: treat as seen
}

__device__ inline uint64_t xorshift64(uint64_t u)
{
  u ^= u >> 12;
  u ^= u << 25;
  u ^= u >> 27;
  return u * 0x2545F4914F6CDD1DULL;
}

// distances of the nodes idx[first .. first+count) to the query; 8 teams of 8 lanes, one row per team.
// Arithmetic (oracle twin: oracle_cagra.c): team lane t accumulates the elements of its 16-byte pieces in
// order with fmaf, then the 8 partial sums are combined by the xor butterfly (1, 2, 4).
template <typename T>
__device__ inline uint32_t team_distances(const T* __restrict__ data, int64_t dim, const float* __restrict__ qf,
                                      uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx, uint32_t first,
                                      uint32_t count, int is_ip, int lane, const float* __restrict__ norms = nullptr,
                                      float qn = 1.f)
{
  constexpr int VL = 16 / sizeof(T);
  const int team = lane >> 3, tl = lane & 7;
  const bool vec = (dim % VL == 0) && ((reinterpret_cast<uintptr_t>(data) & 15) == 0);
  uint32_t n_scored = 0u;  // rows whose distance was computed (children already in the hash are skipped)
  for (uint32_t c0 = 0; c0 < count; c0 += 8) {
    const uint32_t c    = c0 + team;
    const uint32_t node = c < count ? (idx[first + c] & ~kParentFlag) : kInvalidNode;
    const bool ok       = c < count && idx[first + c] != kInvalidNode;
    n_scored += (uint32_t)__popcll(__ballot(ok && tl == 0));
    float acc           = 0.f;
    if (ok) {
      const T* row = data + (int64_t)node * dim;
      int64_t d0   = (int64_t)tl * VL;
      auto accumulate = [&](const uint4& w, const int64_t dd) {
        const T* el = reinterpret_cast<const T*>(&w);
#pragma unroll
        for (int e = 0; e < VL; ++e) {
          const float x = to_float(el[e]), q = qf[dd + e];
          if (is_ip) {
            acc = __fmaf_rn(x, q, acc);
          } else {
            float t = x - q;
            acc     = __fmaf_rn(t, t, acc);
          }
        }
      };
      if (vec) {
        // four 16-byte pieces in flight per lane (the walk is bound by the latency of these dependent row gathers);
        // the pieces are accumulated in the same order as one at a time
        constexpr int64_t S = 8 * VL;
        for (; d0 + 3 * S < dim; d0 += 4 * S) {
          const uint4 w0 = *reinterpret_cast<const uint4*>(row + d0);
          const uint4 w1 = *reinterpret_cast<const uint4*>(row + d0 + S);
          const uint4 w2 = *reinterpret_cast<const uint4*>(row + d0 + 2 * S);
          const uint4 w3 = *reinterpret_cast<const uint4*>(row + d0 + 3 * S);
          accumulate(w0, d0); accumulate(w1, d0 + S); accumulate(w2, d0 + 2 * S); accumulate(w3, d0 + 3 * S);
        }
      }
      for (; d0 < dim; d0 += 8 * VL) {
        T el[VL];
        if (vec) {
          *reinterpret_cast<uint4*>(el) = *reinterpret_cast<const uint4*>(row + d0);
        } else {
#pragma unroll
          for (int e = 0; e < VL; ++e) el[e] = d0 + e < dim ? row[d0 + e] : T(0);
        }
#pragma unroll
        for (int e = 0; e < VL; ++e) {
          if (d0 + e < dim) {
            const float x = to_float(el[e]), q = qf[d0 + e];
            if (is_ip) {
              acc = __fmaf_rn(x, q, acc);
            } else {
              float t = x - q;
              acc     = __fmaf_rn(t, t, acc);
            }
          }
        }
      }
    }
    acc = acc + __shfl_xor(acc, 1, kWave);
    acc = acc + __shfl_xor(acc, 2, kWave);
    acc = acc + __shfl_xor(acc, 4, kWave);
    if (is_ip == 2 && ok) acc = 1.0f - acc / (qn * norms[node]);  // the brute-force cosine epilogue
    if (tl == 0 && c < count) keys[first + c] = ok ? float_to_key(is_ip == 1 ? -acc : acc) : 0xffffffffu;
  }
  return n_scored;
}

__device__ inline float wave_query_norm(const float* qf, int64_t dim, int lane)
{
  float s = 0.f;
  for (int64_t d = lane; d < dim; d += 64) s = __fmaf_rn(qf[d], qf[d], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s = s + __shfl_xor(s, off, kWave);
  return sqrtf(s);
}

template <typename T>
__global__ __launch_bounds__(64) void cagra_search_kernel(search_args a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane   = threadIdx.x;
  const int64_t qi = blockIdx.x;
  float* qf        = reinterpret_cast<float*>(smem);
  uint32_t* keys   = reinterpret_cast<uint32_t*>(qf + ((a.dim + 3) & ~int64_t(3)));
  uint32_t* idx    = keys + a.np2;
  uint32_t* table  = idx + a.np2;
  const uint32_t hsize = 1u << a.hash_bits;
  const T* data    = static_cast<const T*>(a.data);

  for (int64_t d = lane; d < a.dim; d += 64) qf[d] = to_float(static_cast<const T*>(a.queries)[qi * a.dim + d]);
  for (uint32_t i = lane; i < a.np2; i += 64) { keys[i] = 0xffffffffu; idx[i] = kInvalidNode; }
  for (uint32_t i = lane; i < hsize; i += 64) table[i] = kInvalidNode;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  float qn = 1.f;  // cosine: canonical |q| (64 strided fma partial sums + butterfly, as row_norms)
  if (a.is_ip == 2) qn = wave_query_norm(qf, a.dim, lane);

  // ---- seeds (compute_distance_to_random_nodes, device_common_jit.cuh:36-104; call site search_single_cta_jit.cuh:157-171):
  // every slot of the result buffer - itopk + search_width * degree of them - draws num_random_samplings pseudo-random
  // nodes, gid = slot + n_slots * j -> xorshift64(gid ^ rand_xor_mask) % n, and keeps the nearest (the first on ties); a
  // winner that is already in the visited table is dropped. The stream does not depend on the query (as in the reference).
  const uint32_t n_cand = a.width * a.degree;
  const uint32_t n_seed = a.itopk + n_cand;
  uint32_t* tkeys   = table + hsize;
  uint32_t* tidx    = tkeys + n_seed;
  uint32_t* parents = tidx + n_seed;  // [width]
  uint32_t n_dist = 0u;
  for (uint32_t j = 0; j < a.n_distill; ++j) {
    for (uint32_t i = lane; i < n_seed; i += 64) {
      const uint64_t gid = (uint64_t)i + (uint64_t)n_seed * j;
      tidx[i]            = (uint32_t)(xorshift64(gid ^ a.rand_xor_mask) % (uint64_t)a.n);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    n_dist += team_distances<T>(data, a.dim, qf, tkeys, tidx, 0, n_seed, a.is_ip, lane, a.norms, qn);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n_seed; i += 64)
      if (tkeys[i] < keys[i]) { keys[i] = tkeys[i]; idx[i] = tidx[i]; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = lane; i < n_seed; i += 64) {
    const uint32_t node = idx[i];
    if (node != kInvalidNode && !hash_insert(table, a.hash_bits, node)) { idx[i] = kInvalidNode; keys[i] = 0xffffffffu; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t n_rows_read = 0u;  // graph rows read (parents expanded)

  uint32_t iter         = 0;
  while (true) {
    wave_bitonic_sort<uint32_t>(keys, idx, (int)a.np2);
    if (iter >= a.max_iter) break;
    // hash reset (SMALL hash mode of the reference): keep only the current top list
    if (iter > 0 && a.reset_interval > 0 && (iter % a.reset_interval) == 0) {
      for (uint32_t i = lane; i < hsize; i += 64) table[i] = kInvalidNode;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < a.itopk; i += 64)
        if (idx[i] != kInvalidNode) hash_insert(table, a.hash_bits, idx[i] & ~kParentFlag);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- pick up to `width` best entries that were not parents yet (pickup_next_parents)
    uint32_t n_parents = 0;
    for (uint32_t base = 0; base < a.itopk && n_parents < a.width; base += 64) {
      const uint32_t i = base + lane;
      const bool cand  = i < a.itopk && idx[i] != kInvalidNode && !(idx[i] & kParentFlag);
      unsigned long long m = __ballot(cand);
      while (m != 0ull && n_parents < a.width) {
        const int src = (int)__ffsll((long long)m) - 1;
        m &= m - 1ull;
        const uint32_t pos = base + src;
        if (lane == 0) { parents[n_parents] = idx[pos]; idx[pos] |= kParentFlag; }
        ++n_parents;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (n_parents == 0 && iter >= a.min_iter) break;
    // ---- children -> candidate region (dedup through the hash)
    for (uint32_t i = lane; i < n_cand; i += 64) {
      const uint32_t w = i / a.degree, c = i % a.degree;
      uint32_t child   = kInvalidNode;
      if (w < n_parents) {
        child = a.graph[(int64_t)parents[w] * a.degree + c];
        if (child >= a.n || !hash_insert(table, a.hash_bits, child)) child = kInvalidNode;
      }
      idx[a.itopk + i]  = child;
      keys[a.itopk + i] = 0xffffffffu;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    n_dist += team_distances<T>(data, a.dim, qf, keys, idx, a.itopk, n_cand, a.is_ip, lane, a.norms, qn);
    n_rows_read += n_parents;
    ++iter;
  }
  if (a.work != nullptr && lane == 0) {
    atomicAdd(&a.work[0], (unsigned long long)n_dist);
    atomicAdd(&a.work[1], (unsigned long long)n_rows_read);
    atomicAdd(&a.work[2], 1ull);
  }

  // ---- results: first k entries of the sorted list that pass the filter
  uint32_t written = 0;
  for (uint32_t base = 0; base < a.itopk && written < a.k; base += 64) {
    const uint32_t i = base + lane;
    bool ok          = i < a.itopk && idx[i] != kInvalidNode;
    uint32_t node    = ok ? (idx[i] & ~kParentFlag) : 0;
    if (ok && a.filter_bits) ok = (a.filter_bits[node >> 5] >> (node & 31)) & 1u;
    const unsigned long long m = __ballot(ok);
    const uint32_t rank        = written + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (ok && rank < a.k) {
      float d = key_to_float(keys[i]);
      if (a.is_ip == 1) d = -d;
      a.out_dist[qi * a.k + rank] = d;
      if (a.idx64) static_cast<int64_t*>(a.out_idx)[qi * a.k + rank] = node;
      else         static_cast<uint32_t*>(a.out_idx)[qi * a.k + rank] = node;
    }
    written += (uint32_t)__popcll(m);
  }
  for (uint32_t r = min(written, a.k) + lane; r < a.k; r += 64) {
    a.out_dist[qi * a.k + r] = FLT_MAX;
    if (a.idx64) static_cast<int64_t*>(a.out_idx)[qi * a.k + r] = -1;
    else         static_cast<uint32_t*>(a.out_idx)[qi * a.k + r] = kInvalidNode;
  }
}

// ------------------------------------------------------------------ multi-wave search (small batches)
// The reference's MULTI_CTA mode (search_multi_cta_jit.cuh:33-365) gives every query several CTAs, each with its
// own 32-entry top list and ONE parent per iteration, which claim parents through a per-query "traversed" hash
// table in global memory; a final top-k merges the CTAs' lists. Here the unit is a wave: one workgroup per
// query, W waves, each wave runs the reference's per-CTA loop on its own LDS slice, and the traversed table
// lives in LDS (waves of one workgroup share it, so no global table and no extra merge kernel). Waves do not
// synchronise inside the walk; as in the reference the claim order is a race, so results are equal in quality
// (recall tests) but not bit-reproducible.
constexpr uint32_t kMwTopk = 32;  // per-wave top list (the reference's multi_cta_itopk_size)

struct mw_args {
  search_args s;
  uint32_t n_waves, np2_local, vis_bits, trav_bits, merge_np2;
};

// hash table with removal (hashmap.hpp:37-134 SUPPORT_REMOVE semantics): a removed key leaves key|MSB behind
__device__ inline bool trav_insert(uint32_t* table, uint32_t bits, uint32_t key)
{
  const uint32_t mask = (1u << bits) - 1u, removed = key | kParentFlag;
  uint32_t pos = hash_slot(key, bits);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    uint32_t old = atomicCAS(&table[pos], kInvalidNode, key);
    if (old == kInvalidNode) return true;
    if (old == key) return false;
    old = atomicCAS(&table[pos], removed, key);
    if (old == removed) return true;
    if (old == key) return false;
    pos = (pos + 1) & mask;
  }
  return false;
}
__device__ inline bool trav_contains(const uint32_t* table, uint32_t bits, uint32_t key)
{
  const uint32_t mask = (1u << bits) - 1u, removed = key | kParentFlag;
  uint32_t pos = hash_slot(key, bits);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint32_t v = table[pos];
    if (v == key) return true;
    if (v == kInvalidNode || v == removed) return false;
    pos = (pos + 1) & mask;
  }
  return false;
}
__device__ inline void trav_remove(uint32_t* table, uint32_t bits, uint32_t key)
{
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t pos = hash_slot(key, bits);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint32_t old = atomicCAS(&table[pos], key, key | kParentFlag);
    if (old == key || old == kInvalidNode) return;
    pos = (pos + 1) & mask;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void cagra_search_multi_kernel(mw_args m)
{
  const search_args& a = m.s;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane   = threadIdx.x & 63;
  const int wave   = threadIdx.x >> 6;
  const int64_t qi = blockIdx.x;
  const uint32_t W = m.n_waves, np2 = m.np2_local, vsize = 1u << m.vis_bits, tsize = 1u << m.trav_bits;
  float* qf        = reinterpret_cast<float*>(smem);
  uint32_t* trav   = reinterpret_cast<uint32_t*>(qf + ((a.dim + 3) & ~int64_t(3)));
  uint32_t* mkeys  = trav + tsize;             // merge area [merge_np2]
  uint32_t* midx   = mkeys + m.merge_np2;
  uint32_t* wbase  = midx + m.merge_np2 + (size_t)wave * (2 * np2 + vsize);
  uint32_t* m_tmp  = midx + m.merge_np2 + (size_t)W * (2 * np2 + vsize);  // seed candidates: 2 * degree words per wave
  uint32_t* keys   = wbase;                    // this wave's list: [np2] = 32 top + degree candidates
  uint32_t* idx    = keys + np2;
  uint32_t* vis    = idx + np2;                // this wave's visited table (rebuilt every iteration)
  const T* data    = static_cast<const T*>(a.data);

  for (int64_t d = threadIdx.x; d < a.dim; d += blockDim.x) qf[d] = to_float(static_cast<const T*>(a.queries)[qi * a.dim + d]);
  for (uint32_t i = threadIdx.x; i < tsize; i += blockDim.x) trav[i] = kInvalidNode;
  for (uint32_t i = lane; i < np2; i += 64) { keys[i] = 0xffffffffu; idx[i] = kInvalidNode; }
  for (uint32_t i = lane; i < vsize; i += 64) vis[i] = kInvalidNode;
  __syncthreads();

  float qn = 1.f;
  if (a.is_ip == 2) qn = wave_query_norm(qf, a.dim, lane);

  // ---- seeds (search_multi_cta_jit.cuh:134-148): `degree` slots per wave, num_random_samplings candidates per slot with
  // gid = wave + W * (slot + degree * j), the nearest kept; a winner already in this wave's visited table is dropped
  uint32_t n_dist = 0u;
  {
    const uint32_t n_seed = min(a.degree, np2);
    uint32_t* tkeys = m_tmp + (size_t)wave * 2 * a.degree;
    uint32_t* tidx  = tkeys + a.degree;
    for (uint32_t j = 0; j < a.n_distill; ++j) {
      for (uint32_t i = lane; i < n_seed; i += 64) {
        const uint64_t gid = (uint64_t)wave + (uint64_t)W * ((uint64_t)i + (uint64_t)n_seed * j);
        tidx[i]            = (uint32_t)(xorshift64(gid ^ a.rand_xor_mask) % (uint64_t)a.n);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      n_dist += team_distances<T>(data, a.dim, qf, tkeys, tidx, 0, n_seed, a.is_ip, lane, a.norms, qn);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < n_seed; i += 64)
        if (tkeys[i] < keys[i]) { keys[i] = tkeys[i]; idx[i] = tidx[i]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n_seed; i += 64) {
      const uint32_t node = idx[i];
      if (node != kInvalidNode && !hash_insert(vis, m.vis_bits, node)) { idx[i] = kInvalidNode; keys[i] = 0xffffffffu; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  uint32_t n_rows_read = 0u;

  uint32_t iter = 0;
  while (true) {
    wave_bitonic_sort<uint32_t>(keys, idx, (int)np2);
    if (iter + 1 >= a.max_iter) break;
    // ---- next parent (pickup_next_parent, search_multi_cta_helpers.cuh:16-80): the best entry of the first 64
    // that is not a parent yet and that THIS wave manages to claim in the traversed table; entries claimed by
    // another wave are dropped from this list
    uint32_t parent = kInvalidNode;
    {
      const uint32_t e   = lane < (int)np2 ? idx[lane] : kInvalidNode;
      unsigned long long c = __ballot(e != kInvalidNode && !(e & kParentFlag));
      int chosen = -1;
      while (c != 0ull) {
        const int src = (int)__ffsll((long long)c) - 1;
        c &= c - 1ull;
        const uint32_t node = __builtin_amdgcn_readlane(e, src);
        bool mine = false;
        if (lane == 0) mine = trav_insert(trav, m.trav_bits, node);
        mine = __builtin_amdgcn_readfirstlane((int)mine) != 0;
        if (mine) { chosen = src; parent = node; break; }
        if (lane == 0) { idx[src] = kInvalidNode; keys[src] = 0xffffffffu; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (chosen >= 0) {
        if (chosen < (int)kMwTopk) {
          if (lane == 0) idx[chosen] = parent | kParentFlag;
        } else {
          // parent found in the overflow half: move it into the last free slot of the top half, if any
          const uint32_t t     = lane < (int)kMwTopk ? idx[lane] : 0u;
          unsigned long long f = __ballot(lane < (int)kMwTopk && t == kInvalidNode);
          if (f == 0ull) {
            if (lane == 0) trav_remove(trav, m.trav_bits, parent);
            parent = kInvalidNode;
          } else if (lane == 0) {
            const int j = 63 - __builtin_clzll(f);
            idx[j] = parent | kParentFlag; keys[j] = keys[chosen];
            idx[chosen] = kInvalidNode;    keys[chosen] = 0xffffffffu;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (parent == kInvalidNode && iter >= a.min_iter) break;
    // ---- rebuild the visited table from the list; parents pushed out of the top half are released
    for (uint32_t i = lane; i < vsize; i += 64) vis[i] = kInvalidNode;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < np2; i += 64) {
      const uint32_t e = idx[i];
      if (e == kInvalidNode) continue;
      if (i >= kMwTopk && (e & kParentFlag)) {
        trav_remove(trav, m.trav_bits, e & ~kParentFlag);
        idx[i] = kInvalidNode; keys[i] = 0xffffffffu;
      } else {
        hash_insert(vis, m.vis_bits, e & ~kParentFlag);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- children of the parent -> candidate region
    for (uint32_t c = lane; c < np2 - kMwTopk; c += 64) {
      uint32_t child = kInvalidNode;
      if (parent != kInvalidNode && c < a.degree) {
        child = a.graph[(int64_t)parent * a.degree + c];
        if (child >= a.n || !hash_insert(vis, m.vis_bits, child)) child = kInvalidNode;
      }
      idx[kMwTopk + c]  = child;
      keys[kMwTopk + c] = 0xffffffffu;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    n_dist += team_distances<T>(data, a.dim, qf, keys, idx, kMwTopk, a.degree, a.is_ip, lane, a.norms, qn);
    n_rows_read += parent != kInvalidNode ? 1u : 0u;
    // ---- drop what another wave has expanded meanwhile; a parent that fails the filter leaves the list
    for (uint32_t i = lane; i < np2; i += 64) {
      const uint32_t e = idx[i];
      if (e == kInvalidNode) continue;
      bool drop = false;
      if (!(e & kParentFlag)) {
        drop = trav_contains(trav, m.trav_bits, e);
      } else if (a.filter_bits != nullptr && (e & ~kParentFlag) == parent) {
        drop = !((a.filter_bits[parent >> 5] >> (parent & 31)) & 1u);
      }
      if (drop) { idx[i] = kInvalidNode; keys[i] = 0xffffffffu; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ++iter;
  }

  if (a.work != nullptr && lane == 0) {
    atomicAdd(&a.work[0], (unsigned long long)n_dist);
    atomicAdd(&a.work[1], (unsigned long long)n_rows_read);
    atomicAdd(&a.work[2], 1ull);
  }
  // ---- this wave's 32 results: parents are unique by construction, the others must win the traversed table
  // (duplicates across waves lose); filtered-out nodes are dropped
  {
    uint32_t e = lane < (int)kMwTopk ? idx[lane] : kInvalidNode;
    uint32_t kk = lane < (int)kMwTopk ? keys[lane] : 0xffffffffu;
    bool ok = e != kInvalidNode;
    const uint32_t node = e & ~kParentFlag;
    if (ok && a.filter_bits) ok = (a.filter_bits[node >> 5] >> (node & 31)) & 1u;
    if (ok && !(e & kParentFlag)) ok = trav_insert(trav, m.trav_bits, node);
    if (lane < (int)kMwTopk) {
      mkeys[wave * kMwTopk + lane] = ok ? kk : 0xffffffffu;
      midx[wave * kMwTopk + lane]  = ok ? node : kInvalidNode;
    }
  }
  for (uint32_t i = W * kMwTopk + threadIdx.x; i < m.merge_np2; i += blockDim.x) { mkeys[i] = 0xffffffffu; midx[i] = kInvalidNode; }
  __syncthreads();
  if (wave != 0) return;
  wave_bitonic_sort<uint32_t>(mkeys, midx, (int)m.merge_np2);
  for (uint32_t r = lane; r < a.k; r += 64) {
    const bool ok  = r < m.merge_np2 && midx[r] != kInvalidNode;
    float d        = ok ? key_to_float(mkeys[r]) : FLT_MAX;
    if (ok && a.is_ip == 1) d = -d;
    a.out_dist[qi * a.k + r] = d;
    if (a.idx64) static_cast<int64_t*>(a.out_idx)[qi * a.k + r] = ok ? (int64_t)midx[r] : -1;
    else         static_cast<uint32_t*>(a.out_idx)[qi * a.k + r] = ok ? midx[r] : kInvalidNode;
  }
}

template <typename T>
void launch_search_multi(resources& res, const mw_args& m, int64_t nq, size_t smem)
{
  auto kern = cagra_search_multi_kernel<T>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  profile_begin(res, "cagra_search_multi_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64 * m.n_waves), smem, res.stream, m);
  profile_end(res, "cagra_search_multi_kernel");
  HIP_TRY(hipGetLastError());
}

template <typename T>
void launch_search(resources& res, const search_args& a, int64_t nq, size_t smem)
{
  auto kern = cagra_search_kernel<T>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  profile_begin(res, "cagra_search_kernel");
  hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), smem, res.stream, a);
  profile_end(res, "cagra_search_kernel");
  HIP_TRY(hipGetLastError());
}

}  // namespace

__global__ void popcount_kernel(const uint32_t* __restrict__ bits, int64_t n_bits, unsigned long long* __restrict__ out)
{
  const int64_t n_words = (n_bits + 31) / 32;
  unsigned long long c  = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
    uint32_t v = bits[w];
    if (w == n_words - 1 && (n_bits & 31)) v &= (1u << (n_bits & 31)) - 1u;
    c += (unsigned)__popc(v);
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

int64_t count_set_bits(resources& res, const uint32_t* bits, int64_t n_bits)
{
  dev_buf<unsigned long long> cnt(res, 1);
  HIP_TRY(hipMemsetAsync(cnt.data(), 0, sizeof(unsigned long long), res.stream));
  hipLaunchKernelGGL(popcount_kernel, dim3(256), dim3(256), 0, res.stream, bits, n_bits, cnt.data());
  unsigned long long h = 0;
  HIP_TRY(hipMemcpyAsync(&h, cnt.data(), sizeof(h), hipMemcpyDeviceToHost, res.stream));
  HIP_TRY(hipStreamSynchronize(res.stream));
  return (int64_t)h;
}

// ------------------------------------------------------------------ search plan (host only)
// The reference's plan rules restated (search_plan.cuh): algorithm choice of AUTO (:121-131), adjust_search_params
// (:199-245: max_iterations, the filter-rate dependent itopk of MULTI_CTA, itopk rounded to 32) and calc_hashmap_params
// (:248-340: small-hash bit length and reset interval of SINGLE_CTA, visited / traversed hash sizes and CTAs per query
// of MULTI_CTA). `ref_*` fields are what the reference would run with; `run_*` fields are what this library launches
// (see cagra_search for the two places it deliberately differs: AUTO and the LDS hash sizing).
struct cagra_plan {
  uint32_t itopk, max_iterations, search_width;
  int ref_algo;                                   // SINGLE_CTA / MULTI_CTA / MULTI_KERNEL after AUTO resolution
  uint32_t ref_small_hash_bitlen, ref_hash_bitlen, ref_small_hash_reset_interval, ref_mc_num_cta_per_query;
  uint32_t mc_max_iterations;                     // max_iterations if the multi-CTA rule applies (itopk 32 per CTA)
  int run_algo;
  uint32_t run_waves_per_query;
};

inline uint32_t hash_size_of(uint32_t bitlen) { return 1u << bitlen; }  // hashmap.hpp:23

cagra_plan make_cagra_plan(const cuvsCagraSearchParams& p, int64_t n_rows, uint32_t degree, uint32_t topk, int64_t n_queries,
                           int num_cus, float filtering_rate)
{
  cagra_plan pl{};
  const uint32_t width = (uint32_t)std::max<size_t>(1, p.search_width);
  pl.search_width      = width;
  size_t itopk         = std::max<size_t>(p.itopk_size ? p.itopk_size : 64, (size_t)topk);
  // ---- algorithm: search_plan.cuh:121-131 (persistent mode is not offered by the C search entry point)
  int algo = (int)p.algo;
  const size_t max_queries = p.max_queries ? p.max_queries : (size_t)n_queries;  // cagra.cuh: max_queries = min(n_queries, ...)
  if (algo == (int)AUTO) algo = (itopk <= 512 && max_queries >= (size_t)num_cus * 2) ? (int)SINGLE_CTA : (int)MULTI_CTA;
  pl.ref_algo = algo;
  auto reach_iterations = [&](uint32_t it) {
    int64_t reach = 1;
    while (reach < n_rows) { reach *= std::max<int64_t>(2, degree / 2); it += 1; }
    return it;
  };
  // ---- adjust_search_params :199-245
  uint32_t max_it = (uint32_t)p.max_iterations;
  if (p.max_iterations == 0) max_it = reach_iterations(algo == (int)MULTI_CTA ? 32u / 1u : (uint32_t)(itopk / width));
  // the reference tests the ORIGINAL field (search_plan.cuh:216): with max_iterations = 0 and min_iterations > 0 the walk
  // runs exactly min_iterations, not the reach-based count
  if ((uint32_t)p.max_iterations < (uint32_t)p.min_iterations) max_it = (uint32_t)p.min_iterations;
  pl.max_iterations    = std::max<uint32_t>(max_it, (uint32_t)p.max_iterations);
  pl.mc_max_iterations = (uint32_t)p.max_iterations < (uint32_t)p.min_iterations
                           ? (uint32_t)p.min_iterations
                           : (p.max_iterations ? (uint32_t)p.max_iterations : reach_iterations(32u));
  if (algo == (int)MULTI_CTA && filtering_rate > 0.f && filtering_rate < 1.f) {
    size_t adj = (size_t)((float)topk / (1.0 - filtering_rate) + (float)(itopk - topk) / std::sqrt(1.0 - filtering_rate));
    if (adj % 32) adj += 32 - adj % 32;
    if (itopk < adj) itopk = adj;
  }
  if (itopk % 32) itopk += 32 - itopk % 32;
  pl.itopk = (uint32_t)itopk;
  // ---- calc_hashmap_params :248-340
  const float fill = p.hashmap_max_fill_rate > 0.f ? p.hashmap_max_fill_rate : 0.5f;
  const uint32_t min_user = (uint32_t)p.hashmap_min_bitlen;
  pl.ref_small_hash_reset_interval = 1024 * 1024;
  if (algo == (int)MULTI_CTA) {
    pl.ref_mc_num_cta_per_query = (uint32_t)std::max<size_t>(width, (itopk + 31) / 32);
    const uint32_t max_visited  = 32 + degree * 2;
    uint32_t sb = 8;
    while ((float)max_visited > hash_size_of(sb) * fill) ++sb;
    pl.ref_small_hash_bitlen = sb;
    const size_t max_trav = (size_t)pl.ref_mc_num_cta_per_query * std::max<size_t>(32, pl.max_iterations);
    uint32_t hb = std::max<uint32_t>(11, min_user);
    while ((float)max_trav > hash_size_of(hb) * fill) ++hb;
    pl.ref_hash_bitlen = hb;
  } else {
    uint32_t hb = 0;
    if (p.hashmap_mode == AUTO_HASH || p.hashmap_mode == SMALL) {
      const size_t max_visited = itopk + (size_t)width * degree;
      hb = std::max<uint32_t>(8, min_user);
      while ((float)max_visited > hash_size_of(hb) * fill) ++hb;
      if (hb > 13) {
        CUVS_EXPECTS(p.hashmap_mode == AUTO_HASH, "small-hash cannot be used because the required hash size exceeds the limit (%u)",
                     hash_size_of(13));
        hb = 0;
      } else {
        pl.ref_small_hash_bitlen = hb;
        uint32_t interval = 1;
        while ((float)(itopk + (size_t)width * degree * (interval + 1)) <= hash_size_of(hb) * fill) ++interval;
        pl.ref_small_hash_reset_interval = interval;
      }
    }
    if (hb == 0) {
      const size_t max_visited = itopk + (size_t)width * degree * pl.max_iterations;
      hb = std::max<uint32_t>(11, min_user);
      while ((float)max_visited > hash_size_of(hb) * fill) ++hb;
      CUVS_EXPECTS(hb <= 20, "hash_bitlen cannot be largen than 20 (1M). You can decrease itopk_size, search_width or "
                             "max_iterations to reduce the required hashmap size.");
    }
    pl.ref_hash_bitlen = hb;
  }
  return pl;
}

// rows -> source ids of an index that carries them (search_multi_cta.cuh:266-272, search_multi_kernel.cuh:658-668); slots without a
// neighbour stay as they are
__global__ void cagra_source_ids_kernel(void* out_idx, int64_t n_out, int idx64, const uint32_t* __restrict__ source, int64_t n)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  if (idx64) {
    int64_t* o = static_cast<int64_t*>(out_idx);
    if (o[t] >= 0 && o[t] < n) o[t] = (int64_t)source[o[t]];
  } else {
    uint32_t* o = static_cast<uint32_t*>(out_idx);
    if ((int64_t)o[t] < n) o[t] = source[o[t]];
  }
}

void cagra_search(resources& res, const cagra_index& idx, const cuvsCagraSearchParams& p, const void* queries,
                  int64_t nq, int k, void* out_idx, bool idx64, float* out_dist, const uint32_t* filter_bits)
{
  if (nq == 0) return;
  CUVS_EXPECTS(idx.graph.data() != nullptr && idx.data != nullptr, "cagra index has no graph/dataset");
  CUVS_EXPECTS(k >= 1, "k must be positive");
  CUVS_EXPECTS(nq < (int64_t(1) << 24), "cagra::search: split the query batch (max 2^24 queries per call)");
  search_args a;
  a.data = idx.data; a.graph = idx.graph.data(); a.queries = queries; a.filter_bits = filter_bits;
  a.out_idx = out_idx; a.out_dist = out_dist; a.n = idx.n; a.dim = idx.dim; a.degree = idx.degree;
  a.width = (uint32_t)std::max<size_t>(1, p.search_width);
  a.n_distill = (uint32_t)std::max<uint32_t>(1u, p.num_random_samplings);
  // share of rows the bitset removes: the reference derives it from the bitset's population count when the caller
  // does not give one (cagra.cuh:374-381; the C search params have no such field) and widens the multi-CTA itopk
  float filtering_rate = 0.f;
  if (filter_bits != nullptr) {
    const int64_t kept = count_set_bits(res, filter_bits, idx.n);
    filtering_rate     = std::min(std::max((float)(idx.n - kept) / (float)idx.n, 0.0f), 0.999f);
  }
  const cagra_plan pl = make_cagra_plan(p, idx.n, idx.degree, (uint32_t)k, nq, res.num_cus, filtering_rate);
  const uint32_t itopk = pl.itopk;
  CUVS_EXPECTS(itopk <= 1024, "cagra::search: itopk_size up to 1024 is supported");
  a.itopk    = itopk;
  a.min_iter = (uint32_t)p.min_iterations;
  a.max_iter = pl.max_iterations;
  a.k        = (uint32_t)k;
  a.np2      = (uint32_t)next_pow2((int)(itopk + a.width * idx.degree));
  // LDS hash of the walk: exact (open addressing, no false positives), so its size and reset interval change the
  // number of re-visited rows but never a result; sized for <= 50 % fill like the reference's small hash (plan above)
  uint32_t bits = 11;
  while ((1u << bits) < 2 * (itopk + 2 * a.width * idx.degree)) ++bits;
  bits = std::max<uint32_t>(bits, (uint32_t)p.hashmap_min_bitlen);
  CUVS_EXPECTS(bits <= 15, "cagra::search: hash table does not fit the LDS budget");
  a.hash_bits      = bits;
  a.reset_interval = std::max<uint32_t>(1, ((1u << bits) / 2 - itopk) / (a.width * idx.degree));
  a.rand_xor_mask  = p.rand_xor_mask;
  a.is_ip          = idx.metric == M_InnerProduct ? 1 : (idx.metric == M_CosineExpanded ? 2 : 0);
  a.norms          = idx.norms.data();
  a.work           = res.cagra_work;
  CUVS_EXPECTS(a.is_ip != 2 || a.norms != nullptr, "cagra::search: cosine index without dataset norms");
  a.idx64          = idx64 ? 1 : 0;
  // ---- algorithm choice: AUTO follows the reference (search_plan.cuh:121-131: one walker per query once the batch
  // alone fills the GPU, several per query otherwise) so that matched parameters walk the graph the same way.
  // CUVS_AMD_CAGRA_AUTO=multi makes AUTO take the multi-wave walk at every batch size (on MI355X it measured faster at
  // equal recall: 1M x 768 fp16, itopk 64, batch 10k: 18.9 vs 26.7 ms).
  int algo = pl.ref_algo;
  if ((int)p.algo == (int)AUTO) {
    if (res.tune.cagra_auto_multi) algo = (int)MULTI_CTA;
    if (!(itopk <= 16 * kMwTopk && idx.degree <= 224 && (size_t)k <= itopk)) algo = (int)SINGLE_CTA;  // limits of the multi-wave walk
  }
  // MULTI_KERNEL (search_multi_kernel.cuh) exists in the reference because an itopk list above 512 entries does not
  // fit one CTA's shared memory there, so the walk is split into sample / top-k / pickup / distance kernels with the
  // list in global memory. 160 KB of LDS holds the 1024-entry list, its merge buffer and the hash of one walk, so the
  // same request runs as the single-workgroup walk here: same traversal, no round trips through HBM between steps.
  if (algo == (int)MULTI_KERNEL) algo = (int)SINGLE_CTA;
  CUVS_EXPECTS(algo == (int)SINGLE_CTA || algo == (int)MULTI_CTA,
               "cagra::search: algo must be SINGLE_CTA, MULTI_CTA, MULTI_KERNEL or AUTO");
  if (algo == (int)MULTI_CTA && idx.degree <= 224) {
    mw_args m;
    m.s = a;
    // waves per query: the reference's num_cta_per_query = max(search_width, itopk / 32) (search_multi_cta.cuh:124),
    // raised while the launch would leave most of the GPU idle
    uint32_t W = (uint32_t)std::max<size_t>(std::max<size_t>(p.search_width, 1), (itopk + kMwTopk - 1) / kMwTopk);
    while (W < 16 && nq * (int64_t)W * 2 <= 4 * (int64_t)res.num_cus) W *= 2;
    W = std::min<uint32_t>(W, 16);
    CUVS_EXPECTS((uint32_t)k <= W * kMwTopk, "`num_cta_per_query` (%u) * 32 must be equal to or greater than `topk` (%d)", W, k);
    m.n_waves   = W;
    m.np2_local = (uint32_t)next_pow2((int)(kMwTopk + idx.degree));
    m.vis_bits  = 8;
    while ((1u << m.vis_bits) < 2 * (kMwTopk + idx.degree)) ++m.vis_bits;
    m.s.max_iter = pl.mc_max_iterations;  // search_plan.cuh:199-215 with the multi-CTA list size
    m.trav_bits  = 11;  // every wave claims <= max_iter parents and inserts <= 32 results; keep the fill <= 50 %
    while ((1u << m.trav_bits) < 2 * W * (m.s.max_iter + kMwTopk)) ++m.trav_bits;
    m.merge_np2 = (uint32_t)next_pow2((int)(W * kMwTopk));
    size_t msmem = (size_t)((idx.dim + 3) & ~int64_t(3)) * 4 + ((size_t)4 << m.trav_bits) + (size_t)m.merge_np2 * 8 +
                   (size_t)W * (2 * m.np2_local + (1u << m.vis_bits)) * 4 + (size_t)W * 2 * idx.degree * 4;
    CUVS_EXPECTS(msmem <= 160 * 1024, "cagra::search: dim too large for the multi-wave LDS layout");
    switch (idx.dtype) {
      case elem_t::f32: launch_search_multi<float>(res, m, nq, msmem); break;
      case elem_t::f16: launch_search_multi<__half>(res, m, nq, msmem); break;
      case elem_t::i8: launch_search_multi<int8_t>(res, m, nq, msmem); break;
      case elem_t::u8: launch_search_multi<uint8_t>(res, m, nq, msmem); break;
    }
    if (idx.source_indices.data() != nullptr)
      hipLaunchKernelGGL(cagra_source_ids_kernel, dim3((unsigned)grid_blocks(nq * k, 256)), dim3(256), 0, res.stream, out_idx, nq * (int64_t)k,
                         idx64 ? 1 : 0, idx.source_indices.data(), idx.n);
    return;
  }
  size_t smem = (size_t)((idx.dim + 3) & ~int64_t(3)) * 4 + (size_t)a.np2 * 8 + ((size_t)4 << bits) +
                (size_t)(2 * (itopk + a.width * idx.degree) + a.width) * 4;  // + seed candidates, parent list
  CUVS_EXPECTS(smem <= 160 * 1024, "cagra::search: dim/itopk too large for LDS");
  switch (idx.dtype) {
    case elem_t::f32: launch_search<float>(res, a, nq, smem); break;
    case elem_t::f16: launch_search<__half>(res, a, nq, smem); break;
    case elem_t::i8: launch_search<int8_t>(res, a, nq, smem); break;
    case elem_t::u8: launch_search<uint8_t>(res, a, nq, smem); break;
  }
  if (idx.source_indices.data() != nullptr)
    hipLaunchKernelGGL(cagra_source_ids_kernel, dim3((unsigned)grid_blocks(nq * k, 256)), dim3(256), 0, res.stream, out_idx, nq * (int64_t)k,
                       idx64 ? 1 : 0, idx.source_indices.data(), idx.n);
}

// ------------------------------------------------------------------ extend (add_nodes.cuh)
// New rows are added chunk by chunk: every new node searches the CURRENT graph (so later chunks can link to
// earlier ones) and takes the `degree` nearest results as its edges; each existing node that was picked gives its
// last (weakest-ranked) edges - at most degree/2 of them, one atomic slot counter per node - to the new nodes that
// picked it, which is how the reference wires the reverse edges (add_nodes.cuh: rev edges replace the tail of the
// list). The dataset and the graph become storage owned by the index.
__global__ void extend_rows_kernel(const uint32_t* __restrict__ nb, int64_t m, uint32_t degree, int64_t n_cur,
                                   uint32_t* __restrict__ graph, uint32_t* __restrict__ rev_cnt)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * (int64_t)degree) return;
  const int64_t v = i / degree;
  const uint32_t j = (uint32_t)(i % degree);
  uint32_t u = nb[i];
  if (u >= (uint32_t)n_cur) u = (uint32_t)((v * 2654435761ull + j) % (uint64_t)n_cur);  // search gave fewer results
  graph[(n_cur + v) * degree + j] = u;
  if (j < degree / 2) {  // the closer half of the new node's neighbours get a reverse edge
    const uint32_t slot = atomicAdd(&rev_cnt[u], 1u);
    if (slot < degree / 2) graph[(int64_t)u * degree + (degree - 1 - slot)] = (uint32_t)(n_cur + v);
  }
}

void cagra_extend(resources& res, cagra_index& idx, const void* new_rows, bool new_is_host, int64_t m, uint32_t max_chunk)
{
  if (m == 0) return;
  CUVS_EXPECTS(idx.data != nullptr && idx.graph.data() != nullptr, "cagra::extend: index has no graph/dataset");
  CUVS_EXPECTS(idx.n + m < (int64_t(1) << 32) - 1, "cagra: at most 2^32 - 2 rows (uint32 graph)");
  const size_t esz = elem_size(idx.dtype), row_bytes = (size_t)idx.dim * esz;
  const int64_t n0 = idx.n, n1 = idx.n + m;
  auto data  = dev_buf<char>::persistent((size_t)n1 * row_bytes);
  auto graph = dev_buf<uint32_t>::persistent((size_t)n1 * idx.degree);
  copy_async(res, data.data(), idx.data, (size_t)n0 * row_bytes);
  copy_async(res, data.data() + (size_t)n0 * row_bytes, new_rows, (size_t)m * row_bytes);
  copy_async(res, graph.data(), idx.graph.data(), (size_t)n0 * idx.degree * sizeof(uint32_t));
  sync(res);
  (void)new_is_host;
  idx.owned = std::move(data);
  idx.data  = idx.owned.data();
  idx.graph = std::move(graph);
  const int64_t chunk = std::max<int64_t>(1, max_chunk == 0 ? 8192 : (int64_t)max_chunk);
  cuvsCagraSearchParams sp{};
  sp.itopk_size   = std::max<size_t>(64, 2 * (size_t)idx.degree);
  sp.search_width = 2;
  sp.algo         = SINGLE_CTA;  // k = degree may exceed what a 32-entry walker list can return
  sp.hashmap_max_fill_rate = 0.5f;
  sp.num_random_samplings  = 1;
  sp.rand_xor_mask         = 0x128394;
  dev_buf<uint32_t> nb(res, (size_t)chunk * idx.degree), rev_cnt(res, n1);
  dev_buf<float> nd(res, (size_t)chunk * idx.degree);
  // reverse-edge slots are counted over the whole call: a later chunk must not overwrite the reverse edges an
  // earlier chunk placed in the same list tail
  HIP_TRY(hipMemsetAsync(rev_cnt.data(), 0, rev_cnt.bytes(), res.stream));
  for (int64_t c0 = 0; c0 < m; c0 += chunk) {
    const int64_t cm = std::min(chunk, m - c0);
    idx.n = n0 + c0;              // the graph searched by this chunk: old rows + the chunks before it
    cagra_set_norms(res, idx);    // cosine: |x| of everything searchable so far
    const int k = (int)std::min<int64_t>(idx.degree, idx.n);
    cagra_search(res, idx, sp, static_cast<const char*>(idx.data) + (size_t)idx.n * row_bytes, cm, k, nb.data(), false,
                 nd.data(), nullptr);
    if (k < (int)idx.degree) CUVS_FAIL("cagra::extend: the index must hold at least graph_degree rows");
    hipLaunchKernelGGL(extend_rows_kernel, dim3(grid_blocks(cm * (int64_t)idx.degree, 256)), dim3(256), 0, res.stream,
                       nb.data(), cm, idx.degree, idx.n, idx.graph.data(), rev_cnt.data());
    HIP_TRY(hipGetLastError());
  }
  idx.n = n1;
  cagra_set_norms(res, idx);
  sync(res);
}

std::unique_ptr<cagra_index> cagra_build(resources& res, const cuvsCagraIndexParams& p, const void* data, elem_t et,
                                         int64_t n, int64_t dim, bool is_host)
{
  const int metric = (int)p.metric;
  CUVS_EXPECTS(metric_is_l2(metric) || metric == M_InnerProduct || metric == M_CosineExpanded,
               "cagra: unsupported metric %d", metric);
  CUVS_EXPECTS(n > 1, "cagra: need at least two rows");
  auto idx    = std::make_unique<cagra_index>();
  idx->metric = metric;
  idx->dtype  = et;
  idx->n      = n;
  idx->dim    = dim;
  const size_t esz = elem_size(et);
  if (is_host) {
    idx->owned = dev_buf<char>::persistent((size_t)n * dim * esz);
    copy_async(res, idx->owned.data(), data, idx->owned.bytes());
    sync(res);
    idx->data = idx->owned.data();
  } else {
    idx->data = data;  // non-owning view of the caller's device dataset (cagra.hpp: update_dataset semantics)
  }
  uint32_t degree = (uint32_t)std::min<int64_t>((int64_t)p.graph_degree, n - 1);
  uint32_t K      = (uint32_t)std::min<int64_t>(std::max<size_t>(p.intermediate_graph_degree, degree), n - 1);
  idx->degree     = degree;
  dev_buf<uint32_t> knn(res, (size_t)n * K);
  const bool small = n <= 200000 || p.build_algo == ITERATIVE_CAGRA_SEARCH;
  if (p.build_algo == NN_DESCENT) {
    cagra_set_norms(res, *idx);  // cosine: the join needs |x|
    knn_graph_nn_descent(res, idx->data, et, n, dim, K, metric, idx->norms.data(), (int)p.nn_descent_niter, knn.data());
  } else if (small) {
    if (et == elem_t::f32) {
      knn_graph_bruteforce<float>(res, static_cast<const float*>(idx->data), n, dim, K, metric, knn.data());
    } else if (et == elem_t::f16) {
      knn_graph_bruteforce<__half>(res, static_cast<const __half*>(idx->data), n, dim, K, metric, knn.data());
    } else {
      dev_buf<float> f(res, (size_t)n * dim);
      if (et == elem_t::i8)
        hipLaunchKernelGGL((to_float_kernel<int8_t>), dim3(grid_blocks(n * dim, 256)), dim3(256), 0, res.stream,
                           static_cast<const int8_t*>(idx->data), n * dim, f.data());
      else
        hipLaunchKernelGGL((to_float_kernel<uint8_t>), dim3(grid_blocks(n * dim, 256)), dim3(256), 0, res.stream,
                           static_cast<const uint8_t*>(idx->data), n * dim, f.data());
      knn_graph_bruteforce<float>(res, f.data(), n, dim, K, metric, knn.data());
      sync(res);
    }
  } else {
    knn_graph_ivf_pq(res, idx->data, et, n, dim, K, metric, knn.data());
  }
  idx->graph = dev_buf<uint32_t>::persistent((size_t)n * degree);
  optimize_graph(res, knn.data(), n, K, degree, idx->graph.data(), res.cagra_guarantee_connectivity);
  cagra_set_norms(res, *idx);
  sync(res);
  return idx;
}

// mg.hip: rows held and metric of the index behind a cuvsCagraIndex handle
void cagra_index_info(uintptr_t addr, int64_t* size, int* metric)
{
  CUVS_EXPECTS(addr != 0, "cagra index is empty");
  auto* idx = reinterpret_cast<const cagra_index*>(addr);
  *size     = idx->n;
  *metric   = idx->metric;
}

}  // namespace cuvs_amd

using namespace cuvs_amd;

namespace {
cagra_index& get_cagra(cuvsCagraIndex_t index)
{
  CUVS_EXPECTS(index != nullptr && index->addr != 0, "CAGRA index is not built");
  return *reinterpret_cast<cagra_index*>(index->addr);
}
}  // namespace

extern "C" {

cuvsError_t cuvsCagraIndexParamsCreate(cuvsCagraIndexParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] {
    *params                       = new cuvsCagraIndexParams{L2Expanded, 128, 64, IVF_PQ, 20, nullptr, nullptr};
    (*params)->graph_build_params = new cuvsIvfPqParams{nullptr, nullptr, 1};
  });
}
cuvsError_t cuvsCagraIndexParamsDestroy(cuvsCagraIndexParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (!params) return;
    if (params->graph_build_params != nullptr) {
      if (params->build_algo == ACE) {
        auto* ace = static_cast<cuvsAceParams*>(params->graph_build_params);
        if (ace->build_dir) free(const_cast<char*>(ace->build_dir));
        delete ace;
      } else if (params->build_algo == IVF_PQ) {
        delete static_cast<cuvsIvfPqParams*>(params->graph_build_params);
      }
    }
    delete params;
  });
}
cuvsError_t cuvsCagraCompressionParamsCreate(cuvsCagraCompressionParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] { *params = new cuvsCagraCompressionParams{8, 0, 0, 25, 0, 0}; });
}
cuvsError_t cuvsCagraCompressionParamsDestroy(cuvsCagraCompressionParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsAceParamsCreate(cuvsAceParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] { *params = new cuvsAceParams{1, 120, nullptr, false, 0, 0}; });
}
cuvsError_t cuvsAceParamsDestroy(cuvsAceParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (params && params->build_dir) free(const_cast<char*>(params->build_dir));
    delete params;
  });
}
cuvsError_t cuvsCagraIndexParamsFromHnswParams(cuvsCagraIndexParams_t params, int64_t, int64_t, int M, int,
                                               enum cuvsCagraHnswHeuristicType, cuvsDistanceType metric)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "params is null");
    params->metric                    = metric;
    params->graph_degree              = (size_t)std::max(2, 2 * M);
    params->intermediate_graph_degree = (size_t)std::max(4, 3 * M);
  });
}
cuvsError_t cuvsCagraExtendParamsCreate(cuvsCagraExtendParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] { *params = new cuvsCagraExtendParams{0}; });
}
cuvsError_t cuvsCagraExtendParamsDestroy(cuvsCagraExtendParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsCagraSearchParamsCreate(cuvsCagraSearchParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto* p                  = new cuvsCagraSearchParams{};
    p->itopk_size            = 64;
    p->search_width          = 1;
    p->algo                  = AUTO;
    p->hashmap_mode          = AUTO_HASH;
    p->hashmap_max_fill_rate = 0.5f;
    p->num_random_samplings  = 1;
    p->rand_xor_mask         = 0x128394;
    p->persistent            = false;
    p->persistent_lifetime   = 2;
    p->persistent_device_usage = 1.0f;
    *params                  = p;
  });
}
cuvsError_t cuvsCagraSearchParamsDestroy(cuvsCagraSearchParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsCagraIndexCreate(cuvsCagraIndex_t* index)
{
  return (cuvsError_t)translate_exceptions([=] { *index = new cuvsCagraIndex{0, DLDataType{0, 0, 0}}; });
}
cuvsError_t cuvsCagraIndexDestroy(cuvsCagraIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (!index) return;
    delete reinterpret_cast<cagra_index*>(index->addr);
    delete index;
  });
}
cuvsError_t cuvsCagraIndexGetDims(cuvsCagraIndex_t index, int64_t* dim)
{
  return (cuvsError_t)translate_exceptions([=] { *dim = get_cagra(index).dim; });
}
cuvsError_t cuvsCagraIndexGetSize(cuvsCagraIndex_t index, int64_t* size)
{
  return (cuvsError_t)translate_exceptions([=] { *size = get_cagra(index).n; });
}
cuvsError_t cuvsCagraIndexGetGraphDegree(cuvsCagraIndex_t index, int64_t* graph_degree)
{
  return (cuvsError_t)translate_exceptions([=] { *graph_degree = get_cagra(index).degree; });
}
cuvsError_t cuvsCagraIndexGetDataset(cuvsCagraIndex_t index, DLManagedTensor* dataset)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_cagra(index);
    fill_dl_view(dataset, const_cast<void*>(idx.data), index->dtype, idx.n, idx.dim, 2, 0);
  });
}
cuvsError_t cuvsCagraIndexGetGraph(cuvsCagraIndex_t index, DLManagedTensor* graph)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_cagra(index);
    fill_dl_view(graph, idx.graph.data(), DLDataType{kDLUInt, 32, 1}, idx.n, idx.degree, 2, 0);
  });
}

cuvsError_t cuvsCagraBuild(cuvsResources_t res_h, cuvsCagraIndexParams_t params, DLManagedTensor* dataset_tensor,
                           cuvsCagraIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(params && dataset_tensor && index, "null argument");
    CUVS_EXPECTS(params->compression == nullptr, "cagra: VPQ compression is outside the hot path (SURVEY 2.1 #17)");
    auto& ds = dataset_tensor->dl_tensor;
    CUVS_EXPECTS(ds.ndim == 2 && is_c_contiguous(ds), "dataset must be a row-major matrix");
    auto idx = cagra_build(res, *params, dl_data(ds), elem_of(ds.dtype), ds.shape[0], ds.shape[1],
                           !is_device_accessible(ds));
    delete reinterpret_cast<cagra_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = ds.dtype;
  });
}

cuvsError_t cuvsCagraIndexFromArgs(cuvsResources_t res_h, cuvsDistanceType metric, DLManagedTensor* graph_tensor,
                                   DLManagedTensor* dataset_tensor, cuvsCagraIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(graph_tensor && dataset_tensor && index, "null argument");
    auto& g  = graph_tensor->dl_tensor;
    auto& ds = dataset_tensor->dl_tensor;
    CUVS_EXPECTS(dtype_is(g.dtype, kDLUInt, 32) && g.ndim == 2 && is_c_contiguous(g), "graph must be uint32 [n, degree]");
    CUVS_EXPECTS(ds.ndim == 2 && is_c_contiguous(ds) && ds.shape[0] == g.shape[0], "dataset/graph shape mismatch");
    auto idx    = std::make_unique<cagra_index>();
    idx->metric = (int)metric;
    idx->dtype  = elem_of(ds.dtype);
    idx->n      = ds.shape[0];
    idx->dim    = ds.shape[1];
    idx->degree = (uint32_t)g.shape[1];
    const size_t esz = elem_size(idx->dtype);
    if (is_device_accessible(ds)) {
      idx->data = dl_data(ds);
    } else {
      idx->owned = dev_buf<char>::persistent((size_t)idx->n * idx->dim * esz);
      copy_async(res, idx->owned.data(), dl_data(ds), idx->owned.bytes());
      idx->data = idx->owned.data();
    }
    idx->graph = dev_buf<uint32_t>::persistent((size_t)idx->n * idx->degree);
    copy_async(res, idx->graph.data(), dl_data(g), idx->graph.bytes());
    cagra_set_norms(res, *idx);
    sync(res);
    delete reinterpret_cast<cagra_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = ds.dtype;
  });
}

cuvsError_t cuvsCagraSearch(cuvsResources_t res_h, cuvsCagraSearchParams_t params, cuvsCagraIndex_t index_c,
                            DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                            DLManagedTensor* distances_tensor, cuvsFilter filter)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_cagra(index_c);
    CUVS_EXPECTS(params && queries_tensor && neighbors_tensor && distances_tensor, "null argument");
    auto& queries   = queries_tensor->dl_tensor;
    auto& neighbors = neighbors_tensor->dl_tensor;
    auto& distances = distances_tensor->dl_tensor;
    CUVS_EXPECTS(is_device_accessible(queries), "queries should have device compatible memory");
    CUVS_EXPECTS(is_device_accessible(neighbors), "neighbors should have device compatible memory");
    CUVS_EXPECTS(is_device_accessible(distances), "distances should have device compatible memory");
    const bool idx64 = dtype_is(neighbors.dtype, kDLInt, 64);
    CUVS_EXPECTS(idx64 || dtype_is(neighbors.dtype, kDLUInt, 32), "neighbors should be of type uint32_t or int64_t");
    CUVS_EXPECTS(dtype_is(distances.dtype, kDLFloat, 32), "distances should be of type float32");
    CUVS_EXPECTS(queries.ndim == 2 && neighbors.ndim == 2 && distances.ndim == 2, "tensors must be 2-D");
    CUVS_EXPECTS(is_c_contiguous(queries) && is_c_contiguous(neighbors) && is_c_contiguous(distances),
                 "tensors must be C-contiguous");
    CUVS_EXPECTS(queries.dtype.code == index_c->dtype.code && queries.dtype.bits == index_c->dtype.bits,
                 "Unsupported queries DLtensor dtype: %d and bits: %d", (int)queries.dtype.code,
                 (int)queries.dtype.bits);
    CUVS_EXPECTS(queries.shape[1] == idx.dim, "queries dim mismatch");
    int64_t m = queries.shape[0], k = neighbors.shape[1];
    CUVS_EXPECTS(neighbors.shape[0] == m && distances.shape[0] == m && distances.shape[1] == k,
                 "neighbors/distances shape mismatch");
    const uint32_t* bits = nullptr;
    if (filter.type != NO_FILTER) {
      CUVS_EXPECTS(filter.type == BITSET && filter.addr != 0, "cagra: only BITSET filters are supported");
      auto& ft = reinterpret_cast<DLManagedTensor*>(filter.addr)->dl_tensor;
      CUVS_EXPECTS(dtype_is(ft.dtype, kDLUInt, 32) && is_device_accessible(ft), "filter must be a device uint32 tensor");
      bits = static_cast<const uint32_t*>(dl_data(ft));
    }
    cagra_search(res, idx, *params, dl_data(queries), m, (int)k, dl_data(neighbors), idx64,
                 static_cast<float*>(dl_data(distances)), bits);
  });
}

namespace {
constexpr int kCagraRefVersion = 5;  // cagra_serialize.cuh:30
// cudaDataType_t codes the reference stores in front of a strided dataset (dataset_serialize.hpp:82-101)
inline uint32_t cuda_dtype_code(elem_t e) { return e == elem_t::f32 ? 0u : e == elem_t::f16 ? 2u : e == elem_t::i8 ? 3u : 8u; }
inline char npy_kind(elem_t e) { return e == elem_t::f32 ? 'f' : e == elem_t::f16 ? 'e' : e == elem_t::i8 ? 'i' : 'u'; }
}  // namespace

cuvsError_t cuvsCagraSerialize(cuvsResources_t res_h, const char* filename, cuvsCagraIndex_t index, bool include_dataset)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_cagra(index);
    if (write_native_container(res)) {
      file_writer w(filename, KIND_CAGRA);
      w.scalar<int32_t>(idx.metric); w.scalar<int32_t>((int)idx.dtype); w.scalar<int64_t>(idx.n); w.scalar<int64_t>(idx.dim);
      w.scalar<uint32_t>(idx.degree); w.scalar<uint8_t>(index->dtype.code); w.scalar<uint8_t>(index->dtype.bits);
      w.scalar<uint8_t>(include_dataset ? 1 : 0);
      w.device_array(res, idx.graph.data(), idx.graph.bytes());
      if (include_dataset) w.device_array(res, idx.data, (size_t)idx.n * idx.dim * elem_size(idx.dtype));
      return;
    }
    // reference record sequence (cagra_serialize.cuh:49-75, dataset_serialize.hpp:38-60,82-87): dtype prefix,
    // version, size (IdxT = uint32), dim, graph_degree, metric, graph [n, degree], content_map (bit 0 = dataset,
    // bit 1 = source_indices), then the dataset: tag 2 (strided), cudaDataType, n_rows (int64), dim, stride, rows
    npy_writer w(filename);
    char prefix[4];
    elem_prefix(idx.dtype, prefix);
    w.raw(prefix, 4);
    w.scalar<int32_t>(kCagraRefVersion);
    w.scalar<uint32_t>((uint32_t)idx.n);
    w.scalar<uint32_t>((uint32_t)idx.dim);
    w.scalar<uint32_t>(idx.degree);
    w.scalar<int32_t>(idx.metric);
    w.device_array(res, 'u', 4, {idx.n, idx.degree}, idx.graph.data());
    const bool with_data = include_dataset && idx.data != nullptr && idx.n > 0;
    const bool with_src = idx.source_indices.data() != nullptr;
    w.scalar<uint32_t>((with_data ? 1u : 0u) | (with_src ? 2u : 0u));
    if (with_data) {
      const size_t es = elem_size(idx.dtype);
      w.scalar<uint32_t>(2u);
      w.scalar<uint32_t>(cuda_dtype_code(idx.dtype));
      w.scalar<int64_t>(idx.n);
      w.scalar<uint32_t>((uint32_t)idx.dim);
      w.scalar<uint32_t>((uint32_t)(round_up(idx.dim * (int64_t)es, 16) / es));  // 16-byte aligned row stride
      w.device_array(res, npy_kind(idx.dtype), (uint32_t)es, {idx.n, idx.dim}, idx.data);
    }
    if (with_src) w.device_array(res, 'u', 4, {idx.n}, idx.source_indices.data());  // cagra_serialize.cuh:83
    w.close();
  });
}

cuvsError_t cuvsCagraDeserialize(cuvsResources_t res_h, const char* filename, cuvsCagraIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr, "index is null");
    auto idx = std::make_unique<cagra_index>();
    DLDataType dl;
    if (is_native_container(filename)) {
      file_reader r(filename, KIND_CAGRA);
      idx->metric = r.scalar<int32_t>(); idx->dtype = (elem_t)r.scalar<int32_t>(); idx->n = r.scalar<int64_t>();
      idx->dim = r.scalar<int64_t>(); idx->degree = r.scalar<uint32_t>();
      uint8_t code = r.scalar<uint8_t>(), bits = r.scalar<uint8_t>(), has_data = r.scalar<uint8_t>();
      idx->graph = r.device_array<uint32_t>(res);
      if (has_data) {
        idx->owned = r.device_array<char>(res);
        idx->data  = idx->owned.data();
      }
      dl = DLDataType{code, bits, 1};
    } else {
      // reference format (cagra_serialize.cuh:270-330; dtype dispatch c/src/neighbors/cagra.cpp:876-900)
      npy_reader r(filename);
      char prefix[4];
      r.raw(prefix, 4);
      CUVS_EXPECTS(parse_elem_prefix(prefix, &idx->dtype), "Unsupported index dtype in file %s", filename);
      int ver = r.scalar<int32_t>();
      CUVS_EXPECTS(ver == kCagraRefVersion, "serialization version mismatch, expected %d, got %d ", kCagraRefVersion, ver);
      idx->n      = (int64_t)r.scalar<uint32_t>();
      idx->dim    = (int64_t)r.scalar<uint32_t>();
      idx->degree = r.scalar<uint32_t>();
      idx->metric = r.scalar<int32_t>();
      CUVS_EXPECTS(metric_is_l2(idx->metric) || idx->metric == M_InnerProduct || idx->metric == M_CosineExpanded,
                   "cagra::deserialize: unsupported metric value %d", idx->metric);
      CUVS_EXPECTS(idx->degree > 0 && idx->degree <= 1024, "cagra::deserialize: graph_degree=%u exceeds maximum %u",
                   idx->degree, 1024u);
      idx->graph = r.device_array<uint32_t>(res, idx->n * (int64_t)idx->degree);
      uint32_t content = r.scalar<uint32_t>();
      if (content & 1u) {
        uint32_t tag = r.scalar<uint32_t>();
        if (tag == 1u) {
          (void)r.scalar<uint32_t>();  // empty dataset: suggested dim only
        } else {
          CUVS_EXPECTS(tag == 2u, "Failed to deserialize dataset: instance tag %u (VPQ-compressed datasets are not built)", tag);
          uint32_t code = r.scalar<uint32_t>();
          CUVS_EXPECTS(code == cuda_dtype_code(idx->dtype), "cagra::deserialize: dataset element type %u does not match the index dtype", code);
          int64_t rows = r.scalar<int64_t>();
          uint32_t dim = r.scalar<uint32_t>();
          (void)r.scalar<uint32_t>();  // stride: rows are stored unpadded
          CUVS_EXPECTS(rows == idx->n && dim == idx->dim, "cagra::deserialize: dataset shape does not match the graph");
          idx->owned = r.device_bytes(res, (uint32_t)elem_size(idx->dtype), rows * (int64_t)dim);
          idx->data  = idx->owned.data();
        }
      }
      if (content & 2u) idx->source_indices = r.device_array<uint32_t>(res, idx->n);  // cagra_serialize.cuh:314-321
      dl = dl_of(idx->dtype);
    }
    cagra_set_norms(res, *idx);
    sync(res);
    delete reinterpret_cast<cagra_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = dl;
  });
}

// hnswlib "base layer only" export (cagra_serialize.cuh:98-258): 96-byte header, then per node
// {int degree, uint32 links[degree], T row[dim], size_t label}, then one zero int per node (no upper levels).
cuvsError_t cuvsCagraSerializeToHnswlib(cuvsResources_t res_h, const char* filename, cuvsCagraIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_cagra(index);
    CUVS_EXPECTS(idx.data != nullptr && idx.n > 0, "Invalid CAGRA dataset of size 0 during serialization");
    CUVS_EXPECTS(filename != nullptr, "filename is null");
    FILE* f = fopen(filename, "wb");
    CUVS_EXPECTS(f != nullptr, "Cannot open file %s", filename);
    struct closer { FILE* f; ~closer() { if (f) fclose(f); } } guard{f};
    auto put = [&](const void* p, size_t n) { CUVS_EXPECTS(fwrite(p, 1, n, f) == n, "Error writing HNSW file"); };
    const size_t es = elem_size(idx.dtype), n = (size_t)idx.n, dim = (size_t)idx.dim, deg = idx.degree;
    const size_t per_elem = deg * 4 + 4 + dim * es + 8;
    const size_t hdr[6]   = {0, n, n, per_elem, per_elem - 8, deg * 4 + 4};
    put(hdr, sizeof(hdr));            // offset_level_0, max_element, curr_element_count, size_data_per_element,
                                      // label_offset, offset_data
    const int max_level = 1, entry = (int)(n / 2);
    put(&max_level, 4);
    put(&entry, 4);
    const size_t m[3] = {deg / 2, deg, deg / 2};  // max_M, max_M0, M
    put(m, sizeof(m));
    const double mult = 0.42424242;
    put(&mult, 8);
    const size_t ef_construction = 500;
    put(&ef_construction, 8);
    const size_t step = std::max<size_t>(1, (size_t(64) << 20) / (dim * es + deg * 4));
    std::vector<uint32_t> g(step * deg);
    std::vector<char> rows(step * dim * es), out(step * per_elem);
    for (size_t r0 = 0; r0 < n; r0 += step) {
      const size_t nr = std::min(step, n - r0);
      copy_async(res, g.data(), idx.graph.data() + r0 * deg, nr * deg * 4);
      copy_async(res, rows.data(), static_cast<const char*>(idx.data) + r0 * dim * es, nr * dim * es);
      sync(res);
      char* o = out.data();
      for (size_t i = 0; i < nr; ++i) {
        const int d32 = (int)deg;
        const size_t label = r0 + i;
        memcpy(o, &d32, 4); o += 4;
        memcpy(o, g.data() + i * deg, deg * 4); o += deg * 4;
        memcpy(o, rows.data() + i * dim * es, dim * es); o += dim * es;
        memcpy(o, &label, 8); o += 8;
      }
      put(out.data(), nr * per_elem);
    }
    std::vector<int> zeros(std::min<size_t>(n, 1 << 20), 0);
    for (size_t r0 = 0; r0 < n; r0 += zeros.size()) put(zeros.data(), std::min(zeros.size(), n - r0) * 4);
    int rc  = fclose(f);
    guard.f = nullptr;
    CUVS_EXPECTS(rc == 0, "Error writing output %s", filename);
  });
}

cuvsError_t cuvsCagraExtend(cuvsResources_t res_h, cuvsCagraExtendParams_t params, DLManagedTensor* additional_dataset,
                            cuvsCagraIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_cagra(index);
    CUVS_EXPECTS(additional_dataset != nullptr, "additional_dataset is null");
    auto& t = additional_dataset->dl_tensor;
    CUVS_EXPECTS(t.ndim == 2 && is_c_contiguous(t) && t.shape[1] == idx.dim, "additional_dataset must be [m, dim] row-major");
    CUVS_EXPECTS(elem_of(t.dtype) == idx.dtype, "additional_dataset dtype differs from the index dtype");
    cagra_extend(res, idx, dl_data(t), !is_device_accessible(t), t.shape[0], params ? params->max_chunk_size : 0u);
  });
}

// Physical merge (cagra_merge.cuh): the datasets of the input indexes are concatenated in order and a new graph is
// built over all rows with `params`; ids of index i are shifted by the sizes of the indexes before it.
namespace {
// rows[keep[i]] -> out[i] (row_bytes is a multiple of the element size; byte copy, one workgroup per row, strided over the rows)
__global__ void gather_rows_kernel(const char* __restrict__ rows, const int64_t* __restrict__ keep, int64_t n_keep, size_t row_bytes,
                                   char* __restrict__ out)
{
  for (int64_t i = blockIdx.x; i < n_keep; i += gridDim.x) {
    const char* src = rows + (size_t)keep[i] * row_bytes;
    char* dst       = out + (size_t)i * row_bytes;
    for (size_t b = threadIdx.x; b < row_bytes; b += blockDim.x) dst[b] = src[b];
  }
}
}  // namespace

cuvsError_t cuvsCagraMerge(cuvsResources_t res_h, cuvsCagraIndexParams_t params, cuvsCagraIndex_t* indices,
                           size_t num_indices, cuvsFilter filter, cuvsCagraIndex_t output_index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(params && indices && output_index && num_indices > 0, "null argument");
    // a BITSET over the concatenated rows keeps the rows whose bit is set (cagra_merge.cuh:94-131: the merged index is built on
    // the kept rows, in order); bitmaps are refused as in the reference (:45-46)
    CUVS_EXPECTS(filter.type == NO_FILTER || filter.type == BITSET, "Bitmap filter isn't supported inside cagra::merge");
    const uint32_t* bits = nullptr;
    if (filter.type == BITSET) {
      CUVS_EXPECTS(filter.addr != 0, "cagra::merge: the filter has no tensor");
      auto& ft = reinterpret_cast<DLManagedTensor*>(filter.addr)->dl_tensor;
      CUVS_EXPECTS(dtype_is(ft.dtype, kDLUInt, 32) && is_device_accessible(ft), "filter must be a device uint32 tensor");
      bits = static_cast<const uint32_t*>(dl_data(ft));
    }
    auto& first = get_cagra(indices[0]);
    int64_t total = 0;
    for (size_t i = 0; i < num_indices; ++i) {
      auto& ix = get_cagra(indices[i]);
      CUVS_EXPECTS(ix.data != nullptr, "cagra::merge: index %zu has no dataset", i);
      CUVS_EXPECTS(ix.dim == first.dim && ix.dtype == first.dtype, "cagra::merge: indexes differ in dim or dtype");
      total += ix.n;
    }
    const size_t row_bytes = (size_t)first.dim * elem_size(first.dtype);
    auto all = dev_buf<char>::persistent((size_t)total * row_bytes);
    size_t off = 0;
    for (size_t i = 0; i < num_indices; ++i) {
      auto& ix = get_cagra(indices[i]);
      copy_async(res, all.data() + off, ix.data, (size_t)ix.n * row_bytes);
      off += (size_t)ix.n * row_bytes;
    }
    sync(res);
    if (bits != nullptr) {  // the kept rows, compacted in order
      const size_t words = ((size_t)total + 31) / 32;
      std::vector<uint32_t> hb = to_host(res, bits, words);
      std::vector<int64_t> keep;
      keep.reserve((size_t)total);
      for (int64_t r = 0; r < total; ++r)
        if ((hb[(size_t)r >> 5] >> (r & 31)) & 1u) keep.push_back(r);
      CUVS_EXPECTS(!keep.empty(), "cagra::merge: the filter keeps no row");
      dev_buf<int64_t> kd(res, keep.size());
      copy_async(res, kd.data(), keep.data(), keep.size() * sizeof(int64_t));
      auto kept = dev_buf<char>::persistent(keep.size() * row_bytes);
      hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<int64_t>((int64_t)keep.size(), 65535)), dim3(256), 0, res.stream,
                         all.data(), kd.data(), (int64_t)keep.size(), row_bytes, kept.data());
      sync(res);  // (keep is pageable host memory)
      all   = std::move(kept);
      total = (int64_t)keep.size();
    }
    auto idx   = cagra_build(res, *params, all.data(), first.dtype, total, first.dim, false);
    idx->owned = std::move(all);  // the merged index owns the concatenated rows (idx->data already points at them)
    delete reinterpret_cast<cagra_index*>(output_index->addr);
    output_index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    output_index->dtype = indices[0]->dtype;
  });
}

}  // extern "C"


// Measured work of the graph walks run on this handle (CAGRA bytes(q) = n_dist * dim * sizeof(T) + n_iter *
// degree * 4 with n_dist and n_iter measured; the reference has its per-phase _CLK_BREAKDOWN counters,
// search_single_cta_jit.cuh:91-103,425-451). enable != 0: zero the counters and count from now on; enable == 0: stop and
// return {rows scored, graph rows read, walkers} of the searches since (include/cuvs_amd/extensions.h).
extern "C" __attribute__((visibility("default"))) cuvsError_t cuvsAmdCagraWorkCounters(cuvsResources_t res_h, int enable,
                                                                                       uint64_t out[3])
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *cuvs_amd::as_res(res_h);
    if (enable) {
      if (res.cagra_work == nullptr) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&res.cagra_work), 3 * sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(res.cagra_work, 0, 3 * sizeof(unsigned long long), res.stream));
      return;
    }
    CUVS_EXPECTS(res.cagra_work != nullptr && out != nullptr, "cagra work counters were not enabled on this handle");
    unsigned long long h[3];
    HIP_TRY(hipMemcpyAsync(h, res.cagra_work, sizeof(h), hipMemcpyDeviceToHost, res.stream));
    HIP_TRY(hipStreamSynchronize(res.stream));
    for (int i = 0; i < 3; ++i) out[i] = h[i];
    (void)hipFree(res.cagra_work);
    res.cagra_work = nullptr;
  });
}

// cagra::index_params::guarantee_connectivity (cpp/include/cuvs/neighbors/cagra.hpp:193) has no field in the C struct
// cuvsCagraIndexParams (c/include/cuvs/neighbors/cagra.h): the switch lives on the handle and applies to the
// cuvsCagraBuild / cuvsCagraExtend calls made with it (include/cuvs_amd/extensions.h).
extern "C" __attribute__((visibility("default"))) cuvsError_t cuvsAmdCagraSetGuaranteeConnectivity(cuvsResources_t res_h, int on)
{
  return (cuvsError_t)translate_exceptions([=] { cuvs_amd::as_res(res_h)->cagra_guarantee_connectivity = on != 0; });
}

// cagra::helpers::optimize (cpp/include/cuvs/neighbors/cagra_optimize.hpp; graph_core.cuh:1706-1809): kNN graph
// [n, K] uint32 (host or device) -> search graph [n, degree] uint32 (host or device); components_left (optional):
// connected components of the spanning forest (1 when guarantee_connectivity succeeded, 0 when it was not asked for).
extern "C" __attribute__((visibility("default"))) cuvsError_t cuvsAmdCagraOptimize(cuvsResources_t res_h,
                                                                                   DLManagedTensor* knn_tensor,
                                                                                   DLManagedTensor* graph_tensor,
                                                                                   int guarantee_connectivity)
{
  return (cuvsError_t)translate_exceptions([=] {
    using namespace cuvs_amd;
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(knn_tensor && graph_tensor, "null argument");
    auto& kg = knn_tensor->dl_tensor;
    auto& g  = graph_tensor->dl_tensor;
    CUVS_EXPECTS(dtype_is(kg.dtype, kDLUInt, 32) && kg.ndim == 2 && is_c_contiguous(kg), "knn_graph must be uint32 [n, K]");
    CUVS_EXPECTS(dtype_is(g.dtype, kDLUInt, 32) && g.ndim == 2 && is_c_contiguous(g) && g.shape[0] == kg.shape[0],
                 "graph must be uint32 [n, degree]");
    const int64_t n = kg.shape[0];
    const uint32_t K = (uint32_t)kg.shape[1], degree = (uint32_t)g.shape[1];
    dev_buf<uint32_t> knn_d, g_d;
    const uint32_t* knn = static_cast<const uint32_t*>(dl_data(kg));
    if (!is_device_accessible(kg)) {
      knn_d = dev_buf<uint32_t>(res, (size_t)n * K);
      copy_async(res, knn_d.data(), knn, knn_d.bytes());
      knn = knn_d.data();
    }
    uint32_t* out = static_cast<uint32_t*>(dl_data(g));
    if (!is_device_accessible(g)) {
      g_d = dev_buf<uint32_t>(res, (size_t)n * degree);
      out = g_d.data();
    }
    optimize_graph(res, knn, n, K, degree, out, guarantee_connectivity != 0);
    if (!is_device_accessible(g)) copy_async(res, dl_data(g), out, (size_t)n * degree * sizeof(uint32_t));
    sync(res);
  });
}

// Test hook (not part of the reference ABI): the search plan for given parameters, computed on the host without
// touching a GPU - tests pin it to the reference's rules (search_plan.cuh:121-131,199-340).
extern "C" __attribute__((visibility("default"))) int cuvsAmdCagraSearchPlan(cuvsCagraSearchParams_t params, int64_t n_rows,
                                                                              uint32_t graph_degree, uint32_t topk,
                                                                              int64_t n_queries, int num_cus,
                                                                              float filtering_rate, uint32_t out[10])
{
  return translate_exceptions([=] {
    const cuvs_amd::cagra_plan pl =
      cuvs_amd::make_cagra_plan(*params, n_rows, graph_degree, topk, n_queries, num_cus, filtering_rate);
    out[0] = pl.itopk; out[1] = pl.max_iterations; out[2] = pl.search_width; out[3] = (uint32_t)pl.ref_algo;
    out[4] = pl.ref_small_hash_bitlen; out[5] = pl.ref_hash_bitlen; out[6] = pl.ref_small_hash_reset_interval;
    out[7] = pl.ref_mc_num_cta_per_query; out[8] = pl.mc_max_iterations; out[9] = 0;
  });
}
