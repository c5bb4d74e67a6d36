// Pieces shared by the list-major IVF scans (ivf_pq_search.hip, ivf_flat.hip): work-item construction from the
// (query, probe) pairs grouped by list, and the register-resident per-wave top-k list.
#pragma once
#include "ops.hpp"
#include "device_utils.hpp"

#include <algorithm>
#include <functional>
#include <vector>

namespace cuvs_amd {
namespace {

struct work_item {
  uint32_t list;
  uint32_t first;  // first position in the list-sorted pair array
  uint32_t count;  // 1..QPB pairs
  uint32_t pad;
};

// per list: number of work items = ceil(cnt / qpb); item_off = exclusive scan (single workgroup)
// (labels >= split are cut into items of qpb_hi pairs: the two phases of a search may use different kernels)
__global__ __launch_bounds__(1024) void count_items_kernel(const uint32_t* __restrict__ pair_off, int n_lists,
                                                           int qpb_lo, uint32_t* __restrict__ item_off, int split,
                                                           int qpb_hi)
{
  // (a thread takes a run of consecutive labels, one block scan of the run totals: see pair_scan_kernel)
  __shared__ int smem[17];
  const int per = (n_lists + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(n_lists, b + per);
  auto items_of = [&](int i) {
    const int qpb = i < split ? qpb_lo : qpb_hi;
    return ((int)(pair_off[i + 1] - pair_off[i]) + qpb - 1) / qpb;
  };
  int s = 0;
  for (int i = b; i < e; ++i) s += items_of(i);
  int total;
  int run = block_exclusive_scan(s, smem, &total);
  for (int i = b; i < e; ++i) {
    item_off[i] = (uint32_t)run;
    run += items_of(i);
  }
  if (threadIdx.x == 0) item_off[n_lists] = (uint32_t)total;
}

__global__ void fill_items_kernel(const uint32_t* __restrict__ pair_off, const uint32_t* __restrict__ item_off,
                                  int n_lists, int qpb_lo, work_item* __restrict__ items, int split, int qpb_hi)
{
  int L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= n_lists) return;
  const int qpb = L < split ? qpb_lo : qpb_hi;
  uint32_t b = pair_off[L], e = pair_off[L + 1];
  uint32_t w = item_off[L];
  for (uint32_t p = b; p < e; p += qpb, ++w) {
    work_item it;
    it.list  = (uint32_t)L;
    it.first = p;
    it.count = min((uint32_t)qpb, e - p);
    it.pad   = 0;
    items[w] = it;
  }
}

// Two-phase schedule of the list scans: the `head` nearest probes of every query keep their list id as label, the
// others move to n_lists + list, so that one grouping yields [head items | tail items]. The head phase runs first
// and leaves a per-query k-th bound that is already close to final - which is what makes the early stop of the
// tail phase bite. Results do not depend on the order in which pairs are scanned.
// List-sharded search (shard_world > 1): pairs that probe a list of another rank get the label `skip` - a bucket
// past the scanned ranges, so that no work item is ever run for them.
// head_in_tail (partial head): the head pairs get their list's TAIL label as well - the head phase takes its items straight
// from the probes (head_items_kernel) and scores only the first rows of those lists; all their rows are screened with the others.
__global__ void phase_labels_kernel(const uint32_t* __restrict__ probes, int64_t n_pairs, uint32_t n_probes,
                                    uint32_t head, uint32_t n_lists, uint32_t* __restrict__ out,
                                    uint32_t shard_world = 1, uint32_t shard_rank = 0, uint32_t skip = 0,
                                    const int32_t* __restrict__ owner = nullptr, bool head_in_tail = false)
{
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t L = probes[p];
    const bool mine  = shard_world <= 1 || (owner != nullptr ? (uint32_t)owner[L] == shard_rank : L % shard_world == shard_rank);
    out[p] = !mine ? skip : L + ((((uint32_t)(p % n_probes) < head && !head_in_tail) || head == 0) ? 0u : n_lists);
  }
}

// ------------------------------------------------------------------ register-resident sorted top list (one wave)
// rank r lives in lane r % 64, slot r / 64; sorted ascending by (distance, row).
template <int E>
struct wave_top {
  float d[E];
  uint32_t i[E];
  __device__ inline void init()
  {
#pragma unroll
    for (int e = 0; e < E; ++e) { d[e] = INFINITY; i[e] = 0xffffffffu; }
  }
  // value at rank r (wave-uniform r)
  __device__ inline float rank_d(int r) const
  {
    float v = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if ((r >> 6) == e) v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d[e]), r & 63));
    return v;
  }
  __device__ inline uint32_t rank_i(int r) const
  {
    uint32_t v = 0;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if ((r >> 6) == e) v = __builtin_amdgcn_readlane(i[e], r & 63);
    return v;
  }
  // insert wave-uniform candidate (cd, ci); ranks beyond 64*E fall off.
  // The one-lane shift is a DPP wave_shr:1 move (lane 0 takes the carry from the previous slot).
  __device__ inline void insert(float cd, uint32_t ci, int lane)
  {
    int pos = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      bool le = (d[e] < cd) || (d[e] == cd && i[e] <= ci);
      pos += __popcll(__ballot(le));
    }
    uint32_t carry_d = 0, carry_i = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t du   = __float_as_uint(d[e]);
      const uint32_t up_d = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_d, (int)du, 0x138, 0xf, 0xf, false);
      const uint32_t up_i = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_i, (int)i[e], 0x138, 0xf, 0xf, false);
      carry_d = __builtin_amdgcn_readlane(du, 63);
      carry_i = __builtin_amdgcn_readlane(i[e], 63);
      const int rank = e * 64 + lane;
      if (rank > pos) { d[e] = __uint_as_float(up_d); i[e] = up_i; }
      else if (rank == pos) { d[e] = cd; i[e] = ci; }
    }
  }
};


// 64 (distance, row) pairs, one per lane, into ascending (distance, row) order: lane r ends up with rank r - a bitonic network
// of 21 compare-exchange steps. What a scan kernel's wave does with the candidates of its FIRST tile instead of 64 serial
// insertions into an empty list (no NaNs among the distances: the caller checks).
__device__ inline void wave_sort64(float& d, uint32_t& i, const int lane)
{
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      const float od    = __shfl_xor(d, j2, 64);
      const uint32_t oi = (uint32_t)__shfl_xor((int)i, j2, 64);
      const bool other_less = (od < d) || (od == d && oi < i);
      const bool same       = od == d && oi == i;
      const bool take_min   = ((lane & j2) == 0) == ((lane & k2) == 0);  // (k2 == 64: every block ascends)
      if (take_min ? other_less : (!other_less && !same)) { d = od; i = oi; }
    }
  }
}

// ------------------------------------------------------------------ non-fused path: k beyond the register top lists
// The reference falls back from its fused top-k to "write every score, then select_k" when k exceeds the warp-sort
// capacity (is_local_topk_feasible, ivf_pq_compute_similarity_impl.cuh:39-45; ivf_pq_search.cuh:620,
// ivf_flat_search.cuh:180,283). Same here: the scan kernels store the score of every probed row into a
// [n_queries, ld] matrix - the rows of probe p of query q at columns seg[q * n_probes + p] .. (calc_chunk_indices,
// ivf_common.cu:23-51), with the flat row of every column in a second matrix - and select_k picks the k best of every
// row: which of several equal k-th scores survive is decided by column (probe rank, then in-list position), the
// winners come out ordered by (score, flat row) - exactly what the fused path's per-pair lists + merge produce.
// ld = the n_probes largest lists together (the reference's accum_sorted_sizes(n_probes)): no device -> host round trip.
__global__ void pair_segments_kernel(const uint32_t* __restrict__ probes, const uint32_t* __restrict__ list_sizes,
                                     int64_t nq, uint32_t n_probes, uint32_t* __restrict__ seg)
{
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t off = 0;
  for (uint32_t p = 0; p < n_probes; ++p) {
    seg[q * n_probes + p] = off;
    off += list_sizes[probes[q * n_probes + p]];
  }
}

inline size_t largest_lists_total(const std::vector<uint32_t>& sizes, uint32_t n_probes)
{
  std::vector<uint32_t> s(sizes);
  const size_t m = std::min<size_t>(n_probes, s.size());
  std::partial_sort(s.begin(), s.begin() + m, s.end(), std::greater<uint32_t>());
  size_t t = 0;
  for (size_t i = 0; i < m; ++i) t += s[i];
  return t;
}

// ------------------------------------------------------------------ (query, probe) pairs grouped by label, no radix sort
// group_by_label (kmeans_balanced.hip) sorts with hipcub's one-sweep radix sort, whose workgroups wait for their predecessors'
// prefixes (decoupled look-back): next to a long kernel of another stream those workgroups trickle in one by one and the ones
// already resident spin - 0.87 ms for a 0.07 ms sort in the two-stream schedule of the IVF-PQ batch (profiles/r05_*). This
// grouping has no dependency between workgroups: histogram, exclusive scan, scatter by atomic cursor (arbitrary order inside a
// label), then every label's segment is put into ascending pair order by rank counting - pair ids are unique, so the result is
// the stable order of the radix sort, bit for bit, whatever the atomics did.
constexpr int kSegCap = 2048;  // pairs of a label sorted from one wave's LDS region; longer segments: sort_big_segments_kernel

// labels >= n_plain (the list shard's bucket of foreign pairs: up to (world - 1) / world of ALL pairs under ONE label) are
// counted per workgroup and added once - a million atomics on one address run one after the other (13 ms at two ranks)
__global__ __launch_bounds__(256) void pair_histogram_kernel(const uint32_t* __restrict__ labels, int64_t n, uint32_t* __restrict__ counts,
                                                             uint32_t n_plain, uint32_t n_labels)
{
  __shared__ uint32_t bulk;
  if (threadIdx.x == 0) bulk = 0u;
  __syncthreads();
  uint32_t mine = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t L = labels[i];
    if (L < n_plain) atomicAdd(&counts[L], 1u);
    else             ++mine;
  }
  if (n_plain >= n_labels) return;  // (uniform: no bulk label)
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine != 0u) atomicAdd(&bulk, mine);
  __syncthreads();
  if (threadIdx.x == 0 && bulk != 0u) atomicAdd(&counts[n_plain], bulk);  // (one bulk label: n_labels == n_plain + 1)
}
// exclusive scan by ONE workgroup: a thread takes a run of consecutive elements (its own running sum), one block scan of the
// 1024 run totals - two passes over the array instead of n / 1024 block scans with three barriers each
__global__ __launch_bounds__(1024) void pair_scan_kernel(const uint32_t* __restrict__ counts, int n, uint32_t* __restrict__ offsets)
{
  __shared__ int smem[17];
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(n, b + per);
  int s = 0;
  for (int i = b; i < e; ++i) s += (int)counts[i];
  int total;
  int run = block_exclusive_scan(s, smem, &total);
  for (int i = b; i < e; ++i) {
    const int c = (int)counts[i];
    offsets[i] = (uint32_t)run;
    run += c;
  }
  if (threadIdx.x == 0) offsets[n] = (uint32_t)total;
}
// (pairs of a label >= n_plain are not placed: nothing reads the segment of the foreign pairs)
__global__ void pair_scatter_kernel(const uint32_t* __restrict__ labels, int64_t n, const uint32_t* __restrict__ offsets,
                                    uint32_t* __restrict__ cursor, uint32_t* __restrict__ out, uint32_t n_plain)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t L = labels[i];
  if (L >= n_plain) return;
  out[offsets[L] + atomicAdd(&cursor[L], 1u)] = (uint32_t)i;
}
// one wave per label: the segment into LDS, every element to the position of its rank (broadcast reads: all lanes read the
// same word)
__global__ __launch_bounds__(256) void sort_segments_kernel(const uint32_t* __restrict__ offsets, uint32_t n_labels,
                                                            uint32_t* __restrict__ pairs)
{
  __shared__ uint32_t seg_all[4][kSegCap];
  const uint32_t L = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (L >= n_labels) return;  // wave-uniform
  const int lane   = threadIdx.x & 63;
  const uint32_t b = offsets[L], n = offsets[L + 1] - b;
  if (n <= 1u || n > (uint32_t)kSegCap) return;
  uint32_t* seg = seg_all[threadIdx.x >> 6];
  for (uint32_t i = lane; i < n; i += 64u) seg[i] = pairs[b + i];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = lane; i < n; i += 64u) {
    const uint32_t v = seg[i];
    uint32_t rank = 0u;
    for (uint32_t j = 0; j < n; ++j) rank += seg[j] < v ? 1u : 0u;
    pairs[b + rank] = v;
  }
}
// segments beyond kSegCap (a list probed by thousands of queries of the batch): one 1024-thread workgroup per such label. A
// label holds at most ONE pair per query (a query probes a list once), and pair order = query order, so the position of a
// pair is the number of the label's queries before its own: a bitmap over the batch's queries in LDS, a prefix sum over its
// words - O(segment + queries / 32) per label. Batches beyond kBitmapQueries queries: rank counting over LDS tiles (quadratic
// in the segment: correct, slow). Through a second buffer (the segment is read while positions are still being computed).
constexpr uint32_t kBitmapQueries = 1u << 19;  // 2 x 64 KiB of LDS: bitmap + word prefixes

__global__ __launch_bounds__(1024) void sort_big_segments_kernel(const uint32_t* __restrict__ offsets, uint32_t n_labels,
                                                                 uint32_t* __restrict__ pairs, uint32_t* __restrict__ tmp,
                                                                 uint32_t n_probes, uint32_t n_queries)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t sb_smem[];  // bitmap [words] | prefix [words]   or   tile [4096]
  __shared__ uint32_t big[1024];
  __shared__ uint32_t n_big;
  __shared__ int scan_smem[17];
  const bool bitmap_ok = n_probes != 0u && n_queries <= kBitmapQueries;
  const uint32_t words = (n_queries + 31u) / 32u;
  // the workgroup's share of the labels, checked by all threads at once (as a rule there is no long segment at all)
  const uint32_t per = (n_labels + gridDim.x - 1) / gridDim.x;
  const uint32_t l0 = blockIdx.x * per, l1 = min(n_labels, l0 + per);
  for (uint32_t base = l0; base < l1; base += 1024u) {
    if (threadIdx.x == 0) n_big = 0u;
    __syncthreads();
    const uint32_t Lc = base + threadIdx.x;
    if (Lc < l1 && offsets[Lc + 1] - offsets[Lc] > (uint32_t)kSegCap) big[atomicAdd(&n_big, 1u)] = Lc;
    __syncthreads();
    const uint32_t nb = n_big;
    for (uint32_t w = 0; w < nb; ++w) {
      // (the order in which a workgroup takes its long segments does not matter: each is sorted on its own)
      const uint32_t L = big[w];
      const uint32_t b = offsets[L], n = offsets[L + 1] - b;
      if (bitmap_ok) {
        uint32_t* bits = sb_smem;
        uint32_t* pref = sb_smem + words;
        for (uint32_t i = threadIdx.x; i < words; i += 1024u) bits[i] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
          const uint32_t q = pairs[b + i] / n_probes;
          atomicOr(&bits[q >> 5], 1u << (q & 31u));
        }
        __syncthreads();
        // exclusive prefix over the words' popcounts: a run of words per thread, one block scan of the run totals
        const uint32_t run = (words + 1023u) / 1024u, w0 = threadIdx.x * run, w1 = min(words, w0 + run);
        int sum = 0;
        for (uint32_t i = w0; i < w1; ++i) sum += __popc(bits[i]);
        int total;
        int acc = block_exclusive_scan(sum, scan_smem, &total);
        for (uint32_t i = w0; i < w1; ++i) { pref[i] = (uint32_t)acc; acc += __popc(bits[i]); }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
          const uint32_t p = pairs[b + i], q = p / n_probes;
          tmp[b + pref[q >> 5] + (uint32_t)__popc(bits[q >> 5] & ((1u << (q & 31u)) - 1u))] = p;
        }
      } else {
        uint32_t* tile = sb_smem;
        for (uint32_t i0 = 0; i0 < n; i0 += 1024u) {
          const uint32_t i = i0 + threadIdx.x;
          const uint32_t v = i < n ? pairs[b + i] : 0xffffffffu;
          uint32_t rank = 0u;
          for (uint32_t t0 = 0; t0 < n; t0 += 4096u) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < 4096u; t += 1024u) tile[t] = t0 + t < n ? pairs[b + t0 + t] : 0xffffffffu;
            __syncthreads();
            const uint32_t m = min(4096u, n - t0);
            for (uint32_t j = 0; j < m; ++j) rank += tile[j] < v ? 1u : 0u;
          }
          if (i < n) tmp[b + rank] = v;
        }
      }
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n; i += 1024u) pairs[b + i] = tmp[b + i];
      __syncthreads();
    }
    __syncthreads();
  }
}

// labels [n] -> sorted_pairs [n] (pair ids ordered by (label, pair id)), pair_off [n_labels + 1]; scratch: cursor
// [n_labels + 1] (counts, then cursors), tmp [n]. Everything on res.stream, no allocation. Only the first n_sorted labels are
// put into pair order (a list shard's last label is the bucket of the foreign pairs: most pairs of the batch at 8 ranks, never
// scanned - its order does not matter); n_probes / n_queries: pair p belongs to query p / n_probes (0: unknown - long segments
// are then ranked the slow way).
inline void group_pairs(resources& res, const uint32_t* labels, int64_t n, uint32_t n_labels, uint32_t* sorted_pairs, uint32_t* pair_off,
                        uint32_t* cursor, uint32_t* tmp, uint32_t n_sorted, uint32_t n_probes, int64_t n_queries)
{
  CUVS_EXPECTS(n < (int64_t(1) << 32), "group_pairs: more than 2^32 pairs");
  HIP_TRY(hipMemsetAsync(cursor, 0, (size_t)n_labels * sizeof(uint32_t), res.stream));
  n_sorted = std::min(n_sorted, n_labels);
  CUVS_EXPECTS(n_sorted + 1u >= n_labels, "group_pairs: at most one bulk label");
  hipLaunchKernelGGL(pair_histogram_kernel, dim3((unsigned)std::min<int64_t>(grid_blocks(n, 256), 2048)), dim3(256), 0, res.stream, labels, n,
                     cursor, n_sorted, n_labels);
  hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(1024), 0, res.stream, cursor, (int)n_labels, pair_off);
  HIP_TRY(hipMemsetAsync(cursor, 0, (size_t)n_labels * sizeof(uint32_t), res.stream));
  hipLaunchKernelGGL(pair_scatter_kernel, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream, labels, n, pair_off, cursor, sorted_pairs, n_sorted);
  hipLaunchKernelGGL(sort_segments_kernel, dim3(grid_blocks(n_sorted, 4)), dim3(256), 0, res.stream, pair_off, n_sorted, sorted_pairs);
  const bool bitmap   = n_probes != 0u && n_queries <= (int64_t)kBitmapQueries;
  const size_t sbytes = bitmap ? (size_t)2 * ((n_queries + 31) / 32) * sizeof(uint32_t) : (size_t)4096 * sizeof(uint32_t);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_big_segments_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)std::max<size_t>(sbytes, 16384)));
  hipLaunchKernelGGL(sort_big_segments_kernel, dim3(64), dim3(1024), std::max<size_t>(sbytes, 16384), res.stream, pair_off, n_sorted,
                     sorted_pairs, tmp, n_probes, (uint32_t)std::min<int64_t>(n_queries, 0xffffffff));
  HIP_TRY(hipGetLastError());
}

// (query, probe) pairs grouped by list and cut into work items of up to `qpb` pairs of ONE list.
// probes: [n_pairs] list id of pair p (p = query * n_probes + probe rank). Outputs: sorted_pairs[n_pairs],
// items[<= n_pairs / qpb + n_lists + 1], item_off[n_lists] = number of items (device scalar).
// group_scratch: optional [n_lists + 1 + n_pairs] words - with it the grouping runs without the radix sort and without
// allocations (group_pairs above: what the two-stream schedule needs); without it through group_by_label. Same output.
inline void build_work_items(resources& res, const uint32_t* probes, int64_t n_pairs, uint32_t n_lists, int qpb,
                             uint32_t* sorted_pairs, uint32_t* pair_off, uint32_t* item_off, work_item* items,
                             int split = -1, int qpb_hi = 0, uint32_t* group_scratch = nullptr, uint32_t n_sorted = 0xffffffffu,
                             uint32_t n_probes = 0, int64_t n_queries = 0)
{
  if (split < 0) { split = (int)n_lists; qpb_hi = qpb; }
  if (group_scratch != nullptr)
    group_pairs(res, probes, n_pairs, n_lists, sorted_pairs, pair_off, group_scratch, group_scratch + n_lists + 1, n_sorted, n_probes, n_queries);
  else                          group_by_label(res, probes, n_pairs, n_lists, sorted_pairs, pair_off);
  // no items for the labels from n_sorted on (fill_items_kernel writes a label's items from ONE thread, and the foreign pairs of a
  // list shard are never scanned): item_off[n_item_labels] is the end of the last scanned range
  const int n_item_labels = (int)std::min(n_lists, n_sorted);
  hipLaunchKernelGGL(count_items_kernel, dim3(1), dim3(1024), 0, res.stream, pair_off, n_item_labels, qpb, item_off,
                     split, qpb_hi);
  hipLaunchKernelGGL(fill_items_kernel, dim3(grid_blocks(n_item_labels, 256)), dim3(256), 0, res.stream, pair_off,
                     item_off, n_item_labels, qpb, items, split, qpb_hi);
}

}  // namespace
// Top-kp2 of n_l sorted lists (n_l a power of two) by the whole workgroup, in place in LDS: lists l and l + w merge
// into l - C[r] = min(A[r], B[kp2 - 1 - r]) is bitonic and holds the kp2 best of the pair, log2(kp2) compare-exchange
// stages sort it - for w = 1, 2, 4, ... Order: (distance, id) ascending, invalid entries (id 0xffffffff, +inf) last, the
// order of the wave lists. Serial insertion of 16 x 256 candidates by one wave took 300 k cycles per work item in the
// kNN-graph searches of the CAGRA build (k = 256); this takes a few thousand. All threads of the workgroup call it.
template <int NT>
__device__ inline void merge_sorted_lists(float* __restrict__ d, uint32_t* __restrict__ id, const int n_l, const int kp2,
                                          const int tid)
{
  auto less = [](const float da, const uint32_t ia, const float db, const uint32_t ib) {
    return da < db || (da == db && ia < ib);
  };
  for (int w = 1; w < n_l; w <<= 1) {
    const int pairs = n_l / (2 * w);
    for (int t = tid; t < pairs * kp2; t += NT) {
      const int pr = t / kp2, r = t % kp2;
      const int ia = pr * 2 * w * kp2 + r, ib = (pr * 2 * w + w) * kp2 + (kp2 - 1 - r);
      const float da = d[ia], db = d[ib];
      const uint32_t xa = id[ia], xb = id[ib];
      if (less(db, xb, da, xa)) { d[ia] = db; id[ia] = xb; }
    }
    __syncthreads();
    for (int stride = kp2 >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < pairs * (kp2 >> 1); t += NT) {
        const int pr = t / (kp2 >> 1), x = t % (kp2 >> 1);
        const int lo = pr * 2 * w * kp2 + 2 * x - (x & (stride - 1)), hi = lo + stride;
        const float da = d[lo], db = d[hi];
        const uint32_t xa = id[lo], xb = id[hi];
        if (less(db, xb, da, xa)) { d[lo] = db; id[lo] = xb; d[hi] = da; id[hi] = xa; }
      }
      __syncthreads();
    }
  }
}

}  // namespace cuvs_amd
