// CAGRA graph::optimize, the `guarantee_connectivity` pass: a degree-bounded spanning forest over the kNN graph whose
// edges are protected from the prune + reverse-edge merge, so that the search graph has one connected component.
//
// Reference: cpp/src/neighbors/detail/cagra/graph_core.cuh - mst_optimization :1186-1581 (rounds over the edge rank k,
// candidate edge of node i = its rank-k neighbour, final round towards the main cluster :1288-1316), kernels
// kern_mst_opt_update_graph :487-574 (direct edge, else an edge to an incoming neighbour of the target with room
// left), kern_mst_opt_labeling :577-613, kern_mst_opt_cluster_size :616-641, kern_mst_opt_postprocessing :644-718
// (a node whose outgoing slots are full gets one more as long as its row has room), final de-duplication :1551-1576.
// Parameter: cagra::index_params::guarantee_connectivity, cpp/include/cuvs/neighbors/cagra.hpp:193.
//
// MI355X design. The reference keeps the state on the host, copies a column of the kNN graph and four statistics words
// through the PCIe link every round, and lets racing threads decide which edges enter (atomicAdd tickets on the
// incoming counter, labels read while other threads rewrite them): the forest depends on thread timing. Here the whole
// state lives in HBM and every round is a fixed sequence of data-parallel passes over round-START state, so the forest
// is a pure function of the kNN graph (oracle/oracle_cagra_optimize.c: oracle_cagra_mst restates it sequentially):
//   propose   node i (outgoing slots left, rank-k neighbour j in another component) names its target t: j when j has
//             an incoming slot left, else the first incoming neighbour of j that has one; the smallest proposer of
//             every target is kept with one atomicMin per proposal
//   accept    the kept proposer of t writes the edge into its next outgoing slot (front of the row) and into t's next
//             incoming slot (back of the row); a mutual pair (i -> t and t -> i kept in the same round) adds one edge
//   label     components = connected components of the accepted edges over the previous components: min-label hooking
//             (atomicMin on the larger root) repeated until no edge joins two roots, then full pointer compression;
//             the label of a component is its smallest node id whatever the execution order
//   stats     number of components (one readback per round: the loop ends at 1), outgoing-slot growth
// The last round (k == K) links the components outside the largest one to a node inside it (i + 97 m mod n, the
// reference's walk): first only the component roots ask (one edge per component is enough; the reference lets every
// outside node ask, which protects up to n random long edges), then every outside node, repeated with a shifted start
// while it still joins components, since a target takes one incoming edge per round here.
#include "common.hpp"
#include "device_utils.hpp"
#include "ops.hpp"

#include <algorithm>

namespace cuvs_amd {

namespace {

constexpr uint32_t kNone = 0xffffffffu;

__global__ void mst_init_kernel(int64_t n, uint32_t degree, uint32_t* __restrict__ mst, uint32_t* __restrict__ label,
                                uint32_t* __restrict__ out_cnt, uint32_t* __restrict__ in_cnt, uint32_t* __restrict__ out_max,
                                uint32_t* __restrict__ win)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  label[i]   = (uint32_t)i;
  out_cnt[i] = 0;
  in_cnt[i]  = 0;
  out_max[i] = min(2u, degree);  // graph_core.cuh:1228 (incoming_max = degree - outgoing_max)
  win[i]     = kNone;
  for (uint32_t k = 0; k < degree; ++k) mst[i * degree + k] = kNone;
}

// candidate of the last round: the first node of the main component on the walk i + 97 * (shift + m) mod n
__device__ inline uint32_t main_cluster_candidate(int64_t i, int64_t n, const uint32_t* __restrict__ label, uint32_t main_label,
                                                  uint32_t shift)
{
  int64_t j = (i + (int64_t)97 * shift) % n;
  // the stride-97 walk visits one residue class only when 97 divides n: after a full turn (n / 97 + 1 steps) the walk
  // goes on node by node, so it ends after at most n more steps (the main component is never empty)
  for (int64_t s = 0; s <= n / 97 && label[j] != main_label; ++s) j = (j + 97) % n;
  for (int64_t s = 0; s < n && label[j] != main_label; ++s) j = (j + 1) % n;
  return label[j] == main_label ? (uint32_t)j : kNone;
}

__global__ void mst_propose_kernel(const uint32_t* __restrict__ knn, int64_t n, uint32_t K, uint32_t k, uint32_t degree,
                                   const uint32_t* __restrict__ mst, const uint32_t* __restrict__ label,
                                   const uint32_t* __restrict__ out_cnt, const uint32_t* __restrict__ in_cnt,
                                   const uint32_t* __restrict__ out_max, uint32_t* __restrict__ prop, uint32_t* __restrict__ win,
                                   const uint32_t* __restrict__ main_label, uint32_t shift, int roots_only)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t t = kNone;
  if (out_cnt[i] < out_max[i] && !(roots_only && label[i] != (uint32_t)i)) {
    const uint32_t li = label[i];
    uint32_t j        = kNone;
    if (k < K) {
      j = knn[i * K + k];
    } else if (li != *main_label) {
      j = main_cluster_candidate(i, n, label, *main_label, shift);
    }
    if (j < (uint32_t)n && label[j] != li) {
      if (in_cnt[j] < degree - out_max[j]) {
        t = j;
      } else {
        // every incoming neighbour of j is in j's component (labels are compressed at round start)
        for (uint32_t kj = 0; kj < degree; ++kj) {
          const uint32_t l = mst[((int64_t)j + 1) * degree - 1 - kj];
          if (l >= (uint32_t)n) continue;
          if (in_cnt[l] >= degree - out_max[l]) continue;
          t = l;
          break;
        }
      }
    }
  }
  prop[i] = t;
  if (t != kNone) atomicMin(&win[t], (uint32_t)i);
}

__global__ void mst_accept_kernel(int64_t n, uint32_t degree, uint32_t* __restrict__ mst, uint32_t* __restrict__ out_cnt,
                                  uint32_t* __restrict__ in_cnt, uint32_t* __restrict__ prop, const uint32_t* __restrict__ win,
                                  unsigned long long* __restrict__ stats)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = prop[i];
  bool take        = t != kNone && win[t] == (uint32_t)i;
  // mutual pair kept on both sides: the smaller id adds the (bidirectional) edge
  if (take && prop[t] == (uint32_t)i && win[i] == t && t < (uint32_t)i) take = false;
  if (!take) {
    return;
  }
  const uint32_t ki                      = out_cnt[i];
  mst[i * degree + ki]                   = t;  // outgoing: front of the row, written by thread i only
  out_cnt[i]                             = ki + 1;
  const uint32_t kj                      = in_cnt[t];
  mst[((int64_t)t + 1) * degree - 1 - kj] = (uint32_t)i;  // incoming: back of t's row, written by t's kept proposer only
  in_cnt[t]                              = kj + 1;
  atomicAdd(&stats[1], 1ull);
}

// the accepted edge of node i this round, if any (prop/win are still those of the round)
__device__ inline bool accepted_edge(int64_t i, const uint32_t* __restrict__ prop, const uint32_t* __restrict__ win, uint32_t& t)
{
  t = prop[i];
  if (t == kNone || win[t] != (uint32_t)i) return false;
  if (prop[t] == (uint32_t)i && win[i] == t && t < (uint32_t)i) return false;
  return true;
}

__device__ inline uint32_t find_root(const uint32_t* label, uint32_t x)
{
  uint32_t l = __hip_atomic_load(&label[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (l != x) {
    x = l;
    l = __hip_atomic_load(&label[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return x;
}

__global__ void mst_hook_kernel(int64_t n, const uint32_t* __restrict__ prop, const uint32_t* __restrict__ win, uint32_t* label,
                                unsigned long long* __restrict__ stats)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t t;
  if (!accepted_edge(i, prop, win, t)) return;
  const uint32_t ra = find_root(label, (uint32_t)i), rb = find_root(label, t);
  if (ra == rb) return;
  atomicMin(&label[max(ra, rb)], min(ra, rb));
  stats[0] = 1;  // another pass is needed
}

__global__ void mst_compress_kernel(int64_t n, uint32_t* label)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = find_root(label, (uint32_t)i);
  __hip_atomic_store(&label[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// end of round: clear the winners, grow the outgoing budget (graph_core.cuh:693-699), count the components
__global__ void mst_round_end_kernel(int64_t n, uint32_t degree, const uint32_t* __restrict__ label,
                                     const uint32_t* __restrict__ out_cnt, const uint32_t* __restrict__ in_cnt,
                                     uint32_t* __restrict__ out_max, uint32_t* __restrict__ win,
                                     unsigned long long* __restrict__ stats)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool root       = false;
  if (i < n) {
    win[i] = kNone;
    if (out_cnt[i] == out_max[i] && out_cnt[i] + in_cnt[i] < degree) out_max[i] += 1;
    root = label[i] == (uint32_t)i;
  }
  const unsigned long long m = __ballot(root);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&stats[2], (unsigned long long)__popcll(m));
}

__global__ void mst_cluster_size_kernel(int64_t n, const uint32_t* __restrict__ label, uint32_t* __restrict__ size)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&size[label[i]], 1u);
}

// the largest component, the smallest label among equals (graph_core.cuh:1295-1301)
__global__ void mst_main_cluster_kernel(int64_t n, const uint32_t* __restrict__ size, unsigned long long* __restrict__ best)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || size[i] == 0) return;
  atomicMax(best, ((unsigned long long)size[i] << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i));
}

__global__ void mst_main_label_kernel(const unsigned long long* __restrict__ best, uint32_t* __restrict__ main_label)
{
  *main_label = 0xffffffffu - (uint32_t)(*best & 0xffffffffull);
}

// rows compacted in place: valid slots in slot order, duplicates dropped (graph_core.cuh:1551-1576)
__global__ void mst_compact_kernel(int64_t n, uint32_t degree, uint32_t* __restrict__ mst, uint32_t* __restrict__ cnt)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t* row = mst + i * degree;
  uint32_t k    = 0;
  for (uint32_t kj = 0; kj < degree; ++kj) {
    const uint32_t j = row[kj];
    if (j >= (uint32_t)n) continue;
    bool dup = false;
    for (uint32_t ki = 0; ki < k; ++ki) dup |= row[ki] == j;
    if (dup) continue;
    row[k++] = j;
  }
  cnt[i] = k;
  for (uint32_t kj = k; kj < degree; ++kj) row[kj] = kNone;
}

}  // namespace

// mst [n, degree] (device): the protected edges of every node, front-packed; mst_cnt [n]: how many. Returns the number
// of connected components left (1 unless the rows ran out of slots).
int64_t cagra_mst_optimize(resources& res, const uint32_t* knn, int64_t n, uint32_t K, uint32_t degree, uint32_t* mst,
                           uint32_t* mst_cnt)
{
  CUVS_EXPECTS(degree >= 2, "cagra: guarantee_connectivity needs graph_degree >= 2");
  dev_buf<uint32_t> label(res, n), out_cnt(res, n), in_cnt(res, n), out_max(res, n), prop(res, n), win(res, n), main_label(res, 1);
  dev_buf<unsigned long long> stats(res, 4);  // [0] hook flag, [1] edges accepted, [2] components, [3] packed main cluster
  const dim3 grid(grid_blocks(n, 256)), block(256);
  hipLaunchKernelGGL(mst_init_kernel, grid, block, 0, res.stream, n, degree, mst, label.data(), out_cnt.data(), in_cnt.data(),
                     out_max.data(), win.data());
  unsigned long long h[4];
  int64_t clusters = n;
  const int max_last_rounds = 64;
  int last_rounds = 0;
  for (uint32_t k = 0; k <= K && clusters > 1;) {
    if (k == K) {
      // the main cluster of the current labelling
      HIP_TRY(hipMemsetAsync(prop.data(), 0, prop.bytes(), res.stream));  // prop doubles as the size table here
      HIP_TRY(hipMemsetAsync(stats.data() + 3, 0, sizeof(unsigned long long), res.stream));
      hipLaunchKernelGGL(mst_cluster_size_kernel, grid, block, 0, res.stream, n, label.data(), prop.data());
      hipLaunchKernelGGL(mst_main_cluster_kernel, grid, block, 0, res.stream, n, prop.data(), stats.data() + 3);
      hipLaunchKernelGGL(mst_main_label_kernel, dim3(1), dim3(1), 0, res.stream, stats.data() + 3, main_label.data());
    }
    HIP_TRY(hipMemsetAsync(stats.data(), 0, 3 * sizeof(unsigned long long), res.stream));
    hipLaunchKernelGGL(mst_propose_kernel, grid, block, 0, res.stream, knn, n, K, k, degree, mst, label.data(), out_cnt.data(),
                       in_cnt.data(), out_max.data(), prop.data(), win.data(), main_label.data(), (uint32_t)(1 + last_rounds),
                       (int)(k == K && last_rounds == 0));
    hipLaunchKernelGGL(mst_accept_kernel, grid, block, 0, res.stream, n, degree, mst, out_cnt.data(), in_cnt.data(), prop.data(),
                       win.data(), stats.data());
    // components of the accepted edges: hook until no edge joins two roots
    for (;;) {
      hipLaunchKernelGGL(mst_hook_kernel, grid, block, 0, res.stream, n, prop.data(), win.data(), label.data(), stats.data());
      copy_async(res, h, stats.data(), sizeof(unsigned long long));
      HIP_TRY(hipMemsetAsync(stats.data(), 0, sizeof(unsigned long long), res.stream));
      sync(res);
      if (h[0] == 0) break;
    }
    hipLaunchKernelGGL(mst_compress_kernel, grid, block, 0, res.stream, n, label.data());
    hipLaunchKernelGGL(mst_round_end_kernel, grid, block, 0, res.stream, n, degree, label.data(), out_cnt.data(), in_cnt.data(),
                       out_max.data(), win.data(), stats.data());
    copy_async(res, h, stats.data(), 3 * sizeof(unsigned long long));
    sync(res);
    clusters = (int64_t)h[2];
    if (k < K) {
      ++k;
    } else {
      ++last_rounds;
      // the first pass lets only the component roots ask (one edge per component is enough); later passes every node
      if ((h[1] == 0 && last_rounds > 1) || last_rounds >= max_last_rounds) break;  // no edge could be added: rows are full
    }
  }
  hipLaunchKernelGGL(mst_compact_kernel, grid, block, 0, res.stream, n, degree, mst, mst_cnt);
  HIP_TRY(hipGetLastError());
  return clusters;
}

}  // namespace cuvs_amd
