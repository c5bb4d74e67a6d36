// NN-descent kNN-graph builder: CAGRA's `build_algo = NN_DESCENT`.
//
// Reference: cpp/src/neighbors/detail/nn_descent.cuh (GNND: sampled new/old neighbour lists, reverse lists, local
// join, k-list update with locks) and cagra_build.cuh:2208-2217 (the default graph builder when memory allows).
// The algorithm is Dong et al.'s NN-descent; the schedule here is MI355X-first rather than a restatement of the
// reference's kernels:
//   * every node keeps a sorted K-list (id | NEW flag, distance key) in HBM;
//   * per iteration: `sample` (one wave per node) draws up to S new and S old entries and publishes them into the
//     reverse lists of their targets with one atomic slot counter per target; `local_join` (one workgroup per node)
//     evaluates every (new, new) and (new, old) pair of the node's forward + reverse samples with 8-lane teams
//     (the dimension-order fma arithmetic of the CAGRA search) and appends the pairs that beat an endpoint's
//     current worst distance to that endpoint's PROPOSAL buffer (atomic slot counter, overflow dropped);
//     `merge` (one wave per node) sorts list + proposals with the wave bitonic sort in LDS, removes duplicates
//     and keeps the K best - no per-node locks: 288 GB of HBM pays for proposal buffers instead;
//   * iterations stop after `n_iters` rounds or when fewer than 0.01 % of the list slots changed
//     (nn_descent.cuh termination_threshold).
// The result is approximate and depends on atomic ordering (as the reference's): tests check graph recall.
#include "ops.hpp"
#include "device_utils.hpp"

#include <cfloat>

namespace cuvs_amd {

void load_range_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim, int64_t r0,
                         int64_t cnt, float* out);
void load_gather_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim,
                          const uint32_t* d_ids, int64_t cnt, float* out);

namespace {

constexpr uint32_t kNew     = 0x80000000u;
constexpr uint32_t kNone    = 0xffffffffu;
constexpr int kSamples      = 32;   // S: new / old samples per node and direction
constexpr int kJoinThreads  = 256;

__device__ inline uint64_t xs64(uint64_t u)
{
  u ^= u >> 12; u ^= u << 25; u ^= u >> 27;
  return u * 0x2545F4914F6CDD1DULL;
}

// distance key of rows a, b computed by an 8-lane team (lane tl of the team): smaller = closer
template <typename T>
__device__ inline uint32_t team_pair_key(const T* __restrict__ data, int64_t dim, uint32_t a, uint32_t b, int mode,
                                         const float* __restrict__ norms, int tl)
{
  constexpr int VL = 16 / sizeof(T);
  const T* ra = data + (int64_t)a * dim;
  const T* rb = data + (int64_t)b * dim;
  float acc   = 0.f;
  for (int64_t d0 = (int64_t)tl * VL; d0 < dim; d0 += 8 * VL) {
#pragma unroll
    for (int e = 0; e < VL; ++e) {
      if (d0 + e < dim) {
        const float x = to_float(ra[d0 + e]), y = to_float(rb[d0 + e]);
        if (mode == 0) { const float t = x - y; acc = __fmaf_rn(t, t, acc); }
        else           acc = __fmaf_rn(x, y, acc);
      }
    }
  }
  acc = acc + __shfl_xor(acc, 1, 64);
  acc = acc + __shfl_xor(acc, 2, 64);
  acc = acc + __shfl_xor(acc, 4, 64);
  if (mode == 1) acc = -acc;                                   // inner product: larger is closer
  if (mode == 2) acc = 1.0f - acc / (norms[a] * norms[b]);     // cosine
  return float_to_key(acc);
}

// wave bitonic sort of (key, id) pairs ordered by (key, id without flag, flag): the copy of a neighbour that is
// already in the list sorts directly in front of its re-proposal even when other ids tie on the key
__device__ inline void wave_sort_entries(uint32_t* keys, uint32_t* idx, int n)
{
  const int lane = threadIdx.x & 63;
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int t = lane; t < (n >> 1); t += 64) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint32_t ka = keys[lo], kb = keys[hi], ia = idx[lo], ib = idx[hi];
        const uint32_t na = ia & ~kNew, nb = ib & ~kNew;
        const bool a_gt_b = (ka > kb) || (ka == kb && (na > nb || (na == nb && ia > ib)));
        if (a_gt_b == up) { keys[lo] = kb; keys[hi] = ka; idx[lo] = ib; idx[hi] = ia; }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---------------------------------------------------------------- proposals
struct nnd_state {
  uint32_t* ids;       // [n, K] neighbour id | kNew, sorted by key
  uint32_t* keys;      // [n, K]
  uint32_t* worst;     // [n] key of the K-th entry (0xffffffff while the list is not full)
  uint32_t* prop_ids;  // [n, P]
  uint32_t* prop_keys; // [n, P]
  uint32_t* prop_cnt;  // [n]
  uint32_t* fwd_new;   // [n, S]
  uint32_t* fwd_old;   // [n, S]
  uint32_t* rev_new;   // [n, S]
  uint32_t* rev_old;   // [n, S]
  uint32_t* rev_new_cnt;  // [n]
  uint32_t* rev_old_cnt;  // [n]
  unsigned long long* n_updates;  // device scalar
  int64_t n;
  uint32_t K, P;
};

__device__ inline void propose(const nnd_state& st, uint32_t target, uint32_t cand, uint32_t key)
{
  if (key >= st.worst[target]) return;
  const uint32_t slot = atomicAdd(&st.prop_cnt[target], 1u);
  if (slot < st.P) {
    st.prop_ids[(int64_t)target * st.P + slot]  = cand;
    st.prop_keys[(int64_t)target * st.P + slot] = key;
  }
}

// random initial neighbours, delivered as proposals into the (empty) lists
// With a coarse clustering (perm = rows grouped by cluster, pos_of = inverse of perm, cl_off = cluster offsets,
// labels): the first candidates are the rows that follow v inside its own cluster - already related points, which
// saves the many rounds NN-descent needs to get from random lists to neighbourhoods on large datasets.
template <typename T>
__global__ __launch_bounds__(256) void nnd_init_kernel(nnd_state st, const T* __restrict__ data, int64_t dim, int mode,
                                                        const float* __restrict__ norms, uint64_t seed,
                                                        const uint32_t* __restrict__ perm,
                                                        const uint32_t* __restrict__ pos_of,
                                                        const uint32_t* __restrict__ cl_off,
                                                        const uint32_t* __restrict__ labels,
                                                        const uint32_t* __restrict__ hubs, uint32_t n_hubs)
{
  const int team = threadIdx.x >> 3, tl = threadIdx.x & 7;  // 32 teams
  const int64_t v = blockIdx.x;
  const uint32_t n_init = min(st.K, st.P);
  uint32_t c_begin = 0, c_size = 0, c_pos = 0;
  if (perm != nullptr) {
    const uint32_t L = labels[v];
    c_begin = cl_off[L]; c_size = cl_off[L + 1] - c_begin; c_pos = pos_of[v] - c_begin;
  }
  for (uint32_t c = team; c < n_init; c += 32) {
    uint32_t u;
    // even slots: cluster mates (local structure); odd slots: random rows (links out of the cluster - with
    // cluster mates only the descent converges inside the clusters and never sees the neighbours across borders)
    // inner product: every fourth slot is one of the rows of LARGEST NORM - for unnormalised rows the same few rows are
    // among everybody's best inner products; random lists find them through reverse samples of 32 entries that those very
    // rows overflow (graph recall 0.84 at 4000 x 1024, degree 32: the reference's ann_nn_descent table asks for 0.9)
    if (hubs != nullptr && (c & 3u) == 3u && c / 4 < n_hubs) u = hubs[c / 4];
    else if ((c & 1u) == 0u && c / 2 + 1 < c_size) u = perm[c_begin + (c_pos + 1 + c / 2) % c_size];
    else                                           u = (uint32_t)(xs64(((uint64_t)v * st.K + c) ^ seed) % (uint64_t)st.n);
    if (u == (uint32_t)v) u = (u + 1) % (uint32_t)st.n;
    const uint32_t key = team_pair_key<T>(data, dim, (uint32_t)v, u, mode, norms, tl);
    if (tl == 0) {
      st.prop_ids[v * st.P + c]  = u;
      st.prop_keys[v * st.P + c] = key;
    }
  }
  if (threadIdx.x == 0) st.prop_cnt[v] = n_init;
}

// list + proposals -> K best distinct entries (one wave per node); new arrivals get the kNew flag
__global__ __launch_bounds__(256) void nnd_merge_kernel(nnd_state st, int np2)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + wave;
  if (v >= st.n) return;
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * 2 * np2;
  uint32_t* idx  = keys + np2;
  const uint32_t K = st.K, cnt = min(st.prop_cnt[v], st.P);
  for (int i = lane; i < np2; i += 64) {
    uint32_t k = 0xffffffffu, id = kNone;
    if ((uint32_t)i < K) { k = st.keys[v * K + i]; id = st.ids[v * K + i]; }
    else if ((uint32_t)i - K < cnt) {
      k  = st.prop_keys[v * st.P + (i - K)];
      id = st.prop_ids[v * st.P + (i - K)];
      id = (id == kNone || id == (uint32_t)v) ? kNone : (id | kNew);
      if (id == kNone) k = 0xffffffffu;
    }
    keys[i] = k; idx[i] = id;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // order: (key, id without flag, flag) - an entry already in the list sorts before its re-proposal, so the
  // duplicate that is dropped is the proposal
  wave_sort_entries(keys, idx, np2);
  // duplicates are adjacent (same pair => same arithmetic => same key)
  uint32_t kept = 0;
  for (int base = 0; base < np2 && kept < K; base += 64) {
    const int i        = base + lane;
    const uint32_t id  = i < np2 ? idx[i] : kNone;
    const uint32_t k   = i < np2 ? keys[i] : 0xffffffffu;
    bool ok            = id != kNone;
    if (ok && i > 0) {
      const uint32_t pid = idx[i - 1];
      if (pid != kNone && ((pid ^ id) & ~kNew) == 0u) ok = false;
    }
    const unsigned long long m = __ballot(ok);
    const uint32_t pos         = kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (ok && pos < K) {
      st.ids[v * K + pos]  = id;
      st.keys[v * K + pos] = k;
    }
    kept += (uint32_t)__popcll(m);
  }
  kept = min(kept, K);
  for (uint32_t i = kept + lane; i < K; i += 64) { st.ids[v * K + i] = kNone; st.keys[v * K + i] = 0xffffffffu; }
  if (lane == 0) {
    st.worst[v]    = kept == K ? st.keys[v * K + K - 1] : 0xffffffffu;
    st.prop_cnt[v] = 0;
  }
}

// count entries that still carry the kNew flag (= arrived since the last sample) for the convergence test
__global__ void nnd_count_new_kernel(nnd_state st)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool is_new = i < st.n * (int64_t)st.K && st.ids[i] != kNone && (st.ids[i] & kNew);
  const unsigned long long m = __ballot(is_new);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(st.n_updates, (unsigned long long)__popcll(m));
}

// forward samples + reverse publication (one wave per node)
__global__ __launch_bounds__(256) void nnd_sample_kernel(nnd_state st)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + wave;
  if (v >= st.n) return;
  const uint32_t K = st.K;
  uint32_t n_new = 0, n_old = 0;
  for (uint32_t base = 0; base < K; base += 64) {
    const uint32_t i  = base + lane;
    const uint32_t id = i < K ? st.ids[v * K + i] : kNone;
    const bool valid  = id != kNone;
    const bool isnew  = valid && (id & kNew);
    const unsigned long long mn = __ballot(isnew), mo = __ballot(valid && !isnew);
    const uint32_t pn = n_new + (uint32_t)__popcll(mn & ((1ull << lane) - 1ull));
    const uint32_t po = n_old + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull));
    if (isnew && pn < kSamples) {
      const uint32_t u = id & ~kNew;
      st.fwd_new[v * kSamples + pn] = u;
      st.ids[v * K + i]             = u;  // sampled: old from now on
      const uint32_t slot = atomicAdd(&st.rev_new_cnt[u], 1u);
      if (slot < kSamples) st.rev_new[(int64_t)u * kSamples + slot] = (uint32_t)v;
    }
    if (valid && !isnew && po < kSamples) {
      st.fwd_old[v * kSamples + po] = id;
      const uint32_t slot = atomicAdd(&st.rev_old_cnt[id], 1u);
      if (slot < kSamples) st.rev_old[(int64_t)id * kSamples + slot] = (uint32_t)v;
    }
    n_new += (uint32_t)__popcll(mn);
    n_old += (uint32_t)__popcll(mo);
  }
  for (uint32_t i = min(n_new, (uint32_t)kSamples) + lane; i < (uint32_t)kSamples; i += 64) st.fwd_new[v * kSamples + i] = kNone;
  for (uint32_t i = min(n_old, (uint32_t)kSamples) + lane; i < (uint32_t)kSamples; i += 64) st.fwd_old[v * kSamples + i] = kNone;
}

// local join of node v: every (new, new) and (new, old) pair of its forward + reverse samples
template <typename T>
__global__ __launch_bounds__(kJoinThreads) void nnd_join_kernel(nnd_state st, const T* __restrict__ data, int64_t dim,
                                                                int mode, const float* __restrict__ norms)
{
  __shared__ uint32_t cn[2 * kSamples], co[2 * kSamples];
  __shared__ uint32_t n_cn, n_co;
  const int64_t v = blockIdx.x;
  if (threadIdx.x == 0) { n_cn = 0; n_co = 0; }
  __syncthreads();
  // gather (duplicates between the forward and the reverse list only cost a repeated proposal, merged away later)
  if (threadIdx.x < 2 * kSamples) {
    const int i = threadIdx.x;
    const uint32_t rn = min(st.rev_new_cnt[v], (uint32_t)kSamples), ro = min(st.rev_old_cnt[v], (uint32_t)kSamples);
    uint32_t a = i < kSamples ? st.fwd_new[v * kSamples + i] : ((uint32_t)(i - kSamples) < rn ? st.rev_new[v * kSamples + i - kSamples] : kNone);
    uint32_t b = i < kSamples ? st.fwd_old[v * kSamples + i] : ((uint32_t)(i - kSamples) < ro ? st.rev_old[v * kSamples + i - kSamples] : kNone);
    if (a != kNone) cn[atomicAdd(&n_cn, 1u)] = a;
    if (b != kNone) co[atomicAdd(&n_co, 1u)] = b;
  }
  __syncthreads();
  const uint32_t Nn = n_cn, No = n_co;
  const uint32_t pairs_nn = Nn * (Nn - (Nn > 0 ? 1u : 0u)) / 2, pairs = pairs_nn + Nn * No;
  const int team = threadIdx.x >> 3, tl = threadIdx.x & 7;
  for (uint32_t p = team; p < pairs; p += kJoinThreads / 8) {
    uint32_t a, b;
    if (p < pairs_nn) {
      // p -> (i, j), i < j: row i holds Nn-1-i pairs
      uint32_t i = 0, rem = p;
      while (rem >= Nn - 1 - i) { rem -= Nn - 1 - i; ++i; }
      a = cn[i]; b = cn[i + 1 + rem];
    } else {
      const uint32_t q = p - pairs_nn;
      a = cn[q / No]; b = co[q % No];
    }
    if (a == b) continue;  // team-uniform
    const uint32_t key = team_pair_key<T>(data, dim, a, b, mode, norms, tl);
    if (tl == 0) { propose(st, a, b, key); propose(st, b, a, key); }
  }
}

__global__ void nnd_inverse_perm_kernel(const uint32_t* __restrict__ perm, int64_t n, uint32_t* __restrict__ pos_of);

template <typename T>
void nnd_run(resources& res, const T* data, elem_t et, int64_t n, int64_t dim, uint32_t K_out, int mode, const float* norms,
             int n_iters, uint32_t* knn_out, uint32_t* keys_out, float termination_threshold)
{
  // The lists the descent works on are LONGER than the graph it returns, as in the reference (nn_descent_gnnd.hpp:304-310:
  // internal_node_degree = roundUp32(1.3 x intermediate_graph_degree) beyond 32): with lists of exactly the requested length
  // the descent on 4000 x 1024 rows / inner product / 64 settles at a graph recall of 0.84, with 96 at 0.9+ - found by the
  // reference's own ann_nn_descent table (min_recall 0.9), round 5
  const uint32_t K = (uint32_t)std::min<int64_t>(n - 1, K_out <= 32 ? K_out : (uint32_t)round_up((int64_t)(K_out * 1.3), 32));
  const uint32_t P = std::max<uint32_t>(64, 2 * K);  // proposals a node accepts per round (first come, first kept)
  dev_buf<uint32_t> ids(res, (size_t)n * K), keys(res, (size_t)n * K), worst(res, n), prop_ids(res, (size_t)n * P),
    prop_keys(res, (size_t)n * P), prop_cnt(res, n), fwd_new(res, (size_t)n * kSamples), fwd_old(res, (size_t)n * kSamples),
    rev_new(res, (size_t)n * kSamples), rev_old(res, (size_t)n * kSamples), rev_new_cnt(res, n), rev_old_cnt(res, n);
  dev_buf<unsigned long long> n_updates(res, 1);
  nnd_state st;
  st.ids = ids.data(); st.keys = keys.data(); st.worst = worst.data(); st.prop_ids = prop_ids.data();
  st.prop_keys = prop_keys.data(); st.prop_cnt = prop_cnt.data(); st.fwd_new = fwd_new.data(); st.fwd_old = fwd_old.data();
  st.rev_new = rev_new.data(); st.rev_old = rev_old.data(); st.rev_new_cnt = rev_new_cnt.data();
  st.rev_old_cnt = rev_old_cnt.data(); st.n_updates = n_updates.data(); st.n = n; st.K = K; st.P = P;
  HIP_TRY(hipMemsetAsync(ids.data(), 0xff, ids.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(keys.data(), 0xff, keys.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(worst.data(), 0xff, worst.bytes(), res.stream));
  CUVS_EXPECTS(n < (int64_t(1) << 31), "nn_descent: at most 2^31 rows");
  const int np2     = next_pow2((int)(K + P));
  const size_t msm  = (size_t)4 * 2 * np2 * sizeof(uint32_t);
  // (a node's list and its proposals are merged in LDS: K + P = 3 x roundUp32(1.3 x degree) entries <= 4096, i.e. a graph degree up
  // to 1024 - the reference's own tables stop at 64; memory: n x (K + 2 P) x 8 bytes, ~30 GB at 10M rows and degree 128)
  CUVS_EXPECTS(msm <= 160 * 1024,
               "nn_descent: graph degree %u is too large for the merge step (internal lists of %u + %u proposals must fit 4096 LDS entries: "
               "at most degree 1024)", K_out, K, P);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(nnd_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)msm));
  const unsigned g4 = grid_blocks(n, 4);
  // ---- coarse clustering for the initial lists: clusters of ~4 K rows (balanced k-means on a strided sample)
  dev_buf<uint32_t> perm, pos_of, cl_off, labels;
  const int64_t n_clusters = std::min<int64_t>(65536, n / (4 * (int64_t)K));
  if (n_clusters >= 2) {
    const int64_t n_train = std::min<int64_t>(n, std::max<int64_t>(n_clusters * 64, 100000));
    const int64_t stride  = std::max<int64_t>(1, n / n_train);
    const int64_t nt      = n / stride;
    dev_buf<float> train(res, (size_t)nt * dim), centers(res, (size_t)n_clusters * dim);
    {
      std::vector<uint32_t> h_ids(nt);
      for (int64_t i = 0; i < nt; ++i) h_ids[i] = (uint32_t)(i * stride);
      dev_buf<uint32_t> d_ids(res, nt);
      copy_async(res, d_ids.data(), h_ids.data(), nt * sizeof(uint32_t));
      load_gather_as_float(res, data, et, false, dim, d_ids.data(), nt, train.data());
      sync(res);
    }
    if (mode == 2) normalize_rows(res, train.data(), nt, dim);
    kmeans_params kp;
    kp.n_iters = 10;
    kmeans_balanced_fit(res, train.data(), nt, dim, (int)n_clusters, kp, centers.data());
    train.release();
    labels = dev_buf<uint32_t>(res, n);
    const int64_t batch = std::max<int64_t>(1024, std::min<int64_t>(n, (int64_t(1) << 28) / dim));
    dev_buf<float> xb(res, (size_t)std::min(batch, n) * dim);
    for (int64_t r0 = 0; r0 < n; r0 += batch) {
      const int64_t cnt = std::min(batch, n - r0);
      load_range_as_float(res, data, et, false, dim, r0, cnt, xb.data());
      if (mode == 2) normalize_rows(res, xb.data(), cnt, dim);
      kmeans_predict<float>(res, xb.data(), cnt, dim, centers.data(), (int)n_clusters, labels.data() + r0);
    }
    perm   = dev_buf<uint32_t>(res, n);
    pos_of = dev_buf<uint32_t>(res, n);
    cl_off = dev_buf<uint32_t>(res, n_clusters + 1);
    group_by_label(res, labels.data(), n, (uint32_t)n_clusters, perm.data(), cl_off.data());
    hipLaunchKernelGGL(nnd_inverse_perm_kernel, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream, perm.data(), n,
                       pos_of.data());
  }
  // inner product: the K / 4 rows of largest squared norm (exact select over the canonical norms)
  dev_buf<uint32_t> hubs;
  uint32_t n_hubs = 0;
  if (mode == 1 && n > 4 * (int64_t)K) {
    n_hubs = std::max<uint32_t>(1, K / 4);
    dev_buf<float> sq(res, n), hv(res, n_hubs);
    hubs = dev_buf<uint32_t>(res, n_hubs);
    row_norms<T>(res, data, n, dim, dim, sq.data(), false);
    select_k<uint32_t, uint32_t>(res, sq.data(), nullptr, 1, n, n, (int)n_hubs, hv.data(), hubs.data(), false);
  }
  hipLaunchKernelGGL((nnd_init_kernel<T>), dim3((unsigned)n), dim3(256), 0, res.stream, st, data, dim, mode, norms,
                     0x9E3779B97F4A7C15ull, perm.data(), pos_of.data(), cl_off.data(), labels.data(), hubs.data(), n_hubs);
  hipLaunchKernelGGL(nnd_merge_kernel, dim3(g4), dim3(256), msm, res.stream, st, np2);
  for (int it = 0; it < n_iters; ++it) {
    HIP_TRY(hipMemsetAsync(rev_new_cnt.data(), 0, rev_new_cnt.bytes(), res.stream));
    HIP_TRY(hipMemsetAsync(rev_old_cnt.data(), 0, rev_old_cnt.bytes(), res.stream));
    HIP_TRY(hipMemsetAsync(n_updates.data(), 0, sizeof(unsigned long long), res.stream));
    hipLaunchKernelGGL(nnd_count_new_kernel, dim3(grid_blocks(n * K, 256)), dim3(256), 0, res.stream, st);
    unsigned long long upd = 0;
    copy_async(res, &upd, n_updates.data(), sizeof(upd));
    sync(res);
    // nn_descent.cuh: stop when fewer than termination_threshold (1e-4) of the n*K slots changed
    if (it > 0 && (double)upd < (double)termination_threshold * (double)n * (double)K) break;
    hipLaunchKernelGGL(nnd_sample_kernel, dim3(g4), dim3(256), 0, res.stream, st);
    hipLaunchKernelGGL((nnd_join_kernel<T>), dim3((unsigned)n), dim3(kJoinThreads), 0, res.stream, st, data, dim, mode,
                       norms);
    hipLaunchKernelGGL(nnd_merge_kernel, dim3(g4), dim3(256), msm, res.stream, st, np2);
    HIP_TRY(hipGetLastError());
  }
  // ids without flags -> output (invalid entries stay 0xffffffff)
  HIP_TRY(hipMemcpy2DAsync(knn_out, (size_t)K_out * 4, ids.data(), (size_t)K * 4, (size_t)K_out * 4, (size_t)n, hipMemcpyDeviceToDevice,
                           res.stream));
  if (keys_out)
    HIP_TRY(hipMemcpy2DAsync(keys_out, (size_t)K_out * 4, keys.data(), (size_t)K * 4, (size_t)K_out * 4, (size_t)n,
                             hipMemcpyDeviceToDevice, res.stream));
  sync(res);
}

__global__ void nnd_inverse_perm_kernel(const uint32_t* __restrict__ perm, int64_t n, uint32_t* __restrict__ pos_of)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos_of[perm[i]] = (uint32_t)i;
}

__global__ void nnd_strip_flags_kernel(uint32_t* ids, int64_t total)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total && ids[i] != kNone) ids[i] &= ~kNew;
}

}  // namespace

// kNN graph [n, K] (uint32, sorted by distance, self excluded) by NN-descent. data: device rows of element type et.
// metric: L2 family / inner product / cosine (cosine needs canonical row norms |x|).
void knn_graph_nn_descent(resources& res, const void* data, elem_t et, int64_t n, int64_t dim, uint32_t K, int metric,
                          const float* norms, int n_iters, uint32_t* knn, uint32_t* keys_out, float termination_threshold)
{
  CUVS_EXPECTS(n > (int64_t)K, "nn_descent: need more rows than the graph degree");
  const int mode = metric == M_InnerProduct ? 1 : (metric == M_CosineExpanded ? 2 : 0);
  CUVS_EXPECTS(mode != 2 || norms != nullptr, "nn_descent: cosine needs row norms");
  if (n_iters <= 0) n_iters = 20;
  switch (et) {
    case elem_t::f32: nnd_run<float>(res, static_cast<const float*>(data), et, n, dim, K, mode, norms, n_iters, knn, keys_out, termination_threshold); break;
    case elem_t::f16: nnd_run<__half>(res, static_cast<const __half*>(data), et, n, dim, K, mode, norms, n_iters, knn, keys_out, termination_threshold); break;
    case elem_t::i8: nnd_run<int8_t>(res, static_cast<const int8_t*>(data), et, n, dim, K, mode, norms, n_iters, knn, keys_out, termination_threshold); break;
    case elem_t::u8: nnd_run<uint8_t>(res, static_cast<const uint8_t*>(data), et, n, dim, K, mode, norms, n_iters, knn, keys_out, termination_threshold); break;
  }
  hipLaunchKernelGGL(nnd_strip_flags_kernel, dim3(grid_blocks(n * (int64_t)K, 256)), dim3(256), 0, res.stream, knn,
                     n * (int64_t)K);
  HIP_TRY(hipGetLastError());
  sync(res);
}

}  // namespace cuvs_amd

// test / bench hook (not part of the reference ABI): NN-descent kNN graph of device fp32 rows
extern "C" __attribute__((visibility("default"))) int cuvsAmdNnDescent(uintptr_t res_h, const float* data, int64_t n,
                                                                        int64_t dim, uint32_t K, int metric, int n_iters,
                                                                        uint32_t* knn)
{
  using namespace cuvs_amd;
  return translate_exceptions([=] {
    auto& res = *reinterpret_cast<resources*>(res_h);
    dev_buf<float> norms;
    if (metric == M_CosineExpanded) {
      norms = dev_buf<float>(res, n);
      row_norms<float>(res, data, n, dim, dim, norms.data(), true);
    }
    knn_graph_nn_descent(res, data, elem_t::f32, n, dim, K, metric, norms.data(), n_iters, knn);
  });
}

// ------------------------------------------------------------------ cuvsNNDescent* C API (c/src/neighbors/nn_descent.cpp)
#include <cuvs/neighbors/nn_descent.h>

namespace {
struct nnd_index {
  int64_t n = 0;
  uint32_t degree = 0;
  int metric = 0;
  cuvs_amd::dev_buf<uint32_t> graph;  // [n, degree]
  cuvs_amd::dev_buf<float> dist;      // [n, degree] (return_distances)
};

__global__ void nnd_slice_kernel(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ keys, int64_t n, uint32_t K,
                                 uint32_t degree, int metric, uint32_t* __restrict__ g, float* __restrict__ d)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * (int64_t)degree) return;
  const int64_t r = i / degree, c = i % degree;
  g[i] = ids[r * K + c];
  if (d != nullptr) {
    float v = cuvs_amd::key_to_float(keys[r * K + c]);
    if (metric == cuvs_amd::M_InnerProduct) v = -v;
    if (metric == cuvs_amd::M_L2SqrtExpanded || metric == cuvs_amd::M_L2SqrtUnexpanded) v = sqrtf(v);
    d[i] = ids[r * K + c] == 0xffffffffu ? FLT_MAX : v;
  }
}
}  // namespace

extern "C" {
cuvsError_t cuvsNNDescentIndexParamsCreate(cuvsNNDescentIndexParams_t* params)
{
  return (cuvsError_t)cuvs_amd::translate_exceptions([=] {
    CUVS_EXPECTS(params != nullptr, "params is null");
    // defaults of cuvs::neighbors::nn_descent::index_params (nn_descent.hpp:62-67)
    *params = new cuvsNNDescentIndexParams{L2Expanded, 2.0f, 64, 128, 20, 0.0001f, true, NND_DIST_COMP_AUTO};
  });
}
cuvsError_t cuvsNNDescentIndexParamsDestroy(cuvsNNDescentIndexParams_t params)
{
  return (cuvsError_t)cuvs_amd::translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsNNDescentIndexCreate(cuvsNNDescentIndex_t* index)
{
  return (cuvsError_t)cuvs_amd::translate_exceptions([=] {
    CUVS_EXPECTS(index != nullptr, "index is null");
    *index = new cuvsNNDescentIndex{0, DLDataType{kDLUInt, 32, 1}};
  });
}
cuvsError_t cuvsNNDescentIndexDestroy(cuvsNNDescentIndex_t index)
{
  return (cuvsError_t)cuvs_amd::translate_exceptions([=] {
    if (index == nullptr) return;
    delete reinterpret_cast<nnd_index*>(index->addr);
    delete index;
  });
}

cuvsError_t cuvsNNDescentBuild(cuvsResources_t res_h, cuvsNNDescentIndexParams_t params, DLManagedTensor* dataset_tensor,
                               DLManagedTensor* graph_tensor, cuvsNNDescentIndex_t index)
{
  using namespace cuvs_amd;
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *reinterpret_cast<resources*>(res_h);
    CUVS_EXPECTS(params && dataset_tensor && index, "null argument");
    auto& ds = dataset_tensor->dl_tensor;
    CUVS_EXPECTS(ds.ndim == 2 && is_c_contiguous(ds), "dataset must be a row-major matrix");
    const elem_t et = elem_of(ds.dtype);  // fails for unsupported dtypes
    const int64_t n = ds.shape[0], dim = ds.shape[1];
    const int metric = (int)params->metric;
    CUVS_EXPECTS(metric_is_l2(metric) || metric == M_InnerProduct || metric == M_CosineExpanded,
                 "nn_descent: unsupported metric %d", metric);
    const uint32_t degree = (uint32_t)params->graph_degree;
    uint32_t K = (uint32_t)std::max(params->intermediate_graph_degree, params->graph_degree);
    K          = (uint32_t)std::min<int64_t>(K, n - 1);
    CUVS_EXPECTS(degree >= 1 && degree <= K, "nn_descent: graph_degree must be in [1, min(intermediate_graph_degree, n - 1)]");
    const void* data = dl_data(ds);
    dev_buf<char> staged;
    if (!is_device_accessible(ds)) {
      staged = dev_buf<char>(res, (size_t)n * dim * elem_size(et));
      copy_async(res, staged.data(), data, staged.bytes());
      data = staged.data();
    }
    dev_buf<float> norms;
    if (metric == M_CosineExpanded) {
      norms = dev_buf<float>(res, n);
      switch (et) {
        case elem_t::f32: row_norms<float>(res, static_cast<const float*>(data), n, dim, dim, norms.data(), true); break;
        case elem_t::f16: row_norms<__half>(res, static_cast<const __half*>(data), n, dim, dim, norms.data(), true); break;
        case elem_t::i8: row_norms<int8_t>(res, static_cast<const int8_t*>(data), n, dim, dim, norms.data(), true); break;
        case elem_t::u8: row_norms<uint8_t>(res, static_cast<const uint8_t*>(data), n, dim, dim, norms.data(), true); break;
      }
    }
    dev_buf<uint32_t> ids(res, (size_t)n * K), keys(res, (size_t)n * K);
    knn_graph_nn_descent(res, data, et, n, dim, K, metric, norms.data(), (int)params->max_iterations, ids.data(),
                         keys.data(), params->termination_threshold > 0.f ? params->termination_threshold : 1e-4f);
    auto idx    = std::make_unique<nnd_index>();
    idx->n      = n;
    idx->degree = degree;
    idx->metric = metric;
    idx->graph  = dev_buf<uint32_t>::persistent((size_t)n * degree);
    if (params->return_distances) idx->dist = dev_buf<float>::persistent((size_t)n * degree);
    hipLaunchKernelGGL(nnd_slice_kernel, dim3(grid_blocks(n * (int64_t)degree, 256)), dim3(256), 0, res.stream, ids.data(),
                       keys.data(), n, K, degree, metric, idx->graph.data(), idx->dist.data());
    HIP_TRY(hipGetLastError());
    if (graph_tensor != nullptr) {
      auto& g = graph_tensor->dl_tensor;
      CUVS_EXPECTS(dtype_is(g.dtype, kDLUInt, 32) && g.ndim == 2 && g.shape[0] == n && g.shape[1] == degree &&
                     is_c_contiguous(g),
                   "graph must be uint32 [n, graph_degree]");
      copy_async(res, dl_data(g), idx->graph.data(), idx->graph.bytes());
    }
    sync(res);
    delete reinterpret_cast<nnd_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = DLDataType{kDLUInt, 32, 1};
  });
}

cuvsError_t cuvsNNDescentIndexGetGraph(cuvsResources_t res_h, cuvsNNDescentIndex_t index, DLManagedTensor* graph)
{
  using namespace cuvs_amd;
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *reinterpret_cast<resources*>(res_h);
    CUVS_EXPECTS(index && index->addr && graph, "nn-descent index is not built");
    auto& idx = *reinterpret_cast<nnd_index*>(index->addr);
    auto& g   = graph->dl_tensor;
    CUVS_EXPECTS(dtype_is(g.dtype, kDLUInt, 32) && g.ndim == 2 && is_c_contiguous(g), "graph must be a uint32 matrix");
    CUVS_EXPECTS(g.shape[0] == idx.n, "Output graph has incorrect number of rows");
    CUVS_EXPECTS(g.shape[1] == idx.degree, "Output graph has incorrect number of cols");
    copy_async(res, dl_data(g), idx.graph.data(), idx.graph.bytes());
    sync(res);
  });
}

cuvsError_t cuvsNNDescentIndexGetDistances(cuvsResources_t res_h, cuvsNNDescentIndex_t index, DLManagedTensor* distances)
{
  using namespace cuvs_amd;
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *reinterpret_cast<resources*>(res_h);
    CUVS_EXPECTS(index && index->addr && distances, "nn-descent index is not built");
    auto& idx = *reinterpret_cast<nnd_index*>(index->addr);
    CUVS_EXPECTS(idx.dist.data() != nullptr,
                 "nn-descent index doesn't contain distances - set return_distances when building");
    auto& d = distances->dl_tensor;
    CUVS_EXPECTS(dtype_is(d.dtype, kDLFloat, 32) && d.ndim == 2 && is_c_contiguous(d), "distances must be a float32 matrix");
    CUVS_EXPECTS(d.shape[0] == idx.n, "Output distances has incorrect number of rows");
    CUVS_EXPECTS(d.shape[1] == idx.degree, "Output distances has incorrect number of cols");
    copy_async(res, dl_data(d), idx.dist.data(), idx.dist.bytes());
    sync(res);
  });
}
}  // extern "C"
