// Batched exact top-k for wave64 — replaces raft::matrix::select_k, which the reference forwards to
// (cpp/src/selection/select_k.cuh:13-57; RAFT sources are not vendored). One 256-thread workgroup per
// row: three LDS-histogram radix passes (11/11/10 bits of an order-preserving key) find the exact k-th
// key, one collect pass gathers the winners, a bitonic sort in LDS orders them by (value, index).
// HBM/L2-bound: the row is streamed 4 times; everything else lives in LDS.
#include "ops.hpp"
#include "device_utils.hpp"

#include <cfloat>

namespace cuvs_amd {

namespace {

constexpr int kSelThreads = 256;
constexpr int kBins       = 2048;

template <typename InIdxT, bool HAS_IDX>
__device__ inline int64_t src_index(const InIdxT* in_idx_row, int64_t i, int64_t idx_offset)
{
  if constexpr (HAS_IDX) {
    return (int64_t)in_idx_row[i];
  } else {
    return i + idx_offset;
  }
}

template <typename InIdxT, typename OutIdxT, bool HAS_IDX>
__global__ __launch_bounds__(kSelThreads) void select_k_radix_kernel(const float* __restrict__ in,
                                                                     const InIdxT* __restrict__ in_idx,
                                                                     int64_t len,
                                                                     int64_t in_ld,
                                                                     int k,
                                                                     int kp2,
                                                                     float* __restrict__ out_val,
                                                                     OutIdxT* __restrict__ out_idx,
                                                                     bool select_min,
                                                                     int64_t idx_offset,
                                                                     int64_t out_ld,
                                                                     int64_t out_col_offset,
                                                                     char* __restrict__ big_k_scratch,
                                                                     const uint32_t* __restrict__ run_if,
                                                                     const uint8_t* __restrict__ done)
{
  if (run_if != nullptr && *run_if == 0u) return;  // device-side guard: a fallback pass that is not needed
  if (done != nullptr && done[blockIdx.x] != 0u) return;  // this row was served by select_k_minima_kernel
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* hist      = reinterpret_cast<int*>(smem_raw);            // kBins
  int* scan      = hist + kBins;                                 // 32
  int* ctrl      = scan + 32;                                    // 8
  // the k winners live in LDS up to 8192 of them; beyond that in a per-row slice of a global scratch buffer (the
  // bitonic sort then runs on global memory - slow, but k in the tens of thousands is not a hot path)
  char* win      = big_k_scratch ? big_k_scratch + (size_t)blockIdx.x * kp2 * 12 : reinterpret_cast<char*>(ctrl + 8);
  int64_t* s_idx = reinterpret_cast<int64_t*>(win);              // kp2 (8-byte aligned: 8192+160)
  uint32_t* s_key = reinterpret_cast<uint32_t*>(s_idx + kp2);    // kp2

  const int tid       = threadIdx.x;
  const int64_t row   = blockIdx.x;
  const float* r      = in + row * in_ld;
  const InIdxT* ridx  = HAS_IDX ? in_idx + row * in_ld : nullptr;
  const uint32_t flip = select_min ? 0u : 0xffffffffu;

  const int k_eff = (int)((int64_t)k < len ? (int64_t)k : len);

  uint32_t prefix = 0, mask = 0;
  int need = k_eff, cnt_eq = 0;

  if (k_eff > 0 && (int64_t)k_eff < len) {
    const int shifts[3] = {21, 10, 0};
    const int nbits[3]  = {11, 11, 10};
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
      const int shift   = shifts[p];
      const uint32_t dm = (1u << nbits[p]) - 1u;
      for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
      __syncthreads();
      for (int64_t i = tid; i < len; i += kSelThreads) {
        uint32_t key = float_to_key(r[i]) ^ flip;
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & dm], 1);
      }
      __syncthreads();
      int local[8];
      int s = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        local[b] = hist[tid * 8 + b];
        s += local[b];
      }
      int total;
      int excl = block_exclusive_scan(s, scan, &total);
      if (need > excl && need <= excl + s) {
        int c = excl;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (need > c && need <= c + local[b]) {
            ctrl[0] = tid * 8 + b;
            ctrl[1] = need - c;
            ctrl[2] = local[b];
          }
          c += local[b];
        }
      }
      __syncthreads();
      int bucket = ctrl[0];
      need       = ctrl[1];
      cnt_eq     = ctrl[2];
      prefix |= ((uint32_t)bucket) << shift;
      mask |= dm << shift;
      __syncthreads();
    }
  } else {
    // take everything: kth key = max, all elements are "less or equal"
    prefix = 0xffffffffu;
    need   = 0;
    cnt_eq = 0;
  }

  const bool take_all  = !(k_eff > 0 && (int64_t)k_eff < len);
  const uint32_t kth   = prefix;
  const int n_less     = take_all ? k_eff : (k_eff - need);

  if (tid == 0) { ctrl[3] = 0; ctrl[4] = 0; }
  for (int j = tid; j < kp2; j += kSelThreads) {
    s_key[j] = 0xffffffffu;
    s_idx[j] = INT64_MAX;
  }
  __syncthreads();

  if (take_all) {
    for (int64_t i = tid; i < len; i += kSelThreads) {
      s_key[i] = float_to_key(r[i]) ^ flip;
      s_idx[i] = src_index<InIdxT, HAS_IDX>(ridx, i, idx_offset);
    }
  } else if (cnt_eq == need) {
    for (int64_t i = tid; i < len; i += kSelThreads) {
      uint32_t key = float_to_key(r[i]) ^ flip;
      if (key < kth) {
        int pos    = atomicAdd(&ctrl[3], 1);
        s_key[pos] = key;
        s_idx[pos] = src_index<InIdxT, HAS_IDX>(ridx, i, idx_offset);
      } else if (key == kth) {
        int pos    = n_less + atomicAdd(&ctrl[4], 1);
        s_key[pos] = key;
        s_idx[pos] = src_index<InIdxT, HAS_IDX>(ridx, i, idx_offset);
      }
    }
  } else {
    // more elements equal to the k-th key than we may take: earliest columns win
    for (int64_t i = tid; i < len; i += kSelThreads) {
      uint32_t key = float_to_key(r[i]) ^ flip;
      if (key < kth) {
        int pos    = atomicAdd(&ctrl[3], 1);
        s_key[pos] = key;
        s_idx[pos] = src_index<InIdxT, HAS_IDX>(ridx, i, idx_offset);
      }
    }
    int base = 0;
    for (int64_t start = 0; start < len && base < need; start += kSelThreads) {
      int64_t i = start + tid;
      int flag  = 0;
      if (i < len) flag = ((float_to_key(r[i]) ^ flip) == kth) ? 1 : 0;
      int total;
      int excl = block_exclusive_scan(flag, scan, &total);
      if (flag && base + excl < need) {
        int pos    = n_less + base + excl;
        s_key[pos] = kth;
        s_idx[pos] = src_index<InIdxT, HAS_IDX>(ridx, i, idx_offset);
      }
      base += total;
    }
  }
  __syncthreads();

  block_bitonic_sort<int64_t>(s_key, s_idx, kp2);

  float* ov   = out_val + row * out_ld + out_col_offset;
  OutIdxT* oi = out_idx + row * out_ld + out_col_offset;
  for (int j = tid; j < k; j += kSelThreads) {
    if (j < k_eff) {
      ov[j] = key_to_float(s_key[j] ^ flip);
      oi[j] = (OutIdxT)s_idx[j];
    } else {
      ov[j] = select_min ? FLT_MAX : -FLT_MAX;
      oi[j] = (OutIdxT)(-1);
    }
  }
}


// ------------------------------------------------------------------ short rows, small k: ONE read of the row
// The radix kernel above streams a row four times (three histogram passes + the collect pass): at 10k x 16384 -> 128 (the
// coarse search of the IVF-PQ bench shape) that is 2.6 GB of reads for 655 MB of distances, 0.71 of the 7.4 ms of a search.
// For rows of at most 16384 elements the row fits the REGISTERS of a 256-thread workgroup (64 values per thread, all loads
// of a thread in flight at once): every thread keeps the minimum of its elements (of two groups of them for k > 128); the
// k-th smallest of the 256 (512) group minima bounds the row's k-th smallest key from above (k different elements are at
// or below it) and is close to it (k = 128 of 16384: about the 180th smallest); the elements at or below that bound are collected in LDS straight from the
// registers, sorted by (key, position) - the order and the tie rule of the radix kernel - and the first k written.
// A row that collects more than the buffer holds (masses of equal keys) is left to the radix kernel (`done` stays 0).
constexpr int kMinCap = 2048;  // candidates at or below the bound

template <typename OutIdxT, int NV, int G>  // NV: 16-byte vectors per thread (row length <= 1024 NV); G: groups per thread
__global__ __launch_bounds__(kSelThreads) void select_k_minima_kernel(const float* __restrict__ in, int64_t len, int64_t in_ld, int k,
                                                                      float* __restrict__ out_val, OutIdxT* __restrict__ out_idx,
                                                                      bool select_min, int64_t idx_offset, int64_t out_ld,
                                                                      int64_t out_col_offset, uint8_t* __restrict__ done,
                                                                      const uint32_t* __restrict__ run_if)
{
  if (run_if != nullptr && *run_if == 0u) return;
  constexpr int NGR = kSelThreads * G;
  __shared__ __attribute__((aligned(16))) uint32_t tk[NGR];
  __shared__ unsigned long long cand[kMinCap];
  __shared__ uint32_t ctrl[2];
  const int tid        = threadIdx.x;
  const int64_t row    = blockIdx.x;
  const float4* r4     = reinterpret_cast<const float4*>(in + row * in_ld);
  const uint32_t flip  = select_min ? 0u : 0xffffffffu;
  const int n_vec      = (int)(len >> 2);
  // ---- the row into registers: vector m of this thread = elements [4 (tid + 256 m), + 4); all loads issued before any use
  float4 v[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    const int j = tid + kSelThreads * m;
    v[m] = j < n_vec ? r4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid == 0) { ctrl[0] = 0u; ctrl[1] = 0u; }
  uint32_t gmin[G];  // group G tid + (m % G)
#pragma unroll
  for (int w = 0; w < G; ++w) gmin[w] = 0xffffffffu;
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    if (tid + kSelThreads * m < n_vec) {
      const uint32_t k0 = float_to_key(v[m].x) ^ flip, k1 = float_to_key(v[m].y) ^ flip, k2 = float_to_key(v[m].z) ^ flip,
                     k3 = float_to_key(v[m].w) ^ flip;
      gmin[m % G] = min(gmin[m % G], min(min(k0, k1), min(k2, k3)));
    }
  }
#pragma unroll
  for (int w = 0; w < G; ++w) tk[G * tid + w] = gmin[w];
  __syncthreads();
  // ---- the k-th smallest group minimum (ties by group number), by rank counting over broadcast reads
  {
    const uint4* tk4 = reinterpret_cast<const uint4*>(tk);
    int rk[G];
#pragma unroll
    for (int w = 0; w < G; ++w) rk[w] = 0;
#pragma unroll 4
    for (int j = 0; j < NGR / 4; ++j) {
      const uint4 o = tk4[j];
      const uint32_t ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int w = 0; w < G; ++w) rk[w] += (ov[e] < gmin[w] || (ov[e] == gmin[w] && 4 * j + e < G * tid + w)) ? 1 : 0;
    }
#pragma unroll
    for (int w = 0; w < G; ++w)
      if (rk[w] == k - 1) ctrl[1] = gmin[w];
  }
  __syncthreads();
  const uint32_t bound = ctrl[1];
  // ---- the elements at or below the bound, from the registers: (key, position) packed, position = column in the row
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    const int j = tid + kSelThreads * m;
    if (j < n_vec) {
      const float e[4] = {v[m].x, v[m].y, v[m].z, v[m].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t key = float_to_key(e[c]) ^ flip;
        if (key <= bound) {
          const uint32_t pos = atomicAdd(&ctrl[0], 1u);
          if (pos < (uint32_t)kMinCap) cand[pos] = ((unsigned long long)key << 32) | (uint32_t)(4 * j + c);
        }
      }
    }
  }
  __syncthreads();
  const uint32_t cnt = ctrl[0];
  if (cnt > (uint32_t)kMinCap || cnt < (uint32_t)k) {  // (workgroup-uniform) masses of ties at the bound: the radix kernel
    if (tid == 0) done[row] = 0u;
    return;
  }
  int P = 1;
  while (P < (int)cnt) P <<= 1;
  for (int t = (int)cnt + tid; t < P; t += kSelThreads) cand[t] = ~0ull;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (P >> 1); t += kSelThreads) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = cand[lo], b = cand[hi];
        if ((a > b) == up) { cand[lo] = b; cand[hi] = a; }
      }
    }
  }
  __syncthreads();
  float* ov   = out_val + row * out_ld + out_col_offset;
  OutIdxT* oi = out_idx + row * out_ld + out_col_offset;
  for (int j = tid; j < k; j += kSelThreads) {
    const unsigned long long c = cand[j];
    ov[j] = key_to_float((uint32_t)(c >> 32) ^ flip);
    oi[j] = (OutIdxT)((int64_t)(uint32_t)c + idx_offset);
  }
  if (tid == 0) done[row] = 1u;
}


// ------------------------------------------------------------------ rows in GROUPED layout (ops.hpp: pairwise_distance_grouped)
// The distance kernel has already reduced every 16 consecutive positions of a row to their best key. ONE WAVE per row (four
// rows per workgroup, no workgroup barrier): the group keys are cut into 256 G classes - class (m % G, lane, c) holds
// component c of the lane's m-th 16-byte key vector - the k-th best class minimum bounds the row's k-th best key from above
// (k different elements are at or below it; found by a bitwise radix select over the 4 G class minima a lane holds: 32
// ballots per register instead of 256 G compares per thread), only groups at or below the bound are read - 64 bytes each -
// and their elements at or below the bound sorted by (key, column), the order and the tie rule of select_k.
constexpr int kGrpCap = 512;  // candidates at or below the bound, per row

template <int G>
__global__ __launch_bounds__(kSelThreads) void select_k_grouped_kernel(const float* __restrict__ in, int64_t in_ld,
                                                                       const uint32_t* __restrict__ gkeys, int64_t ldg, int n_groups,
                                                                       int64_t rows, int64_t len, int k, float* __restrict__ out_val,
                                                                       uint32_t* __restrict__ out_idx, bool select_min,
                                                                       uint8_t* __restrict__ done)
{
  __shared__ unsigned long long cand_all[kSelThreads / 64][kGrpCap];
  const int lane    = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kSelThreads / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;  // wave-uniform
  unsigned long long* cand = cand_all[threadIdx.x >> 6];
  const uint32_t flip = select_min ? 0u : 0xffffffffu;
  const uint4* gk4    = reinterpret_cast<const uint4*>(gkeys + row * ldg);
  const float* r      = in + row * in_ld;
  const int n_vec     = n_groups >> 2;  // (n_groups is a multiple of 8)
  // ---- class minima
  uint32_t cm[G][4];
#pragma unroll
  for (int w = 0; w < G; ++w)
#pragma unroll
    for (int c = 0; c < 4; ++c) cm[w][c] = 0xffffffffu;
  for (int j = lane, m = 0; j < n_vec; j += 64, ++m) {
    const uint4 v = gk4[j];
#pragma unroll
    for (int w = 0; w < G; ++w)
      if (m % G == w) { cm[w][0] = min(cm[w][0], v.x); cm[w][1] = min(cm[w][1], v.y); cm[w][2] = min(cm[w][2], v.z); cm[w][3] = min(cm[w][3], v.w); }
  }
  // ---- value of the k-th smallest class minimum: one bit per step, counts by ballot
  uint32_t prefix = 0u, mask = 0u;
  int need = k;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t b = 1u << bit;
    int zeros = 0;
#pragma unroll
    for (int w = 0; w < G; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) zeros += __popcll(__ballot(((cm[w][c] & mask) == prefix) && (cm[w][c] & b) == 0u));
    if (need > zeros) { need -= zeros; prefix |= b; }
    mask |= b;
  }
  const uint32_t bound = prefix;
  // ---- the elements at or below the bound in the groups at or below it; positions by ballot + prefix count
  uint32_t cnt = 0u;
  for (int j0 = 0; j0 < n_vec; j0 += 64) {
    const int j   = j0 + lane;
    const uint4 v = j < n_vec ? gk4[j] : make_uint4(~0u, ~0u, ~0u, ~0u);  // (served from L1 / L2: the same words as above)
    const uint32_t gv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool hit = gv[c] <= bound && j < n_vec;
      if (__ballot(hit) == 0ull) continue;  // wave-uniform
      float4 e4[4];
      if (hit) {
        const float4* p4 = reinterpret_cast<const float4*>(r + (int64_t)(4 * j + c) * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) e4[t] = p4[t];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float e[4] = {e4[t].x, e4[t].y, e4[t].z, e4[t].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t key = hit ? (float_to_key(e[u]) ^ flip) : 0xffffffffu;
          const uint32_t col = grouped_col((uint32_t)(4 * j + c) * 16u + (uint32_t)(4 * t + u));
          const bool take    = hit && key <= bound && (int64_t)col < len;
          const unsigned long long mk = __ballot(take);
          if (take) {
            const uint32_t pos = cnt + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
            if (pos < (uint32_t)kGrpCap) cand[pos] = ((unsigned long long)key << 32) | col;
          }
          cnt += (uint32_t)__popcll(mk);
        }
      }
    }
  }
  if (cnt > (uint32_t)kGrpCap || cnt < (uint32_t)k) {  // (wave-uniform) masses of ties at the bound: the radix kernel
    if (lane == 0) done[row] = 0u;
    return;
  }
  int P = 64;
  while (P < (int)cnt) P <<= 1;
  for (int t = (int)cnt + lane; t < P; t += 64) cand[t] = ~0ull;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < (P >> 1); t += 64) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = cand[lo], b = cand[hi];
        if ((a > b) == up) { cand[lo] = b; cand[hi] = a; }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < k; j += 64) {
    const unsigned long long c = cand[j];
    out_val[row * k + j] = key_to_float((uint32_t)(c >> 32) ^ flip);
    out_idx[row * k + j] = (uint32_t)c;
  }
  if (lane == 0) done[row] = 1u;
}

// rows the kernel above left (done == 0) back to column order, in place: the permutation stays inside 128-column tiles
__global__ __launch_bounds__(128) void ungroup_rows_kernel(float* __restrict__ in, int64_t in_ld, int n_tiles, const uint8_t* __restrict__ done)
{
  const int64_t row = blockIdx.x;
  if (done[row] != 0u) return;
  float* r = in + row * in_ld;
  for (int t = 0; t < n_tiles; ++t) {
    const float v = r[t * 128 + threadIdx.x];
    __syncthreads();  // (every position of the tile is read before any is overwritten)
    r[t * 128 + (grouped_col((uint32_t)threadIdx.x))] = v;
    __syncthreads();
  }
}

}  // namespace

bool select_k_grouped_ok(int64_t len, int k) { return len >= 4096 && k >= 8 && k <= 256 && (int64_t)k * 8 <= len && len <= (int64_t(1) << 24); }

void select_k_grouped(resources& res, float* in, int64_t in_ld, const uint32_t* gkeys, int64_t ldg, int64_t rows, int64_t len, int k,
                      float* out_val, uint32_t* out_idx, bool select_min)
{
  if (rows == 0) return;
  CUVS_EXPECTS(select_k_grouped_ok(len, k) && rows < (int64_t(1) << 31), "select_k_grouped: unsupported shape");
  const int n_groups = (int)(round_up(len, 128) / 16);
  dev_buf<uint8_t> done(res, (size_t)rows);
  const unsigned wgs = (unsigned)((rows + kSelThreads / 64 - 1) / (kSelThreads / 64));  // one wave per row
  if (k <= 128)
    hipLaunchKernelGGL(select_k_grouped_kernel<1>, dim3(wgs), dim3(kSelThreads), 0, res.stream, in, in_ld, gkeys, ldg, n_groups,
                       rows, len, k, out_val, out_idx, select_min, done.data());
  else
    hipLaunchKernelGGL(select_k_grouped_kernel<2>, dim3(wgs), dim3(kSelThreads), 0, res.stream, in, in_ld, gkeys, ldg, n_groups,
                       rows, len, k, out_val, out_idx, select_min, done.data());
  // the rows it left (none, as a rule): back to column order, then the radix kernel on those rows only
  hipLaunchKernelGGL(ungroup_rows_kernel, dim3((unsigned)rows), dim3(128), 0, res.stream, in, in_ld, n_groups / 8, done.data());
  const int kp2     = next_pow2(k);
  const size_t smem = (kBins + 32 + 8) * sizeof(int) + (size_t)kp2 * (sizeof(int64_t) + sizeof(uint32_t));
  auto kern = select_k_radix_kernel<uint32_t, uint32_t, false>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(kSelThreads), smem, res.stream, (const float*)in, (const uint32_t*)nullptr, len, in_ld, k, kp2,
                     out_val, out_idx, select_min, (int64_t)0, (int64_t)k, (int64_t)0, (char*)nullptr, (const uint32_t*)nullptr,
                     (const uint8_t*)done.data());
  HIP_TRY(hipGetLastError());
}

template <typename InIdxT, typename OutIdxT>
void select_k(resources& res, const float* in, const InIdxT* in_idx, int64_t rows, int64_t len,
              int64_t in_ld, int k, float* out_val, OutIdxT* out_idx, bool select_min,
              int64_t idx_offset, int64_t out_ld, int64_t out_col_offset, const uint32_t* run_if)
{
  if (rows == 0 || k == 0) return;
  CUVS_EXPECTS(k > 0 && k <= (1 << 24), "select_k: k must be in [1, 2^24], got %d", k);
  CUVS_EXPECTS(rows < (int64_t(1) << 31), "select_k: too many rows");
  if (out_ld < 0) out_ld = k;
  const int kp2      = next_pow2(k);
  const bool in_lds  = kp2 <= 8192;
  const size_t smem  = (kBins + 32 + 8) * sizeof(int) + (in_lds ? (size_t)kp2 * (sizeof(int64_t) + sizeof(uint32_t)) : 0);
  // rows per launch when the winners live in global scratch: bounded by the workspace budget
  const int64_t rows_per = in_lds ? rows : std::max<int64_t>(1, (int64_t)(res.workspace_limit / ((size_t)kp2 * 12)));
  dev_buf<char> scratch(res, in_lds ? 0 : (size_t)std::min(rows, rows_per) * kp2 * 12);
  // short rows and small k without source indices (the coarse searches of the IVF indexes, the per-tile selects of brute
  // force): the one-read kernel first; the radix kernel below then only runs the rows it left (none, as a rule)
  const bool minima = in_idx == nullptr && k >= 8 && k <= 256 && len >= 4096 && len <= 16384 && len % 4 == 0 && in_ld % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (int64_t)k * 8 <= len;
  dev_buf<uint8_t> done(res, minima ? (size_t)rows : 0);
  if (minima) {  // (the kernel writes every row's flag - 0 or 1 - unless the whole launch is guarded off, and then so is the next)
    auto launch_min = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(kSelThreads), 0, res.stream, in, len, in_ld, k, out_val, out_idx, select_min,
                         idx_offset, out_ld, out_col_offset, done.data(), run_if);
    };
    if (k <= 128) {  // one group per thread: the rank counting over 256 minima is the cheaper half of the kernel
      if (len <= 4096)       launch_min(select_k_minima_kernel<OutIdxT, 4, 1>);
      else if (len <= 8192)  launch_min(select_k_minima_kernel<OutIdxT, 8, 1>);
      else                   launch_min(select_k_minima_kernel<OutIdxT, 16, 1>);
    } else {
      if (len <= 4096)       launch_min(select_k_minima_kernel<OutIdxT, 4, 2>);
      else if (len <= 8192)  launch_min(select_k_minima_kernel<OutIdxT, 8, 2>);
      else                   launch_min(select_k_minima_kernel<OutIdxT, 16, 2>);
    }
  }
  for (int64_t r0 = 0; r0 < rows; r0 += rows_per) {
    dim3 grid((unsigned)std::min(rows_per, rows - r0)), block(kSelThreads);
    const uint8_t* done_r = minima ? done.data() + r0 : nullptr;
    const float* in_r      = in + r0 * in_ld;
    const InIdxT* in_idx_r = in_idx ? in_idx + r0 * in_ld : nullptr;
    float* out_val_r       = out_val + r0 * out_ld;
    OutIdxT* out_idx_r     = out_idx + r0 * out_ld;
    if (in_idx != nullptr) {
      auto kern = select_k_radix_kernel<InIdxT, OutIdxT, true>;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(kern, grid, block, smem, res.stream, in_r, in_idx_r, len, in_ld, k, kp2, out_val_r, out_idx_r,
                         select_min, idx_offset, out_ld, out_col_offset, scratch.data(), run_if, done_r);
    } else {
      auto kern = select_k_radix_kernel<InIdxT, OutIdxT, false>;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(kern, grid, block, smem, res.stream, in_r, in_idx_r, len, in_ld, k, kp2, out_val_r, out_idx_r,
                         select_min, idx_offset, out_ld, out_col_offset, scratch.data(), run_if, done_r);
    }
  }
  HIP_TRY(hipGetLastError());
}

#define INST(I, O)                                                                                   \
  template void select_k<I, O>(resources&, const float*, const I*, int64_t, int64_t, int64_t, int,   \
                               float*, O*, bool, int64_t, int64_t, int64_t, const uint32_t*);
INST(uint32_t, uint32_t)
INST(uint32_t, int64_t)
INST(int64_t, int64_t)
#undef INST

}  // namespace cuvs_amd

// Test hook (not part of the reference ABI): exact top-k of a device matrix, used by tests/ to pin the
// kernel against oracle/oracle.c `oracle_select_k`.
extern "C" __attribute__((visibility("default"))) int cuvsAmdSelectK(uintptr_t res, const float* in,
                                                                      const int64_t* in_idx, int64_t rows,
                                                                      int64_t len, int k, float* out_val,
                                                                      int64_t* out_idx, int select_min)
{
  using namespace cuvs_amd;
  return translate_exceptions([=] {
    select_k<int64_t, int64_t>(*as_res(res), in, in_idx, rows, len, len, k, out_val, out_idx,
                               select_min != 0);
  });
}
