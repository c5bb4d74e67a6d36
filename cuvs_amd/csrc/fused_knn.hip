// Fused distance + top-k for brute-force search (k <= 64): the distance tile never leaves the CU.
// Reference: fusedL2kNN (cpp/src/neighbors/detail/fused_l2_knn.cuh:186-330: SIMT tile + FAISS WarpSelect with a
// global-mutex merge across blocks). MI355X design: the same 128x128x16 fp32-MFMA main loop as distance.hip;
// every workgroup owns 128 query rows x one column split and keeps, per row, a sorted top list in LDS plus its
// k-th key as an admission bound. The epilogue compares the 64 accumulator values of each lane with the row
// bounds; the few that pass are appended to an LDS queue (one LDS atomic per wave) and inserted by the wave that
// owns the row (row % 4) with a DPP wave_shr shift — lists are exact under the (value, index) order. The first
// (cold) tile is fed in 16-column slices so that bounds tighten before most of it is looked at. Splits are
// merged by select_k. Distances/ids are bit-identical to the unfused path and to oracle.brute_force_knn.
#include "distance_tile.hpp"

#include <cfloat>

namespace cuvs_amd {
namespace {

constexpr int kQueueCap = 1024;

struct fused_args {
  const float* qn;
  const float* xn;
  int metric;
  float clamp_eps;
  int k, kp;                 // kp = list length in LDS (power of two, 16..64)
  int64_t cols_per_split;    // multiple of BN
  float* out_d;              // [m, n_splits, k]
  int64_t* out_i;
  int n_splits;
};

template <typename TQ, typename TX, bool VEC>
__global__ __launch_bounds__(256) void fused_knn_kernel(const TQ* __restrict__ q, int64_t m, int64_t ldq,
                                                        const TX* __restrict__ x, int64_t n, int64_t ldx,
                                                        int64_t dim, fused_args fa)
{
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDT];
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  // dynamic LDS: per-row lists, bounds, candidate queue
  uint32_t* top_k  = reinterpret_cast<uint32_t*>(dyn);                 // [BM][kp] order-preserving keys
  uint32_t* top_i  = top_k + BM * fa.kp;                                // [BM][kp] column ids
  uint32_t* bound  = top_i + BM * fa.kp;                                // [BM]
  uint32_t* q_key  = bound + BM;                                        // [kQueueCap]
  uint32_t* q_col  = q_key + kQueueCap;
  uint32_t* q_row  = q_col + kQueueCap;
  int* q_cnt       = reinterpret_cast<int*>(q_row + kQueueCap);         // [0] count, [1] "another round" flag

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm   = wave >> 1;
  const int wn   = wave & 1;
  const int l15  = lane & 15;
  const int lg   = lane >> 4;
  const bool neg = fa.metric == M_InnerProduct;  // larger is better: order by the negated value

  const int64_t row0      = (int64_t)blockIdx.y * BM;
  const int64_t col_begin = (int64_t)blockIdx.x * fa.cols_per_split;
  const int64_t col_end   = min(n, col_begin + fa.cols_per_split);
  const int nkt           = (int)((dim + BK - 1) / BK);
  const int sr0 = tid >> 2, sc0 = tid & 3;
  const int sr1 = (tid + 256) >> 2, sc1 = tid & 3;

  for (int t = tid; t < BM * fa.kp; t += 256) { top_k[t] = 0xffffffffu; top_i[t] = 0xffffffffu; }
  for (int t = tid; t < BM; t += 256) bound[t] = 0xffffffffu;
  if (tid < 2) q_cnt[tid] = 0;
  __syncthreads();

  float qnv[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    int64_t row = row0 + wm * 64 + (t >> 2) * 16 + lg * 4 + (t & 3);
    qnv[t]      = (fa.qn != nullptr && row < m) ? fa.qn[row] : 0.f;
  }

  for (int64_t col0 = col_begin; col0 < col_end; col0 += BN) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ra0[4], ra1[4], rb0[4], rb1[4];
    load4<TQ, VEC>(q, row0 + sr0, m, ldq, sc0 * 4, dim, ra0);
    load4<TQ, VEC>(q, row0 + sr1, m, ldq, sc1 * 4, dim, ra1);
    load4<TX, VEC>(x, col0 + sr0, n, ldx, sc0 * 4, dim, rb0);
    load4<TX, VEC>(x, col0 + sr1, n, ldx, sc1 * 4, dim, rb1);
    __syncthreads();
    stage_store(As[0], sr0, sc0, ra0);
    stage_store(As[0], sr1, sc1, ra1);
    stage_store(Bs[0], sr0, sc0, rb0);
    stage_store(Bs[0], sr1, sc1, rb1);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) {
        int64_t k0 = (int64_t)(kt + 1) * BK;
        load4<TQ, VEC>(q, row0 + sr0, m, ldq, k0 + sc0 * 4, dim, ra0);
        load4<TQ, VEC>(q, row0 + sr1, m, ldq, k0 + sc1 * 4, dim, ra1);
        load4<TX, VEC>(x, col0 + sr0, n, ldx, k0 + sc0 * 4, dim, rb0);
        load4<TX, VEC>(x, col0 + sr1, n, ldx, k0 + sc1 * 4, dim, rb1);
      }
      const float* A = As[buf];
      const float* B = Bs[buf];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k   = 4 * c + lg;
        const int swz = c << 3;
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = A[k * LDT + ((wm * 64 + i * 16 + l15) ^ swz)];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = B[k * LDT + ((wn * 64 + j * 16 + l15) ^ swz)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < nkt) {
        stage_store(As[buf ^ 1], sr0, sc0, ra0);
        stage_store(As[buf ^ 1], sr1, sc1, ra1);
        stage_store(Bs[buf ^ 1], sr0, sc0, rb0);
        stage_store(Bs[buf ^ 1], sr1, sc1, rb1);
      }
      __syncthreads();
    }

    // ---- epilogue: admission test against the row bounds, queue, insert
    const bool cold = (col0 == col_begin);
    for (int js = 0; js < (cold ? 4 : 1); ++js) {  // cold tile: one 16-column slice at a time
      const int j_lo = cold ? js : 0, j_hi = cold ? js + 1 : 4;
      // candidate (j, t) <-> bit j * 16 + t of `pend`; keys are recomputed from the accumulators when needed
      float xnv[4];
      unsigned long long pend = 0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = col0 + wn * 64 + j * 16 + l15;
        xnv[j]            = (fa.xn != nullptr && col < col_end) ? fa.xn[col] : 0.f;
        if (j >= j_lo && j < j_hi && col < col_end) {
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int64_t row = row0 + wm * 64 + (t >> 2) * 16 + lg * 4 + (t & 3);
            if (row < m) pend |= 1ull << (j * 16 + t);
          }
        }
      }
      auto key_of = [&](int j, int t) -> uint32_t {
        float d = finish_distance(acc[t >> 2][j][t & 3], qnv[t], xnv[j], fa.metric, fa.clamp_eps);
        return float_to_key(neg ? -d : d);
      };
      for (;;) {
        // admission against the current bounds; every wave submits at most kQueueCap / 4 candidates per round,
        // so the four reservations always fit and no round can stall
        int mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const unsigned long long bit = 1ull << (j * 16 + t);
            if (pend & bit) {
              const int r = wm * 64 + (t >> 2) * 16 + lg * 4 + (t & 3);
              if (key_of(j, t) > bound[r]) pend &= ~bit;  // can never enter this row's list
              else ++mine;
            }
          }
        const int incl   = wave_inclusive_scan(mine);
        const int total  = __shfl(incl, 63, kWave);
        const int submit = min(total, kQueueCap / 4);
        if (submit > 0) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&q_cnt[0], submit);
          base     = __shfl(base, 0, kWave);
          int rank = incl - mine;  // rank of this lane's first candidate within the wave
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              const unsigned long long bit = 1ull << (j * 16 + t);
              if (pend & bit) {
                if (rank < submit) {
                  q_key[base + rank] = key_of(j, t);
                  q_col[base + rank] = (uint32_t)(col0 + wn * 64 + j * 16 + l15);
                  q_row[base + rank] = (uint32_t)(wm * 64 + (t >> 2) * 16 + lg * 4 + (t & 3));
                  pend &= ~bit;
                }
                ++rank;
              }
            }
        }
        if (total > submit && lane == 0) q_cnt[1] = 1;  // this wave has candidates left for another round
        __syncthreads();
        const bool more = q_cnt[1] != 0;
        // ---- drain: wave w inserts the entries of rows with row % 4 == w
        {
          const int cnt = min(q_cnt[0], kQueueCap);
          for (int b0 = 0; b0 < cnt; b0 += 64) {
            const int e         = b0 + lane;
            const bool have     = e < cnt && ((q_row[e < cnt ? e : 0] & 3u) == (uint32_t)wave);
            const uint32_t ek   = have ? q_key[e] : 0u;
            const uint32_t ec   = have ? q_col[e] : 0u;
            const uint32_t er   = have ? q_row[e] : 0u;
            unsigned long long mm = __ballot(have);
            while (mm != 0ull) {
              const int src = (int)__ffsll((long long)mm) - 1;
              mm &= mm - 1ull;
              const uint32_t ck = __builtin_amdgcn_readlane(ek, src);
              const uint32_t cc = __builtin_amdgcn_readlane(ec, src);
              const uint32_t cr = __builtin_amdgcn_readlane(er, src);
              uint32_t lk = lane < fa.kp ? top_k[cr * fa.kp + lane] : 0xffffffffu;
              uint32_t li = lane < fa.kp ? top_i[cr * fa.kp + lane] : 0xffffffffu;
              const uint32_t wk = __builtin_amdgcn_readlane(lk, fa.k - 1);
              const uint32_t wi = __builtin_amdgcn_readlane(li, fa.k - 1);
              if (ck < wk || (ck == wk && cc < wi)) {
                const bool le  = (lk < ck) || (lk == ck && li <= cc);
                const int pos  = __popcll(__ballot(le));
                const uint32_t up_k = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lk, 0x138, 0xf, 0xf, false);
                const uint32_t up_i = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)li, 0x138, 0xf, 0xf, false);
                if (lane > pos) { lk = up_k; li = up_i; }
                else if (lane == pos) { lk = ck; li = cc; }
                if (lane < fa.kp) { top_k[cr * fa.kp + lane] = lk; top_i[cr * fa.kp + lane] = li; }
                if (lane == fa.k - 1) bound[cr] = lk;
              }
            }
          }
        }
        __syncthreads();
        if (tid == 0) { q_cnt[0] = 0; q_cnt[1] = 0; }
        __syncthreads();
        if (!more) break;
      }
    }
  }

  // ---- per-row result of this split
  for (int t = tid; t < BM * fa.k; t += 256) {
    const int r = t / fa.k, j = t % fa.k;
    const int64_t row = row0 + r;
    if (row >= m) continue;
    const uint32_t key = top_k[r * fa.kp + j];
    const uint32_t col = top_i[r * fa.kp + j];
    const size_t o     = ((size_t)row * fa.n_splits + blockIdx.x) * fa.k + j;
    if (col == 0xffffffffu) {
      fa.out_d[o] = neg ? -FLT_MAX : FLT_MAX;
      fa.out_i[o] = -1;
    } else {
      float d     = key_to_float(key);
      fa.out_d[o] = neg ? -d : d;
      fa.out_i[o] = (int64_t)col;
    }
  }
}

}  // namespace

// out_d / out_i: [m, k]; returns false when the shape is outside the fused path (caller uses the tiled path)
template <typename TQ, typename TX>
bool fused_knn(resources& res, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n, int64_t ldx,
               int64_t dim, const float* qn, const float* xn, int metric, int k, float* out_d, int64_t* out_i)
{
  if (k > 64 || k < 1 || n >= (int64_t(1) << 32) - 1 || m == 0 || n == 0) return false;
  const int64_t row_blocks = (m + BM - 1) / BM;
  if (row_blocks > 65535) return false;
  const int64_t col_tiles = (n + BN - 1) / BN;
  int64_t n_splits = std::max<int64_t>(1, std::min<int64_t>(col_tiles, (2 * (int64_t)res.num_cus + row_blocks - 1) / row_blocks));
  int64_t tiles_per_split = (col_tiles + n_splits - 1) / n_splits;
  n_splits                = (col_tiles + tiles_per_split - 1) / tiles_per_split;
  fused_args fa;
  fa.qn = qn; fa.xn = xn; fa.metric = metric; fa.clamp_eps = sizeof(TX) == 2 ? 1e-3f : 1e-6f;
  fa.k = k; fa.kp = std::max(16, next_pow2(k));
  fa.cols_per_split = tiles_per_split * BN;
  fa.n_splits       = (int)n_splits;
  dev_buf<float> pd;
  dev_buf<int64_t> pi;
  if (n_splits > 1) {
    pd = dev_buf<float>(res, (size_t)m * n_splits * k);
    pi = dev_buf<int64_t>(res, (size_t)m * n_splits * k);
    fa.out_d = pd.data(); fa.out_i = pi.data();
  } else {
    fa.out_d = out_d; fa.out_i = out_i;
  }
  size_t dyn = ((size_t)BM * fa.kp * 2 + BM + 3 * kQueueCap + 4) * sizeof(uint32_t);
  dim3 grid((unsigned)n_splits, (unsigned)row_blocks);
  const bool vec = vec_ok(q, ldq, dim) && vec_ok(x, ldx, dim);
  profile_begin(res, "fused_knn_kernel");
  if (vec) {
    auto kern = fused_knn_kernel<TQ, TX, true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, grid, dim3(256), dyn, res.stream, q, m, ldq, x, n, ldx, dim, fa);
  } else {
    auto kern = fused_knn_kernel<TQ, TX, false>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, grid, dim3(256), dyn, res.stream, q, m, ldq, x, n, ldx, dim, fa);
  }
  profile_end(res, "fused_knn_kernel");
  HIP_TRY(hipGetLastError());
  if (n_splits > 1) {
    select_k<int64_t, int64_t>(res, pd.data(), pi.data(), m, n_splits * k, n_splits * k, k, out_d, out_i,
                               metric != M_InnerProduct);
  }
  return true;
}

template bool fused_knn<float, float>(resources&, const float*, int64_t, int64_t, const float*, int64_t, int64_t,
                                      int64_t, const float*, const float*, int, int, float*, int64_t*);
template bool fused_knn<__half, __half>(resources&, const __half*, int64_t, int64_t, const __half*, int64_t, int64_t,
                                        int64_t, const float*, const float*, int, int, float*, int64_t*);

}  // namespace cuvs_amd
