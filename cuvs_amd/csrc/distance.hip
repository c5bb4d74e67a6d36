// Distance "GEMMs" of the hot path on the fp32 matrix cores (v_mfma_f32_16x16x4_f32):
//   * pairwise_distance : Q[m,d] x X[n,d] -> D[m,n] with the L2-expanded / IP / cosine epilogue
//     (reference: cuBLAS/CUTLASS GEMM + map_offset epilogue, knn_brute_force.cuh:183-232,
//      distance_ops/l2_exp.cuh:36-50; IVF coarse search ivf_pq_search.cuh:143-157, ivf_flat_search.cuh:148-162)
//   * fused_l2_argmin   : same main loop, running argmin epilogue (reference: fusedDistanceNNMinReduce,
//     cluster/detail/minClusterDistanceCompute.cu:63-78; predict_core kmeans_balanced.cuh:76)
//   * row_norms         : canonical squared norms
// Tile: 128x128x16 per 256-thread workgroup, 2x2 waves, each wave 4x4 MFMA tiles (64 accumulator VGPRs),
// operands staged k-major in LDS with an XOR swizzle (conflict-free ds_write_b32 and ds_read_b32),
// global->register prefetch of the next k-tile overlapped with the MFMAs of the current one.
// fp32 MFMA is a k-ordered fma chain, so every dot product is reproducible bit for bit (oracle: canon_dot).
#include "ops.hpp"
#include "device_utils.hpp"
#include "distance_tile.hpp"

#include <cfloat>
#include <cstdlib>
#include <type_traits>

namespace cuvs_amd {

namespace {

// ------------------------------------------------------------------ row norms
template <typename T>
__global__ __launch_bounds__(256) void row_norms_kernel(const T* __restrict__ x, int64_t n, int64_t dim,
                                                        int64_t ld, float* __restrict__ out, bool sqrt_out)
{
  // grid-stride over rows: HIP limits gridDim.x * blockDim.x to < 2^32 threads, so 100M+ rows cannot get
  // one wave each from the grid alone
  const int lane = lane_id();
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
    const T* r = x + row * ld;
    float acc  = 0.f;
    for (int64_t j = lane; j < dim; j += kWave) {
      float v = to_float(r[j]);
      acc     = __fmaf_rn(v, v, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = acc + __shfl_xor(acc, off, kWave);
    if (lane == 0) out[row] = sqrt_out ? sqrtf(acc) : acc;
  }
}

// x_r <- x_r * (1 / sqrt(canonical |x_r|^2)) in place; rows of norm 0 become 0 (cosine: ivf_pq_build.cuh:159-166,
// raft::linalg::row_normalize). Oracle twin: oracle.c `normalize_row`.
__global__ __launch_bounds__(256) void normalize_rows_kernel(float* __restrict__ x, int64_t n, int64_t dim)
{
  const int lane = lane_id();
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
    float* r  = x + row * dim;
    float acc = 0.f;
    for (int64_t j = lane; j < dim; j += kWave) acc = __fmaf_rn(r[j], r[j], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = acc + __shfl_xor(acc, off, kWave);
    const float inv = acc > 0.f ? 1.0f / sqrtf(acc) : 0.f;
    for (int64_t j = lane; j < dim; j += kWave) r[j] = r[j] * inv;
  }
}

// MODE 2 (brute force beyond the first column tile): the distance tile never leaves the registers - every element that
// beats its row's current k-th value is appended behind the row's top-k (see brute_force.hip).
struct append_args {
  float* buf_v       = nullptr;  // [m, ldb]: the row's sorted top-k, then the appended candidates
  int64_t* buf_i     = nullptr;
  int* cnt           = nullptr;  // [m] appended so far (may exceed cap: the caller checks)
  int64_t ldb        = 0;
  int k              = 0;
  int cap            = 0;
  int64_t col_off    = 0;        // source id of column 0 of x
  int64_t col_stride = 1;        // source id of column j = col_off + j * col_stride (strided column sets: the coarse search)
  const float* thr   = nullptr;  // optional [m] thresholds instead of the rows' k-th values (ties at the k-th value pass)
  int64_t row_off    = 0;        // query row of row 0 of q (bitmap filters)
  int64_t n_total    = 0;        // indexed rows (bitmap row pitch)
  const uint32_t* bits = nullptr;
  int filter_type    = 0;        // 0 none, 1 bitset, 2 bitmap
  int select_min     = 1;
  int srt = 1, sct = 1;          // supertile: srt row tiles x sct column tiles per 256 consecutive blocks of an XCD
  int64_t n_rt = 0, n_ct = 0;
  unsigned long long* stats = nullptr;  // dbg & 4: cycles of [prologue, main loop, epilogue] summed over workgroups, + count
  int dbg = 0;  // timing experiments only (results are wrong): 1 no staging in the main loop, 2 no LDS reads either
  uint32_t* gkeys = nullptr;  // MODE 3: [m, ldg] best order-preserving key of every 16-column group of a row
  int64_t ldg     = 0;
};

// MODE 0: write D tile.  MODE 1: running argmin over all column tiles (grid.x = 1).  MODE 2: threshold + append.
template <typename TQ, typename TX, int MODE, bool VEC>
__global__ __launch_bounds__(256) void dist_mfma_kernel(const TQ* __restrict__ q, int64_t m, int64_t ldq,
                                                        const TX* __restrict__ x, int64_t n, int64_t ldx,
                                                        int64_t dim, epilogue_args ep,
                                                        float* __restrict__ out, int64_t ldo,
                                                        uint32_t* __restrict__ labels,
                                                        float* __restrict__ min_val, append_args ap)
{
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDT];
  if (ep.run_if != nullptr && *ep.run_if == 0u) return;  // a guarded launch whose condition did not arise

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm   = wave >> 1;
  const int wn   = wave & 1;
  const int l15  = lane & 15;
  const int lg   = lane >> 4;

  int64_t row0 = (int64_t)blockIdx.y * BM;
  int64_t col0_m2 = 0;
  if constexpr (MODE == 2) {
    // XCD x (blocks x, x + 8, ...) owns every 8th supertile; the srt row tiles and sct column tiles of a supertile
    // (~2 MiB of operands) stay in its L2 while its srt * sct blocks run
    const int64_t j  = (int64_t)(blockIdx.x >> 3);
    const int64_t g  = (j / (ap.srt * ap.sct)) * 8 + (blockIdx.x & 7u);
    const int w      = (int)(j % (ap.srt * ap.sct));
    const int64_t sr = (ap.n_rt + ap.srt - 1) / ap.srt;
    const int64_t rt = (g % sr) * ap.srt + (w % ap.srt);
    const int64_t ct = (g / sr) * ap.sct + (w / ap.srt);
    if (rt >= ap.n_rt || ct >= ap.n_ct) return;  // workgroup-uniform
    row0    = rt * BM;
    col0_m2 = ct * BN;
  }
  const int nkt      = (int)((dim + BK - 1) / BK);

  // staging assignment: two (row, chunk) pairs per thread and per operand
  const int sr0 = tid >> 2, sc0 = tid & 3;          // rows 0..63
  const int sr1 = (tid + 256) >> 2, sc1 = tid & 3;  // rows 64..127

  float best_v[16];
  uint32_t best_i[16];
  if constexpr (MODE == 1) {
#pragma unroll
    for (int t = 0; t < 16; ++t) { best_v[t] = FLT_MAX; best_i[t] = 0xffffffffu; }
  }

  const int64_t n_col_tiles = (MODE == 1) ? (n + BN - 1) / BN : 1;
  for (int64_t ct = 0; ct < n_col_tiles; ++ct) {
    const int64_t col0 = (MODE == 1) ? ct * BN : (MODE == 2) ? col0_m2 : (int64_t)blockIdx.x * BN;

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
    __syncthreads();  // previous column tile done with the LDS buffers
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

    // ---- epilogue: C layout col = lane&15, row = (lane>>4)*4 + e
    if constexpr (MODE == 0) {
      float xnv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t col = col0 + wn * 64 + j * 16 + l15;
        xnv[j]      = (ep.xn != nullptr && col < n) ? ep.xn[col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int64_t row = row0 + wm * 64 + i * 16 + lg * 4 + e;
          if (row >= m) continue;
          float qnv = ep.qn != nullptr ? ep.qn[row] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int64_t col = col0 + wn * 64 + j * 16 + l15;
            if (col < n) out[row * ldo + col] = finish_distance(acc[i][j][e], qnv, xnv[j], ep.metric, ep.clamp_eps);
          }
        }
      }
    } else if constexpr (MODE == 2) {
      float xnv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t col = col0 + wn * 64 + j * 16 + l15;
        xnv[j]      = (ep.xn != nullptr && col < n) ? ep.xn[col] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t row = row0 + wm * 64 + i * 16 + lg * 4 + e;
          if (row >= m) continue;
          const float qnv = ep.qn != nullptr ? ep.qn[row] : 0.f;
          const float thr = ap.thr != nullptr ? ap.thr[row] : ap.buf_v[row * ap.ldb + ap.k - 1];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t col = col0 + wn * 64 + j * 16 + l15;
            const float d     = finish_distance(acc[i][j][e], qnv, xnv[j], ep.metric, ep.clamp_eps);
            bool take         = col < n && (ap.select_min ? d < thr : d > thr);
            if (take && ap.filter_type != 0) {
              const int64_t bit = ap.filter_type == 2 ? (ap.row_off + row) * ap.n_total + (ap.col_off + col) : ap.col_off + col;
              take              = (ap.bits[bit >> 5] >> (bit & 31)) & 1u;
            }
            if (take) {
              const int pos = atomicAdd(&ap.cnt[row], 1);
              if (pos < ap.cap) {
                ap.buf_v[row * ap.ldb + ap.k + pos] = d;
                ap.buf_i[row * ap.ldb + ap.k + pos] = ap.col_off + col * ap.col_stride;
              }
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t col = col0 + wn * 64 + j * 16 + l15;
        if (col >= n) continue;
        float xnv = ep.xn[col];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = __fmaf_rn(-2.0f, acc[i][j][e], xnv);
            int t   = i * 4 + e;
            // columns are visited in increasing order per lane: strict < keeps the smallest index
            if (v < best_v[t]) { best_v[t] = v; best_i[t] = (uint32_t)col; }
          }
      }
    }
  }

  if constexpr (MODE == 1) {
    // reduce over the 16 lanes that share a row, then over the two column waves
    __syncthreads();
    float* red_v    = As[0];                                   // [2][128]
    uint32_t* red_i = reinterpret_cast<uint32_t*>(Bs[0]);      // [2][128]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float v    = best_v[t];
      uint32_t i = best_i[t];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        float ov    = __shfl_xor(v, off, kWave);
        uint32_t oi = __shfl_xor(i, off, kWave);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
      }
      if (l15 == 0) {
        int r                 = wm * 64 + (t >> 2) * 16 + lg * 4 + (t & 3);
        red_v[wn * BM + r]    = v;
        red_i[wn * BM + r]    = i;
      }
    }
    __syncthreads();
    if (tid < BM) {
      int64_t row = row0 + tid;
      if (row < m) {
        float v0 = red_v[tid], v1 = red_v[BM + tid];
        uint32_t i0 = red_i[tid], i1 = red_i[BM + tid];
        bool take1 = (v1 < v0) || (v1 == v0 && i1 < i0);
        labels[row] = take1 ? i1 : i0;
        if (min_val != nullptr) min_val[row] = take1 ? v1 : v0;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// dist_tile_kernel: the same 128 x 128 x 16 tile, accumulation order and epilogue as dist_mfma_kernel (bit-identical
// results) for the common shape - 16-byte aligned rows, dim a multiple of 16 - without the edge handling in the main
// loop: rows past the end are CLAMPED to the last row (their results are dropped by the epilogue) instead of being
// branched around, so a k-tile is 4 unconditional 16-byte loads, 64 MFMAs, 16 ds_read2 with immediate offsets (the XOR
// swizzle c << 3 is folded into "fragment i ^ (c >> 1) at one of two base addresses"), 8 ds_write2 and one barrier.
// MODE 0 writes the tile; MODE 2 keeps it in registers: one straight-line pass compares every element with its row's
// k-th value, and only a lane that found something (rare once the thresholds are warm) enters the append loop.
template <typename T>
__device__ inline void load4v(const T* __restrict__ p, float (&v)[4])
{
  if constexpr (sizeof(T) == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const uint2 t   = *reinterpret_cast<const uint2*>(p);
    const __half* h = reinterpret_cast<const __half*>(&t);
    v[0] = __half2float(h[0]); v[1] = __half2float(h[1]); v[2] = __half2float(h[2]); v[3] = __half2float(h[3]);
  }
}

constexpr int kTileOcc = 3;  // waves per SIMD the register budget is set for
// MODE 3 (grouped output, the coarse search: K = dim is short): the epilogue - 64 KiB of stores per tile - waits on HBM writes
// for 38 % of a tile's time (CUVS_AMD_TILE_DBG=4: prologue 7 k, main loop 54 k, epilogue 38 k cycles at 10k x 16384 x 128), and
// with one wave per SIMD and workgroup only the OTHER workgroups of the CU can keep the matrix pipe busy meanwhile. A register
// budget for four workgroups per CU measured the same (0.638 ms for GEMM + selection at that shape, round 5): kept at three.
constexpr int kTileOccGrouped = 3;

template <typename TQ, typename TX, int MODE, int OCC, int METRIC>
__global__ __launch_bounds__(256, OCC) void dist_tile_kernel(const TQ* __restrict__ q, int64_t m, int64_t ldq,
                                                             const TX* __restrict__ x, int64_t n, int64_t ldx, int64_t dim,
                                                             epilogue_args ep, float* __restrict__ out, int64_t ldo,
                                                             append_args ap)
{
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDT];
  if (ep.run_if != nullptr && *ep.run_if == 0u) return;  // a guarded launch whose condition did not arise

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm   = wave >> 1;
  const int wn   = wave & 1;
  const int l15  = lane & 15;
  const int lg   = lane >> 4;

  int64_t row0, col0;
  if constexpr (MODE == 2) {
    const int64_t j  = (int64_t)(blockIdx.x >> 3);
    const int64_t g  = (j / (ap.srt * ap.sct)) * 8 + (blockIdx.x & 7u);
    const int w      = (int)(j % (ap.srt * ap.sct));
    const int64_t sr = (ap.n_rt + ap.srt - 1) / ap.srt;
    const int64_t rt = (g % sr) * ap.srt + (w % ap.srt);
    const int64_t ct = (g / sr) * ap.sct + (w / ap.srt);
    if (rt >= ap.n_rt || ct >= ap.n_ct) return;  // workgroup-uniform
    row0 = rt * BM;
    col0 = ct * BN;
  } else {
    row0 = (int64_t)blockIdx.y * BM;
    col0 = (int64_t)blockIdx.x * BN;
  }
  const int nkt = (int)(dim / BK);
  // the tile's row / column norms and (MODE 2) row thresholds, staged once: visible after the first barrier
  __shared__ __attribute__((aligned(16))) float s_qn[BM], s_xn[BN], s_thr[BM];
  if (tid < BM) {
    const int64_t row = row0 + tid;
    s_qn[tid]         = ep.qn != nullptr ? ep.qn[min(row, m - 1)] : 0.f;
    if constexpr (MODE == 2)  // rows past the end never pass
      s_thr[tid] = row < m ? (ap.thr != nullptr ? ap.thr[row] : ap.buf_v[row * ap.ldb + ap.k - 1]) : (ap.select_min ? -INFINITY : INFINITY);
  } else {
    s_xn[tid - BM] = ep.xn != nullptr ? ep.xn[min(col0 + tid - BM, n - 1)] : 0.f;
  }
  const bool stat = (MODE == 2 || MODE == 3) && (ap.dbg & 4);
  const unsigned long long t_start = stat ? __builtin_readcyclecounter() : 0ull;

  // staging: thread t moves rows t / 4 and 64 + t / 4, k-chunk t % 4 of both operands
  const int sr0 = tid >> 2, sc = tid & 3;
  const TQ* pa0 = q + min(row0 + sr0, m - 1) * ldq + sc * 4;
  const TQ* pa1 = q + min(row0 + sr0 + 64, m - 1) * ldq + sc * 4;
  const TX* pb0 = x + min(col0 + sr0, n - 1) * ldx + sc * 4;
  const TX* pb1 = x + min(col0 + sr0 + 64, n - 1) * ldx + sc * 4;
  const int st0 = (4 * sc) * LDT + (sr0 ^ (sc << 3));         // stage_store's address, e = 0
  const int st1 = (4 * sc) * LDT + ((sr0 + 64) ^ (sc << 3));

  // fragment reads: A[(4c + lg) * LDT + ((wm * 64 + i * 16 + l15) ^ (c << 3))]; bit 3 of the swizzle flips bit 3 of
  // l15 (two base addresses), bit 4 turns fragment i into fragment i ^ 1 (an immediate)
  const int ra_[2] = {lg * LDT + wm * 64 + l15, lg * LDT + wm * 64 + (l15 ^ 8)};
  const int rb_[2] = {lg * LDT + wn * 64 + l15, lg * LDT + wn * 64 + (l15 ^ 8)};

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ga0[4], ga1[4], gb0[4], gb1[4];
  load4v(pa0, ga0); load4v(pa1, ga1); load4v(pb0, gb0); load4v(pb1, gb1);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    As[0][st0 + e * LDT] = ga0[e]; As[0][st1 + e * LDT] = ga1[e];
    Bs[0][st0 + e * LDT] = gb0[e]; Bs[0][st1 + e * LDT] = gb1[e];
  }
  __syncthreads();
  const unsigned long long t_loop = stat ? __builtin_readcyclecounter() : 0ull;

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt && !(ap.dbg & 1)) {
      const int k0 = (kt + 1) * BK;
      load4v(pa0 + k0, ga0); load4v(pa1 + k0, ga1); load4v(pb0 + k0, gb0); load4v(pb1 + k0, gb1);
    }
    const float* A = As[buf];
    const float* B = Bs[buf];
    // fragment reads of step c + 1 are issued before the MFMAs of step c (two register sets): one exposed LDS latency
    // per k-tile instead of four
    float a[2][4], b[2][4];
    auto read_frags = [&](const int c, float (&fa)[4], float (&fb)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = A[ra_[c & 1] + (4 * c) * LDT + ((i ^ (c >> 1)) * 16)];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = B[rb_[c & 1] + (4 * c) * LDT + ((j ^ (c >> 1)) * 16)];
    };
    read_frags(0, a[0], b[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < 3) read_frags(c + 1, a[(c + 1) & 1], b[(c + 1) & 1]);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // the 8 ds_read2 first ...
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][i], b[c & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // ... then the 16 MFMAs of this step
    }
    if (ap.dbg & 1) continue;
    if (kt + 1 < nkt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[buf ^ 1][st0 + e * LDT] = ga0[e]; As[buf ^ 1][st1 + e * LDT] = ga1[e];
        Bs[buf ^ 1][st0 + e * LDT] = gb0[e]; Bs[buf ^ 1][st1 + e * LDT] = gb1[e];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C layout col = lane & 15, row = (lane >> 4) * 4 + e
  const unsigned long long t_epi = stat ? __builtin_readcyclecounter() : 0ull;
  float xnv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xnv[j] = s_xn[wn * 64 + j * 16 + l15];
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 qn4 = *reinterpret_cast<const f32x4*>(&s_qn[wm * 64 + i * 16 + lg * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = row0 + wm * 64 + i * 16 + lg * 4 + e;
        if (row >= m) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t col = col0 + wn * 64 + j * 16 + l15;
          if (col < n) out[row * ldo + col] = finish_distance(acc[i][j][e], qn4[e], xnv[j], METRIC, ep.clamp_eps);
        }
      }
    }
  } else if constexpr (MODE == 3) {
    // GROUPED output for a top-k that follows (the coarse search: 128 of 16384 per row). A lane holds, for each of its 16
    // rows, the columns c0 + j * 16 + l15 (j = 0..3) of its wave's 64-column half: these four values go out as ONE 16-byte
    // store at position c0 + l15 * 4 + j - the row is written in a permuted order, a fixed permutation inside every
    // 128-column tile (grouped_col() is its inverse) - and the 16 values of a lane QUAD form a group whose best
    // order-preserving key (two DPP steps) is written to gkeys[row, column / 16]. The selection then reads the 1 / 16 of
    // keys and only the groups that can hold one of a row's k best (select_k_grouped, select_k.hip) instead of the matrix.
    // Columns past n are written as the worst value and never win.
    constexpr bool smin = METRIC != M_InnerProduct;
    const float worst   = smin ? INFINITY : -INFINITY;
    const uint32_t flip = smin ? 0u : 0xffffffffu;
    bool colok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) colok[j] = col0 + wn * 64 + j * 16 + l15 < n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 qn4 = *reinterpret_cast<const f32x4*>(&s_qn[wm * 64 + i * 16 + lg * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = row0 + wm * 64 + i * 16 + lg * 4 + e;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = colok[j] ? finish_distance(acc[i][j][e], qn4[e], xnv[j], METRIC, ep.clamp_eps) : worst;
        // the best VALUE of the lane's four, then of the quad (two DPP steps), as a float - v_min / v_max return one of their
        // operands, so the key written is the key of an element of the group; an element of the group that compares equal
        // but has a smaller key (-0 next to +0) is covered by writing the key of -|best| for a zero (the smaller of the two)
        float bf = smin ? fminf(fminf(v[0], v[1]), fminf(v[2], v[3])) : fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        float o1 = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(bf), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
        bf       = smin ? fminf(bf, o1) : fmaxf(bf, o1);
        float o2 = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(bf), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
        bf       = smin ? fminf(bf, o2) : fmaxf(bf, o2);
        if (bf == 0.f) bf = smin ? -0.f : 0.f;
        const uint32_t best = float_to_key(bf) ^ flip;
        if (row < m) {
          if (ap.dbg & 8) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(&out[row * ldo + col0 + wn * 64 + l15 * 4]));
          else *reinterpret_cast<f32x4*>(&out[row * ldo + col0 + wn * 64 + l15 * 4]) = v;
          if ((l15 & 3) == 0) ap.gkeys[row * ap.ldg + (col0 >> 4) + wn * 4 + (l15 >> 2)] = best;
        }
      }
    }
    if (stat && tid == 0) {
      const unsigned long long t_end = __builtin_readcyclecounter();
      atomicAdd(&ap.stats[0], t_loop - t_start);
      atomicAdd(&ap.stats[1], t_epi - t_loop);
      atomicAdd(&ap.stats[2], t_end - t_epi);
      atomicAdd(&ap.stats[3], 1ull);
    }
  } else {
    // one straight-line pass: does this lane hold anything that beats its row's k-th value? (norms and thresholds come
    // from LDS four rows at a time: 8 live registers instead of 32)
    // The L2 family is screened with a cheaper SUPERSET test (the append loop below decides exactly): with
    // v = fma(-2, dot, qn + xn), the finished distance is max(v, 0) - or 0 when the self-neighbour clamp fires, which needs
    // v^2 < eps - so "distance < thr" implies v < max(thr', sqrt(eps)), thr' = thr (squared metrics) or thr^2 rounded up
    // (sqrt metrics). 4 instructions per element instead of 9.
    uint32_t any = 0u;
    constexpr bool L2 = METRIC != M_InnerProduct && METRIC != M_CosineExpanded;
    constexpr bool SQ = METRIC == M_L2SqrtExpanded || METRIC == M_L2SqrtUnexpanded;
    const float floor_v = sqrtf(ep.clamp_eps) * 1.0001f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 qn4 = *reinterpret_cast<const f32x4*>(&s_qn[wm * 64 + i * 16 + lg * 4]);
      const f32x4 th4 = *reinterpret_cast<const f32x4*>(&s_thr[wm * 64 + i * 16 + lg * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (L2) {
          const float t  = SQ ? th4[e] * th4[e] * 1.000001f : th4[e];  // -inf (rows past the end) stays out of reach
          const float tt = th4[e] < 0.f ? th4[e] : fmaxf(t, floor_v);
#pragma unroll
          for (int j = 0; j < 4; ++j) any |= (uint32_t)(__fmaf_rn(-2.0f, acc[i][j][e], qn4[e] + xnv[j]) < tt);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = finish_distance(acc[i][j][e], qn4[e], xnv[j], METRIC, ep.clamp_eps);
            any |= (uint32_t)(METRIC != M_InnerProduct ? d < th4[e] : d > th4[e]);
          }
        }
      }
    }
    if (any != 0u) {  // rare once the thresholds are warm
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 qn4 = *reinterpret_cast<const f32x4*>(&s_qn[wm * 64 + i * 16 + lg * 4]);
        const f32x4 th4 = *reinterpret_cast<const f32x4*>(&s_thr[wm * 64 + i * 16 + lg * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t row = row0 + wm * 64 + i * 16 + lg * 4 + e;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t col = col0 + wn * 64 + j * 16 + l15;
            const float d = finish_distance(acc[i][j][e], qn4[e], xnv[j], METRIC, ep.clamp_eps);
            bool take     = col < n && (METRIC != M_InnerProduct ? d < th4[e] : d > th4[e]);
            if (take && ap.filter_type != 0) {
              const int64_t bit = ap.filter_type == 2 ? (ap.row_off + row) * ap.n_total + (ap.col_off + col) : ap.col_off + col;
              take              = (ap.bits[bit >> 5] >> (bit & 31)) & 1u;
            }
            if (take) {
              const int pos = atomicAdd(&ap.cnt[row], 1);
              if (pos < ap.cap) {
                ap.buf_v[row * ap.ldb + ap.k + pos] = d;
                ap.buf_i[row * ap.ldb + ap.k + pos] = ap.col_off + col * ap.col_stride;
              }
            }
          }
        }
      }
    }
    if (stat && tid == 0) {
      const unsigned long long t_end = __builtin_readcyclecounter();
      atomicAdd(&ap.stats[0], t_loop - t_start);
      atomicAdd(&ap.stats[1], t_epi - t_loop);
      atomicAdd(&ap.stats[2], t_end - t_epi);
      atomicAdd(&ap.stats[3], 1ull);
    }
  }
}

template <typename TQ, typename TX, int MODE>
void launch_tile(int metric, dim3 grid, hipStream_t stream, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n,
                 int64_t ldx, int64_t dim, epilogue_args ep, float* out, int64_t ldo, append_args ap)
{
#define TILE_CASE(M)                                                                                                     \
  case M:                                                                                                                \
    hipLaunchKernelGGL((dist_tile_kernel<TQ, TX, MODE, (MODE == 3 ? kTileOccGrouped : kTileOcc), M>), grid, dim3(256), 0, stream, q, m, ldq, x, n, ldx, dim, \
                       ep, out, ldo, ap);                                                                                \
    break;
  switch (metric) {
    TILE_CASE(M_L2Expanded) TILE_CASE(M_L2SqrtExpanded) TILE_CASE(M_CosineExpanded) TILE_CASE(M_L2Unexpanded)
    TILE_CASE(M_L2SqrtUnexpanded) TILE_CASE(M_InnerProduct)
    default: CUVS_FAIL("unsupported metric %d", metric);
  }
#undef TILE_CASE
}

}  // namespace

template <typename T>
void row_norms(resources& res, const T* x, int64_t n, int64_t dim, int64_t ld, float* out, bool sqrt_out)
{
  if (n == 0) return;
  int64_t blocks = std::min<int64_t>((n + 3) / 4, int64_t(1) << 22);  // <= 2^30 threads (HIP grid limit 2^32)
  hipLaunchKernelGGL((row_norms_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, res.stream, x, n, dim, ld,
                     out, sqrt_out);
  HIP_TRY(hipGetLastError());
}

void normalize_rows(resources& res, float* x, int64_t n, int64_t dim)
{
  if (n == 0) return;
  int64_t blocks = std::min<int64_t>((n + 3) / 4, int64_t(1) << 22);
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, res.stream, x, n, dim);
  HIP_TRY(hipGetLastError());
}

template <typename TQ, typename TX>
void pairwise_distance(resources& res, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n,
                       int64_t ldx, int64_t dim, const float* qn, const float* xn, int metric,
                       float* out, int64_t ldo, const uint32_t* run_if)
{
  if (m == 0 || n == 0) return;
  CUVS_EXPECTS(metric_supported(metric), "pairwise_distance: unsupported metric %d", metric);
  if (metric != M_InnerProduct) CUVS_EXPECTS(qn && xn, "pairwise_distance: norms required");
  epilogue_args ep{qn, xn, metric, (sizeof(TX) == 2 ? 1e-3f : 1e-6f), run_if};
  dim3 grid((unsigned)((n + BN - 1) / BN), (unsigned)((m + BM - 1) / BM));
  CUVS_EXPECTS((m + BM - 1) / BM <= 65535, "pairwise_distance: too many query rows per call");
  bool vec = vec_ok(q, ldq, dim) && vec_ok(x, ldx, dim);
  if constexpr (std::is_same_v<TQ, TX> && (sizeof(TQ) == 4 || sizeof(TQ) == 2)) {
    if (vec && dim % BK == 0 && !res.tune.dist_old) {
      launch_tile<TQ, TX, 0>(metric, grid, res.stream, q, m, ldq, x, n, ldx, dim, ep, out, ldo, append_args{});
      HIP_TRY(hipGetLastError());
      return;
    }
  }
  if (vec) {
    hipLaunchKernelGGL((dist_mfma_kernel<TQ, TX, 0, true>), grid, dim3(256), 0, res.stream, q, m, ldq, x, n,
                       ldx, dim, ep, out, ldo, (uint32_t*)nullptr, (float*)nullptr, append_args{});
  } else {
    hipLaunchKernelGGL((dist_mfma_kernel<TQ, TX, 0, false>), grid, dim3(256), 0, res.stream, q, m, ldq, x, n,
                       ldx, dim, ep, out, ldo, (uint32_t*)nullptr, (float*)nullptr, append_args{});
  }
  HIP_TRY(hipGetLastError());
}

// Distances in GROUPED layout for a following select_k_grouped (dist_tile_kernel MODE 3): out [m, ldo] with ldo >= n rounded
// up to 128, gkeys [m, ldg] with ldg >= ldo / 16. Returns false - nothing launched - for shapes the tile kernel does not
// take (rows not 16-byte aligned, dim not a multiple of 16): the caller then uses pairwise_distance + select_k.
bool pairwise_distance_grouped(resources& res, const float* q, int64_t m, int64_t ldq, const float* x, int64_t n, int64_t ldx,
                               int64_t dim, const float* qn, const float* xn, int metric, float* out, int64_t ldo, uint32_t* gkeys,
                               int64_t ldg)
{
  if (m == 0 || n == 0) return true;
  if (!(vec_ok(q, ldq, dim) && vec_ok(x, ldx, dim) && dim % BK == 0) || res.tune.dist_old) return false;
  if ((m + BM - 1) / BM > 65535 || ldo % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) return false;
  CUVS_EXPECTS(ldo >= round_up(n, BN) && ldg * 16 >= round_up(n, BN), "pairwise_distance_grouped: row pitch");
  CUVS_EXPECTS(metric == M_InnerProduct || (qn && xn), "pairwise_distance_grouped: norms required");
  epilogue_args ep{qn, xn, metric, 1e-6f, nullptr};
  append_args ap;
  ap.gkeys = gkeys; ap.ldg = ldg;
  ap.dbg = res.tune.tile_dbg & 4;  // CUVS_AMD_TILE_DBG=4: cycles of prologue / main loop / epilogue per tile (stderr)
  // the selection that follows reads a sixteenth of the matrix (keys) plus the few qualifying groups: from 4096 rows on (the IVF
  // coarse searches: 10k x 16384 = 655 MB) the tile's 16-byte stores are non-temporal - nothing of the matrix is worth a line of
  // L2 (10k x 16384 x 128, GEMM + selection: 0.625 -> 0.568 ms, profiles/r06_coarse_nt_probe.log; 1000 x 100000: 3 % slower, plain)
  if ((m >= 4096 && !(res.tune.tile_dbg & 16)) || (res.tune.tile_dbg & 8)) ap.dbg |= 8;
  dev_buf<unsigned long long> stats;
  if (ap.dbg & 4) {
    stats = dev_buf<unsigned long long>(res, 4);
    HIP_TRY(hipMemsetAsync(stats.data(), 0, stats.bytes(), res.stream));
    ap.stats = stats.data();
  }
  dim3 grid((unsigned)((n + BN - 1) / BN), (unsigned)((m + BM - 1) / BM));
  launch_tile<float, float, 3>(metric, grid, res.stream, q, m, ldq, x, n, ldx, dim, ep, out, ldo, ap);
  HIP_TRY(hipGetLastError());
  if (ap.dbg & 4) {
    unsigned long long h[4];
    HIP_TRY(hipMemcpyAsync(h, stats.data(), sizeof(h), hipMemcpyDeviceToHost, res.stream));
    HIP_TRY(hipStreamSynchronize(res.stream));
    fprintf(stderr, "[dist_tile grouped stats] tiles %llu, cycles per tile (wave 0): prologue %.0f main loop %.0f epilogue %.0f\n", h[3],
            (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3]);
  }
  return true;
}

template <typename TQ, typename TX>
void pairwise_threshold_append(resources& res, const TQ* q, int64_t m, int64_t ldq, const TX* x, int64_t n, int64_t ldx,
                               int64_t dim, const float* qn, const float* xn, int metric, float* buf_v, int64_t* buf_i,
                               int* cnt, int k, int cap, int64_t col_off, int64_t row_off, int64_t n_total,
                               const uint32_t* bits, int filter_type, int64_t col_stride, const float* thr)
{
  if (m == 0 || n == 0) return;
  CUVS_EXPECTS(metric_supported(metric), "pairwise_threshold_append: unsupported metric %d", metric);
  epilogue_args ep{qn, xn, metric, (sizeof(TX) == 2 ? 1e-3f : 1e-6f)};
  append_args ap;
  ap.buf_v = buf_v; ap.buf_i = buf_i; ap.cnt = cnt; ap.ldb = k + cap; ap.k = k; ap.cap = cap;
  ap.col_off = col_off; ap.row_off = row_off; ap.n_total = n_total; ap.bits = bits; ap.filter_type = filter_type;
  ap.col_stride = col_stride; ap.thr = thr;
  ap.select_min = metric != M_InnerProduct;
  ap.n_rt = (m + BM - 1) / BM;
  ap.n_ct = (n + BN - 1) / BN;
  ap.srt  = (int)std::min<int64_t>(16, ap.n_rt);
  ap.sct  = 256 / ap.srt;
  ap.dbg = res.tune.tile_dbg;
  const int64_t supertiles = ((ap.n_rt + ap.srt - 1) / ap.srt) * ((ap.n_ct + ap.sct - 1) / ap.sct);
  const int64_t blocks     = (supertiles + 7) / 8 * 8 * (int64_t)(ap.srt * ap.sct);
  CUVS_EXPECTS(blocks < (int64_t(1) << 31), "pairwise_threshold_append: grid too large");
  const bool vec = vec_ok(q, ldq, dim) && vec_ok(x, ldx, dim);
  if (vec && dim % BK == 0 && !res.tune.dist_old) {
    dev_buf<unsigned long long> stats;
    if (ap.dbg & 4) {
      stats = dev_buf<unsigned long long>(res, 4);
      HIP_TRY(hipMemsetAsync(stats.data(), 0, stats.bytes(), res.stream));
      ap.stats = stats.data();
    }
    launch_tile<TQ, TX, 2>(metric, dim3((unsigned)blocks), res.stream, q, m, ldq, x, n, ldx, dim, ep, (float*)nullptr, (int64_t)0, ap);
    HIP_TRY(hipGetLastError());
    if (ap.dbg & 4) {
      unsigned long long h[4];
      HIP_TRY(hipMemcpyAsync(h, stats.data(), sizeof(h), hipMemcpyDeviceToHost, res.stream));
      HIP_TRY(hipStreamSynchronize(res.stream));
      fprintf(stderr, "[dist_tile stats] tiles %llu, cycles per tile (wave 0): prologue %.0f main loop %.0f epilogue %.0f\n", h[3],
              (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3]);
    }
    return;
  }
  if (vec) {
    hipLaunchKernelGGL((dist_mfma_kernel<TQ, TX, 2, true>), dim3((unsigned)blocks), dim3(256), 0, res.stream, q, m, ldq, x, n,
                       ldx, dim, ep, (float*)nullptr, (int64_t)0, (uint32_t*)nullptr, (float*)nullptr, ap);
  } else {
    hipLaunchKernelGGL((dist_mfma_kernel<TQ, TX, 2, false>), dim3((unsigned)blocks), dim3(256), 0, res.stream, q, m, ldq, x, n,
                       ldx, dim, ep, (float*)nullptr, (int64_t)0, (uint32_t*)nullptr, (float*)nullptr, ap);
  }
  HIP_TRY(hipGetLastError());
}

template <typename TQ>
void fused_l2_argmin(resources& res, const TQ* q, int64_t m, int64_t ldq, const float* centers,
                     int64_t n, int64_t dim, const float* center_norms, uint32_t* labels,
                     float* min_val)
{
  if (m == 0) return;
  CUVS_EXPECTS(n > 0 && center_norms != nullptr, "fused_l2_argmin: need centers and norms");
  epilogue_args ep{nullptr, center_norms, M_L2Expanded, 0.f};
  // grid.y is limited to 65535 blocks: walk the rows in slabs
  const int64_t max_rows = int64_t(65535) * BM;
  bool vec = vec_ok(q, ldq, dim) && vec_ok(centers, dim, dim);
  for (int64_t r0 = 0; r0 < m; r0 += max_rows) {
    int64_t mr = std::min(max_rows, m - r0);
    dim3 grid(1, (unsigned)((mr + BM - 1) / BM));
    const TQ* qq   = q + r0 * ldq;
    uint32_t* lab  = labels + r0;
    float* mv      = min_val ? min_val + r0 : nullptr;
    if (vec) {
      hipLaunchKernelGGL((dist_mfma_kernel<TQ, float, 1, true>), grid, dim3(256), 0, res.stream, qq, mr, ldq,
                         centers, n, dim, dim, ep, (float*)nullptr, (int64_t)0, lab, mv, append_args{});
    } else {
      hipLaunchKernelGGL((dist_mfma_kernel<TQ, float, 1, false>), grid, dim3(256), 0, res.stream, qq, mr, ldq,
                         centers, n, dim, dim, ep, (float*)nullptr, (int64_t)0, lab, mv, append_args{});
    }
  }
  HIP_TRY(hipGetLastError());
}

#define INST_N(T) template void row_norms<T>(resources&, const T*, int64_t, int64_t, int64_t, float*, bool);
INST_N(float) INST_N(__half) INST_N(int8_t) INST_N(uint8_t)
#undef INST_N

#define INST_P(TQ, TX)                                                                                    \
  template void pairwise_distance<TQ, TX>(resources&, const TQ*, int64_t, int64_t, const TX*, int64_t,    \
                                          int64_t, int64_t, const float*, const float*, int, float*, int64_t, const uint32_t*);
INST_P(float, float) INST_P(__half, __half) INST_P(__half, float) INST_P(int8_t, float) INST_P(uint8_t, float)
#undef INST_P

#define INST_T(TQ, TX)                                                                                              \
  template void pairwise_threshold_append<TQ, TX>(resources&, const TQ*, int64_t, int64_t, const TX*, int64_t, int64_t, \
                                                  int64_t, const float*, const float*, int, float*, int64_t*, int*, int, int, \
                                                  int64_t, int64_t, int64_t, const uint32_t*, int, int64_t, const float*);
INST_T(float, float) INST_T(__half, __half)
#undef INST_T

#define INST_A(TQ)                                                                                        \
  template void fused_l2_argmin<TQ>(resources&, const TQ*, int64_t, int64_t, const float*, int64_t,       \
                                    int64_t, const float*, uint32_t*, float*);
INST_A(float) INST_A(__half) INST_A(int8_t) INST_A(uint8_t)
#undef INST_A

}  // namespace cuvs_amd

// Test/bench hooks (not in the reference ABI): the fp32 distance GEMM and the fused argmin on device floats.
extern "C" __attribute__((visibility("default"))) int cuvsAmdPairwiseDistance(uintptr_t res, const float* q,
                                                                               int64_t m, const float* x, int64_t n,
                                                                               int64_t dim, int metric, float* out)
{
  using namespace cuvs_amd;
  return translate_exceptions([=] {
    auto& r = *as_res(res);
    dev_buf<float> qn(r, m), xn(r, n);
    if (metric != M_InnerProduct) {
      row_norms<float>(r, q, m, dim, dim, qn.data(), metric == M_CosineExpanded);
      row_norms<float>(r, x, n, dim, dim, xn.data(), metric == M_CosineExpanded);
    }
    pairwise_distance<float, float>(r, q, m, dim, x, n, dim, dim, qn.data(), xn.data(), metric, out, n);
  });
}

extern "C" __attribute__((visibility("default"))) int cuvsAmdFusedArgmin(uintptr_t res, const float* q, int64_t m,
                                                                          const float* centers, int64_t n, int64_t dim,
                                                                          uint32_t* labels)
{
  using namespace cuvs_amd;
  return translate_exceptions([=] {
    auto& r = *as_res(res);
    dev_buf<float> cn(r, n);
    row_norms<float>(r, centers, n, dim, dim, cn.data(), false);
    fused_l2_argmin<float>(r, q, m, dim, centers, n, dim, cn.data(), labels, nullptr);
  });
}

// Timing hook for the threshold-append tile kernel alone: thresholds that nothing beats, `reps` launches between two
// events. dbg != 0 switches parts of the main loop off (results are wrong; only the duration means anything).
extern "C" __attribute__((visibility("default"))) int cuvsAmdTileBench(uintptr_t res, int64_t m, int64_t n, int64_t dim,
                                                                         int dbg, int reps, float* ms_out)
{
  using namespace cuvs_amd;
  return translate_exceptions([=] {
    auto& r = *as_res(res);
    const int k = 10, cap = 16;
    dev_buf<float> q(r, (size_t)m * dim), x(r, (size_t)n * dim), qn(r, m), xn(r, n), bv(r, (size_t)m * (k + cap));
    dev_buf<int64_t> bi(r, (size_t)m * (k + cap));
    dev_buf<int> cnt(r, m);
    HIP_TRY(hipMemsetAsync(q.data(), 0, q.bytes(), r.stream));
    HIP_TRY(hipMemsetAsync(x.data(), 0, x.bytes(), r.stream));
    HIP_TRY(hipMemsetAsync(qn.data(), 0, qn.bytes(), r.stream));
    HIP_TRY(hipMemsetAsync(xn.data(), 0, xn.bytes(), r.stream));
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bv.data()), 0xBF800000, bv.size(), r.stream));  // thresholds -1: nothing passes
    HIP_TRY(hipMemsetAsync(cnt.data(), 0, cnt.bytes(), r.stream));
    char buf[16];
    snprintf(buf, sizeof(buf), "%d", dbg);
    setenv("CUVS_AMD_TILE_DBG", buf, 1);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int it = 0; it < reps + 1; ++it) {
      if (it == 1) HIP_TRY(hipEventRecord(e0, r.stream));
      pairwise_threshold_append<float, float>(r, q.data(), m, dim, x.data(), n, dim, dim, qn.data(), xn.data(), M_L2Expanded,
                                              bv.data(), bi.data(), cnt.data(), k, cap, 0, 0, n, nullptr, 0);
    }
    HIP_TRY(hipEventRecord(e1, r.stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms_out, e0, e1));
    *ms_out /= reps;
    unsetenv("CUVS_AMD_TILE_DBG");
    HIP_TRY(hipEventDestroy(e0));
    HIP_TRY(hipEventDestroy(e1));
  });
}

#include <cuvs/distance/pairwise_distance.h>

namespace {
// src: column-major [rows, cols] (element (r, c) at c * rows + r)  ->  dst: row-major [rows, cols]
template <typename T>
__global__ void colmajor_to_rowmajor_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t rows, int64_t cols)
{
  __shared__ T tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 tile rows per pass
  for (int j = ty; j < 64; j += 4) {                       // read along r (contiguous in src)
    const int64_t r = r0 + tx, c = c0 + j;
    if (r < rows && c < cols) tile[j][tx] = src[c * rows + r];
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {                       // write along c (contiguous in dst)
    const int64_t r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) dst[r * cols + c] = tile[tx][j];
  }
}

template <typename T>
void pairwise_typed(cuvs_amd::resources& res, const T* xp, int64_t m, const T* yp, int64_t n, int64_t dim, int metric,
                    float* out)
{
  using namespace cuvs_amd;
  const bool sq = metric == M_CosineExpanded;
  dev_buf<float> xn(res, m), yn(res, n);
  if (metric != M_InnerProduct) {
    row_norms<T>(res, xp, m, dim, dim, xn.data(), sq);
    row_norms<T>(res, yp, n, dim, dim, yn.data(), sq);
  }
  for (int64_t r0 = 0; r0 < m; r0 += 32768) {  // the distance kernel takes at most 65535 row tiles per launch
    const int64_t mr = std::min<int64_t>(32768, m - r0);
    pairwise_distance<T, T>(res, xp + r0 * dim, mr, dim, yp, n, dim, dim, xn.data() + r0, yn.data(), metric,
                            out + r0 * n, n);
  }
}

// Column-major inputs and output (c/src/distance/pairwise_distance.cpp:93-121): both operands are re-laid row-major,
// and the column-major [m, n] result is the row-major [n, m] matrix of distance(y, x) - the metrics built are symmetric.
template <typename T>
void pairwise_colmajor(cuvs_amd::resources& res, const T* xp, int64_t m, const T* yp, int64_t n, int64_t dim, int metric,
                       float* out)
{
  using namespace cuvs_amd;
  CUVS_EXPECTS(m <= int64_t(65535) * 64 && n <= int64_t(65535) * 64, "cuvsPairwiseDistance: too many rows for the column-major path");
  dev_buf<T> xr(res, (size_t)m * dim), yr(res, (size_t)n * dim);
  hipLaunchKernelGGL((colmajor_to_rowmajor_kernel<T>), dim3((unsigned)((dim + 63) / 64), (unsigned)((m + 63) / 64)),
                     dim3(256), 0, res.stream, xp, xr.data(), m, dim);
  hipLaunchKernelGGL((colmajor_to_rowmajor_kernel<T>), dim3((unsigned)((dim + 63) / 64), (unsigned)((n + 63) / 64)),
                     dim3(256), 0, res.stream, yp, yr.data(), n, dim);
  HIP_TRY(hipGetLastError());
  pairwise_typed<T>(res, yr.data(), n, xr.data(), m, dim, metric, out);
}
}  // namespace

extern "C" cuvsError_t cuvsPairwiseDistance(cuvsResources_t res_h, DLManagedTensor* x_tensor, DLManagedTensor* y_tensor,
                                            DLManagedTensor* dist_tensor, cuvsDistanceType metric, float metric_arg)
{
  using namespace cuvs_amd;
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    (void)metric_arg;
    CUVS_EXPECTS(x_tensor && y_tensor && dist_tensor, "null argument");
    auto& x = x_tensor->dl_tensor;
    auto& y = y_tensor->dl_tensor;
    auto& d = dist_tensor->dl_tensor;
    CUVS_EXPECTS(is_device_accessible(x) && is_device_accessible(y) && is_device_accessible(d),
                 "Inputs to cuvsPairwiseDistance must all have device compatible memory");
    CUVS_EXPECTS(x.dtype.code == y.dtype.code && x.dtype.bits == y.dtype.bits,
                 "Inputs to cuvsPairwiseDistance must all have the same dtype");
    CUVS_EXPECTS(x.ndim == 2 && y.ndim == 2 && d.ndim == 2, "Inputs to cuvsPairwiseDistance must be matrices");
    const bool c_order = is_c_contiguous(x) && is_c_contiguous(y) && is_c_contiguous(d);
    const bool f_order = is_f_contiguous(x) && is_f_contiguous(y) && is_f_contiguous(d);
    CUVS_EXPECTS(c_order || f_order,
                 "Inputs to cuvsPairwiseDistance must all have the same layout (row-major or col-major)");
    const int64_t m = x.shape[0], n = y.shape[0], dim = x.shape[1];
    CUVS_EXPECTS(y.shape[1] == dim && d.shape[0] == m && d.shape[1] == n, "cuvsPairwiseDistance: shape mismatch");
    CUVS_EXPECTS(dtype_is(d.dtype, kDLFloat, 32), "cuvsPairwiseDistance: distances must be float32");
    CUVS_EXPECTS(metric_supported((int)metric), "cuvsPairwiseDistance: unsupported metric %d", (int)metric);
    float* out = static_cast<float*>(dl_data(d));
    if (dtype_is(x.dtype, kDLFloat, 32)) {
      const float* xp = static_cast<const float*>(dl_data(x));
      const float* yp = static_cast<const float*>(dl_data(y));
      if (c_order) pairwise_typed<float>(res, xp, m, yp, n, dim, (int)metric, out);
      else pairwise_colmajor<float>(res, xp, m, yp, n, dim, (int)metric, out);
    } else if (dtype_is(x.dtype, kDLFloat, 16)) {
      const __half* xp = static_cast<const __half*>(dl_data(x));
      const __half* yp = static_cast<const __half*>(dl_data(y));
      if (c_order) pairwise_typed<__half>(res, xp, m, yp, n, dim, (int)metric, out);
      else pairwise_colmajor<__half>(res, xp, m, yp, n, dim, (int)metric, out);
    } else {
      CUVS_FAIL("Unsupported DLtensor dtype: %d and bits: %d", (int)x.dtype.code, (int)x.dtype.bits);
    }
  });
}

// Test hook (not part of the reference ABI): the coarse search's two forms of "distances of m queries to n rows, the k
// best per query" - grouped != 0: pairwise_distance_grouped + select_k_grouped (the default of the IVF coarse searches);
// grouped == 0: pairwise_distance + select_k (rounds 1-4) - so that tests/ can pin one against the other and against
// oracle/. metric 0 (L2Expanded) or 6 (InnerProduct). Returns 2 when the grouped form does not take the shape.
extern "C" __attribute__((visibility("default"))) int cuvsAmdPairwiseTopK(uintptr_t res_h, const float* q, int64_t m, const float* x,
                                                                         int64_t n, int64_t dim, int metric, int k, float* out_val,
                                                                         uint32_t* out_idx, int grouped)
{
  using namespace cuvs_amd;
  int rc = 1;
  const int ok = translate_exceptions([&] {
    resources& res = *as_res(res_h);
    CUVS_EXPECTS(metric == M_L2Expanded || metric == M_InnerProduct, "cuvsAmdPairwiseTopK: metric %d", metric);
    const bool ip = metric == M_InnerProduct;
    dev_buf<float> qn(res, ip ? 0 : m), xn(res, ip ? 0 : n);
    if (!ip) {
      row_norms<float>(res, q, m, dim, dim, qn.data(), false);
      row_norms<float>(res, x, n, dim, dim, xn.data(), false);
    }
    if (grouped != 0) {
      if (!select_k_grouped_ok(n, k)) { rc = 2; return; }
      const int64_t ldo = round_up(n, 128);
      dev_buf<float> d(res, (size_t)m * ldo);
      dev_buf<uint32_t> gk(res, (size_t)m * (ldo / 16));
      if (!pairwise_distance_grouped(res, q, m, dim, x, n, dim, dim, qn.data(), xn.data(), metric, d.data(), ldo, gk.data(), ldo / 16)) {
        rc = 2;
        return;
      }
      select_k_grouped(res, d.data(), ldo, gk.data(), ldo / 16, m, n, k, out_val, out_idx, !ip);
    } else {
      dev_buf<float> d(res, (size_t)m * n);
      pairwise_distance<float, float>(res, q, m, dim, x, n, dim, dim, qn.data(), xn.data(), metric, d.data(), n);
      select_k<uint32_t, uint32_t>(res, d.data(), nullptr, m, n, n, k, out_val, out_idx, !ip);
    }
    sync(res);
  });
  return ok ? rc : 0;
}
