// IVF-Flat on MI355X: index build/extend, list-major scan, C ABI (drop-in for c/src/neighbors/ivf_flat.cpp).
//
// Reference: cpp/src/neighbors/ivf_flat/ivf_flat_build.cuh (build :394-446, extend :163-390, interleave
// pattern ivf_flat.hpp:184-200), ivf_flat_search.cuh (search_impl :41-309), the scan kernel
// detail/jit_lto_kernels/interleaved_scan_impl.cuh:71-206 with load_and_compute_dist_impl.cuh:690-738 and
// metric_impl.cuh:12-49 (distance accumulated in dimension order with fma, unexpanded L2 for both L2
// variants, ivf_flat_interleaved_scan_jit.cuh:267-278).
//
// MI355X design (same schedule as ivf_pq_search.hip): all lists in one flat allocation, rows interleaved in
// groups of 64 (one wave64 lane per row, 16-byte chunks -> 1 KiB coalesced per wave load); (query, probe)
// pairs are grouped by list and 8 queries that probe the same list share one pass over its rows: the query
// tile of an item ([dim][8] fp32, written by a small pre-kernel) is read with wave-uniform addresses through
// the scalar cache, so the 8 fma chains per loaded element (4 packed fp32 add + 4 packed fp32 fma) cost no LDS
// cycles. Per-wave register top lists + shared k-th bounds, two-phase schedule and a per-tile early stop as in
// the PQ scan. Cosine = dot products + in-kernel row norms. In-list order is ascending source id.
#include "ivf_common.hpp"
#include "ivf_pq_scan3.hpp"

#include <chrono>
#include "serialize.hpp"
#include "npy_io.hpp"

#include <cuvs/neighbors/ivf_flat.h>

#include <algorithm>
#include <type_traits>
#include <cfloat>

namespace cuvs_amd {

void load_range_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim, int64_t r0,
                         int64_t cnt, float* out);
void load_gather_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim,
                          const uint32_t* d_ids, int64_t cnt, float* out);

struct ivf_flat_index {
  int metric   = 0;
  elem_t dtype = elem_t::f32;
  uint32_t n_lists = 0, dim = 0;
  uint32_t veclen = 4, n_chunks = 0;  // elements per 16-byte chunk; chunks per row
  int64_t size = 0, padded_rows = 0;
  dev_buf<float> centers;       // [n_lists, dim]
  dev_buf<float> center_norms;  // [n_lists] canonical |c|^2
  dev_buf<float> center_norms_sqrt;  // [n_lists] |c| (cosine only: the reference's center_norms for that metric)
  dev_buf<uint8_t> data;        // [padded_rows / 64, n_chunks, 64, 16 bytes]
  dev_buf<int64_t> indices;     // [padded_rows]
  dev_buf<uint32_t> list_sizes, list_offsets;
  std::vector<uint32_t> h_list_sizes, h_list_offsets;
  mutable flat3_cache scan3;    // fp16 residual copy for the matrix-core tail phase (ivf_pq_scan3.hip), built on first use
};

namespace {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kFlatThreads = 512;
constexpr int kFlatWaves   = kFlatThreads / 64;
constexpr int kFlatQPB     = 8;
constexpr int kStopEvery   = 4;  // early-stop test every 4 chunks of 16 bytes (power of two)

// out = in * mult (a power of two: exact). The matrix-core tail phase of int8 / uint8 indexes works on the RAW element values - the
// space the scan kernel's integer distances and the head phase's bounds live in - while queries and centres are kept in the
// reference's mapped space (utils::mapping<float>: x / 128, x / 256)
__global__ void scale_floats_kernel(const float* __restrict__ in, int64_t n, float mult, float* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * mult;
}

__global__ void strided_ids_kernel2(uint32_t* ids, int64_t n, int64_t stride)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = (uint32_t)(i * stride);
}

struct pack_args {
  const void* src;           // device rows [*, dim] of T
  const uint32_t* perm;      // sorted new-row ids (whole extend call); src row = perm[j] - src_row0 when src_is_batch
  const uint32_t* labels;
  const uint32_t* new_off;
  const uint32_t* old_sizes;
  const uint32_t* list_off;
  const int64_t* new_ids;
  int64_t id_base, j0, batch;
  int src_is_batch;          // 1: src holds rows j0.. in sorted order (host staging); 0: src is the full device array
  uint32_t dim, veclen, n_chunks;
  uint8_t* data;
  int64_t* indices;
};

// one thread per (sorted row, chunk): 16 bytes of the row -> interleaved slot
template <typename T>
__global__ void pack_rows_kernel(pack_args a)
{
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.batch * a.n_chunks) return;
  int64_t jb  = t / a.n_chunks;
  uint32_t ch = (uint32_t)(t % a.n_chunks);
  int64_t j   = a.j0 + jb;
  uint32_t row = a.perm[j];
  uint32_t L   = a.labels[row];
  int64_t fr   = (int64_t)a.list_off[L] + a.old_sizes[L] + (j - (int64_t)a.new_off[L]);
  const T* src = static_cast<const T*>(a.src) + (a.src_is_batch ? jb : (int64_t)row) * a.dim;
  alignas(16) T tmp[16 / sizeof(T)];
#pragma unroll
  for (uint32_t e = 0; e < 16 / sizeof(T); ++e) {
    uint32_t d = ch * a.veclen + e;
    tmp[e]     = d < a.dim ? src[d] : T(0);
  }
  size_t addr = (((size_t)(fr >> 6) * a.n_chunks + ch) * 64 + (size_t)(fr & 63)) * 16;
  *reinterpret_cast<uint4*>(a.data + addr) = *reinterpret_cast<const uint4*>(tmp);
  if (ch == 0) a.indices[fr] = a.new_ids ? a.new_ids[row] : a.id_base + (int64_t)row;
}

__global__ void relocate_flat_lists_kernel(const uint8_t* __restrict__ old_data, const int64_t* __restrict__ old_ids,
                                           const uint32_t* __restrict__ old_off, const uint32_t* __restrict__ old_sizes,
                                           const uint32_t* __restrict__ new_off, uint32_t n_chunks,
                                           uint8_t* __restrict__ data, int64_t* __restrict__ ids)
{
  const uint32_t L  = blockIdx.x;
  const uint32_t sz = old_sizes[L];
  const int64_t so = old_off[L], dn = new_off[L];
  for (uint32_t i = threadIdx.x; i < sz; i += blockDim.x) ids[dn + i] = old_ids[so + i];
  const size_t n16 = (size_t)((sz + 63) / 64) * n_chunks * 64;
  const uint4* s = reinterpret_cast<const uint4*>(old_data + (size_t)(so >> 6) * n_chunks * 1024);
  uint4* d       = reinterpret_cast<uint4*>(data + (size_t)(dn >> 6) * n_chunks * 1024);
  for (size_t i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
}

struct flat_scan_args {
  const work_item* items;
  const uint32_t* item_begin;  // device scalars: this launch covers items [*item_begin, *item_end); nullptr: from 0
  const uint32_t* item_end;
  uint32_t n_lists;            // item.list >= n_lists: tail-phase label of list item.list - n_lists
  const uint32_t* sorted_pairs;
  const void* queries;  // [n_queries, dim] of T (raw values)
  const float* qtiles;  // [n_items, dim_pad, QPB] fp32 query tile of every work item (flat_query_tiles_kernel)
  const uint8_t* data;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  float* out_d;
  uint32_t* out_i;
  uint32_t* query_kth;
  const uint32_t* filter_bits;  // optional bitset over source ids (1 keeps), sample_filter.cuh semantics
  const int64_t* indices;       // flat row -> source id (only read when filtering)
  uint32_t n_probes, dim, veclen, n_chunks, k;
  int is_ip;  // 0: L2, 1: inner product, 2: cosine
  float* all_scores;         // non-fused path (large k, ivf_common.hpp): [n_queries, scores_ld] score of every probed row
  uint32_t* all_rows;        //   flat row of every column (0xffffffff: nothing there)
  const uint32_t* pair_seg;  //   first column of each pair in its query's row
  size_t scores_ld;
  const uint32_t* run_if = nullptr;  // optional device word: the launch does nothing when it is zero (fallback pass decided on the device)
};

// Query tile of every work item, [dim_pad][QPB] fp32 in HBM. The scan kernel reads it with wave-uniform addresses,
// i.e. through the scalar cache into SGPRs: the 32 bytes of query values an element step needs then cost no LDS
// cycles at all (as LDS broadcast reads they occupied the LDS pipe exactly as long as the 8 packed VALU
// instructions of the step occupy a SIMD, and the two could never fully overlap).
template <typename T>
__global__ void flat_query_tiles_kernel(const work_item* __restrict__ items, const uint32_t* __restrict__ n_items,
                                        const uint32_t* __restrict__ sorted_pairs, const T* __restrict__ queries,
                                        uint32_t n_probes, uint32_t dim, uint32_t dim_pad, int with_norms,
                                        float* __restrict__ tiles)
{
  constexpr int QPB = kFlatQPB;
  const uint32_t w = blockIdx.x;
  if (w >= *n_items) return;
  const work_item item = items[w];
  __shared__ uint32_t qid[QPB];
  if (threadIdx.x < QPB)
    qid[threadIdx.x] = threadIdx.x < item.count ? sorted_pairs[item.first + threadIdx.x] / n_probes : 0xffffffffu;
  __syncthreads();
  float* out = tiles + (size_t)w * (dim_pad + 1) * QPB;  // rows 0..dim_pad-1: components; row dim_pad: |q| (cosine)
  for (uint32_t t = threadIdx.x; t < dim_pad * QPB; t += blockDim.x) {
    const uint32_t d = t / QPB, j = t % QPB;
    float v = 0.f;
    if (qid[j] != 0xffffffffu && d < dim) v = to_float(queries[(size_t)qid[j] * dim + d]);
    out[t] = v;
  }
  if (threadIdx.x < QPB) {
    // |q|: squares accumulated in dimension order with fma, like the row norms inside the scan
    float n2 = 0.f;
    if (with_norms && qid[threadIdx.x] != 0xffffffffu) {
      for (uint32_t d = 0; d < dim; ++d) {
        const float v = to_float(queries[(size_t)qid[threadIdx.x] * dim + d]);
        n2            = __fmaf_rn(v, v, n2);
      }
    }
    out[(size_t)dim_pad * QPB + threadIdx.x] = sqrtf(n2);
  }
}

// int8 / uint8 (L2, inner product): the tile holds the RAW query bytes, four per dword - [chunk][query][4 dwords] - and,
// behind them, the running sum of q^2 after every chunk - [chunk][query] - for v_dot4 (the reference's dp4a path,
// metric_impl.cuh:12-49: integer accumulators, exact).
template <typename T>
__global__ void flat_query_tiles_int_kernel(const work_item* __restrict__ items, const uint32_t* __restrict__ n_items,
                                            const uint32_t* __restrict__ sorted_pairs, const T* __restrict__ queries,
                                            uint32_t n_probes, uint32_t dim, uint32_t n_chunks, size_t tile_floats,
                                            float* __restrict__ tiles)
{
  constexpr int QPB = kFlatQPB;
  const uint32_t w = blockIdx.x;
  if (w >= *n_items) return;
  const work_item item = items[w];
  __shared__ uint32_t qid[QPB];
  if (threadIdx.x < QPB)
    qid[threadIdx.x] = threadIdx.x < item.count ? sorted_pairs[item.first + threadIdx.x] / n_probes : 0xffffffffu;
  __syncthreads();
  uint32_t* words = reinterpret_cast<uint32_t*>(tiles + (size_t)w * tile_floats);
  uint32_t* q2    = words + (size_t)n_chunks * QPB * 4;
  for (uint32_t t = threadIdx.x; t < n_chunks * QPB * 4; t += blockDim.x) {
    const uint32_t ch = t / (QPB * 4), j = (t / 4) % QPB, u = t % 4;
    uint32_t wv = 0u;
    if (qid[j] != 0xffffffffu) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t d = ch * 16 + u * 4 + b;
        if (d < dim) wv |= (uint32_t)(uint8_t)queries[(size_t)qid[j] * dim + d] << (8 * b);
      }
    }
    words[t] = wv;
  }
  if (threadIdx.x < QPB) {
    const uint32_t j = threadIdx.x;
    uint32_t acc = 0u;  // sum of squares: the same bits for int32 and uint32 accumulators
    for (uint32_t ch = 0; ch < n_chunks; ++ch) {
      if (qid[j] != 0xffffffffu)
        for (uint32_t d = ch * 16; d < ch * 16 + 16 && d < dim; ++d) {
          const int v = (int)queries[(size_t)qid[j] * dim + d];
          acc += (uint32_t)(v * v);
        }
      q2[ch * QPB + j] = acc;
    }
  }
}

// the compiler sinks every load of a group to its first use (one load, one wait, sixteen fmas, next load ...); an empty asm that
// takes all four as operands keeps them issued back to back
__device__ inline void keep_loads_together(uint4 (&w)[kStopEvery])
{
  static_assert(kStopEvery == 4, "four chunks per group");
  // (not volatile: an asm with unmodelled side effects counts as a memory clobber, and the wave-uniform loads of the query tile
  // stop being scalar loads)
  asm(""
      : "+v"(w[0].x), "+v"(w[0].y), "+v"(w[0].z), "+v"(w[0].w), "+v"(w[1].x), "+v"(w[1].y), "+v"(w[1].z), "+v"(w[1].w),
                 "+v"(w[2].x), "+v"(w[2].y), "+v"(w[2].z), "+v"(w[2].w), "+v"(w[3].x), "+v"(w[3].y), "+v"(w[3].z), "+v"(w[3].w));
}

// IP (inner product) is a template argument: tested at run time inside the unrolled element loop it became a
// scalar branch per element
// ALL: the non-fused path (every score written out, no top lists) - a template argument so that the fused kernels
// stay exactly as they were
template <typename T, int E, int METRIC, bool ALL = false>  // METRIC 0: L2, 1: inner product, 2: cosine (+ row norms)
__global__ __launch_bounds__(kFlatThreads) void ivf_flat_scan_kernel(flat_scan_args a)
{
  constexpr int QPB = kFlatQPB;
  constexpr int VL  = 16 / sizeof(T);
  constexpr bool IP = METRIC != 0;  // scores that are dot products: no early stop, negated as sort keys
  constexpr bool INT = (std::is_same_v<T, int8_t> || std::is_same_v<T, uint8_t>) && METRIC != 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (a.run_if != nullptr && *a.run_if == 0u) return;
  const uint32_t item0 = a.item_begin ? *a.item_begin : 0u;
  const uint32_t w     = item0 + blockIdx.x;
  if (w >= *a.item_end) return;
  const work_item item = a.items[w];

  const uint32_t dim_pad = a.n_chunks * VL;
  const size_t off = (((size_t)QPB * kFlatWaves * a.k * 8) + 15) & ~size_t(15);  // merge area
  uint32_t* kthb = reinterpret_cast<uint32_t*>(smem + off);
  uint32_t* pid  = kthb + 16;

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t L        = item.list >= a.n_lists ? item.list - a.n_lists : item.list;
  const uint32_t base_row = a.list_offsets[L];
  const uint32_t len      = a.list_sizes[L];

  if (tid < QPB) {
    const uint32_t p = tid < (int)item.count ? a.sorted_pairs[item.first + tid] : 0xffffffffu;
    pid[tid]         = p;
    kthb[tid]        = p != 0xffffffffu ? a.query_kth[p / a.n_probes] : 0u;
  }
  __syncthreads();
  // wave-uniform pointer: the loads below become s_load_dwordx8 (scalar cache -> SGPRs)
  const float* __restrict__ qt = a.qtiles + (size_t)w * (dim_pad + 1) * QPB;

  wave_top<E> top[QPB];
#pragma unroll
  for (int j = 0; j < QPB; ++j) top[j].init();
  const int kr          = (int)a.k - 1;
  uint32_t fresh        = 0xffu;  // bit j: this wave's list of query j is still empty (wave-uniform)
  const size_t g0       = (size_t)(base_row >> 6);
  const uint4* data16   = reinterpret_cast<const uint4*>(a.data);
  const uint32_t n_tile = (len + 63) / 64;

  for (uint32_t tile = wave; tile < n_tile; tile += kFlatWaves) {
    const uint32_t tile0 = tile * 64;
    const uint32_t v     = tile0 + lane;
    const bool valid     = v < len;
    // two queries per packed fp32 instruction (v_pk_add_f32 / v_pk_fma_f32): lane by lane the IEEE operations of
    // the scalar form; (q - x)^2 == (x - q)^2 exactly
    f32x2_t accv[QPB / 2];
#pragma unroll
    for (int j = 0; j < QPB / 2; ++j) accv[j] = f32x2_t{0.f, 0.f};
    float xn2 = 0.f;  // cosine: |x|^2 of this lane's row, accumulated next to the dot products (metric_impl.cuh)
    const uint4* cp = data16 + ((g0 + tile) * a.n_chunks) * 64 + lane;
    // early stop (L2): partial sums of squares only grow, so once every row of the tile is above the k-th bound of
    // every query of the item the rest of the row data is neither loaded nor accumulated. Bounds are read once
    // per tile and only decrease: rows dropped here are also rejected by the filter below.
    float bf[QPB];
#pragma unroll
    for (int j = 0; j < QPB; ++j) {
      const uint32_t kk = __builtin_amdgcn_readfirstlane(kthb[j]);
      bf[j] = (IP || j >= (int)item.count) ? -INFINITY : (kk >= 0xff800000u ? INFINITY : key_to_float(kk));
    }
    float acc[QPB];
    if constexpr (INT) {
      // integer path: sum (x - q)^2 = sum x^2 + sum q^2 - 2 sum x q, every term exact in 32-bit integers (v_dot4: four
      // elements per instruction; 36 instructions per 16-byte chunk and 8 queries instead of 144)
      using iacc_t = std::conditional_t<std::is_same_v<T, int8_t>, int32_t, uint32_t>;
      const uint32_t* qw = reinterpret_cast<const uint32_t*>(qt);  // wave-uniform: scalar loads
      const uint32_t* q2 = qw + (size_t)a.n_chunks * QPB * 4;
      iacc_t dot[QPB];
#pragma unroll
      for (int j = 0; j < QPB; ++j) dot[j] = 0;
      iacc_t sx2 = 0;
      auto dot4 = [](const uint32_t x, const uint32_t y, const iacc_t c) -> iacc_t {
        if constexpr (std::is_same_v<T, int8_t>) return __builtin_amdgcn_sdot4((int)x, (int)y, c, false);
        else return __builtin_amdgcn_udot4(x, y, c, false);
      };
      uint32_t done = 0;
      // the kStopEvery chunk loads between two early-stop tests are issued together (one load, one wait per chunk left the
      // wave with 1 KiB in flight: the head phase of C2 ran at half the HBM rate)
      for (uint32_t ch0 = 0; ch0 < a.n_chunks; ch0 += kStopEvery) {
        if (!IP && !ALL && ch0 > 0) {
          bool below = false;
#pragma unroll
          for (int j = 0; j < QPB; ++j)
            below = below || ((float)(iacc_t)(sx2 + (iacc_t)q2[(ch0 - 1) * QPB + j] - 2 * dot[j]) <= bf[j]);
          if (__ballot(valid && below) == 0ull) break;  // wave-uniform
        }
        uint4 cws[kStopEvery];
#pragma unroll
        for (int c = 0; c < kStopEvery; ++c) cws[c] = cp[(size_t)min(ch0 + (uint32_t)c, a.n_chunks - 1u) * 64];
        keep_loads_together(cws);
#pragma unroll
        for (int c = 0; c < kStopEvery; ++c) {
          const uint32_t ch = ch0 + (uint32_t)c;
          if (ch >= a.n_chunks) break;  // wave-uniform
          const uint32_t xw[4] = {cws[c].x, cws[c].y, cws[c].z, cws[c].w};
          const uint32_t* qr   = qw + (size_t)ch * QPB * 4;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!IP) sx2 = dot4(xw[u], xw[u], sx2);
#pragma unroll
            for (int j = 0; j < QPB; ++j) dot[j] = dot4(xw[u], qr[j * 4 + u], dot[j]);
          }
          done = ch + 1;
        }
      }
      // after an early stop the partial sums (all above every bound) stand in for the scores: rejected below
#pragma unroll
      for (int j = 0; j < QPB; ++j)
        acc[j] = IP ? (float)dot[j] : (float)(iacc_t)(sx2 + (iacc_t)q2[(done - 1) * QPB + j] - 2 * dot[j]);
    } else {
    auto chunk_step = [&](const uint4& cw, const uint32_t ch) {
      const T* el = reinterpret_cast<const T*>(&cw);
#pragma unroll
      for (int e = 0; e < VL; ++e) {
        const float x   = to_float(el[e]);
        const float* qr = qt + (size_t)(ch * VL + e) * QPB;
        const f32x2_t qv[QPB / 2] = {f32x2_t{qr[0], qr[1]}, f32x2_t{qr[2], qr[3]}, f32x2_t{qr[4], qr[5]}, f32x2_t{qr[6], qr[7]}};
        const f32x2_t xx = f32x2_t{x, x};
        if (METRIC == 2) xn2 = __fmaf_rn(x, x, xn2);
#pragma unroll
        for (int j = 0; j < QPB / 2; ++j) {
          if (!IP) {
            const f32x2_t t = qv[j] - xx;
            accv[j]         = __builtin_elementwise_fma(t, t, accv[j]);
          } else {
            accv[j] = __builtin_elementwise_fma(xx, qv[j], accv[j]);
          }
        }
      }
    };
    if constexpr (IP) {
      // dot products have no early stop: the rows stream through, four chunk loads in flight per lane (see the integer path;
      // measured at C2: inner product 12.98 -> 12.38 ms on this kernel alone, cosine 11.70 -> 11.20)
      for (uint32_t ch0 = 0; ch0 < a.n_chunks; ch0 += kStopEvery) {
        uint4 cws[kStopEvery];  // padded rows of a group are zero-filled: always readable
#pragma unroll
        for (int c = 0; c < kStopEvery; ++c) cws[c] = cp[(size_t)min(ch0 + (uint32_t)c, a.n_chunks - 1u) * 64];
        keep_loads_together(cws);
#pragma unroll
        for (int c = 0; c < kStopEvery; ++c) {
          const uint32_t ch = ch0 + (uint32_t)c;
          if (ch >= a.n_chunks) break;  // wave-uniform
          chunk_step(cws[c], ch);
        }
      }
    } else {
      // L2: one chunk at a time (the grouped form costs 20 registers - a workgroup less per CU - and the head phase of C2, which
      // is bound by the insertions under cold bounds, not by the loads, got 2 % slower with it)
      for (uint32_t ch = 0; ch < a.n_chunks; ++ch) {
        if (!ALL && ch > 0 && (ch & (kStopEvery - 1)) == 0) {
          bool below = false;
#pragma unroll
          for (int j = 0; j < QPB; ++j) below = below || (accv[j >> 1][j & 1] <= bf[j]);
          if (__ballot(valid && below) == 0ull) break;  // wave-uniform
        }
        chunk_step(cp[(size_t)ch * 64], ch);  // padded rows of a group are zero-filled: always readable
      }
    }
#pragma unroll
    for (int j = 0; j < QPB; ++j) acc[j] = accv[j >> 1][j & 1];
    }
    if (METRIC == 2) {
      // cos = dot / (|q| * |x|); reported later as 1 - cos (post_process_compose)
      const float xn = sqrtf(xn2);
#pragma unroll
      for (int j = 0; j < QPB; ++j) acc[j] = acc[j] / (qt[(size_t)dim_pad * QPB + j] * xn);
    }
#pragma unroll
    for (int j = 0; j < QPB; ++j) {
      if (j >= (int)item.count) break;
      const float dj       = IP ? -acc[j] : acc[j];  // smaller is better
      if constexpr (ALL) {  // non-fused path: every score goes to the query's row (filtered rows keep the fill)
        bool keep = valid;
        if (keep && a.filter_bits != nullptr) {
          const int64_t sid = a.indices[(size_t)base_row + v];
          keep              = (a.filter_bits[sid >> 5] >> (sid & 31)) & 1u;
        }
        const uint32_t p = pid[j];
        if (keep) {
          const size_t o  = (size_t)(p / a.n_probes) * a.scores_ld + a.pair_seg[p] + v;
          a.all_scores[o] = dj;
          a.all_rows[o]   = base_row + v;
        }
        continue;
      }
      const uint32_t bound = kthb[j];
      unsigned long long m = __ballot(valid && float_to_key(dj) <= bound);
      if (m == 0ull) continue;
      // the wave's first candidates of query j (its list is empty: cold bounds let nearly every row of the first tile through):
      // one sorting network instead of up to 64 serial insertions - the list that results is the same, entry for entry
      const bool first = ((fresh >> j) & 1u) != 0u;
      fresh &= ~(1u << j);
      if (first && a.filter_bits == nullptr && __popcll(m) >= 12 && __ballot(dj != dj) == 0ull) {
        const bool c = ((m >> lane) & 1ull) != 0ull;
        float sd     = c ? dj : INFINITY;
        uint32_t si  = c ? tile0 + (uint32_t)lane : 0xffffffffu;
        wave_sort64(sd, si, lane);
        top[j].d[0] = sd;
        top[j].i[0] = si;
        const float kd0 = top[j].rank_d(kr);
        if (lane == 0 && kd0 < INFINITY) atomicMin(&kthb[j], float_to_key(kd0));
        continue;
      }
      float kd      = top[j].rank_d(kr);
      uint32_t ki   = top[j].rank_i(kr);
      bool improved = false;
      while (m != 0ull) {
        const int src = (int)__ffsll((long long)m) - 1;
        m &= m - 1ull;
        const float cd    = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj), src));
        const uint32_t ci = tile0 + (uint32_t)src;
        if (a.filter_bits != nullptr) {  // pre-filter: rows whose source id is masked out never enter the top list
          const int64_t sid = a.indices[(size_t)base_row + ci];
          if (!((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u)) continue;
        }
        if ((cd < kd) || (cd == kd && ci < ki)) {
          top[j].insert(cd, ci, lane);
          kd       = top[j].rank_d(kr);
          ki       = top[j].rank_i(kr);
          improved = true;
        }
      }
      if (improved && lane == 0 && kd < INFINITY) atomicMin(&kthb[j], float_to_key(kd));
    }
  }

  if constexpr (ALL) return;
  // ---- merge the wave lists (the query tile is no longer needed)
  __syncthreads();
  if constexpr (E > 1) {
    // k > 64: the sorted wave lists are merged by the whole workgroup, one query after the other (ivf_common.hpp)
    constexpr int KP2 = E == 2 ? 128 : 256;
    float* sd    = reinterpret_cast<float*>(smem);
    uint32_t* si = reinterpret_cast<uint32_t*>(smem + (size_t)kFlatWaves * KP2 * 4);
    for (int j = 0; j < QPB; ++j) {
      if (j >= (int)item.count) break;  // workgroup-uniform
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int r = e * 64 + lane;
        const bool in = r < (int)a.k;
        sd[wave * KP2 + r] = in ? top[j].d[e] : INFINITY;
        si[wave * KP2 + r] = in ? top[j].i[e] : 0xffffffffu;
      }
      __syncthreads();
      merge_sorted_lists<kFlatThreads>(sd, si, kFlatWaves, KP2, tid);
      const size_t o = (size_t)pid[j] * a.k;
      for (int r = tid; r < (int)a.k; r += kFlatThreads) {
        const bool ok  = si[r] != 0xffffffffu;
        a.out_d[o + r] = ok ? sd[r] : FLT_MAX;
        a.out_i[o + r] = ok ? base_row + si[r] : 0xffffffffu;
      }
      if (tid == 0 && si[a.k - 1] != 0xffffffffu && sd[a.k - 1] < INFINITY)
        atomicMin(&a.query_kth[pid[j] / a.n_probes], float_to_key(sd[a.k - 1]));
      __syncthreads();  // the next query reuses the area
    }
    return;
  }
  float* mg_d    = reinterpret_cast<float*>(smem);
  uint32_t* mg_i = reinterpret_cast<uint32_t*>(smem + (size_t)QPB * kFlatWaves * a.k * 4);
#pragma unroll
  for (int j = 0; j < QPB; ++j) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int r = e * 64 + lane;
      if (r < (int)a.k) {
        mg_d[((size_t)j * kFlatWaves + wave) * a.k + r] = top[j].d[e];
        mg_i[((size_t)j * kFlatWaves + wave) * a.k + r] = top[j].i[e];
      }
    }
  }
  __syncthreads();
  if (wave < QPB && wave < (int)item.count) {
    const int j = wave;
    wave_top<E> fin;
    fin.init();
    float kd    = INFINITY;
    uint32_t ki = 0xffffffffu;
    const int n = kFlatWaves * (int)a.k;
    for (int b0 = 0; b0 < n; b0 += 64) {
      float md    = INFINITY;
      uint32_t mi = 0xffffffffu;
      if (b0 + lane < n) { md = mg_d[(size_t)j * n + b0 + lane]; mi = mg_i[(size_t)j * n + b0 + lane]; }
      unsigned long long m = __ballot(mi != 0xffffffffu && ((md < kd) || (md == kd && mi < ki)));
      while (m != 0ull) {
        const int src = (int)__ffsll((long long)m) - 1;
        m &= m - 1ull;
        const float cd    = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(md), src));
        const uint32_t ci = __builtin_amdgcn_readlane(mi, src);
        if ((cd < kd) || (cd == kd && ci < ki)) {
          fin.insert(cd, ci, lane);
          kd = fin.rank_d(kr);
          ki = fin.rank_i(kr);
        }
      }
    }
    const size_t o = (size_t)pid[j] * a.k;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      int r = e * 64 + lane;
      if (r < (int)a.k) {
        bool ok        = fin.i[e] != 0xffffffffu;
        a.out_d[o + r] = ok ? fin.d[e] : FLT_MAX;
        a.out_i[o + r] = ok ? base_row + fin.i[e] : 0xffffffffu;
      }
    }
    if (lane == 0 && kd < INFINITY) atomicMin(&a.query_kth[pid[j] / a.n_probes], float_to_key(kd));
  }
}

// the first `seg` slots of every query's candidate row (the head pairs' segments) start out "nothing found"
__global__ void flat_init_head_rows_kernel(float* __restrict__ cand_d, uint32_t* __restrict__ cand_i, int64_t nq, int64_t row_len, uint32_t seg)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * (int64_t)seg) return;
  const int64_t o = (t / seg) * row_len + t % seg;
  cand_d[o] = FLT_MAX;
  cand_i[o] = 0xffffffffu;
}

// the tail slots of every query's candidate row back to "nothing found" - only when *run_if is set (the matrix-core tail
// phase gave up: its pool entries must not be read as per-pair lists by the scan-kernel pass that follows)
__global__ void flat_reset_rows_if_kernel(const uint32_t* __restrict__ run_if, float* __restrict__ cand_d, uint32_t* __restrict__ cand_i,
                                          int64_t nq, int64_t row_len, uint32_t head_len)
{
  if (run_if != nullptr && *run_if == 0u) return;  // (nullptr: unconditional)
  const int64_t tail = row_len - head_len;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nq * tail; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = (t / tail) * row_len + head_len + t % tail;
    cand_d[o] = FLT_MAX;
    cand_i[o] = 0xffffffffu;
  }
}

__global__ void flat_postprocess_kernel(const uint32_t* __restrict__ pos, const float* __restrict__ d_in, int64_t n,
                                        const int64_t* __restrict__ indices, int metric,
                                        int64_t* __restrict__ neighbors, float* __restrict__ distances)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t p   = pos[i];
  neighbors[i] = p == 0xffffffffu ? INT64_MAX : indices[p];
  float d      = d_in[i];
  if (p == 0xffffffffu) d = FLT_MAX;
  else if (metric == M_InnerProduct) d = -d;
  else if (metric == M_CosineExpanded) d = 1.0f + d;  // d = -cos
  else if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) d = sqrtf(d);
  distances[i] = d;
}

__global__ void unpack_flat_list_kernel(const uint8_t* __restrict__ data, uint32_t n_chunks, uint32_t veclen,
                                        uint32_t esz, uint32_t dim, int64_t flat_row0, uint32_t n_rows,
                                        uint8_t* __restrict__ out)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_rows * dim) return;
  int64_t r  = i / dim;
  uint32_t d = (uint32_t)(i % dim);
  int64_t fr = flat_row0 + r;
  size_t addr = (((size_t)(fr >> 6) * n_chunks + d / veclen) * 64 + (size_t)(fr & 63)) * 16 + (size_t)(d % veclen) * esz;
  for (uint32_t b = 0; b < esz; ++b) out[i * esz + b] = data[addr + b];
}

template <typename T, int E, int METRIC, bool ALL = false>
void launch_flat_scan_kern(resources& res, const flat_scan_args& a, size_t smem, unsigned grid)
{
  auto kern = ivf_flat_scan_kernel<T, E, METRIC, ALL>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kFlatThreads), smem, res.stream, a);
}

template <typename T>
void launch_flat_scan(resources& res, const flat_scan_args& a, size_t smem, unsigned grid, bool big_k)
{
  if (a.all_scores != nullptr) {
    if (a.is_ip == 2)      launch_flat_scan_kern<T, 1, 2, true>(res, a, smem, grid);
    else if (a.is_ip == 1) launch_flat_scan_kern<T, 1, 1, true>(res, a, smem, grid);
    else                   launch_flat_scan_kern<T, 1, 0, true>(res, a, smem, grid);
  } else if (big_k) {
    if (a.is_ip == 2)      launch_flat_scan_kern<T, 4, 2>(res, a, smem, grid);
    else if (a.is_ip == 1) launch_flat_scan_kern<T, 4, 1>(res, a, smem, grid);
    else                   launch_flat_scan_kern<T, 4, 0>(res, a, smem, grid);
  } else {
    if (a.is_ip == 2)      launch_flat_scan_kern<T, 1, 2>(res, a, smem, grid);
    else if (a.is_ip == 1) launch_flat_scan_kern<T, 1, 1>(res, a, smem, grid);
    else                   launch_flat_scan_kern<T, 1, 0>(res, a, smem, grid);
  }
  HIP_TRY(hipGetLastError());
}

template <typename T>
void launch_pack(resources& res, const pack_args& a)
{
  int64_t total = a.batch * a.n_chunks;
  hipLaunchKernelGGL((pack_rows_kernel<T>), dim3(grid_blocks(total, 256)), dim3(256), 0, res.stream, a);
}

}  // namespace

void ivf_flat_extend(resources& res, ivf_flat_index& idx, const void* data, elem_t et, int64_t n_new, bool is_host,
                     const int64_t* new_ids, bool ids_on_host)
{
  if (n_new == 0) return;
  CUVS_EXPECTS(et == idx.dtype, "extend: vector dtype differs from the index dtype");
  CUVS_EXPECTS(new_ids != nullptr || idx.size == 0, "You must pass data indices when the index is non-empty.");
  CUVS_EXPECTS(idx.size + n_new < (int64_t(1) << 32) - 64 * (int64_t)idx.n_lists, "index too large for 32-bit row offsets");
  const int64_t dim = idx.dim;
  const size_t esz  = elem_size(et);
  dev_buf<int64_t> ids_dev;
  if (new_ids && ids_on_host) {
    ids_dev = dev_buf<int64_t>(res, n_new);
    copy_async(res, ids_dev.data(), new_ids, n_new * sizeof(int64_t));
    new_ids = ids_dev.data();
  }
  // labels: predicted with the index metric, like the k-means that trained the centres (ivf_flat_build.cuh:179-200,
  // :438): L2 argmin; inner product -> the centre with the largest dot product (argmin of |c|^2 - 2 x.c with the |c|^2
  // term dropped), which is also how the search ranks the probes; cosine -> L2 on unit-length copies
  dev_buf<uint32_t> labels(res, n_new);
  dev_buf<float> zero_norms(res, idx.metric == M_InnerProduct ? idx.n_lists : 0);
  if (idx.metric == M_InnerProduct) HIP_TRY(hipMemsetAsync(zero_norms.data(), 0, zero_norms.bytes(), res.stream));
  const float* label_norms = idx.metric == M_InnerProduct ? zero_norms.data() : idx.center_norms.data();
  const int64_t batch_rows = std::max<int64_t>(1024, std::min<int64_t>(n_new, (int64_t(1) << 28) / dim));
  {
    dev_buf<float> xb(res, (size_t)std::min(batch_rows, n_new) * dim);
    for (int64_t r0 = 0; r0 < n_new; r0 += batch_rows) {
      int64_t cnt = std::min(batch_rows, n_new - r0);
      load_range_as_float(res, data, et, is_host, dim, r0, cnt, xb.data());
      if (idx.metric == M_CosineExpanded)
        normalize_rows(res, xb.data(), cnt, dim);  // cosine: rows are assigned to lists on unit-length copies
      fused_l2_argmin<float>(res, xb.data(), cnt, dim, idx.centers.data(), idx.n_lists, dim, label_norms,
                             labels.data() + r0, nullptr);
    }
  }
  dev_buf<uint32_t> perm(res, n_new), new_off(res, idx.n_lists + 1);
  group_by_label(res, labels.data(), n_new, idx.n_lists, perm.data(), new_off.data());
  std::vector<uint32_t> h_new_off = to_host(res, new_off.data(), idx.n_lists + 1);
  std::vector<uint32_t> sizes(idx.n_lists), offs(idx.n_lists + 1);
  int64_t total = 0;
  for (uint32_t L = 0; L < idx.n_lists; ++L) {
    sizes[L] = idx.h_list_sizes[L] + (h_new_off[L + 1] - h_new_off[L]);
    offs[L]  = (uint32_t)total;
    total += round_up(sizes[L], 64);
  }
  offs[idx.n_lists] = (uint32_t)total;
  auto ndata    = dev_buf<uint8_t>::persistent((size_t)total * idx.n_chunks * 16);
  auto nindices = dev_buf<int64_t>::persistent((size_t)total);
  HIP_TRY(hipMemsetAsync(ndata.data(), 0, ndata.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(nindices.data(), 0xff, nindices.bytes(), res.stream));
  dev_buf<uint32_t> d_list_off(res, idx.n_lists + 1);
  copy_async(res, d_list_off.data(), offs.data(), offs.size() * sizeof(uint32_t));
  if (idx.size > 0) {
    hipLaunchKernelGGL(relocate_flat_lists_kernel, dim3(idx.n_lists), dim3(256), 0, res.stream, idx.data.data(),
                       idx.indices.data(), idx.list_offsets.data(), idx.list_sizes.data(), d_list_off.data(),
                       idx.n_chunks, ndata.data(), nindices.data());
  }
  pack_args a;
  a.perm = perm.data(); a.labels = labels.data(); a.new_off = new_off.data(); a.old_sizes = idx.list_sizes.data();
  a.list_off = d_list_off.data(); a.new_ids = new_ids; a.id_base = idx.size;
  a.dim = idx.dim; a.veclen = idx.veclen; a.n_chunks = idx.n_chunks; a.data = ndata.data(); a.indices = nindices.data();
  auto launch = [&](const pack_args& pa) {
    switch (et) {
      case elem_t::f32: launch_pack<float>(res, pa); break;
      case elem_t::f16: launch_pack<__half>(res, pa); break;
      case elem_t::i8: launch_pack<int8_t>(res, pa); break;
      case elem_t::u8: launch_pack<uint8_t>(res, pa); break;
    }
  };
  const int64_t pb = std::max<int64_t>(64, (int64_t(1) << 22));  // rows per pack launch
  if (!is_host) {
    a.src = data; a.src_is_batch = 0;
    for (int64_t j0 = 0; j0 < n_new; j0 += pb) {
      a.j0 = j0; a.batch = std::min(pb, n_new - j0);
      launch(a);
    }
  } else {
    std::vector<uint32_t> h_perm = to_host(res, perm.data(), n_new);
    const int64_t hb = std::max<int64_t>(1, std::min<int64_t>(n_new, (int64_t(1) << 28) / (dim * (int64_t)esz)));
    std::vector<char> host((size_t)hb * dim * esz);
    dev_buf<char> stage(res, host.size());
    const char* src = static_cast<const char*>(data);
    for (int64_t j0 = 0; j0 < n_new; j0 += hb) {
      int64_t cnt = std::min(hb, n_new - j0);
      for (int64_t i = 0; i < cnt; ++i)
        memcpy(host.data() + (size_t)i * dim * esz, src + (size_t)h_perm[j0 + i] * dim * esz, dim * esz);
      copy_async(res, stage.data(), host.data(), (size_t)cnt * dim * esz);
      a.src = stage.data(); a.src_is_batch = 1; a.j0 = j0; a.batch = cnt;
      launch(a);
      sync(res);
    }
  }
  HIP_TRY(hipGetLastError());
  sync(res);
  idx.data    = std::move(ndata);
  idx.indices = std::move(nindices);
  copy_async(res, idx.list_sizes.data(), sizes.data(), sizes.size() * sizeof(uint32_t));
  copy_async(res, idx.list_offsets.data(), offs.data(), offs.size() * sizeof(uint32_t));
  sync(res);
  idx.h_list_sizes   = sizes;
  idx.h_list_offsets = offs;
  idx.size += n_new;
  idx.padded_rows = total;
}

static void flat_set_center_norms(resources& res, ivf_flat_index& idx)
{
  idx.center_norms = dev_buf<float>::persistent(idx.n_lists);
  row_norms<float>(res, idx.centers.data(), idx.n_lists, idx.dim, idx.dim, idx.center_norms.data(), false);
  if (idx.metric == M_CosineExpanded) {
    idx.center_norms_sqrt = dev_buf<float>::persistent(idx.n_lists);
    row_norms<float>(res, idx.centers.data(), idx.n_lists, idx.dim, idx.dim, idx.center_norms_sqrt.data(), true);
  }
}

std::unique_ptr<ivf_flat_index> ivf_flat_build(resources& res, const cuvsIvfFlatIndexParams& p, const void* data,
                                               elem_t et, int64_t n, int64_t dim, bool is_host)
{
  CUVS_EXPECTS(n > 0 && dim > 0, "empty dataset");
  CUVS_EXPECTS(n >= p.n_lists, "number of rows can't be less than n_lists");
  const int metric = (int)p.metric;
  CUVS_EXPECTS(metric_is_l2(metric) || metric == M_InnerProduct || metric == M_CosineExpanded,
               "ivf_flat: unsupported metric %d (L2, inner product and cosine are built)", metric);
  CUVS_EXPECTS(metric != M_CosineExpanded || dim > 1, "Cosine metric requires more than one dim");
  auto idx      = std::make_unique<ivf_flat_index>();
  idx->metric   = metric;
  idx->dtype    = et;
  idx->n_lists  = p.n_lists;
  idx->dim      = (uint32_t)dim;
  idx->veclen   = (uint32_t)(16 / elem_size(et));
  idx->n_chunks = (uint32_t)ceil_div(dim, idx->veclen);
  idx->list_sizes   = dev_buf<uint32_t>::persistent(p.n_lists);
  idx->list_offsets = dev_buf<uint32_t>::persistent(p.n_lists + 1);
  HIP_TRY(hipMemsetAsync(idx->list_sizes.data(), 0, idx->list_sizes.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(idx->list_offsets.data(), 0, idx->list_offsets.bytes(), res.stream));
  idx->h_list_sizes.assign(p.n_lists, 0);
  idx->h_list_offsets.assign(p.n_lists + 1, 0);
  // trainset: every ratio-th row (ivf_flat_build.cuh:409-425 uses a strided subsample as well)
  const int64_t ratio   = std::max<int64_t>(1, n / std::max<int64_t>((int64_t)(p.kmeans_trainset_fraction * n), p.n_lists));
  const int64_t n_train = n / ratio;
  dev_buf<float> trainset(res, (size_t)n_train * dim);
  {
    dev_buf<uint32_t> ids(res, n_train);
    hipLaunchKernelGGL(strided_ids_kernel2, dim3(grid_blocks(n_train, 256)), dim3(256), 0, res.stream, ids.data(),
                       n_train, ratio);
    load_gather_as_float(res, data, et, is_host, dim, ids.data(), n_train, trainset.data());
  }
  idx->centers      = dev_buf<float>::persistent((size_t)p.n_lists * dim);
  kmeans_params kp;
  kp.n_iters       = (int)p.kmeans_n_iters;
  kp.inner_product = metric == M_InnerProduct;  // the index metric drives the clustering (ivf_flat_build.cuh:188)
  if (metric == M_CosineExpanded)
    normalize_rows(res, trainset.data(), n_train, dim);
  kmeans_balanced_fit(res, trainset.data(), n_train, dim, (int)p.n_lists, kp, idx->centers.data());
  flat_set_center_norms(res, *idx);
  trainset.release();
  if (p.add_data_on_build) ivf_flat_extend(res, *idx, data, et, n, is_host, nullptr, false);
  sync(res);
  return idx;
}

void ivf_flat_search(resources& res, const ivf_flat_index& idx, uint32_t n_probes_in, const void* queries, elem_t et,
                     int64_t n_queries, int k, int64_t* neighbors, float* distances, const uint32_t* filter_bits)
{
  CUVS_EXPECTS(k > 0, "ivf_flat::search: k must be positive");
  // (k beyond the index size is served with padded slots, as in the reference: ivf_flat_search.cuh has no such check)
  CUVS_EXPECTS(n_probes_in > 0, "n_probes must be positive");
  CUVS_EXPECTS(et == idx.dtype, "queries dtype differs from the index dtype");
  if (n_queries == 0) return;
  const uint32_t n_probes = std::min<uint32_t>(n_probes_in, idx.n_lists);
  const int qpb           = kFlatQPB;
  const bool large_k      = k > 256;  // beyond the register top lists: non-fused path (ivf_common.hpp)
  const bool big_k        = k > 64 && !large_k;
  const int k_scan        = large_k ? 1 : k;
  const uint32_t dim_pad  = idx.n_chunks * idx.veclen;
  size_t smem = ((((size_t)qpb * kFlatWaves * k_scan * 8) + 15) & ~size_t(15)) + 2 * 16 * 4;  // (>= 16 KiB at k > 64: the workgroup merge's 8 x 256 entries)
  const size_t scores_ld  = large_k ? largest_lists_total(idx.h_list_sizes, n_probes) : 0;

  // the matrix-core tail phase serves this index: beyond 256 dimensions fp32 / fp16 rows only (the re-score's fp32 chain over int8 /
  // uint8 rows equals the scan kernel's integer sum while every partial sum stays below 2^24: up to 256 dimensions)
  const bool flat3_ok = flat3_supported(idx.dim, k) && (idx.dim <= 256 || et == elem_t::f32 || et == elem_t::f16);
  int64_t max_batch = 1 << 15;
  {
    int64_t per_q = (int64_t)idx.n_lists * 4 + (int64_t)n_probes * k_scan * 8 + idx.dim * 4 +
                    ((int64_t)n_probes / qpb + 1) * (dim_pad + 1) * qpb * 4;  // + the query tiles of its work items
    if (large_k) per_q += (int64_t)scores_ld * 8 + (int64_t)k * 12;
    // the matrix-core tail phase: fp16 B operands + thresholds + norms + candidate rows of every pair, the raw / unit copies of
    // the query, the survivor list (16 entries per pair) and work units; the coarse search's grouped distances + keys
    if (!large_k && flat3_ok && res.tune.flat_scan3 != 0)
      per_q += (int64_t)n_probes * ((int64_t)dim_pad * 2 + 4 + 16 + (int64_t)k * 4 + 16 * 8 + 16) + (int64_t)idx.dim * 8;
    per_q += round_up((int64_t)idx.n_lists, 128) * 4 + round_up((int64_t)idx.n_lists, 128) / 4;
    max_batch     = balanced_batch(n_queries, std::min(max_batch, std::max<int64_t>(1, (int64_t)res.ivf_batch_limit / per_q)));
  }
  const int64_t bs = std::min<int64_t>(max_batch, n_queries);
  const int64_t np_max = bs * n_probes;
  host_trace trace((res.tune.scan_debug & 8192) != 0, "ivf_flat_search");  // declared before the buffers: destroyed after them
  dev_buf<float> qf(res, (size_t)bs * idx.dim), qn(res, bs), pd(res, (size_t)np_max);
  dev_buf<float> dist;  // [bs, n_lists] coarse distances of the plain form: allocated by the first batch that does not take the grouped form
  // two-phase schedule (ivf_common.hpp): nearest probe of every query first
  // (inner product, round 5: a head phase only where the matrix-core tail phase follows it - the scan kernel has no early stop
  // for dot products, but the filter prunes on the full-score bound the head phase leaves)
  const bool cos3 = idx.metric == M_CosineExpanded;
  const bool ip3 = (idx.metric == M_InnerProduct || cos3) && !large_k && n_queries >= 256 && flat3_ok &&
                   res.tune.flat_scan3 != 0;
  uint32_t head = (n_probes > 8 && (metric_is_l2(idx.metric) || ip3) && !large_k) ? 1u : 0u;
  if (res.tune.flat_head_probes >= 0) head = std::min<uint32_t>((uint32_t)res.tune.flat_head_probes, n_probes);
  const uint32_t n_labels = head > 0 ? 2 * idx.n_lists : idx.n_lists;
  dev_buf<uint32_t> probes(res, np_max), sorted_pairs(res, np_max), pair_off(res, n_labels + 1),
    item_off(res, n_labels + 1), cand_i(res, large_k ? (size_t)bs * scores_ld : (size_t)np_max * k), top_i(res, (size_t)bs * k),
    query_kth(res, bs), pair_seg(res, large_k ? (size_t)np_max : 0);
  dev_buf<uint32_t> phase_labels(res, head > 0 ? (size_t)np_max : 0);
  const size_t max_items = (size_t)(np_max / qpb + n_labels + 1);
  dev_buf<work_item> items(res, max_items);
  dev_buf<float> qtiles(res, max_items * (dim_pad + 1) * qpb);
  dev_buf<float> cand_d(res, large_k ? (size_t)bs * scores_ld : (size_t)np_max * k), top_d(res, (size_t)bs * k);
  const size_t esz = elem_size(et);
  // warm-bounds phase on the matrix cores (ivf_pq_scan3.hip): every row type (int8 / uint8 values and the distances between them
  // are exact in fp16 / fp32 up to dim 256), L2, batches large enough for a head phase
  const bool use3 = head > 0 && (metric_is_l2(idx.metric) || ip3) && !large_k && n_queries >= 256 &&
                    flat3_ok && res.tune.flat_scan3 != 0;
  uint32_t max_list_len = 0;
  for (uint32_t v : idx.h_list_sizes) max_list_len = std::max(max_list_len, v);
  const uint32_t unit_rows = std::max<uint32_t>(4096u, (uint32_t)round_up((int64_t)(max_list_len + 15) / 16, 64));
  const size_t max_units   = use3 ? (size_t)16 * ((size_t)np_max / 32 + idx.n_lists + 1) : 0;
  uint32_t surv_cap        = use3 ? (uint32_t)std::min<int64_t>(std::max<int64_t>(np_max * 16, 1 << 22), 1 << 28) : 0u;
  if (use3 && res.tune.pq3_surv_cap > 0) surv_cap = (uint32_t)res.tune.pq3_surv_cap;
  const uint32_t overflow_cap = use3 ? (res.tune.pq3_surv_cap > 0 ? (uint32_t)res.tune.pq3_surv_cap : (1u << 22)) : 0u;
  dev_buf<uint32_t> cand_r(res, use3 ? (size_t)np_max * k : 0), qstate(res, use3 ? (size_t)4 * bs + 8 + pq3_regions(res) + 1 : 0);
  dev_buf<uint32_t> unit_off(res, use3 ? (size_t)idx.n_lists + 1 : 0), tickets3(res, use3 ? 8 * 32 : 0);
  dev_buf<uint2> surv(res, surv_cap);
  dev_buf<uint4> units3(res, 2 * max_units), overflow3(res, (size_t)2 * overflow_cap);
  // flat_filter2_kernel's pre-pass: fp16 B operand (2 x dim bytes) and threshold of every pair
  const bool f2 = use3 && res.tune.flat_filter2 != 0 && idx.dim <= 128;
  const bool fw = use3 && flat3_wide(idx.dim);  // the wide filter (ivf_pq_wide.hip): operands in blocks of 32 pairs per list
  dev_buf<uint4> bq3(res, f2 ? (size_t)np_max * (idx.dim / 8) : fw ? ((size_t)np_max + (size_t)32 * (idx.n_lists + 1)) * (idx.dim / 8) : 0);
  dev_buf<float> thr3(res, (f2 || fw) ? (size_t)np_max : 0);
  // Bound-only head phase (round 6, L2 family without a bitset filter; CUVS_AMD_FLAT_BOUND_HEAD=0: the exact head phase on the scan
  // kernel): the nearest list of every query is screened through its fp16 copy like the other 63 - the head pass only has to leave
  // an upper bound of the query's k-th best score (ivf_pq_scan3.hpp: flat3_head_bounds)
  // (measured and rejected for the wide filter's dimensions: the emit build + select + the k exact rows against the exact head phase on
  // the scan kernel - 1M x 768 fp16: 3.63 vs 3.50 ms, 512: 2.60 vs 2.54, 256: 2.06 vs 1.92; the emit pass reads the whole copy for ~10
  // queries per list)
  const bool bound_head = f2 && head == 1 && metric_is_l2(idx.metric) && filter_bits == nullptr && res.tune.flat_bound_head != 0;
  // (rows of at least 4096 values: select_k's one-read kernel serves them; the padding is -inf)
  const uint32_t hb_ldx = (uint32_t)std::max<int64_t>(4096, round_up((int64_t)max_list_len + 64, 64));
  dev_buf<float> hb_x(res, bound_head ? (size_t)bs * hb_ldx : 0), hb_kv(res, bound_head ? (size_t)bs * k : 0), hb_thr(res, bound_head ? (size_t)bs : 0);
  dev_buf<uint32_t> hb_ki(res, bound_head ? (size_t)bs * k : 0), hb_tk(res, bound_head ? 8 * 32 : 0);
  dev_buf<float4> hb_nm(res, bound_head ? (size_t)bs : 0);
  const float raw_mult = et == elem_t::i8 ? 128.0f : et == elem_t::u8 ? 256.0f : 1.0f;
  const bool raw3      = use3 && raw_mult != 1.0f;
  dev_buf<float> q_unit(res, use3 && cos3 ? (size_t)bs * idx.dim : 0);  // cosine: unit-length queries for the filter
  dev_buf<float> q_raw(res, raw3 ? (size_t)bs * idx.dim : 0), c_raw(res, raw3 ? (size_t)idx.n_lists * idx.dim : 0);
  if (raw3)
    hipLaunchKernelGGL(scale_floats_kernel, dim3(grid_blocks((int64_t)c_raw.n, 256)), dim3(256), 0, res.stream, idx.centers.data(),
                       (int64_t)c_raw.n, raw_mult, c_raw.data());
  trace.mark("buffers allocated");

  for (int64_t q0 = 0; q0 < n_queries; q0 += max_batch) {
    const int64_t nq      = std::min(max_batch, n_queries - q0);
    const int64_t n_pairs = nq * n_probes;
    const char* qptr      = static_cast<const char*>(queries) + (size_t)q0 * idx.dim * esz;
    load_range_as_float(res, queries, et, false, idx.dim, q0, nq, qf.data());
    // coarse search (ivf_flat_search.cuh:104-187). Common shapes: distances in grouped layout + the best key of every 16-centre
    // group, selection from the keys and the qualifying groups only (ops.hpp: pairwise_distance_grouped / select_k_grouped,
    // DESIGN 3.1f) - the values, their order and the tie rule are those of the plain form below
    bool coarse_done = false;
    if (res.tune.coarse_grouped != 0 && select_k_grouped_ok(idx.n_lists, (int)n_probes)) {
      const int64_t ldo = round_up((int64_t)idx.n_lists, 128);
      dev_buf<float> gdist(res, (size_t)nq * ldo);
      dev_buf<uint32_t> gkeys(res, (size_t)nq * (ldo / 16));
      const bool ipm = idx.metric == M_InnerProduct, cosm = idx.metric == M_CosineExpanded;
      if (!ipm) row_norms<float>(res, qf.data(), nq, idx.dim, idx.dim, qn.data(), cosm);
      coarse_done = pairwise_distance_grouped(res, qf.data(), nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim, idx.dim,
                                              ipm ? nullptr : qn.data(), ipm ? nullptr : (cosm ? idx.center_norms_sqrt.data() : idx.center_norms.data()),
                                              ipm ? (int)M_InnerProduct : (cosm ? (int)M_CosineExpanded : (int)M_L2Expanded), gdist.data(), ldo,
                                              gkeys.data(), ldo / 16);
      if (coarse_done) select_k_grouped(res, gdist.data(), ldo, gkeys.data(), ldo / 16, nq, idx.n_lists, (int)n_probes, pd.data(), probes.data(), !ipm);
    }
    if (!coarse_done && dist.data() == nullptr) dist = dev_buf<float>(res, (size_t)bs * idx.n_lists);
    if (coarse_done) {
    } else if (idx.metric == M_InnerProduct) {
      pairwise_distance<float, float>(res, qf.data(), nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim, idx.dim,
                                      nullptr, nullptr, M_InnerProduct, dist.data(), idx.n_lists);
      select_k<uint32_t, uint32_t>(res, dist.data(), nullptr, nq, idx.n_lists, idx.n_lists, (int)n_probes, pd.data(),
                                   probes.data(), false);
    } else if (idx.metric == M_CosineExpanded) {
      // 1 - q.c / (|q| |c|): the same ranking as the reference's -q.c / (|q| |c|) (ivf_flat_search.cuh:130-175)
      row_norms<float>(res, qf.data(), nq, idx.dim, idx.dim, qn.data(), true);
      pairwise_distance<float, float>(res, qf.data(), nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim, idx.dim,
                                      qn.data(), idx.center_norms_sqrt.data(), M_CosineExpanded, dist.data(), idx.n_lists);
      select_k<uint32_t, uint32_t>(res, dist.data(), nullptr, nq, idx.n_lists, idx.n_lists, (int)n_probes, pd.data(),
                                   probes.data(), true);
    } else {
      row_norms<float>(res, qf.data(), nq, idx.dim, idx.dim, qn.data(), false);
      pairwise_distance<float, float>(res, qf.data(), nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim, idx.dim,
                                      qn.data(), idx.center_norms.data(), M_L2Expanded, dist.data(), idx.n_lists);
      select_k<uint32_t, uint32_t>(res, dist.data(), nullptr, nq, idx.n_lists, idx.n_lists, (int)n_probes, pd.data(),
                                   probes.data(), true);
    }
    trace.mark("coarse search");
    const uint32_t* labels = probes.data();
    if (head > 0) {
      hipLaunchKernelGGL(phase_labels_kernel, dim3(grid_blocks(n_pairs, 256)), dim3(256), 0, res.stream, probes.data(),
                         n_pairs, n_probes, head, idx.n_lists, phase_labels.data());
      labels = phase_labels.data();
    }
    build_work_items(res, labels, n_pairs, n_labels, qpb, sorted_pairs.data(), pair_off.data(), item_off.data(),
                     items.data());
    trace.mark("work items");
    HIP_TRY(hipMemsetAsync(query_kth.data(), 0xff, (size_t)nq * sizeof(uint32_t), res.stream));
    if (use3) {
      // only the head segments of a query's candidate row are read before they are written (the pool behind them is filled by
      // count; the guarded fallback pass resets the tail itself): no fill of all n_pairs x k slots
      hipLaunchKernelGGL(flat_init_head_rows_kernel, dim3(grid_blocks(nq * (int64_t)head * k, 256)), dim3(256), 0, res.stream,
                         cand_d.data(), cand_i.data(), nq, (int64_t)n_probes * k, (uint32_t)(head * k));
      HIP_TRY(hipMemsetAsync(qstate.data(), 0, qstate.bytes(), res.stream));
      HIP_TRY(hipMemsetAsync(tickets3.data(), 0, tickets3.bytes(), res.stream));
    }
    if (large_k) {
      HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cand_d.data()), 0x7f7fffff, (size_t)nq * scores_ld, res.stream));
      HIP_TRY(hipMemsetAsync(cand_i.data(), 0xff, (size_t)nq * scores_ld * sizeof(uint32_t), res.stream));
      hipLaunchKernelGGL(pair_segments_kernel, dim3(grid_blocks(nq, 256)), dim3(256), 0, res.stream, probes.data(),
                         idx.list_sizes.data(), nq, n_probes, pair_seg.data());
    }
    flat_scan_args a;
    a.all_scores = large_k ? cand_d.data() : nullptr; a.all_rows = cand_i.data(); a.pair_seg = pair_seg.data(); a.scores_ld = scores_ld;
    a.items = items.data(); a.sorted_pairs = sorted_pairs.data(); a.n_lists = idx.n_lists;
    {
      const uint32_t* n_all = item_off.data() + n_labels;
      const int with_norms  = idx.metric == M_CosineExpanded;
      const unsigned g      = (unsigned)(n_pairs / qpb + n_labels + 1);
      switch (et) {
        case elem_t::f32: hipLaunchKernelGGL(flat_query_tiles_kernel<float>, dim3(g), dim3(256), 0, res.stream, items.data(), n_all, sorted_pairs.data(), reinterpret_cast<const float*>(qptr), n_probes, idx.dim, dim_pad, with_norms, qtiles.data()); break;
        case elem_t::f16: hipLaunchKernelGGL(flat_query_tiles_kernel<__half>, dim3(g), dim3(256), 0, res.stream, items.data(), n_all, sorted_pairs.data(), reinterpret_cast<const __half*>(qptr), n_probes, idx.dim, dim_pad, with_norms, qtiles.data()); break;
        case elem_t::i8:
          if (!with_norms) hipLaunchKernelGGL(flat_query_tiles_int_kernel<int8_t>, dim3(g), dim3(256), 0, res.stream, items.data(), n_all, sorted_pairs.data(), reinterpret_cast<const int8_t*>(qptr), n_probes, idx.dim, idx.n_chunks, (size_t)(dim_pad + 1) * qpb, qtiles.data());
          else hipLaunchKernelGGL(flat_query_tiles_kernel<int8_t>, dim3(g), dim3(256), 0, res.stream, items.data(), n_all, sorted_pairs.data(), reinterpret_cast<const int8_t*>(qptr), n_probes, idx.dim, dim_pad, with_norms, qtiles.data());
          break;
        case elem_t::u8:
          if (!with_norms) hipLaunchKernelGGL(flat_query_tiles_int_kernel<uint8_t>, dim3(g), dim3(256), 0, res.stream, items.data(), n_all, sorted_pairs.data(), reinterpret_cast<const uint8_t*>(qptr), n_probes, idx.dim, idx.n_chunks, (size_t)(dim_pad + 1) * qpb, qtiles.data());
          else hipLaunchKernelGGL(flat_query_tiles_kernel<uint8_t>, dim3(g), dim3(256), 0, res.stream, items.data(), n_all, sorted_pairs.data(), reinterpret_cast<const uint8_t*>(qptr), n_probes, idx.dim, dim_pad, with_norms, qtiles.data());
          break;
      }
      a.qtiles = qtiles.data();
    }
    a.queries = qptr; a.data = idx.data.data(); a.list_offsets = idx.list_offsets.data();
    a.list_sizes = idx.list_sizes.data(); a.out_d = cand_d.data(); a.out_i = cand_i.data();
    a.filter_bits = filter_bits; a.indices = idx.indices.data();
    a.query_kth = query_kth.data(); a.n_probes = n_probes; a.dim = idx.dim; a.veclen = idx.veclen;
    a.n_chunks = idx.n_chunks; a.k = (uint32_t)k_scan; a.is_ip = idx.metric == M_InnerProduct ? 1 : (idx.metric == M_CosineExpanded ? 2 : 0);
    auto launch = [&](const flat_scan_args& fa, unsigned grid) {
      profile_begin(res, "ivf_flat_scan_kernel");
      switch (et) {
        case elem_t::f32: launch_flat_scan<float>(res, fa, smem, grid, big_k); break;
        case elem_t::f16: launch_flat_scan<__half>(res, fa, smem, grid, big_k); break;
        case elem_t::i8: launch_flat_scan<int8_t>(res, fa, smem, grid, big_k); break;
        case elem_t::u8: launch_flat_scan<uint8_t>(res, fa, smem, grid, big_k); break;
      }
      profile_end(res, "ivf_flat_scan_kernel");
    };
    trace.mark("memsets + query tiles");
    // grids are upper bounds of the (device-side) item counts of each phase; surplus workgroups exit at once
    bool merged = false;
    if (head > 0) {
      // the bound-only head phase needs the tail phase's run description (pairs, queries in the tail's space): set up below, the exact
      // head launch happens there when the bound-only pass is not taken
      bool head_done = false;
      auto exact_head = [&]() {
        a.item_begin = nullptr; a.item_end = item_off.data() + idx.n_lists;
        launch(a, (unsigned)(nq * head / qpb + idx.n_lists + 1));
        trace.mark("head launch");
        a.item_begin = item_off.data() + idx.n_lists; a.item_end = item_off.data() + 2 * idx.n_lists;
        head_done = true;
      };
      if (!(use3 && bound_head)) exact_head();
      if (use3) {
        pq3_run r{};
        r.nq = nq; r.n_probes = n_probes; r.k = (uint32_t)k; r.head = head; r.is_ip = (idx.metric == M_InnerProduct || cos3) ? 1 : 0;
        r.sorted_pairs = sorted_pairs.data(); r.pair_off = pair_off.data(); r.probes = probes.data();
        if (raw3)
          hipLaunchKernelGGL(scale_floats_kernel, dim3(grid_blocks(nq * (int64_t)idx.dim, 256)), dim3(256), 0, res.stream, qf.data(),
                             nq * (int64_t)idx.dim, raw_mult, q_raw.data());
        r.rot_queries = raw3 ? q_raw.data() : qf.data(); r.query_kth = query_kth.data();
        if (cos3) {
          // cos = q^ . x^ with x^ = c + (x^ - c): the filter's inner-product form on unit-length queries and rows (scale-free:
          // the centres are means of unit-length rows); the exact chain works on the values the scan kernel reads
          HIP_TRY(hipMemcpyAsync(q_unit.data(), qf.data(), (size_t)nq * idx.dim * sizeof(float), hipMemcpyDeviceToDevice, res.stream));
          normalize_rows(res, q_unit.data(), nq, idx.dim);
          r.rescore_queries = r.rot_queries;
          r.rot_queries     = q_unit.data();
          r.cosine          = 1;
        }
        r.cand_d = cand_d.data(); r.cand_i = cand_i.data(); r.cand_r = cand_r.data();
        r.qflag = qstate.data(); r.qcnt = qstate.data() + bs; r.counters = qstate.data() + 2 * bs;
        r.surv_cnt = qstate.data() + 2 * bs + 2;
        r.ov_cnt = qstate.data() + 2 * bs + 4 + pq3_regions(res); r.ov_off = r.ov_cnt + bs;
        r.fail = qstate.data() + 4 * bs + 8 + pq3_regions(res);
        r.surv = surv.data(); r.surv_cap = surv_cap; r.units = units3.data(); r.unit_off = unit_off.data();
        r.unit_rows = unit_rows; r.xcd_ticket = tickets3.data(); r.filter_bits = filter_bits;
        r.overflow = overflow3.data(); r.overflow_cap = overflow_cap;
        r.bq = (f2 || fw) ? bq3.data() : nullptr; r.thr = thr3.data();
        r.filter_dbg = (res.tune.scan_debug >> 16) & 255;  // CUVS_AMD_SCAN_DEBUG bits 16..23: ablations of the filter (timing only)
        flat3_view v{idx.data.data(), (raw3 && !cos3) ? c_raw.data() : idx.centers.data(), idx.list_offsets.data(), idx.list_sizes.data(), idx.indices.data(),
                     idx.n_lists, idx.dim, idx.n_chunks, idx.padded_rows, idx.size, max_list_len,
                     et == elem_t::f32 ? 0 : et == elem_t::f16 ? 1 : et == elem_t::i8 ? 2 : 3, cos3};
        const bool tdbg = (res.tune.scan_debug & 1024) != 0;
        auto now = [&]() { if (tdbg) sync(res); return std::chrono::steady_clock::now(); };
        bool hb_used = false;
        if (!head_done) {
          HIP_TRY(hipMemsetAsync(hb_tk.data(), 0, hb_tk.bytes(), res.stream));
          flat3_head_bufs hb{hb_x.data(), hb_ldx, hb_kv.data(), hb_ki.data(), hb_thr.data(), hb_nm.data(), hb_tk.data()};
          if (flat3_head_bounds(res, v, idx.scan3, r, hb)) {
            r.hb_xbuf = hb_x.data(); r.hb_ldx = hb_ldx; r.hb_thr = hb_thr.data();
            a.item_begin = item_off.data() + idx.n_lists; a.item_end = item_off.data() + 2 * idx.n_lists;
            hb_used = true;
            trace.mark("bound-only head enqueued");
          } else {
            exact_head();  // (no room for the fp16 copy)
          }
        }
        const auto t0 = now();
        if (flat3_tail(res, v, idx.scan3, r)) {
          trace.mark("tail enqueued");
          const auto t1 = now();
          pq3_merge(res, r, top_d.data(), top_i.data());
          const auto t2 = now();
          if (tdbg)
            fprintf(stderr, "[flat3] tail %.3f ms, merge %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count());
          // a buffer ran over (bounds too loose to filter: e.g. lists shorter than k): the tail phase again, on the scan
          // kernel + select_k - launched now, decided ON THE DEVICE (the word r.fail guards every kernel of the pass: the
          // call stays asynchronous, no flag comes back to the host). The pass starts from clean per-pair rows.
          trace.mark("merge enqueued");
          hipLaunchKernelGGL(flat_reset_rows_if_kernel, dim3(1024), dim3(256), 0, res.stream, r.fail, cand_d.data(), cand_i.data(),
                             (int64_t)nq, (int64_t)n_probes * k, (uint32_t)(head * k));
          a.run_if = r.fail;
          if (hb_used) {  // the head pairs' lists as well: the bound-only head phase left bounds, not candidates
            flat_scan_args ah = a;
            ah.item_begin = nullptr; ah.item_end = item_off.data() + idx.n_lists;
            launch(ah, (unsigned)(nq * head / qpb + idx.n_lists + 1));
          }
          launch(a, (unsigned)(nq * (n_probes - head) / qpb + idx.n_lists + 1));
          a.run_if = nullptr;
          select_k<uint32_t, uint32_t>(res, cand_d.data(), cand_i.data(), nq, (int64_t)n_probes * k, (int64_t)n_probes * k,
                                       k, top_d.data(), top_i.data(), true, 0, -1, 0, r.fail);
          merged = true;
          trace.mark("guarded fallback enqueued");
        }
      }
      if (!merged) {
        if (use3) {  // (no room for the fp16 copy: the plain tail phase needs every pair's row "nothing found")
          hipLaunchKernelGGL(flat_reset_rows_if_kernel, dim3(1024), dim3(256), 0, res.stream, static_cast<const uint32_t*>(nullptr),
                             cand_d.data(), cand_i.data(), (int64_t)nq, (int64_t)n_probes * k, (uint32_t)(head * k));
        }
        launch(a, (unsigned)(nq * (n_probes - head) / qpb + idx.n_lists + 1));
      }
    } else {
      a.item_begin = nullptr; a.item_end = item_off.data() + idx.n_lists;
      launch(a, (unsigned)(n_pairs / qpb + idx.n_lists + 1));
    }
    if (merged) {
      // pq3_merge: head lists + pools
    } else if (!large_k) {
      select_k<uint32_t, uint32_t>(res, cand_d.data(), cand_i.data(), nq, (int64_t)n_probes * k, (int64_t)n_probes * k,
                                   k, top_d.data(), top_i.data(), true);
    } else {
      select_k<uint32_t, uint32_t>(res, cand_d.data(), cand_i.data(), nq, (int64_t)scores_ld, (int64_t)scores_ld, k,
                                   top_d.data(), top_i.data(), true);
    }
    hipLaunchKernelGGL(flat_postprocess_kernel, dim3(grid_blocks(nq * k, 256)), dim3(256), 0, res.stream, top_i.data(),
                       top_d.data(), nq * k, idx.indices.data(), idx.metric, neighbors + q0 * k, distances + q0 * k);
    trace.mark("select + post-process");
  }
  HIP_TRY(hipGetLastError());
}

// mg.hip: rows held and metric of the index behind a cuvsIvfFlatIndex handle (the C ABI has no getter for them)
void ivf_flat_index_info(uintptr_t addr, int64_t* size, int* metric)
{
  CUVS_EXPECTS(addr != 0, "ivf_flat index is empty");
  auto* idx = reinterpret_cast<const ivf_flat_index*>(addr);
  *size     = idx->size;
  *metric   = idx->metric;
}

}  // namespace cuvs_amd

using namespace cuvs_amd;

namespace {
ivf_flat_index& get_flat(cuvsIvfFlatIndex_t index)
{
  CUVS_EXPECTS(index != nullptr && index->addr != 0, "IVF-Flat index is not built");
  return *reinterpret_cast<ivf_flat_index*>(index->addr);
}
}  // namespace

extern "C" {

cuvsError_t cuvsIvfFlatIndexParamsCreate(cuvsIvfFlatIndexParams_t* params)
{
  return (cuvsError_t)translate_exceptions(
    [=] { *params = new cuvsIvfFlatIndexParams{L2Expanded, 2.0f, true, 1024, 20, 0.5, false, false}; });
}
cuvsError_t cuvsIvfFlatIndexParamsDestroy(cuvsIvfFlatIndexParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsIvfFlatSearchParamsCreate(cuvsIvfFlatSearchParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] { *params = new cuvsIvfFlatSearchParams{20}; });
}
cuvsError_t cuvsIvfFlatSearchParamsDestroy(cuvsIvfFlatSearchParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsIvfFlatIndexCreate(cuvsIvfFlatIndex_t* index)
{
  return (cuvsError_t)translate_exceptions([=] { *index = new cuvsIvfFlatIndex{0, DLDataType{0, 0, 0}}; });
}
cuvsError_t cuvsIvfFlatIndexDestroy(cuvsIvfFlatIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (!index) return;
    delete reinterpret_cast<ivf_flat_index*>(index->addr);
    delete index;
  });
}
cuvsError_t cuvsIvfFlatIndexGetNLists(cuvsIvfFlatIndex_t index, int64_t* n_lists)
{
  return (cuvsError_t)translate_exceptions([=] { *n_lists = get_flat(index).n_lists; });
}
cuvsError_t cuvsIvfFlatIndexGetDim(cuvsIvfFlatIndex_t index, int64_t* dim)
{
  return (cuvsError_t)translate_exceptions([=] { *dim = get_flat(index).dim; });
}
cuvsError_t cuvsIvfFlatIndexGetCenters(cuvsIvfFlatIndex_t index, DLManagedTensor* centers)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_flat(index);
    fill_dl_view(centers, idx.centers.data(), DLDataType{kDLFloat, 32, 1}, idx.n_lists, idx.dim, 2, 0);
  });
}

cuvsError_t cuvsIvfFlatBuild(cuvsResources_t res_h, cuvsIvfFlatIndexParams_t params, DLManagedTensor* dataset_tensor,
                             cuvsIvfFlatIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(params && dataset_tensor && index, "null argument");
    auto& ds = dataset_tensor->dl_tensor;
    CUVS_EXPECTS(ds.ndim == 2 && is_c_contiguous(ds), "dataset must be a row-major matrix");
    auto idx = ivf_flat_build(res, *params, dl_data(ds), elem_of(ds.dtype), ds.shape[0], ds.shape[1],
                              !is_device_accessible(ds));
    delete reinterpret_cast<ivf_flat_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = ds.dtype;
  });
}

cuvsError_t cuvsIvfFlatSearch(cuvsResources_t res_h, cuvsIvfFlatSearchParams_t params, cuvsIvfFlatIndex_t index_c,
                              DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                              DLManagedTensor* distances_tensor, cuvsFilter filter)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_flat(index_c);
    CUVS_EXPECTS(params && queries_tensor && neighbors_tensor && distances_tensor, "null argument");
    const uint32_t* bits = nullptr;
    if (filter.type != NO_FILTER) {
      CUVS_EXPECTS(filter.type == BITSET && filter.addr != 0, "cuvsIvfFlatSearch: only BITSET filters are supported");
      auto& ft = reinterpret_cast<DLManagedTensor*>(filter.addr)->dl_tensor;
      CUVS_EXPECTS(dtype_is(ft.dtype, kDLUInt, 32) && is_device_accessible(ft), "filter must be a device uint32 tensor");
      bits = static_cast<const uint32_t*>(dl_data(ft));
    }
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
    CUVS_EXPECTS(queries.dtype.code == index_c->dtype.code && queries.dtype.bits == index_c->dtype.bits,
                 "Unsupported queries DLtensor dtype: %d and bits: %d", (int)queries.dtype.code,
                 (int)queries.dtype.bits);
    CUVS_EXPECTS(queries.shape[1] == idx.dim, "queries dim %ld != index dim %u", (long)queries.shape[1], idx.dim);
    int64_t m = queries.shape[0], k = neighbors.shape[1];
    CUVS_EXPECTS(neighbors.shape[0] == m && distances.shape[0] == m && distances.shape[1] == k,
                 "neighbors/distances shape mismatch");
    ivf_flat_search(res, idx, params->n_probes, dl_data(queries), elem_of(queries.dtype), m, (int)k,
                    static_cast<int64_t*>(dl_data(neighbors)), static_cast<float*>(dl_data(distances)), bits);
  });
}

cuvsError_t cuvsIvfFlatExtend(cuvsResources_t res_h, DLManagedTensor* new_vectors, DLManagedTensor* new_indices,
                              cuvsIvfFlatIndex_t index_c)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_flat(index_c);
    CUVS_EXPECTS(new_vectors != nullptr, "new_vectors is null");
    auto& v = new_vectors->dl_tensor;
    CUVS_EXPECTS(v.ndim == 2 && is_c_contiguous(v) && v.shape[1] == idx.dim, "new_vectors must be [n, dim] row-major");
    const int64_t* ids = nullptr;
    bool ids_host      = false;
    if (new_indices != nullptr) {
      auto& t = new_indices->dl_tensor;
      CUVS_EXPECTS(dtype_is(t.dtype, kDLInt, 64) && t.shape[0] == v.shape[0], "new_indices must be int64 [n]");
      ids      = static_cast<const int64_t*>(dl_data(t));
      ids_host = !is_device_accessible(t);
    }
    ivf_flat_extend(res, idx, dl_data(v), elem_of(v.dtype), v.shape[0], !is_device_accessible(v), ids, ids_host);
  });
}

}  // extern "C"

namespace {
constexpr int kFlatRefVersion = 4;  // ivf_flat_serialize.cuh:30

// the reference's interleave inside one list record [rows32, dim] (ivf_flat.hpp:184-200): groups of 32 rows,
// chunks of `vr` consecutive components; element offset of (row r, component d):
inline size_t ref_flat_offset(uint32_t r, uint32_t d, uint32_t dim, uint32_t vr)
{
  return (size_t)(r / 32) * 32 * dim + (size_t)(d / vr) * 32 * vr + (size_t)(r % 32) * vr + d % vr;
}
// ivf_flat.hpp:284-294
inline uint32_t ref_flat_veclen(uint32_t dim, size_t es)
{
  uint32_t v = std::max<uint32_t>(1, (uint32_t)(16 / es));
  return dim % v != 0 ? 1 : v;
}

void flat_write_native(resources& res, const char* filename, const ivf_flat_index& idx, DLDataType dl)
{
  file_writer w(filename, KIND_IVF_FLAT);
  w.scalar<int32_t>(idx.metric); w.scalar<int32_t>((int)idx.dtype); w.scalar<uint32_t>(idx.n_lists);
  w.scalar<uint32_t>(idx.dim); w.scalar<uint32_t>(idx.veclen); w.scalar<uint32_t>(idx.n_chunks);
  w.scalar<int64_t>(idx.size); w.scalar<int64_t>(idx.padded_rows);
  w.scalar<uint8_t>(dl.code); w.scalar<uint8_t>(dl.bits);
  w.device_array(res, idx.centers.data(), idx.centers.bytes());
  w.device_array(res, idx.center_norms.data(), idx.center_norms.bytes());
  w.device_array(res, idx.list_sizes.data(), idx.list_sizes.bytes());
  w.device_array(res, idx.list_offsets.data(), idx.list_offsets.bytes());
  w.device_array(res, idx.data.data(), idx.data.bytes());
  w.device_array(res, idx.indices.data(), idx.indices.bytes());
}

std::unique_ptr<ivf_flat_index> flat_read_native(resources& res, const char* filename, DLDataType* dl)
{
  file_reader r(filename, KIND_IVF_FLAT);
  auto idx = std::make_unique<ivf_flat_index>();
  idx->metric = r.scalar<int32_t>(); idx->dtype = (elem_t)r.scalar<int32_t>(); idx->n_lists = r.scalar<uint32_t>();
  idx->dim = r.scalar<uint32_t>(); idx->veclen = r.scalar<uint32_t>(); idx->n_chunks = r.scalar<uint32_t>();
  idx->size = r.scalar<int64_t>(); idx->padded_rows = r.scalar<int64_t>();
  uint8_t code = r.scalar<uint8_t>(), bits = r.scalar<uint8_t>();
  idx->centers      = r.device_array<float>(res);
  idx->center_norms = r.device_array<float>(res);
  idx->list_sizes   = r.device_array<uint32_t>(res);
  idx->list_offsets = r.device_array<uint32_t>(res);
  idx->data         = r.device_array<uint8_t>(res);
  idx->indices      = r.device_array<int64_t>(res);
  idx->h_list_sizes   = to_host(res, idx->list_sizes.data(), idx->n_lists);
  idx->h_list_offsets = to_host(res, idx->list_offsets.data(), idx->n_lists + 1);
  if (idx->metric == M_CosineExpanded) {
    idx->center_norms_sqrt = dev_buf<float>::persistent(idx->n_lists);
    row_norms<float>(res, idx->centers.data(), idx->n_lists, idx->dim, idx->dim, idx->center_norms_sqrt.data(), true);
  }
  *dl = DLDataType{code, bits, 1};
  return idx;
}

// Reference record sequence (ivf_flat_serialize.cuh:42-77, ivf_list.cuh:108-131): dtype prefix, version, size,
// dim, n_lists, metric, adaptive_centers, conservative_memory_allocation, centers, has_norms [, center_norms],
// list_sizes, then per list: rows32 = roundUp(size, 32) [, data [rows32, dim] in the reference's 32-row
// interleave, ids [rows32]]. Our lists are stored in 64-row groups, so each list is re-interleaved on the host.
void flat_write_ref(resources& res, const char* filename, const ivf_flat_index& idx)
{
  npy_writer w(filename);
  char prefix[4];
  elem_prefix(idx.dtype, prefix);
  w.raw(prefix, 4);
  w.scalar<int32_t>(kFlatRefVersion);
  w.scalar<int64_t>(idx.size);
  w.scalar<uint32_t>(idx.dim);
  w.scalar<uint32_t>(idx.n_lists);
  w.scalar<int32_t>(idx.metric);
  w.scalar<bool>(false);  // adaptive_centers
  w.scalar<bool>(true);   // conservative_memory_allocation: lists hold no slack beyond their group padding
  w.device_array(res, 'f', 4, {idx.n_lists, idx.dim}, idx.centers.data());
  const bool has_norms = idx.metric != M_InnerProduct;  // ivf_flat_index.cpp:179-191
  w.scalar<bool>(has_norms);
  if (has_norms)
    w.device_array(res, 'f', 4, {idx.n_lists},
                   idx.metric == M_CosineExpanded ? idx.center_norms_sqrt.data() : idx.center_norms.data());
  w.host_array<uint32_t>(idx.h_list_sizes.data(), {idx.n_lists});
  const size_t es   = elem_size(idx.dtype);
  const uint32_t vr = ref_flat_veclen(idx.dim, es);
  const char kind   = idx.dtype == elem_t::f32 ? 'f' : idx.dtype == elem_t::f16 ? 'e' : idx.dtype == elem_t::i8 ? 'i' : 'u';
  std::vector<uint8_t> ours, theirs;
  std::vector<int64_t> ids;
  for (uint32_t L = 0; L < idx.n_lists; ++L) {
    const uint32_t size = idx.h_list_sizes[L], rows32 = (uint32_t)round_up(size, 32);
    w.scalar<uint32_t>(rows32);
    if (rows32 == 0) continue;
    const uint32_t cap = idx.h_list_offsets[L + 1] - idx.h_list_offsets[L];
    ours.resize((size_t)cap * idx.n_chunks * 16);
    ids.assign(rows32, -1);
    copy_async(res, ours.data(), idx.data.data() + (size_t)idx.h_list_offsets[L] * idx.n_chunks * 16, ours.size());
    copy_async(res, ids.data(), idx.indices.data() + idx.h_list_offsets[L], (size_t)size * sizeof(int64_t));
    sync(res);
    theirs.assign((size_t)rows32 * idx.dim * es, 0);
    for (uint32_t r = 0; r < size; ++r) {
      const uint8_t* row = ours.data() + ((size_t)(r / 64) * idx.n_chunks * 64 + r % 64) * 16;
      for (uint32_t d = 0; d < idx.dim; ++d)
        memcpy(theirs.data() + ref_flat_offset(r, d, idx.dim, vr) * es,
               row + (size_t)(d / idx.veclen) * 64 * 16 + (d % idx.veclen) * es, es);
    }
    w.header(kind, (uint32_t)es, {rows32, idx.dim});
    w.raw(theirs.data(), theirs.size());
    w.host_array<int64_t>(ids.data(), {rows32});
  }
  w.close();
}

std::unique_ptr<ivf_flat_index> flat_read_ref(resources& res, const char* filename, DLDataType* dl)
{
  npy_reader r(filename);
  auto idx = std::make_unique<ivf_flat_index>();
  char prefix[4];
  r.raw(prefix, 4);
  CUVS_EXPECTS(parse_elem_prefix(prefix, &idx->dtype), "Unsupported dtype in file %s", filename);
  int ver = r.scalar<int32_t>();
  CUVS_EXPECTS(ver == kFlatRefVersion, "serialization version mismatch, expected %d, got %d ", kFlatRefVersion, ver);
  idx->size    = r.scalar<int64_t>();
  idx->dim     = r.scalar<uint32_t>();
  idx->n_lists = r.scalar<uint32_t>();
  idx->metric  = r.scalar<int32_t>();
  (void)r.scalar<bool>();  // adaptive_centers
  (void)r.scalar<bool>();  // conservative_memory_allocation
  CUVS_EXPECTS(metric_is_l2(idx->metric) || idx->metric == M_InnerProduct || idx->metric == M_CosineExpanded,
               "ivf_flat::deserialize: unsupported metric value %d", idx->metric);
  CUVS_EXPECTS(idx->dim > 0 && idx->n_lists > 0 && idx->n_lists <= (1u << 24), "ivf_flat::deserialize: bad header");
  const size_t es = elem_size(idx->dtype);
  idx->veclen     = (uint32_t)(16 / es);
  idx->n_chunks   = (uint32_t)ceil_div((int64_t)idx->dim, (int64_t)idx->veclen);
  idx->centers    = r.device_array<float>(res, (int64_t)idx->n_lists * idx->dim);
  if (r.scalar<bool>()) (void)r.host_array<float>(idx->n_lists);
  // canonical |c|^2 (the build's own rounding) rather than the file's
  flat_set_center_norms(res, *idx);
  idx->h_list_sizes = r.host_array<uint32_t>(idx->n_lists);
  idx->h_list_offsets.assign(idx->n_lists + 1, 0);
  int64_t total = 0, live = 0;
  for (uint32_t L = 0; L < idx->n_lists; ++L) {
    idx->h_list_offsets[L] = (uint32_t)total;
    total += round_up(idx->h_list_sizes[L], 64);
    live += idx->h_list_sizes[L];
  }
  CUVS_EXPECTS(total < (int64_t(1) << 32), "ivf_flat::deserialize: index too large");
  CUVS_EXPECTS(live == idx->size, "ivf_flat::deserialize: list sizes (%ld) do not add up to the index size (%ld)",
               (long)live, (long)idx->size);
  idx->h_list_offsets[idx->n_lists] = (uint32_t)total;
  idx->padded_rows                  = total;
  idx->list_sizes   = dev_buf<uint32_t>::persistent(idx->n_lists);
  idx->list_offsets = dev_buf<uint32_t>::persistent(idx->n_lists + 1);
  copy_async(res, idx->list_sizes.data(), idx->h_list_sizes.data(), idx->n_lists * sizeof(uint32_t));
  copy_async(res, idx->list_offsets.data(), idx->h_list_offsets.data(), (idx->n_lists + 1) * sizeof(uint32_t));
  idx->data    = dev_buf<uint8_t>::persistent((size_t)total * idx->n_chunks * 16);
  idx->indices = dev_buf<int64_t>::persistent((size_t)total);
  HIP_TRY(hipMemsetAsync(idx->data.data(), 0, idx->data.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(idx->indices.data(), 0xff, idx->indices.bytes(), res.stream));
  sync(res);
  const uint32_t vr = ref_flat_veclen(idx->dim, es);
  std::vector<uint8_t> ours;
  std::vector<char> theirs;
  for (uint32_t L = 0; L < idx->n_lists; ++L) {
    const uint32_t rows = r.scalar<uint32_t>(), size = idx->h_list_sizes[L];
    CUVS_EXPECTS(rows >= size, "ivf_flat::deserialize: list %u holds %u rows, list_sizes says %u", L, rows, size);
    if (rows == 0) continue;
    r.array((uint32_t)es, (int64_t)rows * idx->dim, theirs);
    std::vector<int64_t> ids = r.host_array<int64_t>(rows);
    if (size == 0) continue;
    const uint32_t cap = (uint32_t)round_up(size, 64);
    ours.assign((size_t)cap * idx->n_chunks * 16, 0);
    for (uint32_t rr = 0; rr < size; ++rr) {
      uint8_t* row = ours.data() + ((size_t)(rr / 64) * idx->n_chunks * 64 + rr % 64) * 16;
      for (uint32_t d = 0; d < idx->dim; ++d)
        memcpy(row + (size_t)(d / idx->veclen) * 64 * 16 + (d % idx->veclen) * es,
               theirs.data() + ref_flat_offset(rr, d, idx->dim, vr) * es, es);
    }
    copy_async(res, idx->data.data() + (size_t)idx->h_list_offsets[L] * idx->n_chunks * 16, ours.data(), ours.size());
    copy_async(res, idx->indices.data() + idx->h_list_offsets[L], ids.data(), (size_t)size * sizeof(int64_t));
    sync(res);
  }
  *dl = dl_of(idx->dtype);
  return idx;
}
}  // namespace

extern "C" {
cuvsError_t cuvsIvfFlatSerialize(cuvsResources_t res_h, const char* filename, cuvsIvfFlatIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_flat(index);
    if (write_native_container(res)) flat_write_native(res, filename, idx, index->dtype);
    else flat_write_ref(res, filename, idx);
  });
}
cuvsError_t cuvsIvfFlatDeserialize(cuvsResources_t res_h, const char* filename, cuvsIvfFlatIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr, "index is null");
    DLDataType dl;
    auto idx = is_native_container(filename) ? flat_read_native(res, filename, &dl) : flat_read_ref(res, filename, &dl);
    delete reinterpret_cast<ivf_flat_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = dl;
  });
}

// test hooks (not in the reference ABI): list size, and a row-major copy of one list + its source ids
__attribute__((visibility("default"))) int cuvsAmdIvfFlatListSize(cuvsIvfFlatIndex_t index, uint32_t label,
                                                                    uint32_t* size)
{
  return translate_exceptions([=] {
    auto& idx = get_flat(index);
    CUVS_EXPECTS(label < idx.n_lists, "label out of range");
    *size = idx.h_list_sizes[label];
  });
}

__attribute__((visibility("default"))) int cuvsAmdIvfFlatUnpackList(cuvsResources_t res_h, cuvsIvfFlatIndex_t index,
                                                                      uint32_t label, void* out_rows, int64_t* out_ids)
{
  return translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_flat(index);
    CUVS_EXPECTS(label < idx.n_lists, "label out of range");
    uint32_t sz = idx.h_list_sizes[label];
    if (sz == 0) return;
    int64_t total = (int64_t)sz * idx.dim;
    hipLaunchKernelGGL(unpack_flat_list_kernel, dim3(grid_blocks(total, 256)), dim3(256), 0, res.stream,
                       idx.data.data(), idx.n_chunks, idx.veclen, (uint32_t)elem_size(idx.dtype), idx.dim,
                       (int64_t)idx.h_list_offsets[label], sz, static_cast<uint8_t*>(out_rows));
    copy_async(res, out_ids, idx.indices.data() + idx.h_list_offsets[label], (size_t)sz * sizeof(int64_t));
    HIP_TRY(hipGetLastError());
  });
}

}  // extern "C"
