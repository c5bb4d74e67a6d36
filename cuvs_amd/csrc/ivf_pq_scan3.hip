// IVF-PQ search, warm-bounds phase on the matrix cores: "decode + MFMA filter, exact re-score of the survivors".
//
// Reference semantics: the score of (query, probed row) is the sum over the pq_dim subspaces of the query's LUT entry
// of the row's code (compute_score_impl.cuh:52-79, create_lut_impl.cuh:17-78), the k smallest win
// (ivf_pq_search.cuh:421-669). Once a query has a k-th bound (after the head phase, ivf_pq_search.hip) all but ~1e-4
// of the (row, query) pairs of the remaining probes are above it. The LUT scan still gathers one LDS entry per code
// byte and pair to find that out; here the bulk of the pairs never touch a LUT:
//
//   filter    For L2 the real-valued score is |r - d|^2 = |r|^2 - 2 r.d + |d|^2 (r: rotated query minus rotated list
//             centre, d: the row's decoded residual); for inner product it is -(q.c + q.d). r.d over (32 rows x 64
//             queries) is a GEMM tile: a wave DECODES 32 rows straight into MFMA A operands (lane = (row, half of the
//             K slice): 4 ds_read_b32 gathers of fp16x2 codebook entries per v_mfma_f32_32x32x16_f16, 32 per row for
//             all 64 subspaces - per row and 64 queries, not per row and query) against the queries' residuals held as
//             fp16 B operands in registers. A pair is DROPPED only when a lower bound of its exact score - the fp16
//             GEMM value minus rigorous rounding margins, see filter_threshold() - is above the query's bound; every
//             other pair is appended to a survivor list (pair id, flat row).
//   re-score  one lane per survivor computes the reference's score with the reference's arithmetic for the requested
//             LUT / score types (pq_lut_math.hpp: the same entry roundings, the same summation order as the LUT scan:
//             bit-identical scores) and, if it is still within the bound and passes the pre-filter, appends it to the
//             query's candidate pool (the tail of the query's per-pair candidate rows);
//   merge     one wave per query selects the k best of head lists + pool by (score, probe rank, row) - the order the
//             per-pair lists + select_k of the LUT path produce - and writes them sorted by (score, row).
// The bound is the head phase's and does not move during the filter, so the survivor set - and with it every result -
// is independent of scheduling. Queries the scheme cannot serve (no finite bound yet, residuals beyond the fp16
// range, a full pool) are flagged and re-done by the LUT scan kernel, pair by pair (ivf_pq_search.hip).
//
// Roofline: the decode is bound by the random LDS gathers (5.2 cycles per wave-level ds_read_b32 and CU, measured:
// profiles/r03_lds_gather_bench.json): 32 gathers per 32 rows and 64 queries = 0.081 cycles per pair, against 16
// MFMAs x 8 cycles per CU = 0.0625 cycles per pair on the matrix cores.
#include "ivf_pq.hpp"
#include "ops.hpp"
#include "device_utils.hpp"
#include "ivf_common.hpp"
#include "pq_lut_math.hpp"
#include "ivf_pq_scan3.hpp"
#include "ivf_pq_filter_common.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace cuvs_amd {

namespace {

constexpr int kFThreads = 512;  // 8 waves: two per SIMD, up to 256 registers each (64 of them the B operands)
constexpr int kFWaves   = kFThreads / 64;



// ------------------------------------------------------------------ per-index tables
// (PER_CLUSTER: pq_dim = n_lists - one table per list, shared by its subspaces; the indexing is the same)
__global__ void cb16_kernel(const float* __restrict__ pq_centers, uint32_t pq_dim, uint32_t pq_len, uint32_t book, float sc,
                            uint16_t* __restrict__ cb16)
{
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;  // s * 256 + code
  if (e >= pq_dim * 256u) return;
  const uint32_t s = e >> 8, code = e & 255u;
  // [subspace][code][component]: an entry's pq_len fp16 values side by side (pq_len 2: one 32-bit word)
  // (codes of fewer than 8 bits: the table keeps 256 slots per subspace, those past the codebook are never looked up)
  for (uint32_t l = 0; l < pq_len; ++l)
    cb16[(size_t)e * pq_len + l] =
      code < book ? __builtin_bit_cast(uint16_t, (_Float16)(sc * pq_centers[(size_t)(s * pq_len + l) * book + code])) : (uint16_t)0;
}

// the fp32 codebook entry-major (PER_SUBSPACE): [subspace][256 codes][component] - a code's pq_len values side by side
__global__ void cbt_kernel(const float* __restrict__ pq_centers, uint32_t pq_dim, uint32_t pq_len, uint32_t book, float* __restrict__ cbt)
{
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;  // s * 256 + code
  if (e >= pq_dim * 256u) return;
  const uint32_t s = e >> 8, code = e & 255u;
  for (uint32_t l = 0; l < pq_len; ++l) cbt[(size_t)e * pq_len + l] = code < book ? pq_centers[(size_t)(s * pq_len + l) * book + code] : 0.f;
}

// codes of fewer than 8 bits (a little-endian bit stream of codes_per_chunk codes per 16-byte chunk,
// ivf_pq_codepacking.cuh:22-52) expanded to one byte per code, 16 per chunk: the layout every kernel of this file reads.
// One thread per (row, 16-code chunk)
__global__ void expand_codes_kernel(const uint8_t* __restrict__ codes, int64_t rows, uint32_t n_chunks, uint32_t cpc, uint32_t bits,
                                    uint32_t pq_dim, uint8_t* __restrict__ out)
{
  const uint32_t nch8 = pq_dim / 16;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nch8) return;
  const int64_t row = i / nch8;
  const uint32_t c8 = (uint32_t)(i % nch8);
  uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const uint32_t s = c8 * 16 + b, ch = s / cpc, bit = (s % cpc) * bits;
    const uint8_t* base = codes + ((size_t)(row >> 6) * n_chunks + ch) * 1024 + (size_t)(row & 63) * 16;
    uint32_t v = base[bit >> 3];
    if ((bit & 7) + bits > 8) v |= (uint32_t)base[(bit >> 3) + 1] << 8;
    w[b >> 2] |= ((v >> (bit & 7)) & ((1u << bits) - 1u)) << ((b & 3) * 8);
  }
  reinterpret_cast<uint4*>(out)[((size_t)(row >> 6) * nch8 + c8) * 64 + (row & 63)] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ inline float wave_reduce_max_f32(float v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// |d|^2 of every row's decoded residual, shrunk by the relative margin (see filter_threshold), as an extra K element of
// the GEMM: x = -|d|^2 (1 - 2^-9) sc^2 / 2 split into two fp16 values (hi + lo = x to 2^-22; |x| <= 16384 by the choice
// of sc), so that the accumulator of a (row, query) pair ends up holding sc^2 (r.d - |d|^2 (1 - 2^-9) / 2). One thread per row
__global__ void row_term_kernel(const uint8_t* __restrict__ codes, const float* __restrict__ pq_centers, int64_t rows, float sc,
                                uint32_t* __restrict__ term, uint32_t* __restrict__ dn_max_bits, int n_chunks, int fp32, int pq_len,
                                uint32_t book, const uint32_t* __restrict__ list_offsets, uint32_t n_lists)
{
  const int64_t r0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r  = min(r0, rows - 1);  // (no early exit: the wave reduction below needs every lane)
  // PER_CLUSTER (list_offsets != nullptr): the codebook is the row's list's - the last list starting at or before the row
  uint32_t L = 0u;
  if (list_offsets != nullptr) {
    uint32_t lo = 0u, hi = n_lists;  // list_offsets[lo] <= r < list_offsets[hi]
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((int64_t)list_offsets[mid] <= r) lo = mid; else hi = mid;
    }
    L = lo;
  }
  const uint4* cp = reinterpret_cast<const uint4*>(codes) + ((size_t)(r >> 6) * n_chunks) * 64 + (r & 63);
  float dn = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    const uint4 cw       = cp[c * 64];
    const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const uint32_t code = (ws[b >> 2] >> ((b & 3) * 8)) & 0xffu;
      const uint32_t s    = c * 16 + b;
      for (int l = 0; l < pq_len; ++l) {
        const float p = pq_centers[(size_t)((list_offsets != nullptr ? L : s) * pq_len + l) * book + code];
        dn = __fmaf_rn(p, p, dn);
      }
    }
  }
  // the largest |d|^2 of the index (bits of a non-negative float order like unsigned integers)
  const float wmax = wave_reduce_max_f32(dn);
  if ((threadIdx.x & 63) == 0) atomicMax(dn_max_bits, __float_as_uint(wmax));
  const float x     = -0.5f * sc * sc * (dn * (1.0f - 1.0f / 512.0f));
  const _Float16 hi = (_Float16)x;
  const _Float16 lo = (_Float16)(x - (float)hi);
  // (pq_filter4_kernel takes the term as the fp32 initial value of its accumulators, pq_filter_kernel as two fp16 K elements)
  if (r0 < rows)
    term[r] = fp32 ? __float_as_uint(x) : ((uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16));
}

// ------------------------------------------------------------------ work units: (list, <= 64 pairs, row chunk)

__global__ __launch_bounds__(1024) void count_units_kernel(const uint32_t* __restrict__ pair_off, uint32_t n_lists,
                                                           const uint32_t* __restrict__ list_sizes, uint32_t unit_rows,
                                                           uint32_t* __restrict__ unit_off, uint32_t group, uint32_t lbase)
{
  // a thread takes a run of consecutive lists, one block scan of the 1024 run totals (two passes instead of n_lists / 1024
  // block scans with three barriers each)
  __shared__ int smem[17];
  const uint32_t per = (n_lists + 1023u) / 1024u;
  const uint32_t b = threadIdx.x * per, e = min(n_lists, b + per);
  auto units_of = [&](uint32_t i) {
    const uint32_t np = pair_off[lbase + i + 1] - pair_off[lbase + i];  // tail labels: lbase = n_lists (head labels: 0)
    const uint32_t len = list_sizes[i];
    return (int)(((np + group - 1u) / group) * ((len + unit_rows - 1u) / unit_rows));  // (empty lists: no units)
  };
  int s = 0;
  for (uint32_t i = b; i < e; ++i) s += units_of(i);
  int total;
  int run = block_exclusive_scan(s, smem, &total);
  for (uint32_t i = b; i < e; ++i) {
    unit_off[i] = (uint32_t)run;
    run += units_of(i);
  }
  if (threadIdx.x == 0) unit_off[n_lists] = (uint32_t)total;
}

__global__ void fill_units_kernel(const uint32_t* __restrict__ pair_off, uint32_t n_lists,
                                  const uint32_t* __restrict__ list_offsets, const uint32_t* __restrict__ list_sizes,
                                  uint32_t unit_rows, const uint32_t* __restrict__ unit_off, filter_unit* __restrict__ units,
                                  uint32_t group, uint32_t lbase)
{
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= n_lists) return;
  const uint32_t b = pair_off[lbase + L], e = pair_off[lbase + L + 1], len = list_sizes[L];
  uint32_t w = unit_off[L];
  if (b == e || len == 0u) return;
  // the list is cut into ceil(len / unit_rows) chunks of equal length (a multiple of 64 rows); row chunk major: the query
  // groups of one chunk run next to each other and share its rows in L2
  const uint32_t n_ch = (len + unit_rows - 1u) / unit_rows;
  const uint32_t rows = ((len + n_ch - 1u) / n_ch + 63u) & ~63u;
  const uint32_t base = list_offsets[L];
  for (uint32_t c = 0; c < n_ch; ++c) {
    const uint32_t r0 = c * rows, r1 = min(len, r0 + rows);
    if (r0 >= r1) continue;  // (the rounding can leave the last chunk empty: its units stay zero-length)
    for (uint32_t p = b; p < e; p += group) units[w++] = filter_unit{L, p, min(group, e - p), r0, base, r1, 0u, 0u};
  }
  // units the count reserved but the equal-length cut did not need: zero rows
  const uint32_t w_end = unit_off[L] + ((e - b + group - 1u) / group) * n_ch;
  for (; w < w_end; ++w) units[w] = filter_unit{L, b, 1u, 0u, base, 0u, 0u, 0u};
}

// ------------------------------------------------------------------ the filter
struct filter_params {
  const filter_unit* units;
  const uint32_t* n_units;  // device scalar
  uint32_t* xcd_ticket;     // 8 counters, 32 words apart
  const uint32_t* sorted_pairs;
  const float* rot_queries;
  const float* centers_rot;
  const uint32_t* cb16;
  const uint8_t* codes;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  const uint32_t* row_term;  // packed fp16 (hi, lo) of the row's K-extension term (L2), nullptr for inner product
  const uint32_t* query_kth;
  uint32_t* qflag;
  uint2* surv;         // one region of surv_cap entries per workgroup (a single global counter for all survivors made
  uint32_t* surv_cnt;  // ~500 k same-address atomics per search the bottleneck of the kernel: 6.6 of them per us)
  uint32_t surv_cap;   // entries per region; surv_cnt[b]: fill of workgroup b's region, written once at kernel end
  uint32_t spill_cap;  // entries of the shared spill region behind the workgroups' regions (counter: surv_cnt[gridDim.x])
  uint32_t n_probes, rot_dim, unit_rows;
  float sc;        // power of two applied to both GEMM operands before the fp16 rounding
  float c1;        // -2 / sc^2 (L2) or -1 / sc^2 (inner product)
  float eps, alpha;  // exact score >= real score * (1 - eps) - alpha for the requested LUT / score types
  float cbmax, dmax; // largest codebook value, largest decoded residual norm
  float bound_max;   // bounds beyond this are not served here (fp8 LUT saturation)
  int is_ip;
  unsigned long long* stats;  // optional [8]: pairs tested, survivors, subtiles, slow-path subtiles, cycles (see the kernel), units
  int dbg;                    // ablations (timing only, wrong results): 1 conflict-free gathers, 2 no gathers, 4 no MFMAs
  // IVF-Flat (FLAT builds of the kernel): the rows' residuals against their list centre as scaled fp16, laid out as the
  // A operands themselves - [32-row tile][K step][lane] x 16 bytes - so a subtile is NST coalesced 1 KiB loads
  const uint4* rows16;
  // IVF-Flat has no per-pair hand-back kernel: a query that cannot be served survives every test (its rows are all
  // re-scored), and a full buffer raises *fail - the caller re-runs the batch's tail phase on the scan kernel
  uint32_t* fail;
  // flat_filter2_kernel: B operands and thresholds of the tail pairs from the pre-pass (pq_bprep_kernel, ivf_pq_filter4.hip)
  const uint4* bq;
  const float* thr_pair;
  const uint32_t* pair_off;
  uint32_t n_lists;
  // EMIT build of flat_filter2_kernel (the bound-only head phase): the screened value of every (head pair, row) goes to
  // xbuf[pair position][row of the list], ldx floats per pair; lbase: first label of the pairs served (0: head pairs)
  float* xbuf;
  uint32_t ldx, lbase;
};

// NCH: 16-byte code chunks per row = pq_dim / 16 (pq_len 2: rot_dim = 32 NCH, 2 NCH MFMA K steps). Up to 4 chunks a
// work unit holds two groups of 32 queries (B operands: 16 NCH registers) and the decoded rows are double-buffered;
// beyond (pq_dim 80 .. 128) one group and one buffer.
// DBG: ablations (timing only, wrong results): 1 conflict-free gathers, 2 no gathers, 4 no MFMAs, 8 no code loads
template <int NCH, int DBG, bool FLAT = false>
__global__ __launch_bounds__(kFThreads) void pq_filter_kernel(const filter_params a)
{
  constexpr int NST = 2 * NCH;            // MFMA K steps
  constexpr int NG  = NCH <= 4 ? 2 : 1;   // groups of 32 queries per work unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* cb = reinterpret_cast<uint32_t*>(smem);  // [pq_dim subspaces][256 codes] fp16x2 (64 KiB at pq_dim 64)
  uint32_t* wg_fill = reinterpret_cast<uint32_t*>(smem + (FLAT ? 0 : NCH * 16 * 1024));  // survivors of this workgroup so far
  uint2* my_surv    = a.surv + (size_t)blockIdx.x * a.surv_cap;
  if (threadIdx.x == 0) *wg_fill = 0u;
  if constexpr (!FLAT)
    for (uint32_t i = threadIdx.x; i < (uint32_t)NCH * 16u * 256u / 4u; i += kFThreads)
      reinterpret_cast<uint4*>(cb)[i] = reinterpret_cast<const uint4*>(a.cb16)[i];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const uint32_t ql = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
  const uint32_t* cbh = cb + h * (8u * 256u);  // this lane's half of every 16-subspace chunk

  // XCD x owns the x-th eighth of the (list-sorted) unit array; its waves draw units through one ticket counter (the
  // units of a list run on one XCD at about the same time and share its L2). A wave whose XCD has run dry moves on to
  // the next XCD's share: the shares are equal in units, not in work.
  const uint32_t n_units = *a.n_units;
  const uint32_t chunk = (n_units + 7u) / 8u;
  uint32_t xcd = blockIdx.x & 7u, hops = 0u;
  const uint4* codes16 = reinterpret_cast<const uint4*>(a.codes);

  unsigned long long st_pairs = 0, st_surv = 0, st_sub = 0, st_slow = 0, st_units = 0, st_t[3] = {0, 0, 0};
  for (;;) {
    const uint32_t share0 = min(n_units, xcd * chunk), share_len = min(chunk, n_units - share0);
    const filter_unit* share = a.units + share0;
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(a.xcd_ticket + xcd * 32, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= share_len) {
      if (++hops == 8u) break;
      xcd = (xcd + 1u) & 7u;
      continue;
    }
    const uint4 uu = *reinterpret_cast<const uint4*>(share + t);
    const uint32_t L = __builtin_amdgcn_readfirstlane(uu.x), first = __builtin_amdgcn_readfirstlane(uu.y),
                   count = __builtin_amdgcn_readfirstlane(uu.z), row0 = __builtin_amdgcn_readfirstlane(uu.w);
    const uint32_t base_row = a.list_offsets[L], len = a.list_sizes[L];
    const bool two = NG == 2 && count > 32u;  // wave-uniform: the second group of 32 queries
    const unsigned long long t_unit = a.stats != nullptr ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_slow = 0ull;

    // ---- B operands: the item's query residuals as fp16, lane = (query ql of group g, K half h)
    f16x8_t bop[NG][NST];
    float thr[2] = {INFINITY, INFINITY};
    uint32_t pairid[2] = {0u, 0u};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t jj = g * 32u + ql;
      const bool valid  = jj < count;
      const uint32_t p  = valid ? a.sorted_pairs[first + jj] : 0u;
      const uint32_t q  = p / a.n_probes;
      pairid[g]         = p;
      const float* rq   = a.rot_queries + (size_t)q * a.rot_dim;
      const float* ct   = a.centers_rot + (size_t)L * a.rot_dim;
      float rn = 0.f, big = 0.f, qc = 0.f, cn = 0.f;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const uint32_t s0 = 16u * (st >> 1) + 8u * h + 4u * (st & 1);
        const float4 q0 = *reinterpret_cast<const float4*>(rq + 2 * s0), q1 = *reinterpret_cast<const float4*>(rq + 2 * s0 + 4);
        float r[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const float4 c0 = *reinterpret_cast<const float4*>(ct + 2 * s0), c1 = *reinterpret_cast<const float4*>(ct + 2 * s0 + 4);
        const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        if (!a.is_ip) {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] -= c[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { qc = __fmaf_rn(r[e], c[e], qc); cn = __fmaf_rn(c[e], c[e], cn); }
        }
        f16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          rn   = __fmaf_rn(r[e], r[e], rn);
          const float x = a.sc * r[e];
          big  = fmaxf(big, fabsf(x));
          v[e] = (_Float16)fminf(fmaxf(x, -60000.f), 60000.f);  // (finite whatever happens: unserved queries may still be screened)
        }
        bop[g][st] = v;
      }
      rn  += __shfl_xor(rn, 32);
      qc  += __shfl_xor(qc, 32);
      cn  += __shfl_xor(cn, 32);
      big  = fmaxf(big, __shfl_xor(big, 32));
      const uint32_t kk = valid ? a.query_kth[q] : 0u;
      const float bound = key_to_float(kk);
      // no finite bound yet, an operand beyond the fp16 range, a bound the LUT type cannot represent, or a query that
      // is re-done by the LUT scan anyway: nothing of this query survives here
      const bool served = valid && kk < 0xff800000u && big < 60000.f && fabsf(bound) <= a.bound_max && (FLAT || a.qflag[q] == 0u);
      if (!FLAT && valid && !served) a.qflag[q] = 1u;
      const float t = a.is_ip ? filter_threshold_ip(bound, rn, cn, qc, a) : filter_threshold(bound, rn, a);
      // in accumulator units (c1 < 0: the test flips); an unserved query drops everything (PQ: handed back) or survives
      // everything (FLAT: all its rows are re-scored)
      thr[g] = served ? t / a.c1 : ((FLAT && valid) ? -INFINITY : INFINITY);
    }
    // B operand of the K-extension step: the row term's two halves times one
    const u32x4_t oq   = {h == 0u ? 0x3c003c00u : 0u, 0u, 0u, 0u};
    const f16x8_t ones = __builtin_bit_cast(f16x8_t, oq);

    // ---- rows of the unit, 32 at a time. Software pipeline of a wave: the code words are loaded two subtiles ahead;
    // the 32 gathers that decode subtile u + 1 are issued right after the MFMAs of subtile u and land while its
    // accumulators are screened; the other wave of the SIMD fills the matrix pipe meanwhile.
    const uint32_t r_end = __builtin_amdgcn_readfirstlane(share[t].r_end);
    const uint32_t u0 = row0 >> 5, u1 = (r_end + 31u) >> 5;
    auto load_codes = [&](const uint32_t u, uint2 (&cw)[NCH]) {
      const uint32_t fr = base_row + (min(u, u1 - 1u) << 5) + ql;  // this lane's row (padded rows of a group are readable)
      const char* p = reinterpret_cast<const char*>(codes16 + ((size_t)(fr >> 6) * NCH) * 64 + (fr & 63u)) + 8u * h;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if constexpr ((DBG & 8) != 0) cw[c] = make_uint2(u * 2654435761u + c, fr);  // ablation: no code loads
        else cw[c] = *reinterpret_cast<const uint2*>(p + (size_t)c * 64 * 16);
      }
    };
    auto decode = [&](const uint2 (&cw)[NCH], u32x4_t (&av)[NST]) {
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const uint32_t w = (st & 1) ? cw[st >> 1].y : cw[st >> 1].x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t code = (w >> (8 * e)) & 0xffu;
          if constexpr ((DBG & 1) != 0) code = ql;  // ablation: every lane of a half in its own bank
          av[st][e] = (DBG & 2) ? code : cbh[((16u * (st >> 1) + 4u * (st & 1) + e) << 8) + code];
        }
      }
    };
    // FLAT: the A operands of subtile u are NST 16-byte loads per lane from the fp16 residual copy
    auto load_rows = [&](const uint32_t u, u32x4_t (&av)[NST]) {
      const uint4* p = a.rows16 + ((size_t)((base_row >> 5) + min(u, u1 - 1u)) * NST) * 64 + lane;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const uint4 v = p[(size_t)st * 64];
        av[st] = u32x4_t{v.x, v.y, v.z, v.w};
      }
    };
    auto run = [&](auto two_tag) {
      constexpr bool TWO = decltype(two_tag)::value;
      constexpr bool DOUBLE = NCH <= 4;  // decoded rows of the current / the next subtile in two buffers (roles alternate)
      uint2 cw1[NCH], cw2[NCH];
      u32x4_t avA[NST], avB[DOUBLE ? NST : 1];
      if constexpr (FLAT) {
        load_rows(u0, avA);
      } else {
        load_codes(u0, cw1);
        load_codes(u0 + 1, cw2);
        decode(cw1, avA);
#pragma unroll
        for (int c = 0; c < NCH; ++c) cw1[c] = cw2[c];
        load_codes(u0 + 2, cw2);
      }
      // one subtile: the gathers that decode subtile u + 1 are issued BEFORE the MFMAs of subtile u, so their trip
      // through the LDS queue (shared with seven other waves) overlaps this wave's own matrix work
      auto step = [&](const uint32_t u, u32x4_t (&cur)[NST], auto& nxt) {
        uint32_t term = 0u;  // the rows' K-extension term (one dword per row, lanes of half 0)
        if (a.row_term != nullptr && h == 0u) term = a.row_term[base_row + (u << 5) + ql];
        if constexpr (FLAT) {
          if constexpr (DOUBLE) load_rows(u + 1, nxt);
        } else if constexpr (DOUBLE) {
          decode(cw1, nxt);  // (clamped to the last subtile: a harmless repeat at the end)
#pragma unroll
          for (int c = 0; c < NCH; ++c) cw1[c] = cw2[c];
          load_codes(u + 3, cw2);
        }
        f32x16_t acc0 = {}, acc1 = {};
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          const f16x8_t aop = __builtin_bit_cast(f16x8_t, cur[st]);
          if constexpr ((DBG & 4) != 0) { acc0[st] += (float)aop[0]; continue; }  // ablation: no MFMAs
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, bop[0][st], acc0, 0, 0, 0);
          if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, bop[NG - 1][st], acc1, 0, 0, 0);
        }
        if constexpr (!DOUBLE) {  // one buffer: the next subtile is decoded once this one's MFMAs are issued
          if constexpr (FLAT) {
            load_rows(u + 1, cur);
          } else {
            decode(cw1, cur);
#pragma unroll
            for (int c = 0; c < NCH; ++c) cw1[c] = cw2[c];
            load_codes(u + 3, cw2);
          }
        }
        if (a.row_term != nullptr) {  // wave-uniform: the extra K step adds -|d|^2 (1 - 2^-9) sc^2 / 2 to every pair of the row
          const u32x4_t tq  = {term, 0u, 0u, 0u};
          const f16x8_t top = __builtin_bit_cast(f16x8_t, tq);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(top, ones, acc0, 0, 0, 0);
          if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(top, ones, acc1, 0, 0, 0);
        }
        // ---- screen: accumulator register i of this lane is row (i & 3) + 8 (i >> 2) + 4 h of the subtile; a pair
        // survives when c1 * acc <= thr, i.e. acc >= thr / c1 (c1 < 0, a power of two)
        float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          m0 = fmaxf(m0, acc0[i]);
          if constexpr (TWO) m1 = fmaxf(m1, acc1[i]);
        }
        const bool any = (m0 >= thr[0]) || (TWO && m1 >= thr[1]);
        if (a.stats != nullptr) { st_pairs += 32u * count; st_sub += 1u; }
        if (__ballot(any) == 0ull) return;  // the usual case
        // ---- slow path (9 % of the subtiles, one or two survivors each): the few lanes that hold a survivor go through
        // their own 16 (32) values and append them one by one through the workgroup's LDS counter
        if (a.stats != nullptr) st_slow += 1u;
        const unsigned long long t_s0 = a.stats != nullptr ? __builtin_readcyclecounter() : 0ull;
        if (any) {
#pragma unroll
          for (int g = 0; g < (TWO ? 2 : 1); ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const uint32_t v = (u << 5) + (uint32_t)((i & 3) + 8 * (i >> 2)) + 4u * h;
              const float x    = g == 0 ? acc0[i] : acc1[i];
              if (x >= thr[g] && v < r_end) {
                const uint32_t pos = atomicAdd(wg_fill, 1u);  // LDS
                if (pos < a.surv_cap) {
                  my_surv[pos] = make_uint2(pairid[g], base_row + v);
                } else {
                  // this workgroup's region is full: the spill region shared by all (one global counter, rarely touched);
                  // only when that is full too is the query handed back to the LUT scan
                  const uint32_t sp = atomicAdd(a.surv_cnt + gridDim.x, 1u);
                  if (sp < a.spill_cap) a.surv[(size_t)gridDim.x * a.surv_cap + sp] = make_uint2(pairid[g], base_row + v);
                  else if (FLAT) *a.fail = 1u;
                  else a.qflag[pairid[g] / a.n_probes] = 1u;
                }
                if (a.stats != nullptr) st_surv += 1u;
              }
            }
        }
        if (a.stats != nullptr) t_slow += __builtin_readcyclecounter() - t_s0;
      };
      if constexpr (DOUBLE) {
        for (uint32_t u = u0; u < u1; u += 2) {
          step(u, avA, avB);
          if (u + 1 < u1) step(u + 1, avB, avA);
        }
      } else {
        for (uint32_t u = u0; u < u1; ++u) step(u, avA, avA);
      }
    };
    const unsigned long long t_loop = a.stats != nullptr ? __builtin_readcyclecounter() : 0ull;
    if (u0 < u1) {
      if (two) run(std::true_type{}); else run(std::false_type{});
    }
    if (a.stats != nullptr) {
      const unsigned long long t_end = __builtin_readcyclecounter();
      st_t[0] += t_loop - t_unit;  // unit prologue (B operands, thresholds)
      st_t[1] += t_end - t_loop;   // subtile loop
      st_t[2] += t_slow;           // of which slow path
      st_units += 1u;
    }
  }
  if (a.stats != nullptr) atomicAdd(&a.stats[1], st_surv);  // counted per lane
  if (a.stats != nullptr && lane == 0) {  // one flush per wave (atomics inside the loop cost more than the loop)
    atomicAdd(&a.stats[0], st_pairs); atomicAdd(&a.stats[2], st_sub);
    atomicAdd(&a.stats[3], st_slow);  atomicAdd(&a.stats[4], st_t[0]); atomicAdd(&a.stats[5], st_t[1]);
    atomicAdd(&a.stats[6], st_t[2]);  atomicAdd(&a.stats[7], st_units);
  }
  __syncthreads();
  if (threadIdx.x == 0) a.surv_cnt[blockIdx.x] = min(*wg_fill, a.surv_cap);
}

// ------------------------------------------------------------------ IVF-Flat: the filter with the B operands in LDS (round 6)
// pq_filter_kernel<.., FLAT> is bound by HBM on its OWN re-reads (profiles/r04_pmc_c4_c2.json: 6.57 GB per search at 7.8 TB/s
// for a 2.56 GB fp16 copy): a unit is (list chunk, 64 queries) and every wave streams its unit's rows by itself, so a list probed
// by 156 queries (C2's mean) is fetched 2.6 times - the units of a chunk run on one XCD, but 256 streams share its 4 MiB of L2
// and a line is gone before the neighbour asks for it. Here a unit is (list chunk, up to 256 queries) and belongs to the whole
// WORKGROUP: its waves build the B operands of the unit's eight query groups into LDS once (8 KiB per group at 128 dimensions),
// then every wave takes every eighth 32-row subtile, loads its A operands from the fp16 copy ONCE and multiplies them with group
// after group (B operand: eight ds_read_b128 per group - the LDS delivers them ten times faster than HBM delivers rows).
// Same operands, same accumulation order (K steps, then the row term's K-extension step), same thresholds as the kernel
// above: the same survivors.
constexpr int kF2Threads = 256;  // 4 waves; two workgroups per CU (one builds its unit's operands while the other streams rows)
constexpr int kF2Waves   = kF2Threads / 64;

// EMIT (the bound-only head phase, 3.1c): no screen - the value of every (pair, row) is written to a.xbuf (rows past the chunk's
// end: -inf); the pairs are the head pairs (a.lbase = 0), thresholds and pair ids are not read
template <int NST, bool EMIT = false>
__global__ __launch_bounds__(kF2Threads, 2) void flat_filter2_kernel(const filter_params a)
{
  constexpr int NGM      = 8;          // query groups per unit
  constexpr bool DOUBLE  = NST <= 8;   // A operands of the wave's next two subtiles in registers of their own (beyond: one set)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Bs        = reinterpret_cast<uint4*>(smem);                             // [NGM][NST][64 lanes] x 16 B
  float* s_thr     = reinterpret_cast<float*>(smem + (size_t)NGM * NST * 1024);  // [NGM][32]
  uint32_t* s_pair = reinterpret_cast<uint32_t*>(s_thr + NGM * 32);              // [NGM][32]
  uint32_t* ctrl   = s_pair + NGM * 32;                                          // [0] survivors of this workgroup, [1] current unit
  uint2* my_surv   = a.surv + (size_t)blockIdx.x * a.surv_cap;
  if (threadIdx.x == 0) ctrl[0] = 0u;

  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t ql = lane & 31u, h = lane >> 5;
  const uint32_t n_units = *a.n_units;
  const uint32_t chunk   = (n_units + 7u) / 8u;
  const uint32_t s_base  = a.pair_off[EMIT ? a.lbase : a.n_lists];  // first pair served (tail pairs; EMIT: head pairs) in sorted_pairs
  uint32_t xcd = blockIdx.x & 7u, hops = 0u;  // (thread 0's: the workgroup's position in the XCD shares)
  unsigned long long st_pairs = 0, st_surv = 0, st_sub = 0, st_units = 0;
  const u32x4_t oq   = {h == 0u ? 0x3c003c00u : 0u, 0u, 0u, 0u};
  const f16x8_t ones = __builtin_bit_cast(f16x8_t, oq);  // B operand of the K-extension step: the row term's two halves times one

  for (;;) {
    __syncthreads();  // nobody reads the previous unit's operands any more
    if (threadIdx.x == 0) {
      uint32_t ui = 0xffffffffu;
      for (;;) {  // XCD x owns the x-th eighth of the (list-sorted) units; a workgroup whose XCD has run dry moves on to the next share
        const uint32_t share0 = min(n_units, xcd * chunk), share_len = min(chunk, n_units - share0);
        const uint32_t t = atomicAdd(a.xcd_ticket + xcd * 32, 1u);
        if (t < share_len) { ui = share0 + t; break; }
        if (++hops == 8u) break;
        xcd = (xcd + 1u) & 7u;
      }
      ctrl[1] = ui;
    }
    __syncthreads();
    const uint32_t ui = ctrl[1];
    if (ui == 0xffffffffu) break;  // workgroup-uniform
    const filter_unit* up = a.units + ui;
    const uint4 uu = *reinterpret_cast<const uint4*>(up);
    const uint32_t L = __builtin_amdgcn_readfirstlane(uu.x), first = __builtin_amdgcn_readfirstlane(uu.y),
                   count = __builtin_amdgcn_readfirstlane(uu.z), row0 = __builtin_amdgcn_readfirstlane(uu.w);
    const uint32_t r_end    = __builtin_amdgcn_readfirstlane(up->r_end);
    const uint32_t base_row = a.list_offsets[L];
    const uint32_t ng       = (count + 31u) >> 5;  // 1 .. NGM, workgroup-uniform
    const uint32_t u0 = row0 >> 5, u1 = (r_end + 31u) >> 5;
    // the rows' K-extension terms (one dword per row, used by the lanes of half 0) travel WITH the rows: loaded where the rows are
    // loaded, unconditionally (inner product: a readable dummy address) - a conditional load at the top of a subtile made the
    // compiler wait for EVERY outstanding load (s_waitcnt vmcnt(0)) before the term's MFMA, i.e. for the rows it had just asked
    // for two subtiles ahead: one exposed HBM round trip per subtile
    const uint32_t* term_base = a.row_term != nullptr ? a.row_term + base_row : reinterpret_cast<const uint32_t*>(a.rows16);
    auto load_rows = [&](const uint32_t u, u32x4_t (&av)[NST], uint32_t& tm) {
      const uint32_t uc = min(u, u1 - 1u);
      const uint4* p = a.rows16 + ((size_t)((base_row >> 5) + uc) * NST) * 64 + lane;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const uint4 v = p[(size_t)st * 64];
        av[st] = u32x4_t{v.x, v.y, v.z, v.w};
      }
      tm = term_base[(a.row_term != nullptr ? (uc << 5) : 0u) + ql];
    };
    // the wave's first two subtiles are on their way while the unit's operands are built (three register sets: a subtile's rows
    // are asked for two subtiles - ~1.5 us of matrix work - before they are multiplied; one subtile ahead, 0.8 us, is less than
    // a loaded HBM takes to answer)
    u32x4_t avA[NST], avB[DOUBLE ? NST : 1], avC[DOUBLE ? NST : 1];
    uint32_t tmA = 0u, tmB = 0u, tmC = 0u;
    if (u0 < u1) {
      load_rows(u0 + wave, avA, tmA);
      if constexpr (DOUBLE) load_rows(u0 + wave + kF2Waves, avB, tmB);
    }
    // ---- B operands, thresholds and pair ids of the unit's queries, from the pre-pass (fp16 residuals in B-operand layout, 32 NST
    // bytes per pair; thresholds in accumulator units): wave w copies the groups w, w + 4; lane = (query ql, K half h)
    for (uint32_t g = wave; g < ng; g += kF2Waves) {
      const uint32_t jj = g * 32u + ql;
      const uint32_t jc = min(jj, count - 1u);
      const uint4* bp   = a.bq + ((size_t)(first - s_base + jc) * NST * 2 + h);
#pragma unroll
      for (int st = 0; st < NST; ++st) Bs[(g * NST + st) * 64 + lane] = bp[st * 2];
      if (!EMIT && h == 0u) {
        s_thr[g * 32 + ql]  = jj < count ? a.thr_pair[first - s_base + jc] : INFINITY;  // (a padding slot keeps nothing)
        s_pair[g * 32 + ql] = a.sorted_pairs[first + jc];
      }
    }
    __syncthreads();

    // ---- the unit's rows: wave w takes the subtiles u0 + w, u0 + w + 4, ...; the A operands of its next subtile are loaded
    // while this one is multiplied (two register sets up to 128 dimensions)
    // B operands of a group in two halves (K steps [0, H) and [H, NST)): the halves of group g + 1 are read from LDS while the
    // MFMAs of the other half of group g run - one register set, no exposed LDS round trip per group (a wave that reads all of a
    // group's operands and then waits for them spends ~700 cycles per group on ~300 cycles of matrix work)
    constexpr int H = NST >= 2 ? NST / 2 : 1;
    auto load_b_half = [&](const uint32_t g, const int half, u32x4_t (&bq)[NST]) {
      const uint4* p = Bs + (size_t)g * NST * 64 + lane;
      if (half == 0) {
#pragma unroll
        for (int st = 0; st < H; ++st) {
          const uint4 v = p[st * 64];
          bq[st] = u32x4_t{v.x, v.y, v.z, v.w};
        }
      } else {
#pragma unroll
        for (int st = H; st < NST; ++st) {
          const uint4 v = p[st * 64];
          bq[st] = u32x4_t{v.x, v.y, v.z, v.w};
        }
      }
    };
    u32x4_t bq[NST];
    float th_next = INFINITY;
    load_b_half(0u, 0, bq);
    load_b_half(0u, 1, bq);
    if constexpr (!EMIT) th_next = s_thr[ql];
    auto step = [&](const uint32_t u, u32x4_t (&cur)[NST], uint32_t& tcur, auto& nxt, uint32_t& tnxt) {
      const uint32_t term = h == 0u ? tcur : 0u;
      if constexpr (DOUBLE) load_rows(u + 2u * kF2Waves, nxt, tnxt);  // (nxt: the set the subtile before this one has just left)
      const u32x4_t tq  = {term, 0u, 0u, 0u};
      const f16x8_t top = __builtin_bit_cast(f16x8_t, tq);
      auto one = [&](const uint32_t gg, const bool last) {
        const uint32_t gn = gg + 1u < ng ? gg + 1u : 0u;  // the group whose operands are read next (group 0: of the wave's next subtile)
        const float thg   = th_next;
        f32x16_t acc = {};
#pragma unroll
        for (int st = 0; st < H; ++st)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, cur[st]), __builtin_bit_cast(f16x8_t, bq[st]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_b_half(gn, 0, bq);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = H; st < NST; ++st)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, cur[st]), __builtin_bit_cast(f16x8_t, bq[st]), acc, 0, 0, 0);
        if (a.row_term != nullptr) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(top, ones, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_b_half(gn, 1, bq);
        if constexpr (!EMIT) th_next = s_thr[gn * 32 + ql];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!DOUBLE) {
          if (last) load_rows(u + kF2Waves, cur, tcur);  // one register set: the next subtile's rows once this one's last MFMAs are issued
        }
        if constexpr (EMIT) {
          // registers 4 j .. 4 j + 3 of a lane are the rows 8 j + 4 h .. + 3 of the subtile: one 16-byte store each
          const uint32_t jj = gg * 32u + ql;
          if (jj < count) {
            float* xp = a.xbuf + (size_t)(first - s_base + jj) * a.ldx + (u << 5) + 4u * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t v0 = (u << 5) + 8u * j + 4u * h;
              // (measured and rejected, round 6: the same stores from inline asm, hidden from the compiler's wait insertion - no faster,
              // and a parity test failed; the values staged in LDS and flushed once per unit; the pass on round 3's wave-independent
              // kernel, B operands in registers, no LDS, no barrier: 0.53 - 0.57 ms = 4.4 - 4.7 TB/s every time - the emit pass is not
              // behind its stores, its barriers or its prefetch depth)
              float4 o;
              o.x = v0 + 0u < r_end ? acc[4 * j + 0] : -INFINITY; o.y = v0 + 1u < r_end ? acc[4 * j + 1] : -INFINITY;
              o.z = v0 + 2u < r_end ? acc[4 * j + 2] : -INFINITY; o.w = v0 + 3u < r_end ? acc[4 * j + 3] : -INFINITY;
              *reinterpret_cast<float4*>(xp + 8 * j) = o;
            }
          }
          return;
        }
        // accumulator register i of this lane is row (i & 3) + 8 (i >> 2) + 4 h of the subtile; a pair survives when acc >= thr
        float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, acc[i]), acc[i + 1]);
        m = fmaxf(m, acc[15]);
        const bool any = m >= thg;
        if (a.stats != nullptr) { st_pairs += 32u * min(32u, count - gg * 32u); }
        if (__ballot(any) == 0ull) return;
        // ---- survivors. On fp32 rows screened through their fp16 copy they are NOT rare (C2: 2.3e-3 per pair, two per 32 x 32
        // tile): the lanes that hold some count them (a 16-bit mask), ONE wave scan + ONE LDS atomic hand out the positions
        // (round 3's per-lane loop drew one returning LDS atomic per survivor: most of the kernel's wait cycles at this rate)
        uint32_t hits = 0u;
        if (any) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const uint32_t v = (u << 5) + (uint32_t)((i & 3) + 8 * (i >> 2)) + 4u * h;
            hits |= (acc[i] >= thg && v < r_end) ? (1u << i) : 0u;
          }
        }
        const uint32_t n_mine = (uint32_t)__popc(hits);
        // the few lanes that hold hits, one after the other under scalar control: a lane's offset = the hits of the lanes before
        // it (a wave scan by shuffles is six dependent trips through the LDS crossbar: ~600 cycles for usually two survivors)
        unsigned long long lm = __ballot(hits != 0u);
        uint32_t total = 0u, my_off = 0u;
        while (lm != 0ull) {
          const uint32_t src = (uint32_t)__ffsll((long long)lm) - 1u;
          lm &= lm - 1ull;
          if (lane == src) my_off = total;
          total += (uint32_t)__builtin_amdgcn_readlane(n_mine, src);
        }
        if (total == 0u) return;  // (every hit lay past the chunk's end)
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(&ctrl[0], total);  // LDS
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t pid = s_pair[gg * 32 + ql];
        uint32_t pos = base + my_off;
        uint32_t hb = hits;
        if (base + total <= a.surv_cap) {  // wave-uniform, the usual case: plain stores (nothing here waits for a memory operation)
          while (hb != 0u) {
            const uint32_t i = (uint32_t)__ffs((int)hb) - 1u;
            hb &= hb - 1u;
            my_surv[pos++] = make_uint2(pid, base_row + (u << 5) + (i & 3u) + 8u * (i >> 2) + 4u * h);
          }
        } else {  // this workgroup's region is (nearly) full: the shared spill region; when that is full too the batch is re-run on the scan kernel
          while (hb != 0u) {
            const uint32_t i = (uint32_t)__ffs((int)hb) - 1u;
            hb &= hb - 1u;
            const uint32_t v = (u << 5) + (i & 3u) + 8u * (i >> 2) + 4u * h;
            if (pos < a.surv_cap) {
              my_surv[pos] = make_uint2(pid, base_row + v);
            } else {
              const uint32_t sp = atomicAdd(a.surv_cnt + gridDim.x, 1u);
              if (sp < a.spill_cap) a.surv[(size_t)gridDim.x * a.surv_cap + sp] = make_uint2(pid, base_row + v);
              else *a.fail = 1u;
            }
            ++pos;
          }
        }
        if (a.stats != nullptr) st_surv += n_mine;
      };
      for (uint32_t g = 0; g < ng; ++g) one(g, g + 1u == ng);
      if (a.stats != nullptr) st_sub += 1u;
    };
    if constexpr (DOUBLE) {
      for (uint32_t u = u0 + wave; u < u1; u += 3u * kF2Waves) {
        step(u, avA, tmA, avC, tmC);
        if (u + kF2Waves < u1) step(u + kF2Waves, avB, tmB, avA, tmA);
        if (u + 2u * kF2Waves < u1) step(u + 2u * kF2Waves, avC, tmC, avB, tmB);
      }
    } else {
      for (uint32_t u = u0 + wave; u < u1; u += kF2Waves) step(u, avA, tmA, avA, tmA);
    }
    if (a.stats != nullptr) st_units += 1u;
  }
  if (a.stats != nullptr) atomicAdd(&a.stats[1], st_surv);  // counted per lane
  if (a.stats != nullptr && lane == 0) {
    atomicAdd(&a.stats[0], st_pairs); atomicAdd(&a.stats[2], st_sub);
    if (wave == 0) atomicAdd(&a.stats[7], st_units);
  }
  __syncthreads();
  if (!EMIT && threadIdx.x == 0) a.surv_cnt[blockIdx.x] = min(ctrl[0], a.surv_cap);
}

template <typename T>
__device__ inline void chunk_to_float(const uint4& w, float (&x)[16 / sizeof(T)]);  // (defined with the FLAT kernels below)

// ---- the bound-only head phase of IVF-Flat (L2; 3.1c): bounds and head-pair survivors from the emitted values
// value of a (pair, row): acc = sc^2 (r . d16) - |d|^2 (1 - 2^-9) sc^2 / 2 - larger is nearer, up to the fp16 roundings. The bound of a
// query is NOT derived from the values' error terms (in a 32-d latent space a bound 10 % looser in the squared distance lets 4.6 x
// the rows through: measured, re-score 0.35 -> 2.7 ms): the k rows with the largest values are scored EXACTLY - the re-score's
// chain, t = q - x, acc = fma(t, t, acc) in dimension order - and the largest of those k exact scores is the bound: k rows of the
// list are at or below it, so it bounds the query's k-th best score from above, and it IS the exact head phase's bound whenever
// the screen ranks the list's best k rows first.
struct head_bound_params {
  const float* kth_val;          // [n_head, k] the k largest values of every head pair (select_k, select_min = false)
  const uint32_t* kth_idx;       // [n_head, k] their rows (inside the list)
  const float4* norms;           // [n_head] (|r|^2, ., ., largest scaled operand) from the pre-pass
  const uint32_t* sorted_pairs;  // head pairs first (labels [0, n_lists))
  const uint32_t* pair_off;
  const uint32_t* probes;
  const uint32_t* list_offsets;
  const uint8_t* data;           // the index's rows (interleaved chunks)
  const float* queries;          // the exact chain's queries (rescore_params::rot_queries)
  uint32_t n_lists, n_probes, k, dim, n_chunks;
  float sc, eps, alpha, cbmax, c1;
  uint32_t rot_dim;
  uint32_t* query_kth;           // out: bound keys
  float* thr_head;               // out: [n_head] the head pair's threshold in accumulator units (as pq_bprep_kernel forms it)
  float bound_max;
};
template <typename T>
__global__ __launch_bounds__(256) void flat_head_bound_kernel(const head_bound_params a)
{
  const uint32_t n_head = a.pair_off[a.n_lists];
  const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (i >= n_head) return;  // wave-uniform
  const uint32_t p = a.sorted_pairs[i], q = p / a.n_probes, L = a.probes[p];
  const uint32_t base_row = a.list_offsets[L];
  const float* rq = a.queries + (size_t)q * a.dim;
  float worst = 0.f;   // largest exact score among the k candidates (scores are sums of squares: >= 0)
  bool short_list = false;
  for (uint32_t j0 = 0; j0 < a.k; j0 += 64u) {
    const uint32_t j = j0 + lane;
    if (j < a.k) {
      if (!(a.kth_val[(size_t)i * a.k + j] > -INFINITY)) {
        short_list = true;  // fewer than k rows in the list: no finite bound
      } else {
        const uint32_t row = base_row + a.kth_idx[(size_t)i * a.k + j];
        const uint4* cp = reinterpret_cast<const uint4*>(a.data) + ((size_t)(row >> 6) * a.n_chunks) * 64 + (row & 63u);
        constexpr int VL = 16 / sizeof(T);
        float acc = 0.f;
        for (uint32_t c = 0; c < a.n_chunks; ++c) {
          float x[VL];
          chunk_to_float<T>(cp[(size_t)c * 64], x);
#pragma unroll
          for (int e4 = 0; e4 < VL / 4; ++e4) {
            const float4 qv   = *reinterpret_cast<const float4*>(rq + c * VL + e4 * 4);
            const float qq[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = qq[e] - x[e4 * 4 + e];
              acc = __fmaf_rn(t, t, acc);
            }
          }
        }
        worst = fmaxf(worst, acc);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
  const bool no_bound = __ballot(short_list) != 0ull;
  if (lane != 0u) return;
  const float rn = a.norms[i].x, big = a.norms[i].w;
  const float bound  = no_bound ? INFINITY : worst;
  const uint32_t key = no_bound ? 0xffffffffu : float_to_key(bound);
  a.query_kth[q] = key;
  const bool served = key < 0xff800000u && big < 60000.f && fabsf(bound) <= a.bound_max;
  a.thr_head[i] = served ? filter_threshold(bound, rn, a) / a.c1 : -INFINITY;  // (unserved: everything survives, as in the pre-pass)
}

// the head pair's own candidates: rows of its list whose value reaches the threshold -> the survivor buffer. Runs BEHIND the filter:
// the regions' fills are final, a wave appends to the region (head pair index mod regions) with one atomic - spread over all regions,
// because the re-score gives every region one workgroup column (all of them in the shared region: 8 workgroups for 300 k entries,
// 1.5 ms) - and to the shared region only when its region is full
__global__ __launch_bounds__(256) void flat_head_survivors_kernel(const float* __restrict__ xbuf, uint32_t ldx, const float* __restrict__ thr_head,
                                                                  const uint32_t* __restrict__ sorted_pairs, const uint32_t* __restrict__ pair_off,
                                                                  uint32_t n_lists, const uint32_t* __restrict__ probes,
                                                                  const uint32_t* __restrict__ list_offsets, const uint32_t* __restrict__ list_sizes,
                                                                  uint2* __restrict__ surv, uint32_t* __restrict__ surv_cnt, uint32_t n_regions,
                                                                  uint32_t surv_cap, uint32_t spill_cap, uint32_t* __restrict__ fail)
{
  const uint32_t n_head = pair_off[n_lists];
  const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (i >= n_head) return;  // wave-uniform
  const uint32_t p = sorted_pairs[i], L = probes[p];
  const uint32_t len = list_sizes[L], base_row = list_offsets[L];
  const float th = thr_head[i];
  const float* x = xbuf + (size_t)i * ldx;
  uint32_t total = 0u;
  for (uint32_t r0 = 0; r0 < len; r0 += 64u) {
    const uint32_t r = r0 + lane;
    total += (uint32_t)__popcll(__ballot(r < len && x[r] >= th));
  }
  if (total == 0u) return;
  const uint32_t ri = i % n_regions;
  uint32_t base = 0u;
  if (lane == 0u) base = atomicAdd(surv_cnt + ri, total);
  base = __builtin_amdgcn_readfirstlane(base);
  const uint32_t n_fit = base >= surv_cap ? 0u : min(total, surv_cap - base);  // what the region still holds; the rest: shared region
  uint32_t sbase = 0u;
  if (n_fit < total) {  // wave-uniform
    if (lane == 0u) { atomicMin(surv_cnt + ri, surv_cap); sbase = atomicAdd(surv_cnt + n_regions, total - n_fit); }
    sbase = __builtin_amdgcn_readfirstlane(sbase);
    if (sbase + (total - n_fit) > spill_cap && lane == 0u) *fail = 1u;  // (the batch is re-run on the scan kernel; what fits is still
                                                                         // written: the re-score reads min(count, capacity) entries)
  }
  uint2* region = surv + (size_t)ri * surv_cap;
  uint2* shared = surv + (size_t)n_regions * surv_cap;
  uint32_t done = 0u;
  for (uint32_t r0 = 0; r0 < len; r0 += 64u) {
    const uint32_t r = r0 + lane;
    const bool hit = r < len && x[r] >= th;
    const unsigned long long m = __ballot(hit);
    const uint32_t kth = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (hit) {
      if (kth < n_fit) region[base + kth] = make_uint2(p, base_row + r);
      else if (sbase + (kth - n_fit) < spill_cap) shared[sbase + (kth - n_fit)] = make_uint2(p, base_row + r);
    }
    done += (uint32_t)__popcll(m);
  }
}

__global__ void fill_f32_kernel(float* __restrict__ p, size_t n, float v)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------ re-score
struct rescore_params {
  const uint2* surv;         // regions of surv_cap entries, one per workgroup of the filter
  const uint32_t* surv_cnt;  // [n_regions + 1] fill of every region, of the spill region
  uint32_t surv_cap, spill_cap;
  uint32_t n_regions, sub;   // regions of the filter (one per workgroup), regions per workgroup of this kernel; sub == 0: the
                             // buffer was handed out in chunks of 256 entries (pq_filter4_kernel), surv_cnt[0] = chunks drawn,
                             // unused entries are invalid pairs (0xffffffff)
  const uint32_t* probes;  // [n_pairs] list of every pair
  const float* rot_queries;
  const float* centers_rot;
  const float* pq_centers;
  const uint8_t* codes;
  const uint32_t* query_kth;
  uint32_t* qflag;
  uint32_t* qcnt;
  float* cand_d;
  uint32_t* cand_i;
  uint32_t* cand_r;
  uint32_t n_probes, rot_dim, k, head, n_chunks;
  uint32_t head_rows;             // partial head: survivors of a head pair below this row of its list were scored by the head phase
  const uint32_t* list_offsets;   //   (the list's first flat row)
  int is_ip;
  const uint32_t* filter_bits;
  const int64_t* indices;
  uint4* overflow;        // candidates of queries whose pool is full: (query, score bits, probe rank, row)
  uint32_t* overflow_cnt;
  uint32_t overflow_cap;
  uint32_t* fail;         // IVF-Flat: raised when the overflow list is full (nullptr: the query is flagged instead)
  uint32_t dim;           // IVF-Flat: row length
  int cb_lds;             // IVF-PQ: the fp32 codebook fits the LDS of a workgroup
  uint32_t pq_len, book;
  int per_cluster;        // IVF-PQ: one codebook per list (pq_centers [n_lists][pq_len][book])
  const float* cbt;       // the wide path: the codebook entry-major, [subspace][256][pq_len] fp32 (pq_exact_score_wave)
};

// A re-scored survivor goes into its query's pool if it is within the bound, beyond the pool's capacity (a loose head
// bound) into a list shared by all such queries (binned by query before the merge); only when that is full too is the
// query handed back (IVF-Flat: the batch is re-run). A wave operation - every lane calls it, `want`: this lane holds a
// re-scored survivor: the survivors of a query arrive together (a loose bound means thousands of them in neighbouring
// chunks), so the lanes of a wave that append to the same pool draw their positions with ONE atomic, and so do the lanes
// that run over into the overflow list (one atomic per survivor on a few hot counters was half of the kernel's time)
__device__ inline void pool_append_wave(const rescore_params& a, bool want, const uint32_t q, const uint32_t pair, const uint32_t row,
                                        const float score)
{
  const uint32_t lane = threadIdx.x & 63u;
  want = want && float_to_key(score) <= a.query_kth[want ? q : 0u];
  const uint32_t cap = (a.n_probes - a.head) * a.k;
  bool over = false;
  unsigned long long todo = __ballot(want);
  while (todo != 0ull) {
    const uint32_t q0 = __builtin_amdgcn_readlane(q, (int)__ffsll((long long)todo) - 1);
    const unsigned long long same = __ballot(want && q == q0);
    const uint32_t leader = (uint32_t)__ffsll((long long)same) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(&a.qcnt[q0], (uint32_t)__popcll(same));
    base = __builtin_amdgcn_readlane(base, (int)leader);
    if (want && q == q0) {
      const uint32_t pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      if (pos < cap) {
        const size_t o = (size_t)q * a.n_probes * a.k + (size_t)a.head * a.k + pos;
        a.cand_d[o] = score;
        a.cand_i[o] = row;
        a.cand_r[o] = pair % a.n_probes;
      } else {
        over = true;
      }
    }
    todo &= ~same;
  }
  const unsigned long long om = __ballot(over);
  if (om != 0ull) {
    const uint32_t leader = (uint32_t)__ffsll((long long)om) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(a.overflow_cnt, (uint32_t)__popcll(om));
    base = __builtin_amdgcn_readlane(base, (int)leader);
    if (over) {
      const uint32_t ov = base + (uint32_t)__popcll(om & ((1ull << lane) - 1ull));
      if (ov < a.overflow_cap) a.overflow[ov] = make_uint4(q, __float_as_uint(score), pair % a.n_probes, row);
      else if (a.fail != nullptr) *a.fail = 1u;
      else a.qflag[q] = 1u;
    }
  }
}

constexpr int kRThreads = 1024;

// The score of (query q, row `row` of list L) in the reference's arithmetic for the requested LUT / score types: every LUT entry
// from its components in order (create_lut_impl.cuh:17-78), rounded to the LUT type, summed in subspace order in the score type
// (compute_score_impl.cuh:52-79) - bit-identical to the LUT scan's score. cb: the fp32 codebook staged in LDS (CB_LDS).
template <int LUT, bool ACC_HALF, bool CB_LDS>  // LUT: 0 fp32, 1 fp16, 2 fp8 (fp_8bit<5>)
__device__ inline float pq_exact_score(const rescore_params& a, const float* cb, const uint32_t q, const uint32_t L, const uint32_t row)
{
  const float* pqc = CB_LDS ? cb : a.pq_centers;  // (the generic branch below: any address space)
  float af       = 0.f;
  _Float16 ah    = (_Float16)0.f;
  const float* rq  = a.rot_queries + (size_t)q * a.rot_dim;
  const float* ct  = a.centers_rot + (size_t)L * a.rot_dim;
  const uint4* cp  = reinterpret_cast<const uint4*>(a.codes) + ((size_t)(row >> 6) * a.n_chunks) * 64 + (row & 63u);
  auto add_entry = [&](float v) {  // one LUT entry in the reference's arithmetic: LUT type, then the score type's sum
    if constexpr (LUT == 2) v = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(v, a.is_ip != 0);
    if constexpr (LUT == 0 || (LUT == 2 && !ACC_HALF)) {
      af += v;  // fp32 entries
    } else {
      const _Float16 e = to_lut_half(v);
      if constexpr (ACC_HALF) ah += e; else af += (float)e;
    }
  };
  if (a.pq_len != 2u || a.per_cluster) {
    // any pq_len, PER_CLUSTER codebooks: entry (s, code) = the components' chain in order (create_lut_impl.cuh:17-78), subspaces in order
#pragma unroll 1
    for (int c = 0; c < (int)a.n_chunks; ++c) {
      const uint4 cw       = cp[c * 64];
      const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const uint32_t code = (ws[b >> 2] >> ((b & 3) * 8)) & 0xffu;
        const uint32_t d0   = (uint32_t)(c * 16 + b) * a.pq_len;
        float v = 0.f;
        for (uint32_t l = 0; l < a.pq_len; ++l) {
          const float p = pqc[(size_t)(a.per_cluster ? L * a.pq_len + l : d0 + l) * a.book + code], qv = rq[d0 + l], cv = ct[d0 + l];
          if (!a.is_ip) {
            const float d = (qv - cv) - p;
            v = __fmaf_rn(d, d, v);
          } else {
            v = __fmaf_rn(-qv, cv, v);
            v = __fmaf_rn(-qv, p, v);
          }
        }
        add_entry(v);
      }
    }
  } else {
    // pq_len 2, PER_SUBSPACE: half a chunk (8 subspaces) at a time - its 16 codebook values (LDS when the codebook fits) and
    // 8 query / centre vectors are all requested before the first is used: a thread holds one or two survivors, so the
    // kernel's time is its chains of memory latencies, not its arithmetic
#pragma unroll 1
    for (int c = 0; c < (int)a.n_chunks; ++c) {
      const uint4 cw       = cp[c * 64];
      const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float qq[16], cc[16], p0[8], p1[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 qv = *reinterpret_cast<const float4*>(rq + c * 32 + hh * 16 + j * 4),
                       cv = *reinterpret_cast<const float4*>(ct + c * 32 + hh * 16 + j * 4);
          qq[j * 4] = qv.x; qq[j * 4 + 1] = qv.y; qq[j * 4 + 2] = qv.z; qq[j * 4 + 3] = qv.w;
          cc[j * 4] = cv.x; cc[j * 4 + 1] = cv.y; cc[j * 4 + 2] = cv.z; cc[j * 4 + 3] = cv.w;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int bb        = hh * 8 + b;
          const uint32_t code = (ws[bb >> 2] >> ((bb & 3) * 8)) & 0xffu;
          const uint32_t e0   = (uint32_t)((c * 16 + bb) * 2) * a.book + code;
          if constexpr (CB_LDS) { p0[b] = cb[e0]; p1[b] = cb[e0 + a.book]; }
          else                  { p0[b] = a.pq_centers[e0]; p1[b] = a.pq_centers[e0 + a.book]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const float q0 = qq[b * 2], q1 = qq[b * 2 + 1], c0 = cc[b * 2], c1 = cc[b * 2 + 1];
          float v;
          if (!a.is_ip) {
            const float d0 = (q0 - c0) - p0[b], d1 = (q1 - c1) - p1[b];
            v = __fmaf_rn(d1, d1, __fmaf_rn(d0, d0, 0.f));
          } else {
            v = __fmaf_rn(-q0, c0, 0.f);
            v = __fmaf_rn(-q0, p0[b], v);
            v = __fmaf_rn(-q1, c1, v);
            v = __fmaf_rn(-q1, p1[b], v);
          }
          add_entry(v);
        }
      }
    }
  }
  return ACC_HALF ? (float)ah : af;
}

// The codebook (128 KiB of fp32 at pq_dim 64) is staged in LDS once per workgroup: a survivor's 128 codebook values were
// 128 scattered L2 reads per lane; the query / centre values of a 16-subspace chunk are read as 16-byte vectors.
template <int LUT, bool ACC_HALF, bool CB_LDS>  // LUT: 0 fp32, 1 fp16, 2 fp8 (fp_8bit<5>); CB_LDS: the codebook is staged in LDS
__global__ __launch_bounds__(kRThreads) void pq_rescore_kernel(const rescore_params a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cb = reinterpret_cast<float*>(smem);  // [pq_dim * pq_len][256] when it fits (cb_lds), else read from memory
  if constexpr (CB_LDS) {
    const uint32_t n4 = a.n_chunks * 16u * a.pq_len * a.book / 4u;
    for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x)
      reinterpret_cast<float4*>(cb)[i] = reinterpret_cast<const float4*>(a.pq_centers)[i];
    __syncthreads();
  }
  // region blockIdx.x of the filter's workgroups, the last workgroup takes the shared spill region; chunked buffer
  // (pq_filter4_kernel): all workgroups stride over the chunks drawn
  const bool chunked = a.sub == 0u;
  const bool spill = blockIdx.x + 1 == gridDim.x;
  const uint32_t ri = spill ? a.n_regions : blockIdx.x;
  const uint32_t n = chunked ? min(a.surv_cnt[0], a.surv_cap / 256u) * 256u : (spill ? min(a.surv_cnt[ri], a.spill_cap) : a.surv_cnt[ri]);
  const uint2* region = chunked ? a.surv : a.surv + (size_t)ri * a.surv_cap;
  const uint32_t s_first  = chunked ? (blockIdx.x * gridDim.y + blockIdx.y) * blockDim.x + threadIdx.x : blockIdx.y * blockDim.x + threadIdx.x;
  const uint32_t s_stride = chunked ? gridDim.x * gridDim.y * blockDim.x : gridDim.y * blockDim.x;
  const uint32_t lane_ = threadIdx.x & 63u;
  for (uint32_t sb = s_first - lane_; sb < n; sb += s_stride) {  // (wave-uniform trip count: pool_append_wave is a wave operation)
    const uint32_t s = sb + lane_;
    uint2 sv = make_uint2(0xffffffffu, 0u);
    if (s < n) sv = region[s];
    bool ok = sv.x != 0xffffffffu;  // (padding of a chunk's tail)
    const uint32_t pair = ok ? sv.x : 0u, row = ok ? sv.y : 0u, q = pair / a.n_probes;
    ok = ok && a.qflag[q] == 0u;    // flagged: re-done by the LUT scan
    if (ok && a.filter_bits != nullptr) {
      const int64_t sid = a.indices[row];
      ok = ((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u) != 0u;
    }
    // partial head: the first head_rows rows of a head pair's list were scored by the head phase (no duplicates)
    if (ok && a.head_rows != 0u && pair % a.n_probes < a.head) ok = row - a.list_offsets[a.probes[pair]] >= a.head_rows;
    float score = 0.f;
    if (ok) score = pq_exact_score<LUT, ACC_HALF, CB_LDS>(a, cb, q, a.probes[pair], row);
    pool_append_wave(a, ok, q, pair, row, score);
  }
}

// ------------------------------------------------------------------ the exact score by a whole wave (the wide path)
// pq_exact_score's generic branch gives every LANE a survivor: its pq_dim x pq_len codebook values, query and centre components are
// scalar loads of 64 unrelated (query, list, row)s per instruction - 51 ms for 15.8 M survivors at pq_dim 64 x pq_len 12 (3.3 ns each,
// measured: most of the wide path's batch). Here a WAVE takes a survivor: lane = subspace (64 at a time) - its code byte, then per
// component one codebook value and the query's / centre's component (48-byte stride across the lanes: whole lines) - the entry in the
// LUT type, and the sum over the subspaces in subspace order in the score type, as the reference's loop (compute_score_impl.cuh:52-79)
// adds them: lane by lane, every lane computing the same chain (v_readlane + add). Bit-identical to pq_exact_score.
// cache: the residual (query - centre; L2) of the lane's subspace for the pair (q, L) scored last, kept across calls while the index has
// at most 64 subspaces: consecutive survivors of a region mostly belong to the same pair (a head pair's rows; a bound's candidates from
// the nearest list), and two thirds of a score's bytes were the query's and the centre's components read again
struct wave_score_cache {
  uint32_t q = 0xffffffffu, L = 0xffffffffu;
  float r[16], c[16];  // L2: the residual q - c; inner product: the query's and the centre's components
};
template <int LUT, bool ACC_HALF>
__device__ inline float pq_exact_score_wave(const rescore_params& a, const uint32_t q, const uint32_t L, const uint32_t row)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t pq_dim = a.n_chunks * 16u;
  const float* rq = a.rot_queries + (size_t)q * a.rot_dim;
  const float* ct = a.centers_rot + (size_t)L * a.rot_dim;
  const uint8_t* cr = a.codes + ((size_t)(row >> 6) * a.n_chunks) * 1024 + (size_t)(row & 63u) * 16;
  float af    = 0.f;
  _Float16 ah = (_Float16)0.f;
  for (uint32_t s0 = 0; s0 < pq_dim; s0 += 64u) {  // wave-uniform
    const uint32_t sub = min(s0 + lane, pq_dim - 1u);
    const uint32_t code = cr[(size_t)(sub >> 4) * 1024 + (sub & 15u)];
    const uint32_t d0 = sub * a.pq_len;
    // the codebook entry from the entry-major copy (a.cbt: [subspace][code][component], 4 pq_len bytes side by side): with the
    // index's own layout ([subspace * pq_len + component][code]) every lane read a different 1 KiB row per component - 64 lines per
    // load instruction, pq_len times; here a lane's whole entry, the query's and the centre's components are 16-byte loads, all
    // asked for before the first is used
    const float* e = a.cbt + ((size_t)sub * 256 + code) * a.pq_len;
    float v = 0.f;
    auto add = [&](const float p, const float qv, const float cv) {
      if (!a.is_ip) {
        const float d = (qv - cv) - p;
        v = __fmaf_rn(d, d, v);
      } else {
        v = __fmaf_rn(-qv, cv, v);
        v = __fmaf_rn(-qv, p, v);
      }
    };
    if ((a.pq_len & 3u) == 0u && a.pq_len <= 16u) {  // wave-uniform
      float4 p4[4], q4[4], c4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t o = min((uint32_t)i * 4u, a.pq_len - 4u);  // (past the entry: a repeat of its last piece, not used)
        p4[i] = *reinterpret_cast<const float4*>(e + o);
        q4[i] = *reinterpret_cast<const float4*>(rq + d0 + o);
        c4[i] = *reinterpret_cast<const float4*>(ct + d0 + o);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((uint32_t)i * 4u < a.pq_len) {
          add(p4[i].x, q4[i].x, c4[i].x); add(p4[i].y, q4[i].y, c4[i].y); add(p4[i].z, q4[i].z, c4[i].z); add(p4[i].w, q4[i].w, c4[i].w);
        }
      }
    } else {
      for (uint32_t l = 0; l < a.pq_len; ++l) add(e[l], rq[d0 + l], ct[d0 + l]);
    }
    // the entry as the LUT holds it
    if constexpr (LUT == 2) v = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(v, a.is_ip != 0);
    uint32_t bits;
    if constexpr (LUT == 0 || (LUT == 2 && !ACC_HALF)) bits = __float_as_uint(v);
    else bits = (uint32_t)__builtin_bit_cast(uint16_t, to_lut_half(v));
    auto chain = [&](const uint32_t eb) {
      if constexpr (LUT == 0 || (LUT == 2 && !ACC_HALF)) {
        af += __uint_as_float(eb);
      } else {
        const _Float16 e = __builtin_bit_cast(_Float16, (uint16_t)eb);
        if constexpr (ACC_HALF) ah += e; else af += (float)e;
      }
    };
    const uint32_t n_here = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, pq_dim - s0));
    if (n_here == 64u) {  // the usual case: lane indices in the instructions (a loop with a run-time bound was five instructions and a branch per entry)
#pragma unroll
      for (int j = 0; j < 64; ++j) chain(__builtin_amdgcn_readlane(bits, j));
    } else {
      for (uint32_t j = 0; j < n_here; ++j) chain(__builtin_amdgcn_readlane(bits, j));
    }
  }
  return ACC_HALF ? (float)ah : af;
}

// (Measured and rejected for pq_len 12: THREE lanes per codebook entry - a step = 16 subspaces on 48 lanes, one load fetching 16 whole
// 48-byte entries instead of 64 lanes x one line each, the entry's 12-term chain handed from lane to lane by DPP - is bit-identical but a
// score becomes four dependent steps instead of one: 1M x 768, pq_dim 64: 2.75 -> 3.61 ms per 10 k queries.)
// Up to 64 (query, list, row) items, one per lane (`ok`), scored one after the other by the whole wave; lane j receives item j's score.
// L2 with pq_len 2 / 4 / 8 / 12 / 16 runs as a software pipeline over STEPS = (item, block of 64 subspaces): a step is a chain of two
// dependent memory round trips (the row's code bytes, then the codebook entries they name - and, beyond 64 subspaces, the block's
// residual components): the entries of step t + 1 and the codes of step t + 2 are asked for BEFORE step t's entries are formed
// (unconditionally: past the last step the last one is repeated - a conditional load would make the compiler wait for everything
// outstanding at the branch's end).
// The sum over the subspaces, in subspace order, is NOT done item by item (64 x (v_readlane + add) per item was most of a score):
// item i's entries go to row i of a tile of this wave in LDS (rows of pq_dim entries + padding; as many items per pass as the tile
// holds), and when a pass is through every lane that owns one of its items reads that row and adds the entries up one after the
// other - one chain per lane.
template <int LUT, bool ACC_HALF>
struct wave_score_tile {
  static constexpr bool kHalf = !(LUT == 0 || (LUT == 2 && !ACC_HALF));  // entries are fp16 values
  using ent_t = std::conditional_t<kHalf, uint16_t, uint32_t>;
  static constexpr uint32_t kWords = 64u * 33u * (kHalf ? 1u : 2u) + 64u;  // 32-bit words of a wave's tile (64 rows of 64 entries + padding)
  static constexpr size_t kBytes = (size_t)kWords * 4u;
};

template <int LUT, bool ACC_HALF, int PL, bool IP>
__device__ inline float wave_score_batch_pl(const rescore_params& a, const bool ok, const uint32_t q, const uint32_t L, const uint32_t row,
                                            wave_score_cache& cache, void* lds_tile)
{
  using tile = wave_score_tile<LUT, ACC_HALF>;
  typename tile::ent_t* ent = static_cast<typename tile::ent_t*>(lds_tile);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t pq_dim = a.n_chunks * 16u, nb = (pq_dim + 63u) >> 6;  // nb: blocks of 64 subspaces
  // a row: pq_dim entries, an odd number of 32-bit words (conflict-free when the lanes read their rows side by side)
  const uint32_t row_words = ((tile::kHalf ? (pq_dim + 1u) / 2u : pq_dim) | 1u), stride = tile::kHalf ? 2u * row_words : row_words;
  const uint32_t per_pass = min(64u, tile::kWords / row_words);  // items per pass (>= 1: pq_dim up to 4096)
  float score = 0.f;
  const unsigned long long all_items = __ballot(ok);
  if (all_items == 0ull) return score;
  const uint32_t n_items = (uint32_t)__popcll(all_items);
  const uint32_t mine = (uint32_t)__popcll(all_items & ((1ull << lane) - 1ull));  // this lane's item is the mine-th
  constexpr int NV = PL >= 4 ? PL / 4 : 1;                                         // 16-byte (pq_len 2: 8-byte) pieces of an entry
  struct piece { float x[PL]; };
  auto load_pl = [&](const float* p) {
    piece o;
    if constexpr (PL == 2) {
      const float2 v = *reinterpret_cast<const float2*>(p);
      o.x[0] = v.x; o.x[1] = v.y;
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
        o.x[4 * i] = v.x; o.x[4 * i + 1] = v.y; o.x[4 * i + 2] = v.z; o.x[4 * i + 3] = v.w;
      }
    }
    return o;
  };
  for (uint32_t base = 0; base < n_items; base += per_pass) {  // a pass: the items [base, base + per_pass) of the batch
    const uint32_t n_pass = min(per_pass, n_items - base);
    // the lanes of this pass's items, in order
    unsigned long long todo = all_items;
    for (uint32_t i = 0; i < base; ++i) todo &= todo - 1ull;
    int j_cur = (int)__ffsll((long long)todo) - 1;
    uint32_t left = n_pass;
    // step cursor: (item lane, block); past the end: the last step again
    struct cursor { int j; uint32_t blk; };
    cursor nxt{j_cur, 0u};
    auto advance = [&]() {
      const cursor c = nxt;
      if (nxt.blk + 1u < nb) { nxt.blk += 1u; }
      else if (left > 1u) { left -= 1u; todo &= todo - 1ull; nxt.j = (int)__ffsll((long long)todo) - 1; nxt.blk = 0u; }
      return c;
    };
    auto sub_of = [&](const cursor c) { return min(c.blk * 64u + lane, pq_dim - 1u); };
    auto load_code = [&](const cursor c) {
      const uint32_t r = __builtin_amdgcn_readlane(row, c.j), sub = sub_of(c);
      const uint8_t* cr = a.codes + ((size_t)(r >> 6) * a.n_chunks) * 1024 + (size_t)(r & 63u) * 16;
      return (uint32_t)cr[(size_t)(sub >> 4) * 1024 + (sub & 15u)];
    };
    auto load_entry = [&](const cursor c, const uint32_t code) { return load_pl(a.cbt + ((size_t)sub_of(c) * 256 + code) * PL); };
    // the block's residual components q - c (L2); inner product: the query's components, and the centre's (load_ctr)
    auto load_res = [&](const cursor c) {
      const uint32_t qj = __builtin_amdgcn_readlane(q, c.j), Lj = __builtin_amdgcn_readlane(L, c.j), d0 = sub_of(c) * PL;
      const piece qv = load_pl(a.rot_queries + (size_t)qj * a.rot_dim + d0);
      if constexpr (IP) return qv;
      const piece cv = load_pl(a.centers_rot + (size_t)Lj * a.rot_dim + d0);
      piece o;
#pragma unroll
      for (int l = 0; l < PL; ++l) o.x[l] = qv.x[l] - cv.x[l];
      return o;
    };
    auto load_ctr = [&](const cursor c) {
      const uint32_t Lj = __builtin_amdgcn_readlane(L, c.j), d0 = sub_of(c) * PL;
      return load_pl(a.centers_rot + (size_t)Lj * a.rot_dim + d0);
    };
    const uint32_t n_steps = n_pass * nb;
    cursor c0 = advance(), c1 = advance();
    uint32_t code1 = load_code(c1);
    piece p_cur = load_entry(c0, load_code(c0)), r_cur{}, p_nxt, r_nxt{}, c_cur{}, c_nxt{};
    if (nb > 1u) {
      r_cur = load_res(c0);
      if constexpr (IP) c_cur = load_ctr(c0);
    }
    uint32_t it = 0;  // position of c0's item in the pass
    for (uint32_t t = 0; t < n_steps; ++t) {
      if (nb == 1u) {  // wave-uniform: the pair's residual, kept across items and calls
        const uint32_t qj = __builtin_amdgcn_readlane(q, c0.j), Lj = __builtin_amdgcn_readlane(L, c0.j);
        if (cache.q != qj || cache.L != Lj) {
          const piece rr = load_res(c0);
#pragma unroll
          for (int l = 0; l < PL; ++l) cache.r[l] = rr.x[l];
          if constexpr (IP) {
            const piece cc = load_ctr(c0);
#pragma unroll
            for (int l = 0; l < PL; ++l) cache.c[l] = cc.x[l];
          }
          cache.q = qj; cache.L = Lj;
        }
      }
      p_nxt = load_entry(c1, code1);  // step t + 1
      if (nb > 1u) {
        r_nxt = load_res(c1);
        if constexpr (IP) c_nxt = load_ctr(c1);
      }
      const cursor c2 = advance();
      const uint32_t code2 = load_code(c2);  // step t + 2
      float v = 0.f;
#pragma unroll
      for (int l = 0; l < PL; ++l) {
        const float rl = nb == 1u ? cache.r[l] : r_cur.x[l];
        if constexpr (!IP) {
          const float d = rl - p_cur.x[l];
          v = __fmaf_rn(d, d, v);
        } else {  // (create_lut_impl.cuh:17-78: -q c - q p, component by component)
          v = __fmaf_rn(-rl, nb == 1u ? cache.c[l] : c_cur.x[l], v);
          v = __fmaf_rn(-rl, p_cur.x[l], v);
        }
      }
      if constexpr (LUT == 2) v = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(v, IP);
      uint32_t bits;
      if constexpr (!tile::kHalf) bits = __float_as_uint(v);
      else bits = (uint32_t)__builtin_bit_cast(uint16_t, to_lut_half(v));
      if (c0.blk * 64u + lane < pq_dim) ent[it * stride + c0.blk * 64u + lane] = (typename tile::ent_t)bits;  // row = the item's position in the pass
      if (c0.blk + 1u == nb) it += 1u;
      c0 = c1; c1 = c2; code1 = code2; p_cur = p_nxt;
      if (nb > 1u) {
        r_cur = r_nxt;
        if constexpr (IP) c_cur = c_nxt;
      }
    }
    if (ok && mine >= base && mine < base + n_pass) {
      const typename tile::ent_t* my = ent + (mine - base) * stride;
      float af    = 0.f;
      _Float16 ah = (_Float16)0.f;
      for (uint32_t s2 = 0; s2 < pq_dim; ++s2) {
        if constexpr (!tile::kHalf) {
          af += __uint_as_float(my[s2]);
        } else {
          const _Float16 e = __builtin_bit_cast(_Float16, my[s2]);
          if constexpr (ACC_HALF) ah += e; else af += (float)e;
        }
      }
      score = ACC_HALF ? (float)ah : af;
    }
  }
  return score;
}

template <int LUT, bool ACC_HALF>
__device__ inline float wave_score_batch(const rescore_params& a, const bool ok, const uint32_t q, const uint32_t L, const uint32_t row,
                                         wave_score_cache& cache, void* lds_tile)
{
  if (!a.per_cluster) {  // wave-uniform
    auto go = [&](auto pl_tag) {
      constexpr int PL = decltype(pl_tag)::value;
      return a.is_ip ? wave_score_batch_pl<LUT, ACC_HALF, PL, true>(a, ok, q, L, row, cache, lds_tile)
                     : wave_score_batch_pl<LUT, ACC_HALF, PL, false>(a, ok, q, L, row, cache, lds_tile);
    };
    switch (a.pq_len) {
      case 2:  return go(std::integral_constant<int, 2>{});
      case 4:  return go(std::integral_constant<int, 4>{});
      case 8:  return go(std::integral_constant<int, 8>{});
      case 12: return go(std::integral_constant<int, 12>{});
      case 16: return go(std::integral_constant<int, 16>{});
      default: break;
    }
  }
  const uint32_t lane = threadIdx.x & 63u;
  float score = 0.f;
  unsigned long long todo = __ballot(ok);
  while (todo != 0ull) {
    const int j = (int)__ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const float sc = pq_exact_score_wave<LUT, ACC_HALF>(a, __builtin_amdgcn_readlane(q, j), __builtin_amdgcn_readlane(L, j),
                                                        __builtin_amdgcn_readlane(row, j));
    if ((int)lane == j) score = sc;
  }
  return score;
}

// the re-score with a wave per survivor: a wave takes 64 consecutive survivors of its region, scores them one after the other
// (lane j keeps the score of the j-th) and appends them together (pool_append_wave)
template <int LUT, bool ACC_HALF>
__global__ __launch_bounds__(256) void pq_rescore_wave_kernel(const rescore_params a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];  // one wave_score_tile per wave, then the regions' batch offsets
  // The regions' fills differ (a filter workgroup whose lists are dense leaves several times the survivors of another; a wave works
  // through its 64 survivors one after the other): the batches of 64 of ALL regions are numbered through and dealt to the waves of the
  // whole grid round-robin (a region's own 32 waves: 2.9 ms for 1 M survivors at 1M x 768, the longest region's time)
  uint32_t* s_off = reinterpret_cast<uint32_t*>(smem + 4 * wave_score_tile<LUT, ACC_HALF>::kBytes);  // [n_regions + 2] first batch of every region
  const uint32_t n_reg = a.n_regions + 1u;  // + the shared spill region
  if (threadIdx.x == 0) {
    uint32_t run = 0u;
    for (uint32_t r = 0; r < n_reg; ++r) {
      s_off[r] = run;
      const uint32_t n = r == a.n_regions ? min(a.surv_cnt[r], a.spill_cap) : min(a.surv_cnt[r], a.surv_cap);
      run += (n + 63u) >> 6;
    }
    s_off[n_reg] = run;
  }
  __syncthreads();
  const uint32_t total = s_off[n_reg];
  const uint32_t n_waves = gridDim.x * gridDim.y * (blockDim.x >> 6);
  const uint32_t wave0 = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  wave_score_cache cache;
  for (uint32_t b = wave0; b < total; b += n_waves) {  // wave-uniform
    uint32_t lo = 0u, hi = n_reg;  // s_off[lo] <= b < s_off[hi]
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_off[mid] <= b) lo = mid; else hi = mid;
    }
    const uint32_t ri = lo;
    const uint32_t n = ri == a.n_regions ? min(a.surv_cnt[ri], a.spill_cap) : min(a.surv_cnt[ri], a.surv_cap);
    const uint2* region = a.surv + (size_t)ri * a.surv_cap;
    const uint32_t s = (b - s_off[ri]) * 64u + lane;
    uint2 sv = make_uint2(0xffffffffu, 0u);
    if (s < n) sv = region[s];
    bool ok = sv.x != 0xffffffffu;
    const uint32_t pair = ok ? sv.x : 0u, row = ok ? sv.y : 0u, q = pair / a.n_probes;
    ok = ok && a.qflag[q] == 0u;  // flagged: re-done by the LUT scan
    if (ok && a.filter_bits != nullptr) {
      const int64_t sid = a.indices[row];
      ok = ((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u) != 0u;
    }
    const uint32_t L = ok ? a.probes[pair] : 0u;
    const float score = wave_score_batch<LUT, ACC_HALF>(a, ok, q, L, row, cache, smem + (threadIdx.x >> 6) * wave_score_tile<LUT, ACC_HALF>::kBytes);
    pool_append_wave(a, ok, q, pair, row, score);
  }
}

// ------------------------------------------------------------------ re-score with the codebook staged block by block (pq_len 2)
// pq_rescore_wave_kernel reads a survivor's codebook entries from memory: at pq_len 2 that is an 8-byte piece per subspace, each in a
// line of its own - 268 M L2 requests for ~1 M survivors at pq_dim 384 (profiles/r06_pmc_wide.json), the kernel's bound. The narrow
// path's pq_rescore_kernel keeps the whole fp32 codebook in LDS (128 KiB at pq_dim 64); beyond that it does not fit - but a BLOCK of 64
// subspaces does, and the score is a chain over the subspaces in order: a thread keeps the running sums of its survivors in
// registers, the workgroup stages block after block (128 KiB, coalesced) and every thread adds the block's 64 entries of each of its
// survivors - looked up in LDS - to their sums, in subspace order. Same arithmetic as pq_exact_score's pq_len-2 branch, entry by entry.
constexpr int kRBThreads = 1024;  // 16 waves, one workgroup per CU
constexpr int kRBItems   = 2;     // survivors per thread and round: 2048 per workgroup between two stagings of the codebook

template <int LUT, bool ACC_HALF>
__global__ __launch_bounds__(kRBThreads) void pq_rescore_blocks_kernel(const rescore_params a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cb = reinterpret_cast<float*>(smem);  // the block's codebook: [64 subspaces][2 components][book]
  const uint32_t blk_floats = 64u * 2u * a.book;
  uint32_t* s_off = reinterpret_cast<uint32_t*>(smem + (size_t)blk_floats * 4u);  // [n_regions + 2] first batch of every region
  const uint32_t n_reg = a.n_regions + 1u;
  if (threadIdx.x == 0) {
    uint32_t run = 0u;
    for (uint32_t r = 0; r < n_reg; ++r) {
      s_off[r] = run;
      const uint32_t n = r == a.n_regions ? min(a.surv_cnt[r], a.spill_cap) : min(a.surv_cnt[r], a.surv_cap);
      run += (n + 63u) >> 6;
    }
    s_off[n_reg] = run;
  }
  __syncthreads();
  const uint32_t total = s_off[n_reg];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = kRBThreads / 64;
  const uint32_t nb = a.n_chunks / 4u;  // blocks of 64 subspaces (pq_dim a multiple of 64)
  const uint32_t per_round = n_waves * kRBItems;
  for (uint32_t b0 = blockIdx.x * per_round; b0 < total; b0 += gridDim.x * per_round) {  // workgroup-uniform
    bool ok[kRBItems];
    uint32_t pair[kRBItems], row[kRBItems], q[kRBItems], L[kRBItems];
    float af[kRBItems];
    _Float16 ah[kRBItems];
#pragma unroll
    for (int i = 0; i < kRBItems; ++i) {
      const uint32_t b = b0 + (uint32_t)i * n_waves + wave;  // wave-uniform
      uint2 sv = make_uint2(0xffffffffu, 0u);
      if (b < total) {
        uint32_t lo = 0u, hi = n_reg;  // s_off[lo] <= b < s_off[hi]
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_off[mid] <= b) lo = mid; else hi = mid;
        }
        const uint32_t n  = lo == a.n_regions ? min(a.surv_cnt[lo], a.spill_cap) : min(a.surv_cnt[lo], a.surv_cap);
        const uint32_t sx = (b - s_off[lo]) * 64u + lane;
        if (sx < n) sv = a.surv[(size_t)lo * a.surv_cap + sx];
      }
      ok[i]   = sv.x != 0xffffffffu;
      pair[i] = ok[i] ? sv.x : 0u; row[i] = ok[i] ? sv.y : 0u; q[i] = pair[i] / a.n_probes;
      ok[i]   = ok[i] && a.qflag[q[i]] == 0u;  // flagged: re-done by the LUT scan
      if (ok[i] && a.filter_bits != nullptr) {
        const int64_t sid = a.indices[row[i]];
        ok[i] = ((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u) != 0u;
      }
      L[i]  = ok[i] ? a.probes[pair[i]] : 0u;
      af[i] = 0.f; ah[i] = (_Float16)0.f;
    }
    for (uint32_t blk = 0; blk < nb; ++blk) {
      __syncthreads();  // the previous block's lookups are done
      {
        const float4* src = reinterpret_cast<const float4*>(a.pq_centers + (size_t)blk * blk_floats);
        for (uint32_t i = threadIdx.x; i < blk_floats / 4u; i += kRBThreads) reinterpret_cast<float4*>(cb)[i] = src[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kRBItems; ++i) {
        if (!ok[i]) continue;
        const float* rq = a.rot_queries + (size_t)q[i] * a.rot_dim;
        const float* ct = a.centers_rot + (size_t)L[i] * a.rot_dim;
        const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + ((size_t)(row[i] >> 6) * a.n_chunks) * 64 + (row[i] & 63u);
        auto add_entry = [&](float v) {  // one LUT entry in the reference's arithmetic: LUT type, then the score type's sum
          if constexpr (LUT == 2) v = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(v, a.is_ip != 0);
          if constexpr (LUT == 0 || (LUT == 2 && !ACC_HALF)) {
            af[i] += v;
          } else {
            const _Float16 e = to_lut_half(v);
            if constexpr (ACC_HALF) ah[i] += e; else af[i] += (float)e;
          }
        };
#pragma unroll 1
        for (int cc = 0; cc < 4; ++cc) {
          const int c          = (int)blk * 4 + cc;
          const uint4 cw       = cp[c * 64];
          const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float qq[16], cv[16], p0[8], p1[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 qv = *reinterpret_cast<const float4*>(rq + c * 32 + hh * 16 + j * 4),
                           c4 = *reinterpret_cast<const float4*>(ct + c * 32 + hh * 16 + j * 4);
              qq[j * 4] = qv.x; qq[j * 4 + 1] = qv.y; qq[j * 4 + 2] = qv.z; qq[j * 4 + 3] = qv.w;
              cv[j * 4] = c4.x; cv[j * 4 + 1] = c4.y; cv[j * 4 + 2] = c4.z; cv[j * 4 + 3] = c4.w;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
              const int bb        = hh * 8 + b;
              const uint32_t code = (ws[bb >> 2] >> ((bb & 3) * 8)) & 0xffu;
              const uint32_t e0   = (uint32_t)((cc * 16 + bb) * 2) * a.book + code;  // (inside the staged block)
              p0[b] = cb[e0]; p1[b] = cb[e0 + a.book];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
              const float q0 = qq[b * 2], q1 = qq[b * 2 + 1], c0 = cv[b * 2], c1 = cv[b * 2 + 1];
              float v;
              if (!a.is_ip) {
                const float d0 = (q0 - c0) - p0[b], d1 = (q1 - c1) - p1[b];
                v = __fmaf_rn(d1, d1, __fmaf_rn(d0, d0, 0.f));
              } else {
                v = __fmaf_rn(-q0, c0, 0.f);
                v = __fmaf_rn(-q0, p0[b], v);
                v = __fmaf_rn(-q1, c1, v);
                v = __fmaf_rn(-q1, p1[b], v);
              }
              add_entry(v);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kRBItems; ++i) pool_append_wave(a, ok[i], q[i], pair[i], row[i], ACC_HALF ? (float)ah[i] : af[i]);
  }
}

// ------------------------------------------------------------------ the wide path's bound-only head phase (ivf_pq_wide.hip)
// As IVF-Flat's (flat_head_bound_kernel): the emit pass left a value for every (head pair, row) - larger is nearer, comparable
// across a query's head lists (the pair's constant -|r|^2 sc^2 / 2 is added) - and select_k the k largest of every QUERY over the
// union of its `heads` lists. Those k rows are scored EXACTLY, in the reference's arithmetic: the largest of the k scores bounds
// the query's k-th best score from above (k rows are at or below it). One wave per query.
struct wbound_params {
  rescore_params rs;  // the exact score's inputs
  const float* kth_val;
  const uint32_t* kth_idx;  // position in the query's value row: probe rank * ldx + row of the list
  const float4* norms;      // [query * heads + rank] (|r|^2, ., ., largest scaled operand)
  const uint32_t* probes;
  const uint32_t* list_offsets;
  int64_t nq;
  uint32_t k, heads, ldx, n_probes, rot_dim;
  uint32_t* query_kth;
  uint32_t* qflag;
  float* thr_head;          // out: [query * heads + rank] threshold in the units of the value buffer
  float sc, c1, eps, alpha, cbmax, dmax, bound_max;
  int is_ip;
};
template <int LUT, bool ACC_HALF>
__global__ __launch_bounds__(256) void pqw_head_bound_kernel(const wbound_params a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];  // one wave_score_tile per wave
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (q >= a.nq) return;  // wave-uniform
  float worst = -INFINITY;  // (wave-uniform)
  bool short_list = false;
  wave_score_cache cache;
  for (uint32_t j0 = 0; j0 < a.k; j0 += 64u) {
    const uint32_t j = j0 + lane;
    bool have = false;
    uint32_t L = 0u, row = 0u;
    if (j < a.k) {
      if (!(a.kth_val[(size_t)q * a.k + j] > -INFINITY)) {
        short_list = true;  // fewer than k rows in the head lists: no finite bound
      } else {
        const uint32_t id = a.kth_idx[(size_t)q * a.k + j], rank = id / a.ldx, r = id - rank * a.ldx;
        L    = a.probes[(size_t)q * a.n_probes + rank];
        row  = a.list_offsets[L] + r;
        have = true;
      }
    }
    // a wave per candidate (wave_score_batch); lane j holds candidate j's score
    float sc = wave_score_batch<LUT, ACC_HALF>(a.rs, have, (uint32_t)q, L, row, cache, smem + (threadIdx.x >> 6) * wave_score_tile<LUT, ACC_HALF>::kBytes);
    sc = have ? sc : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sc = fmaxf(sc, __shfl_xor(sc, o));
    worst = fmaxf(worst, sc);
  }
  const bool no_bound = __ballot(short_list) != 0ull;
  const float bound  = no_bound ? INFINITY : worst;
  const uint32_t key = no_bound ? 0xffffffffu : float_to_key(bound);
  bool hand_back = no_bound;
  for (uint32_t rank = lane; rank < a.heads; rank += 64u) {
    const float4 nm   = a.norms[(size_t)q * a.heads + rank];
    const bool served = key < 0xff800000u && nm.w < 60000.f && fabsf(bound) <= a.bound_max;
    hand_back = hand_back || !served;
    // (the same expression as the emit pass's constant: x = fl(acc + c) >= fl(t + c) whenever acc >= t)
    // nm = (|r|^2 or |q|^2, |c|^2, q.c, .); + the pair's constant of the emit pass (the same expression as pqw_bprep_kernel's)
    const float t = a.is_ip ? filter_threshold_ip(bound, nm.x, nm.y, nm.z, a) : filter_threshold(bound, nm.x, a);
    a.thr_head[(size_t)q * a.heads + rank] = served ? t / a.c1 + (a.is_ip ? a.sc * a.sc * nm.z : -0.5f * a.sc * a.sc * nm.x) : INFINITY;
  }
  const bool any_back = __ballot(hand_back) != 0ull;
  if (lane == 0u) {
    a.query_kth[q] = key;
    if (any_back) a.qflag[q] = 1u;  // no bound, or a pair the filter cannot serve: the query goes back to the LUT scan, all its pairs
  }
}

// pq_len 2: the bound's candidates scored with the codebook staged block by block in LDS (pq_rescore_blocks_kernel's scheme: a wave
// per candidate reads 8-byte entries one line each; 0.37 ms of a 3.2 ms search at k = 10, most of the head phase at k = 100). A thread
// takes candidates (query q, j-th best value); the largest exact score of a query's k candidates is collected by atomicMax on its
// order-preserving key (tmp[q], zeroed), a candidate slot without a row raises tmp[nq + q]; pqw_bound_finish_kernel turns both into
// the query's bound and the head pairs' thresholds.
template <int LUT, bool ACC_HALF>
__global__ __launch_bounds__(kRBThreads) void pqw_bound_blocks_kernel(const wbound_params a, uint32_t* __restrict__ tmp)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cb = reinterpret_cast<float*>(smem);  // the block's codebook: [64 subspaces][2 components][book]
  const rescore_params& rs = a.rs;
  const uint32_t blk_floats = 64u * 2u * rs.book;
  const uint32_t nb = rs.n_chunks / 4u;
  const uint64_t total = (uint64_t)a.nq * a.k;
  const uint64_t per_round = (uint64_t)kRBThreads * kRBItems;
  for (uint64_t i0 = (uint64_t)blockIdx.x * per_round; i0 < total; i0 += (uint64_t)gridDim.x * per_round) {  // workgroup-uniform
    bool ok[kRBItems];
    uint32_t q[kRBItems], L[kRBItems], row[kRBItems];
    float af[kRBItems];
    _Float16 ah[kRBItems];
#pragma unroll
    for (int i = 0; i < kRBItems; ++i) {
      const uint64_t ix = i0 + (uint64_t)i * kRBThreads + threadIdx.x;
      ok[i] = false; q[i] = 0u; L[i] = 0u; row[i] = 0u; af[i] = 0.f; ah[i] = (_Float16)0.f;
      if (ix < total) {
        q[i] = (uint32_t)(ix / a.k);
        if (!(a.kth_val[ix] > -INFINITY)) {
          tmp[a.nq + q[i]] = 1u;  // fewer than k rows in the head lists: no finite bound
        } else {
          const uint32_t id = a.kth_idx[ix], rank = id / a.ldx, r = id - rank * a.ldx;
          L[i]   = a.probes[(size_t)q[i] * a.n_probes + rank];
          row[i] = a.list_offsets[L[i]] + r;
          ok[i]  = true;
        }
      }
    }
    for (uint32_t blk = 0; blk < nb; ++blk) {
      __syncthreads();
      {
        const float4* src = reinterpret_cast<const float4*>(rs.pq_centers + (size_t)blk * blk_floats);
        for (uint32_t i = threadIdx.x; i < blk_floats / 4u; i += kRBThreads) reinterpret_cast<float4*>(cb)[i] = src[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kRBItems; ++i) {
        if (!ok[i]) continue;
        const float* rq = rs.rot_queries + (size_t)q[i] * rs.rot_dim;
        const float* ct = rs.centers_rot + (size_t)L[i] * rs.rot_dim;
        const uint4* cp = reinterpret_cast<const uint4*>(rs.codes) + ((size_t)(row[i] >> 6) * rs.n_chunks) * 64 + (row[i] & 63u);
        auto add_entry = [&](float v) {
          if constexpr (LUT == 2) v = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(v, rs.is_ip != 0);
          if constexpr (LUT == 0 || (LUT == 2 && !ACC_HALF)) {
            af[i] += v;
          } else {
            const _Float16 e = to_lut_half(v);
            if constexpr (ACC_HALF) ah[i] += e; else af[i] += (float)e;
          }
        };
#pragma unroll 1
        for (int cc = 0; cc < 4; ++cc) {
          const int c          = (int)blk * 4 + cc;
          const uint4 cw       = cp[c * 64];
          const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float qq[16], cv[16], p0[8], p1[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 qv = *reinterpret_cast<const float4*>(rq + c * 32 + hh * 16 + j * 4),
                           c4 = *reinterpret_cast<const float4*>(ct + c * 32 + hh * 16 + j * 4);
              qq[j * 4] = qv.x; qq[j * 4 + 1] = qv.y; qq[j * 4 + 2] = qv.z; qq[j * 4 + 3] = qv.w;
              cv[j * 4] = c4.x; cv[j * 4 + 1] = c4.y; cv[j * 4 + 2] = c4.z; cv[j * 4 + 3] = c4.w;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
              const int bb        = hh * 8 + b;
              const uint32_t code = (ws[bb >> 2] >> ((bb & 3) * 8)) & 0xffu;
              const uint32_t e0   = (uint32_t)((cc * 16 + bb) * 2) * rs.book + code;
              p0[b] = cb[e0]; p1[b] = cb[e0 + rs.book];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
              const float q0 = qq[b * 2], q1 = qq[b * 2 + 1], c0 = cv[b * 2], c1 = cv[b * 2 + 1];
              float v;
              if (!rs.is_ip) {
                const float d0 = (q0 - c0) - p0[b], d1 = (q1 - c1) - p1[b];
                v = __fmaf_rn(d1, d1, __fmaf_rn(d0, d0, 0.f));
              } else {
                v = __fmaf_rn(-q0, c0, 0.f);
                v = __fmaf_rn(-q0, p0[b], v);
                v = __fmaf_rn(-q1, c1, v);
                v = __fmaf_rn(-q1, p1[b], v);
              }
              add_entry(v);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kRBItems; ++i)
      if (ok[i]) atomicMax(&tmp[q[i]], float_to_key(ACC_HALF ? (float)ah[i] : af[i]));
  }
}

__global__ __launch_bounds__(256) void pqw_bound_finish_kernel(const wbound_params a, const uint32_t* __restrict__ tmp)
{
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (q >= a.nq) return;  // wave-uniform
  const bool no_bound = tmp[a.nq + q] != 0u || tmp[q] == 0u;  // (no candidate scored: k rows there are not)
  const uint32_t key  = no_bound ? 0xffffffffu : tmp[q];
  const float bound   = no_bound ? INFINITY : key_to_float(key);
  bool hand_back = no_bound;
  for (uint32_t rank = lane; rank < a.heads; rank += 64u) {
    const float4 nm   = a.norms[(size_t)q * a.heads + rank];
    const bool served = key < 0xff800000u && nm.w < 60000.f && fabsf(bound) <= a.bound_max;
    hand_back = hand_back || !served;
    const float t = a.is_ip ? filter_threshold_ip(bound, nm.x, nm.y, nm.z, a) : filter_threshold(bound, nm.x, a);
    a.thr_head[(size_t)q * a.heads + rank] = served ? t / a.c1 + (a.is_ip ? a.sc * a.sc * nm.z : -0.5f * a.sc * a.sc * nm.x) : INFINITY;
  }
  const bool any_back = __ballot(hand_back) != 0ull;
  if (lane == 0u) {
    a.query_kth[q] = key;
    if (any_back) a.qflag[q] = 1u;
  }
}

// the head pairs' own candidates: rows whose value reaches the pair's threshold -> the survivor regions, BEHIND the filter (the
// regions' fills are final); one wave per head pair (flat_head_survivors_kernel's scheme; a full buffer flags the query)
__global__ __launch_bounds__(256) void pqw_head_survivors_kernel(const float* __restrict__ xbuf, uint32_t ldx, const float* __restrict__ thr_head,
                                                                 uint32_t heads, const uint32_t* __restrict__ sorted_pairs,
                                                                 const uint32_t* __restrict__ pair_off, uint32_t n_lists,
                                                                 const uint32_t* __restrict__ probes, uint32_t n_probes,
                                                                 const uint32_t* __restrict__ list_offsets, const uint32_t* __restrict__ list_sizes,
                                                                 uint2* __restrict__ surv, uint32_t* __restrict__ surv_cnt, uint32_t n_regions,
                                                                 uint32_t surv_cap, uint32_t spill_cap, uint32_t* __restrict__ qflag)
{
  const uint32_t n_head = pair_off[n_lists];
  const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (i >= n_head) return;  // wave-uniform
  const uint32_t p = sorted_pairs[i], L = probes[p], q = p / n_probes;
  if (qflag[q] != 0u) return;  // (handed back: its survivors would not be re-scored)
  const uint32_t len = list_sizes[L], base_row = list_offsets[L];
  const size_t xrow = (size_t)q * heads + p % n_probes;
  const float th = thr_head[xrow];
  const float* x = xbuf + xrow * ldx;
  uint32_t total = 0u;
  for (uint32_t r0 = 0; r0 < len; r0 += 64u) {
    const uint32_t r = r0 + lane;
    total += (uint32_t)__popcll(__ballot(r < len && x[r] >= th));
  }
  if (total == 0u) return;
  const uint32_t ri = i % n_regions;
  uint32_t base = 0u;
  if (lane == 0u) base = atomicAdd(surv_cnt + ri, total);
  base = __builtin_amdgcn_readfirstlane(base);
  const uint32_t n_fit = base >= surv_cap ? 0u : min(total, surv_cap - base);  // what the region still holds; the rest: shared region
  uint32_t sbase = 0u;
  if (n_fit < total) {  // wave-uniform
    if (lane == 0u) { atomicMin(surv_cnt + ri, surv_cap); sbase = atomicAdd(surv_cnt + n_regions, total - n_fit); }
    sbase = __builtin_amdgcn_readfirstlane(sbase);
    if (sbase + (total - n_fit) > spill_cap && lane == 0u) qflag[q] = 1u;
  }
  uint2* region = surv + (size_t)ri * surv_cap;
  uint2* shared = surv + (size_t)n_regions * surv_cap;
  uint32_t done = 0u;
  for (uint32_t r0 = 0; r0 < len; r0 += 64u) {
    const uint32_t r = r0 + lane;
    const bool hit = r < len && x[r] >= th;
    const unsigned long long m = __ballot(hit);
    const uint32_t kth = done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (hit) {
      if (kth < n_fit) region[base + kth] = make_uint2(p, base_row + r);
      else if (sbase + (kth - n_fit) < spill_cap) shared[sbase + (kth - n_fit)] = make_uint2(p, base_row + r);
    }
    done += (uint32_t)__popcll(m);
  }
}

// ------------------------------------------------------------------ IVF-Flat through the same filter (fp32 rows, L2)
// The A operands of the GEMM are the rows' residuals against their list centre, rounded to fp16 after a power-of-two
// scaling: a derived copy of the index (half its size; 288 GB of HBM pay for it) laid out as the MFMA wants it.
// the VL = 16 / sizeof(T) elements of a 16-byte chunk as floats (fp32, fp16, int8 or uint8 rows; exact)
template <typename T>
__device__ inline void chunk_to_float(const uint4& w, float (&x)[16 / sizeof(T)])
{
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = __uint_as_float(ws[e]);
  } else if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)__builtin_bit_cast(_Float16, (uint16_t)(ws[e >> 1] >> ((e & 1) * 16)));
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (float)(T)(uint8_t)(ws[e >> 2] >> ((e & 3) * 8));
  }
}

// max |x - c| over the index (scaling) and |x - c|^2 per row (K-extension term); one thread per row
template <typename T>
__global__ void flat_residual_stats_kernel(const uint8_t* __restrict__ data, const float* __restrict__ centers,
                                           const uint32_t* __restrict__ row_list, int64_t rows, uint32_t dim, uint32_t n_chunks,
                                           float* __restrict__ dn, uint32_t* __restrict__ max_bits, float* __restrict__ inv_norm)
{
  const int64_t r0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r  = min(r0, rows - 1);
  const uint32_t L = row_list[r >> 6];  // list of the row's 64-row group (0xffffffff: padding beyond the last list)
  float acc = 0.f, mx = 0.f;
  if (L != 0xffffffffu) {
    const uint4* cp = reinterpret_cast<const uint4*>(data) + ((size_t)(r >> 6) * n_chunks) * 64 + (r & 63);
    const float* ct = centers + (size_t)L * dim;
    constexpr int VL = 16 / sizeof(T);
    float inv = 1.0f;
    if (inv_norm != nullptr) {  // cosine: residuals of the unit-length row
      float n2 = 0.f;
      for (uint32_t c = 0; c < n_chunks; ++c) {
        float x[VL];
        chunk_to_float<T>(cp[(size_t)c * 64], x);
#pragma unroll
        for (int e = 0; e < VL; ++e) n2 = __fmaf_rn(x[e], x[e], n2);
      }
      inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
    }
    for (uint32_t c = 0; c < n_chunks; ++c) {
      float x[VL];
      chunk_to_float<T>(cp[(size_t)c * 64], x);
#pragma unroll
      for (int e = 0; e < VL; ++e) {
        const float d = x[e] * inv - ct[c * VL + e];
        acc = __fmaf_rn(d, d, acc);
        mx  = fmaxf(mx, fabsf(d));
      }
    }
    if (inv_norm != nullptr && r0 < rows) inv_norm[r] = inv;
  } else if (inv_norm != nullptr && r0 < rows) {
    inv_norm[r] = 0.f;
  }
  if (r0 < rows) dn[r] = acc;
  const float wm = wave_reduce_max_f32(mx), wn = wave_reduce_max_f32(acc);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(max_bits, __float_as_uint(wm));
    atomicMax(max_bits + 1, __float_as_uint(wn));  // (inner product: the threshold's bound on |x - c|)
  }
}

// list id of every 64-row group (lists start at multiples of 64)
__global__ void flat_group_lists_kernel(const uint32_t* __restrict__ list_offsets, const uint32_t* __restrict__ list_sizes,
                                        uint32_t n_lists, uint32_t* __restrict__ row_list)
{
  const uint32_t L = blockIdx.x;
  if (L >= n_lists) return;
  const uint32_t g0 = list_offsets[L] >> 6, g1 = (list_offsets[L] + list_sizes[L] + 63u) >> 6;
  for (uint32_t g = g0 + threadIdx.x; g < g1; g += blockDim.x) row_list[g] = L;
}

// rows16[tile][step][lane]: lane (row r = lane & 31 of the tile, half h = lane >> 5) holds the residuals of dimensions
// [32 (step / 2) + 16 h + 8 (step % 2), + 8) - the K-slot order of the B operands - as scaled fp16; term[row]: the K-extension halves of -|x - c|^2 (1 - 2^-9) sc^2 / 2
template <typename T>
__global__ void flat_rows16_kernel(const uint8_t* __restrict__ data, const float* __restrict__ centers,
                                   const uint32_t* __restrict__ row_list, const float* __restrict__ dn, int64_t rows, uint32_t dim,
                                   uint32_t n_chunks, float sc, uint4* __restrict__ rows16, uint32_t* __restrict__ term,
                                   const float* __restrict__ inv_norm, int natural, float* __restrict__ term32)
{
  const uint32_t nst = dim / 16;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (tile, step, lane)
  if (t >= rows / 32 * nst * 64) return;
  const uint32_t lane = (uint32_t)(t & 63), st = (uint32_t)((t >> 6) % nst);
  const int64_t tile  = (t >> 6) / nst;
  const int64_t r     = tile * 32 + (lane & 31);
  // the K-slot order of the B operands: pq_bprep_kernel's (pq_len 2) - or, for the wide filter (pqw_bprep_kernel), the natural one
  const uint32_t h    = lane >> 5, d0 = natural ? 16u * st + 8u * h : 32u * (st >> 1) + 16u * h + 8u * (st & 1);
  const uint32_t L    = row_list[r >> 6];
  _Float16 v[8];
  if (L != 0xffffffffu) {
    const uint4* cp = reinterpret_cast<const uint4*>(data) + ((size_t)(r >> 6) * n_chunks) * 64 + (r & 63);
    const float* ct = centers + (size_t)L * dim + d0;
    const float inv = inv_norm != nullptr ? inv_norm[r] : 1.0f;
    constexpr int VL = 16 / sizeof(T);  // 8 dimensions = two chunks of fp32, one of fp16 or half a chunk of int8 / uint8
    if constexpr (VL <= 8) {
#pragma unroll
      for (int c = 0; c < 8 / VL; ++c) {
        float x[VL];
        chunk_to_float<T>(cp[(size_t)(d0 / VL + c) * 64], x);
#pragma unroll
        for (int e = 0; e < VL; ++e) v[c * VL + e] = (_Float16)(sc * (x[e] * inv - ct[c * VL + e]));
      }
    } else {
      float x[VL];
      chunk_to_float<T>(cp[(size_t)(d0 / VL) * 64], x);
      const bool upper = (d0 & 8u) != 0u;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (_Float16)(sc * ((upper ? x[8 + e] : x[e]) * inv - ct[e]));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
  }
  uint32_t w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    w[e] = (uint32_t)__builtin_bit_cast(uint16_t, v[2 * e]) | ((uint32_t)__builtin_bit_cast(uint16_t, v[2 * e + 1]) << 16);
  rows16[t] = make_uint4(w[0], w[1], w[2], w[3]);
  if (st == 0 && h == 0) {
    const float x     = -0.5f * sc * sc * (dn[r] * (1.0f - 1.0f / 512.0f));
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    term[r] = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
    if (term32 != nullptr) term32[r] = x;  // the wide filter's accumulators start from the term itself
  }
}

// one lane per survivor: the scan kernel's arithmetic (ivf_flat.hip: t = q - x, acc = fma(t, t, acc) - inner product: acc =
// fma(x, q, acc), score -acc - in dimension order;
// int8 / uint8 rows: the scan kernel's integer sum converted to float - every partial sum here is an integer below 2^24 for
// dim <= 256, so the fp32 chain gives that same number)
template <typename T>
__global__ __launch_bounds__(256) void flat_rescore_kernel(const rescore_params a)
{
  const bool spill = blockIdx.x + 1 == gridDim.x;
  const uint32_t ri = spill ? a.n_regions : blockIdx.x;
  const uint32_t n = spill ? min(a.surv_cnt[ri], a.spill_cap) : a.surv_cnt[ri];
  const uint2* region = a.surv + (size_t)ri * a.surv_cap;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t sb = blockIdx.y * blockDim.x + threadIdx.x - lane; sb < n; sb += gridDim.y * blockDim.x) {  // (wave-uniform trip count)
    const uint32_t s = sb + lane;
    bool ok = s < n;
    const uint2 sv = ok ? region[s] : make_uint2(0u, 0u);
    const uint32_t pair = sv.x, row = sv.y, q = pair / a.n_probes;
    if (ok && a.filter_bits != nullptr) {
      const int64_t sid = a.indices[row];
      ok = ((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u) != 0u;
    }
    float acc = 0.f, xn2 = 0.f, qn2 = 0.f;  // (cosine: |x|^2 and |q|^2 by the scan kernel's / the query tiles' chains)
    if (ok) {
      const float* rq = a.rot_queries + (size_t)q * a.dim;
      const uint4* cp = reinterpret_cast<const uint4*>(a.codes) + ((size_t)(row >> 6) * a.n_chunks) * 64 + (row & 63u);
      constexpr int VL = 16 / sizeof(T);
      for (uint32_t c = 0; c < a.n_chunks; ++c) {
        float x[VL];
        chunk_to_float<T>(cp[(size_t)c * 64], x);
#pragma unroll
        for (int e4 = 0; e4 < VL / 4; ++e4) {
          const float4 qv   = *reinterpret_cast<const float4*>(rq + c * VL + e4 * 4);
          const float qq[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (a.is_ip) {
              acc = __fmaf_rn(x[e4 * 4 + e], qq[e], acc);
              if (a.is_ip == 2) {
                xn2 = __fmaf_rn(x[e4 * 4 + e], x[e4 * 4 + e], xn2);
                qn2 = __fmaf_rn(qq[e], qq[e], qn2);
              }
            } else {
              const float t = qq[e] - x[e4 * 4 + e];
              acc = __fmaf_rn(t, t, acc);
            }
          }
        }
      }
    }
    if (a.is_ip == 2) acc = acc / (sqrtf(qn2) * sqrtf(xn2));  // cos, as ivf_flat_scan_kernel<.., 2> forms it
    pool_append_wave(a, ok, q, pair, row, a.is_ip ? -acc : acc);  // (inner product: smaller is better, as in the scan kernel)
  }
}

// ------------------------------------------------------------------ single-query list scan (head phase, handed-back pairs)
// One (query, list) pair per work item with no bound to prune against: the LUT scan kernel of ivf_pq_search.hip spends
// such an item on top-list bookkeeping (16 wave lists with serial insertions while the bounds are cold, then their
// merge: ~130 k cycles for 6 k rows). Here the workgroup builds the query's LUT, writes the score KEY of every row of
// the list to LDS and selects the k smallest by (score, row) with three histogram passes - select_k's scheme
// (select_k.hip) on data that never leaves the LDS. Scores are the reference's (same entries, same summation order).
struct head_params {
  const work_item* items;
  const uint32_t* item_begin;  // device scalars (item_begin == nullptr: from 0)
  const uint32_t* item_end;
  uint32_t n_lists;
  uint32_t* xcd_ticket;
  const uint32_t* sorted_pairs;
  const float* rot_queries;
  const float* centers_rot;
  const float* pq_centers;
  const uint8_t* codes;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  float* out_d;
  uint32_t* out_i;
  uint32_t* query_kth;
  uint32_t n_probes, rot_dim, k, cap_rows, pq_dim, n_chunks, pq_len, book;
  int is_ip, per_cluster;
  int hcand;  // capacity of a candidate buffer (head_cand(k))
  uint32_t row_limit; // > 0: only the first row_limit rows of a list are scored (partial head)
  uint32_t one_shot;  // > 0: that many items, ONE per workgroup (item blockIdx.x), no tickets - the two-stream schedule's head launch:
                      // workgroup slots free up item by item, so the helper stream's kernels get their share of the CUs
  const uint32_t* filter_bits;
  const int64_t* indices;
  unsigned long long* stats;  // optional [8]: workgroup cycles in header / LUT / scores / select / output, items
};

// candidates of a list chunk at or below its threshold (about k of them) + the kept ones (up to k)
static inline int head_cand(int k) { return k <= 128 ? 512 : 1024; }

template <int LUT, bool ACC_HALF, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void pq_head_kernel(const head_params a)
{
  constexpr bool LUT32 = LUT == 0 || (LUT == 2 && !ACC_HALF);
  using lut_t = std::conditional_t<LUT32, float, _Float16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lut_t* lut      = reinterpret_cast<lut_t*>(smem);                                      // [pq_dim][256]
  float* rv       = reinterpret_cast<float*>(smem + (size_t)a.pq_dim * 256 * sizeof(lut_t));  // [256] residual (L2) / query (IP)
  float* cv       = rv + 256;                                                            // [256] list centre
  unsigned long long* min64 = reinterpret_cast<unsigned long long*>(cv + 256);           // [2]
  int* ctrl       = reinterpret_cast<int*>(min64 + 2);                           // [8]
  // (no static LDS in this kernel: the LUT sits at LDS address 0 and a lookup's address is (code byte) << 1 - one SDWA shift)
  work_item& cur  = *reinterpret_cast<work_item*>(ctrl + 8);                     // the current item, 16 bytes
  uint32_t* tk    = reinterpret_cast<uint32_t*>(ctrl + 12);                      // [NT] smallest key of every thread
  const int kHCand = a.hcand;
  uint2* cand     = reinterpret_cast<uint2*>(tk + NT);                           // [2][kHCand] candidates (key, row), two buffers: 16-byte aligned -
                                                                                 // the rank pass reads two entries per ds_read_b128
  uint32_t* keys  = reinterpret_cast<uint32_t*>(cand + 2 * kHCand);              // [cap_rows] score keys of the current chunk
  static_assert(sizeof(work_item) == 16, "work_item");

  const int tid = threadIdx.x;
  const uint32_t item0   = (a.one_shot == 0u && a.item_begin) ? *a.item_begin : 0u;
  const uint32_t n_items = a.one_shot != 0u ? a.one_shot : *a.item_end - item0;
  const uint32_t xcd = blockIdx.x & 7u, chunk = (n_items + 7u) / 8u;
  const uint32_t share0 = min(n_items, xcd * chunk), share_len = min(chunk, n_items - share0);
  const work_item* share = a.items + item0 + share0;
  const uint4* codes16   = reinterpret_cast<const uint4*>(a.codes);

  for (uint32_t round = 0;; ++round) {
    __syncthreads();
    if (tid == 0) {
      if (a.one_shot != 0u) {
        cur = (round == 0u && blockIdx.x < a.one_shot) ? a.items[blockIdx.x] : work_item{0u, 0u, 0u, 0xffffffffu};
      } else {
        const uint32_t t = atomicAdd(a.xcd_ticket + xcd * 32, 1u);
        cur = t < share_len ? share[t] : work_item{0u, 0u, 0u, 0xffffffffu};
      }
    }
    __syncthreads();
    const work_item item = cur;
    if (item.pad == 0xffffffffu) break;  // workgroup-uniform
    if (item.count == 0u) continue;      // (direct head items: the nearest list of this query lives on another rank)
    unsigned long long t_prev = a.stats != nullptr ? __builtin_readcyclecounter() : 0ull;
    auto phase = [&](int which) {
      if (a.stats != nullptr && tid == 0) {
        const unsigned long long t = __builtin_readcyclecounter();
        atomicAdd(&a.stats[which], t - t_prev);
        t_prev = t;
      }
    };
    const uint32_t L        = item.list >= a.n_lists ? item.list - a.n_lists : item.list;
    const uint32_t base_row = a.list_offsets[L], len = a.row_limit != 0u ? min(a.list_sizes[L], a.row_limit) : a.list_sizes[L];
    const uint32_t pair     = a.sorted_pairs[item.first];
    const uint32_t q        = pair / a.n_probes;
    if (tid < (int)a.rot_dim) {
      const float c = a.centers_rot[(size_t)L * a.rot_dim + tid];
      float v       = a.rot_queries[(size_t)q * a.rot_dim + tid];
      if (!a.is_ip) v -= c;
      rv[tid] = v;
      cv[tid] = c;
    }
    __syncthreads();
    phase(0);
    // ---- LUT (create_lut_impl.cuh:17-78), entry (s, code) at s * 256 + code; the codebook values of 16 entries are
    // loaded before any is used (one L2 round trip per batch instead of one per entry; round 6: 16 instead of 8 - two round trips
    // per item at pq_dim 64 instead of four)
    constexpr int LB = 16;
#pragma unroll 1
    for (uint32_t e0 = tid; e0 < ((a.pq_len == 2u && !a.per_cluster) ? a.pq_dim * 256u : 0u); e0 += (uint32_t)LB * NT) {
      float p0[LB], p1[LB];
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        // (codes of fewer than 8 bits: the LUT keeps 256 slots per subspace, those past the codebook are never looked up)
        const uint32_t e = min(e0 + (uint32_t)j * NT, a.pq_dim * 256u - 1u), sb = e >> 8, code = min(e & 255u, a.book - 1u);
        p0[j] = a.pq_centers[(size_t)(sb * 2 + 0) * a.book + code];
        p1[j] = a.pq_centers[(size_t)(sb * 2 + 1) * a.book + code];
      }
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        const uint32_t e = e0 + (uint32_t)j * NT, sb = e >> 8;
        if (e >= a.pq_dim * 256u) break;
        const float q0 = rv[sb * 2], q1 = rv[sb * 2 + 1];
        float v;
        if (!a.is_ip) {
          const float d0 = q0 - p0[j], d1 = q1 - p1[j];
          v = __fmaf_rn(d1, d1, __fmaf_rn(d0, d0, 0.f));
        } else {
          v = __fmaf_rn(-q0, cv[sb * 2], 0.f);
          v = __fmaf_rn(-q0, p0[j], v);
          v = __fmaf_rn(-q1, cv[sb * 2 + 1], v);
          v = __fmaf_rn(-q1, p1[j], v);
        }
        if constexpr (LUT == 2) v = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(v, a.is_ip != 0);
        if constexpr (LUT32) lut[e] = v; else lut[e] = to_lut_half(v);
      }
    }
    if (a.pq_len != 2u || a.per_cluster) {  // any pq_len / PER_CLUSTER codebooks: the components' chain in order, four entries in flight
#pragma unroll 1
      for (uint32_t e0 = tid; e0 < a.pq_dim * 256u; e0 += 4u * NT) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t l = 0; l < a.pq_len; ++l) {
          float p[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t e = min(e0 + (uint32_t)j * NT, a.pq_dim * 256u - 1u), sb = e >> 8, code = min(e & 255u, a.book - 1u);
            p[j] = a.pq_centers[(size_t)((a.per_cluster ? L : sb) * a.pq_len + l) * a.book + code];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t e = min(e0 + (uint32_t)j * NT, a.pq_dim * 256u - 1u), dd = (e >> 8) * a.pq_len + l;
            const float qv = rv[dd];
            if (!a.is_ip) {
              const float d = qv - p[j];
              v[j] = __fmaf_rn(d, d, v[j]);
            } else {
              v[j] = __fmaf_rn(-qv, cv[dd], v[j]);
              v[j] = __fmaf_rn(-qv, p[j], v[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t e = e0 + (uint32_t)j * NT;
          if (e >= a.pq_dim * 256u) break;
          float x = v[j];
          if constexpr (LUT == 2) x = fp8_round_trip<std::conditional_t<ACC_HALF, __half, float>>(x, a.is_ip != 0);
          if constexpr (LUT32) lut[e] = x; else lut[e] = to_lut_half(x);
        }
      }
    }
    if (tid == 0) ctrl[0] = 0;  // candidates kept so far (buffer 0)
    __syncthreads();
    phase(1);
    const int k = (int)a.k;
    int buf = 0;  // candidate buffer holding the kept candidates
    // ---- the list in chunks of cap_rows rows (one chunk for all but very long lists)
    for (uint32_t c0 = 0; c0 < len; c0 += a.cap_rows) {
      const uint32_t clen = min(a.cap_rows, len - c0);
      // score keys of the chunk's rows (filtered rows: invalid)
      uint32_t my_min = 0xffffffffu;
      for (uint32_t v = tid; v < clen; v += NT) {
        const uint32_t fr = base_row + c0 + v;
        const uint4* cp   = codes16 + ((size_t)(fr >> 6) * a.n_chunks) * 64 + (fr & 63u);
        float af    = 0.f;
        _Float16 ah = (_Float16)0.f;
        for (int c0c = 0; c0c < (int)a.n_chunks; c0c += 4) {  // four chunk loads in flight
          uint4 cw[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) cw[c] = cp[(size_t)min(c0c + c, (int)a.n_chunks - 1) * 64];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int c = c0c + cc;
            if (c >= (int)a.n_chunks) break;
            const uint32_t ws[4] = {cw[cc].x, cw[cc].y, cw[cc].z, cw[cc].w};
            // the chunk's 16 lookups are issued together, the sum follows in order (left to the compiler every lookup was
            // followed by a wait for it: one LDS latency per entry and wave)
            lut_t e[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) e[b] = lut[((c * 16 + b) << 8) + ((ws[b >> 2] >> ((b & 3) * 8)) & 0xffu)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 16; ++b) {
              if constexpr (LUT32) af += e[b];
              else if constexpr (ACC_HALF) ah += e[b];
              else af += (float)e[b];
            }
          }
        }
        const float score = ACC_HALF ? (float)ah : af;
        bool keep = true;
        if (a.filter_bits != nullptr) {
          const int64_t sid = a.indices[fr];
          keep = ((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u) != 0u;
        }
        const uint32_t key = keep ? float_to_key(score) : 0xffffffffu;  // (a NaN score has a key beyond every finite one as well)
        keys[v] = key;
        my_min  = min(my_min, key);
      }
      // minima of groups of four threads (k <= 64), of two (k <= 128) or of one (k <= 256: at least 2 k groups keep the
      // threshold tight) by DPP: each group is a set of rows of its own
      const int gshift = k > 128 ? 0 : (k > 64 ? 1 : 2);
      if (gshift >= 1)
        my_min = min(my_min, (uint32_t)__builtin_amdgcn_update_dpp((int)my_min, (int)my_min, 0xb1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
      if (gshift == 2)
        my_min = min(my_min, (uint32_t)__builtin_amdgcn_update_dpp((int)my_min, (int)my_min, 0x4e, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
      if ((tid & ((1 << gshift) - 1)) == 0) tk[tid >> gshift] = my_min;
      __syncthreads();
      phase(2);
      // ---- candidates of the chunk. The scores of a list share their leading bits: a histogram select serializes on a
      // few LDS counters (33 k cycles measured), a bitonic sort is a chain of barriers. Instead: the k-th smallest of the
      // group minima bounds the chunk's k-th smallest key from above (k groups hold k different rows at or below it),
      // found by counting ranks over broadcast reads; the keys at or below it - about k - join the candidates.
      const int NGr   = NT >> gshift;
      const int k_c   = (int)min((uint32_t)k, clen);
      const int kept  = ctrl[0];
      if (tid < NGr) {
        const uint32_t mine = tk[tid];
        const uint4* tk4    = reinterpret_cast<const uint4*>(tk);
        int r = 0;
#pragma unroll 4
        for (int j = 0; j < NGr / 4; ++j) {  // 16-byte broadcast reads, four in flight
          const uint4 o = tk4[j];
          r += (o.x < mine || (o.x == mine && 4 * j + 0 < tid)) ? 1 : 0;
          r += (o.y < mine || (o.y == mine && 4 * j + 1 < tid)) ? 1 : 0;
          r += (o.z < mine || (o.z == mine && 4 * j + 2 < tid)) ? 1 : 0;
          r += (o.w < mine || (o.w == mine && 4 * j + 3 < tid)) ? 1 : 0;
        }
        if (r == k_c - 1) ctrl[1] = (int)mine;
        if (tid == 0) ctrl[2] = kept;  // append position
      }
      __syncthreads();
      const uint32_t tau = (uint32_t)ctrl[1];
      uint2* bc = cand + buf * kHCand;
      for (uint32_t i = tid; i < clen; i += NT) {
        const uint32_t key = keys[i];
        if (key <= tau && key != 0xffffffffu) {
          const int pos = atomicAdd(&ctrl[2], 1);
          if (pos < kHCand) bc[pos] = make_uint2(key, c0 + i);
        }
      }
      __syncthreads();
      int cnt = ctrl[2];
      if (cnt > kHCand) {
        // masses of equal scores at the threshold: the chunk's k smallest (key, row) one after the other instead
        unsigned long long last = 0ull;
        bool first = true;
        cnt = kept;
        for (int r = 0; r < k_c; ++r) {
          if (tid == 0) min64[0] = ~0ull;
          __syncthreads();
          unsigned long long best = ~0ull;
          for (uint32_t i = tid; i < clen; i += NT) {
            const unsigned long long v = ((unsigned long long)keys[i] << 32) | (unsigned long long)(c0 + i);
            if ((first || v > last) && keys[i] != 0xffffffffu && v < best) best = v;
          }
          if (best != ~0ull) atomicMin(&min64[0], best);
          __syncthreads();
          last  = min64[0];
          first = false;
          if (last == ~0ull) break;  // workgroup-uniform
          if (tid == 0) bc[cnt] = make_uint2((uint32_t)(last >> 32), (uint32_t)last);
          ++cnt;
          __syncthreads();
        }
      }
      // ---- keep the k best of the candidates, in (key, row) order, in the other buffer
      // (rank counting over broadcast reads; (key, row) pairs side by side: one 16-byte read serves two comparisons - the pass is a
      // chain of LDS round trips behind the OTHER workgroup's gathers, 41 k of an item's 127 k cycles at k = 60)
      uint2* nc = cand + (buf ^ 1) * kHCand;
      const uint4* b4 = reinterpret_cast<const uint4*>(bc);
      for (int t = tid; t < cnt; t += NT) {
        const uint2 me = bc[t];
        int r = 0;
#pragma unroll 4
        for (int j = 0; j < (cnt + 1) / 2; ++j) {
          const uint4 o = b4[j];
          r += (o.x < me.x || (o.x == me.x && o.y < me.y)) ? 1 : 0;
          r += (2 * j + 1 < cnt && (o.z < me.x || (o.z == me.x && o.w < me.y))) ? 1 : 0;
        }
        if (r < k) nc[r] = me;
      }
      if (tid == 0) ctrl[0] = min(cnt, k);
      buf ^= 1;
      __syncthreads();
      phase(3);
    }
    // ---- the pair's candidate row, ordered by (score, row) (the caller pre-filled it "invalid")
    {
      const int cnt = ctrl[0];
      const uint2* bc = cand + buf * kHCand;
      if (tid < cnt) {
        const uint32_t key = bc[tid].x;
        const size_t o     = (size_t)pair * a.k + tid;
        a.out_d[o]         = key_to_float(key);
        a.out_i[o]         = base_row + bc[tid].y;
        if (tid == k - 1 && key < 0xff800000u) atomicMin(&a.query_kth[q], key);
      }
    }
    phase(4);
    if (a.stats != nullptr && tid == 0) atomicAdd(&a.stats[5], 1ull);
  }
}

// ------------------------------------------------------------------ flagged queries: back to the LUT scan, pair by pair
__global__ void reset_flagged_kernel(const uint32_t* __restrict__ qflag, int64_t nq, uint32_t n_probes, uint32_t k, uint32_t head,
                                     float* __restrict__ cand_d, uint32_t* __restrict__ cand_i)
{
  const int64_t q = blockIdx.x;
  if (q >= nq || qflag[q] == 0u) return;
  const size_t o = (size_t)q * n_probes * k;
  for (uint32_t s = head * k + threadIdx.x; s < n_probes * k; s += blockDim.x) {
    cand_d[o + s] = FLT_MAX;
    cand_i[o + s] = 0xffffffffu;
  }
}

__global__ void fallback_items_kernel(const uint32_t* __restrict__ sorted_pairs, const uint32_t* __restrict__ pair_off,
                                      uint32_t n_lists, const uint32_t* __restrict__ probes, uint32_t n_probes,
                                      const uint32_t* __restrict__ qflag, work_item* __restrict__ items,
                                      uint32_t* __restrict__ n_items, int from_head)
{
  // from_head: the head pairs too (the wide path's bound-only head phase left no candidates of theirs)
  const uint32_t h_end = pair_off[n_lists];
  const uint32_t b = from_head ? pair_off[0] : h_end, e = pair_off[2 * n_lists];
  for (uint32_t s = b + blockIdx.x * blockDim.x + threadIdx.x; s < e; s += gridDim.x * blockDim.x) {
    const uint32_t p = sorted_pairs[s];
    if (qflag[p / n_probes] == 0u) continue;
    const uint32_t w = atomicAdd(n_items, 1u);
    items[w] = work_item{(s < h_end ? 0u : n_lists) + probes[p], s, 1u, 0u};
  }
}

// ------------------------------------------------------------------ overflow list binned by query (count, scan, fill)
// (the overflow entries of a query arrive in bursts - its survivors sit in the same survivor chunks - so a wave's 64 entries
// belong to a handful of queries: one atomic per distinct query of a wave instead of one per entry on a few hot counters)
__global__ void ov_count_kernel(const uint4* __restrict__ ov, const uint32_t* __restrict__ n_ov, uint32_t cap, uint32_t* __restrict__ cnt)
{
  const uint32_t n = min(*n_ov, cap);
  for (uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; i0 < n; i0 += gridDim.x * blockDim.x) {
    const uint32_t i = i0 + (threadIdx.x & 63u);
    const uint32_t q = i < n ? ov[i].x : 0xffffffffu;
    unsigned long long todo = __ballot(q != 0xffffffffu);
    while (todo != 0ull) {
      const uint32_t q0 = __builtin_amdgcn_readlane(q, (int)__ffsll((long long)todo) - 1);
      const unsigned long long same = __ballot(q == q0);
      if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)same) - 1u) atomicAdd(&cnt[q0], (uint32_t)__popcll(same));
      todo &= ~same;
    }
  }
}

__global__ __launch_bounds__(1024) void ov_scan_kernel(uint32_t* __restrict__ cnt, int64_t nq, uint32_t* __restrict__ off)
{
  __shared__ int smem[17];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nq; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = i < nq ? (int)cnt[i] : 0;
    int total;
    const int excl = block_exclusive_scan(v, smem, &total);
    if (i < nq) { off[i] = (uint32_t)(carry + excl); cnt[i] = 0u; }  // cnt becomes the fill cursor
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) off[nq] = (uint32_t)carry;
}

__global__ void ov_fill_kernel(const uint4* __restrict__ ov, const uint32_t* __restrict__ n_ov, uint32_t cap,
                               const uint32_t* __restrict__ off, uint32_t* __restrict__ cursor, uint4* __restrict__ sorted)
{
  const uint32_t n = min(*n_ov, cap);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; i0 < n; i0 += gridDim.x * blockDim.x) {
    const uint32_t i = i0 + lane;
    uint4 e = make_uint4(0xffffffffu, 0u, 0u, 0u);
    if (i < n) e = ov[i];
    unsigned long long todo = __ballot(e.x != 0xffffffffu);
    while (todo != 0ull) {
      const uint32_t q0 = __builtin_amdgcn_readlane(e.x, (int)__ffsll((long long)todo) - 1);
      const unsigned long long same = __ballot(e.x == q0);
      const uint32_t leader = (uint32_t)__ffsll((long long)same) - 1u;
      uint32_t base = 0u;
      if (lane == leader) base = atomicAdd(&cursor[q0], (uint32_t)__popcll(same));
      base = __builtin_amdgcn_readlane(base, (int)leader);
      if (e.x == q0) sorted[off[q0] + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = e;
      todo &= ~same;
    }
  }
}

// ------------------------------------------------------------------ merge: one wave per query
template <int E>
struct top3 {  // sorted ascending by (d, rank, row); rank r lives in lane r % 64, slot r / 64
  float d[E];
  uint32_t rk[E], row[E];
  __device__ inline void init()
  {
#pragma unroll
    for (int e = 0; e < E; ++e) { d[e] = INFINITY; rk[e] = 0xffffffffu; row[e] = 0xffffffffu; }
  }
  __device__ static inline bool before(float da, uint32_t ra, uint32_t wa, float db, uint32_t rb, uint32_t wb)
  {
    return da < db || (da == db && (ra < rb || (ra == rb && wa < wb)));
  }
  // the entry of (wave-uniform) rank r
  __device__ inline void at(int r, float& od, uint32_t& ork, uint32_t& orow) const
  {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if ((r >> 6) == e) {
        od   = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d[e]), r & 63));
        ork  = __builtin_amdgcn_readlane(rk[e], r & 63);
        orow = __builtin_amdgcn_readlane(row[e], r & 63);
      }
  }
  __device__ inline void insert(float cd, uint32_t cr, uint32_t cw, int lane)
  {
    int pos = 0;  // entries at or before the candidate
#pragma unroll
    for (int e = 0; e < E; ++e) pos += __popcll(__ballot(!before(cd, cr, cw, d[e], rk[e], row[e])));
    uint32_t c_d = 0, c_r = 0, c_w = 0;  // the entry that moves from lane 63 of a slot to lane 0 of the next
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t du = __float_as_uint(d[e]);
      const uint32_t ud = (uint32_t)__builtin_amdgcn_update_dpp((int)c_d, (int)du, 0x138, 0xf, 0xf, false);  // wave_shr:1
      const uint32_t ur = (uint32_t)__builtin_amdgcn_update_dpp((int)c_r, (int)rk[e], 0x138, 0xf, 0xf, false);
      const uint32_t uw = (uint32_t)__builtin_amdgcn_update_dpp((int)c_w, (int)row[e], 0x138, 0xf, 0xf, false);
      c_d = __builtin_amdgcn_readlane(du, 63);
      c_r = __builtin_amdgcn_readlane(rk[e], 63);
      c_w = __builtin_amdgcn_readlane(row[e], 63);
      const int rank = e * 64 + lane;
      if (rank > pos) { d[e] = __uint_as_float(ud); rk[e] = ur; row[e] = uw; }
      else if (rank == pos) { d[e] = cd; rk[e] = cr; row[e] = cw; }
    }
  }
};

template <int E>  // k <= 64 E
__global__ __launch_bounds__(256) void pool_merge_kernel(const float* __restrict__ cand_d, const uint32_t* __restrict__ cand_i,
                                                         const uint32_t* __restrict__ cand_r, const uint32_t* __restrict__ qcnt,
                                                         const uint32_t* __restrict__ qflag, int64_t nq, uint32_t n_probes,
                                                         uint32_t k, uint32_t head, float* __restrict__ top_d,
                                                         uint32_t* __restrict__ top_i, const uint4* __restrict__ overflow,
                                                         const uint32_t* __restrict__ ov_off)
{
  const int lane  = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const size_t o      = (size_t)q * n_probes * k;
  const bool flagged  = qflag[q] != 0u;
  const uint32_t pool = min(qcnt[q], (n_probes - head) * k);
  const uint32_t n    = flagged ? n_probes * k : head * k + pool;
  const int kr        = (int)k - 1;
  top3<E> best;
  best.init();
  float kd = INFINITY;
  uint32_t krk = 0xffffffffu, krow = 0xffffffffu;
  // a query whose pool ran over also has candidates in the overflow list (binned by query)
  const uint32_t ov0  = ov_off[q];
  const uint32_t n_ov = flagged ? 0u : ov_off[q + 1] - ov0;
  for (uint32_t s0 = 0; s0 < n + n_ov; s0 += 64) {
    const uint32_t s = s0 + lane;
    float d = INFINITY;
    uint32_t rk = 0xffffffffu, row = 0xffffffffu;
    if (s < n) {
      d   = cand_d[o + s];
      row = cand_i[o + s];
      rk  = (flagged || s < head * k) ? s / k : cand_r[o + s];
    } else if (s < n + n_ov) {
      const uint4 e = overflow[ov0 + s - n];
      d = __uint_as_float(e.y); rk = e.z; row = e.w;
    }
    unsigned long long m = __ballot(row != 0xffffffffu && top3<E>::before(d, rk, row, kd, krk, krow));
    while (m != 0ull) {
      const int src = (int)__ffsll((long long)m) - 1;
      m &= m - 1ull;
      const float cd    = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), src));
      const uint32_t cr = __builtin_amdgcn_readlane(rk, src), cw = __builtin_amdgcn_readlane(row, src);
      if (top3<E>::before(cd, cr, cw, kd, krk, krow)) {
        best.insert(cd, cr, cw, lane);
        best.at(kr, kd, krk, krow);
      }
    }
  }
  // the winners, ordered by (score, row) as select_k orders them
  wave_top<E> out;
  out.init();
  for (int r = 0; r < (int)k; ++r) {
    float cd = 0.f;
    uint32_t cr = 0u, cw = 0xffffffffu;
    best.at(r, cd, cr, cw);
    if (cw == 0xffffffffu) break;
    out.insert(cd, cw, lane);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = e * 64 + lane;
    if (r < (int)k) {
      const bool ok = out.i[e] != 0xffffffffu;
      top_d[q * k + r] = ok ? out.d[e] : FLT_MAX;
      top_i[q * k + r] = ok ? out.i[e] : 0xffffffffu;
    }
  }
}

}  // namespace

// ------------------------------------------------------------------ host side
unsigned pq3_grid(const resources& res) { return (unsigned)std::max(8, res.num_cus / 8 * 8); }
unsigned pq3_regions(const resources& res) { return 2 * pq3_grid(res); }  // room for the survivor-region counters: pq_filter_kernel one region per
                                                                           // workgroup, flat_filter2_kernel two workgroups per CU (pq_filter4_kernel: chunks)

bool pq3_supported(const ivf_pq_index& idx, int k)
{
  // codes of 4 .. 8 bits (fewer than 8: through a one-byte-per-code copy, pq3_codes), pq_dim a multiple of 16; pq_len 1 / 2 / 4 / 8 (a lane's 8 K elements of an MFMA step = 8 / pq_len codebook
  // entries); at most 16 K steps (the B operands of a unit's queries live in registers) = a decode table of at most
  // 128 KiB; pq_dim <= 128 (the head phase's LUT in LDS); PER_SUBSPACE or PER_CLUSTER codebooks. Round 3's pq_filter_kernel
  // (comparator, inner product): pq_len 2, PER_SUBSPACE
  return idx.pq_bits >= 4 && idx.pq_bits <= 8 && (idx.pq_len == 1 || idx.pq_len == 2 || idx.pq_len == 4 || idx.pq_len == 8) && idx.pq_dim % 16 == 0 &&
         idx.pq_dim >= 16 && idx.pq_dim <= 128 && idx.rot_dim == idx.pq_len * idx.pq_dim && idx.rot_dim <= 256 &&
         k <= 256;
}

// The tail phase prunes with the k-th score of the query's NEAREST list: useful while k is a small fraction of a list
// (k = 10 of 6.1 k rows at the bench shape; k = 100: still 4x faster than the LUT scan), useless when it is not - at
// k = 256 of 2.4 k-row lists (the IVF-PQ searches of a CAGRA build) a tenth of all pairs survived the screen and the
// re-score took 2.8x the LUT scan's time. Measured crossover: k around 4 % of the mean list length.
bool pq3_bound_useful(const ivf_pq_index& idx, int k)
{
  // A list shard whose searches hold collectives (communicator attached): every rank must take the same decision - for
  // inner product / cosine it decides whether the search has a head phase at all, and the head phase ends in an
  // all-reduce of the bounds - while the list sizes differ from rank to rank. The ranks therefore decide on the rows and
  // non-empty lists of ALL shards (exchanged once, ivf_pq_search.hip: shard_exchange_stats): the same number everywhere,
  // and the measured crossover of the unsharded index. Without a communicator a search has no collective and a shard
  // decides on its own lists (foreign lists are empty and do not count).
  uint64_t rows = 0, lists = 0;
  if (idx.shard_world > 1 && idx.shard_comm != nullptr) {
    if (!idx.shard_stats_valid) return k <= 128;  // (not reached through cuvsIvfPqSearch: the exchange precedes the decision)
    rows = idx.shard_global_rows; lists = idx.shard_global_lists;
  } else {
    for (uint32_t v : idx.h_list_sizes) { rows += v; lists += v != 0u; }
  }
  return lists != 0 && (uint64_t)k * 25u * lists <= rows;
}

static std::recursive_mutex g_pq3_mu;

// the codes as the kernels of this file read them - one byte per code, 16 per chunk - and the chunks per row: the index's
// own array for 8-bit codes, a derived copy (built on first use, rebuilt when the lists change) for 4 .. 7 bits
const uint8_t* pq3_codes(resources& res, const ivf_pq_index& idx, uint32_t* n_chunks)
{
  *n_chunks = idx.pq_dim / 16;
  if (idx.pq_bits == 8) return idx.codes.data();
  std::lock_guard<std::recursive_mutex> lock(g_pq3_mu);
  auto& c = idx.scan3;
  if (c.codes8_src != idx.codes.data() || c.codes8_rows != idx.padded_rows || c.codes8_size != idx.size) {
    const int64_t rows = std::max<int64_t>(idx.padded_rows, 1);
    c.codes8 = dev_buf<uint8_t>::persistent((size_t)rows * idx.pq_dim);
    if (idx.padded_rows > 0)
      hipLaunchKernelGGL(expand_codes_kernel, dim3(grid_blocks(idx.padded_rows * (int64_t)(idx.pq_dim / 16), 256)), dim3(256), 0, res.stream,
                         idx.codes.data(), idx.padded_rows, idx.n_chunks, idx.codes_per_chunk, idx.pq_bits, idx.pq_dim, c.codes8.data());
    c.codes8_src = idx.codes.data(); c.codes8_rows = idx.padded_rows; c.codes8_size = idx.size;
  }
  return c.codes8.data();
}

pq3_tables pq3_prepare(resources& res, const ivf_pq_index& idx, const bool term_fp32)
{
  std::lock_guard<std::recursive_mutex> lock(g_pq3_mu);
  auto& c = idx.scan3;
  if (c.codes_ptr != idx.codes.data() || c.rows != idx.padded_rows || c.size != idx.size || c.pq_ptr != idx.pq_centers.data() ||
      c.term_fp32 != term_fp32) {
    std::vector<float> h = to_host(res, idx.pq_centers.data(), idx.pq_centers.size());
    float mx = 0.f;
    for (float v : h) mx = std::max(mx, std::fabs(v));
    // both GEMM operands are scaled by a power of two so that the largest codebook value lands in (8, 16]: query
    // residuals up to 4096 times larger still fit the fp16 range, values 2^17 times smaller are still normal numbers
    c.sc    = mx > 0.f ? std::exp2(std::floor(std::log2(16.0f / mx))) : 1.0f;
    c.cbmax = mx;
    const uint32_t n_tables = idx.codebook_kind == 0 ? idx.pq_dim : idx.n_lists;  // PER_CLUSTER: one decode table per list
    c.cb16  = dev_buf<uint32_t>::persistent((size_t)n_tables * 256 * idx.pq_len / 2 + 1);
    hipLaunchKernelGGL(cb16_kernel, dim3(grid_blocks((int64_t)n_tables * 256, 256)), dim3(256), 0, res.stream,
                       idx.pq_centers.data(), n_tables, idx.pq_len, idx.pq_book, c.sc, reinterpret_cast<uint16_t*>(c.cb16.data()));
    c.row_term = dev_buf<uint32_t>::persistent((size_t)std::max<int64_t>(idx.padded_rows, 1));
    dev_buf<uint32_t> mxd(res, 1);
    HIP_TRY(hipMemsetAsync(mxd.data(), 0, sizeof(uint32_t), res.stream));
    uint32_t nch8 = 0;
    const uint8_t* codes8 = pq3_codes(res, idx, &nch8);
    if (idx.padded_rows > 0)
      hipLaunchKernelGGL(row_term_kernel, dim3(grid_blocks(idx.padded_rows, 256)), dim3(256), 0, res.stream, codes8,
                         idx.pq_centers.data(), idx.padded_rows, c.sc, c.row_term.data(), mxd.data(), (int)nch8, term_fp32 ? 1 : 0,
                         (int)idx.pq_len, idx.pq_book, idx.codebook_kind == 0 ? nullptr : idx.list_offsets.data(), idx.n_lists);
    const uint32_t mbits = to_host(res, mxd.data(), 1)[0];
    float dn_max;
    memcpy(&dn_max, &mbits, 4);
    c.dmax = std::sqrt(dn_max) * 1.0001f;
    c.codes_ptr = idx.codes.data(); c.rows = idx.padded_rows; c.size = idx.size; c.pq_ptr = idx.pq_centers.data();
    c.term_fp32 = term_fp32;
  }
  return pq3_tables{c.cb16.data(), c.row_term.data(), c.sc, c.cbmax, c.dmax};
}

size_t pq3_max_units(const ivf_pq_index& idx, int64_t n_pairs, uint32_t* unit_rows, bool filter4)
{
  uint32_t max_len = 0;
  for (uint32_t v : idx.h_list_sizes) max_len = std::max(max_len, v);
  // a unit is a chunk of 4096 rows of a list for one group of queries (the B operands and thresholds are built once per
  // unit: ~23 k cycles against ~250 k for the rows); a list is cut into at most 16 row chunks
  // (pq_filter4_kernel: the prologue is one memory round trip; chunks of up to 8192 rows, lists cut into equal parts)
  const uint32_t ur = std::max<uint32_t>(filter4 ? 8192u : 4096u, (uint32_t)round_up((int64_t)(max_len + 15) / 16, 64));
  *unit_rows = ur;
  return (size_t)16 * ((size_t)n_pairs / 32 + idx.n_lists + 1);
}

void pq3_warm(resources& res, const ivf_pq_index& idx, bool filter4)
{
  uint32_t n_chunks = 0;
  (void)pq3_codes(res, idx, &n_chunks);
  (void)pq3_prepare(res, idx, filter4);
}

void pq3_tail(resources& res, const ivf_pq_index& idx, const pq3_run& r)
{
  const bool f4 = r.bq != nullptr;  // pq_filter4_kernel (one wave per SIMD, up to four query groups per unit)
  CUVS_EXPECTS(r.stage == 0 || f4, "ivf_pq: the two-stream schedule is pq_filter4_kernel's");
  const pq3_tables tb = pq3_prepare(res, idx, f4);
  if (r.stage != 1) profile_begin(res, "pq_scan_kernel");  // bench.py sums the scan phases under this name
  const int nch        = (int)idx.pq_dim / 16;          // 16-byte code chunks per row
  const int nst        = nch * (int)idx.pq_len;         // MFMA K steps = rot_dim / 16
  // queries per work unit: B-operand groups of 32, four (two beyond 8 K steps) with 512 registers per wave, else two (one)
  const uint32_t group = f4 ? (nst <= 8 ? 128u : 64u) : (nch <= 4 ? 64u : 32u);
  CUVS_EXPECTS(f4 || (idx.pq_len == 2 && idx.codebook_kind == 0), "ivf_pq: pq_filter_kernel decodes pq_len 2, PER_SUBSPACE only");
  auto* units = static_cast<filter_unit*>(r.units);
  if (r.stage != 2) {
    hipLaunchKernelGGL(count_units_kernel, dim3(1), dim3(1024), 0, res.stream, r.pair_off, idx.n_lists, idx.list_sizes.data(),
                       r.unit_rows, r.unit_off, group, idx.n_lists);
    hipLaunchKernelGGL(fill_units_kernel, dim3(grid_blocks(idx.n_lists, 256)), dim3(256), 0, res.stream, r.pair_off, idx.n_lists,
                       idx.list_offsets.data(), idx.list_sizes.data(), r.unit_rows, r.unit_off, units, group, idx.n_lists);
  }
  filter_params f{};
  f.units = units; f.n_units = r.unit_off + idx.n_lists; f.xcd_ticket = r.xcd_ticket;
  f.sorted_pairs = r.sorted_pairs; f.rot_queries = r.rot_queries; f.centers_rot = idx.centers_rot.data();
  uint32_t nch8 = 0;
  const uint8_t* codes8 = pq3_codes(res, idx, &nch8);
  f.cb16 = tb.cb16; f.codes = codes8; f.list_offsets = idx.list_offsets.data(); f.list_sizes = idx.list_sizes.data();
  f.row_term = r.is_ip ? nullptr : tb.row_term; f.query_kth = r.query_kth; f.qflag = r.qflag;
  const unsigned grid = pq3_grid(res);
  // three quarters of the survivor buffer are cut into one region per workgroup (pq_filter4_kernel: per wave), the rest is
  // the shared spill region
  const unsigned regions = grid;
  f.surv = static_cast<uint2*>(r.surv); f.surv_cnt = r.surv_cnt; f.surv_cap = (uint32_t)((uint64_t)r.surv_cap * 3 / 4 / regions);
  f.spill_cap = r.surv_cap - f.surv_cap * regions;
  f.n_probes = r.n_probes; f.rot_dim = idx.rot_dim; f.unit_rows = r.unit_rows;
  f.sc = tb.sc; f.c1 = (r.is_ip ? -1.0f : -2.0f) / (tb.sc * tb.sc);
  f.cbmax = tb.cbmax; f.dmax = tb.dmax; f.is_ip = r.is_ip; f.stats = r.stats; f.dbg = r.filter_dbg;
  // exact score >= real score * (1 - eps) - alpha (L2: all entries are >= 0)
  //   fp32 LUT / fp32 score: 2 roundings per entry + 64 adds
  //   fp16 LUT: + 2^-11 per entry (2^-24 absolute below the normal range); fp16 score: + 2^-11 of the partial sum per add
  //   fp8 LUT (5 exponent bits, 3 value bits, truncation, half an ulp added back): 2^-4 per entry, 2^-15 absolute below its
  //   range; it saturates at 1.875 * 2^16, so bounds near that are not served
  f.bound_max = FLT_MAX;
  // (the figures are those of pq_dim 64; the per-add and per-entry terms grow with the number of entries summed)
  const float ne = std::max(1.0f, (float)idx.pq_dim / 64.0f);
  if (r.lut_mode == 0)      { f.eps = ne / 65536.0f; f.alpha = 0.f; }
  else if (r.lut_mode == 1) { f.eps = r.acc_half ? 0.04f * ne : 1.0f / 1024.0f; f.alpha = ne * 64.0f / 16777216.0f; f.bound_max = 60000.f; }
  else                      { f.eps = r.acc_half ? 0.07f + 0.04f * ne : 0.07f; f.alpha = ne * 64.0f / 32768.0f; f.bound_max = 30000.f; }
  if (r.is_ip && r.lut_mode == 2) f.eps = r.acc_half ? 0.14f + 0.04f * ne : 0.14f;  // signed fp8: one value bit less (2^-3 per entry)
  const size_t fsmem = (size_t)nch * 16 * 1024 + 16;
  if (f4) {
    // pre-pass (B operands and thresholds of every tail pair), then the filter: ivf_pq_filter4.hip
    filter4_launch l{};
    l.units = units; l.n_units = f.n_units; l.xcd_ticket = r.xcd_ticket; l.sorted_pairs = r.sorted_pairs; l.pair_off = r.pair_off;
    l.n_lists = idx.n_lists; l.probes = r.probes; l.rot_queries = r.rot_queries; l.centers_rot = f.centers_rot;
    l.query_kth = r.query_kth; l.qflag = r.qflag; l.bq = r.bq; l.thr = r.thr; l.cb16 = tb.cb16; l.codes = f.codes;
    l.list_offsets = f.list_offsets; l.list_sizes = f.list_sizes; l.row_term = reinterpret_cast<const float*>(f.row_term);
    l.surv = f.surv; l.surv_cnt = f.surv_cnt; l.surv_entries = r.surv_cap; l.n_probes = r.n_probes;
    l.rot_dim = idx.rot_dim; l.unit_rows = r.unit_rows; l.sc = f.sc; l.c1 = f.c1; l.eps = f.eps; l.alpha = f.alpha;
    l.cbmax = f.cbmax; l.dmax = f.dmax; l.bound_max = f.bound_max; l.is_ip = r.is_ip; l.dbg = r.filter_dbg; l.nch = nch;
    l.pl = (int)idx.pq_len; l.per_cluster = idx.codebook_kind != 0 ? 1 : 0;
    l.n_pairs = r.nq * (int64_t)r.n_probes; l.stats = r.stats; l.grid = grid;
    l.stage = r.stage; l.pair_norms = r.pair_norms;
    pq4_filter(res, l);
    if (r.stage == 1) return;  // (the helper stream's share: units, B operands, norms)
  } else {
  auto launch_filter = [&](auto kern) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem));
    profile_begin(res, "pq_filter_kernel");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kFThreads), fsmem, res.stream, f);
    profile_end(res, "pq_filter_kernel");
  };
  if (nch == 4) {
    switch (r.filter_dbg) {  // CUVS_AMD_SCAN_DEBUG bits 16..19: ablation builds of the kernel (bench shape only)
      case 1:  launch_filter(pq_filter_kernel<4, 1>); break;
      case 2:  launch_filter(pq_filter_kernel<4, 2>); break;
      case 4:  launch_filter(pq_filter_kernel<4, 4>); break;
      default: launch_filter(pq_filter_kernel<4, 0>); break;
    }
  } else {
    switch (nch) {
      case 1: launch_filter(pq_filter_kernel<1, 0>); break;
      case 2: launch_filter(pq_filter_kernel<2, 0>); break;
      case 3: launch_filter(pq_filter_kernel<3, 0>); break;
      case 5: launch_filter(pq_filter_kernel<5, 0>); break;
      case 6: launch_filter(pq_filter_kernel<6, 0>); break;
      case 7: launch_filter(pq_filter_kernel<7, 0>); break;
      default: launch_filter(pq_filter_kernel<8, 0>); break;
    }
  }
  }

  rescore_params s{};
  s.surv = f.surv; s.surv_cnt = r.surv_cnt; s.surv_cap = f.surv_cap; s.spill_cap = f.spill_cap; s.probes = r.probes; s.rot_queries = r.rot_queries;
  s.n_regions = regions; s.sub = f4 ? 0u : 1u;
  if (f4) s.surv_cap = r.surv_cap;  // chunked: the whole buffer
  s.centers_rot = idx.centers_rot.data(); s.pq_centers = idx.pq_centers.data(); s.codes = codes8;
  s.query_kth = r.query_kth; s.qflag = r.qflag; s.qcnt = r.qcnt; s.cand_d = r.cand_d; s.cand_i = r.cand_i; s.cand_r = r.cand_r;
  s.n_probes = r.n_probes; s.rot_dim = idx.rot_dim; s.k = r.k; s.head = r.head; s.is_ip = r.is_ip; s.n_chunks = nch8;
  s.head_rows = r.head_rows; s.list_offsets = idx.list_offsets.data();
  s.filter_bits = r.filter_bits; s.indices = idx.indices.data();
  s.overflow = static_cast<uint4*>(r.overflow); s.overflow_cnt = r.counters + 1; s.overflow_cap = r.overflow_cap; s.fail = nullptr;
  const size_t cb_bytes = (size_t)idx.rot_dim * idx.pq_book * sizeof(float);
  s.pq_len = idx.pq_len; s.book = idx.pq_book; s.per_cluster = idx.codebook_kind != 0 ? 1 : 0;
  s.cb_lds = (cb_bytes <= 128 * 1024 && !s.per_cluster) ? 1 : 0;
  const size_t rsmem = s.cb_lds ? cb_bytes : 16;
  // chunked survivor buffer (pq_filter4_kernel): every workgroup strides over all chunks, so ONE workgroup per CU is a full
  // wave of work and stages the codebook (128 KiB of LDS: one workgroup per CU anyway) once - with (grid + 1) x 2 workgroups
  // the staging ran twice per CU and was most of the kernel (0.16 ms for ~1 M survivors)
  const dim3 rg(f4 && s.cb_lds ? grid : grid + 1, f4 && s.cb_lds ? 1 : 2), rb(kRThreads);
  profile_begin(res, "pq_rescore_kernel");
  auto launch_rescore = [&](auto kern) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsmem));
    hipLaunchKernelGGL(kern, rg, rb, rsmem, res.stream, s);
  };
  auto pick_rescore = [&](auto lut_tag, auto acc_tag) {
    constexpr int LUT = decltype(lut_tag)::value;
    constexpr bool ACC = decltype(acc_tag)::value;
    if (s.cb_lds) launch_rescore(pq_rescore_kernel<LUT, ACC, true>); else launch_rescore(pq_rescore_kernel<LUT, ACC, false>);
  };
  using L0 = std::integral_constant<int, 0>; using L1 = std::integral_constant<int, 1>; using L2 = std::integral_constant<int, 2>;
  if (r.lut_mode == 0)      pick_rescore(L0{}, std::false_type{});
  else if (r.lut_mode == 1) { if (r.acc_half) pick_rescore(L1{}, std::true_type{}); else pick_rescore(L1{}, std::false_type{}); }
  else                      { if (r.acc_half) pick_rescore(L2{}, std::true_type{}); else pick_rescore(L2{}, std::false_type{}); }
  profile_end(res, "pq_rescore_kernel");
  // flagged queries: their candidate rows go back to "per-pair segments, nothing found yet", their tail pairs become
  // single-pair work items of the LUT scan kernel (launched by the caller)
  hipLaunchKernelGGL(reset_flagged_kernel, dim3((unsigned)r.nq), dim3(256), 0, res.stream, r.qflag, r.nq, r.n_probes, r.k, r.head,
                     r.cand_d, r.cand_i);
  hipLaunchKernelGGL(fallback_items_kernel, dim3(grid * 4), dim3(256), 0, res.stream, r.sorted_pairs, r.pair_off, idx.n_lists,
                     r.probes, r.n_probes, r.qflag, static_cast<work_item*>(r.fb_items), r.counters, 0);
  profile_end(res, "pq_scan_kernel");
}

static size_t head_smem_fixed(const ivf_pq_index& idx, int lut_mode, bool acc_half, int nt, int k)
{
  const bool lut32 = lut_mode == 0 || (lut_mode == 2 && !acc_half);
  return (size_t)idx.pq_dim * 256 * (lut32 ? 4 : 2) + 2 * 256 * 4 + 16 + 32 + 16 + (size_t)nt * 4 + 4 * (size_t)head_cand(k) * 4;
}

void pq3_head_scan(resources& res, const ivf_pq_index& idx, const pq3_head& h)
{
  head_params a{};
  a.items = static_cast<const work_item*>(h.items); a.item_begin = h.item_begin; a.item_end = h.item_end; a.n_lists = idx.n_lists;
  a.xcd_ticket = h.xcd_ticket; a.sorted_pairs = h.sorted_pairs; a.rot_queries = h.rot_queries; a.centers_rot = idx.centers_rot.data();
  a.pq_centers = idx.pq_centers.data(); a.codes = pq3_codes(res, idx, &a.n_chunks); a.list_offsets = idx.list_offsets.data();
  a.list_sizes = idx.list_sizes.data(); a.out_d = h.cand_d; a.out_i = h.cand_i; a.query_kth = h.query_kth;
  a.n_probes = h.n_probes; a.rot_dim = idx.rot_dim; a.k = h.k; a.is_ip = h.is_ip; a.pq_dim = idx.pq_dim;
  a.pq_len = idx.pq_len; a.book = idx.pq_book; a.per_cluster = idx.codebook_kind != 0 ? 1 : 0;
  a.filter_bits = h.filter_bits; a.indices = idx.indices.data(); a.stats = h.stats; a.hcand = head_cand((int)h.k);
  // A LUT of up to 32 KiB (fp16 entries at pq_dim 64): two 512-thread workgroups per CU, one streams its list while the
  // other selects; beyond: one 1024-thread workgroup. The rest of the LDS holds the score keys of a list chunk; longer
  // lists are scanned in chunks.
  const bool lut32 = h.lut_mode == 0 || (h.lut_mode == 2 && !h.acc_half);
  const bool small = (size_t)idx.pq_dim * 256 * (lut32 ? 4 : 2) <= 32 * 1024;
  const int nt = small ? 512 : 1024;
  const size_t budget = (small ? 80 : 160) * 1024 - 64, fixed = head_smem_fixed(idx, h.lut_mode, h.acc_half != 0, nt, (int)h.k);
  CUVS_EXPECTS(budget > fixed + 4096, "ivf_pq: the head-phase LUT does not fit the LDS");
  a.cap_rows = (uint32_t)(((budget - fixed) / 4) & ~size_t(63));
  if (h.max_list_len > 0) a.cap_rows = std::min<uint32_t>(a.cap_rows, (uint32_t)round_up(h.max_list_len, 64));
  const size_t smem   = fixed + (size_t)a.cap_rows * 4;
  a.one_shot          = h.one_shot;
  a.row_limit         = h.row_limit;
  const unsigned grid = h.one_shot != 0u ? h.one_shot : pq3_grid(res) * (small ? 2u : 1u);
  auto go = [&](auto kern) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    profile_begin(res, "pq_scan_kernel");
    profile_begin(res, "pq_head_kernel");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), smem, res.stream, a);
    profile_end(res, "pq_head_kernel");
    profile_end(res, "pq_scan_kernel");
  };
  auto pick = [&](auto lut_tag, auto acc_tag) {
    constexpr int LUT = decltype(lut_tag)::value;
    constexpr bool ACC = decltype(acc_tag)::value;
    if (small) go(pq_head_kernel<LUT, ACC, 512>); else go(pq_head_kernel<LUT, ACC, 1024>);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  if (h.lut_mode == 0)      pick(I0{}, std::false_type{});
  else if (h.lut_mode == 1) { if (h.acc_half) pick(I1{}, std::true_type{}); else pick(I1{}, std::false_type{}); }
  else                      { if (h.acc_half) pick(I2{}, std::true_type{}); else pick(I2{}, std::false_type{}); }
}

// up to 256 dimensions: pq_filter_kernel's FLAT build / flat_filter2_kernel; 256 / 384 / 512 / 768: the wide filter (ivf_pq_wide.hip)
bool flat3_wide(uint32_t dim) { return dim > 128 && pqw_shape(dim); }
bool flat3_supported(uint32_t dim, int k) { return ((dim % 32 == 0 && dim >= 32 && dim <= 256) || flat3_wide(dim)) && k <= 128; }

static bool flat3_prepare(resources& res, const flat3_view& v, flat3_cache& c)
{
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (c.data_ptr == v.data && c.rows == v.padded_rows && c.size == v.size) return true;
  // (an attempt that found no room is remembered for this state of the index: it is not repeated on every search)
  if (c.failed_ptr == v.data && c.failed_rows == v.padded_rows && c.failed_size == v.size) return false;
  const int64_t rows = std::max<int64_t>(v.padded_rows, 64);
  {
    // the copy takes half the size of the fp32 rows again: only when the device has that much to spare (idle scratch
    // blocks kept by the handles count as spare: they are given back first)
    c.rows16   = dev_buf<uint4>();
    c.row_term = dev_buf<uint32_t>();
    c.data_ptr = nullptr;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)rows * v.dim * 2 + (size_t)rows * 16 + (size_t(1) << 30);  // (fp16 rows: as large as the index again)
    if (free_b < need) {
      scratch_cache_flush_all();
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    }
    if (free_b < need) {
      c.failed_ptr = v.data; c.failed_rows = v.padded_rows; c.failed_size = v.size;
      return false;
    }
  }
  dev_buf<uint32_t> row_list(res, (size_t)rows / 64), mxd(res, 2);
  dev_buf<float> dn(res, (size_t)rows), inv_norm(res, v.unit_rows ? (size_t)rows : 0);
  HIP_TRY(hipMemsetAsync(row_list.data(), 0xff, row_list.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(mxd.data(), 0, 2 * sizeof(uint32_t), res.stream));
  hipLaunchKernelGGL(flat_group_lists_kernel, dim3(v.n_lists), dim3(64), 0, res.stream, v.list_offsets, v.list_sizes, v.n_lists,
                     row_list.data());
  c.rows16   = dev_buf<uint4>::persistent((size_t)rows / 32 * (v.dim / 16) * 64);
  c.row_term = dev_buf<uint32_t>::persistent((size_t)rows);
  const bool wide = flat3_wide(v.dim);  // the wide filter: natural K order, fp32 row terms
  c.row_term32 = wide ? dev_buf<float>::persistent((size_t)rows) : dev_buf<float>();
  c.zeros      = wide ? dev_buf<float>::persistent(32) : dev_buf<float>();
  if (wide) HIP_TRY(hipMemsetAsync(c.zeros.data(), 0, 32 * sizeof(float), res.stream));
  if (v.padded_rows > 0) {
    auto stats = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(grid_blocks(v.padded_rows, 256)), dim3(256), 0, res.stream, v.data, v.centers, row_list.data(),
                         v.padded_rows, v.dim, v.n_chunks, dn.data(), mxd.data(), inv_norm.data());
    };
    switch (v.elem) {
      case 0: stats(flat_residual_stats_kernel<float>); break;
      case 1: stats(flat_residual_stats_kernel<__half>); break;
      case 2: stats(flat_residual_stats_kernel<int8_t>); break;
      default: stats(flat_residual_stats_kernel<uint8_t>); break;
    }
    const auto mbits = to_host(res, mxd.data(), 2);
    float mx, mn2;
    memcpy(&mx, &mbits[0], 4);
    memcpy(&mn2, &mbits[1], 4);
    c.maxnorm = std::sqrt(mn2) * (1.0f + 1e-6f);
    // both GEMM operands are scaled by a power of two so that the largest residual component lands in (8, 16] (as the
    // codebook values of the IVF-PQ filter: the K-extension term stays below 16384)
    c.maxres = mx;
    c.sc     = mx > 0.f ? std::exp2(std::floor(std::log2(16.0f / mx))) : 1.0f;
    const int64_t n_t = v.padded_rows / 32 * (v.dim / 16) * 64;
    auto rows16 = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(grid_blocks(n_t, 256)), dim3(256), 0, res.stream, v.data, v.centers, row_list.data(), dn.data(),
                         v.padded_rows, v.dim, v.n_chunks, c.sc, c.rows16.data(), c.row_term.data(), inv_norm.data(), wide ? 1 : 0,
                         wide ? c.row_term32.data() : nullptr);
    };
    switch (v.elem) {
      case 0: rows16(flat_rows16_kernel<float>); break;
      case 1: rows16(flat_rows16_kernel<__half>); break;
      case 2: rows16(flat_rows16_kernel<int8_t>); break;
      default: rows16(flat_rows16_kernel<uint8_t>); break;
    }
  }
  sync(res);
  c.data_ptr = v.data; c.rows = v.padded_rows; c.size = v.size;
  return true;
}

bool flat3_head_bounds(resources& res, const flat3_view& v, flat3_cache& cache, const pq3_run& r, const flat3_head_bufs& hb)
{
  if (!flat3_prepare(res, v, cache)) return false;
  CUVS_EXPECTS(v.dim <= 128 && r.bq != nullptr && !r.is_ip, "ivf_flat: the bound-only head phase serves L2 up to 128 dimensions");
  profile_begin(res, "ivf_flat_scan_kernel");
  profile_begin(res, "flat_head_kernel");
  const float c1 = -2.0f / (cache.sc * cache.sc);
  hipLaunchKernelGGL(fill_f32_kernel, dim3(1024), dim3(256), 0, res.stream, hb.xbuf, (size_t)r.nq * hb.ldx, -INFINITY);
  // B operands + norms of the head pairs (the pre-pass without thresholds)
  filter4_launch l{};
  l.sorted_pairs = r.sorted_pairs; l.pair_off = r.pair_off; l.n_lists = v.n_lists; l.probes = r.probes;
  l.rot_queries = r.rot_queries; l.centers_rot = v.centers; l.query_kth = r.query_kth; l.qflag = r.qflag;
  l.bq = r.bq; l.thr = hb.thr_head; l.n_probes = r.n_probes; l.rot_dim = v.dim;
  l.sc = cache.sc; l.c1 = c1; l.eps = 1.0f / 65536.0f; l.alpha = 0.f; l.cbmax = cache.maxres; l.dmax = 0.f; l.bound_max = FLT_MAX;
  l.is_ip = 0; l.nch = (int)v.dim / 32; l.pl = 2; l.n_pairs = r.nq; l.flat = 1; l.head_labels = 1; l.stage = 1; l.pair_norms = hb.norms;
  pq4_filter(res, l);
  auto* units = static_cast<filter_unit*>(r.units);
  hipLaunchKernelGGL(count_units_kernel, dim3(1), dim3(1024), 0, res.stream, r.pair_off, v.n_lists, v.list_sizes, r.unit_rows, r.unit_off,
                     256u, 0u);
  hipLaunchKernelGGL(fill_units_kernel, dim3(grid_blocks(v.n_lists, 256)), dim3(256), 0, res.stream, r.pair_off, v.n_lists, v.list_offsets,
                     v.list_sizes, r.unit_rows, r.unit_off, units, 256u, 0u);
  filter_params f{};
  f.units = units; f.n_units = r.unit_off + v.n_lists; f.xcd_ticket = hb.tickets; f.sorted_pairs = r.sorted_pairs;
  f.list_offsets = v.list_offsets; f.list_sizes = v.list_sizes; f.row_term = cache.row_term.data(); f.rows16 = cache.rows16.data();
  f.bq = static_cast<const uint4*>(r.bq); f.pair_off = r.pair_off; f.n_lists = v.n_lists; f.lbase = 0u; f.xbuf = hb.xbuf; f.ldx = hb.ldx;
  f.n_probes = r.n_probes; f.rot_dim = v.dim; f.sc = cache.sc; f.c1 = c1;
  const unsigned grid = 2 * pq3_grid(res);
  auto launch_emit = [&](auto kern) {
    const size_t fsmem = (size_t)8 * (v.dim / 16) * 1024 + 2 * 8 * 32 * 4 + 16;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kF2Threads), fsmem, res.stream, f);
  };
  switch (v.dim / 16) {
    case 2:  launch_emit(flat_filter2_kernel<2, true>); break;
    case 4:  launch_emit(flat_filter2_kernel<4, true>); break;
    case 6:  launch_emit(flat_filter2_kernel<6, true>); break;
    default: launch_emit(flat_filter2_kernel<8, true>); break;
  }
  // the k largest values of every head pair (value = -(approximate score) up to a per-pair constant: larger is nearer)
  select_k<uint32_t, uint32_t>(res, hb.xbuf, nullptr, r.nq, (int64_t)hb.ldx, (int64_t)hb.ldx, (int)r.k, hb.kth_val, hb.kth_idx, false);
  head_bound_params b{};
  b.kth_val = hb.kth_val; b.kth_idx = hb.kth_idx; b.norms = static_cast<const float4*>(hb.norms); b.sorted_pairs = r.sorted_pairs;
  b.pair_off = r.pair_off; b.probes = r.probes; b.list_offsets = v.list_offsets; b.data = v.data;
  b.queries = r.rescore_queries != nullptr ? r.rescore_queries : r.rot_queries;
  b.n_lists = v.n_lists; b.n_probes = r.n_probes; b.k = r.k; b.dim = v.dim; b.n_chunks = v.n_chunks;
  b.sc = cache.sc; b.eps = l.eps; b.alpha = l.alpha; b.cbmax = cache.maxres; b.c1 = c1; b.rot_dim = v.dim;
  b.query_kth = const_cast<uint32_t*>(r.query_kth); b.thr_head = hb.thr_head; b.bound_max = FLT_MAX;
  const dim3 bgrid((unsigned)grid_blocks(r.nq, 4));
  switch (v.elem) {
    case 0: hipLaunchKernelGGL(flat_head_bound_kernel<float>, bgrid, dim3(256), 0, res.stream, b); break;
    case 1: hipLaunchKernelGGL(flat_head_bound_kernel<__half>, bgrid, dim3(256), 0, res.stream, b); break;
    case 2: hipLaunchKernelGGL(flat_head_bound_kernel<int8_t>, bgrid, dim3(256), 0, res.stream, b); break;
    default: hipLaunchKernelGGL(flat_head_bound_kernel<uint8_t>, bgrid, dim3(256), 0, res.stream, b); break;
  }
  profile_end(res, "flat_head_kernel");
  profile_end(res, "ivf_flat_scan_kernel");
  HIP_TRY(hipGetLastError());
  return true;
}

bool flat3_tail(resources& res, const flat3_view& v, flat3_cache& cache, const pq3_run& r)
{
  if (!flat3_prepare(res, v, cache)) return false;  // no room for the fp16 copy: the caller stays on the scan kernel
  profile_begin(res, "ivf_flat_scan_kernel");  // bench.py sums the scan phases under this name
  const int nch        = (int)v.dim / 32;   // the filter kernel's "chunk" = two K steps of 16 dimensions
  // flat_filter2_kernel: units of up to 256 queries per workgroup, B operands in LDS (the caller provides the pre-pass buffers)
  // (up to 128 dimensions: beyond, the operand registers of the subtile loop spill - 17 .. 148 registers - and round 3's kernel stays)
  const bool f2        = res.tune.flat_filter2 != 0 && r.bq != nullptr && v.dim <= 128;
  // 256 / 384 / 512 / 768 dimensions: the wide filter of the IVF-PQ path (ivf_pq_wide.hip) over the same fp16 copy (natural K order,
  // fp32 row terms) - units of up to 96 queries, B operands from its pre-pass in LDS; exact head phase, the same re-score and merge
  const bool wide      = flat3_wide(v.dim);
  CUVS_EXPECTS(!wide || r.bq != nullptr, "ivf_flat: the wide filter needs the pre-pass buffers");
  const uint32_t group = wide ? pqw_group() : f2 ? 256u : (nch <= 4 ? 64u : 32u);
  auto* units = static_cast<filter_unit*>(r.units);
  hipLaunchKernelGGL(count_units_kernel, dim3(1), dim3(1024), 0, res.stream, r.pair_off, v.n_lists, v.list_sizes, r.unit_rows, r.unit_off,
                     group, v.n_lists);
  hipLaunchKernelGGL(fill_units_kernel, dim3(grid_blocks(v.n_lists, 256)), dim3(256), 0, res.stream, r.pair_off, v.n_lists, v.list_offsets,
                     v.list_sizes, r.unit_rows, r.unit_off, units, group, v.n_lists);
  // flat_filter2_kernel: two 256-thread workgroups per CU while their operands fit the LDS twice (up to 128 dimensions)
  const unsigned grid = (f2 && !wide) ? 2 * pq3_grid(res) : pq3_grid(res);  // (two 256-thread workgroups per CU: 2 x 66 KiB of LDS)
  filter_params f{};
  f.units = units; f.n_units = r.unit_off + v.n_lists; f.xcd_ticket = r.xcd_ticket;
  f.sorted_pairs = r.sorted_pairs; f.rot_queries = r.rot_queries; f.centers_rot = v.centers;
  f.codes = v.data; f.list_offsets = v.list_offsets; f.list_sizes = v.list_sizes;
  f.row_term = r.is_ip ? nullptr : cache.row_term.data(); f.query_kth = r.query_kth; f.qflag = r.qflag;
  f.surv = static_cast<uint2*>(r.surv); f.surv_cnt = r.surv_cnt; f.surv_cap = (uint32_t)((uint64_t)r.surv_cap * 3 / 4 / grid);
  f.spill_cap = r.surv_cap - f.surv_cap * grid;
  f.n_probes = r.n_probes; f.rot_dim = v.dim; f.unit_rows = r.unit_rows;
  f.sc = cache.sc; f.c1 = (r.is_ip ? -1.0f : -2.0f) / (cache.sc * cache.sc); f.cbmax = cache.maxres; f.dmax = r.is_ip ? cache.maxnorm : 0.f;
  f.is_ip = r.is_ip; f.stats = r.stats; f.dbg = r.filter_dbg;
  f.eps = 1.0f / 65536.0f; f.alpha = 0.f; f.bound_max = FLT_MAX;  // fp32 fma chain over (q - x)^2: 2 roundings per term + 'dim' adds
  // inner product: score -(q . x) with x = c + d; the exact chain's error is below dim 2^-24 sum |q_i x_i| <= dim 2^-24 |q| (|c| + |d|)
  if (r.is_ip) f.eps = (float)v.dim * (1.0f / 4194304.0f);
  f.rows16 = cache.rows16.data(); f.fail = r.fail;
  auto launch_filter = [&](auto kern) {
    profile_begin(res, "flat_filter_kernel");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kFThreads), 16, res.stream, f);
    profile_end(res, "flat_filter_kernel");
  };
  auto launch_filter2 = [&](auto kern) {
    const size_t fsmem = (size_t)8 * (v.dim / 16) * 1024 + 2 * 8 * 32 * 4 + 16;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem));
    profile_begin(res, "flat_filter_kernel");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kF2Threads), fsmem, res.stream, f);
    profile_end(res, "flat_filter_kernel");
  };
  if (wide) {
    dev_buf<uint32_t> blk_off(res, (size_t)v.n_lists + 1);
    wide_prep l{};
    l.sorted_pairs = r.sorted_pairs; l.pair_off = r.pair_off; l.n_lists = v.n_lists; l.probes = r.probes; l.rot_queries = r.rot_queries;
    l.centers_rot = v.centers; l.query_kth = r.query_kth; l.qflag = r.qflag; l.bq = r.bq; l.blk_off = blk_off.data(); l.thr = r.thr;
    l.norms = nullptr; l.n_probes = r.n_probes; l.rot_dim = v.dim; l.heads = r.head; l.sc = f.sc; l.c1 = f.c1; l.eps = f.eps; l.alpha = f.alpha;
    l.cbmax = f.cbmax; l.dmax = f.dmax; l.bound_max = f.bound_max; l.head = 0; l.is_ip = r.is_ip; l.flat = 1;
    l.n_pairs = r.nq * (int64_t)r.n_probes;
    pqw_bprep(res, l);
    wide_filter w{};
    w.units = units; w.n_units = f.n_units; w.xcd_ticket = r.xcd_ticket; w.sorted_pairs = r.sorted_pairs; w.pair_off = r.pair_off;
    w.n_lists = v.n_lists; w.bq = r.bq; w.blk_off = blk_off.data(); w.thr = r.thr; w.rows16 = cache.rows16.data();
    w.row_term = r.is_ip ? nullptr : cache.row_term32.data(); w.zeros = cache.zeros.data(); w.filter_bits = nullptr; w.indices = nullptr;
    w.qflag = r.qflag; w.surv = f.surv; w.surv_cnt = r.surv_cnt; w.surv_cap = f.surv_cap; w.spill_cap = f.spill_cap; w.n_probes = r.n_probes;
    w.rot_dim = v.dim; w.xbuf = nullptr; w.ldx = 0; w.heads = r.head; w.emit = 0; w.grid = grid; w.stats = r.stats; w.fail = r.fail;
    w.profile_name = "flat_filter_kernel";
    pqw_filter(res, w);
  } else if (f2) {
    // pre-pass: every tail pair's fp16 B operand + threshold (the filter's unit prologue is a copy into LDS)
    filter4_launch l{};
    l.sorted_pairs = r.sorted_pairs; l.pair_off = r.pair_off; l.n_lists = v.n_lists; l.probes = r.probes;
    l.rot_queries = r.rot_queries; l.centers_rot = v.centers; l.query_kth = r.query_kth; l.qflag = r.qflag;
    l.bq = r.bq; l.thr = r.thr; l.n_probes = r.n_probes; l.rot_dim = v.dim;
    l.sc = f.sc; l.c1 = f.c1; l.eps = f.eps; l.alpha = f.alpha; l.cbmax = f.cbmax; l.dmax = f.dmax; l.bound_max = f.bound_max;
    l.is_ip = r.is_ip; l.nch = (int)v.dim / 32; l.pl = 2; l.n_pairs = r.nq * (int64_t)r.n_probes; l.flat = 1; l.bprep_only = 1;
    pq4_filter(res, l);
    f.bq = static_cast<const uint4*>(r.bq); f.thr_pair = r.thr; f.pair_off = r.pair_off; f.n_lists = v.n_lists;
    switch (v.dim / 16) {
      case 2:  launch_filter2(flat_filter2_kernel<2>); break;
      case 4:  launch_filter2(flat_filter2_kernel<4>); break;
      case 6:  launch_filter2(flat_filter2_kernel<6>); break;
      default: launch_filter2(flat_filter2_kernel<8>); break;
    }
  } else
  switch (nch) {
    case 1: launch_filter(pq_filter_kernel<1, 0, true>); break;
    case 2: launch_filter(pq_filter_kernel<2, 0, true>); break;
    case 3: launch_filter(pq_filter_kernel<3, 0, true>); break;
    case 4: launch_filter(pq_filter_kernel<4, 0, true>); break;
    case 5: launch_filter(pq_filter_kernel<5, 0, true>); break;
    case 6: launch_filter(pq_filter_kernel<6, 0, true>); break;
    case 7: launch_filter(pq_filter_kernel<7, 0, true>); break;
    default: launch_filter(pq_filter_kernel<8, 0, true>); break;
  }
  if (r.hb_xbuf != nullptr)  // the bound-only head phase: the head pairs' own candidates join the survivors (shared region)
    hipLaunchKernelGGL(flat_head_survivors_kernel, dim3((unsigned)grid_blocks(r.nq, 4)), dim3(256), 0, res.stream, r.hb_xbuf, r.hb_ldx, r.hb_thr,
                       r.sorted_pairs, r.pair_off, v.n_lists, r.probes, v.list_offsets, v.list_sizes,
                       f.surv, r.surv_cnt, grid, f.surv_cap, f.spill_cap, r.fail);
  rescore_params s{};
  s.surv = f.surv; s.surv_cnt = r.surv_cnt; s.surv_cap = f.surv_cap; s.spill_cap = f.spill_cap; s.probes = r.probes;
  s.n_regions = grid; s.sub = 1;
  s.rot_queries = r.rescore_queries != nullptr ? r.rescore_queries : r.rot_queries; s.codes = v.data; s.query_kth = r.query_kth; s.qflag = r.qflag; s.qcnt = r.qcnt;
  s.cand_d = r.cand_d; s.cand_i = r.cand_i; s.cand_r = r.cand_r; s.n_probes = r.n_probes; s.k = r.k; s.head = r.head;
  s.n_chunks = v.n_chunks; s.dim = v.dim; s.filter_bits = r.filter_bits; s.indices = v.indices; s.is_ip = r.cosine ? 2 : r.is_ip;
  s.overflow = static_cast<uint4*>(r.overflow); s.overflow_cnt = r.counters + 1; s.overflow_cap = r.overflow_cap; s.fail = r.fail;
  profile_begin(res, "flat_rescore_kernel");
  switch (v.elem) {
    case 0: hipLaunchKernelGGL(flat_rescore_kernel<float>, dim3(grid + 1, 8), dim3(256), 0, res.stream, s); break;
    case 1: hipLaunchKernelGGL(flat_rescore_kernel<__half>, dim3(grid + 1, 8), dim3(256), 0, res.stream, s); break;
    case 2: hipLaunchKernelGGL(flat_rescore_kernel<int8_t>, dim3(grid + 1, 8), dim3(256), 0, res.stream, s); break;
    default: hipLaunchKernelGGL(flat_rescore_kernel<uint8_t>, dim3(grid + 1, 8), dim3(256), 0, res.stream, s); break;
  }
  profile_end(res, "flat_rescore_kernel");
  profile_end(res, "ivf_flat_scan_kernel");
  return true;
}

// ------------------------------------------------------------------ the wide path (ivf_pq_wide.hip)
bool pqw_supported(const ivf_pq_index& idx, int k)
{
  return idx.codebook_kind == 0 && idx.pq_bits >= 4 && idx.pq_bits <= 8 && idx.pq_dim % 16 == 0 && idx.pq_dim >= 16 &&
         idx.rot_dim == idx.pq_len * idx.pq_dim && pqw_shape(idx.rot_dim) && k <= 256 && idx.shard_world <= 1;
}

uint32_t pqw_heads(const ivf_pq_index& idx, int k, uint32_t n_probes)
{
  uint64_t rows = 0, lists = 0;
  for (uint32_t v : idx.h_list_sizes) { rows += v; lists += v != 0u; }
  if (lists == 0 || rows == 0) return 0u;
  // k <= 2.5 % of the head lists' rows: heads * rows / lists >= 40 k (pq3_bound_useful's crossover is 4 %; measured at k = 256 of
  // 1.4 k-row lists, fp32 scores: 16.8 / 15.1 / 13.9 / 13.6 ms per batch at 3 / 5 / 8 / 12 head lists - a tighter bound saves more
  // re-scoring in the tail than the extra head lists cost in the emit pass)
  const uint64_t h = std::max<uint64_t>(1, ((uint64_t)40 * (uint64_t)k * lists + rows - 1) / rows);
  return 2 * h <= n_probes ? (uint32_t)h : 0u;
}

// exact score >= real score * (1 - eps) - alpha for the requested LUT / score types (L2: all entries are >= 0)
//   fp32 LUT / fp32 score: 2 roundings per entry + 64 adds
//   fp16 LUT: + 2^-11 per entry (2^-24 absolute below the normal range); fp16 score: + 2^-11 of the partial sum per add
//   fp8 LUT (5 exponent bits, 3 value bits, truncation, half an ulp added back): 2^-4 per entry, 2^-15 absolute below its
//   range; it saturates at 1.875 * 2^16, so bounds near that are not served
// (the figures are those of pq_dim 64 and pq_len 2; the per-add and per-entry terms grow with the number of entries summed, the
// entries' own chains - pq_len fused multiply-adds in fp32 - with pq_len: 2^-24 each, far inside the fp32 figure's slack)
static void pqw_margins(const ivf_pq_index& idx, const pq3_run& r, float* eps, float* alpha, float* bound_max)
{
  const float ne = std::max(1.0f, (float)idx.pq_dim / 64.0f) * std::max(1.0f, (float)idx.pq_len / 8.0f);
  *bound_max = FLT_MAX;
  if (r.lut_mode == 0)      { *eps = ne / 65536.0f; *alpha = 0.f; }
  else if (r.lut_mode == 1) { *eps = r.acc_half ? 0.04f * ne : 1.0f / 1024.0f; *alpha = ne * 64.0f / 16777216.0f; *bound_max = 60000.f; }
  else                      { *eps = r.acc_half ? 0.07f + 0.04f * ne : 0.07f; *alpha = ne * 64.0f / 32768.0f; *bound_max = 30000.f; }
  if (r.is_ip && r.lut_mode == 2) *eps = r.acc_half ? 0.14f + 0.04f * ne : 0.14f;  // signed fp8: one value bit less (2^-3 per entry)
}

// the decoded rows (nullptr: no room on the device - remembered for this state of the index)
static const void* pqw_rows(resources& res, const ivf_pq_index& idx, pq3_tables* tb_out)
{
  std::lock_guard<std::recursive_mutex> lock(g_pq3_mu);
  const pq3_tables tb = pq3_prepare(res, idx, true);
  if (tb_out != nullptr) *tb_out = tb;
  auto& c = idx.scan3;
  if (c.w_codes == idx.codes.data() && c.w_rows == idx.padded_rows && c.w_size == idx.size && c.w_pq == idx.pq_centers.data())
    return c.rows16w.data();
  if (c.w_failed_codes == idx.codes.data() && c.w_failed_rows == idx.padded_rows && c.w_failed_size == idx.size) return nullptr;
  c.rows16w = dev_buf<uint4>();
  c.w_codes = nullptr;
  const int64_t rows = std::max<int64_t>(idx.padded_rows, 64);
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t need = (size_t)rows * idx.rot_dim * 2 + (size_t(1) << 30);
  if (free_b < need) {
    scratch_cache_flush_all();
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  }
  if (free_b < need) {
    c.w_failed_codes = idx.codes.data(); c.w_failed_rows = idx.padded_rows; c.w_failed_size = idx.size;
    return nullptr;
  }
  c.rows16w = dev_buf<uint4>::persistent((size_t)rows * idx.rot_dim / 8);
  c.cbt     = dev_buf<float>::persistent((size_t)idx.pq_dim * 256 * idx.pq_len);
  c.zeros   = dev_buf<float>::persistent(32);
  HIP_TRY(hipMemsetAsync(c.zeros.data(), 0, 32 * sizeof(float), res.stream));
  hipLaunchKernelGGL(cbt_kernel, dim3(grid_blocks((int64_t)idx.pq_dim * 256, 256)), dim3(256), 0, res.stream, idx.pq_centers.data(), idx.pq_dim,
                     idx.pq_len, idx.pq_book, c.cbt.data());
  uint32_t nch8 = 0;
  const uint8_t* codes8 = pq3_codes(res, idx, &nch8);
  pqw_decode(res, codes8, nch8, tb.cb16, idx.pq_len, idx.padded_rows, idx.rot_dim, c.rows16w.data());
  sync(res);  // published (to the other threads sharing the index) only once it is filled
  c.w_codes = idx.codes.data(); c.w_rows = idx.padded_rows; c.w_size = idx.size; c.w_pq = idx.pq_centers.data();
  return c.rows16w.data();
}

bool pqw_ready(resources& res, const ivf_pq_index& idx) { return pqw_rows(res, idx, nullptr) != nullptr; }

static rescore_params pqw_score_inputs(const ivf_pq_index& idx, const pq3_run& r, const uint8_t* codes8, uint32_t nch8)
{
  rescore_params s{};
  s.probes = r.probes; s.rot_queries = r.rot_queries; s.centers_rot = idx.centers_rot.data(); s.pq_centers = idx.pq_centers.data();
  s.codes = codes8; s.n_chunks = nch8; s.rot_dim = idx.rot_dim; s.pq_len = idx.pq_len; s.book = idx.pq_book; s.per_cluster = 0;
  s.is_ip = r.is_ip; s.n_probes = r.n_probes; s.k = r.k; s.head = r.head; s.cbt = idx.scan3.cbt.data();
  return s;
}

bool pqw_head_bounds(resources& res, const ivf_pq_index& idx, const pq3_run& r, const pqw_bufs& hb)
{
  pq3_tables tb{};
  const void* rows16 = pqw_rows(res, idx, &tb);
  if (rows16 == nullptr) return false;
  profile_begin(res, "pq_scan_kernel");
  profile_begin(res, "pq_head_kernel");
  const float c1 = (r.is_ip ? -1.0f : -2.0f) / (tb.sc * tb.sc);
  float eps = 0.f, alpha = 0.f, bound_max = FLT_MAX;
  pqw_margins(idx, r, &eps, &alpha, &bound_max);
  hipLaunchKernelGGL(fill_f32_kernel, dim3(1024), dim3(256), 0, res.stream, hb.xbuf, (size_t)r.nq * r.head * hb.ldx, -INFINITY);
  // B operands, norms and constants of the head pairs
  wide_prep l{};
  l.sorted_pairs = r.sorted_pairs; l.pair_off = r.pair_off; l.n_lists = idx.n_lists; l.probes = r.probes; l.rot_queries = r.rot_queries;
  l.centers_rot = idx.centers_rot.data(); l.query_kth = r.query_kth; l.qflag = r.qflag; l.bq = r.bq; l.thr = hb.head_c; l.norms = hb.norms;
  l.n_probes = r.n_probes; l.rot_dim = idx.rot_dim; l.heads = r.head; l.sc = tb.sc; l.c1 = c1; l.eps = eps; l.alpha = alpha;
  l.cbmax = tb.cbmax; l.dmax = tb.dmax; l.bound_max = bound_max; l.head = 1; l.n_pairs = r.nq * (int64_t)r.head; l.blk_off = hb.blk_off; l.is_ip = r.is_ip;
  pqw_bprep(res, l);
  auto* units = static_cast<filter_unit*>(r.units);
  hipLaunchKernelGGL(count_units_kernel, dim3(1), dim3(1024), 0, res.stream, r.pair_off, idx.n_lists, idx.list_sizes.data(), r.unit_rows,
                     r.unit_off, pqw_group(), 0u);
  hipLaunchKernelGGL(fill_units_kernel, dim3(grid_blocks(idx.n_lists, 256)), dim3(256), 0, res.stream, r.pair_off, idx.n_lists,
                     idx.list_offsets.data(), idx.list_sizes.data(), r.unit_rows, r.unit_off, units, pqw_group(), 0u);
  wide_filter f{};
  f.units = units; f.n_units = r.unit_off + idx.n_lists; f.xcd_ticket = hb.tickets; f.sorted_pairs = r.sorted_pairs; f.pair_off = r.pair_off;
  f.n_lists = idx.n_lists; f.bq = r.bq; f.blk_off = hb.blk_off; f.thr = hb.head_c; f.rows16 = rows16;
  f.row_term = r.is_ip ? nullptr : reinterpret_cast<const float*>(tb.row_term); f.zeros = idx.scan3.zeros.data();
  f.qflag = r.qflag; f.surv = r.surv; f.surv_cnt = r.surv_cnt; f.surv_cap = 0; f.spill_cap = 0; f.n_probes = r.n_probes; f.rot_dim = idx.rot_dim;
  f.xbuf = hb.xbuf; f.ldx = hb.ldx; f.heads = r.head; f.emit = 1; f.grid = pq3_grid(res); f.stats = nullptr;
  f.filter_bits = r.filter_bits; f.indices = idx.indices.data();
  pqw_filter(res, f);
  // the k largest values of every query over its head lists
  select_k<uint32_t, uint32_t>(res, hb.xbuf, nullptr, r.nq, (int64_t)r.head * hb.ldx, (int64_t)r.head * hb.ldx, (int)r.k, hb.kth_val, hb.kth_idx,
                               false);
  uint32_t nch8 = 0;
  const uint8_t* codes8 = pq3_codes(res, idx, &nch8);
  wbound_params b{};
  b.rs = pqw_score_inputs(idx, r, codes8, nch8);
  b.kth_val = hb.kth_val; b.kth_idx = hb.kth_idx; b.norms = static_cast<const float4*>(hb.norms); b.probes = r.probes;
  b.list_offsets = idx.list_offsets.data(); b.nq = r.nq; b.k = r.k; b.heads = r.head; b.ldx = hb.ldx; b.n_probes = r.n_probes;
  b.rot_dim = idx.rot_dim; b.query_kth = const_cast<uint32_t*>(r.query_kth); b.qflag = r.qflag; b.thr_head = hb.thr_head;
  b.sc = tb.sc; b.c1 = c1; b.eps = eps; b.alpha = alpha; b.cbmax = tb.cbmax; b.dmax = tb.dmax; b.bound_max = bound_max; b.is_ip = r.is_ip;
  const dim3 bgrid((unsigned)grid_blocks(r.nq, 4));
  const bool bblocks = idx.pq_len == 2 && idx.pq_dim % 64 == 0 && res.tune.pq_wide_blocks != 0 && hb.bound_tmp != nullptr;
  auto bound = [&](auto lut_tag, auto acc_tag) {
    constexpr int LUT = decltype(lut_tag)::value;
    constexpr bool ACC = decltype(acc_tag)::value;
    if (bblocks) {
      HIP_TRY(hipMemsetAsync(hb.bound_tmp, 0, (size_t)2 * r.nq * sizeof(uint32_t), res.stream));
      const size_t smb = (size_t)64 * 2 * idx.pq_book * sizeof(float);
      auto kb = pqw_bound_blocks_kernel<LUT, ACC>;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smb));
      const unsigned gb = (unsigned)std::min<int64_t>(pq3_grid(res), grid_blocks(r.nq * (int64_t)r.k, kRBThreads * kRBItems));
      hipLaunchKernelGGL(kb, dim3(gb), dim3(kRBThreads), smb, res.stream, b, hb.bound_tmp);
      hipLaunchKernelGGL(pqw_bound_finish_kernel, bgrid, dim3(256), 0, res.stream, b, hb.bound_tmp);
      return;
    }
    const size_t sm = 4 * wave_score_tile<LUT, ACC>::kBytes;
    auto kern = pqw_head_bound_kernel<LUT, ACC>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL(kern, bgrid, dim3(256), sm, res.stream, b);
  };
  using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>; using B2 = std::integral_constant<int, 2>;
  if (r.lut_mode == 0)      bound(B0{}, std::false_type{});
  else if (r.lut_mode == 1) { if (r.acc_half) bound(B1{}, std::true_type{}); else bound(B1{}, std::false_type{}); }
  else                      { if (r.acc_half) bound(B2{}, std::true_type{}); else bound(B2{}, std::false_type{}); }
  profile_end(res, "pq_head_kernel");
  profile_end(res, "pq_scan_kernel");
  HIP_TRY(hipGetLastError());
  return true;
}

void pqw_tail(resources& res, const ivf_pq_index& idx, const pq3_run& r, const pqw_bufs& hb)
{
  pq3_tables tb{};
  const void* rows16 = pqw_rows(res, idx, &tb);
  CUVS_EXPECTS(rows16 != nullptr, "ivf_pq: the wide path's decoded rows are gone between the head and the tail phase");
  profile_begin(res, "pq_scan_kernel");
  const float c1 = (r.is_ip ? -1.0f : -2.0f) / (tb.sc * tb.sc);
  float eps = 0.f, alpha = 0.f, bound_max = FLT_MAX;
  pqw_margins(idx, r, &eps, &alpha, &bound_max);
  wide_prep l{};
  l.sorted_pairs = r.sorted_pairs; l.pair_off = r.pair_off; l.n_lists = idx.n_lists; l.probes = r.probes; l.rot_queries = r.rot_queries;
  l.centers_rot = idx.centers_rot.data(); l.query_kth = r.query_kth; l.qflag = r.qflag; l.bq = r.bq; l.thr = r.thr; l.norms = nullptr;
  l.n_probes = r.n_probes; l.rot_dim = idx.rot_dim; l.heads = r.head; l.sc = tb.sc; l.c1 = c1; l.eps = eps; l.alpha = alpha;
  l.cbmax = tb.cbmax; l.dmax = tb.dmax; l.bound_max = bound_max; l.head = 0; l.n_pairs = r.nq * (int64_t)r.n_probes; l.blk_off = hb.blk_off; l.is_ip = r.is_ip;
  profile_begin(res, "pq_bprep_kernel");
  pqw_bprep(res, l);
  profile_end(res, "pq_bprep_kernel");
  auto* units = static_cast<filter_unit*>(r.units);
  hipLaunchKernelGGL(count_units_kernel, dim3(1), dim3(1024), 0, res.stream, r.pair_off, idx.n_lists, idx.list_sizes.data(), r.unit_rows,
                     r.unit_off, pqw_group(), idx.n_lists);
  hipLaunchKernelGGL(fill_units_kernel, dim3(grid_blocks(idx.n_lists, 256)), dim3(256), 0, res.stream, r.pair_off, idx.n_lists,
                     idx.list_offsets.data(), idx.list_sizes.data(), r.unit_rows, r.unit_off, units, pqw_group(), idx.n_lists);
  const unsigned grid = pq3_grid(res);
  wide_filter f{};
  f.units = units; f.n_units = r.unit_off + idx.n_lists; f.xcd_ticket = r.xcd_ticket; f.sorted_pairs = r.sorted_pairs; f.pair_off = r.pair_off;
  f.n_lists = idx.n_lists; f.bq = r.bq; f.blk_off = hb.blk_off; f.thr = r.thr; f.rows16 = rows16;
  f.row_term = r.is_ip ? nullptr : reinterpret_cast<const float*>(tb.row_term); f.zeros = idx.scan3.zeros.data();
  f.qflag = r.qflag; f.surv = r.surv; f.surv_cnt = r.surv_cnt; f.surv_cap = (uint32_t)((uint64_t)r.surv_cap * 3 / 4 / grid);
  f.spill_cap = r.surv_cap - f.surv_cap * grid; f.n_probes = r.n_probes; f.rot_dim = idx.rot_dim; f.xbuf = nullptr; f.ldx = 0; f.heads = r.head;
  f.emit = 0; f.grid = grid; f.stats = r.stats;
  pqw_filter(res, f);
  hipLaunchKernelGGL(pqw_head_survivors_kernel, dim3((unsigned)grid_blocks(r.nq * (int64_t)r.head, 4)), dim3(256), 0, res.stream, hb.xbuf, hb.ldx,
                     hb.thr_head, r.head, r.sorted_pairs, r.pair_off, idx.n_lists, r.probes, r.n_probes, idx.list_offsets.data(),
                     idx.list_sizes.data(), static_cast<uint2*>(r.surv), r.surv_cnt, grid, f.surv_cap, f.spill_cap, r.qflag);
  uint32_t nch8 = 0;
  const uint8_t* codes8 = pq3_codes(res, idx, &nch8);
  rescore_params s = pqw_score_inputs(idx, r, codes8, nch8);
  s.surv = static_cast<const uint2*>(r.surv); s.surv_cnt = r.surv_cnt; s.surv_cap = f.surv_cap; s.spill_cap = f.spill_cap;
  s.n_regions = grid; s.sub = 1u;
  s.query_kth = r.query_kth; s.qflag = r.qflag; s.qcnt = r.qcnt; s.cand_d = r.cand_d; s.cand_i = r.cand_i; s.cand_r = r.cand_r;
  s.head_rows = 0u; s.list_offsets = idx.list_offsets.data(); s.filter_bits = r.filter_bits; s.indices = idx.indices.data();
  s.overflow = static_cast<uint4*>(r.overflow); s.overflow_cnt = r.counters + 1; s.overflow_cap = r.overflow_cap; s.fail = nullptr;
  s.cb_lds = 0;
  // a wave per survivor (pq_rescore_wave_kernel): the batches of all regions dealt to 8 workgroups of 4 waves per CU
  const dim3 rg(grid, 8), rb(256);
  profile_begin(res, "pq_rescore_kernel");
  // pq_len 2 with whole blocks of 64 subspaces: the codebook staged block by block in LDS (pq_rescore_blocks_kernel). (Measured and
  // rejected for longer entries - a lane per survivor walking the block's subspaces one by one, pq_len lookups each: 1M x 768, pq_len 4:
  // 3.74 -> 4.25 ms per 10k queries, pq_len 12: 2.90 -> 3.31 - the wave per survivor reads such entries as whole 16-byte pieces.)
  const bool blocks = idx.pq_len == 2 && idx.pq_dim % 64 == 0 && res.tune.pq_wide_blocks != 0;
  auto rescore = [&](auto lut_tag, auto acc_tag) {
    constexpr int LUT = decltype(lut_tag)::value;
    constexpr bool ACC = decltype(acc_tag)::value;
    if (blocks) {
      const size_t sm = (size_t)64 * 2 * idx.pq_book * sizeof(float) + ((size_t)grid + 3) * sizeof(uint32_t);
      auto kern = pq_rescore_blocks_kernel<LUT, ACC>;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(kRBThreads), sm, res.stream, s);
      return;
    }
    const size_t sm = 4 * wave_score_tile<LUT, ACC>::kBytes + ((size_t)grid + 3) * sizeof(uint32_t);
    auto kern = pq_rescore_wave_kernel<LUT, ACC>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL(kern, rg, rb, sm, res.stream, s);
  };
  using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
  if (r.lut_mode == 0)      rescore(R0{}, std::false_type{});
  else if (r.lut_mode == 1) { if (r.acc_half) rescore(R1{}, std::true_type{}); else rescore(R1{}, std::false_type{}); }
  else                      { if (r.acc_half) rescore(R2{}, std::true_type{}); else rescore(R2{}, std::false_type{}); }
  profile_end(res, "pq_rescore_kernel");
  // flagged queries: ALL their candidate rows back to "nothing found yet", ALL their pairs single-pair items of the LUT scan
  hipLaunchKernelGGL(reset_flagged_kernel, dim3((unsigned)r.nq), dim3(256), 0, res.stream, r.qflag, r.nq, r.n_probes, r.k, 0u, r.cand_d,
                     r.cand_i);
  hipLaunchKernelGGL(fallback_items_kernel, dim3(grid * 4), dim3(256), 0, res.stream, r.sorted_pairs, r.pair_off, idx.n_lists, r.probes,
                     r.n_probes, r.qflag, static_cast<work_item*>(r.fb_items), r.counters, 1);
  profile_end(res, "pq_scan_kernel");
  HIP_TRY(hipGetLastError());
}

// (Measured and rejected, round 6: the k > 64 merge as a workgroup SORT - candidates to LDS as (score key, rank << 32 | row), one bitonic
// sort, the first k re-sorted by (score, row); exact (245 tests) but 66 barrier-separated stages per query cost what the serial
// insertions cost: a CAGRA build on 10M x 128 rows 8.4 -> 8.2 s, PQ-768 at k = 100 4.39 -> 4.35 ms.)
void pq3_merge(resources& res, const pq3_run& r, float* top_d, uint32_t* top_i)
{
  auto* ov = static_cast<uint4*>(r.overflow);
  hipLaunchKernelGGL(ov_count_kernel, dim3(256), dim3(256), 0, res.stream, ov, r.counters + 1, r.overflow_cap, r.ov_cnt);
  hipLaunchKernelGGL(ov_scan_kernel, dim3(1), dim3(1024), 0, res.stream, r.ov_cnt, r.nq, r.ov_off);
  hipLaunchKernelGGL(ov_fill_kernel, dim3(256), dim3(256), 0, res.stream, ov, r.counters + 1, r.overflow_cap, r.ov_off, r.ov_cnt,
                     ov + r.overflow_cap);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid_blocks(r.nq, 4)), dim3(256), 0, res.stream, r.cand_d, r.cand_i, r.cand_r, r.qcnt, r.qflag, r.nq,
                       r.n_probes, r.k, r.head, top_d, top_i, ov + r.overflow_cap, r.ov_off);
  };
  if (r.k <= 64)       go(pool_merge_kernel<1>);
  else if (r.k <= 128) go(pool_merge_kernel<2>);
  else if (r.k <= 192) go(pool_merge_kernel<3>);
  else                 go(pool_merge_kernel<4>);
}

}  // namespace cuvs_amd
