// IVF-PQ search, warm-bounds phase on the matrix cores for the shapes pq_filter4_kernel does not decode (ivf_pq_scan3.hip has
// the scheme): rot_dim beyond 256 (the reference's default at 768 dimensions: pq_dim 384 x pq_len 2; a CAGRA build's kNN-graph
// search: pq_dim 64 x pq_len 12), pq_len outside {1, 2, 4, 8}, and k a large fraction of a list (the head phase then bounds with
// several lists). Reference semantics: compute_score_impl.cuh:52-79 / ivf_pq_search.cuh:421-669 - the filter only decides which
// (row, query) pairs the reference's arithmetic has to look at.
//
// pq_filter4_kernel decodes rows into MFMA A operands on the fly, through a decode table in LDS: rot_dim x 512 bytes - 384 KiB at
// 768 dimensions, against 160 KiB of LDS. Here the decode is hoisted out of the search: the index keeps its rows DECODED, as scaled
// fp16, laid out as the A operands themselves ([32-row tile][K step][lane] x 16 bytes - the layout of IVF-Flat's fp16 copy; 288 GB
// of HBM pay for rows x rot_dim x 2 bytes) - the same fp16 values pq_filter4_kernel's table holds, so the same thresholds
// (filter_threshold) make the same guarantee. The filter is then a GEMM with a compare for an epilogue:
//   unit       (row chunk of a list) x (up to 96 queries probing it); the queries' fp16 residuals - B operands, from the pre-pass -
//              sit in LDS for the whole unit ([3 groups][K steps][64 lanes] x 16 bytes = 144 KiB at 768 dimensions);
//   wave       one per SIMD with up to 512 registers; takes strips of 64 rows: 2 subtiles x 3 query groups of fp32 accumulators
//              (96 registers), initialised with the rows' terms; the A operands of the strip arrive in chunks of 8 K steps
//              (384 dimensions: of 4) through a ring of 2 (384-d: 3) register sets, a chunk is asked for one chunk (~1.5 k
//              cycles of matrix work) before it is multiplied; per K step 3 ds_read_b128 (B) feed 6 MFMAs;
//   epilogue   at the end of the strip's K loop: a pair survives when acc >= thr (as in the other filters).
// EMIT build: the bound-only head phase (values to a buffer instead of the compare), as IVF-Flat's (3.1c). Inner product / cosine:
// the query itself is the operand and the accumulators start from zero (pqw_bprep_kernel, filter_threshold_ip).
#include "ivf_pq_filter_common.hpp"

#include <cfloat>
#include <type_traits>

namespace cuvs_amd {

namespace {

// ------------------------------------------------------------------ the decoded copy
// one thread per 16-byte piece of the copy: (32-row tile, K step st, lane = (row ql, K half h)) holds the rotated dimensions
// 16 st + 8 h .. + 7 of the row's decoded residual, scaled (the decode table's values: cb16_kernel)
__global__ void pqw_decode_kernel(const uint8_t* __restrict__ codes8, uint32_t n_chunks, const uint16_t* __restrict__ cb16,
                                  uint32_t pq_len, int64_t n_pieces, uint32_t nst, uint4* __restrict__ rows16)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pieces) return;
  const uint32_t lane = (uint32_t)(i & 63), ql = lane & 31u, h = lane >> 5;
  const int64_t ts   = i >> 6;
  const uint32_t st  = (uint32_t)(ts % nst);
  const int64_t row  = (ts / nst) * 32 + ql;
  const uint8_t* cr  = codes8 + (size_t)(row >> 6) * n_chunks * 1024 + (size_t)(row & 63) * 16;
  uint32_t w[4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    uint32_t v[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const uint32_t j = 16u * st + 8u * h + (uint32_t)(e + o);
      const uint32_t s = j / pq_len, l = j - s * pq_len;
      const uint32_t code = cr[(size_t)(s >> 4) * 1024 + (s & 15u)];
      v[o] = cb16[((size_t)s * 256 + code) * pq_len + l];
    }
    w[e >> 1] = v[0] | (v[1] << 16);
  }
  rows16[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------ pre-pass: B operands, norms / thresholds
// The B operands are written in the layout a work unit reads them in: the pairs of a list are cut into BLOCKS of 32 (the last one
// padded with zero operands), block b of the batch is [K step][lane = (K half h, pair ql)] x 16 bytes - NST KiB in one piece - and the
// blocks of a list follow each other (blk_off[L]: the list's first block). A unit's query group is one block: its operands go to LDS
// as whole 1 KiB rows (the pair-major layout of the first version made the unit prologue - 16-byte pieces 1.5 KiB apart - a third of
// the filter kernel: 54 k cycles per unit against 107 k for its rows, measured).
__global__ __launch_bounds__(1024) void pqw_pair_blocks_kernel(const uint32_t* __restrict__ pair_off, uint32_t n_lists, uint32_t lbase,
                                                               uint32_t* __restrict__ blk_off)
{
  __shared__ int smem[17];
  const uint32_t per = (n_lists + 1023u) / 1024u;
  const uint32_t b = threadIdx.x * per, e = min(n_lists, b + per);
  auto blocks_of = [&](uint32_t i) { return (int)((pair_off[lbase + i + 1] - pair_off[lbase + i] + 31u) >> 5); };
  int sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += blocks_of(i);
  int total;
  int run = block_exclusive_scan(sum, smem, &total);
  for (uint32_t i = b; i < e; ++i) {
    blk_off[i] = (uint32_t)run;
    run += blocks_of(i);
  }
  if (threadIdx.x == 0) blk_off[n_lists] = (uint32_t)total;
}

struct wprep_params {
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;
  uint32_t n_lists, lbase;
  const uint32_t* probes;
  const float* rot_queries;
  const float* centers_rot;
  const uint32_t* query_kth;
  uint32_t* qflag;
  const uint32_t* blk_off;  // [n_lists + 1] first block of every list
  uint4* bq;      // [block][K step][K half h][pair ql] x 16 bytes
  float* thr;     // tail pairs: [pair position] threshold in accumulator units; head pairs: the pair's constant -|r|^2 sc^2 / 2
  float4* norms;  // head pairs: [query * heads + probe rank] (|r|^2, ., ., largest scaled operand)
  uint32_t n_probes, rot_dim, heads;
  float sc, c1, eps, alpha, cbmax, dmax, bound_max;
  int head, is_ip, flat;
};

// one wave per block of 32 pairs: lane = (K half h, pair ql); K step st holds, in K half h, the rotated dimensions 16 st + 8 h .. + 7
__global__ __launch_bounds__(256) void pqw_bprep_kernel(const wprep_params a)
{
  const uint32_t blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u, ql = lane & 31u, h = lane >> 5;
  if (blk >= a.blk_off[a.n_lists]) return;  // wave-uniform
  uint32_t lo = 0u, hi = a.n_lists;  // the block's list: blk_off[lo] <= blk < blk_off[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.blk_off[mid] <= blk) lo = mid; else hi = mid;
  }
  const uint32_t L = lo, s0 = a.pair_off[a.lbase + L], np = a.pair_off[a.lbase + L + 1] - s0;
  const uint32_t pos = (blk - a.blk_off[L]) * 32u + ql;
  const bool have = pos < np;
  const uint32_t s = s0 + min(pos, np - 1u), s_base = a.pair_off[a.lbase];
  const uint32_t p = a.sorted_pairs[s], q = p / a.n_probes;
  const float* rq = a.rot_queries + (size_t)q * a.rot_dim + 8u * h;
  const float* ct = a.centers_rot + (size_t)L * a.rot_dim + 8u * h;
  const uint32_t nst = a.rot_dim / 16u;
  uint4* out = a.bq + (size_t)blk * nst * 64 + lane;
  // L2: the operand is the residual q - c; inner product / cosine: the query itself (score -(q.c + q.d): q.c and |c|^2 enter the threshold)
  float rn = 0.f, big = 0.f, qc = 0.f, cn = 0.f;
#pragma unroll 4
  for (uint32_t st = 0; st < nst; ++st) {
    const float4 q0 = *reinterpret_cast<const float4*>(rq + st * 16u), q1 = *reinterpret_cast<const float4*>(rq + st * 16u + 4u);
    const float4 c0 = *reinterpret_cast<const float4*>(ct + st * 16u), c1 = *reinterpret_cast<const float4*>(ct + st * 16u + 4u);
    float r[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
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
      rn = __fmaf_rn(r[e], r[e], rn);
      const float x = a.sc * r[e];
      big  = fmaxf(big, fabsf(x));
      v[e] = have ? (_Float16)fminf(fmaxf(x, -60000.f), 60000.f) : (_Float16)0.f;  // (a padding slot: zero operands)
    }
    out[(size_t)st * 64] = __builtin_bit_cast(uint4, v);
  }
  rn += __shfl_xor(rn, 32); qc += __shfl_xor(qc, 32); cn += __shfl_xor(cn, 32);
  big = fmaxf(big, __shfl_xor(big, 32));
  if (!have || h != 0u) return;
  if (a.head) {
    // the pair's constant: what turns the accumulator into a value comparable across the query's head lists
    //   L2: score = |r|^2 - 2 r.d + |d|^2 -> acc - |r|^2 sc^2 / 2;   inner product: score = -(q.c + q.d) -> acc + (q.c) sc^2
    a.norms[(size_t)q * a.heads + p % a.n_probes] = make_float4(rn, cn, qc, big);
    a.thr[s - s_base] = a.is_ip ? a.sc * a.sc * qc : -0.5f * a.sc * a.sc * rn;
  } else {
    const uint32_t kk = a.query_kth[q];
    const float bound = key_to_float(kk);
    const bool served = kk < 0xff800000u && big < 60000.f && fabsf(bound) <= a.bound_max;
    if (!served && !a.flat) a.qflag[q] = 1u;  // handed back to the LUT scan (IVF-Flat: everything of it survives, all its rows are re-scored)
    const float t = a.is_ip ? filter_threshold_ip(bound, rn, cn, qc, a) : filter_threshold(bound, rn, a);
    a.thr[s - s_base] = served ? t / a.c1 : (a.flat ? -INFINITY : INFINITY);
  }
}

// ------------------------------------------------------------------ the filter
constexpr int kWThreads = 256;  // 4 waves: one per SIMD
constexpr int kWWaves   = 4;
constexpr int kWS       = 2;    // 32-row subtiles of a wave's strip
constexpr int kWNG      = 3;    // query groups of a unit

struct wide_params {
  const filter_unit* units;
  const uint32_t* n_units;
  uint32_t* xcd_ticket;
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;
  uint32_t n_lists, lbase;
  const uint4* bq;
  const uint32_t* blk_off;  // [n_lists + 1] first operand block of every list (pqw_pair_blocks_kernel)
  const float* thr;
  const uint4* rows16;
  const float* row_term;
  const float* zeros;  // 32 zero floats (the rows' terms of an inner-product search)
  const uint32_t* filter_bits;  // pre-filter (bitset over source ids) - the EMIT build writes -inf for the rows it rejects: the head
  const int64_t* indices;       //   phase's k best rows, and with them the bound, are admissible ones
  uint32_t* qflag;
  uint32_t* fail;  // IVF-Flat: raised when the survivor buffer is full (nullptr: the query is flagged)
  uint2* surv;
  uint32_t* surv_cnt;
  uint32_t surv_cap, spill_cap, n_probes;
  float* xbuf;
  uint32_t ldx, heads;
  unsigned long long* stats;  // optional [8]: [0] pairs screened, [1] survivors, [2] strips, [7] units
};

template <int NSTEPS, bool EMIT>
__global__ __launch_bounds__(kWThreads) void pqw_filter_kernel(const wide_params a)
{
  constexpr int NST    = NSTEPS;  // K steps of 16 dimensions: 4 / 6 / 8 (64 / 96 / 128 dimensions: searches whose k is large against a list), 16 / 24 / 32 / 48
  // K steps of an A-operand chunk: 8 where that gives an even number of chunks, else 4 (384 and 128 dimensions) or 2 (64, 96)
  constexpr int KC     = (NST % 16 == 0) ? 8 : (NST % 8 == 0) ? 4 : 2;
  constexpr int NCHUNK = NST / KC;
  // The ring of A-operand register sets: a chunk is asked for R - 1 chunks before it is multiplied. Three sets (192 registers, 16 K steps
  // ~ 3 k cycles ahead) made the 768-d kernel spill 69 registers - and a reload from scratch is the YOUNGEST entry of the wave's in-order
  // memory queue: waiting for it drains every row load in flight (23 k cycles per strip against 9.2 k of MFMA issue). Two sets: no spill,
  // 19.8 k per strip (with cache-hot rows 14.5 k: a quarter of the strip is still exposed row latency). Chunks of 2 K steps in a ring of
  // 6 or 8 (the depth of three sets for the registers of two) spilled again (9 .. 93 registers: the longer unrolled body) and measured
  // the same 20.5 k.
  constexpr int R   = (NST / KC) % 3 == 0 && NST <= 24 ? 3 : 2;  // register sets of the A-operand ring (96 and 384 dimensions: three sets fit without a spill)
  static_assert(NCHUNK % R == 0, "a chunk's register set must not depend on the strip");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Bs        = reinterpret_cast<uint4*>(smem);                              // [kWNG][NST][64 lanes] x 16 B
  float* s_thr     = reinterpret_cast<float*>(smem + (size_t)kWNG * NST * 1024);  // [kWNG][32] thresholds (EMIT: the pairs' constants)
  uint32_t* s_pair = reinterpret_cast<uint32_t*>(s_thr + kWNG * 32);              // [kWNG][32] pair ids (EMIT: rows of the value buffer)
  uint32_t* ctrl   = s_pair + kWNG * 32;                                          // [0] survivors of this workgroup, [1] current unit
  uint2* my_surv   = a.surv + (size_t)blockIdx.x * a.surv_cap;
  if (threadIdx.x == 0) ctrl[0] = 0u;

  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (scalar: the strips' addresses are)
  const uint32_t ql = lane & 31u, h = lane >> 5;
  const uint32_t n_units = *a.n_units;
  const uint32_t chunk   = (n_units + 7u) / 8u;
  const uint32_t s_base  = a.pair_off[a.lbase];
  uint32_t xcd = blockIdx.x & 7u, hops = 0u;
  unsigned long long st_pairs = 0, st_surv = 0, st_strips = 0, st_units = 0, st_t[2] = {0, 0};

  for (;;) {
    __syncthreads();  // nobody reads the previous unit's operands any more
    if (threadIdx.x == 0) {
      uint32_t ui = 0xffffffffu;
      for (;;) {  // XCD x owns the x-th eighth of the (list-sorted) units; a workgroup whose XCD has run dry moves on to the next share
        // (measured and rejected: the next unit's ticket drawn while this unit runs - the atomic sits ahead of wave 0's row loads in
        // its in-order memory queue: 97 k -> 108 k cycles per unit for the rows, nothing gained in the prologue)
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
    const uint2 uv = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(up) + 16);
    const uint32_t list = __builtin_amdgcn_readfirstlane(uu.x);
    const uint32_t first = __builtin_amdgcn_readfirstlane(uu.y), count = __builtin_amdgcn_readfirstlane(uu.z),
                   row0 = __builtin_amdgcn_readfirstlane(uu.w), base_row = __builtin_amdgcn_readfirstlane(uv.x),
                   r_end = __builtin_amdgcn_readfirstlane(uv.y);
    const uint32_t ng = (count + 31u) >> 5;  // 1 .. kWNG, workgroup-uniform
    const uint32_t u0 = row0 >> 5, u1 = (r_end + 31u) >> 5;
    // strips of this wave: strip t = subtiles u0 + (4 t + wave) kWS .. + kWS - 1; n_mine of them start below u1
    const uint32_t n_sub  = u1 > u0 ? u1 - u0 : 0u;
    const uint32_t w_sub  = wave * kWS;
    const uint32_t n_mine = n_sub > w_sub ? (n_sub - w_sub + kWWaves * kWS - 1u) / (kWWaves * kWS) : 0u;
    const uint4* a_base   = a.rows16 + (size_t)(base_row >> 5) * NST * 64;  // (wave-uniform bases + a lane offset: scalar address arithmetic)
    // (inner product: no row terms - the loads stay, unconditional, on a block of zeros: stride 0)
    const float* t_base   = a.row_term != nullptr ? a.row_term + base_row : a.zeros;
    const uint32_t t_step = a.row_term != nullptr ? 32u : 0u;

    // a chunk of a strip: KC K steps of its kWS subtiles (subtiles past the end repeat the last one and are not screened)
    auto load_chunk = [&](u32x4_t (&av)[KC][kWS], const uint32_t t, const int c) {
#pragma unroll
      for (int s = 0; s < kWS; ++s) {
        const uint32_t uc = min(u0 + (t * kWWaves + wave) * kWS + (uint32_t)s, u1 - 1u);
        const uint4* p = a_base + ((size_t)uc * NST + (size_t)c * KC) * 64;
#pragma unroll
        for (int st = 0; st < KC; ++st) {
          const uint4 v = p[st * 64 + lane];
          av[st][s] = u32x4_t{v.x, v.y, v.z, v.w};
        }
      }
    };
    // accumulator register i of a lane is row (i & 3) + 8 (i >> 2) + 4 h of the subtile: its initial value is that row's term
    auto load_terms = [&](f32x16_t (&tv)[kWS], const uint32_t t) {
#pragma unroll
      for (int s = 0; s < kWS; ++s) {
        const uint32_t uc = min(u0 + (t * kWWaves + wave) * kWS + (uint32_t)s, u1 - 1u);
        const float* p = t_base + (size_t)uc * t_step;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(p + 8 * j + 4u * h);
          tv[s][4 * j] = v.x; tv[s][4 * j + 1] = v.y; tv[s][4 * j + 2] = v.z; tv[s][4 * j + 3] = v.w;
        }
      }
    };
    const unsigned long long t_unit = a.stats != nullptr ? __builtin_readcyclecounter() : 0ull;
    u32x4_t av[R][KC][kWS];
    f32x16_t tv[kWS];
    if (n_mine != 0u) {  // the first chunks of the wave's first strip are on their way while the unit's operands are copied
      load_terms(tv, 0u);
#pragma unroll
      for (int r = 0; r < R - 1; ++r) load_chunk(av[r], 0u, r);
    }
    // ---- B operands of the unit's query groups: whole blocks of the pre-pass's layout (NST KiB each, one after the other), copied as
    // 1 KiB rows; then thresholds and pair ids
    {
      const uint4* src = a.bq + ((size_t)a.blk_off[list] + ((first - a.pair_off[a.lbase + list]) >> 5)) * NST * 64 + lane;
      // (measured and rejected: all 36 rows of a wave's share asked for before the first is stored - 144 more registers next to the
      // first chunks' prefetch: 124 spilled, the prologue 45 k -> 80 k cycles)
#pragma unroll 12
      for (uint32_t it = wave; it < ng * NST; it += kWWaves) Bs[it * 64 + lane] = src[(size_t)it * 64];
    }
    if (threadIdx.x < kWNG * 32) {
      const uint32_t jj = threadIdx.x;
      const uint32_t jc = min(jj, count - 1u);
      const uint32_t p  = a.sorted_pairs[first + jc];
      if constexpr (EMIT) {
        s_thr[jj]  = a.thr[first - s_base + jc];
        s_pair[jj] = (p / a.n_probes) * a.heads + p % a.n_probes;
      } else {
        s_thr[jj]  = jj < count ? a.thr[first - s_base + jc] : INFINITY;  // (a padding slot keeps nothing)
        s_pair[jj] = p;
      }
    }
    __syncthreads();
    const unsigned long long t_loop = a.stats != nullptr ? __builtin_readcyclecounter() : 0ull;

    auto run = [&](auto ng_tag) {
      constexpr int NG = decltype(ng_tag)::value;
      float thr[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) thr[g] = s_thr[g * 32 + ql];
      f16x8_t bc[NG], bn[NG];
      auto load_b = [&](f16x8_t (&b)[NG], const int ks) {
        const uint4* bp = Bs + (size_t)ks * 64 + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const uint4 v = bp[(size_t)g * NST * 64];
          b[g] = __builtin_bit_cast(f16x8_t, v);
        }
      };
      load_b(bc, 0);
      for (uint32_t t = 0; t < n_mine; ++t) {
        const uint32_t tn = min(t + 1u, n_mine - 1u);
        f32x16_t acc[kWS][NG];
#pragma unroll
        for (int s = 0; s < kWS; ++s)
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[s][g] = tv[s];
        __builtin_amdgcn_sched_barrier(0);
        load_terms(tv, tn);  // the next strip's
        __builtin_amdgcn_sched_barrier(0);
        // program order is the schedule (sched_barrier): the compiler moved the chunk loads down to where their registers come
        // free - eight loads, a few hundred cycles, ahead of their use instead of two chunks
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
          const int cp = c + R - 1;  // the chunk asked for now: two (one) chunks ahead, of this strip or the wave's next
          if (cp < NCHUNK) load_chunk(av[cp % R], t, cp); else load_chunk(av[cp % R], tn, cp - NCHUNK);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int st = 0; st < KC; ++st) {
            // B operands of the NEXT K step are read while this step's MFMAs run (the step after the strip's last one: the first again)
            const int kn = (c * KC + st + 1) % NST;
            load_b(bn, kn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
              for (int s = 0; s < kWS; ++s)
                acc[s][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, av[c % R][st][s]), bc[g], acc[s][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) bc[g] = bn[g];
          }
        }
        if (a.stats != nullptr) st_strips += 1u;
        // ---- epilogue of the strip
#pragma unroll
        for (int s = 0; s < kWS; ++s) {
          const uint32_t u = u0 + (t * kWWaves + wave) * kWS + (uint32_t)s;
          if (u >= u1) continue;  // wave-uniform: a repeat of the last subtile
          uint32_t adm = 0xffffffffu;  // EMIT: bit v = row v of the subtile passes the pre-filter
          if constexpr (EMIT) {
            if (a.filter_bits != nullptr) {  // wave-uniform
              const uint32_t rr = (u << 5) + ql;
              bool okr = false;
              if (h == 0u && rr < r_end) {
                const int64_t sid = a.indices[base_row + rr];
                okr = ((a.filter_bits[sid >> 5] >> (sid & 31)) & 1u) != 0u;
              }
              adm = (uint32_t)__ballot(okr);  // (lanes 0 .. 31 are K half 0)
            }
          }
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const f32x16_t& ac = acc[s][g];
            if constexpr (EMIT) {
              // registers 4 j .. 4 j + 3 of a lane are the rows 8 j + 4 h .. + 3 of the subtile: one 16-byte store each
              const uint32_t jj = g * 32u + ql;
              if (jj < count) {
                const float cadd = thr[g];
                float* xp = a.xbuf + (size_t)s_pair[jj] * a.ldx + (u << 5) + 4u * h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const uint32_t v0 = (u << 5) + 8u * j + 4u * h;
                  float4 o;
                  const uint32_t am = adm >> (8u * j + 4u * h);  // rows 8 j + 4 h .. + 3 of the subtile
                  o.x = (v0 + 0u < r_end && (am & 1u)) ? ac[4 * j + 0] + cadd : -INFINITY; o.y = (v0 + 1u < r_end && (am & 2u)) ? ac[4 * j + 1] + cadd : -INFINITY;
                  o.z = (v0 + 2u < r_end && (am & 4u)) ? ac[4 * j + 2] + cadd : -INFINITY; o.w = (v0 + 3u < r_end && (am & 8u)) ? ac[4 * j + 3] + cadd : -INFINITY;
                  *reinterpret_cast<float4*>(xp + 8 * j) = o;
                }
              }
              continue;
            }
            float m = fmaxf(fmaxf(ac[0], ac[1]), ac[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, ac[i]), ac[i + 1]);
            m = fmaxf(m, ac[15]);
            const float thg = thr[g];
            const bool any  = m >= thg;
            if (a.stats != nullptr) st_pairs += 32u * min(32u, count - g * 32u);
            if (__ballot(any) == 0ull) continue;
            // ---- survivors: the lanes that hold some count them (a 16-bit mask), positions under scalar control, ONE LDS atomic
            uint32_t hits = 0u;
            if (any) {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const uint32_t v = (u << 5) + (uint32_t)((i & 3) + 8 * (i >> 2)) + 4u * h;
                hits |= (ac[i] >= thg && v < r_end) ? (1u << i) : 0u;
              }
            }
            const uint32_t n_hit = (uint32_t)__popc(hits);
            unsigned long long lm = __ballot(hits != 0u);
            uint32_t total = 0u, my_off = 0u;
            while (lm != 0ull) {
              const uint32_t src = (uint32_t)__ffsll((long long)lm) - 1u;
              lm &= lm - 1ull;
              if (lane == src) my_off = total;
              total += (uint32_t)__builtin_amdgcn_readlane(n_hit, src);
            }
            if (total == 0u) continue;  // (every hit lay past the chunk's end)
            uint32_t base = 0u;
            if (lane == 0u) base = atomicAdd(&ctrl[0], total);  // LDS
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t pid = s_pair[g * 32 + ql];
            uint32_t pos = base + my_off;
            uint32_t hb = hits;
            while (hb != 0u) {
              const uint32_t i = (uint32_t)__ffs((int)hb) - 1u;
              hb &= hb - 1u;
              const uint32_t v = (u << 5) + (i & 3u) + 8u * (i >> 2) + 4u * h;
              if (pos < a.surv_cap) {
                my_surv[pos] = make_uint2(pid, base_row + v);
              } else {  // this workgroup's region is full: the shared spill region; when that is full too the query goes back to the LUT scan
                const uint32_t sp = atomicAdd(a.surv_cnt + gridDim.x, 1u);
                if (sp < a.spill_cap) a.surv[(size_t)gridDim.x * a.surv_cap + sp] = make_uint2(pid, base_row + v);
                else if (a.fail != nullptr) *a.fail = 1u;
                else a.qflag[pid / a.n_probes] = 1u;
              }
              ++pos;
            }
            if (a.stats != nullptr) st_surv += n_hit;
          }
        }
      }
    };
    if (ng >= 3u) run(std::integral_constant<int, 3>{});
    else if (ng == 2u) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
    if (a.stats != nullptr) {
      st_units += 1u;
      const unsigned long long t_end = __builtin_readcyclecounter();
      st_t[0] += t_loop - t_unit;  // unit prologue (ticket, B operands into LDS)
      st_t[1] += t_end - t_loop;   // strips
    }
  }
  if (a.stats != nullptr) atomicAdd(&a.stats[1], st_surv);  // counted per lane
  if (a.stats != nullptr && lane == 0) {
    atomicAdd(&a.stats[0], st_pairs); atomicAdd(&a.stats[2], st_strips);
    if (wave == 0) { atomicAdd(&a.stats[7], st_units); atomicAdd(&a.stats[4], st_t[0]); atomicAdd(&a.stats[5], st_t[1]); }
  }
  __syncthreads();
  if (!EMIT && threadIdx.x == 0) a.surv_cnt[blockIdx.x] = min(ctrl[0], a.surv_cap);
}

}  // namespace

bool pqw_shape(uint32_t rot_dim)
{
  return rot_dim == 64u || rot_dim == 96u || rot_dim == 128u || rot_dim == 256u || rot_dim == 384u || rot_dim == 512u || rot_dim == 768u;
}

uint32_t pqw_group() { return 32u * kWNG; }

void pqw_decode(resources& res, const uint8_t* codes8, uint32_t n_chunks, const uint32_t* cb16, uint32_t pq_len, int64_t padded_rows,
                uint32_t rot_dim, void* rows16)
{
  const uint32_t nst     = rot_dim / 16u;
  const int64_t n_pieces = padded_rows / 32 * nst * 64;
  if (n_pieces == 0) return;
  hipLaunchKernelGGL(pqw_decode_kernel, dim3((unsigned)grid_blocks(n_pieces, 256)), dim3(256), 0, res.stream, codes8, n_chunks,
                     reinterpret_cast<const uint16_t*>(cb16), pq_len, n_pieces, nst, static_cast<uint4*>(rows16));
  HIP_TRY(hipGetLastError());
}

void pqw_bprep(resources& res, const wide_prep& l)
{
  wprep_params b{};
  b.sorted_pairs = l.sorted_pairs; b.pair_off = l.pair_off; b.n_lists = l.n_lists; b.lbase = l.head ? 0u : l.n_lists; b.probes = l.probes;
  b.rot_queries = l.rot_queries; b.centers_rot = l.centers_rot; b.query_kth = l.query_kth; b.qflag = l.qflag;
  b.bq = static_cast<uint4*>(l.bq); b.thr = l.thr; b.norms = static_cast<float4*>(l.norms); b.n_probes = l.n_probes; b.rot_dim = l.rot_dim;
  b.blk_off = l.blk_off; b.is_ip = l.is_ip; b.flat = l.flat;
  b.heads = l.heads; b.sc = l.sc; b.c1 = l.c1; b.eps = l.eps; b.alpha = l.alpha; b.cbmax = l.cbmax; b.dmax = l.dmax; b.bound_max = l.bound_max;
  b.head = l.head;
  if (l.n_pairs == 0) return;
  hipLaunchKernelGGL(pqw_pair_blocks_kernel, dim3(1), dim3(1024), 0, res.stream, l.pair_off, l.n_lists, b.lbase, l.blk_off);
  // (grid: an upper bound of the blocks - every list's last block may be a partial one)
  hipLaunchKernelGGL(pqw_bprep_kernel, dim3((unsigned)grid_blocks(l.n_pairs / 32 + l.n_lists + 1, 4)), dim3(256), 0, res.stream, b);
  HIP_TRY(hipGetLastError());
}

void pqw_filter(resources& res, const wide_filter& l)
{
  wide_params g{};
  g.units = l.units; g.n_units = l.n_units; g.xcd_ticket = l.xcd_ticket; g.sorted_pairs = l.sorted_pairs; g.pair_off = l.pair_off;
  g.n_lists = l.n_lists; g.lbase = l.emit ? 0u : l.n_lists; g.bq = static_cast<const uint4*>(l.bq); g.thr = l.thr; g.blk_off = l.blk_off;
  g.rows16 = static_cast<const uint4*>(l.rows16); g.row_term = l.row_term; g.zeros = l.zeros; g.filter_bits = l.filter_bits; g.indices = l.indices; g.qflag = l.qflag; g.fail = l.fail; g.surv = static_cast<uint2*>(l.surv);
  g.surv_cnt = l.surv_cnt; g.surv_cap = l.surv_cap; g.spill_cap = l.spill_cap; g.n_probes = l.n_probes; g.xbuf = l.xbuf; g.ldx = l.ldx;
  g.heads = l.heads; g.stats = l.stats;
  const uint32_t nst = l.rot_dim / 16u;
  const size_t fsmem = (size_t)kWNG * nst * 1024 + 2 * kWNG * 32 * 4 + 16;
  auto launch = [&](auto kern) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem));
    profile_begin(res, l.profile_name != nullptr ? l.profile_name : "pq_filter_kernel");
    hipLaunchKernelGGL(kern, dim3(l.grid), dim3(kWThreads), fsmem, res.stream, g);
    profile_end(res, l.profile_name != nullptr ? l.profile_name : "pq_filter_kernel");
  };
  CUVS_EXPECTS(pqw_shape(l.rot_dim), "ivf_pq: rot_dim %u is outside the wide matrix-core filter", l.rot_dim);
  auto pick = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    if (l.emit) launch(pqw_filter_kernel<N, true>); else launch(pqw_filter_kernel<N, false>);
  };
  switch (nst) {
    case 4:  pick(std::integral_constant<int, 4>{}); break;
    case 6:  pick(std::integral_constant<int, 6>{}); break;
    case 8:  pick(std::integral_constant<int, 8>{}); break;
    case 16: pick(std::integral_constant<int, 16>{}); break;
    case 24: pick(std::integral_constant<int, 24>{}); break;
    case 32: pick(std::integral_constant<int, 32>{}); break;
    default: pick(std::integral_constant<int, 48>{}); break;
  }
  HIP_TRY(hipGetLastError());
}

}  // namespace cuvs_amd
