// IVF-PQ search, warm-bounds phase on the matrix cores (see ivf_pq_scan3.hip for the scheme): the filter kernel of
// round 4 and its pre-pass. Reference semantics: compute_score_impl.cuh:52-79 / ivf_pq_search.cuh:421-669 - the filter
// only decides which (row, query) pairs the reference's arithmetic has to look at.
#include "ivf_pq_filter_common.hpp"

#include <cfloat>
#include <type_traits>

namespace cuvs_amd {

namespace {

// ================================================================== round 4: the filter with ONE wave per SIMD
// What the two-waves-per-SIMD kernel above left on the table (profiles/r03_*): a list probed by 65 .. 128 queries decoded
// its rows twice (64-query units), the unit prologue (B operands from fp32 residuals, thresholds in double: ~23 k cycles
// per unit and wave) was 10 % of the kernel, the K-extension step was a ninth of the matrix work, and 256 registers per
// wave did not hold the 64 B-operand registers + double-buffered A operands without spilling (51 VGPRs). Here:
//   * a pre-pass (pq_bprep_kernel) writes, for every tail pair, the fp16 B operand in MFMA layout (256 B at rot_dim 128)
//     and the threshold in accumulator units: the filter's prologue is 32 16-byte loads per lane;
//   * a work unit holds up to FOUR groups of 32 queries (128 B-operand registers): the rows of a list chunk are fetched
//     and decoded once for up to 128 probing queries - half the LDS gathers, VALU decode work and code fetches per pair;
//   * the row term -|d|^2 (1 - 2^-9) sc^2 / 2 enters as the INITIAL VALUE of the accumulators (fp32, four 16-byte loads per
//     subtile, shared by the query groups) instead of through a ninth MFMA step;
//   * one 256-thread workgroup per CU = one wave per SIMD with up to 512 registers; the overlap that two waves per SIMD
//     provided by chance is written into the program order: the gathers that decode subtile u + 1 are spread over the K
//     steps of subtile u's MFMAs, and the accumulators of subtile u - 1 are screened while those of u are being computed
//     (two accumulator sets), so the matrix pipe never waits for a screen.
// Results are those of the kernel above: the same threshold, a superset test (the row term is more precise in fp32).
struct bprep_params {
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;  // [2 n_lists + 1]: the tail pairs are sorted_pairs[pair_off[n_lists] .. pair_off[2 n_lists])
  uint32_t n_lists;
  uint32_t lbase;          // first label of the pairs served: n_lists (tail pairs) or 0 (head pairs: IVF-Flat's bound-only head phase)
  const uint32_t* probes;  // [n_pairs] list of every pair
  const float* rot_queries;
  const float* centers_rot;
  const uint32_t* query_kth;
  uint32_t* qflag;
  uint4* bq;   // [tail pair][K step][K half] x 16 bytes: the pair's scaled fp16 residual as the filter's B operand
  float* thr;  // [tail pair] threshold in accumulator units (+inf: nothing survives, -inf: everything does)
  uint32_t n_probes, rot_dim;
  float sc, c1, eps, alpha, cbmax, dmax, bound_max;
  int is_ip, flat;
  uint32_t lpl;  // log2(pq_len)
  float4* norms; // PART 1 / pq_thr_kernel: [tail pair] (|r|^2, |c|^2, q.c, largest scaled operand)
};

// One wave per 32 consecutive tail pairs, LPP lanes per pair - lane slot = (K step st, K half h) = one 16-byte piece of the
// pair's B operand in the layout of v_mfma_f32_32x32x16_f16 - so a pair's 2 NST pieces leave the wave as one contiguous run
// (round 4: lane = (pair, K half), every store instruction wrote 32-byte pieces 256 bytes apart and every load pulled 16-byte
// pieces of 64 different rows: 0.20 ms for a 325 MB stream) and its rotated query / list centre are read as whole rows.
// K step st = c * pq_len + t of the filter holds, in K half h, the 8 rotated dimensions from pq_len (16 c + 8 h) + 8 t on:
// the components of the subspaces whose code bytes are the h-th 8 bytes of the row's 16-byte code chunk c, in order.
// PART 0: B operands and thresholds (the head phase's bounds are known). PART 1: B operands and the norms the thresholds
// need - this part does not depend on the head phase and runs next to it on a helper stream; pq_thr_kernel finishes.
template <int NST, int PART>
__global__ __launch_bounds__(256) void pq_bprep_kernel(const bprep_params a)
{
  constexpr uint32_t LPP = 2 * NST <= 2 ? 2u : 2 * NST <= 4 ? 4u : 2 * NST <= 8 ? 8u : 2 * NST <= 16 ? 16u : 32u;  // lanes per pair
  constexpr uint32_t PPW = 64u / LPP;                                                                            // pairs per pass
  const uint32_t s_base = a.pair_off[a.lbase], s_end = a.pair_off[a.lbase + a.n_lists];
  const uint32_t lane = threadIdx.x & 63u, slot = lane % LPP, st = slot >> 1, h = slot & 1u;
  const uint32_t w0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 32u;
  if (s_base + w0 >= s_end) return;  // wave-uniform
  const uint32_t d0 = ((16u * (st >> a.lpl) + 8u * h) << a.lpl) + 8u * (st & ((1u << a.lpl) - 1u));
#pragma unroll 2
  for (uint32_t pass = 0; pass < 32u / PPW; ++pass) {
    const uint32_t s  = s_base + w0 + pass * PPW + lane / LPP;
    const bool pair_ok = s < s_end;
    const bool valid  = pair_ok && slot < 2u * NST;
    const uint32_t p  = a.sorted_pairs[pair_ok ? s : s_end - 1u];
    const uint32_t q  = p / a.n_probes, L = a.probes[p];
    float rn = 0.f, big = 0.f, qc = 0.f, cn = 0.f;
    if (valid) {
      const float* rq = a.rot_queries + (size_t)q * a.rot_dim + d0;
      const float* ct = a.centers_rot + (size_t)L * a.rot_dim + d0;
      const float4 q0 = *reinterpret_cast<const float4*>(rq), q1 = *reinterpret_cast<const float4*>(rq + 4);
      float r[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      const float4 c0 = *reinterpret_cast<const float4*>(ct), c1 = *reinterpret_cast<const float4*>(ct + 4);
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
      a.bq[(size_t)(s - s_base) * NST * 2 + slot] = __builtin_bit_cast(uint4, v);
    }
    // the pair's norms: sums / maximum over its LPP lanes (idle lanes hold zeros)
#pragma unroll
    for (uint32_t o = 1; o < LPP; o <<= 1) {
      rn += __shfl_xor(rn, o); qc += __shfl_xor(qc, o); cn += __shfl_xor(cn, o);
      big = fmaxf(big, __shfl_xor(big, o));
    }
    if (!pair_ok || slot != 0u) continue;
    if constexpr (PART == 1) {
      a.norms[s - s_base] = make_float4(rn, cn, qc, big);
    } else {
      const uint32_t kk = a.query_kth[q];
      const float bound = key_to_float(kk);
      // no finite bound yet, an operand beyond the fp16 range or a bound the LUT type cannot represent: nothing of this query
      // survives here, it is handed back to the LUT scan (IVF-Flat: everything survives, all its rows are re-scored)
      const bool served = kk < 0xff800000u && big < 60000.f && fabsf(bound) <= a.bound_max;
      if (!a.flat && !served) a.qflag[q] = 1u;
      const float t = a.is_ip ? filter_threshold_ip(bound, rn, cn, qc, a) : filter_threshold(bound, rn, a);
      a.thr[s - s_base] = served ? t / a.c1 : (a.flat ? -INFINITY : INFINITY);
    }
  }
}

// the second half of the pre-pass (two-stream schedule): thresholds of the tail pairs from the head phase's bounds
__global__ __launch_bounds__(256) void pq_thr_kernel(const bprep_params a)
{
  const uint32_t s_base = a.pair_off[a.lbase], s_end = a.pair_off[a.lbase + a.n_lists];
  const uint32_t s = s_base + blockIdx.x * 256u + threadIdx.x;
  if (s >= s_end) return;
  const uint32_t q  = a.sorted_pairs[s] / a.n_probes;
  const float4 nm   = a.norms[s - s_base];
  const float rn = nm.x, cn = nm.y, qc = nm.z, big = nm.w;
  const uint32_t kk = a.query_kth[q];
  const float bound = key_to_float(kk);
  const bool served = kk < 0xff800000u && big < 60000.f && fabsf(bound) <= a.bound_max;
  if (!a.flat && !served) a.qflag[q] = 1u;
  const float t = a.is_ip ? filter_threshold_ip(bound, rn, cn, qc, a) : filter_threshold(bound, rn, a);
  a.thr[s - s_base] = served ? t / a.c1 : (a.flat ? -INFINITY : INFINITY);
}

constexpr int kF4Threads = 256;  // 4 waves: one per SIMD, up to 512 registers each
constexpr uint32_t kSurvChunk = 256u;

struct filter4_params {
  const filter_unit* units;
  const uint32_t* n_units;  // device scalar
  uint32_t* xcd_ticket;     // 8 counters, 32 words apart
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;
  uint32_t n_lists;
  const uint4* bq;
  const float* thr;
  const uint32_t* cb16;
  const uint8_t* codes;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  const float* row_term;  // [padded_rows] -|d|^2 (1 - 2^-9) sc^2 / 2 (L2), nullptr for inner product
  uint32_t* qflag;
  uint2* surv;         // n_chunks x kSurvChunk entries
  uint32_t* surv_cnt;  // [1] chunks handed out so far (may run past n_chunks)
  uint32_t n_chunks, n_probes, unit_rows;
  unsigned long long* stats;  // optional [8] as in pq_filter_kernel
};

// LDS byte address of a decode-table entry in ONE instruction: the low word of `addr` becomes (byte BYTE of w) << two (the
// shift = log2 of an entry's bytes), its high word - the lane's K half: the two halves' tables lie 64 KiB apart - stays (v_bfe_u32 + v_lshl_add_u32 otherwise:
// with one wave per SIMD every instruction of the subtile loop is an issue slot the matrix pipe waits behind)
template <int BYTE>
__device__ inline void table_addr(uint32_t& addr, const uint32_t two, const uint32_t w)
{
  if constexpr (BYTE == 0)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0" : "+v"(addr) : "v"(two), "v"(w));
  else if constexpr (BYTE == 1)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1" : "+v"(addr) : "v"(two), "v"(w));
  else if constexpr (BYTE == 2)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(addr) : "v"(two), "v"(w));
  else
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3" : "+v"(addr) : "v"(two), "v"(w));
}

typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;

// TERM: L2 (the rows' terms are the accumulators' initial values); inner product starts from zero
// DBG: ablations (timing only, wrong results): 1 conflict-free gathers, 2 no gathers, 4 no MFMAs, 8 no code loads
// STATS: the counters / cycle stamps of CUVS_AMD_SCAN_DEBUG=1024 (s_memtime is a scalar memory operation: in the subtile
// loop it would make every wait on the gathers a wait for everything)
// PL: pq_len (1, 2, 4, 8): a code byte stands for PL fp16 values - the lane's 8 K elements of an MFMA step are 8 / PL
// codebook entries of 2 PL bytes; the 8 code bytes a lane holds of a chunk feed PL consecutive K steps
// PC: PER_CLUSTER codebooks - one decode table of 256 entries per LIST, shared by its subspaces: every wave keeps the table
// of its current unit's list in a 4 KiB LDS region of its own (loaded at the start of a unit, 32 PL 16-byte loads per wave)
template <int NCH, int PL, bool TERM, int DBG, bool STATS = false, bool PC = false>
__global__ __launch_bounds__(kF4Threads) void pq_filter4_kernel(const filter4_params a)
{
  constexpr int NST = PL * NCH;          // MFMA K steps
  constexpr int NGM = NST <= 8 ? 4 : 2;  // groups of 32 queries per work unit
  constexpr int NGA = 8 / PL;            // gathers (and table addresses) per K step
  constexpr uint32_t kEntry = 2u * PL;   // bytes of a decode-table entry
  constexpr uint32_t kSlot  = 256u * kEntry;  // bytes of a subspace's table
  // LDS: K half 0 of every 16-subspace chunk at [0, NCH * 8 slots), K half 1 at 64 KiB + the same (table_addr)
  constexpr uint32_t kHalf1 = 65536u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // survivors: the buffer is handed out in chunks of kSurvChunk entries - a wave draws a chunk with ONE global atomic and
  // fills it by itself, position in a scalar register (an LDS counter's returning atomic would sit behind the wave's
  // gathers in the LDS queue, which returns in order: a full drain per survivor; fixed per-wave regions run over on the
  // waves whose lists hold the dense regions of the corpus). A chunk's unused tail is padded with invalid entries.
  uint32_t s_chunk = 0xffffffffu, s_fill = kSurvChunk;  // wave-uniform: current chunk, entries written to it
  for (uint32_t i = threadIdx.x; i < (PC ? 0u : (uint32_t)NCH * 16u * (kSlot / 16u)); i += kF4Threads) {
    // a.cb16: [subspace s][256 codes][PL]; subspace s = 16 c + 8 half + j goes to half's table at slot 8 c + j
    const uint32_t s = i / (kSlot / 16u), half = (s >> 3) & 1u, slot = (s >> 4) * 8u + (s & 7u);
    *reinterpret_cast<uint4*>(smem + half * kHalf1 + slot * kSlot + (i % (kSlot / 16u)) * 16u) = reinterpret_cast<const uint4*>(a.cb16)[i];
  }
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u, ql = lane & 31u, h = lane >> 5;
  const uint32_t n_units = *a.n_units;
  const uint32_t chunk   = (n_units + 7u) / 8u;
  const uint32_t s_base  = a.pair_off[a.n_lists];
  uint32_t xcd = blockIdx.x & 7u, hops = 0u;
  const uint32_t two = PL == 1 ? 1u : PL == 2 ? 2u : PL == 4 ? 3u : 4u;  // log2(kEntry)
  const uint32_t lane_code_off = ql * 16u + h * 8u, lane_term_off = h * 16u;  // byte offsets of this lane inside a subtile

  unsigned long long st_pairs = 0, st_surv = 0, st_sub = 0, st_slow = 0, st_units = 0, st_t[3] = {0, 0, 0};
  // ---- unit pipeline: the ticket of the NEXT unit is drawn when a unit starts and its descriptor is loaded a few
  // subtiles later, so that a unit's prologue is ONE memory round trip (B operands, codes and terms together) instead
  // of a chain of five (ticket -> descriptor -> list offsets -> operands -> codes)
  uint32_t tk_v = 0u;  // lane 0: the pending ticket
  uint4 d0 = {0u, 0u, 0u, 0u};
  uint2 d1 = {0u, 0u};
  bool desc_ok = false;
  auto draw_ticket = [&]() {
    if (lane == 0) tk_v = atomicAdd(a.xcd_ticket + xcd * 32, 1u);
  };
  auto load_desc = [&]() {  // resolves the pending ticket; false: this XCD's share has run dry
    const uint32_t share0 = min(n_units, xcd * chunk), share_len = min(chunk, n_units - share0);
    const uint32_t t = __builtin_amdgcn_readfirstlane(tk_v);
    desc_ok = t < share_len;
    if (desc_ok) {
      const filter_unit* up = a.units + share0 + t;
      d0 = *reinterpret_cast<const uint4*>(up);
      d1 = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(up) + 16);
    }
  };
  draw_ticket();
  load_desc();
  for (;;) {
    while (!desc_ok) {  // a wave whose XCD has run dry moves on to the next XCD's share (equal in units, not in work)
      if (++hops == 8u) break;
      xcd = (xcd + 1u) & 7u;
      draw_ticket();
      load_desc();
    }
    if (!desc_ok) break;
    if constexpr (PC) {  // this list's decode table into the wave's region (LDS operations of a wave complete in order)
      const uint32_t list = __builtin_amdgcn_readfirstlane(d0.x);
      const uint4* src    = reinterpret_cast<const uint4*>(a.cb16) + (size_t)list * (kSlot / 16u);
      char* dst           = smem + (threadIdx.x >> 6) * 4096u;
#pragma unroll
      for (uint32_t i = 0; i < (kSlot / 16u + 63u) / 64u; ++i)
        if (i * 64u + lane < kSlot / 16u) *reinterpret_cast<uint4*>(dst + (i * 64u + lane) * 16u) = src[i * 64u + lane];
    }
    const uint32_t first = __builtin_amdgcn_readfirstlane(d0.y), count = __builtin_amdgcn_readfirstlane(d0.z),
                   row0 = __builtin_amdgcn_readfirstlane(d0.w), base_row = __builtin_amdgcn_readfirstlane(d1.x),
                   r_end = __builtin_amdgcn_readfirstlane(d1.y);
    const uint32_t u0 = row0 >> 5, u1 = (r_end + 31u) >> 5;
    const unsigned long long t_unit = STATS ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_loop = 0ull;
    draw_ticket();  // the next unit's (resolved a few subtiles into this one)
    bool desc_pending = true;
    // wave-uniform bases (lists start at multiples of 64 rows): subtile u of the list is half (u & 1) of 64-row group u / 2
    const char* code_base = reinterpret_cast<const char*>(a.codes) + (size_t)(base_row >> 6) * NCH * 1024;
    const char* term_base = reinterpret_cast<const char*>(a.row_term + base_row);

    auto load_codes = [&](const uint32_t u, uint2 (&cw)[NCH]) {
      uint32_t uc = min(u, u1 - 1u);  // (scalar; the subtiles past the end repeat the last one)
      if constexpr ((DBG & 64) != 0) uc = u0;  // ablation: cache-hot code loads
      const char* p = code_base + ((size_t)(uc >> 1) * NCH * 1024 + (uc & 1u) * 512u) + lane_code_off;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if constexpr ((DBG & 8) != 0) cw[c] = make_uint2(u * 2654435761u + c, lane);  // ablation: no code loads
        else cw[c] = *reinterpret_cast<const uint2*>(p + c * 1024);
      }
    };
    // the gathers of K step st = c * PL + t: lane (row, half) looks up the codebook entries of 8 / PL of its row's codes -
    // bytes t * 8 / PL .. of the 8 it holds of chunk c
    uint32_t ad[NGA];  // table addresses: high word = K half, low word rewritten per gather
#pragma unroll
    for (int e = 0; e < NGA; ++e) ad[e] = h << 16;
    auto decode_addr = [&](const uint2 (&cw)[NCH], const int st) {
      const int c = st / PL, b0 = (st % PL) * NGA;  // first code byte of the step
      if constexpr (PC) {
        const uint32_t base = (threadIdx.x >> 6) * 4096u;
#pragma unroll
        for (int e = 0; e < NGA; ++e) {
          const int b = b0 + e;
          const uint32_t w = b < 4 ? cw[c].x : cw[c].y;
          ad[e] = base + (((w >> (8 * (b & 3))) & 255u) << two);
        }
        return;
      }
      if constexpr ((DBG & 1) != 0) {  // ablation: every lane of a half in its own bank
        const uint32_t w = ql * 0x01010101u;
        if constexpr (NGA >= 1) table_addr<0>(ad[0], two, w);
        if constexpr (NGA >= 2) table_addr<1>(ad[1], two, w);
        if constexpr (NGA >= 4) { table_addr<2>(ad[2], two, w); table_addr<3>(ad[3], two, w); }
        return;
      }
      auto one = [&](const int e) {  // byte b0 + e of the lane's 8
        const int b = b0 + e;
        const uint32_t w = b < 4 ? cw[c].x : cw[c].y;
        switch (b & 3) {
          case 0: table_addr<0>(ad[e], two, w); break;
          case 1: table_addr<1>(ad[e], two, w); break;
          case 2: table_addr<2>(ad[e], two, w); break;
          default: table_addr<3>(ad[e], two, w); break;
        }
      };
#pragma unroll
      for (int e = 0; e < NGA; ++e) one(e);
    };
    auto decode_gather = [&](const int st, u32x4_t (&av)[NST]) {
      constexpr uint32_t kNone = 0u;
      const uint32_t slot0 = PC ? 0u : (8u * (st / PL) + (uint32_t)((st % PL) * NGA)) * kSlot;
      constexpr uint32_t kStep = PC ? 0u : kSlot;  // (PER_CLUSTER: every subspace reads the same table)
      if constexpr ((DBG & 2) != 0) {  // ablation: no gathers
#pragma unroll
        for (int e = 0; e < 4; ++e) av[st][e] = ad[e % NGA] + kNone;
        return;
      }
      if constexpr (PL == 1) {
        // eight 2-byte entries: ds_read_u16_d16 / _d16_hi fill the two halves of a register
        typedef __attribute__((address_space(3))) const _Float16 lds_f16_t;
        f16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<lds_f16_t*>(ad[e] + slot0 + e * kStep);
        av[st] = __builtin_bit_cast(u32x4_t, v);
      } else if constexpr (PL == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[st][e] = *reinterpret_cast<lds_u32_t*>(ad[e] + slot0 + e * kStep);
      } else if constexpr (PL == 4) {
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const u32x2_t lds_u64_t;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const u32x2_t v = *reinterpret_cast<lds_u64_t*>(ad[e] + slot0 + e * kStep);
          av[st][2 * e] = v[0]; av[st][2 * e + 1] = v[1];
        }
      } else {
        typedef __attribute__((address_space(3))) const u32x4_t lds_u128_t;
        av[st] = *reinterpret_cast<lds_u128_t*>(ad[0] + slot0);
      }
    };
    auto decode_step = [&](const uint2 (&cw)[NCH], const int st, u32x4_t (&av)[NST]) {
      decode_addr(cw, st);
      decode_gather(st, av);
    };
    // accumulator register i of a lane is row (i & 3) + 8 (i >> 2) + 4 h of the subtile: its initial value is that row's term
    auto load_term = [&](const uint32_t u, f32x16_t& tv) {
      if constexpr (!TERM) { tv = f32x16_t{}; return; }
      uint32_t uc = min(u, u1 - 1u);
      if constexpr ((DBG & 16) != 0) uc = u0;  // ablation: cache-hot term loads
      const char* p = term_base + (size_t)uc * 128u + lane_term_off;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(p + j * 32);
        tv[4 * j] = v.x; tv[4 * j + 1] = v.y; tv[4 * j + 2] = v.z; tv[4 * j + 3] = v.w;
      }
    };

    auto run = [&](auto ng_tag) {
      constexpr int NG = decltype(ng_tag)::value;
      // ---- B operands, thresholds and pair ids of the unit's queries: lane = (query ql of group g, K half h)
      f16x8_t bop[NG][NST];
      float thr[NG];
      uint32_t pairid[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const uint32_t jj  = g * 32u + ql;
        const uint32_t jc  = min(jj, count - 1u);
        const uint4* bp    = a.bq + ((size_t)(first - s_base + jc) * NST * 2 + h);
#pragma unroll
        for (int st = 0; st < NST; ++st) bop[g][st] = __builtin_bit_cast(f16x8_t, bp[st * 2]);
        thr[g]    = jj < count ? a.thr[first - s_base + jc] : INFINITY;
        pairid[g] = a.sorted_pairs[first + jc];
      }
      // code words: a ring of four subtiles (slot k % 4 holds subtile u0 + k), loaded four subtiles ahead of their decode;
      // decoded rows, row terms and accumulators: two sets whose roles alternate from subtile to subtile
      uint2 cw[4][NCH];
      u32x4_t av[2][NST];
      f32x16_t tv[2], acc[2][NG];
#pragma unroll
      for (int k = 0; k < 4; ++k) load_codes(u0 + k, cw[k]);
      load_term(u0, tv[0]);
      load_term(u0 + 1, tv[1]);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          // the B operands live in the accumulation registers for the whole unit (MFMA reads them there directly); the
          // 256 architectural registers hold both accumulator sets, the row terms and everything the VALU touches
          asm volatile("" : "+a"(bop[g][st]));
        }
#pragma unroll
      for (int st = 0; st < NST; ++st) decode_step(cw[0], st, av[0]);
      load_codes(u0 + 4, cw[0]);
      if (STATS) t_loop = __builtin_readcyclecounter();

      // ---- screen of a finished subtile: a pair survives when c1 * acc <= threshold, i.e. acc >= thr (c1 < 0). Part 1, per
      // query group: the lanes whose 16 accumulator registers hold a survivor; part 2: the (rare) survivors themselves
      unsigned long long gmask[NG];
      auto screen_group = [&](const f32x16_t (&ac)[NG], const int g, const bool valid) {
        float m = fmaxf(fmaxf(ac[g][0], ac[g][1]), ac[g][2]);  // (v_max3_f32 chains: this file is built with -fno-honor-nans)
#pragma unroll
        for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, ac[g][i]), ac[g][i + 1]);
        m        = fmaxf(m, ac[g][15]);
        gmask[g] = valid ? __ballot(m >= thr[g]) : 0ull;
      };
      auto screen_finish = [&](const f32x16_t (&ac)[NG], const uint32_t u, const bool valid) {
        unsigned long long any_mask = 0ull;
#pragma unroll
        for (int g = 0; g < NG; ++g) any_mask |= gmask[g];
        if (STATS && valid) { st_pairs += 32u * count; st_sub += 1u; }
        if (any_mask == 0ull) return;  // the usual case
        // slow path (one or two survivors): the lanes that hold one collect their hits in a bit mask; from there on the wave
        // works through them with scalar control - one lane writes, the fill count lives in a scalar register. (Measured
        // against a lane-parallel append - ballot per accumulator register, positions by mbcnt - at the C3 shape: 3.2 vs
        // 3.7 ms; at the survivor rates of unnormalised inner products neither beats pq_filter_kernel's per-lane loop.)
        if (STATS) st_slow += 1u;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (gmask[g] == 0ull) continue;
          uint32_t hits = 0u;
#pragma unroll
          for (int i = 0; i < 16; ++i) hits |= (ac[g][i] >= thr[g] ? 1u : 0u) << i;
          unsigned long long lm = gmask[g];
          while (lm != 0ull) {
            const uint32_t src = (uint32_t)__ffsll((long long)lm) - 1u;
            lm &= lm - 1ull;
            uint32_t hb        = __builtin_amdgcn_readlane(hits, src);
            const uint32_t pid = __builtin_amdgcn_readlane(pairid[g], src);
            while (hb != 0u) {
              const uint32_t i = (uint32_t)__ffs((int)hb) - 1u;
              hb &= hb - 1u;
              const uint32_t v = (u << 5) + (i & 3u) + 8u * (i >> 2) + 4u * (src >> 5);  // lane = (query, K half -> rows + 4)
              if (v >= r_end) continue;
              if (s_fill == kSurvChunk) {  // the next chunk (the first one: at the wave's first survivor)
                uint32_t c = 0u;
                if (lane == 0u) c = atomicAdd(a.surv_cnt, 1u);
                s_chunk = __builtin_amdgcn_readfirstlane(c);
                s_fill  = 0u;
              }
              if (s_chunk < a.n_chunks) {
                if (lane == 0u) a.surv[(size_t)s_chunk * kSurvChunk + s_fill] = make_uint2(pid, base_row + v);
              } else if (lane == 0u) {
                a.qflag[pid / a.n_probes] = 1u;  // the buffer is full: the query is handed back to the LUT scan
              }
              s_fill += 1u;
              if (STATS) st_surv += 1u;
            }
          }
        }
      };
      // ---- one subtile, phase P = (u - u0) % 4. A single wave issues in order: an MFMA that finds the matrix pipe busy holds
      // up everything behind it, so the other work of a K step is placed BETWEEN the step's MFMAs (one per query group, 32
      // cycles each): the four table addresses of the step's gathers for subtile u + 1, the gathers, and one more piece -
      // the row terms of subtile u + 2 (into the registers the first step's MFMAs have just read), the screen of one query
      // group of subtile u - 1, or the code words of subtile u + 5 (replacing those of u + 1 once they are decoded)
      auto step = [&](auto p_tag, const uint32_t u) {
        constexpr int P = decltype(p_tag)::value;
        constexpr int C = P & 1, N = (P + 1) & 1, S = (P + 1) & 3;
        const bool prev = u > u0 && u <= u1;  // subtile u - 1 exists (the loop runs whole groups of four: subtiles past the
                                              // end repeat the last one and are not screened)
        auto part_chunk = [](const int p) { return p + 1 < NST ? p + 1 : NST - 1; };  // screen part p runs in K step ...
#pragma unroll
        for (int st = 0; st < NST; ++st) {
          const f16x8_t aop = __builtin_bit_cast(f16x8_t, av[C][st]);
          auto mfma = [&](const int g) {
            if constexpr ((DBG & 4) != 0) {  // ablation: no MFMAs
              if (st == 0) acc[C][g] = tv[C];
              acc[C][g][st & 15] += (float)aop[0] * (float)bop[g][st][0];
            } else {
              acc[C][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, bop[g][st], st == 0 ? tv[C] : acc[C][g], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          };
          mfma(0);
          if (st == 1) load_term(u + 2, tv[C]);  // (every MFMA of step 0 has read tv[C] by now)
          decode_addr(cw[S], st);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (NG >= 2) mfma(1);
          decode_gather(st, av[N]);  // (clamped to the last subtile: a harmless repeat at the end)
          if (st == NST - 1) load_codes(u + 5, cw[S]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (NG >= 3) mfma(2);
#pragma unroll
          for (int p = 0; p < NG; ++p)
            if (part_chunk(p) == st) screen_group(acc[N], p, prev);
          if (part_chunk(NG) == st) screen_finish(acc[N], u - 1u, prev);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (NG >= 4) mfma(3);
        }
        if constexpr (NST == 1) load_term(u + 2, tv[C]);
      };
      using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
      using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;
      uint32_t u = u0;
      for (; u < u1; u += 4u) {
        step(P0{}, u);
        step(P1{}, u + 1u);
        step(P2{}, u + 2u);
        step(P3{}, u + 3u);
        if (desc_pending && u >= u0 + 4u) { load_desc(); desc_pending = false; }  // (wave-uniform)
      }
      if (u == u1) {  // (otherwise the last real subtile was screened inside the loop)
#pragma unroll
        for (int g = 0; g < NG; ++g) screen_group(acc[1], g, true);
        screen_finish(acc[1], u - 1u, true);
      }
    };
    if (u0 < u1) {
      const uint32_t ng = (count + 31u) >> 5;  // wave-uniform
      if constexpr (NGM == 4) {
        if (ng >= 4u) run(std::integral_constant<int, 4>{});
        else if (ng == 3u) run(std::integral_constant<int, 3>{});
        else if (ng == 2u) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 1>{});
      } else {
        if (ng >= 2u) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 1>{});
      }
    }
    if (desc_pending) load_desc();
    if (STATS) {
      const unsigned long long t_end = __builtin_readcyclecounter();
      st_t[0] += t_loop - t_unit;  // unit prologue (B operands, first decode)
      st_t[1] += t_end - t_loop;   // subtile loop
      st_units += 1u;
    }
  }
  if (STATS && lane == 0) {
    atomicAdd(&a.stats[1], st_surv);
    atomicAdd(&a.stats[0], st_pairs); atomicAdd(&a.stats[2], st_sub);
    atomicAdd(&a.stats[3], st_slow);  atomicAdd(&a.stats[4], st_t[0]); atomicAdd(&a.stats[5], st_t[1]);
    atomicAdd(&a.stats[6], st_t[2]);  atomicAdd(&a.stats[7], st_units);
  }
  if (s_chunk < a.n_chunks)
    for (uint32_t i = s_fill + lane; i < kSurvChunk; i += 64u) a.surv[(size_t)s_chunk * kSurvChunk + i] = make_uint2(0xffffffffu, 0u);
}

}  // namespace

void pq4_filter(resources& res, const filter4_launch& l)
{
  bprep_params b{};
  b.sorted_pairs = l.sorted_pairs; b.pair_off = l.pair_off; b.n_lists = l.n_lists; b.probes = l.probes;
  b.rot_queries = l.rot_queries; b.centers_rot = l.centers_rot; b.query_kth = l.query_kth; b.qflag = l.qflag;
  b.bq = static_cast<uint4*>(l.bq); b.thr = l.thr; b.n_probes = l.n_probes; b.rot_dim = l.rot_dim;
  b.sc = l.sc; b.c1 = l.c1; b.eps = l.eps; b.alpha = l.alpha; b.cbmax = l.cbmax; b.dmax = l.dmax; b.bound_max = l.bound_max;
  b.is_ip = l.is_ip; b.flat = l.flat; b.lbase = l.head_labels ? 0u : l.n_lists;
  b.lpl = l.pl == 1 ? 0u : l.pl == 2 ? 1u : l.pl == 4 ? 2u : 3u;
  const int nst = l.nch * l.pl;  // MFMA K steps
  const unsigned pgrid = (unsigned)grid_blocks(l.n_pairs, 128);
  b.norms = static_cast<float4*>(l.pair_norms);
  if (l.stage != 2) {
    profile_begin(res, "pq_bprep_kernel");
    auto prep = [&](auto nst_tag) {
      constexpr int N = decltype(nst_tag)::value;
      if (l.stage == 1) hipLaunchKernelGGL((pq_bprep_kernel<N, 1>), dim3(pgrid), dim3(256), 0, res.stream, b);
      else              hipLaunchKernelGGL((pq_bprep_kernel<N, 0>), dim3(pgrid), dim3(256), 0, res.stream, b);
    };
    switch (nst) {
#ifndef CUVS_AMD_F4_DEV
      case 1: prep(std::integral_constant<int, 1>{}); break;
      case 2: prep(std::integral_constant<int, 2>{}); break;
      case 3: prep(std::integral_constant<int, 3>{}); break;
      case 4: prep(std::integral_constant<int, 4>{}); break;
      case 5: prep(std::integral_constant<int, 5>{}); break;
      case 6: prep(std::integral_constant<int, 6>{}); break;
      case 7: prep(std::integral_constant<int, 7>{}); break;
      case 10: prep(std::integral_constant<int, 10>{}); break;
      case 12: prep(std::integral_constant<int, 12>{}); break;
      case 14: prep(std::integral_constant<int, 14>{}); break;
      case 16: prep(std::integral_constant<int, 16>{}); break;
#endif
      default: prep(std::integral_constant<int, 8>{}); break;
    }
    profile_end(res, "pq_bprep_kernel");
    HIP_TRY(hipGetLastError());
    if (l.stage == 1 || l.bprep_only) return;
  } else {
    hipLaunchKernelGGL(pq_thr_kernel, dim3((unsigned)grid_blocks(l.n_pairs, 256)), dim3(256), 0, res.stream, b);
  }
  filter4_params g{};
  g.units = l.units; g.n_units = l.n_units; g.xcd_ticket = l.xcd_ticket; g.sorted_pairs = l.sorted_pairs; g.pair_off = l.pair_off;
  g.n_lists = l.n_lists; g.bq = b.bq; g.thr = l.thr; g.cb16 = l.cb16; g.codes = l.codes; g.list_offsets = l.list_offsets;
  g.list_sizes = l.list_sizes; g.row_term = l.row_term; g.qflag = l.qflag; g.surv = static_cast<uint2*>(l.surv);
  g.surv_cnt = l.surv_cnt; g.n_chunks = l.surv_entries / kSurvChunk; g.n_probes = l.n_probes; g.unit_rows = l.unit_rows;
  g.stats = l.stats;
  // the two K halves of the decode table lie 64 KiB apart; PER_CLUSTER: a 4 KiB table region per wave
  const size_t fsmem = l.per_cluster ? 4 * 4096 : 65536 + (size_t)l.nch * 8 * 512 * l.pl;
  const bool term = l.row_term != nullptr;
  auto launch = [&](auto kern) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem));
    profile_begin(res, "pq_filter_kernel");
    hipLaunchKernelGGL(kern, dim3(l.grid), dim3(kF4Threads), fsmem, res.stream, g);
    profile_end(res, "pq_filter_kernel");
  };
#ifndef CUVS_AMD_F4_DEV
  if (l.nch == 4 && l.pl == 2 && !l.per_cluster) {
    switch (l.dbg) {  // CUVS_AMD_SCAN_DEBUG bits 16..19: ablation builds of the kernel (bench shape only)
      case 1:  launch(pq_filter4_kernel<4, 2, true, 1>); break;
      case 2:  launch(pq_filter4_kernel<4, 2, true, 2>); break;
      case 4:  launch(pq_filter4_kernel<4, 2, true, 4>); break;
      case 8:  launch(pq_filter4_kernel<4, 2, true, 8>); break;
      case 16: launch(pq_filter4_kernel<4, 2, true, 16>); break;
      case 64: launch(pq_filter4_kernel<4, 2, true, 64>); break;
      case 80: launch(pq_filter4_kernel<4, 2, true, 80>); break;
      case 81: launch(pq_filter4_kernel<4, 2, true, 81>); break;
      default:
        if (l.stats != nullptr && term) launch(pq_filter4_kernel<4, 2, true, 0, true>);
        else if (term) launch(pq_filter4_kernel<4, 2, true, 0>);
        else launch(pq_filter4_kernel<4, 2, false, 0>);
        break;
    }
    return;
  }
  auto pick = [&](auto nch_tag, auto pl_tag) {
    constexpr int N = decltype(nch_tag)::value, P = decltype(pl_tag)::value;
    if (l.per_cluster) {
      if (term) launch(pq_filter4_kernel<N, P, true, 0, false, true>); else launch(pq_filter4_kernel<N, P, false, 0, false, true>);
    } else {
      if (term) launch(pq_filter4_kernel<N, P, true, 0>); else launch(pq_filter4_kernel<N, P, false, 0>);
    }
  };
  using N1 = std::integral_constant<int, 1>; using N2 = std::integral_constant<int, 2>; using N3 = std::integral_constant<int, 3>;
  using N4 = std::integral_constant<int, 4>; using N5 = std::integral_constant<int, 5>; using N6 = std::integral_constant<int, 6>;
  using N7 = std::integral_constant<int, 7>; using N8 = std::integral_constant<int, 8>;
  CUVS_EXPECTS(nst >= 1 && nst <= 16 && l.nch >= 1 && l.nch <= 8, "ivf_pq: shape outside the matrix-core filter (pq3_supported)");
  if (l.pl == 2) {
    switch (l.nch) {
      case 1: pick(N1{}, N2{}); break;
      case 2: pick(N2{}, N2{}); break;
      case 3: pick(N3{}, N2{}); break;
      case 4: pick(N4{}, N2{}); break;
      case 5: pick(N5{}, N2{}); break;
      case 6: pick(N6{}, N2{}); break;
      case 7: pick(N7{}, N2{}); break;
      default: pick(N8{}, N2{}); break;
    }
  } else if (l.pl == 1) {
    switch (l.nch) {
      case 1: pick(N1{}, N1{}); break;
      case 2: pick(N2{}, N1{}); break;
      case 3: pick(N3{}, N1{}); break;
      case 4: pick(N4{}, N1{}); break;
      case 5: pick(N5{}, N1{}); break;
      case 6: pick(N6{}, N1{}); break;
      case 7: pick(N7{}, N1{}); break;
      default: pick(N8{}, N1{}); break;
    }
  } else if (l.pl == 4) {
    switch (l.nch) {
      case 1: pick(N1{}, N4{}); break;
      case 2: pick(N2{}, N4{}); break;
      case 3: pick(N3{}, N4{}); break;
      default: pick(N4{}, N4{}); break;
    }
  } else {
    if (l.nch == 1) pick(N1{}, N8{}); else pick(N2{}, N8{});
  }
#else
  launch(pq_filter4_kernel<4, 2, true, 0>);
#endif
}

}  // namespace cuvs_amd
