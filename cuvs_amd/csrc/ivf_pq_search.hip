// IVF-PQ search on MI355X: coarse selection (fp32-MFMA distances + radix select_k), then a LIST-MAJOR
// LUT scan, then the per-query merge.
//
// Reference path: cpp/src/neighbors/ivf_pq/ivf_pq_search.cuh (search :881-1050, select_clusters :60-168,
// ivfpq_search_worker :421-669) and the kernel detail/jit_lto_kernels/compute_similarity_impl.cuh:77-173
// (create_lut_impl.cuh:17-78, compute_distances_impl.cuh:16-106, compute_score_impl.cuh:20-79).
// The reference launches one block per (query, probe) pair, each building a private LUT and gathering one
// LUT entry per code byte. On CDNA4 that loop is bound by LDS gather issue (ds_read_b32: 64 lanes / >=2 clk),
// not by HBM, so the schedule here is different:
//   * (query, probe) pairs are grouped by list (stable radix sort by list id) and cut into work items of
//     QPB pairs that probe the SAME list; a batch of 256+ queries is scheduled in two phases (every query's
//     nearest probe first), so that the per-query k-th bound is warm when the bulk of the probes runs;
//   * a persistent 1024-thread workgroup per CU draws work items from a per-XCD ticket counter and builds ONE
//     interleaved LUT for its QPB queries in LDS - entry (s, code) holds the QPB partial distances side by
//     side (8 bytes: 2 x fp32 or 4 x fp16), code-major with padded rows so that a gather address is a single
//     SDWA multiply - and a single ds_read_b64 gather serves QPB queries; the list's code bytes are read once
//     per work item (1 KiB coalesced per wave and chunk from the 64-row interleaved layout, L2 hits for all but
//     the first item of a list);
//   * every wave keeps a private sorted top list per query in registers (lane i holds rank i; up to 4 ranks per
//     lane for k <= 256; DPP insertion) - no workgroup barrier inside the scan; 64-row tiles are handed to the
//     waves through an LDS ticket; lanes whose partial sums already exceed every k-th bound sit out the rest of
//     the row (early stop), whole waves skip it; the 16 wave lists of a query are merged once per item, and only
//     for queries that inserted anything.
// Results: exact for the given codes/LUT precision - every (query, probe) pair yields its true top-k by
// (distance, row) order, independent of scheduling.
#include "ivf_pq.hpp"
#include "ops.hpp"
#include "device_utils.hpp"
#include "ivf_common.hpp"
#include "pq_lut_math.hpp"
#include "ivf_pq_scan3.hpp"

#include <cfloat>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace cuvs_amd {
extern unsigned long long g_pq3_last_stats[6];  // core.hip: cuvsAmdIvfPqLastFilterStats

void load_range_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim, int64_t r0,
                         int64_t cnt, float* out);

namespace {

constexpr int kScanThreads = 1024;
constexpr int kScanWaves   = kScanThreads / 64;
constexpr int kQueueRows   = 320;  // survivor queue of a wave: < 64 carried over + a block of 4 tiles

inline unsigned nblk(int64_t n, int per) { return grid_blocks(n, per); }

// ------------------------------------------------------------------ accumulators over interleaved LUT entries
template <typename LutT, typename AccT, int QPB>
struct lut_acc;

// native vector types: one LUT entry is ONE LDS load (b16/b32/b64) and accumulates with packed adds
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

template <>
struct lut_acc<float, float, 1> {
  using entry_t = float;
  float a       = 0.f;
  __device__ inline void add(entry_t e) { a += e; }
  __device__ inline float get(int) const { return a; }
  __device__ static inline entry_t pack(const float (&v)[1]) { return v[0]; }
};
template <>
struct lut_acc<float, float, 2> {
  using entry_t = f32x2_t;
  f32x2_t a     = {0.f, 0.f};
  __device__ inline void add(entry_t e) { a += e; }
  __device__ inline float get(int j) const { return j == 0 ? a.x : a.y; }
  __device__ static inline entry_t pack(const float (&v)[2]) { return f32x2_t{v[0], v[1]}; }
};
template <>
struct lut_acc<__half, float, 1> {
  using entry_t = _Float16;
  float a       = 0.f;
  __device__ inline void add(entry_t e) { a += (float)e; }
  __device__ inline float get(int) const { return a; }
  __device__ static inline entry_t pack(const float (&v)[1]) { return to_lut_half(v[0]); }
};
template <>
struct lut_acc<__half, __half, 1> {
  using entry_t = _Float16;
  _Float16 a    = (_Float16)0.f;
  __device__ inline void add(entry_t e) { a += e; }
  __device__ inline float get(int) const { return (float)a; }
  __device__ static inline entry_t pack(const float (&v)[1]) { return to_lut_half(v[0]); }
};
// fp16 LUT entries summed in fp32: v_fma_mix_f32 reads one half of a packed pair, widens it exactly and adds it to the
// fp32 accumulator in ONE instruction ((float)e + a, a single rounding - the same value as convert-then-add)
__device__ inline void add_lo_half(float& a, uint32_t pair)
{
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(pair));
}
__device__ inline void add_hi_half(float& a, uint32_t pair)
{
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(pair));
}
template <>
struct lut_acc<__half, float, 2> {
  using entry_t = f16x2_t;
  float a0 = 0.f, a1 = 0.f;
  __device__ inline void add(entry_t e)
  {
    const uint32_t ev = __builtin_bit_cast(uint32_t, e);
    add_lo_half(a0, ev); add_hi_half(a1, ev);
  }
  __device__ inline float get(int j) const { return j == 0 ? a0 : a1; }
  __device__ static inline entry_t pack(const float (&v)[2]) { return f16x2_t{to_lut_half(v[0]), to_lut_half(v[1])}; }
};
template <>
struct lut_acc<__half, __half, 2> {
  using entry_t = f16x2_t;
  f16x2_t a     = {(_Float16)0.f, (_Float16)0.f};
  __device__ inline void add(entry_t e) { a += e; }
  __device__ inline float get(int j) const { return j == 0 ? (float)a.x : (float)a.y; }
  __device__ static inline entry_t pack(const float (&v)[2]) { return f16x2_t{to_lut_half(v[0]), to_lut_half(v[1])}; }
};
template <>
struct lut_acc<__half, float, 4> {
  using entry_t = f16x4_t;
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  __device__ inline void add(entry_t e)
  {
    const u32x2_t ev = __builtin_bit_cast(u32x2_t, e);
    add_lo_half(a0, ev.x); add_hi_half(a1, ev.x); add_lo_half(a2, ev.y); add_hi_half(a3, ev.y);
  }
  __device__ inline float get(int j) const { return j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3; }
  __device__ static inline entry_t pack(const float (&v)[4])
  {
    return f16x4_t{to_lut_half(v[0]), to_lut_half(v[1]), to_lut_half(v[2]), to_lut_half(v[3])};
  }
};
template <>
struct lut_acc<__half, __half, 4> {
  using entry_t = f16x4_t;
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  u32x2_t a     = {0u, 0u};  // four fp16 partial distances
  // Both packed adds are pinned with volatile asm: left to itself hipcc sinks the upper-half chain
  // below the scan loop and spills every gathered entry to scratch.
  __device__ inline void add(entry_t e)
  {
    u32x2_t ev = __builtin_bit_cast(u32x2_t, e);
    asm volatile("v_pk_add_f16 %0, %0, %2\n\tv_pk_add_f16 %1, %1, %3" : "+v"(a.x), "+v"(a.y) : "v"(ev.x), "v"(ev.y));
  }
  __device__ inline float get(int j) const
  {
    f16x4_t h = __builtin_bit_cast(f16x4_t, a);
    return (float)h[j];
  }
  __device__ static inline entry_t pack(const float (&v)[4])
  {
    return f16x4_t{to_lut_half(v[0]), to_lut_half(v[1]), to_lut_half(v[2]), to_lut_half(v[3])};
  }
};

// dbg 128 statistics (CUVS_AMD_SCAN_DEBUG=128 prints them per search): wave cycles per phase, rows per stage
enum scan_stat { ST_HEADER, ST_LUT, ST_SCAN, ST_STAGE2, ST_MERGE, ST_ROWS, ST_QUEUED, ST_S2_CALLS, ST_ALIVE1, ST_ALIVE2,
                 ST_ALIVE3, ST_CAND, ST_ITEMS, ST_F_LOAD, ST_F_GATHER, ST_F_FLUSH, ST_COUNT };

struct scan_args {
  uint32_t one_shot = 0;         // head launch of the two-stream schedule: the item count, one workgroup per item (pq3_head::one_shot)
  uint32_t row_limit = 0;        //   and the partial head's row limit (pq3_head::row_limit)
  const work_item* items;
  const uint32_t* item_begin;    // device scalars: this launch walks items [*item_begin, *item_end)
  const uint32_t* item_end;      //   (item_begin == nullptr: from 0)
  uint32_t n_lists;              // item.list >= n_lists: tail-phase label of list item.list - n_lists
  uint32_t* xcd_ticket;          // 8 zeroed counters, 32 words apart: next unclaimed item of each XCD's share
  const uint32_t* sorted_pairs;  // pair ids (q * n_probes + probe rank) grouped by list
  const float* rot_queries;      // [n_queries, rot_dim]
  const float* centers_rot;      // [n_lists, rot_dim]
  const float* pq_centers;       // [pq_dim, pq_len, book], or [n_lists, pq_len, book] when per_cluster
  int per_cluster;
  const uint8_t* codes;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  float* out_d;                  // [n_pairs, k]
  uint32_t* out_i;               // [n_pairs, k] flat row
  uint32_t n_probes, rot_dim, pq_dim, pq_len, pq_bits, n_chunks, cpc, k;
  int is_ip;
  int lut_fp8;  // LUT entries pass through the reference's fp_8bit<5, is_ip> (see fp8_round_trip)
  uint32_t* query_kth;  // [n_queries] order-preserving key of the best known k-th distance (shared by probes)
  float* all_scores;             // non-fused path (large k): [n_queries, scores_ld] score of every probed row
  uint32_t* all_rows;            //   flat row of every column (0xffffffff: nothing there)
  const uint32_t* pair_seg;      //   first column of each pair in its query's row
  size_t scores_ld;
  unsigned long long* stats;  // dbg 128: per-phase wave cycles and row counters (see scan_stat)
  int dbg;  // ablation switches (CUVS_AMD_SCAN_DEBUG): 1 no LUT build, 2 no gathers, 4 no top-k, 8 no early stop
  // LUT that does not fit the LDS (e.g. pq_dim 384 x 8 bit, the default for 768-d data): a per-workgroup LUT in global
  // memory, served from L2 - the reference's non-shared-memory LUT mode (ivf_pq_compute_similarity_impl.cuh:449-465)
  char* global_lut = nullptr;
  size_t global_lut_stride = 0;
  // pre-filter (compute_distances_impl.cuh:78-80, ivf_pq_search.cuh:1111-1134): bitset over SOURCE ids, 1 keeps the row
  const uint32_t* filter_bits = nullptr;
  const int64_t* indices      = nullptr;  // flat row -> source id (only read when filtering)
  uint32_t qcap = 0;  // pq_scan2_kernel: rows per survivor queue (0: the compile-time capacity; smaller: test hook)
};

// gathers of one 16-byte chunk of 8-bit codes, issued 8 at a time (8 independent ds_reads in flight;
// 16 at a time spills at the 128-VGPR budget of a 1024-thread workgroup)
template <typename acc_t>
__device__ inline void gather16(acc_t& acc, const typename acc_t::entry_t* __restrict__ lut_chunk, const uint4 cw)
{
  using entry_t = typename acc_t::entry_t;
  const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    entry_t e[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int bb  = h * 8 + b;
      uint32_t code = (ws[bb >> 2] >> ((bb & 3) * 8)) & 0xffu;
      e[b]          = lut_chunk[(bb << 8) + code];
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) acc.add(e[b]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- code-major LUT (FAST4 kernels): entry (s, code) lives at byte  code * kRow + s * sizeof(entry),
// kRow = 64 entries + 8 bytes of padding. The byte address of a gather is then ONE VALU instruction - a 24-bit
// multiply whose operand is a byte of the code word picked by SDWA - plus an immediate offset in the ds_read,
// instead of bit-field extract + shift-add (the subspace-major layout needs the 17-bit s * 2048 in the address
// register). The padding keeps both the LUT writes (64 consecutive codes of one subspace: stride kRow) and the
// random gathers spread over all banks. The LUT must start at LDS address 0 (checked once per launch): the
// FAST4 kernels use no static LDS.
template <typename entry_t>
struct cm_lut {
  static constexpr uint32_t esz  = sizeof(entry_t);
  static constexpr uint32_t kRow = 64 * esz + 8;
  typedef __attribute__((address_space(3))) const entry_t* rd_ptr;
  typedef __attribute__((address_space(3))) entry_t* wr_ptr;
  static constexpr size_t bytes() { return (size_t)256 * kRow; }
  __device__ static inline void store(uint32_t s, uint32_t code, entry_t v)
  {
    *(wr_ptr)(uintptr_t)(code * kRow + s * esz) = v;
  }
};

template <int BYTE>
__device__ inline uint32_t sdwa_byte_times(uint32_t word, uint32_t factor)
{
  uint32_t r;
  if constexpr (BYTE == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(factor), "v"(word));
  if constexpr (BYTE == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(factor), "v"(word));
  if constexpr (BYTE == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(factor), "v"(word));
  if constexpr (BYTE == 3) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(factor), "v"(word));
  return r;
}

// the 16 gathers of chunk CH (subspaces CH*16 .. CH*16+15), 8 at a time
template <typename acc_t, int CH>
__device__ inline void gather16_cm(acc_t& acc, const uint4 cw)
{
  using entry_t = typename acc_t::entry_t;
  using L       = cm_lut<entry_t>;
  const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    entry_t e[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int bb       = h * 8 + b;
      const uint32_t w   = ws[bb >> 2];
      const uint32_t row = (bb & 3) == 0 ? sdwa_byte_times<0>(w, L::kRow)
                         : (bb & 3) == 1 ? sdwa_byte_times<1>(w, L::kRow)
                         : (bb & 3) == 2 ? sdwa_byte_times<2>(w, L::kRow)
                                         : sdwa_byte_times<3>(w, L::kRow);
      e[b] = *(typename L::rd_ptr)(uintptr_t)(row + (uint32_t)(CH * 16 + bb) * L::esz);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) acc.add(e[b]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// pre-filter test of a flat row (wave-uniform or per lane): bit `source id` of the bitset, 1 keeps the row
__device__ inline bool row_passes(const scan_args& a, const uint32_t flat_row)
{
  if (a.filter_bits == nullptr) return true;
  const int64_t sid = a.indices[flat_row];
  return (a.filter_bits[sid >> 5] >> (sid & 31)) & 1u;
}

// The register-resident codebook (pq_dim 64 x pq_len 2 x 256 codes = 32 values per thread of a 1024-thread workgroup):
// thread (wave w, lane l) owns subspaces (l & 15) + 16 sg, sg = 0..3, and the codes pq_code0(w, l) .. + 3. A 16-lane
// LDS store group then writes 16 DIFFERENT subspaces of one code: contiguous in the code-major exact LUT, and one bank
// quad each in the unpadded filter LUT of pq_scan2_kernel (whose reads rely on "bank = subspace").
__device__ inline uint32_t pq_code0(const int wave, const int lane) { return (uint32_t)wave * 16u + ((uint32_t)lane >> 4) * 4u; }
__device__ inline void pq_regs_load(float (&pqreg)[4][2][4], const float* __restrict__ pq_centers, const bool on)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sl = (uint32_t)lane & 15u, cb = pq_code0(wave, lane);
#pragma unroll
  for (int sg = 0; sg < 4; ++sg)
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        pqreg[sg][l][t] = on ? pq_centers[(size_t)((sl + sg * 16) * 2 + l) * 256 + cb + t] : 0.f;
}

// LDS carve of the scan kernel (all offsets multiples of 16). The FAST4 LUT must sit at LDS address 0, so the kernel
// keeps no static LDS: [LUT | merge area (reuses the LUT region)] [query residuals] [list centre] [kthb 16 words]
// [pair ids 16 words] [2 work-item hand-over slots] [per-wave survivor queues: 16 x kQueueRows rows]
struct scan_layout {
  size_t qv, cv, kthb, pid, slots, queue, total;
  __host__ __device__ scan_layout(size_t lut_bytes, int qpb, uint32_t rot_dim, uint32_t k)
  {
    size_t off = (lut_bytes + 15) & ~size_t(15);
    size_t mg = (size_t)qpb * kScanWaves * k * 8;  // merge area reuses the LUT region after the scan
    if (k > 64) mg = std::max<size_t>(mg, (size_t)kScanWaves * 256 * 8);  // workgroup merge: 16 lists padded to 256 entries
    if (mg > off) off = (mg + 15) & ~size_t(15);
    qv = off;    off += (((size_t)qpb * rot_dim * 4) + 15) & ~size_t(15);
    cv = off;    off += (((size_t)rot_dim * 4) + 15) & ~size_t(15);
    kthb = off;  off += 16 * 4;
    pid = off;   off += 16 * 4;
    slots = off; off += 2 * 16;
    queue = off; off += (size_t)kScanWaves * kQueueRows * 4;
    total = off;
  }
};

// FAST4: pq_bits == 8, pq_dim == 64 (4 full chunks): the four chunk loads of a tile are issued back to back. Otherwise the generic path handles any pq_dim / pq_bits.
template <typename LutT, typename AccT, int QPB, bool FAST4, int E, bool ALL, bool GLUT = false>
__device__ inline void pq_scan_item(const scan_args& a, const work_item item, char* smem,
                                    const float (&pqreg)[4][2][4], const bool pq_in_regs,
                                    const work_item* __restrict__ share, const uint32_t share_len,
                                    const uint32_t next_ticket, const int next_slot_id)
{
  using acc_t   = lut_acc<LutT, AccT, QPB>;
  using entry_t = typename acc_t::entry_t;

  const uint32_t book      = 1u << a.pq_bits;
  const uint32_t lut_elems = a.pq_dim * book;
  static_assert(!(GLUT && FAST4), "the code-major LUT lives in LDS");
  const scan_layout lay(GLUT ? (size_t)0 : (FAST4 ? cm_lut<entry_t>::bytes() : (size_t)lut_elems * sizeof(entry_t)), QPB,
                        a.rot_dim, a.k);
  entry_t* lut     = GLUT ? reinterpret_cast<entry_t*>(a.global_lut + (size_t)blockIdx.x * a.global_lut_stride)
                          : reinterpret_cast<entry_t*>(smem);
  float* qv        = reinterpret_cast<float*>(smem + lay.qv);
  float* cv        = reinterpret_cast<float*>(smem + lay.cv);
  uint32_t* kthb   = reinterpret_cast<uint32_t*>(smem + lay.kthb);
  uint32_t* pid    = reinterpret_cast<uint32_t*>(smem + lay.pid);
  uint4* next_slot = reinterpret_cast<uint4*>(smem + lay.slots) + next_slot_id;

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const bool stat  = (a.dbg & 128) != 0;  // phase cycles: a few atomics per wave and item
  const bool statc = (a.dbg & 512) != 0;  // row counters: atomics per tile (slow; counts only)
  // every wave owns a row of counters (no contention); the host sums the rows
  auto stat_add = [&](int which, unsigned long long v) {
    // fire-and-forget atomic on the wave's private slot: a plain += would wait (vmcnt) for every load in flight - e.g. the
    // prefetched code loads - and book their latency to the phase being closed
    if (lane == 0) atomicAdd(&a.stats[((size_t)blockIdx.x * kScanWaves + wave) * ST_COUNT + which], v);
  };
  unsigned long long t_prev = stat ? __builtin_readcyclecounter() : 0ull, t_s2 = 0ull;
  auto stat_phase = [&](int which) {
    if (stat) { const unsigned long long t = __builtin_readcyclecounter(); stat_add(which, t - t_prev); t_prev = t; }
  };

  const uint32_t L        = item.list >= a.n_lists ? item.list - a.n_lists : item.list;
  const uint32_t base_row = a.list_offsets[L];
  const uint32_t len      = a.list_sizes[L];

  if (tid < QPB) {
    const uint32_t p = tid < (int)item.count ? a.sorted_pairs[item.first + tid] : 0xffffffffu;
    pid[tid]         = p;
    kthb[tid]        = p != 0xffffffffu ? a.query_kth[p / a.n_probes] : 0u;
    kthb[8 + tid]    = 0u;  // set once any wave inserts a candidate for query `tid` of this item
    if (tid == 0) kthb[12] = 0u;  // tile ticket of this item (see the scan loop)
  }
  // query residuals (L2) or raw rotated queries + list centre (IP); every thread resolves its pair id itself so
  // that this phase needs no barrier after the header loads above
  for (uint32_t t = tid; t < QPB * a.rot_dim; t += kScanThreads) {
    uint32_t j = t / a.rot_dim, dd = t % a.rot_dim;
    float v = 0.f;
    if (j < item.count) {
      uint32_t q = a.sorted_pairs[item.first + j] / a.n_probes;
      v          = a.rot_queries[(size_t)q * a.rot_dim + dd];
      if (!a.is_ip) v -= a.centers_rot[(size_t)L * a.rot_dim + dd];
    }
    qv[t] = v;
  }
  for (uint32_t t = tid; t < a.rot_dim; t += kScanThreads) cv[t] = a.centers_rot[(size_t)L * a.rot_dim + t];
  __syncthreads();
  stat_phase(ST_HEADER);

  // ---- LUT (create_lut_impl.cuh:17-78): entry (s, c) = QPB partial scores side by side
  if (pq_in_regs && !(a.dbg & 1)) {
    // pq_dim 64 x pq_len 2 x 256 codes: this thread's 32 codebook values stay in registers for the whole
    // persistent launch (pq_scan_kernel), so the LUT build touches no global memory at all
    // The metric test sits OUTSIDE the unrolled loops (inside, it became scalar branches around every entry),
    // and a scheduling fence after every entry keeps the compiler from overlapping entries: that overlap costs
    // ~45 spilled VGPRs at the 128-register budget, i.e. the codebook registers end up in scratch.
    // A thread owns subspaces (lane & 15) + 16 sg and the four codes pq_code0(..) .. + 3 (see pq_regs_load): the 16
    // lanes of an LDS store group write 16 neighbouring subspaces of ONE code - 128 contiguous bytes.
    const uint32_t sl = (uint32_t)lane & 15u, cb = pq_code0(wave, lane);
    // (the fp8 test is a compile-time argument of the loop body: as a run-time test inside it hipcc emitted one scalar
    // branch per LUT value - 64 per thread and LUT)
    auto build_regs = [&](auto fp8_tag) {
    constexpr bool FP8 = decltype(fp8_tag)::value;
    if (!a.is_ip) {
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const uint32_t s = sl + sg * 16;
        float q[2][QPB];
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
          for (int j = 0; j < QPB; ++j) q[l][j] = qv[j * a.rot_dim + s * 2 + l];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float p0 = pqreg[sg][0][t], p1 = pqreg[sg][1][t];
          float sc[QPB];
          // (packed fp32 pairs - v_pk_add_f32 / v_pk_fma_f32 - were measured here: ~50 spilled VGPRs, slower)
#pragma unroll
          for (int j = 0; j < QPB; ++j) {
            float d0 = q[0][j] - p0;
            float d1 = q[1][j] - p1;
            sc[j]    = __fmaf_rn(d1, d1, __fmaf_rn(d0, d0, 0.f));
            if constexpr (FP8) sc[j] = fp8_round_trip<AccT>(sc[j], false);
          }
          cm_lut<entry_t>::store(s, cb + t, acc_t::pack(sc));  // pq_in_regs implies FAST4
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const uint32_t s = sl + sg * 16;
        float q[2][QPB];
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
          for (int j = 0; j < QPB; ++j) q[l][j] = qv[j * a.rot_dim + s * 2 + l];
        const float cc0 = cv[s * 2], cc1 = cv[s * 2 + 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float sc[QPB];
#pragma unroll
          for (int j = 0; j < QPB; ++j) {
            float v = __fmaf_rn(-q[0][j], cc0, 0.f);
            v       = __fmaf_rn(-q[0][j], pqreg[sg][0][t], v);
            v       = __fmaf_rn(-q[1][j], cc1, v);
            sc[j]   = __fmaf_rn(-q[1][j], pqreg[sg][1][t], v);
            if constexpr (FP8) sc[j] = fp8_round_trip<AccT>(sc[j], true);
          }
          cm_lut<entry_t>::store(s, cb + t, acc_t::pack(sc));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    };
    if (a.lut_fp8) build_regs(std::true_type{}); else build_regs(std::false_type{});
  } else if (book >= 64 && !(a.dbg & 1)) {
    // wave-per-subspace: the QPB query residuals of subspace s are read from LDS once per wave (they are the
    // same for all 64 lanes) and reused for the book/64 code blocks; the codebook loads of a subspace are
    // independent and in flight together
    for (uint32_t s = wave; s < a.pq_dim; s += kScanWaves) {
      for (uint32_t c0 = 0; c0 < book; c0 += 256) {
        float sc[4][QPB];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < QPB; ++j) sc[t][j] = 0.f;
        // the codebook values of FOUR components are loaded before any is used (the build of a pq_len = 12 LUT was a chain
        // of 12 dependent L2 latencies per subspace); components are accumulated in the same order
        for (uint32_t l0 = 0; l0 < a.pq_len; l0 += 4) {
          float p[4][4];
#pragma unroll
          for (int li = 0; li < 4; ++li) {
            const uint32_t l  = min(l0 + li, a.pq_len - 1);
            const uint32_t dd = s * a.pq_len + l;
            const float* pqr  = a.pq_centers + (size_t)(a.per_cluster ? L * a.pq_len + l : dd) * book + c0 + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) p[li][t] = (c0 + t * 64 < book) ? pqr[t * 64] : 0.f;
          }
#pragma unroll
          for (int li = 0; li < 4; ++li) {
            if (l0 + li >= a.pq_len) break;  // wave-uniform
            const uint32_t dd = s * a.pq_len + l0 + li;
            float q[QPB];
#pragma unroll
            for (int j = 0; j < QPB; ++j) q[j] = qv[j * a.rot_dim + dd];
            const float cc = cv[dd];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
              for (int j = 0; j < QPB; ++j) {
                if (!a.is_ip) {
                  float diff = q[j] - p[li][t];
                  sc[t][j]   = __fmaf_rn(diff, diff, sc[t][j]);
                } else {
                  sc[t][j] = __fmaf_rn(-q[j], cc, sc[t][j]);
                  sc[t][j] = __fmaf_rn(-q[j], p[li][t], sc[t][j]);
                }
              }
            }
          }
        }
        if (a.lut_fp8) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < QPB; ++j) sc[t][j] = fp8_round_trip<AccT>(sc[t][j], a.is_ip != 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (c0 + t * 64 < book) {
            if (FAST4) cm_lut<entry_t>::store(s, c0 + t * 64 + lane, acc_t::pack(sc[t]));
            else       lut[(s << a.pq_bits) + c0 + t * 64 + lane] = acc_t::pack(sc[t]);
          }
      }
    }
  } else {
#pragma unroll 2
    for (uint32_t e = tid; e < ((a.dbg & 1) ? 0u : lut_elems); e += kScanThreads) {
      const uint32_t s = e >> a.pq_bits, c = e & (book - 1);
      float sc[QPB];
#pragma unroll
      for (int j = 0; j < QPB; ++j) sc[j] = 0.f;
      for (uint32_t l = 0; l < a.pq_len; ++l) {
        const uint32_t dd = s * a.pq_len + l;
        const float p     = a.pq_centers[(size_t)(a.per_cluster ? L * a.pq_len + l : dd) * book + c];
        if (!a.is_ip) {
#pragma unroll
          for (int j = 0; j < QPB; ++j) {
            float diff = qv[j * a.rot_dim + dd] - p;
            sc[j]      = __fmaf_rn(diff, diff, sc[j]);
          }
        } else {
          const float cc = cv[dd];
#pragma unroll
          for (int j = 0; j < QPB; ++j) {
            float q = qv[j * a.rot_dim + dd];
            sc[j]   = __fmaf_rn(-q, cc, sc[j]);
            sc[j]   = __fmaf_rn(-q, p, sc[j]);
          }
        }
      }
      if (a.lut_fp8) {
#pragma unroll
        for (int j = 0; j < QPB; ++j) sc[j] = fp8_round_trip<AccT>(sc[j], a.is_ip != 0);
      }
      lut[e] = acc_t::pack(sc);
    }
  }
  if constexpr (GLUT) __threadfence_block();  // the LUT went to global memory: visible to the workgroup's other waves
  __syncthreads();
  stat_phase(ST_LUT);

  // header of the workgroup's next item: its ticket was drawn at the start of this item and has arrived by
  // now; the load issued here lands during the scan and is handed over through LDS before the merge barrier
  uint4 next_hdr = make_uint4(0u, 0u, 0u, 0xffffffffu);  // plain registers (a struct here lived in scratch)
  if (threadIdx.x == 0 && next_ticket < share_len) next_hdr = *reinterpret_cast<const uint4*>(share + next_ticket);

  // ---- scan: every wave keeps a private sorted top list per query in registers; no workgroup barrier in
  // the loop. kthb[j] (LDS) is the tightest k-th bound any wave (or an earlier probe of the same query)
  // has established; only candidates at or below it are looked at.
  wave_top<E> top[QPB];
#pragma unroll
  for (int j = 0; j < QPB; ++j) top[j].init();

  const uint32_t n_iter = (a.dbg & 16) ? 0u : (len + kScanThreads - 1) / kScanThreads;  // dbg 16: no scan loop
  const size_t g0       = (size_t)(base_row >> 6);
  const uint4* codes16  = reinterpret_cast<const uint4*>(a.codes);
  const int kr          = (int)a.k - 1;
  const bool prune      = FAST4 && !ALL && !a.is_ip && !(a.dbg & 8);  // dbg 8: early stop off (ablation)

  // k-th bounds of the item's queries as floats (+inf while a query has fewer than k candidates); read once per
  // tile - they only ever decrease, so a row dropped against these is also rejected by the (fresher) filter
  auto load_bounds = [&](float (&bf)[QPB]) {
#pragma unroll
    for (int j = 0; j < QPB; ++j) {
      const uint32_t kk = __builtin_amdgcn_readfirstlane(kthb[j]);
      bf[j] = j >= (int)item.count ? -INFINITY : (kk >= 0xff800000u ? INFINITY : key_to_float(kk));
    }
  };

  // candidate filter + insertion for the rows held one per lane (v = in-list row of this lane)
  auto offer = [&](const acc_t& acc, const bool cand, const uint32_t v) {
    if constexpr (ALL) {  // non-fused path: every score goes to the query's row, select_k runs afterwards
#pragma unroll
      for (int j = 0; j < QPB; ++j) {
        if (j >= (int)item.count) break;
        const uint32_t p = pid[j];
        if (cand && row_passes(a, base_row + v)) {  // filtered rows keep the "invalid" fill
          const size_t o  = (size_t)(p / a.n_probes) * a.scores_ld + a.pair_seg[p] + v;
          a.all_scores[o] = acc.get(j);
          a.all_rows[o]   = base_row + v;
        }
      }
      return;
    }
    if (a.dbg & 4) return;
    if (__ballot(cand) == 0ull) return;  // the usual case once the bounds are warm: nothing in this tile
    // one ballot over "any query passes" first (almost always empty once the bounds are warm), then per query
    float dj[QPB];
    uint32_t djk[QPB];
    bool any = false;
#pragma unroll
    for (int j = 0; j < QPB; ++j) {
      dj[j]  = acc.get(j);
      // L2 scores are >= 0: their order-preserving key is just the sign bit set
      djk[j] = a.is_ip ? float_to_key(dj[j]) : (__float_as_uint(dj[j]) | 0x80000000u);
      any    = any || (j < (int)item.count && djk[j] <= kthb[j]);  // kthb: LDS broadcast reads
    }
    if (__ballot(cand && any) == 0ull) return;
#pragma unroll
    for (int j = 0; j < QPB; ++j) {
      if (j >= (int)item.count) break;
      unsigned long long m = __ballot(cand && djk[j] <= kthb[j]);
      if (m == 0ull) continue;
      float kd    = top[j].rank_d(kr);
      uint32_t ki = top[j].rank_i(kr);
      bool improved = false;
      while (m != 0ull) {
        const int src = (int)__ffsll((long long)m) - 1;
        m &= m - 1ull;
        const float cd    = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj[j]), src));
        const uint32_t ci = __builtin_amdgcn_readlane(v, src);
        if (!row_passes(a, base_row + ci)) continue;  // pre-filter: masked rows never enter a top list
        if ((cd < kd) || (cd == kd && ci < ki)) {
          top[j].insert(cd, ci, lane);
          kd       = top[j].rank_d(kr);
          ki       = top[j].rank_i(kr);
          improved = true;
        }
      }
      // this wave's k-th best bounds the list's k-th best from above: publish it
      if (improved && lane == 0) {
        kthb[8 + j] = 1u;
        if (kd < INFINITY) atomicMin(&kthb[j], float_to_key(kd));
      }
    }
  };

  // FAST4: all 64 subspaces of the rows held one per lane (any rows of the list: a contiguous tile reads 1 KiB per
  // chunk, queued survivors read 16 bytes each). With `prune`, the early stop (compute_score_impl.cuh:70-71): L2 LUT
  // entries are >= 0, so a row whose partial sums already exceed the k-th bound of every query of the item cannot
  // enter any top list; its lane sits out the remaining chunks - fewer active lanes mean fewer LDS bank conflicts.
  auto scan_rows4 = [&](const uint32_t v, const bool valid) {
    acc_t acc;
    bool cand = valid;  // lanes that may still hold a candidate for some query of the item
    const uint32_t fr = base_row + (valid ? v : 0u);  // padded rows of a group are zero-filled: readable
    const uint4* cp   = codes16 + ((size_t)(fr >> 6) * 4) * 64 + (fr & 63u);
    uint4 cur[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) cur[ch] = cp[ch * 64];
    if (!(a.dbg & 2) && prune) {
      float bf[QPB];
      load_bounds(bf);
      bool alive = valid;
      auto still_below = [&]() {
        bool below = false;
#pragma unroll
        for (int j = 0; j < QPB; ++j) below = below || (acc.get(j) <= bf[j]);
        return below;
      };
      if (alive) gather16_cm<acc_t, 0>(acc, cur[0]);
      alive = alive && still_below();
      if (alive) gather16_cm<acc_t, 1>(acc, cur[1]);
      alive = alive && still_below();
      if (statc) stat_add(ST_ALIVE1, __popcll(__ballot(alive)));
      if (alive) gather16_cm<acc_t, 2>(acc, cur[2]);
      alive = alive && still_below();
      if (statc) stat_add(ST_ALIVE2, __popcll(__ballot(alive)));
      if (alive) gather16_cm<acc_t, 3>(acc, cur[3]);
      cand = alive && still_below();
      if (statc) stat_add(ST_ALIVE3, __popcll(__ballot(cand)));
    } else if (!(a.dbg & 2)) {
      gather16_cm<acc_t, 0>(acc, cur[0]);
      gather16_cm<acc_t, 1>(acc, cur[1]);
      gather16_cm<acc_t, 2>(acc, cur[2]);
      gather16_cm<acc_t, 3>(acc, cur[3]);
    } else {
      acc.add(lut[cur[0].x & 0xff]);
    }
    offer(acc, cand, v);
  };

  // The ~20 work items of a list run on ~20 CUs of one XCD at about the same time. Each starts its pass over
  // the list at a different rotation, so that at any moment the CUs touch different parts of the list: one of them
  // pulls a line into L2, the others find it there later, instead of all of them missing on it together.
  const uint32_t rot = (a.dbg & 64) ? 0u : (item.first / QPB);
  // Tiles (64 rows) are handed to the waves through a ticket in LDS: a tile costs anything between a chunk of
  // gathers and a pass over its survivors, and a fixed tile -> wave map left waves idle at the merge barrier.
  // A wave draws its next ticket before it works on the current tile, so the LDS round trip is off the path.
  const uint32_t n_tiles = n_iter == 0 ? 0u : (len + 63u) / 64u;
  const uint32_t rot_t   = n_tiles ? (rot * kScanWaves) % n_tiles : 0u;
  uint32_t ticket = 0u;
  if (lane == 0) ticket = atomicAdd(&kthb[12], 1u);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (FAST4 && prune && !(a.dbg & 256)) {
    // ---- two stages. Once the bounds are warm (after the head phase) all but ~1 % of the (row, query) pairs are out
    // after the first 16 subspaces, but a wave only saves work when all 64 of its rows are out. So stage 1 reads
    // ONLY chunk 0 of the rows (1 KiB per tile instead of 4) - a ticket is a block of 4 tiles, whose four loads are
    // in flight together: with one load per wave the loop ran at the latency of an L2 hit - and queues the rows
    // that are still below a bound; stage 2 takes 64 survivors at a time (one per lane, a full wave of useful
    // gathers) and scores them over all 64 subspaces in the order every other path uses, so the sums are
    // bit-identical. dbg 256: single stage.
    uint32_t* wq = reinterpret_cast<uint32_t*>(smem + lay.queue) + wave * kQueueRows;
    uint32_t qn  = 0u;  // queued rows of this wave (wave-uniform), < 64 between blocks
    const uint32_t n_blocks = (n_tiles + 3u) / 4u;
    const uint32_t rot_b    = n_blocks ? (rot * kScanWaves) % n_blocks : 0u;
    while (ticket < n_blocks) {
      const uint32_t blk = (ticket + rot_b) % n_blocks;
      uint32_t next = 0u;
      if (lane == 0) next = atomicAdd(&kthb[12], 1u);
      uint4 c0[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t tile = min(blk * 4 + t, n_tiles - 1);
        c0[t] = codes16[((g0 + (size_t)tile) * 4) * 64 + lane];
      }
      float bf[QPB];
      load_bounds(bf);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t tile = blk * 4 + t;
        const uint32_t v    = tile * 64 + lane;
        acc_t acc;
        gather16_cm<acc_t, 0>(acc, c0[t]);
        bool alive = false;
#pragma unroll
        for (int j = 0; j < QPB; ++j) alive = alive || (acc.get(j) <= bf[j]);
        alive = alive && v < len;  // also drops the clamped tiles past the end of the list
        const unsigned long long m = __ballot(alive);
        if (m != 0ull) {
          const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (alive) wq[pos] = v;
          qn += (uint32_t)__popcll(m);
          if (statc) stat_add(ST_QUEUED, __popcll(m));
        }
      }
      while (qn >= 64u) {  // newest 64 first: no compaction of the queue needed
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        qn -= 64u;
        const uint32_t r = wq[qn + lane];
        const unsigned long long t0 = stat ? __builtin_readcyclecounter() : 0ull;
        scan_rows4(r, true);
        if (stat) t_s2 += __builtin_readcyclecounter() - t0;
        if (statc) stat_add(ST_S2_CALLS, 1);
      }
      ticket = __builtin_amdgcn_readfirstlane(next);
    }
    if (qn != 0u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const bool valid = (uint32_t)lane < qn;
      const uint32_t r = wq[valid ? lane : 0];
      const unsigned long long t0 = stat ? __builtin_readcyclecounter() : 0ull;
      scan_rows4(r, valid);
      if (stat) t_s2 += __builtin_readcyclecounter() - t0;
      if (statc) stat_add(ST_S2_CALLS, 1);
    }
  } else {
    while (ticket < n_tiles) {
      const uint32_t tile  = (ticket + rot_t) % n_tiles;
      uint32_t next = 0u;
      if (lane == 0) next = atomicAdd(&kthb[12], 1u);
      const uint32_t v = tile * 64 + lane;  // in-list position of this lane's row
      const bool valid = v < len;
      if (FAST4) {
        scan_rows4(v, valid);
      } else {
        acc_t acc;
        const size_t g  = g0 + (size_t)tile;
        const uint4* cp = codes16 + (g * a.n_chunks) * 64 + lane;
        if (a.pq_bits == 8) {
          for (uint32_t ch = 0; ch < a.n_chunks; ++ch) {
            uint4 cw          = cp[(size_t)ch * 64];
            const uint32_t s0 = ch * 16;
            if (s0 + 16 <= a.pq_dim) {
              gather16(acc, lut + (s0 << 8), cw);
            } else {
              const uint32_t ws[4] = {cw.x, cw.y, cw.z, cw.w};
              for (uint32_t b = 0; b < 16 && s0 + b < a.pq_dim; ++b) {
                uint32_t code = (ws[b >> 2] >> ((b & 3) * 8)) & 0xffu;
                acc.add(lut[((s0 + b) << 8) + code]);
              }
            }
          }
        } else {
          const uint32_t msk = book - 1;
          for (uint32_t ch = 0; ch < a.n_chunks; ++ch) {
            uint4 cw             = cp[(size_t)ch * 64];
            const uint32_t ws[5] = {cw.x, cw.y, cw.z, cw.w, 0u};
            for (uint32_t b = 0; b < a.cpc; ++b) {
              uint32_t s = ch * a.cpc + b;
              if (s >= a.pq_dim) break;
              uint32_t bit  = b * a.pq_bits;
              uint64_t two  = (uint64_t)ws[bit >> 5] | ((uint64_t)ws[(bit >> 5) + 1] << 32);
              uint32_t code = (uint32_t)(two >> (bit & 31)) & msk;
              acc.add(lut[(s << a.pq_bits) + code]);
            }
          }
        }
        offer(acc, valid, v);
      }
      ticket = __builtin_amdgcn_readfirstlane(next);
    }
  }

  if (stat) {
    stat_phase(ST_SCAN);
    stat_add(ST_STAGE2, t_s2);
    if (wave == 0) { stat_add(ST_ROWS, len); stat_add(ST_ITEMS, 1); }
  }
  if (threadIdx.x == 0) *next_slot = next_hdr;
  if (ALL || (a.dbg & 32)) return;  // dbg 32: no merge / output (workgroup-uniform); ALL: non-fused path
  // ---- merge the 16 wave lists of every query (the LUT region is free now)
  __syncthreads();
  float* mg_d    = reinterpret_cast<float*>(smem);
  uint32_t* mg_i = reinterpret_cast<uint32_t*>(smem + (size_t)QPB * kScanWaves * a.k * 4);
  // Queries for which no wave inserted anything (the usual case once the bounds are warm) skip the merge and the
  // output: the host pre-fills the per-pair candidate rows with "invalid".
  {
    uint32_t any_ins = 0u;
#pragma unroll
    for (int j = 0; j < QPB; ++j) any_ins |= kthb[8 + j];
    if (any_ins == 0u) return;  // workgroup-uniform: nothing to merge for this item
  }
  if constexpr (E > 1) {
    // k > 64: the wave lists (sorted, rank e * 64 + lane) are merged by the whole workgroup, one query after the other
    constexpr int KP2 = E == 2 ? 128 : 256;  // E = 2: k <= 128, E = 4: k <= 256
    float* sd    = reinterpret_cast<float*>(smem);
    uint32_t* si = reinterpret_cast<uint32_t*>(smem + (size_t)kScanWaves * KP2 * 4);
    for (int j = 0; j < QPB; ++j) {
      if (j >= (int)item.count || kthb[8 + j] == 0u) continue;  // workgroup-uniform
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int r = e * 64 + lane;
        const bool in = r < (int)a.k;
        sd[wave * KP2 + r] = in ? top[j].d[e] : INFINITY;
        si[wave * KP2 + r] = in ? top[j].i[e] : 0xffffffffu;
      }
      __syncthreads();
      merge_sorted_lists<kScanThreads>(sd, si, kScanWaves, KP2, tid);
      const size_t o = (size_t)pid[j] * a.k;
      for (int r = tid; r < (int)a.k; r += kScanThreads) {
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
#pragma unroll
  for (int j = 0; j < QPB; ++j) {
    if (kthb[8 + j] == 0u) continue;  // workgroup-uniform
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int r = e * 64 + lane;
      if (r < (int)a.k) {
        mg_d[((size_t)j * kScanWaves + wave) * a.k + r] = top[j].d[e];
        mg_i[((size_t)j * kScanWaves + wave) * a.k + r] = top[j].i[e];
      }
    }
  }
  __syncthreads();
  if (wave < QPB && wave < (int)item.count && kthb[8 + wave] != 0u) {
    const int j = wave;
    wave_top<E> fin;
    fin.init();
    float kd    = INFINITY;
    uint32_t ki = 0xffffffffu;
    const int n = kScanWaves * (int)a.k;
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
    // per-pair result (pairs of empty lists get all-invalid rows)
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
    // tighten the bound shared by the other probes of this query
    if (lane == 0 && kd < INFINITY) atomicMin(&a.query_kth[pid[j] / a.n_probes], float_to_key(kd));
  }
}

// Persistent launch: one 1024-thread workgroup per CU walks the work items. Items are sorted by list, and the
// observed dispatch places workgroup b on XCD b % 8 (used for speed only): XCD x takes the x-th eighth of the
// item array and its 32 CUs work on 32 consecutive items, so the ~20 work items of a list run on ONE XCD at
// about the same time and share its 4 MiB L2 instead of pulling the list into all eight L2s.
// ALL: the non-fused path (every score written out, no top lists) - a template argument so that the fused kernels stay
// exactly as they were
template <typename LutT, typename AccT, int QPB, bool FAST4, int E, bool ALL = false, bool GLUT = false>
__global__ __launch_bounds__(kScanThreads) void pq_scan_kernel(scan_args a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t item0   = a.item_begin ? *a.item_begin : 0u;
  const uint32_t n_items = *a.item_end - item0;
  const uint32_t xcd = blockIdx.x & 7u, lb = blockIdx.x >> 3, per = gridDim.x >> 3;
  const uint32_t chunk = (n_items + 7u) / 8u;
  float pqreg[4][2][4];
  const bool pq_in_regs = FAST4 && a.pq_len == 2 && !a.per_cluster;  // FAST4: pq_dim 64, 8-bit codes
  // the code-major LUT addresses LDS absolutely (see cm_lut): fail loudly if the dynamic LDS does not start at 0
  if (FAST4 && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  pq_regs_load(pqreg, a.pq_centers, pq_in_regs);
  // Work distribution: XCD x owns the x-th eighth of the (list-sorted) item array and its workgroups draw
  // items from it through one ticket counter, so at any moment the XCD's 32 CUs work on ~32 CONSECUTIVE items
  // (~1.6 lists) and every list is pulled into that XCD's L2 once. A static stride let the CUs drift apart by
  // many steps over the ~1200 items each one processes, which spread a list's ~20 items over time and had L2
  // refetch it (TCC hit rate 53 %).
  using entry_t = typename lut_acc<LutT, AccT, QPB>::entry_t;
  work_item* sh_item;
  {
    const scan_layout lay(GLUT ? (size_t)0 : (FAST4 ? cm_lut<entry_t>::bytes() : (size_t)a.pq_dim * (1u << a.pq_bits) * sizeof(entry_t)),
                          QPB, a.rot_dim, a.k);
    sh_item = reinterpret_cast<work_item*>(smem + lay.slots);
  }
  const uint32_t share0     = min(n_items, xcd * chunk);
  const uint32_t share_len  = min(chunk, n_items - share0);
  const work_item* share    = a.items + item0 + share0;
  uint32_t* ticket          = a.xcd_ticket + xcd * 32;
  (void)lb; (void)per;
  if (threadIdx.x == 0) {
    const uint32_t t = atomicAdd(ticket, 1u);
    sh_item[0]       = t < share_len ? share[t] : work_item{0u, 0u, 0u, 0xffffffffu};
  }
  __syncthreads();
  for (int buf = 0;; buf ^= 1) {
    const work_item cur = sh_item[buf];
    if (cur.pad == 0xffffffffu) break;  // workgroup-uniform
    uint32_t next_ticket = 0xffffffffu;
    if (threadIdx.x == 0) next_ticket = atomicAdd(ticket, 1u);
    const unsigned long long t0 = (a.dbg & 128) ? __builtin_readcyclecounter() : 0ull;
    pq_scan_item<LutT, AccT, QPB, FAST4, E, ALL, GLUT>(a, cur, smem, pqreg, pq_in_regs, share, share_len, next_ticket,
                                             buf ^ 1);
    __syncthreads();
    // ST_MERGE accumulates the whole item; the host subtracts the other phases
    if ((a.dbg & 128) && (threadIdx.x & 63) == 0)
      a.stats[((size_t)blockIdx.x * kScanWaves + (threadIdx.x >> 6)) * ST_COUNT + ST_MERGE] += __builtin_readcyclecounter() - t0;
  }
}

// =====================================================================================================================
// pq_scan2_kernel: the warm-bounds scan (tail phase of a batch; pq_dim 64, 8-bit codes, pq_len 2, L2 metrics, k <= 64).
//
// Once every query has a k-th bound (after the head phase), ~96 % of the (row, query) pairs of a probed list are out
// after the first 16 of the 64 subspaces, but a wave only saves work when all 64 of its rows are out for all queries
// of the item. So the work item is cut differently here - (list, 8 queries in NG groups of EQ) - and runs in three steps:
//   filter  a LUT of the first 16 subspaces only, 16-byte entries = the 8 queries' partial distances as fp16 side by
//           side, and ONE pass over chunk 0 of the list's rows (1 KiB per 64 rows, four tiles in flight per wave). A
//           random gather of `ds_read_b128` (16 lanes per LDS cycle, 4 banks each) costs ~3 cycles per lane group in
//           bank conflicts when all lanes look up the same subspace at random codes (round 2: 54 % of the LDS cycles
//           of this kernel were conflict cycles). Round 3: the LUT is laid out WITHOUT padding - entry (s, code) at
//           byte code * 256 + s * 16, i.e. in bank quad s whatever the code - and at step t lane l looks up subspace
//           (l + t) mod 16: the 16 lanes of a group hit 16 different bank quads by construction. The code bytes of a
//           row are rotated by (l mod 16) once per tile (v_alignbyte), the gather address is one v_perm_b32.
//           Every lane therefore sums its 16 entries in its own order, and in fp16 whatever the score type; the filter
//           stays SAFE because it only drops a row when that sum exceeds the bound by more than 1/32: the two sums of
//           the same 16 non-negative numbers differ by at most a factor (1 + 2^-11)^15 / (1 - 2^-11)^15 < 1.016 (fp16,
//           any order), the canonical partial sum only grows over the remaining 48 subspaces, so such a row is above
//           the bound in the exact order too. Bounds at or beyond the fp16 range keep every row. For an fp32 LUT the
//           filter entries are the exact entries rounded toward zero to fp16 (a lower bound), the argument is the same.
//           Rows still below the bound of some query of group g go to queue g in LDS;
//   exact   for every group: the exact 64-subspace LUT of pq_scan_kernel (8-byte entries, EQ queries: 4 x fp16 or
//           2 x fp32) replaces the filter LUT, and the queued rows are scored 64 at a time, one row per lane, over all
//           64 subspaces in the order every other path uses (bit-identical sums). The code loads of a wave's first
//           batch are issued BEFORE the LUT build, which hides their latency;
//   merge   per group, as in pq_scan_kernel.
// Results are identical to pq_scan_kernel's: a dropped row can never have reached a top list. A queue that overflows
// (cold bounds) makes the group fall back to scoring every row.
constexpr int kQueueBytes = 24 * 1024;  // survivor queues of a workgroup (NG queues of kQueueBytes / 4 / NG rows)

struct filter_lut {  // code-major, 16 subspaces x 16 bytes per code row, no padding: bank quad == subspace
  static constexpr uint32_t kRow = 256;
  static constexpr size_t bytes() { return (size_t)256 * kRow; }
};

struct scan2_layout {
  size_t qv, cv, ctrl, pid, slots, queue, total;
  __host__ __device__ scan2_layout(size_t lut_bytes, int fq, uint32_t rot_dim)
  {
    size_t off = (lut_bytes + 15) & ~size_t(15);
    qv = off;    off += (((size_t)fq * rot_dim * 4) + 15) & ~size_t(15);
    cv = off;    off += (((size_t)rot_dim * 4) + 15) & ~size_t(15);
    ctrl = off;  off += 32 * 4;  // [0..7] k-th bound keys, [8..15] "inserted" flags, [16] block ticket, [17..20] queue
                                 // lengths
    pid = off;   off += 8 * 4;
    slots = off; off += 2 * 16;
    queue = off; off += (size_t)kQueueBytes;
    total = off;
  }
};

// L2 LUT entries of TWO queries at once with packed fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: the
// cost of one plain fp32 instruction each, measured): (q0 - p0)^2 + (q1 - p1)^2 per component, with exactly the
// roundings of fma(d1, d1, fma(d0, d0, 0)) - a product plus zero rounds like the product alone.
// p = {p0, p1}, the two components of one codebook entry in a register pair: op_sel broadcasts one half of the pair to
// both lanes of the packed subtract, so the codebook stays in 32 registers (hipcc materialises {p0, p0} pairs otherwise)
__device__ inline f32x2_t pq_l2_entry2(const f32x2_t q0, const f32x2_t q1, const f32x2_t p)
{
  f32x2_t d0, d1;
  asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "v"(q0), "v"(p));               // q0 - p.x
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "v"(q1), "v"(p));  // q1 - p.y
  return __builtin_elementwise_fma(d1, d1, d0 * d0);
}
// the codebook slice of pq_regs_load as (component 0, component 1) pairs
__device__ inline void pq_regs_load2(f32x2_t (&pq2)[4][4], const float* __restrict__ pq_centers)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sl = (uint32_t)lane & 15u, cb = pq_code0(wave, lane);
#pragma unroll
  for (int sg = 0; sg < 4; ++sg)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      pq2[sg][t] = f32x2_t{pq_centers[(size_t)((sl + sg * 16) * 2 + 0) * 256 + cb + t],
                           pq_centers[(size_t)((sl + sg * 16) * 2 + 1) * 256 + cb + t]};
}

// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts (no LDS)
__device__ inline uint32_t wave_inclusive_scan_dpp(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// two floats -> packed fp16, rounded toward zero (a lower bound of non-negative entries)
__device__ inline uint32_t pack_half_rtz(float a, float b)
{
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// smallest fp16 >= v for 0 <= v < 60000 (bits)
__device__ inline uint32_t half_bits_round_up(float v)
{
  if (v < 6.2e-5f) return 0x0400u;  // below the normal range: the smallest normal number bounds it
  uint32_t h = pack_half_rtz(v, v) & 0xffffu;
  if ((float)__builtin_bit_cast(_Float16, (uint16_t)h) < v) h += 1u;
  return h;
}
__device__ inline uint32_t pk_sub_f16(uint32_t a, uint32_t b)  // a - b, two halves
{
  uint32_t r;
  asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// X: the kernel-wide per-lane constants of the rotated filter gathers - byte b of xoff[w] = ((lane + 4 w + b) & 15) << 4,
// the byte offset of the subspace this lane looks up at step 4 w + b inside a 256-byte code row
template <typename LutT, typename AccT, int EQ, int NG, int E>
__device__ inline void pq_scan2_item(const scan_args& a, const work_item item, char* smem, const f32x2_t (&pq2)[4][4],
                                     const work_item* __restrict__ share,
                                     const uint32_t share_len, const uint32_t next_ticket, const int next_slot_id)
{
  constexpr int FQ = EQ * NG;
  static_assert(FQ == 8 && (NG == 2 || NG == 4), "the filter entry holds 8 fp16 partial distances");
  constexpr uint32_t kCap = kQueueBytes / 4 / NG;
  using acc_t   = lut_acc<LutT, AccT, EQ>;
  using entry_t = typename acc_t::entry_t;
  using XL      = cm_lut<entry_t>;
  using FL      = filter_lut;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  static_assert(sizeof(entry_t) == 8, "exact LUT entries are 8 bytes (4 x fp16 or 2 x fp32)");

  const scan2_layout lay(XL::bytes(), FQ, a.rot_dim);
  float* qv        = reinterpret_cast<float*>(smem + lay.qv);
  uint32_t* ctrl   = reinterpret_cast<uint32_t*>(smem + lay.ctrl);
  uint32_t* pid    = reinterpret_cast<uint32_t*>(smem + lay.pid);
  uint4* next_slot = reinterpret_cast<uint4*>(smem + lay.slots) + next_slot_id;
  uint32_t* queue  = reinterpret_cast<uint32_t*>(smem + lay.queue);
  const uint32_t qcap = a.qcap != 0u ? min(a.qcap, kCap) : kCap;

  // Everything derived from the thread id is recomputed per item: the empty asm hides the value from hipcc's
  // loop-invariant code motion, which otherwise keeps dozens of per-thread addresses (LUT slots, query residual slots,
  // codebook pointers) alive across the persistent item loop - at the 128-register budget of a 1024-thread workgroup
  // they end up in scratch, and every use waits for a scratch load.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // per-lane constants of the rotated filter gathers: byte b of xoff[w] = ((lane + 4 w + b) & 15) << 4, the byte offset
  // of the subspace this lane looks up at step 4 w + b inside a 256-byte code row
  (void)0;
  uint32_t xoff[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    xoff[w] = 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b) xoff[w] |= ((((uint32_t)lane + 4u * w + b) & 15u) << 4) << (8 * b);
  }

  const bool stat = (a.dbg & 128) != 0;
  auto stat_add = [&](int which, unsigned long long v) {
    // fire-and-forget atomic on the wave's private slot: a plain += would wait (vmcnt) for every load in flight - e.g. the
    // prefetched code loads - and book their latency to the phase being closed
    if (lane == 0) atomicAdd(&a.stats[((size_t)blockIdx.x * kScanWaves + wave) * ST_COUNT + which], v);
  };
  unsigned long long t_prev = stat ? __builtin_readcyclecounter() : 0ull;
  auto stat_phase = [&](int which) {
    if (stat) { const unsigned long long t = __builtin_readcyclecounter(); stat_add(which, t - t_prev); t_prev = t; }
  };

  const uint32_t L        = item.list >= a.n_lists ? item.list - a.n_lists : item.list;
  const uint32_t base_row = a.list_offsets[L];
  const uint32_t len      = a.list_sizes[L];

  // The codebook is NOT kept in registers across the item (round 2 pinned 32 registers per thread for it; the filter pass
  // and the exact passes then had to fit their gathers into what was left): every LUT build fetches this thread's
  // slice - (pq_len = 2) x 4 codes of subspace sl + 16 sg: one 16-byte load per component from the 128 KiB codebook,
  // L2-resident - and the loads are issued BEFORE the barrier that precedes the build, so their latency falls into
  // the barrier wait.
  const uint32_t sl = (uint32_t)lane & 15u, cb = pq_code0(wave, lane);  // this thread's subspaces / codes
  auto load_pq = [&](const int sg, float4 (&dst)[2]) {
#pragma unroll
    for (int l = 0; l < 2; ++l)
      dst[l] = (a.dbg & 32768) ? make_float4(0.f, 0.f, 0.f, 0.f)  // dbg 32768: no codebook loads (ablation)
                               : *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.pq_centers) +
                                                                  (uint32_t)((((sl + sg * 16) * 2 + l) * 256 + cb) * 4u));
  };

  if (tid < FQ) {
    const uint32_t p = tid < (int)item.count ? a.sorted_pairs[item.first + tid] : 0xffffffffu;
    pid[tid]         = p;
    ctrl[tid]        = p != 0xffffffffu ? a.query_kth[p / a.n_probes] : 0u;
    ctrl[8 + tid]    = 0u;
  }
  if (tid >= 16 && tid < 24) ctrl[tid] = 0u;
  // query residuals (L2 only here); every thread resolves its pair id itself: no barrier after the header loads
  for (uint32_t t = tid; t < FQ * a.rot_dim; t += kScanThreads) {
    const uint32_t j = t / a.rot_dim, dd = t % a.rot_dim;
    float v = 0.f;
    if (j < item.count && !(a.dbg & 65536)) {  // dbg 65536: no residual loads (ablation)
      const uint32_t q = a.sorted_pairs[item.first + j] / a.n_probes;
      v = a.rot_queries[(size_t)q * a.rot_dim + dd] - a.centers_rot[(size_t)L * a.rot_dim + dd];
    }
    qv[t] = v;
  }
  __syncthreads();
  stat_phase(ST_HEADER);

  // ---- filter LUT: subspaces 0..15 (this thread: subspace sl, codes cb .. cb + 3), the 8 queries' entries as fp16
  auto build_filter_lut = [&](auto fp8_tag) {
    constexpr bool FP8 = decltype(fp8_tag)::value;  // compile-time: see pq_scan_item
    f32x2_t q[2][FQ / 2];  // query pairs (j, j + 1)
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int j = 0; j < FQ; ++j) q[l][j >> 1][j & 1] = qv[j * a.rot_dim + sl * 2 + l];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u32x4_t ev;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x2_t e2 = pq_l2_entry2(q[0][jj], q[1][jj], pq2[0][t]);
        float sc[2] = {e2.x, e2.y};
        if constexpr (FP8) { sc[0] = fp8_round_trip<AccT>(sc[0], false); sc[1] = fp8_round_trip<AccT>(sc[1], false); }  // L2 only here: fp_8bit<5, false>
        // fp16 LUT: the exact entry itself (round to nearest, as the exact LUT stores it); fp32 LUT: rounded toward zero
        if constexpr (sizeof(LutT) == 2) ev[jj] = __builtin_bit_cast(uint32_t, f16x2_t{to_lut_half(sc[0]), to_lut_half(sc[1])});
        else                             ev[jj] = pack_half_rtz(sc[0], sc[1]);
      }
      *(__attribute__((address_space(3))) u32x4_t*)(uintptr_t)((cb + t) * FL::kRow + sl * 16) = ev;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (!(a.dbg & 1)) { if (a.lut_fp8) build_filter_lut(std::true_type{}); else build_filter_lut(std::false_type{}); }
  stat_phase(ST_LUT);
  __syncthreads();
  stat_phase(ST_ALIVE1);  // (pq_scan2: barrier wait after the filter LUT build)

  // header of the workgroup's next item (its ticket was drawn at the start of this item)
  uint4 next_hdr = make_uint4(0u, 0u, 0u, 0xffffffffu);
  if (threadIdx.x == 0 && next_ticket < share_len) next_hdr = *reinterpret_cast<const uint4*>(share + next_ticket);

  const size_t g0      = (size_t)(base_row >> 6);
  const uint4* codes16 = reinterpret_cast<const uint4*>(a.codes);
  const int kr         = (int)a.k - 1;
  const uint32_t n_tiles = (a.dbg & 16) ? 0u : (len + 63u) / 64u;

  auto key_bound = [&](const uint32_t kk, const bool live) {
    return !live ? -INFINITY : (kk >= 0xff800000u ? INFINITY : key_to_float(kk));
  };

  // ---- filter pass: chunk 0 of every row against the 8 bounds. Tiles (64 rows, 1 KiB of chunk-0 codes) are dealt to
  // the waves statically - wave w takes tiles w, w + 16, ... (a rotated start per item) - because every tile costs the
  // same here: a ticket per block of 4 tiles left 12 waves with two blocks and 4 with one (28 blocks for 16 waves), and
  // every block began with an exposed load. The codes of the next two tiles of a wave are in flight while it works.
  {
    const uint32_t rot   = (a.dbg & 64) ? 0u : (item.first / FQ);
    const uint32_t rot_t = n_tiles ? (rot * kScanWaves) % n_tiles : 0u;
    const uint32_t rr    = (uint32_t)lane & 15u;  // byte rotation of this lane's code words
    const uint32_t rb    = rr & 3u;               //   = rb bytes after (rr >> 2) dwords
    auto tile_of = [&](const uint32_t i) {  // the i-th tile of this wave (i * 16 + wave < n_tiles)
      uint32_t t = i * kScanWaves + (uint32_t)wave + rot_t;
      return t >= n_tiles ? t - n_tiles : t;
    };
    auto load_tile = [&](const uint32_t i) {
      const uint32_t t = min(tile_of(min(i, 0x00ffffffu)), n_tiles ? n_tiles - 1 : 0u);
      return codes16[((g0 + (size_t)t) * 4) * 64 + lane];
    };
    const uint32_t my_tiles = n_tiles > (uint32_t)wave ? (n_tiles - (uint32_t)wave + kScanWaves - 1) / kScanWaves : 0u;
    uint4 cb0 = make_uint4(0u, 0u, 0u, 0u), cb1 = cb0, cb2 = cb0;
    if (my_tiles > 0u) cb0 = load_tile(0);
    if (my_tiles > 1u) cb1 = load_tile(1);
    // the 8 bounds as fp16, widened by 1/32 and rounded up (lane j < 8 converts query j's), once per item; a query
    // without a bound yet, or with one beyond the fp16 range, keeps every row of its group alive (`force`)
    uint32_t bfh[4];
    uint32_t force = 0u;  // bit g: group g keeps every row
    {
      const int j       = lane & 7;
      const uint32_t kk = ctrl[j];
      uint32_t hb       = 0xfc00u;  // -inf: no such query in this item
      bool fo           = false;
      if (j < (int)item.count) {
        const float bw = (kk >= 0xff800000u ? INFINITY : key_to_float(kk)) * 1.03125f;
        if (!(bw < 60000.0f) || (a.dbg & 8)) { fo = true; hb = 0x7bffu; }  // dbg 8: early stop off (ablation)
        else hb = half_bits_round_up(bw);
      }
      const uint32_t fm = (uint32_t)__ballot(fo) & 0xffu;
      uint32_t hj[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) hj[q] = __builtin_amdgcn_readlane(hb, q);
#pragma unroll
      for (int d = 0; d < 4; ++d) bfh[d] = hj[2 * d] | (hj[2 * d + 1] << 16);
#pragma unroll
      for (int g = 0; g < NG; ++g) force |= ((fm >> (g * EQ)) & ((1u << EQ) - 1u)) ? (1u << g) : 0u;
    }
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    // survivors are collected in a bit field per lane (bit NG * slot + g: this lane's row of the wave's tile `slot` of
    // the current chunk of 8 tiles is alive for group g) and appended to the queues once per chunk: one LDS atomic per
    // group and 8 tiles
    uint32_t flags = 0u;
    auto flush = [&](const uint32_t i0) {  // tiles i0 .. of this wave (the set bits of `flags`)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        uint32_t gm = flags & (NG == 2 ? 0x5555u << g : 0x11111111u << g);
        const uint32_t cnt = (uint32_t)__popc(gm);            // this lane's surviving rows of the chunk for group g
        const uint32_t incl = wave_inclusive_scan_dpp(cnt);   // six DPP adds
        const uint32_t n = __builtin_amdgcn_readlane(incl, 63);
        if (n == 0u) continue;
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(&ctrl[17 + g], n);
        uint32_t pos = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
        while (gm != 0u) {  // at most 8 rounds, usually one or two: 11 % of the rows survive
          const uint32_t sl2 = ((uint32_t)__ffs((int)gm) - 1u) / NG;
          gm &= gm - 1u;
          if (pos < qcap) queue[g * kCap + pos] = tile_of(i0 + sl2) * 64 + lane;
          ++pos;
        }
      }
      flags = 0u;
    };
    unsigned long long tf_load = 0ull, tf_gather = 0ull, tf_flush = 0ull;  // dbg 128: where the pass spends its cycles
    auto process = [&](const uint32_t i, const uint4& c0) {
      const uint32_t slot = i & 7u;
      if (stat) {  // wait for this tile's codes (and only them)
        const unsigned long long t0 = __builtin_readcyclecounter();
        asm volatile("" ::"v"(c0.x), "v"(c0.y), "v"(c0.z), "v"(c0.w));
        tf_load += __builtin_readcyclecounter() - t0;
      }
      // rotate the 16 code bytes right by rr: byte k of R = code of subspace (k + rr) & 15
      uint32_t w0 = c0.x, w1 = c0.y, w2 = c0.z, w3 = c0.w;
      {
        const bool r1 = (rr & 4u) != 0u, r2 = (rr & 8u) != 0u;
        const uint32_t a0 = r1 ? w1 : w0, a1 = r1 ? w2 : w1, a2 = r1 ? w3 : w2, a3 = r1 ? w0 : w3;
        w0 = r2 ? a2 : a0; w1 = r2 ? a3 : a1; w2 = r2 ? a0 : a2; w3 = r2 ? a1 : a3;
      }
      uint32_t R[4];
      R[0] = __builtin_amdgcn_alignbyte(w1, w0, rb);
      R[1] = __builtin_amdgcn_alignbyte(w2, w1, rb);
      R[2] = __builtin_amdgcn_alignbyte(w3, w2, rb);
      R[3] = __builtin_amdgcn_alignbyte(w0, w3, rb);
      h2_t acc[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) acc[d] = h2_t{(_Float16)0.f, (_Float16)0.f};
      if (!(a.dbg & 2)) {
        // The 16 gathers and 64 packed adds of a tile as ONE hand-scheduled block: eight ds_read_b128 in flight, gather
        // s + 8 is issued right behind the adds of gather s (registers v96..v127 are the eight 16-byte slots). The
        // waves of the workgroup leave the barrier together and run this loop in lockstep, so LDS time and VALU time
        // only overlap when they overlap inside each wave; hipcc's own schedule - all 16 gathers, one wait, 64 adds -
        // gave 3.3 k cycles per round of tiles against ~1.7 k of VALU work and ~1.0 k of LDS time. lgkmcnt counts this
        // block's own reads only: it starts by draining what the compiler may have in flight and ends drained.
        uint32_t a0 = 0u, a1 = 0u, a2 = 0u, a3 = 0u, ta;
        const unsigned long long tg0 = stat ? __builtin_readcyclecounter() : 0ull;
        asm volatile(
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_perm_b32 %[t], %[R0], %[X0], %[s0]\n\t"
          "ds_read_b128 v[96:99], %[t]\n\t"
          "v_perm_b32 %[t], %[R0], %[X0], %[s1]\n\t"
          "ds_read_b128 v[100:103], %[t]\n\t"
          "v_perm_b32 %[t], %[R0], %[X0], %[s2]\n\t"
          "ds_read_b128 v[104:107], %[t]\n\t"
          "v_perm_b32 %[t], %[R0], %[X0], %[s3]\n\t"
          "ds_read_b128 v[108:111], %[t]\n\t"
          "v_perm_b32 %[t], %[R1], %[X1], %[s0]\n\t"
          "ds_read_b128 v[112:115], %[t]\n\t"
          "v_perm_b32 %[t], %[R1], %[X1], %[s1]\n\t"
          "ds_read_b128 v[116:119], %[t]\n\t"
          "v_perm_b32 %[t], %[R1], %[X1], %[s2]\n\t"
          "ds_read_b128 v[120:123], %[t]\n\t"
          "v_perm_b32 %[t], %[R1], %[X1], %[s3]\n\t"
          "ds_read_b128 v[124:127], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v96\n\t"
          "v_pk_add_f16 %[a1], %[a1], v97\n\t"
          "v_pk_add_f16 %[a2], %[a2], v98\n\t"
          "v_pk_add_f16 %[a3], %[a3], v99\n\t"
          "v_perm_b32 %[t], %[R2], %[X2], %[s0]\n\t"
          "ds_read_b128 v[96:99], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v100\n\t"
          "v_pk_add_f16 %[a1], %[a1], v101\n\t"
          "v_pk_add_f16 %[a2], %[a2], v102\n\t"
          "v_pk_add_f16 %[a3], %[a3], v103\n\t"
          "v_perm_b32 %[t], %[R2], %[X2], %[s1]\n\t"
          "ds_read_b128 v[100:103], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v104\n\t"
          "v_pk_add_f16 %[a1], %[a1], v105\n\t"
          "v_pk_add_f16 %[a2], %[a2], v106\n\t"
          "v_pk_add_f16 %[a3], %[a3], v107\n\t"
          "v_perm_b32 %[t], %[R2], %[X2], %[s2]\n\t"
          "ds_read_b128 v[104:107], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v108\n\t"
          "v_pk_add_f16 %[a1], %[a1], v109\n\t"
          "v_pk_add_f16 %[a2], %[a2], v110\n\t"
          "v_pk_add_f16 %[a3], %[a3], v111\n\t"
          "v_perm_b32 %[t], %[R2], %[X2], %[s3]\n\t"
          "ds_read_b128 v[108:111], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v112\n\t"
          "v_pk_add_f16 %[a1], %[a1], v113\n\t"
          "v_pk_add_f16 %[a2], %[a2], v114\n\t"
          "v_pk_add_f16 %[a3], %[a3], v115\n\t"
          "v_perm_b32 %[t], %[R3], %[X3], %[s0]\n\t"
          "ds_read_b128 v[112:115], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v116\n\t"
          "v_pk_add_f16 %[a1], %[a1], v117\n\t"
          "v_pk_add_f16 %[a2], %[a2], v118\n\t"
          "v_pk_add_f16 %[a3], %[a3], v119\n\t"
          "v_perm_b32 %[t], %[R3], %[X3], %[s1]\n\t"
          "ds_read_b128 v[116:119], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v120\n\t"
          "v_pk_add_f16 %[a1], %[a1], v121\n\t"
          "v_pk_add_f16 %[a2], %[a2], v122\n\t"
          "v_pk_add_f16 %[a3], %[a3], v123\n\t"
          "v_perm_b32 %[t], %[R3], %[X3], %[s2]\n\t"
          "ds_read_b128 v[120:123], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v124\n\t"
          "v_pk_add_f16 %[a1], %[a1], v125\n\t"
          "v_pk_add_f16 %[a2], %[a2], v126\n\t"
          "v_pk_add_f16 %[a3], %[a3], v127\n\t"
          "v_perm_b32 %[t], %[R3], %[X3], %[s3]\n\t"
          "ds_read_b128 v[124:127], %[t]\n\t"
          "s_waitcnt lgkmcnt(7)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v96\n\t"
          "v_pk_add_f16 %[a1], %[a1], v97\n\t"
          "v_pk_add_f16 %[a2], %[a2], v98\n\t"
          "v_pk_add_f16 %[a3], %[a3], v99\n\t"
          "s_waitcnt lgkmcnt(6)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v100\n\t"
          "v_pk_add_f16 %[a1], %[a1], v101\n\t"
          "v_pk_add_f16 %[a2], %[a2], v102\n\t"
          "v_pk_add_f16 %[a3], %[a3], v103\n\t"
          "s_waitcnt lgkmcnt(5)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v104\n\t"
          "v_pk_add_f16 %[a1], %[a1], v105\n\t"
          "v_pk_add_f16 %[a2], %[a2], v106\n\t"
          "v_pk_add_f16 %[a3], %[a3], v107\n\t"
          "s_waitcnt lgkmcnt(4)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v108\n\t"
          "v_pk_add_f16 %[a1], %[a1], v109\n\t"
          "v_pk_add_f16 %[a2], %[a2], v110\n\t"
          "v_pk_add_f16 %[a3], %[a3], v111\n\t"
          "s_waitcnt lgkmcnt(3)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v112\n\t"
          "v_pk_add_f16 %[a1], %[a1], v113\n\t"
          "v_pk_add_f16 %[a2], %[a2], v114\n\t"
          "v_pk_add_f16 %[a3], %[a3], v115\n\t"
          "s_waitcnt lgkmcnt(2)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v116\n\t"
          "v_pk_add_f16 %[a1], %[a1], v117\n\t"
          "v_pk_add_f16 %[a2], %[a2], v118\n\t"
          "v_pk_add_f16 %[a3], %[a3], v119\n\t"
          "s_waitcnt lgkmcnt(1)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v120\n\t"
          "v_pk_add_f16 %[a1], %[a1], v121\n\t"
          "v_pk_add_f16 %[a2], %[a2], v122\n\t"
          "v_pk_add_f16 %[a3], %[a3], v123\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_pk_add_f16 %[a0], %[a0], v124\n\t"
          "v_pk_add_f16 %[a1], %[a1], v125\n\t"
          "v_pk_add_f16 %[a2], %[a2], v126\n\t"
          "v_pk_add_f16 %[a3], %[a3], v127\n\t"
          : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t] "=&v"(ta)
          : [R0] "v"(R[0]), [R1] "v"(R[1]), [R2] "v"(R[2]), [R3] "v"(R[3]), [X0] "v"(xoff[0]), [X1] "v"(xoff[1]),
            [X2] "v"(xoff[2]), [X3] "v"(xoff[3]), [s0] "s"(0x0c0c0400u), [s1] "s"(0x0c0c0501u), [s2] "s"(0x0c0c0602u),
            [s3] "s"(0x0c0c0703u)
          : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
            "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
            "v124", "v125", "v126", "v127", "memory");
        if (stat) tf_gather += __builtin_readcyclecounter() - tg0;
        acc[0] = __builtin_bit_cast(h2_t, a0); acc[1] = __builtin_bit_cast(h2_t, a1);
        acc[2] = __builtin_bit_cast(h2_t, a2); acc[3] = __builtin_bit_cast(h2_t, a3);
      }
      // bound - sum per half: a set sign bit = that query is out (no inf - inf here: a +inf bound is `force`)
      uint32_t dv[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) dv[d] = pk_sub_f16(bfh[d], __builtin_bit_cast(uint32_t, acc[d]));
      const bool valid = tile_of(i) * 64 + lane < len;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        bool dead;
        if constexpr (NG == 2) dead = ((dv[2 * g] & dv[2 * g + 1]) & 0x80008000u) == 0x80008000u;
        else                   dead = (dv[g] & 0x80008000u) == 0x80008000u;
        const bool alive = valid && (!dead || ((force >> g) & 1u));
        flags |= (alive ? 1u : 0u) << (NG * slot + g);
      }
      if ((slot == 7u || i + 1 == my_tiles) && !(a.dbg & 8192)) {  // dbg 8192: survivors dropped (ablation)
        const unsigned long long t0 = stat ? __builtin_readcyclecounter() : 0ull;
        flush(i - slot);
        if (stat) tf_flush += __builtin_readcyclecounter() - t0;
      }
    };
    // three code buffers in rotation, the loop unrolled by three: the loads of the next TWO tiles stay in flight while
    // a tile is worked on (with two buffers and a copy hipcc waited for the newest load - vmcnt(0) - at every tile)
    for (uint32_t i = 0; i < my_tiles; i += 3) {
      if (i + 2 < my_tiles) cb2 = load_tile(i + 2);
      process(i, cb0);
      if (i + 1 >= my_tiles) break;
      if (i + 3 < my_tiles) cb0 = load_tile(i + 3);
      process(i + 1, cb1);
      if (i + 2 >= my_tiles) break;
      if (i + 4 < my_tiles) cb1 = load_tile(i + 4);
      process(i + 2, cb2);
    }
    if (stat) { stat_add(ST_F_LOAD, tf_load); stat_add(ST_F_GATHER, tf_gather); stat_add(ST_F_FLUSH, tf_flush); }
  }
  if (stat) { stat_phase(ST_SCAN); if (wave == 0) { stat_add(ST_ROWS, len); stat_add(ST_ITEMS, 1); } }
  __syncthreads();
  stat_phase(ST_ALIVE2);  // (pq_scan2: barrier wait after the filter pass)
  if (stat && wave == 0) {
    uint32_t nq = 0u;
#pragma unroll
    for (int g = 0; g < NG; ++g) nq += ctrl[17 + g];
    stat_add(ST_QUEUED, nq);
  }

  // ---- exact passes
  wave_top<E> top[EQ];
  auto load_codes = [&](const uint32_t v, uint4 (&cur)[4]) {
    const uint32_t fr = base_row + v;
    const uint4* cp   = codes16 + ((size_t)(fr >> 6) * 4) * 64 + (fr & 63u);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) cur[ch] = cp[ch * 64];
  };
  const int tid_item = tid;
  for (int g = 0; g < NG; ++g) {
    if (g * EQ >= (int)item.count) break;  // workgroup-uniform: no queries in this group
    // (per-group copy of the thread id behind an empty asm: keeps the per-thread LDS addresses of this loop body from
    // being hoisted out of the loop and spilled - see the top of the function)
    int tid_g = tid_item;
    asm volatile("" : "+v"(tid_g));
    const int lane = tid_g & 63, wave = tid_g >> 6;
    const uint32_t sl = (uint32_t)lane & 15u, cb = pq_code0(wave, lane);
    auto load_pq = [&](const int sg, float4 (&dst)[2]) {
#pragma unroll
      for (int l = 0; l < 2; ++l)
        dst[l] = (a.dbg & 32768) ? make_float4(0.f, 0.f, 0.f, 0.f)
                                 : *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.pq_centers) +
                                                                    (uint32_t)((((sl + sg * 16) * 2 + l) * 256 + cb) * 4u));
    };
    uint32_t* kthb = ctrl + g * EQ;        // bounds of this group's queries
    uint32_t* ins  = ctrl + 8 + g * EQ;
    const int cnt_g = min(EQ, (int)item.count - g * EQ);
    const uint32_t n_q   = ctrl[17 + g];
    const bool overflow  = n_q > qcap;
    const uint32_t n_bat = (a.dbg & 16384) ? 0u : (overflow ? n_tiles : (n_q + 63u) / 64u);  // dbg 16384: no exact batches
    const uint32_t* qg   = queue + g * kCap;
    auto batch_row = [&](const uint32_t b, bool& valid) {
      uint32_t v;
      if (overflow) { v = b * 64 + lane; valid = v < len; }
      else          { valid = b * 64 + lane < n_q; v = qg[valid ? b * 64 + lane : 0]; }
      return valid ? v : 0u;  // row 0 of the list is always readable
    };
    // first batch of this wave: its code loads fly during the LUT build
    uint4 cur[4];
    bool valid0 = false;
    uint32_t v0 = 0u;
    if ((uint32_t)wave < n_bat) { v0 = batch_row(wave, valid0); load_codes(v0, cur); }
    stat_phase(ST_MERGE);
    if (g > 0) __syncthreads();  // every wave is done with the previous LUT / merge area
    stat_phase(ST_S2_CALLS);  // (pq_scan2: barrier wait before the next exact LUT build)
    // ---- exact LUT of the group's EQ queries (same arithmetic and layout as pq_scan_kernel)
    auto build_exact_lut = [&](auto fp8_tag) {
      constexpr bool FP8 = decltype(fp8_tag)::value;
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const uint32_t s = sl + sg * 16;
        f32x2_t q[2][EQ / 2];  // query pairs (j, j + 1)
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
          for (int j = 0; j < EQ; ++j) q[l][j >> 1][j & 1] = qv[(g * EQ + j) * a.rot_dim + s * 2 + l];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float sc[EQ];
#pragma unroll
          for (int jj = 0; jj < EQ / 2; ++jj) {
            const f32x2_t e2 = pq_l2_entry2(q[0][jj], q[1][jj], pq2[sg][t]);
            sc[2 * jj] = e2.x; sc[2 * jj + 1] = e2.y;
            if constexpr (FP8) { sc[2 * jj] = fp8_round_trip<AccT>(sc[2 * jj], false); sc[2 * jj + 1] = fp8_round_trip<AccT>(sc[2 * jj + 1], false); }
          }
          XL::store(s, cb + t, acc_t::pack(sc));
        }
      }
    };
    if (!(a.dbg & 1)) { if (a.lut_fp8) build_exact_lut(std::true_type{}); else build_exact_lut(std::false_type{}); }
#pragma unroll
    for (int j = 0; j < EQ; ++j) top[j].init();
    stat_phase(ST_LUT);
    __syncthreads();
    stat_phase(ST_ALIVE3);  // (pq_scan2: barrier wait after an exact LUT build)

    for (uint32_t b = wave; b < n_bat; b += kScanWaves) {
      bool valid = valid0;
      uint32_t v = v0;
      if (b != (uint32_t)wave) { v = batch_row(b, valid); load_codes(v, cur); }
      float bf[EQ];
#pragma unroll
      for (int j = 0; j < EQ; ++j) bf[j] = key_bound(__builtin_amdgcn_readfirstlane(kthb[j]), j < cnt_g);
      acc_t acc;
      bool alive = valid;
      auto still_below = [&]() {
        bool below = false;
#pragma unroll
        for (int j = 0; j < EQ; ++j) below = below || (acc.get(j) <= bf[j]);
        return below;
      };
      if (!(a.dbg & 8)) {
        if (alive) gather16_cm<acc_t, 0>(acc, cur[0]);
        alive = alive && still_below();
        if (alive) gather16_cm<acc_t, 1>(acc, cur[1]);
        alive = alive && still_below();
        if (alive) gather16_cm<acc_t, 2>(acc, cur[2]);
        alive = alive && still_below();
        if (alive) gather16_cm<acc_t, 3>(acc, cur[3]);
        alive = alive && still_below();
      } else {
        gather16_cm<acc_t, 0>(acc, cur[0]);
        gather16_cm<acc_t, 1>(acc, cur[1]);
        gather16_cm<acc_t, 2>(acc, cur[2]);
        gather16_cm<acc_t, 3>(acc, cur[3]);
      }
      if (__ballot(alive) == 0ull) continue;
      // ---- candidate filter + insertion (pq_scan_kernel's)
      float dj[EQ];
      uint32_t djk[EQ];
      bool any = false;
#pragma unroll
      for (int j = 0; j < EQ; ++j) {
        dj[j]  = acc.get(j);
        djk[j] = __float_as_uint(dj[j]) | 0x80000000u;  // L2 scores are >= 0
        any    = any || (j < cnt_g && djk[j] <= kthb[j]);
      }
      if (__ballot(alive && any) == 0ull) continue;
#pragma unroll
      for (int j = 0; j < EQ; ++j) {
        if (j >= cnt_g) break;
        unsigned long long m = __ballot(alive && djk[j] <= kthb[j]);
        if (m == 0ull) continue;
        float kd    = top[j].rank_d(kr);
        uint32_t ki = top[j].rank_i(kr);
        bool improved = false;
        while (m != 0ull) {
          const int src = (int)__ffsll((long long)m) - 1;
          m &= m - 1ull;
          const float cd    = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dj[j]), src));
          const uint32_t ci = __builtin_amdgcn_readlane(v, src);
          if (!row_passes(a, base_row + ci)) continue;  // pre-filter: masked rows never enter a top list
          if ((cd < kd) || (cd == kd && ci < ki)) {
            top[j].insert(cd, ci, lane);
            kd       = top[j].rank_d(kr);
            ki       = top[j].rank_i(kr);
            improved = true;
          }
        }
        if (improved && lane == 0) {
          atomicOr(&ins[j], 1u << wave);  // which waves hold candidates of query j: only their lists are merged
          if (kd < INFINITY) atomicMin(&kthb[j], float_to_key(kd));
        }
      }
    }
    stat_phase(ST_STAGE2);
    if (a.dbg & (32 | 2048)) continue;  // dbg 32: no merge / output; 2048: in this kernel only (the head phase still warms the bounds)
    // ---- merge the 16 wave lists of the group's queries (the LUT region is free once every wave is here)
    __syncthreads();
    stat_phase(ST_CAND);  // (pq_scan2: barrier wait after an exact pass)
    {
      uint32_t any_ins = 0u;
#pragma unroll
      for (int j = 0; j < EQ; ++j) any_ins |= ins[j];
      if (any_ins == 0u) { stat_phase(ST_MERGE); continue; }  // workgroup-uniform: nothing to merge for this group
    }
    float* mg_d    = reinterpret_cast<float*>(smem);
    uint32_t* mg_i = reinterpret_cast<uint32_t*>(smem + (size_t)EQ * kScanWaves * a.k * 4);
#pragma unroll
    for (int j = 0; j < EQ; ++j) {
      if (!((ins[j] >> wave) & 1u)) continue;  // wave-uniform: this wave holds nothing for query j
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int r = e * 64 + lane;
        if (r < (int)a.k) {
          mg_d[((size_t)j * kScanWaves + wave) * a.k + r] = top[j].d[e];
          mg_i[((size_t)j * kScanWaves + wave) * a.k + r] = top[j].i[e];
        }
      }
    }
    __syncthreads();
    if (wave < cnt_g && ins[wave] != 0u) {
      const int j = wave;
      wave_top<E> fin;
      fin.init();
      float kd    = INFINITY;
      uint32_t ki = 0xffffffffu;
      const int n = kScanWaves * (int)a.k;
      static_assert(E == 1, "k <= 64: one chunk of lanes per wave list");
      for (uint32_t wm = ins[j]; wm != 0u; wm &= wm - 1u) {  // the lists of the waves that inserted, ascending wave
        const int b0 = (__ffs((int)wm) - 1) * (int)a.k;
        float md    = INFINITY;
        uint32_t mi = 0xffffffffu;
        if (lane < (int)a.k) { md = mg_d[(size_t)j * n + b0 + lane]; mi = mg_i[(size_t)j * n + b0 + lane]; }
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
      const uint32_t pj = pid[g * EQ + j];
      const size_t o    = (size_t)pj * a.k;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int r = e * 64 + lane;
        if (r < (int)a.k) {
          const bool ok  = fin.i[e] != 0xffffffffu;
          a.out_d[o + r] = ok ? fin.d[e] : FLT_MAX;
          a.out_i[o + r] = ok ? base_row + fin.i[e] : 0xffffffffu;
        }
      }
      if (lane == 0 && kd < INFINITY) atomicMin(&a.query_kth[pj / a.n_probes], float_to_key(kd));
    }
    stat_phase(ST_MERGE);
  }
  if (threadIdx.x == 0) *next_slot = next_hdr;
}

template <typename LutT, typename AccT, int EQ, int NG, int E>
__global__ __launch_bounds__(kScanThreads) void pq_scan2_kernel(scan_args a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using entry_t = typename lut_acc<LutT, AccT, EQ>::entry_t;
  const uint32_t item0   = a.item_begin ? *a.item_begin : 0u;
  const uint32_t n_items = *a.item_end - item0;
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t chunk = (n_items + 7u) / 8u;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();  // see cm_lut
  f32x2_t pq2[4][4];  // this thread's slice of the codebook, in registers for the whole launch (pq_regs_load2)
  pq_regs_load2(pq2, a.pq_centers);
  const scan2_layout lay(cm_lut<entry_t>::bytes(), EQ * NG, a.rot_dim);
  work_item* sh_item       = reinterpret_cast<work_item*>(smem + lay.slots);
  const uint32_t share0    = min(n_items, xcd * chunk);
  const uint32_t share_len = min(chunk, n_items - share0);
  const work_item* share   = a.items + item0 + share0;
  uint32_t* ticket         = a.xcd_ticket + xcd * 32;
  if (threadIdx.x == 0) {
    const uint32_t t = atomicAdd(ticket, 1u);
    sh_item[0]       = t < share_len ? share[t] : work_item{0u, 0u, 0u, 0xffffffffu};
  }
  __syncthreads();
  for (int buf = 0;; buf ^= 1) {
    const work_item cur = sh_item[buf];
    if (cur.pad == 0xffffffffu) break;  // workgroup-uniform
    uint32_t next_ticket = 0xffffffffu;
    if (threadIdx.x == 0) next_ticket = atomicAdd(ticket, 1u);
    pq_scan2_item<LutT, AccT, EQ, NG, E>(a, cur, smem, pq2, share, share_len, next_ticket, buf ^ 1);
    __syncthreads();
  }
}

template <typename LutT, typename AccT, int EQ, int NG>
void launch_scan2(resources& res, const scan_args& a, unsigned grid)
{
  using entry_t = typename lut_acc<LutT, AccT, EQ>::entry_t;
  const size_t smem = scan2_layout(cm_lut<entry_t>::bytes(), EQ * NG, a.rot_dim).total;
  CUVS_EXPECTS(smem <= res.lds_per_block, "pq_scan2_kernel: %zu bytes of LDS", smem);
  auto kern = pq_scan2_kernel<LutT, AccT, EQ, NG, 1>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem));
  profile_begin(res, "pq_scan_kernel");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanThreads), smem, res.stream, a);
  profile_end(res, "pq_scan_kernel");
  HIP_TRY(hipGetLastError());
}

template <typename LutT, typename AccT, int QPB>
size_t scan_smem_bytes(const ivf_pq_index& idx, int k)
{
  using entry_t = typename lut_acc<LutT, AccT, QPB>::entry_t;
  const bool fast4 = idx.pq_bits == 8 && idx.pq_dim == 64;  // code-major LUT: 8 bytes of padding per code row
  return scan_layout(fast4 ? cm_lut<entry_t>::bytes() : (size_t)idx.pq_dim * idx.pq_book * sizeof(entry_t), QPB,
                     idx.rot_dim, (uint32_t)k).total;
}

template <typename LutT, typename AccT, int QPB, bool FAST4, int E, bool ALL = false, bool GLUT = false>
void launch_scan(resources& res, const scan_args& a, size_t smem, unsigned grid)
{
  auto kern = pq_scan_kernel<LutT, AccT, QPB, FAST4, E, ALL, GLUT>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem));
  profile_begin(res, "pq_scan_kernel");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanThreads), smem, res.stream, a);
  profile_end(res, "pq_scan_kernel");
  HIP_TRY(hipGetLastError());
}

template <typename LutT, typename AccT, int QPB>
void launch_scan_qpb(resources& res, const scan_args& a, size_t smem, unsigned grid, bool bits8, bool big_k)
{
  if (a.all_scores != nullptr) {
    if (bits8) launch_scan<LutT, AccT, QPB, true, 1, true>(res, a, smem, grid);
    else       launch_scan<LutT, AccT, QPB, false, 1, true>(res, a, smem, grid);
  } else if (bits8) {
    if (big_k) launch_scan<LutT, AccT, QPB, true, 4>(res, a, smem, grid);
    else       launch_scan<LutT, AccT, QPB, true, 1>(res, a, smem, grid);
  } else {
    if (big_k) launch_scan<LutT, AccT, QPB, false, 4>(res, a, smem, grid);
    else       launch_scan<LutT, AccT, QPB, false, 1>(res, a, smem, grid);
  }
}

// LUT in global memory: the interleave is not bounded by the LDS any more, so one 8-byte load serves QPB queries
template <typename LutT, typename AccT, int QPB>
void launch_scan_glut(resources& res, const scan_args& a, size_t smem, unsigned grid, bool big_k)
{
  if (a.all_scores != nullptr) launch_scan<LutT, AccT, QPB, false, 1, true, true>(res, a, smem, grid);
  else if (big_k)              launch_scan<LutT, AccT, QPB, false, 4, false, true>(res, a, smem, grid);
  else                         launch_scan<LutT, AccT, QPB, false, 1, false, true>(res, a, smem, grid);
}

// ---- reduced-precision coarse search (search_params.coarse_search_dtype; ivf_pq_search.cuh:171-340, :995-1017)
// The reference converts queries, centres (with their |c|^2 column) and the rotation matrix to half or int8 and runs
// the coarse GEMM and the rotation GEMM in that type. Here the operands are rounded the same way and kept as
// floats - products of two halves / two int8 are exact in fp32 - so the fp32 MFMA kernels produce: int8 the exact
// integer result (bit-identical to any int8 GEMM), fp16 the fp32-accumulated dot product, rounded to half like the
// reference's half output (cuBLAS's accumulation order is not specified, so fp16 is "the same rounding", not bit-pinned).
__global__ void round_to_half_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int64_t ld_in,
                                     int64_t cols)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)(_Float16)in[(i / cols) * ld_in + i % cols];
}
__device__ inline float to_int8_value(float v)  // static_cast<int8_t>(clamp(v, -128, 127)): truncation toward zero
{
  return truncf(fmaxf(-128.0f, fminf(127.0f, v)));
}
__global__ void scale_to_int8_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int64_t ld_in,
                                     int64_t cols)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = to_int8_value(in[(i / cols) * ld_in + i % cols] * 128.0f);
}
// the y / z columns of centers_int8 times the query's constant columns (ivf_pq_index.cu:674-732, ivf_pq_search.cuh:205-215)
__global__ void int8_norm_term_kernel(const float* __restrict__ norms, int n_lists, int m, bool l2, float* __restrict__ out)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_lists) return;
  const float c  = 64.0f / (float)(m - 1);
  const float y  = fmaxf(-128.0f, fminf(127.0f, norms[j] * c));
  const float z  = to_int8_value((y - roundf(y)) * 128.0f);
  const float yr = truncf(roundf(y));
  // query columns: (1 - m) at `dim`, then (m - 1) times norm_factor (-128 for L2, 0 for inner product / cosine)
  out[j] = z * (float)(1 - m) + (l2 ? (float)(m - 1) * yr * -128.0f : 0.0f);
}
// qc_distances of the reduced-precision GEMM from the exact dot products: half: half(alpha * (dot - 0.5 |c|^2_h));
// int8: alpha * (dot + norm term), alpha = -2 (L2) / -1 (inner product, cosine)
__global__ void coarse_finish_kernel(float* __restrict__ d, int64_t nq, int n_lists, const float* __restrict__ term,
                                     float alpha, bool half_out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * n_lists) return;
  float v = d[i];
  if (term != nullptr) v = v + (half_out ? -0.5f * term[i % n_lists] : term[i % n_lists]);
  v = alpha * v;
  d[i] = half_out ? (float)(_Float16)v : v;
}

static std::mutex g_coarse_cache_mu;  // the reduced-precision copies of an index's centres / rotation, built on first use

// n_probes clusters closest to each query (select_clusters, ivf_pq_search.cuh:60-168 fp32, :171-340 int8 / half).
// qf: [nq, dim] fp32 queries (normalised for cosine); qc: the copy in the coarse type (== qf for fp32)
void select_clusters(resources& res, const ivf_pq_index& idx, const float* qf, int64_t nq, uint32_t n_probes,
                     uint32_t* probes, int coarse_dtype, float* qc)
{
  dev_buf<float> pd(res, (size_t)nq * n_probes);
  const bool ip = idx.metric == M_InnerProduct || idx.metric == M_CosineExpanded;  // cosine: unit queries x unit centres
  if (coarse_dtype != 0) {
    dev_buf<float> dist(res, (size_t)nq * idx.n_lists);
    const bool half_t  = coarse_dtype == 2;
    const int64_t n_el = (int64_t)idx.n_lists * idx.dim;
    dev_buf<float>& cc = half_t ? idx.coarse_centers_h : idx.coarse_centers_i8;
    dev_buf<float>& ct = half_t ? idx.coarse_norms_h : idx.coarse_normterm_i8;
    {
      // Derived copies of the index, made by the first search that asks for them - possibly while other host threads search the
      // same index through handles of their own (benchmark.hpp:296-307). A copy is PUBLISHED (the member's pointer set) under the
      // lock and only after the kernels that fill it have run: a thread that finds the pointer set may read it from any stream
      // (tests/test_concurrent_search_gpu.py caught three threads reading the fp16 centres while the fourth's fill was still queued).
      std::lock_guard<std::mutex> lock(g_coarse_cache_mu);
      if (cc.data() == nullptr) {
        auto cc_new = dev_buf<float>::persistent((size_t)n_el);
        auto ct_new = dev_buf<float>::persistent(idx.n_lists);
        if (half_t) {
          hipLaunchKernelGGL(round_to_half_kernel, dim3(nblk(n_el, 256)), dim3(256), 0, res.stream, idx.centers.data(),
                             cc_new.data(), n_el, (int64_t)idx.dim_ext, (int64_t)idx.dim);
          hipLaunchKernelGGL(round_to_half_kernel, dim3(nblk(idx.n_lists, 256)), dim3(256), 0, res.stream,
                             idx.center_norms.data(), ct_new.data(), (int64_t)idx.n_lists, (int64_t)1, (int64_t)1);
        } else {
          hipLaunchKernelGGL(scale_to_int8_kernel, dim3(nblk(n_el, 256)), dim3(256), 0, res.stream, idx.centers.data(),
                             cc_new.data(), n_el, (int64_t)idx.dim_ext, (int64_t)idx.dim);
          const int m = (int)(round_up((int64_t)idx.dim + 2, 16) - idx.dim);  // dim_ext_int8 - dim (ivf_pq_index.cu:668)
          hipLaunchKernelGGL(int8_norm_term_kernel, dim3(nblk(idx.n_lists, 256)), dim3(256), 0, res.stream,
                             idx.center_norms.data(), (int)idx.n_lists, m, !ip, ct_new.data());
        }
        sync(res);
        ct = std::move(ct_new);
        cc = std::move(cc_new);
      }
    }
    const int64_t nqe = nq * idx.dim;
    if (half_t) hipLaunchKernelGGL(round_to_half_kernel, dim3(nblk(nqe, 256)), dim3(256), 0, res.stream, qf, qc, nqe, (int64_t)idx.dim, (int64_t)idx.dim);
    else        hipLaunchKernelGGL(scale_to_int8_kernel, dim3(nblk(nqe, 256)), dim3(256), 0, res.stream, qf, qc, nqe, (int64_t)idx.dim, (int64_t)idx.dim);
    // half: the norm column only takes part for L2 (norm_factor 0 otherwise); int8: the z column always does
    const float* term = (half_t && ip) ? nullptr : ct.data();
    if (res.tune.coarse_lowp != 0) {
      // the products on the matrix cores of the coarse type, the output arithmetic in the kernel's epilogue
      const size_t row_b = (size_t)coarse_lowp_ksteps(!half_t, idx.dim) * 32;
      dev_buf<uint32_t>& cp = half_t ? idx.coarse_pack_h : idx.coarse_pack_i8;
      {
        std::lock_guard<std::mutex> lock(g_coarse_cache_mu);
        if (cp.data() == nullptr) {
          auto cp_new = dev_buf<uint32_t>::persistent((size_t)idx.n_lists * row_b / 4);
          coarse_lowp_pack(res, !half_t, idx.centers.data(), idx.n_lists, idx.dim_ext, idx.dim, cp_new.data());
          sync(res);
          cp = std::move(cp_new);
        }
      }
      dev_buf<uint32_t> qp(res, (size_t)nq * row_b / 4);
      coarse_lowp_pack(res, !half_t, qf, nq, idx.dim, idx.dim, qp.data());
      coarse_lowp_distances(res, !half_t, qp.data(), nq, cp.data(), idx.n_lists, idx.dim, term, ip ? -1.0f : -2.0f, dist.data(),
                            idx.n_lists);
    } else {
      pairwise_distance<float, float>(res, qc, nq, idx.dim, cc.data(), idx.n_lists, idx.dim, idx.dim, nullptr, nullptr,
                                      M_InnerProduct, dist.data(), idx.n_lists);
      hipLaunchKernelGGL(coarse_finish_kernel, dim3(nblk(nq * idx.n_lists, 256)), dim3(256), 0, res.stream, dist.data(), nq,
                         (int)idx.n_lists, term, ip ? -1.0f : -2.0f, half_t);
    }
    select_k<uint32_t, uint32_t>(res, dist.data(), nullptr, nq, idx.n_lists, idx.n_lists, (int)n_probes, pd.data(),
                                 probes, true);
    return;
  }
  // fp32 coarse search. Common shapes: the distance tile writes rows in GROUPED layout + the best key of every 16-centre
  // group, and the selection reads the keys and only the groups that can hold one of the n_probes nearest (ops.hpp:
  // pairwise_distance_grouped / select_k_grouped) - the values, their order and the tie rule are those of the plain form below
  if (res.tune.coarse_grouped != 0 && select_k_grouped_ok(idx.n_lists, (int)n_probes)) {
    const int64_t ldo = round_up((int64_t)idx.n_lists, 128);
    dev_buf<float> gdist(res, (size_t)nq * ldo);
    dev_buf<uint32_t> gkeys(res, (size_t)nq * (ldo / 16));
    dev_buf<float> qn(res, ip ? 0 : nq);
    if (!ip) row_norms<float>(res, qf, nq, idx.dim, idx.dim, qn.data(), false);
    if (pairwise_distance_grouped(res, qf, nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim_ext, idx.dim, ip ? nullptr : qn.data(),
                                  ip ? nullptr : idx.center_norms.data(), ip ? (int)M_InnerProduct : (int)M_L2Expanded, gdist.data(), ldo,
                                  gkeys.data(), ldo / 16)) {
      select_k_grouped(res, gdist.data(), ldo, gkeys.data(), ldo / 16, nq, idx.n_lists, (int)n_probes, pd.data(), probes, !ip);
      return;
    }
  }
  dev_buf<float> dist(res, (size_t)nq * idx.n_lists);
  if (ip) {
    pairwise_distance<float, float>(res, qf, nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim_ext, idx.dim,
                                    nullptr, nullptr, M_InnerProduct, dist.data(), idx.n_lists);
    select_k<uint32_t, uint32_t>(res, dist.data(), nullptr, nq, idx.n_lists, idx.n_lists, (int)n_probes, pd.data(),
                                 probes, false);
  } else {
    dev_buf<float> qn(res, nq);
    row_norms<float>(res, qf, nq, idx.dim, idx.dim, qn.data(), false);
    pairwise_distance<float, float>(res, qf, nq, idx.dim, idx.centers.data(), idx.n_lists, idx.dim_ext, idx.dim,
                                    qn.data(), idx.center_norms.data(), M_L2Expanded, dist.data(), idx.n_lists);
    select_k<uint32_t, uint32_t>(res, dist.data(), nullptr, nq, idx.n_lists, idx.n_lists, (int)n_probes, pd.data(),
                                 probes, true);
  }
}

// rotation of the queries in the coarse type (ivf_pq_search.cuh:995-1017): R rounded like the queries, fp32 output;
// int8: alpha = 1 / 128 / 128
__global__ void scale_kernel(float* __restrict__ x, int64_t n, float a)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= a;
}
void rotate_queries(resources& res, const ivf_pq_index& idx, const float* qf, const float* qc, int64_t nq, int coarse_dtype,
                    float* rot_q)
{
  if (coarse_dtype == 0) {
    pairwise_distance<float, float>(res, qf, nq, idx.dim, idx.rotation.data(), idx.rot_dim, idx.dim, idx.dim, nullptr,
                                    nullptr, M_InnerProduct, rot_q, idx.rot_dim);
    return;
  }
  const bool half_t  = coarse_dtype == 2;
  dev_buf<float>& rr = half_t ? idx.coarse_rot_h : idx.coarse_rot_i8;
  const int64_t n_el = (int64_t)idx.rot_dim * idx.dim;
  {
    std::lock_guard<std::mutex> lock(g_coarse_cache_mu);  // (published once filled: see select_clusters)
    if (rr.data() == nullptr) {
      auto rr_new = dev_buf<float>::persistent((size_t)n_el);
      if (half_t) hipLaunchKernelGGL(round_to_half_kernel, dim3(nblk(n_el, 256)), dim3(256), 0, res.stream, idx.rotation.data(), rr_new.data(), n_el, (int64_t)idx.dim, (int64_t)idx.dim);
      else        hipLaunchKernelGGL(scale_to_int8_kernel, dim3(nblk(n_el, 256)), dim3(256), 0, res.stream, idx.rotation.data(), rr_new.data(), n_el, (int64_t)idx.dim, (int64_t)idx.dim);
      sync(res);
      rr = std::move(rr_new);
    }
  }
  pairwise_distance<float, float>(res, qc, nq, idx.dim, rr.data(), idx.rot_dim, idx.dim, idx.dim, nullptr, nullptr,
                                  M_InnerProduct, rot_q, idx.rot_dim);
  if (!half_t) hipLaunchKernelGGL(scale_kernel, dim3(nblk(nq * idx.rot_dim, 256)), dim3(256), 0, res.stream, rot_q, nq * idx.rot_dim, 1.0f / 128.0f / 128.0f);
}

// Two-stream schedule: the head phase scans ONE pair per query - its nearest probe - so its work items need no grouping:
// item q = (list probes[q, 0], pair q n_probes), count 0 when that list lives on another rank of a list-sharded index.
// The head kernel can start as soon as the probes are known, while the grouping of all pairs, the work units and the B
// operands of the tail phase are made on the helper stream.
__global__ void head_items_kernel(const uint32_t* __restrict__ probes, int64_t nq, uint32_t n_probes, uint32_t shard_world,
                                  uint32_t shard_rank, const int32_t* __restrict__ owner, work_item* __restrict__ items,
                                  uint32_t* __restrict__ pairs, uint32_t* __restrict__ n_items)
{
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0) *n_items = (uint32_t)nq;
  if (q >= nq) return;
  const uint32_t L = probes[q * n_probes];
  const bool mine  = shard_world <= 1 || (owner != nullptr ? (uint32_t)owner[L] == shard_rank : L % shard_world == shard_rank);
  items[q] = work_item{L, (uint32_t)q, mine ? 1u : 0u, 0u};
  pairs[q] = (uint32_t)(q * n_probes);
}

// the first `seg` slots of every query's candidate row (the head pairs' segments) start out "nothing found"
__global__ void init_head_rows_kernel(float* __restrict__ cand_d, uint32_t* __restrict__ cand_i, int64_t nq, int64_t row_len, uint32_t seg)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * (int64_t)seg) return;
  const int64_t o = (t / seg) * row_len + t % seg;
  cand_d[o] = FLT_MAX;
  cand_i[o] = 0xffffffffu;
}

// flat row -> source id; distance fix-ups (ivf_common.cuh:114-171 postprocess_neighbors, :176-253)
__global__ void postprocess_kernel(const uint32_t* __restrict__ pos, const float* __restrict__ d_in, int64_t n,
                                   const int64_t* __restrict__ indices, int metric, float scale2,
                                   int64_t* __restrict__ neighbors, float* __restrict__ distances)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t p   = pos[i];
  neighbors[i] = p == 0xffffffffu ? INT64_MAX : indices[p];  // kOutOfBoundsRecord, ivf_common.cuh:31
  float d      = d_in[i];
  if (p == 0xffffffffu) {
    d = FLT_MAX;
  } else if (metric == M_InnerProduct) {
    d = -d * scale2;
  } else if (metric == M_CosineExpanded) {
    d = 1.0f + d;  // the scan minimised -cos of unit vectors
  } else if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) {
    d = sqrtf(d * scale2);
  } else {
    d = d * scale2;
  }
  distances[i] = d;
}

}  // namespace

// rows and non-empty lists over all ranks of a list-sharded index: ONE in-place all-gather of three words per rank, made by
// the first search after cuvsAmdIvfPqSetShardComm (attaching is collective, so every rank is in the same state)
static void shard_exchange_stats(resources& res, const ivf_pq_index& idx)
{
  // (threads sharing the index: one of them exchanges - the all-gather is collective ACROSS RANKS, every rank's first searcher
  // makes it once - the others find the numbers in place)
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (idx.shard_stats_valid) return;
  uint64_t rows = 0;
  uint32_t lists = 0;
  for (uint32_t v : idx.h_list_sizes) { rows += v; lists += v != 0u; }
  const size_t world = (size_t)std::max(1, idx.shard_world);
  std::vector<uint32_t> h(3 * world, 0u);
  h[3 * idx.shard_rank + 0] = (uint32_t)rows; h[3 * idx.shard_rank + 1] = (uint32_t)(rows >> 32); h[3 * idx.shard_rank + 2] = lists;
  dev_buf<uint32_t> d(res, h.size());
  copy_async(res, d.data(), h.data(), h.size() * sizeof(uint32_t));
  sync(res);  // (h is pageable: the copy must have read it before the vector is reused below)
  shard_allgather_inplace_u32(res, idx.shard_comm, d.data(), 3);
  h = to_host(res, d.data(), h.size());
  idx.shard_global_rows = 0; idx.shard_global_lists = 0;
  for (size_t r = 0; r < world; ++r) {
    idx.shard_global_rows += (uint64_t)h[3 * r] | ((uint64_t)h[3 * r + 1] << 32);
    idx.shard_global_lists += h[3 * r + 2];
  }
  idx.shard_stats_valid = true;
}

void ivf_pq_search(resources& res, const ivf_pq_search_params& p, const ivf_pq_index& idx, const void* queries,
                   elem_t et, int64_t n_queries, int k, int64_t* neighbors, float* distances, const uint32_t* filter_bits)
{
  CUVS_EXPECTS(k > 0, "parameter `k` in top-k must be positive.");
  // (a list shard may hold fewer than k rows - its peers pad: every rank must reach the collectives of this call)
  CUVS_EXPECTS((int64_t)k <= idx.size || idx.shard_world > 1,
               "parameter `k` (%d) in top-k must not be larger that the total size of the index (%ld)", k,
               (long)idx.size);
  CUVS_EXPECTS(p.n_probes > 0, "n_probes (number of clusters to probe in the search) must be positive.");
  CUVS_EXPECTS(p.internal_distance_dtype == 0 || p.internal_distance_dtype == 2,
               "internal_distance_dtype must be either CUDA_R_16F or CUDA_R_32F");
  CUVS_EXPECTS(p.lut_dtype == 0 || p.lut_dtype == 2 || p.lut_dtype == 8 || p.lut_dtype == 3,
               "lut_dtype must be CUDA_R_16F, CUDA_R_32F, CUDA_R_8U or CUDA_R_8I");
  CUVS_EXPECTS(p.coarse_search_dtype == 0 || p.coarse_search_dtype == 2 || p.coarse_search_dtype == 3,
               "Unsupported coarse_search_dtype (only CUDA_R_32F, CUDA_R_16F, and CUDA_R_8I are supported)");
  CUVS_EXPECTS(!idx.dtype_known || et == idx.dtype, "queries dtype differs from the index dtype");
  if (n_queries == 0) return;
  if (idx.shard_comm != nullptr && !idx.shard_stats_valid) shard_exchange_stats(res, idx);
  const uint32_t n_probes = std::min<uint32_t>(p.n_probes, idx.n_lists);
  // fp8 LUT (the reference's fp_8bit<5, signed>): entries are rounded through that type and kept in the score type
  const bool lut_fp8      = p.lut_dtype == 8 || p.lut_dtype == 3;
  const bool acc_half     = p.lut_dtype != 0 && p.internal_distance_dtype == 2;
  const bool lut_half     = lut_fp8 ? acc_half : p.lut_dtype != 0;
  const bool bits8        = idx.pq_bits == 8 && idx.pq_dim == 64;  // FAST4 path: 4 full 16-byte chunks
  const bool large_k      = k > 256;  // beyond the register top lists: non-fused path (ivf_common.hpp)
  const bool big_k        = k > 64 && !large_k;
  const int k_scan        = large_k ? 1 : k;  // top-list length the scan kernel is launched with
  const size_t lds_cap    = 160 * 1024;

  // choose the widest interleave (queries per work item) whose LUT fits the 160 KiB LDS
  int qpb = 0;
  size_t smem = 0;
  if (!lut_half) {
    if ((smem = scan_smem_bytes<float, float, 2>(idx, k_scan)) <= lds_cap) qpb = 2;
    else if ((smem = scan_smem_bytes<float, float, 1>(idx, k_scan)) <= lds_cap) qpb = 1;
  } else {
    if ((smem = scan_smem_bytes<__half, float, 4>(idx, k_scan)) <= lds_cap) qpb = 4;
    else if ((smem = scan_smem_bytes<__half, float, 2>(idx, k_scan)) <= lds_cap) qpb = 2;
    else if ((smem = scan_smem_bytes<__half, float, 1>(idx, k_scan)) <= lds_cap) qpb = 1;
  }
  // no fit: the LUT goes to global memory (L2), as the reference does when its LUT exceeds shared memory
  const bool glut = qpb == 0;
  if (glut) {
    qpb  = lut_half ? 4 : 2;  // 8-byte entries
    smem = scan_layout(0, qpb, idx.rot_dim, (uint32_t)k_scan).total;
    CUVS_EXPECTS(smem <= lds_cap, "ivf_pq::search: rot_dim %u / k %d do not fit 160 KiB of LDS", idx.rot_dim, k_scan);
  }

  // rows of the n_probes largest lists: the width of the score matrix of the non-fused path. With a shard communicator
  // attached the batch size derived from it must be the same on every rank (the ranks issue one probe all-gather and
  // one bound all-reduce per batch: different batch counts would hang the collectives), so the maximum over the
  // ranks is used (one 4-byte all-reduce, only on this rarely taken path).
  size_t largest_total = large_k ? largest_lists_total(idx.h_list_sizes, n_probes) : 0;
  if (large_k && idx.shard_comm != nullptr) {
    dev_buf<uint32_t> key(res, 1);
    const uint32_t mine = ~(uint32_t)std::min<size_t>(largest_total, 0xfffffffeu);  // min over ~x = max over x
    copy_async(res, key.data(), &mine, sizeof(mine));
    shard_allreduce_min_u32(res, idx.shard_comm, key.data(), 1);
    largest_total = (size_t)~to_host(res, key.data(), 1)[0];
  }
  // The wide matrix-core path (ivf_pq_wide.hip): shapes pq_filter4_kernel does not decode (rot_dim beyond 256, pq_len not a power
  // of two) and searches whose k is too large a fraction of ONE list for its bound to prune (pq3_bound_useful) - the bound then
  // comes from the union of `wheads` head lists. Not on a list shard, batches large enough for a head phase. (A pre-filter is applied
  // by the emit pass - rejected rows get the value -inf, so the k rows whose exact scores make the bound are admissible ones - and by
  // the re-score.)
  uint32_t wheads = 0;
  if (!large_k && n_probes > 8 && n_queries >= 256 && res.tune.pq_scan3 != 0 && res.tune.pq_wide != 0 && res.tune.pq_head_probes < 0 &&
      idx.shard_world <= 1 && idx.shard_comm == nullptr && pqw_supported(idx, k) &&
      !(pq3_supported(idx, k) && pq3_bound_useful(idx, k) && ((idx.pq_len == 2 && idx.codebook_kind == 0) || res.tune.pq_filter4 != 0))) {
    wheads = res.tune.pq_wide_heads > 0 ? std::min<uint32_t>((uint32_t)res.tune.pq_wide_heads, n_probes / 2) : pqw_heads(idx, k, n_probes);
    if (wheads > 0 && !pqw_ready(res, idx)) wheads = 0;  // (no room for the decoded rows)
  }
  const bool usew = wheads > 0;
  if (res.tune.scan_debug & 1024)
    fprintf(stderr, "[pq_wide] heads %u (supported %d, matrix-core tail of the narrow shapes %d, probes %u, queries %ld, k %d)\n", wheads,
            (int)pqw_supported(idx, k), (int)(pq3_supported(idx, k) && pq3_bound_useful(idx, k)), n_probes, (long)n_queries, k);
  uint32_t max_list_len = 0;
  for (uint32_t v : idx.h_list_sizes) max_list_len = std::max(max_list_len, v);
  const uint32_t w_ldx = usew ? (uint32_t)round_up((int64_t)max_list_len + 64, 64) : 0u;
  // batch of queries per pass (reference: max_internal_batch_size bounds the coarse batch, :814-857)
  int64_t max_batch = std::max<uint32_t>(1, p.max_internal_batch_size);
  {
    // keep the coarse distance matrix and the candidate buffers inside the workspace budget
    int64_t per_q = (int64_t)idx.n_lists * 4 + (int64_t)n_probes * k_scan * 8 + (int64_t)idx.rot_dim * 4 + idx.dim * 4;
    if (large_k) per_q += (int64_t)largest_total * 8;
    // the matrix-core tail phase's buffers, per (query, probe) pair: fp16 B operand, threshold, probe ranks of the pool, >= 16
    // survivor entries, a fallback work item, two unit descriptors' share, norms and grouping scratch of the two-stream schedule
    if (!large_k && (pq3_supported(idx, k) || usew) && res.tune.pq_scan3 != 0)
      per_q += (int64_t)n_probes * ((int64_t)idx.rot_dim * 2 + (int64_t)k * 4 + 128 + 16 + 4 + 16 + 8);
    if (usew) per_q += (int64_t)wheads * ((int64_t)w_ldx * 4 + 32) + (int64_t)k * (8 + 32 * 8);  // values of the head lists' rows, the k best
    int64_t fit   = std::max<int64_t>(1, (int64_t)res.ivf_batch_limit / per_q);
    max_batch     = balanced_batch(n_queries, std::min(max_batch, fit));  // (the same on every rank of a list shard: same inputs)
  }
  const int64_t bs_alloc = std::min<int64_t>(max_batch, n_queries);
  const int64_t n_pairs_max = bs_alloc * n_probes;
  dev_buf<float> qf(res, (size_t)bs_alloc * idx.dim);
  dev_buf<float> rot_q(res, (size_t)bs_alloc * idx.rot_dim);
  dev_buf<float> qc(res, p.coarse_search_dtype != 0 ? (size_t)bs_alloc * idx.dim : 0);  // queries in the coarse type
  dev_buf<uint32_t> probes(res, (size_t)n_pairs_max + (size_t)std::max(1, idx.shard_world) * n_probes);  // + slice padding
  // Two-phase schedule: the `head` nearest probes of every query are scanned first (labels 0..n_lists-1), the
  // rest afterwards (labels n_lists..2 n_lists-1). After the head phase each query's k-th bound (query_kth) is
  // already close to final, which is what makes the early stop in the scan loop bite. Results do not depend on
  // the order in which pairs are scanned.
  // (measured at 100M x 128, n_probes 128: batch 1000 3.2 vs 4.0 ms with the head phase, batch 100 1.7 vs 1.3 ms without:
  // a second launch and a twice as long label range only pay off once the batch is large)
  uint32_t head = (n_probes > 8 && n_queries >= 256 && !large_k) ? 1u : 0u;
  if (res.tune.pq_head_probes >= 0) head = std::min<uint32_t>((uint32_t)res.tune.pq_head_probes, n_probes);
  // signed LUT entries: no early stop in the LUT scan kernels - unless the tail phase runs on the matrix-core filter,
  // which needs no non-negativity (a full-score bound): then the head phase supplies its bounds as for L2
  const bool pq3_ok = !large_k && pq3_supported(idx, k) && pq3_bound_useful(idx, k) && res.tune.pq_scan3 != 0 && res.tune.pq_head_probes != 0 &&
                      ((idx.pq_len == 2 && idx.codebook_kind == 0) || res.tune.pq_filter4 != 0);
  if ((idx.metric == M_InnerProduct || idx.metric == M_CosineExpanded) && !pq3_ok) head = 0;
  if (usew) head = wheads;
  const bool sharded      = idx.shard_world > 1;  // list-sharded index: foreign probes go to a bucket that is never scanned
  const uint32_t n_ranges = head > 0 ? 2 * idx.n_lists : idx.n_lists;
  const uint32_t n_labels = n_ranges + (sharded ? 1u : 0u);
  dev_buf<uint32_t> sorted_pairs(res, (size_t)n_pairs_max), pair_off(res, n_labels + 1), item_off(res, n_labels + 1);
  dev_buf<uint32_t> phase_labels(res, (head > 0 || sharded) ? (size_t)n_pairs_max : 0);
  // (with the matrix-core tail phase the head pairs become single-pair items: one item per head pair on top of the
  // qpb-pair items of the tail labels - the bound used to leave them out and the item array ran over by (queries x head)
  // items whenever a batch was large against the number of lists: found in round 4 by a 1500-query x 24-list test)
  const int64_t max_items = n_pairs_max / qpb + bs_alloc * (int64_t)head + n_labels + 1;
  dev_buf<work_item> items(res, (size_t)max_items);
  const size_t scores_ld = largest_total;
  dev_buf<float> cand_d(res, large_k ? (size_t)bs_alloc * scores_ld : (size_t)n_pairs_max * k);
  dev_buf<uint32_t> cand_i(res, large_k ? (size_t)bs_alloc * scores_ld : (size_t)n_pairs_max * k);
  dev_buf<uint32_t> pair_seg(res, large_k ? (size_t)n_pairs_max : 0);
  dev_buf<float> top_d(res, (size_t)bs_alloc * k);
  dev_buf<uint32_t> top_i(res, (size_t)bs_alloc * k);
  dev_buf<uint32_t> query_kth(res, (size_t)bs_alloc);
  dev_buf<uint32_t> tickets(res, 4 * 8 * 32);
  // warm-bounds phase on the matrix cores (ivf_pq_scan3.hip): decode + MFMA filter, exact re-score of the survivors
  const bool metric_ip = idx.metric == M_InnerProduct || idx.metric == M_CosineExpanded;
  // (pq_len other than 2 is decoded by pq_filter4_kernel only)
  const bool use3 = !usew && head > 0 && !large_k && pq3_supported(idx, k) && pq3_bound_useful(idx, k) && res.tune.pq_scan3 != 0 && ((idx.pq_len == 2 && idx.codebook_kind == 0) || res.tune.pq_filter4 != 0);
  const bool use3x = use3 || usew;  // the buffers both matrix-core paths need
  uint32_t unit_rows = 0;
  const size_t max_units = usew ? pq3_max_units(idx, n_pairs_max, &unit_rows, false) : use3 ? pq3_max_units(idx, n_pairs_max, &unit_rows, res.tune.pq_filter4 != 0 && (idx.metric != M_InnerProduct || idx.pq_len != 2 || idx.codebook_kind != 0)) : 0;
  uint32_t surv_cap = use3x ? (uint32_t)std::min<int64_t>(std::max<int64_t>(n_pairs_max * 16, 1 << 22), 1 << 28) : 0u;
  // (the wide path at large k: a query has at least k survivors by construction and a few times k with the margins of fp16 scores -
  // 1200 per query measured at k = 256 of 1.4 k-row lists - and the head pairs' rows within the bound join them)
  if (usew) surv_cap = (uint32_t)std::min<int64_t>(std::max<int64_t>((int64_t)surv_cap, bs_alloc * (int64_t)k * 32), 1 << 28);
  if (use3x && res.tune.pq3_surv_cap > 0) surv_cap = (uint32_t)res.tune.pq3_surv_cap;
  dev_buf<uint32_t> cand_r(res, use3x ? (size_t)n_pairs_max * k : 0), qstate(res, use3x ? (size_t)4 * bs_alloc + 8 + pq3_regions(res) : 0);
  dev_buf<uint32_t> unit_off(res, use3x ? (size_t)idx.n_lists + 1 : 0);
  dev_buf<uint2> surv(res, surv_cap);
  dev_buf<uint4> units3(res, 2 * max_units);  // 32-byte unit descriptors
  const uint32_t overflow_cap = use3x ? (res.tune.pq3_surv_cap > 0 ? (uint32_t)res.tune.pq3_surv_cap : (1u << 22)) : 0u;
  dev_buf<uint4> overflow3(res, (size_t)2 * overflow_cap);
  dev_buf<work_item> fb_items(res, use3x ? (size_t)n_pairs_max : 0);
  // pq_filter4_kernel serves L2 and cosine; unnormalised inner products (loose margins: ~8x the survivors per pair) keep
  // pq_filter_kernel, whose per-lane survivor loop is cheaper at that rate (C3 shape: 3.9 vs 6.7 ms)
  const bool use_f4 = use3 && res.tune.pq_filter4 != 0 && (idx.metric != M_InnerProduct || idx.pq_len != 2 || idx.codebook_kind != 0);
  // fp16 B operands of the tail pairs (the wide path: in blocks of 32 pairs per list, every list's last block padded)
  dev_buf<uint4> bq3(res, (use_f4 || usew) ? ((size_t)n_pairs_max + (usew ? (size_t)32 * (idx.n_lists + 1) : 0)) * (idx.rot_dim / 8) : 0);
  dev_buf<float> thr3(res, (use_f4 || usew) ? (size_t)n_pairs_max : 0);
  // the wide path's bound-only head phase: values of every (head pair, row), the k best of every query, the head pairs' thresholds
  dev_buf<float> w_x(res, usew ? (size_t)bs_alloc * wheads * w_ldx : 0), w_kv(res, usew ? (size_t)bs_alloc * k : 0);
  dev_buf<float> w_thr(res, usew ? (size_t)bs_alloc * wheads : 0), w_c(res, usew ? (size_t)bs_alloc * wheads : 0);
  dev_buf<uint32_t> w_ki(res, usew ? (size_t)bs_alloc * k : 0), w_blk(res, usew ? (size_t)idx.n_lists + 1 : 0), w_bt(res, usew ? (size_t)2 * bs_alloc : 0);
  dev_buf<float4> w_nm(res, usew ? (size_t)bs_alloc * wheads : 0);
  // two-stream schedule (the bench shape and every other search whose head phase is one single-pair item per query and whose
  // tail phase runs pq_filter4_kernel): grouping, work units and B operands on the helper stream, next to the head kernel
  const bool overlap = use3 && use_f4 && head == 1 && !glut && res.tune.pq_overlap != 0;
  dev_buf<work_item> hitems(res, overlap ? (size_t)bs_alloc : 0);
  dev_buf<uint32_t> hpairs(res, overlap ? (size_t)bs_alloc + 1 : 0);  // + the item count
  dev_buf<float4> pair_norms(res, overlap ? (size_t)n_pairs_max : 0);
  dev_buf<uint32_t> group_scratch(res, overlap ? (size_t)n_pairs_max + n_labels + 1 : 0);  // cursors + second buffer of group_pairs
  resources aux = res;
  if (overlap) {
    ensure_aux_stream(res);
    aux.aux_stream = res.aux_stream;
    aux.stream     = res.aux_stream;
  }
  resources& gres = overlap ? aux : res;  // the stream the grouping and the tail phase's preparation are queued on
  // Between fork and join the helper stream's kernels read and write scratch blocks that belong to the handle's stream
  // (sorted_pairs, pair_off, item_off, items, group_scratch, bq3, pair_norms, units3, ...). If anything throws in between
  // (a HIP error, a shard collective's timeout), unwinding would hand those blocks back to the handle's scratch cache while
  // the helper stream may still be using them, and the next call on the handle's stream would re-use them at once. The guard
  // is declared AFTER every such buffer (destroyed first): while armed, its destructor drains the helper stream.
  struct aux_fork_guard {
    hipStream_t s = nullptr;
    bool armed    = false;
    ~aux_fork_guard() { if (armed && s != nullptr) (void)hipStreamSynchronize(s); }
  } fork_guard;
  fork_guard.s = overlap ? aux.stream : nullptr;
  // Partial head (two-stream schedule only: the head items come straight from the probes): the head phase scores the first
  // head_rows rows of a query's nearest list - its k-th best of those bounds the query's final k-th score like the whole list's
  // does, a little less tightly - and the list's remaining rows are screened by the filter with all the other probes.
  uint32_t head_rows = 0u;
  if (overlap) {
    if (res.tune.pq_head_rows >= 0) head_rows = (uint32_t)res.tune.pq_head_rows / 64u * 64u;
    else                            head_rows = 0u;  // (default rule: see DESIGN 3.1f - set by measurement)
    if (head_rows != 0u && head_rows < 4u * (uint32_t)k) head_rows = 0u;  // (a bound needs a few times k rows to mean anything)
  }
  const bool q_is_host = false;  // the C layer guarantees device-accessible queries

  for (int64_t q0 = 0; q0 < n_queries; q0 += max_batch) {
    const int64_t nq      = std::min(max_batch, n_queries - q0);
    const int64_t n_pairs = nq * n_probes;
    load_range_as_float(res, queries, et, q_is_host, idx.dim, q0, nq, qf.data());
    if (idx.metric == M_CosineExpanded) normalize_rows(res, qf.data(), nq, idx.dim);
    // list-sharded index with a communicator: the coarse search is sharded by QUERY - this rank ranks the lists for its
    // slice of the batch and one all-gather of the probe lists (n_probes x 4 B per query) replaces world - 1 replicas of
    // the coarse GEMM + select_k (the same deterministic kernels on the same inputs: identical probes)
    const bool shard_coarse = idx.shard_comm != nullptr && p.coarse_search_dtype == 0 &&  // (one rank: the same calls)
                              !res.tune.shard_coarse_replicated;
    if (shard_coarse) {
      const int64_t slice = (nq + idx.shard_world - 1) / idx.shard_world;
      const int64_t s0    = std::min<int64_t>(nq, (int64_t)idx.shard_rank * slice);
      const int64_t s1    = std::min<int64_t>(nq, s0 + slice);
      if (s1 > s0)
        select_clusters(res, idx, qf.data() + s0 * idx.dim, s1 - s0, n_probes,
                        probes.data() + (size_t)idx.shard_rank * slice * n_probes, 0, qc.data());
      shard_allgather_inplace_u32(res, idx.shard_comm, probes.data(), (size_t)slice * n_probes);
    } else {
      select_clusters(res, idx, qf.data(), nq, n_probes, probes.data(), p.coarse_search_dtype, qc.data());
    }
    rotate_queries(res, idx, qf.data(), qc.data(), nq, p.coarse_search_dtype, rot_q.data());  // ivf_pq_search.cuh:995-1017
    if (p.coarse_search_dtype != 0 && idx.metric == M_CosineExpanded) normalize_rows(res, rot_q.data(), nq, idx.rot_dim);
    // list-major grouping of the (query, probe) pairs
    const uint32_t* labels = probes.data();
    if (overlap) {  // fork: the helper stream starts behind the probes and the rotated queries
      pq3_warm(res, idx, true);  // (derived tables of the index: built here, on the handle's stream, if they are not there yet)
      HIP_TRY(hipEventRecord(res.aux_events[0], res.stream));
      HIP_TRY(hipStreamWaitEvent(gres.stream, res.aux_events[0], 0));
      fork_guard.armed = true;
    }
    if (head > 0 || sharded) {
      hipLaunchKernelGGL(phase_labels_kernel, dim3(nblk(n_pairs, 256)), dim3(256), 0, gres.stream, probes.data(),
                         n_pairs, n_probes, head, idx.n_lists, phase_labels.data(), (uint32_t)idx.shard_world,
                         (uint32_t)idx.shard_rank, n_ranges, idx.list_owner.data(), head_rows != 0u);
      labels = phase_labels.data();
    }
    // the tail phase (warm bounds) of the common configuration runs pq_scan2_kernel on items of 2 * qpb pairs
    // (8 pairs whatever the LUT type: two groups of four with an fp16 LUT, four groups of two with an fp32 LUT)
    bool use2 = head > 0 && bits8 && idx.pq_len == 2 && idx.codebook_kind == 0 && k <= 64 &&  // (k <= 64 excludes the non-fused path)
                ((lut_half && qpb == 4) || (!lut_half && qpb == 2));
    use2 = use2 && res.tune.pq_scan2 != 0 && !use3x;
    // with the matrix-core tail phase the LUT scan only sees the head pairs - nearly always one query per list at the
    // bench shape (10k queries, 16384 lists) - and the pairs of handed-back queries: single-query items and a
    // single-query LUT (a quarter of the LUT build and of the accumulate work of the 4-query interleave)
    const int lut_mode = lut_fp8 ? 2 : (p.lut_dtype != 0 ? 1 : 0);
    // (the wide path: pq_head_kernel only serves the pairs of handed-back queries, when its LUT fits next to its score keys)
    const bool head1 = (use3 && !glut) || (usew && !glut && idx.rot_dim <= 256 /* its residual buffers */ && (size_t)idx.pq_dim * 256 * ((p.lut_dtype == 0 || (lut_fp8 && !acc_half)) ? 4 : 2) <= 96 * 1024);
    build_work_items(gres, labels, n_pairs, n_labels, head1 ? 1 : qpb, sorted_pairs.data(), pair_off.data(), item_off.data(),
                     items.data(), (int)idx.n_lists, use2 ? 8 : qpb, overlap ? group_scratch.data() : nullptr,
                     n_ranges /* the shard's bucket of foreign pairs stays as the scatter left it */, n_probes, nq);
    if (overlap)
      hipLaunchKernelGGL(head_items_kernel, dim3(nblk(nq, 256)), dim3(256), 0, res.stream, probes.data(), nq, n_probes,
                         (uint32_t)idx.shard_world, (uint32_t)idx.shard_rank, idx.list_owner.data(), hitems.data(), hpairs.data(),
                         hpairs.data() + bs_alloc);
    HIP_TRY(hipMemsetAsync(query_kth.data(), 0xff, (size_t)nq * sizeof(uint32_t), res.stream));
    HIP_TRY(hipMemsetAsync(tickets.data(), 0, tickets.bytes(), res.stream));
    // per-pair candidate rows start out invalid: the scan only writes the rows of pairs that found something
    if (use3x) {
      // matrix-core tail phase: only the head segments of a query's row are read before they are written (the pool behind
      // them is filled by count, the rows of handed-back queries are reset by reset_flagged_kernel) - no fill of all
      // n_pairs x k slots (205 MB per batch at the bench shape; on a list shard most of them belong to foreign pairs)
      hipLaunchKernelGGL(init_head_rows_kernel, dim3(nblk(nq * (int64_t)head * k, 256)), dim3(256), 0, res.stream, cand_d.data(),
                         cand_i.data(), nq, (int64_t)n_probes * k, (uint32_t)(head * k));
    } else if (!large_k) {
      HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cand_d.data()), 0x7f7fffff, (size_t)n_pairs * k, res.stream));
      HIP_TRY(hipMemsetAsync(cand_i.data(), 0xff, (size_t)n_pairs * k * sizeof(uint32_t), res.stream));
    } else {
      HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cand_d.data()), 0x7f7fffff, (size_t)nq * scores_ld, res.stream));
      HIP_TRY(hipMemsetAsync(cand_i.data(), 0xff, (size_t)nq * scores_ld * sizeof(uint32_t), res.stream));
      hipLaunchKernelGGL(pair_segments_kernel, dim3(nblk(nq, 256)), dim3(256), 0, res.stream, probes.data(),
                         idx.list_sizes.data(), nq, n_probes, pair_seg.data());
    }
    scan_args a;
    a.query_kth = query_kth.data();
    a.items = items.data(); a.sorted_pairs = sorted_pairs.data(); a.n_lists = idx.n_lists;
    a.rot_queries = rot_q.data(); a.centers_rot = idx.centers_rot.data(); a.pq_centers = idx.pq_centers.data();
    a.per_cluster = idx.codebook_kind == 1;
    a.codes = idx.codes.data(); a.list_offsets = idx.list_offsets.data(); a.list_sizes = idx.list_sizes.data();
    a.out_d = cand_d.data(); a.out_i = cand_i.data();
    a.all_scores = large_k ? cand_d.data() : nullptr; a.all_rows = cand_i.data(); a.pair_seg = pair_seg.data(); a.scores_ld = scores_ld;
    a.n_probes = n_probes; a.rot_dim = idx.rot_dim; a.pq_dim = idx.pq_dim; a.pq_len = idx.pq_len;
    a.pq_bits = idx.pq_bits; a.n_chunks = idx.n_chunks; a.cpc = idx.codes_per_chunk; a.k = (uint32_t)k_scan;
    a.is_ip = idx.metric == M_InnerProduct || idx.metric == M_CosineExpanded;
    a.lut_fp8 = lut_fp8 ? 1 : 0;
    a.dbg   = res.tune.scan_debug;
    a.qcap  = (uint32_t)std::max(0, res.tune.pq_qcap);
    a.filter_bits = filter_bits; a.indices = idx.indices.data();
    const unsigned grid = (unsigned)std::max(8, res.num_cus / 8 * 8);  // persistent: one workgroup per CU
    dev_buf<unsigned long long> stats(res, (a.dbg & (128 | 512)) ? (size_t)ST_COUNT * grid * kScanWaves : 0);
    a.stats = stats.data();
    if (a.dbg & (128 | 512)) HIP_TRY(hipMemsetAsync(stats.data(), 0, stats.bytes(), res.stream));
    dev_buf<char> glut_buf;
    if (glut) {
      const size_t entry = 8;  // 4 x fp16 or 2 x fp32
      a.global_lut_stride = ((size_t)idx.pq_dim * idx.pq_book * entry + 255) & ~size_t(255);
      glut_buf     = dev_buf<char>(res, a.global_lut_stride * grid);
      a.global_lut = glut_buf.data();
    }
    auto launch = [&](const scan_args& sa) {
      if (glut) {
        if (!lut_half)      launch_scan_glut<float, float, 2>(res, sa, smem, grid, big_k);
        else if (!acc_half) launch_scan_glut<__half, float, 4>(res, sa, smem, grid, big_k);
        else                launch_scan_glut<__half, __half, 4>(res, sa, smem, grid, big_k);
        return;
      }
      if (!lut_half) {
        if (qpb == 2) launch_scan_qpb<float, float, 2>(res, sa, smem, grid, bits8, big_k);
        else          launch_scan_qpb<float, float, 1>(res, sa, smem, grid, bits8, big_k);
      } else if (!acc_half) {
        if (qpb == 4)      launch_scan_qpb<__half, float, 4>(res, sa, smem, grid, bits8, big_k);
        else if (qpb == 2) launch_scan_qpb<__half, float, 2>(res, sa, smem, grid, bits8, big_k);
        else               launch_scan_qpb<__half, float, 1>(res, sa, smem, grid, bits8, big_k);
      } else {
        if (qpb == 4)      launch_scan_qpb<__half, __half, 4>(res, sa, smem, grid, bits8, big_k);
        else if (qpb == 2) launch_scan_qpb<__half, __half, 2>(res, sa, smem, grid, bits8, big_k);
        else               launch_scan_qpb<__half, __half, 1>(res, sa, smem, grid, bits8, big_k);
      }
    };
    auto launch1 = [&](const scan_args& sa) {  // single-pair work items: scores of the whole list in LDS, k smallest selected there
      pq3_head h{};
      h.items = sa.items; h.item_begin = sa.item_begin; h.item_end = sa.item_end; h.xcd_ticket = sa.xcd_ticket;
      h.sorted_pairs = sa.sorted_pairs; h.rot_queries = sa.rot_queries; h.cand_d = sa.out_d; h.cand_i = sa.out_i;
      h.query_kth = sa.query_kth; h.n_probes = n_probes; h.k = (uint32_t)k; h.max_list_len = max_list_len; h.is_ip = sa.is_ip;
      h.lut_mode = lut_mode; h.acc_half = acc_half ? 1 : 0; h.filter_bits = filter_bits;
      h.one_shot = sa.one_shot; h.row_limit = sa.row_limit;
      dev_buf<unsigned long long> hst(res, (sa.dbg & 2048) ? 8 : 0);
      if (sa.dbg & 2048) HIP_TRY(hipMemsetAsync(hst.data(), 0, hst.bytes(), res.stream));
      h.stats = hst.data();
      pq3_head_scan(res, idx, h);
      if (sa.dbg & 2048) {
        auto hs = to_host(res, hst.data(), 8);
        const double n = (double)std::max<unsigned long long>(1, hs[5]);
        fprintf(stderr, "[pq_head] items %llu; workgroup cycles per item: header %.0f, LUT %.0f, scores %.0f, select %.0f, output %.0f\n", hs[5],
                hs[0] / n, hs[1] / n, hs[2] / n, hs[3] / n, hs[4] / n);
      }
    };
    if (head > 0) {
      a.xcd_ticket = tickets.data();
      a.item_begin = nullptr;                        a.item_end = item_off.data() + idx.n_lists;
      if (usew) {
        // the wide path's head phase is a bound-only pass through the filter: below, with the tail phase's run description
      } else if (overlap) {  // head phase straight from the probes (one single-pair item per query), no grouping in its way
        scan_args ah = a;
        ah.items = hitems.data(); ah.sorted_pairs = hpairs.data(); ah.item_end = hpairs.data() + bs_alloc;
        ah.one_shot = (uint32_t)nq;  // one workgroup per item: slots free up item by item, the helper stream's kernels fit in between
        ah.row_limit = head_rows;
        launch1(ah);
      } else if (head1) launch1(a); else launch(a);  // head phase: the nearest probes, cold bounds
      // list-sharded index with a communicator: every rank continues with the bound of the query's globally nearest
      // probe (one all-reduce of nq keys), not only the rank that owns that probe
      if (idx.shard_comm != nullptr) shard_allreduce_min_u32(res, idx.shard_comm, query_kth.data(), (size_t)nq);
      a.xcd_ticket = tickets.data() + 8 * 32;
      a.item_begin = item_off.data() + idx.n_lists;  a.item_end = item_off.data() + 2 * idx.n_lists;
      if (use3x) {
        HIP_TRY(hipMemsetAsync(qstate.data(), 0, qstate.bytes(), res.stream));
        pq3_run r{};
        r.pair_norms = pair_norms.data(); r.head_rows = head_rows;
        r.nq = nq; r.n_probes = n_probes; r.k = (uint32_t)k; r.head = head; r.is_ip = a.is_ip;
        r.lut_mode = lut_fp8 ? 2 : (p.lut_dtype != 0 ? 1 : 0); r.acc_half = acc_half ? 1 : 0;
        r.sorted_pairs = sorted_pairs.data(); r.pair_off = pair_off.data(); r.probes = probes.data();
        r.rot_queries = rot_q.data(); r.query_kth = query_kth.data();
        r.cand_d = cand_d.data(); r.cand_i = cand_i.data(); r.cand_r = cand_r.data();
        r.qflag = qstate.data(); r.qcnt = qstate.data() + bs_alloc; r.counters = qstate.data() + 2 * bs_alloc;
        r.surv_cnt = qstate.data() + 2 * bs_alloc + 2;
        r.ov_cnt = qstate.data() + 2 * bs_alloc + 4 + pq3_regions(res); r.ov_off = r.ov_cnt + bs_alloc;
        r.surv = surv.data(); r.surv_cap = surv_cap; r.units = units3.data(); r.unit_off = unit_off.data();
        r.unit_rows = unit_rows; r.xcd_ticket = tickets.data() + 2 * 8 * 32; r.fb_items = fb_items.data();
        r.filter_bits = filter_bits; r.overflow = overflow3.data(); r.overflow_cap = overflow_cap;
        r.bq = (use_f4 || usew) ? bq3.data() : nullptr; r.thr = thr3.data();
        dev_buf<unsigned long long> st3(res, (a.dbg & 1024) ? 8 : 0);
        if (a.dbg & 1024) HIP_TRY(hipMemsetAsync(st3.data(), 0, st3.bytes(), res.stream));
        r.stats = st3.data(); r.filter_dbg = (a.dbg >> 16) & 255;  // CUVS_AMD_SCAN_DEBUG bits 16..23
        if (overlap) {
          // the helper stream: work units, B operands and norms of the tail pairs (nothing here reads the head phase's bounds);
          // join; then thresholds, filter and re-score behind the head kernel (and the bound all-reduce) on the handle's stream
          r.stage = 1;
          pq3_tail(gres, idx, r);
          HIP_TRY(hipEventRecord(res.aux_events[1], gres.stream));
          HIP_TRY(hipStreamWaitEvent(res.stream, res.aux_events[1], 0));
          fork_guard.armed = false;  // joined: everything the helper stream was given is ordered before the handle's stream again
          r.stage = 2;
        }
        if (usew) {
          const pqw_bufs hb{w_x.data(), w_ldx, w_kv.data(), w_ki.data(), w_thr.data(), w_c.data(), w_nm.data(), tickets.data(), w_blk.data(), w_bt.data()};
          const bool ok = pqw_head_bounds(res, idx, r, hb);
          CUVS_EXPECTS(ok, "ivf_pq: the wide path's decoded rows are gone");
          pqw_tail(res, idx, r, hb);
        } else {
          pq3_tail(res, idx, r);
        }
        // queries the filter could not serve (no finite bound, operands beyond fp16, full pool): LUT scan of their pairs
        a.items = fb_items.data(); a.item_begin = nullptr; a.item_end = r.counters;
        a.xcd_ticket = tickets.data() + 3 * 8 * 32;
        if (head1) launch1(a); else launch(a);
        pq3_merge(res, r, top_d.data(), top_i.data());
        if (a.dbg & 1024) {
          auto hs = to_host(res, st3.data(), 8);
          g_pq3_last_stats[0] = hs[0]; g_pq3_last_stats[1] = hs[1]; g_pq3_last_stats[2] = hs[2]; g_pq3_last_stats[3] = hs[7];
          fprintf(stderr, "[pq_scan3] units %llu; wave cycles per unit: prologue %.0f, loop %.0f (slow path %.0f); per subtile %.0f\n", hs[7],
                  (double)hs[4] / std::max<unsigned long long>(1, hs[7]), (double)hs[5] / std::max<unsigned long long>(1, hs[7]),
                  (double)hs[6] / std::max<unsigned long long>(1, hs[7]), (double)hs[5] / std::max<unsigned long long>(1, hs[2]));
          auto hc = to_host(res, r.counters, 2);
          g_pq3_last_stats[4] = hc[0]; g_pq3_last_stats[5] = hc[1];
          fprintf(stderr, "[pq_scan3] overflow entries %u\n", hc[1]);
          hc[1] = hc[0];
          fprintf(stderr, "[pq_scan3] pairs screened %llu, survivors %llu (%.4f%%), subtiles %llu (slow path %llu), fallback pairs %u\n",
                  hs[0], hs[1], 100.0 * hs[1] / (double)std::max<unsigned long long>(1, hs[0]), hs[2], hs[3], hc[1]);
        }
      }
      else if (!use2)     launch(a);  // tail phase: warm bounds
      else if (!lut_half) launch_scan2<float, float, 2, 4>(res, a, grid);
      else if (!acc_half) launch_scan2<__half, float, 4, 2>(res, a, grid);
      else                launch_scan2<__half, __half, 4, 2>(res, a, grid);
    } else {
      a.xcd_ticket = tickets.data();
      a.item_begin = nullptr; a.item_end = item_off.data() + idx.n_lists;
      launch(a);
    }
    if (a.dbg & (128 | 512)) {
      std::vector<unsigned long long> hw(stats.n);
      HIP_TRY(hipMemcpyAsync(hw.data(), stats.data(), stats.bytes(), hipMemcpyDeviceToHost, res.stream));
      HIP_TRY(hipStreamSynchronize(res.stream));
      unsigned long long h[ST_COUNT] = {};
      for (size_t i = 0; i < hw.size(); ++i) h[i % ST_COUNT] += hw[i];
      const double w = 1.0 / (16.0 * grid);  // wave cycles -> average cycles per wave
      if (a.dbg & 4096) {  // per-wave view of the filter pass (which wave of a workgroup runs late?)
        for (int which : {(int)ST_SCAN, (int)ST_ALIVE2, (int)ST_F_GATHER, (int)ST_F_FLUSH, (int)ST_STAGE2, (int)ST_CAND, (int)ST_S2_CALLS}) {
          fprintf(stderr, "[pq_scan per-wave stat %d, Mcycles]", which);
          for (int wv = 0; wv < kScanWaves; ++wv) {
            unsigned long long t = 0;
            for (unsigned b = 0; b < grid; ++b) t += hw[((size_t)b * kScanWaves + wv) * ST_COUNT + which];
            fprintf(stderr, " %.2f", (double)t / grid * 1e-6);
          }
          fprintf(stderr, "\n");
        }
      }
      fprintf(stderr,
              "[pq_scan stats] items %llu rows %llu queued %llu (%.2f%%) stage2 calls %llu alive after chunk1/2/3 %llu/%llu/%llu"
              " | cycles per wave: header %.3g lut %.3g scan %.3g (stage2 %.3g) merge+sync %.3g\n",
              h[ST_ITEMS], h[ST_ROWS], h[ST_QUEUED], 100.0 * h[ST_QUEUED] / (double)std::max<unsigned long long>(1, h[ST_ROWS]),
              h[ST_S2_CALLS], h[ST_ALIVE1], h[ST_ALIVE2], h[ST_ALIVE3], h[ST_HEADER] * w, h[ST_LUT] * w, h[ST_SCAN] * w,
              h[ST_STAGE2] * w, (double)(h[ST_MERGE] - h[ST_HEADER] - h[ST_LUT] - h[ST_SCAN]) * w);
      fprintf(stderr, "[pq_scan2 waits, cycles per wave] after filter LUT %.3g, after filter pass %.3g, before LUT B %.3g, after exact LUT "
              "%.3g, after exact pass %.3g, merge %.3g | filter pass: code-load wait %.3g, gather block %.3g, flush %.3g\n",
              h[ST_ALIVE1] * w, h[ST_ALIVE2] * w, h[ST_S2_CALLS] * w, h[ST_ALIVE3] * w, h[ST_CAND] * w, h[ST_MERGE] * w,
              h[ST_F_LOAD] * w, h[ST_F_GATHER] * w, h[ST_F_FLUSH] * w);
    }
    // per-query merge of n_probes * k candidates (ivf_pq_search.cuh:646-655)
    if (use3x) {
      // merged already (pq3_merge: head lists + pool)
    } else if (!large_k) {
      select_k<uint32_t, uint32_t>(res, cand_d.data(), cand_i.data(), nq, (int64_t)n_probes * k, (int64_t)n_probes * k,
                                   k, top_d.data(), top_i.data(), true);
    } else {
      select_k<uint32_t, uint32_t>(res, cand_d.data(), cand_i.data(), nq, (int64_t)scores_ld, (int64_t)scores_ld, k,
                                   top_d.data(), top_i.data(), true);
    }
    const float sc = ivf_pq_index::scale(et);
    hipLaunchKernelGGL(postprocess_kernel, dim3(nblk(nq * k, 256)), dim3(256), 0, res.stream, top_i.data(),
                       top_d.data(), nq * k, idx.indices.data(), idx.metric, sc * sc, neighbors + q0 * k,
                       distances + q0 * k);
  }
  HIP_TRY(hipGetLastError());
}

}  // namespace cuvs_amd
