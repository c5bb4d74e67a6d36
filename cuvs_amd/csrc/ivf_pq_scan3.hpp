// Interface of the warm-bounds IVF-PQ phase on the matrix cores (ivf_pq_scan3.hip), driven by ivf_pq_search.hip.
#pragma once
#include "ivf_pq.hpp"

namespace cuvs_amd {

struct pq3_tables {
  const uint32_t* cb16;   // [pq_dim][256] codebook entries as scaled fp16 pairs (decode table of the filter)
  const uint32_t* row_term;  // [padded_rows] fp16 (hi, lo) of -|decoded residual|^2 (1 - 2^-9) sc^2 / 2: an extra K element of the GEMM
  float sc, cbmax, dmax;
};

// one batch of the tail phase; all pointers device memory owned by the caller (ivf_pq_search)
struct pq3_run {
  int64_t nq;
  uint32_t n_probes, k, head;
  int is_ip, lut_mode /* 0 fp32, 1 fp16, 2 fp8 */, acc_half;
  const uint32_t* sorted_pairs;  // pair ids grouped by phase label
  const uint32_t* pair_off;      // [2 n_lists + 1] label offsets into sorted_pairs (head labels, then tail labels)
  const uint32_t* probes;        // [n_pairs] list of every pair
  const float* rot_queries;
  const float* rescore_queries = nullptr;  // IVF-Flat cosine: the raw queries of the exact chain (rot_queries: unit length)
  int cosine = 0;
  const uint32_t* query_kth;     // bound keys after the head phase
  float* cand_d;                 // [nq, n_probes * k]: head segments first, the rest is the query's pool
  uint32_t* cand_i;
  uint32_t* cand_r;              // probe rank of the pool entries
  uint32_t* qflag;               // [nq] zeroed: queries handed back to the LUT scan
  uint32_t* qcnt;                // [nq] zeroed: pool fill
  uint32_t* counters;            // [2] zeroed: fallback work items, overflow entries
  void* overflow;                // [2 * overflow_cap] x 16 bytes: candidates of queries whose pool ran over, then the same
  uint32_t overflow_cap;         //   binned by query
  uint32_t* ov_cnt;              // [nq] zeroed
  uint32_t* ov_off;              // [nq + 1]
  uint32_t* surv_cnt;            // [pq3_regions() + 1] zeroed: fill of every survivor region, of the spill region
  void* surv;                    // [surv_cap] (pair, flat row), cut into one region per workgroup of the filter
  uint32_t surv_cap;
  void* units;                   // work units of the filter, pq3_max_units() entries
  uint32_t* unit_off;            // [n_lists + 1]
  uint32_t unit_rows;
  uint32_t* xcd_ticket;          // 8 x 32 zeroed words
  void* fb_items;                // work_item[n tail pairs]: single-pair items of the flagged queries
  const uint32_t* filter_bits;
  uint32_t* fail;                // IVF-Flat: device word raised when a buffer ran over (nullptr for IVF-PQ)
  void* bq;                      // pq_filter4_kernel: [tail pairs] x rot_dim fp16 B operands (nullptr: pq_filter_kernel, two waves per SIMD)
  float* thr;                    //   and [tail pairs] thresholds, both written by the pre-pass
  int filter_dbg;                // ablation bits of the filter kernel (timing only)
  unsigned long long* stats;     // optional device [8]
  // Two-stream schedule of a batch (ivf_pq_search.hip, round 5): everything of the tail phase that does not depend on the head
  // phase's bounds - work units, B operands - runs on a helper stream WHILE the head kernel runs.
  //   stage 0: the whole tail phase on one stream (every shape outside the two-stream schedule)
  //   stage 1: work units + pre-pass without thresholds (B operands, and the norms the thresholds need -> pair_norms)
  //   stage 2: thresholds from pair_norms and the head bounds, filter, re-score, fallback work items
  int stage = 0;
  void* pair_norms = nullptr;
  // Partial head (round 5): the head phase scored only the first head_rows rows of every query's nearest list; that pair is
  // ALSO a tail pair of its list (all its rows are screened against the bound), and the re-score drops what the head phase
  // has already scored - survivors of a head pair (probe rank < head) below head_rows. 0: the head phase scored whole lists.
  // IVF-Flat's bound-only head phase (flat3_head_bounds): values of every (head pair, row) and the head pairs' thresholds; the tail
  // phase turns them into the head pairs' survivors. nullptr: the head phase scored its lists exactly (ivf_flat_scan_kernel)
  const float* hb_xbuf = nullptr;
  uint32_t hb_ldx = 0;
  const float* hb_thr = nullptr;
  uint32_t head_rows = 0;    // [tail pairs] x 16 bytes (|r|^2, |c|^2, q.c, largest scaled operand), written by stage 1
};

// single-query list scan with the k smallest selected in LDS (head phase, pairs of handed-back queries)
struct pq3_head {
  const void* items;             // work_item[]: single-pair items
  const uint32_t* item_begin;    // device scalars (nullptr: from 0)
  const uint32_t* item_end;
  uint32_t* xcd_ticket;          // 8 x 32 zeroed words
  const uint32_t* sorted_pairs;
  const float* rot_queries;
  float* cand_d;
  uint32_t* cand_i;
  uint32_t* query_kth;
  uint32_t n_probes, k, max_list_len;
  int is_ip, lut_mode, acc_half;
  const uint32_t* filter_bits;
  unsigned long long* stats;     // optional device [8] (CUVS_AMD_SCAN_DEBUG=2048)
  uint32_t one_shot = 0;         // > 0: the number of items, one workgroup each (no tickets): the two-stream schedule's head launch
  uint32_t row_limit = 0;        // > 0: score only the first row_limit rows of a list (partial head: pq3_run::head_rows)
};
void pq3_head_scan(resources& res, const ivf_pq_index& idx, const pq3_head& h);

// ---- IVF-Flat (fp32 or fp16 rows, L2) through the same filter: a derived fp16 copy of the rows' residuals laid out as MFMA operands
struct flat3_cache {
  dev_buf<uint4> rows16;       // [padded_rows / 32][dim / 16][64 lanes] x 16 bytes
  dev_buf<uint32_t> row_term;  // [padded_rows] K-extension halves of -|x - c|^2 (1 - 2^-9) sc^2 / 2
  dev_buf<float> row_term32;   // the same terms in fp32 and 32 zero floats (the wide filter: 256 / 384 / 512 / 768 dimensions; rows16 is then
  dev_buf<float> zeros;        //   in the natural K order)
  float sc = 1.f, maxres = 0.f, maxnorm = 0.f;  // scaling, largest |component| and largest norm of a residual
  const void* data_ptr = nullptr;
  int64_t rows = -1, size = -1;
  const void* failed_ptr = nullptr;  // index state for which the device had no room for the copy
  int64_t failed_rows = -1, failed_size = -1;
};
struct flat3_view {  // the IVF-Flat index as ivf_flat.hip holds it
  const uint8_t* data;
  const float* centers;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  const int64_t* indices;
  uint32_t n_lists, dim, n_chunks;
  int64_t padded_rows, size;
  uint32_t max_list_len;
  int elem;  // row type: 0 fp32, 1 fp16, 2 int8, 3 uint8
  bool unit_rows = false;  // cosine: the fp16 copy holds x / |x| - c (the centres are means of unit-length rows)
};
bool flat3_supported(uint32_t dim, int k);
bool flat3_wide(uint32_t dim);  // the dimensions served by the wide filter (ivf_pq_wide.hip): the caller provides r.bq / r.thr
// filter + re-score of the tail phase (units from r.pair_off); r.rot_queries = the fp32 queries [nq, dim]; *r.fail is
// raised when a buffer ran over: the caller then re-runs the tail phase on the scan kernel. Returns false (nothing
// launched) when the device has no room for the fp16 copy.
bool flat3_tail(resources& res, const flat3_view& v, flat3_cache& cache, const pq3_run& r);
// The head phase as a BOUND-ONLY pass through the fp16 copy (L2, up to 128 dimensions): the head pairs (labels [0, n_lists) of
// r.pair_off) grouped by list go through flat_filter2_kernel in its emit form, the k-th largest value of every pair + the error terms
// of the screen give r.query_kth (an upper bound of the query's k-th best exact score), hb.thr_head the pairs' thresholds. Buffers are
// the caller's. Returns false (nothing launched) when the fp16 copy cannot be made.
struct flat3_head_bufs {
  float* xbuf; uint32_t ldx;          // [nq, ldx] values, ldx >= the longest list rounded up to 64
  float* kth_val; uint32_t* kth_idx;  // [nq, k]
  float* thr_head;                    // [nq]
  void* norms;                        // [nq] x 16 bytes
  uint32_t* tickets;                  // 8 x 32 zeroed words
};
bool flat3_head_bounds(resources& res, const flat3_view& v, flat3_cache& cache, const pq3_run& r, const flat3_head_bufs& hb);

unsigned pq3_grid(const resources& res);     // workgroups of the filter
unsigned pq3_regions(const resources& res);  // survivor regions of pq_filter_kernel (pq_filter4_kernel hands out chunks instead)
bool pq3_supported(const ivf_pq_index& idx, int k);
bool pq3_bound_useful(const ivf_pq_index& idx, int k);  // k is a small enough fraction of a list for the head bound to prune
size_t pq3_max_units(const ivf_pq_index& idx, int64_t n_pairs, uint32_t* unit_rows, bool filter4);
// the index's derived tables of this path (decode table, row terms, one-byte-per-code copy), built on first use and when the
// lists change - on the stream of `res`: the two-stream schedule calls this BEFORE it forks, so that no helper-stream kernel
// ever runs next to the build
void pq3_warm(resources& res, const ivf_pq_index& idx, bool filter4);
// filter + re-score + fallback work items of the flagged queries (the caller launches the LUT scan on them)
void pq3_tail(resources& res, const ivf_pq_index& idx, const pq3_run& r);
void pq3_merge(resources& res, const pq3_run& r, float* top_d, uint32_t* top_i);

// ---- the wide path (ivf_pq_wide.hip): rot_dim 256 .. 768 in steps the kernel is built for, any pq_len, PER_SUBSPACE, every metric, pre-filters; the head
// phase is a BOUND-ONLY pass over the `heads` nearest lists of every query (their union bounds the k-th score: k may be a large
// fraction of ONE list), through the filter in its emit form; both phases read the index's decoded fp16 rows (scan3_cache::rows16w)
bool pqw_supported(const ivf_pq_index& idx, int k);
// head lists per query so that k is at most ~2.5 % of their rows; 0: more than half of
// the probes would be head lists - the LUT scan stays
uint32_t pqw_heads(const ivf_pq_index& idx, int k, uint32_t n_probes);
// the decoded copy is there (made now, on the stream of `res`, if the device has the room)
bool pqw_ready(resources& res, const ivf_pq_index& idx);
struct pqw_bufs {
  float* xbuf; uint32_t ldx;          // [nq * heads, ldx] values of (head pair, row), ldx >= the longest list rounded up to 64
  float* kth_val; uint32_t* kth_idx;  // [nq, k]
  float* thr_head;                    // [nq * heads] thresholds of the head pairs, in the units of xbuf
  float* head_c;                      // [nq * heads] the head pairs' constants (by pair position)
  void* norms;                        // [nq * heads] x 16 bytes
  uint32_t* tickets;                  // 8 x 32 zeroed words
  uint32_t* blk_off;                  // [n_lists + 1] scratch: first operand block of every list (head pairs, then tail pairs)
  uint32_t* bound_tmp;                // [2 nq] scratch of the bound kernels (largest score key, short-list flag)
};
// r.head = heads. Returns false (nothing launched) when the decoded copy cannot be made: the caller runs the exact head phase and
// the LUT scan. Leaves r.query_kth (bounds) and hb.thr_head; queries without a bound (fewer than k rows in their head lists) flagged.
bool pqw_head_bounds(resources& res, const ivf_pq_index& idx, const pq3_run& r, const pqw_bufs& hb);
// filter + head-pair survivors + re-score + fallback work items of the flagged queries (ALL their pairs: the head phase left no
// candidates); the caller launches the LUT scan on those and pq3_merge
void pqw_tail(resources& res, const ivf_pq_index& idx, const pq3_run& r, const pqw_bufs& hb);

}  // namespace cuvs_amd
