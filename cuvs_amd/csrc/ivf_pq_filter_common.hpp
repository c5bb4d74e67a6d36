// Shared by the matrix-core filter kernels of the IVF-PQ / IVF-Flat tail phase (ivf_pq_scan3.hip, ivf_pq_filter4.hip).
#pragma once
#include "ivf_pq.hpp"
#include "device_utils.hpp"

#include <cmath>

namespace cuvs_amd {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// work unit of the filter: a row chunk of a list x a block of the pairs probing it
struct filter_unit {
  uint32_t list, first, count, row0;  // list, first pair (position in sorted_pairs) and number of pairs, first row of the chunk
  uint32_t base_row, r_end;           // flat row of the list's first row; end of the chunk (row index inside the list)
  uint32_t pad0, pad1;
};

// Largest value B with: exact score > bound  whenever  (row term - 2 dot16 / sc^2) > B   (L2; see the file header).
// With T the real-valued score, S the score in the reference's arithmetic and A the filter's value
//   S >= T (1 - eps) - alpha                          entry roundings of the LUT type, summation in the score type
//   T >= A - 2^-9 (rn + dn) - mabs                    fp16 rounding of both GEMM operands: |x^ y^ - x y| <= 2^-10 |x y| +
//                                                     2^-25 (|x| + |y|) per element (normal + subnormal range), fp32
//                                                     accumulation; 2 |r.d| <= rn + dn
// so S > bound is implied by  dn (1 - 2^-9) - 2 dot > (bound + alpha) / (1 - eps) + mabs - rn (1 - 2^-9). The right side
// is evaluated in double and rounded up.
template <typename P>
__device__ inline float filter_threshold(const float bound, const float rn, const P& a)
{
  const double mabs = 1.1920929e-07 /* 2^-23 */ / (double)a.sc *
                      (sqrt((double)a.rot_dim * (double)rn) + (double)a.rot_dim * (double)a.cbmax);
  const double b = ((double)bound + (double)a.alpha) * (1.0 + 2.0 * (double)a.eps) + mabs - (double)rn * (1.0 - 1.0 / 512.0);
  float f = (float)b;
  f += fabsf(f) * 2.4e-7f + 1e-37f;
  return f;
}

// Inner product / cosine: T = -(q.c + q.d) (q the rotated query, c the list centre, d the row's decoded residual); the LUT
// entries have both signs, so the roundings of S scale with sum |entry| <= |q| (|c| + |d|) instead of with T:
//   S >= T - eps |q| (|c| + |d|) - alpha,   T >= A - 2^-17 |q| |c| - 2^-9.9 |q| |d| - mabs,   A = -(qc + dot16 / sc^2)
// with qc = q.c in fp32 and |d| <= dmax (the largest decoded norm of the index). S > bound is implied by
//   -dot16 / sc^2 > bound + qc + |q| ((eps + 2^-17) |c| + (eps + 2^-9) dmax) + mabs + alpha.
template <typename P>
__device__ inline float filter_threshold_ip(const float bound, const float qn, const float cn, const float qc, const P& a)
{
  const double nq = sqrt((double)qn), nc = sqrt((double)cn);
  const double mabs = 1.1920929e-07 /* 2^-23 */ / (double)a.sc * (sqrt((double)a.rot_dim) * nq + (double)a.rot_dim * (double)a.cbmax);
  const double m = nq * (((double)a.eps + 7.63e-6) * nc + ((double)a.eps + 1.0 / 512.0) * (double)a.dmax);
  const double b = (double)bound + (double)qc + (fabs((double)qc) + fabs((double)bound)) * 1e-6 + m + mabs + (double)a.alpha;
  float f = (float)b;
  f += fabsf(f) * 2.4e-7f + 1e-37f;
  return f;
}


// ---- pq_filter4_kernel (ivf_pq_filter4.hip): pre-pass + filter, one wave per SIMD
struct filter4_launch {
  const filter_unit* units;
  const uint32_t* n_units;
  uint32_t* xcd_ticket;
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;
  uint32_t n_lists;
  const uint32_t* probes;
  const float* rot_queries;
  const float* centers_rot;
  const uint32_t* query_kth;
  uint32_t* qflag;
  void* bq;
  float* thr;
  const uint32_t* cb16;
  const uint8_t* codes;
  const uint32_t* list_offsets;
  const uint32_t* list_sizes;
  const float* row_term;
  void* surv;
  uint32_t* surv_cnt;
  uint32_t surv_entries, n_probes, rot_dim, unit_rows;  // surv_entries: size of the survivor buffer (handed out in chunks of 256)
  float sc, c1, eps, alpha, cbmax, dmax, bound_max;
  int is_ip, dbg, nch, pl, per_cluster;  // nch: 16-byte code chunks per row (pq_dim / 16); pl: pq_len
  int64_t n_pairs;
  unsigned long long* stats;
  unsigned grid;
  int stage = 0;              // pq3_run::stage: 1 = pre-pass without thresholds only, 2 = thresholds + filter, 0 = both in one go
  void* pair_norms = nullptr; // [tail pairs] float4 between stage 1 and stage 2
  int flat = 0;               // IVF-Flat's pairs (an unserved query survives everything instead of being handed back)
  int bprep_only = 0;         // the pre-pass alone (B operands + thresholds): IVF-Flat's flat_filter2_kernel follows it
  int head_labels = 0;        // the pre-pass over the HEAD pairs (labels [0, n_lists)) instead of the tail pairs: IVF-Flat's bound-only head phase
};
void pq4_filter(resources& res, const filter4_launch& l);

// ---- pqw_filter_kernel (ivf_pq_wide.hip): the filter over the index's DECODED rows (fp16, A-operand layout), for rot_dim beyond
// pq_filter4_kernel's decode table and for bounds from several head lists
bool pqw_shape(uint32_t rot_dim);  // rot_dim the kernel is built for (256, 384, 512, 768)
uint32_t pqw_group();              // queries per work unit
// rows16 [padded_rows / 32][rot_dim / 16][64 lanes] x 16 bytes from the one-byte-per-code copy and the decode table (cb16_kernel)
void pqw_decode(resources& res, const uint8_t* codes8, uint32_t n_chunks, const uint32_t* cb16, uint32_t pq_len, int64_t padded_rows,
                uint32_t rot_dim, void* rows16);
struct wide_prep {  // pre-pass: fp16 B operands of the head (labels [0, n_lists)) or tail pairs, + norms (head) / thresholds (tail)
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;
  uint32_t n_lists;
  const uint32_t* probes;
  const float* rot_queries;
  const float* centers_rot;
  const uint32_t* query_kth;
  uint32_t* qflag;
  void* bq;      // [(n_pairs / 32 + n_lists) blocks of 32 pairs][rot_dim / 16][64 lanes] x 16 bytes
  uint32_t* blk_off;  // out: [n_lists + 1] first block of every list
  float* thr;    // tail: thresholds in accumulator units; head: the pairs' constants -|r|^2 sc^2 / 2 (values of different lists become comparable)
  void* norms;   // head: [query * heads + probe rank] x 16 bytes
  uint32_t n_probes, rot_dim, heads;
  float sc, c1, eps, alpha, cbmax, dmax, bound_max;
  int head;
  int is_ip;        // inner product / cosine: the operand is the query (not the residual), thresholds by filter_threshold_ip
  int flat = 0;     // IVF-Flat's pairs: a query the filter cannot serve survives every test instead of being handed back
  int64_t n_pairs;  // upper bound of the pairs served (grid size)
};
void pqw_bprep(resources& res, const wide_prep& l);
struct wide_filter {
  const filter_unit* units;
  const uint32_t* n_units;
  uint32_t* xcd_ticket;
  const uint32_t* sorted_pairs;
  const uint32_t* pair_off;
  uint32_t n_lists;
  const void* bq;
  const uint32_t* blk_off;
  const float* thr;
  const void* rows16;
  const float* row_term;  // [padded_rows] fp32 -|d|^2 (1 - 2^-9) sc^2 / 2: the accumulators' initial values (nullptr: inner product)
  const float* zeros;     // 32 zero floats
  const uint32_t* filter_bits;  // emit: the pre-filter's bitset over source ids (nullptr: none) and the rows' source ids
  const int64_t* indices;
  uint32_t* qflag;
  void* surv;             // one region of surv_cap entries per workgroup + a shared spill region of spill_cap entries
  uint32_t* surv_cnt;     // [grid + 1]
  uint32_t surv_cap, spill_cap, n_probes, rot_dim;
  float* xbuf;            // emit: values of (head pair, row) at xbuf[(query * heads + probe rank) * ldx + row of the list]
  uint32_t ldx, heads;
  int emit;
  unsigned grid;
  unsigned long long* stats;
  uint32_t* fail = nullptr;            // IVF-Flat: device word raised when the survivor buffer is full (nullptr: the query is flagged)
  const char* profile_name = nullptr;  // (default "pq_filter_kernel")
};
void pqw_filter(resources& res, const wide_filter& l);

}  // namespace cuvs_amd
