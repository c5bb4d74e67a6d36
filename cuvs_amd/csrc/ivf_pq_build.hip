// IVF-PQ build / extend / code packing for the MI355X layout (ivf_pq.hpp).
// Restates cpp/src/neighbors/ivf_pq/ivf_pq_build.cuh: build (:1231-1389), set_centers (:262-300),
// train_per_subset (:328-407), extend (:981-1211), make_rotation_matrix (ivf_pq_build_common.cu:235-261),
// encode_vectors (ivf_pq_process_and_fill_codes.cuh:65-112), bit packing (ivf_pq_codepacking.cuh:22-137).
#include "ivf_pq.hpp"
#include "ops.hpp"
#include "device_utils.hpp"

#include <algorithm>
#include <cmath>
#include <random>

namespace cuvs_amd {

namespace {

inline unsigned nblk(int64_t n, int per) { return grid_blocks(n, per); }

// ------------------------------------------------------------------ loading rows as float
template <typename T>
__global__ void convert_kernel(const T* __restrict__ in, int64_t n, float mult, float* __restrict__ out)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = to_float(in[i]) * mult;
}
template <typename T>
__global__ void gather_convert_kernel(const T* __restrict__ in, int64_t dim, const uint32_t* __restrict__ ids,
                                      int64_t cnt, float mult, float* __restrict__ out)
{
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= cnt) return;
  const T* src = in + (int64_t)ids[row] * dim;
  for (int64_t d = threadIdx.x & 63; d < dim; d += 64) out[row * dim + d] = to_float(src[d]) * mult;
}

float mapping_mult(elem_t et) { return et == elem_t::u8 ? 1.0f / 256.0f : (et == elem_t::i8 ? 1.0f / 128.0f : 1.0f); }

template <typename T>
void convert_typed(resources& res, const void* d_in, int64_t n, float mult, float* out)
{
  hipLaunchKernelGGL((convert_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, res.stream,
                     static_cast<const T*>(d_in), n, mult, out);
}
void convert_any(resources& res, const void* d_in, elem_t et, int64_t n, float* out)
{
  float mult = mapping_mult(et);
  switch (et) {
    case elem_t::f32: convert_typed<float>(res, d_in, n, mult, out); break;
    case elem_t::f16: convert_typed<__half>(res, d_in, n, mult, out); break;
    case elem_t::i8: convert_typed<int8_t>(res, d_in, n, mult, out); break;
    case elem_t::u8: convert_typed<uint8_t>(res, d_in, n, mult, out); break;
  }
}
template <typename T>
void gather_typed(resources& res, const void* d_in, int64_t dim, const uint32_t* ids, int64_t cnt, float mult,
                  float* out)
{
  hipLaunchKernelGGL((gather_convert_kernel<T>), dim3(nblk(cnt, 4)), dim3(256), 0, res.stream,
                     static_cast<const T*>(d_in), dim, ids, cnt, mult, out);
}

}  // namespace

// rows [r0, r0 + cnt) -> float [cnt, dim] on the device
void load_range_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim, int64_t r0,
                         int64_t cnt, float* out)
{
  const size_t esz  = elem_size(et);
  const char* src   = static_cast<const char*>(data) + (size_t)r0 * dim * esz;
  if (!is_host) {
    convert_any(res, src, et, cnt * dim, out);
  } else {
    dev_buf<char> stage(res, (size_t)cnt * dim * esz);
    copy_async(res, stage.data(), src, stage.bytes());
    convert_any(res, stage.data(), et, cnt * dim, out);
    sync(res);  // the host buffer may be pageable: finish before the caller moves on
  }
}

// rows ids[0..cnt) (device ids) -> float [cnt, dim] on the device
void load_gather_as_float(resources& res, const void* data, elem_t et, bool is_host, int64_t dim,
                          const uint32_t* d_ids, int64_t cnt, float* out)
{
  float mult = mapping_mult(et);
  if (!is_host) {
    switch (et) {
      case elem_t::f32: gather_typed<float>(res, data, dim, d_ids, cnt, mult, out); break;
      case elem_t::f16: gather_typed<__half>(res, data, dim, d_ids, cnt, mult, out); break;
      case elem_t::i8: gather_typed<int8_t>(res, data, dim, d_ids, cnt, mult, out); break;
      case elem_t::u8: gather_typed<uint8_t>(res, data, dim, d_ids, cnt, mult, out); break;
    }
  } else {
    const size_t esz = elem_size(et);
    std::vector<uint32_t> ids = to_host(res, d_ids, cnt);
    std::vector<char> host((size_t)cnt * dim * esz);
    const char* src = static_cast<const char*>(data);
    for (int64_t i = 0; i < cnt; ++i)
      memcpy(host.data() + (size_t)i * dim * esz, src + (size_t)ids[i] * dim * esz, dim * esz);
    dev_buf<char> stage(res, host.size());
    copy_async(res, stage.data(), host.data(), host.size());
    convert_any(res, stage.data(), et, cnt * dim, out);
    sync(res);
  }
}

namespace {

__global__ void strided_ids_kernel(uint32_t* ids, int64_t n, int64_t stride)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = (uint32_t)(i * stride);
}

// centers [n_lists, dim] -> padded [n_lists, dim_ext] with |c|^2 in column dim (ivf_pq_build.cuh:262-284)
__global__ void pad_centers_kernel(const float* __restrict__ c, const float* __restrict__ norms, int n_lists,
                                   int dim, int dim_ext, float* __restrict__ out)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_lists * dim_ext) return;
  int r = (int)(i / dim_ext), col = (int)(i % dim_ext);
  out[i] = col < dim ? c[(int64_t)r * dim + col] : (col == dim ? norms[r] : 0.f);
}

// resid[i, :] -= centers_rot[labels[i], :]
__global__ void subtract_center_kernel(float* __restrict__ resid, int64_t n, int rot_dim,
                                       const uint32_t* __restrict__ labels, const float* __restrict__ centers_rot)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * rot_dim) return;
  int64_t r = i / rot_dim;
  int d     = (int)(i % rot_dim);
  resid[i] -= centers_rot[(int64_t)labels[r] * rot_dim + d];
}

// centers_tmp [pq_dim][book][pq_len] -> pq_centers [pq_dim][pq_len][book] (ivf_pq_build.cuh:303-325)
__global__ void transpose_pq_centers_kernel(const float* __restrict__ src, float* __restrict__ dst, int pq_dim,
                                            int pq_len, int book)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pq_dim * pq_len * book) return;
  int c = i % book, l = (i / book) % pq_len, s = i / (book * pq_len);
  dst[i] = src[((int64_t)s * book + c) * pq_len + l];
}

// Modified Gram-Schmidt QR of an n x n Gaussian matrix; returns the top-left [rot_dim, dim] block of Q
// (orthonormal columns when rot_dim >= dim). Reference: make_rotation_matrix, seed 7; the RNG stream is
// RAFT-specific, so only the distribution is reproduced.
std::vector<float> make_rotation_matrix(uint32_t rot_dim, uint32_t dim, bool force_random)
{
  std::vector<float> out((size_t)rot_dim * dim, 0.f);
  if (!force_random && rot_dim == dim) {
    for (uint32_t i = 0; i < dim; ++i) out[(size_t)i * dim + i] = 1.f;
    return out;
  }
  const uint32_t n = std::max(rot_dim, dim);
  std::mt19937_64 rng(7ULL);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> q((size_t)n * n);  // column-major: column j at q[j*n ...]
  for (auto& v : q) v = nd(rng);
  for (uint32_t j = 0; j < n; ++j) {
    double* cj = &q[(size_t)j * n];
    for (int pass = 0; pass < 2; ++pass) {  // re-orthogonalise once for fp safety
      for (uint32_t i = 0; i < j; ++i) {
        const double* ci = &q[(size_t)i * n];
        double dot       = 0;
        for (uint32_t t = 0; t < n; ++t) dot += ci[t] * cj[t];
        for (uint32_t t = 0; t < n; ++t) cj[t] -= dot * ci[t];
      }
    }
    double nrm = 0;
    for (uint32_t t = 0; t < n; ++t) nrm += cj[t] * cj[t];
    nrm = std::sqrt(nrm);
    for (uint32_t t = 0; t < n; ++t) cj[t] /= nrm;
  }
  for (uint32_t r = 0; r < rot_dim; ++r)
    for (uint32_t c = 0; c < dim; ++c) out[(size_t)r * dim + c] = (float)q[(size_t)c * n + r];
  return out;
}

// ------------------------------------------------------------------ encode + pack
__device__ inline size_t code_chunk_addr(int64_t flat_row, uint32_t n_chunks, uint32_t chunk)
{
  return (((size_t)(flat_row >> 6) * n_chunks + chunk) * 64 + (size_t)(flat_row & 63)) * 16;
}

struct encode_args {
  const float* rx;            // [batch, rot_dim] rotated rows of this batch (sorted order)
  const uint32_t* perm;       // [n_new] new-row ids sorted by label (whole extend call)
  const uint32_t* labels;     // [n_new] label of each new row
  const uint32_t* new_off;    // [n_lists + 1] first sorted position of each list among the new rows
  const uint32_t* old_sizes;  // [n_lists] list sizes before this extend
  const uint32_t* list_off;   // [n_lists + 1] flat row offsets of the new layout
  const float* centers_rot;
  const float* pq_centers;
  const int64_t* new_ids;     // optional user ids (device), indexed by new-row id
  int64_t id_base;            // used when new_ids == nullptr
  int64_t j0, batch;          // sorted positions [j0, j0 + batch)
  uint32_t rot_dim, pq_dim, pq_len, pq_bits, book, n_chunks, cpc;
  uint32_t dc_sub;            // subspaces staged in LDS per pass (column tile of the residuals)
  int per_cluster;            // codebook of a row = pq_centers[list] instead of pq_centers[subspace]
  uint8_t* codes;
  int64_t* indices;
  bool pack_each_pass = false;  // the codes of 64 rows do not fit the LDS at once (pq_dim in the thousands): packed per pass
};

// bit-packs the staged codes of 64 rows into their 16-byte chunks: one thread per (row, chunk); ctile holds the codes of
// subspaces [sub0, ...) with row pitch ct_ld
__device__ inline void pack_codes(const encode_args& a, const uint8_t* ctile, uint32_t ct_ld, uint32_t sub0, uint32_t n_ch,
                                  const int64_t* flat_row, int tid)
{
  const uint32_t ch0 = sub0 / a.cpc;
  for (uint32_t t = tid; t < 64 * n_ch; t += 256) {
    uint32_t r = t & 63, ch = t >> 6;
    if (flat_row[r] < 0) continue;
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < a.cpc; ++b) {
      uint32_t s = ch * a.cpc + b;
      if (sub0 + s >= a.pq_dim) break;
      uint32_t code = ctile[r * ct_ld + s];
      uint32_t bit  = b * a.pq_bits;
      w[bit >> 5] |= code << (bit & 31);
      if ((bit & 31) + a.pq_bits > 32) w[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
    }
    *reinterpret_cast<uint4*>(a.codes + code_chunk_addr(flat_row[r], a.n_chunks, ch0 + ch)) =
      make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// 64 sorted rows per workgroup; wave w encodes subspaces w, w+4, ...; lane = row.
__global__ __launch_bounds__(256) void encode_kernel(encode_args a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int dc      = a.dc_sub * a.pq_len;                                  // columns staged per pass
  const int ldr     = dc + 1;
  float* r_tile     = reinterpret_cast<float*>(smem);                       // [64][dc + 1]
  int64_t* flat_row = reinterpret_cast<int64_t*>(r_tile + (((size_t)64 * ldr + 1) & ~size_t(1)));  // [64]
  uint32_t* lab     = reinterpret_cast<uint32_t*>(flat_row + 64);           // [64]
  uint8_t* ctile    = reinterpret_cast<uint8_t*>(lab + 64);                 // [64][pq_dim], or [64][dc_sub] packed pass by pass
  const uint32_t ct_ld = a.pack_each_pass ? a.dc_sub : a.pq_dim;

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t jb0 = (int64_t)blockIdx.x * 64;  // first batch-local row of this block

  if (tid < 64) {
    int64_t jb = jb0 + tid;
    if (jb < a.batch) {
      int64_t j    = a.j0 + jb;
      uint32_t row = a.perm[j];
      uint32_t L   = a.labels[row];
      int64_t fr   = (int64_t)a.list_off[L] + a.old_sizes[L] + (j - (int64_t)a.new_off[L]);
      flat_row[tid] = fr;
      lab[tid]      = L;
      a.indices[fr] = a.new_ids ? a.new_ids[row] : a.id_base + (int64_t)row;
    } else {
      flat_row[tid] = -1;
      lab[tid]      = 0;
    }
  }
  __syncthreads();
  for (uint32_t sub0 = 0; sub0 < a.pq_dim; sub0 += a.dc_sub) {
    const int col0  = sub0 * a.pq_len;
    const int ncols = min((int)a.rot_dim - col0, dc);
    for (int t = tid; t < 64 * ncols; t += 256) {
      int r = t / ncols, d = t % ncols;
      float v = 0.f;
      if (flat_row[r] >= 0)
        v = a.rx[(jb0 + r) * a.rot_dim + col0 + d] - a.centers_rot[(int64_t)lab[r] * a.rot_dim + col0 + d];
      r_tile[r * ldr + d] = v;
    }
    __syncthreads();
    const uint32_t sub_end = min(a.pq_dim, sub0 + a.dc_sub);
    for (uint32_t s0 = sub0 + wave; s0 < sub_end; s0 += 4) {
      const uint32_t s    = __builtin_amdgcn_readfirstlane(s0);
      // PER_CLUSTER: every lane (row) reads the codebook of its own list (rows are sorted by list, so a wave
      // touches one or two codebooks)
      const float* pq     = a.pq_centers + (size_t)(a.per_cluster ? lab[lane] : s) * a.pq_len * a.book;
      const float* rrow   = r_tile + lane * ldr + (s - sub0) * a.pq_len;
      float best          = INFINITY;
      uint32_t code       = 0;
      if (a.pq_len <= 8) {
        float rv[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) rv[l] = l < (int)a.pq_len ? rrow[l] : 0.f;
        for (uint32_t c = 0; c < a.book; ++c) {
          float d = 0.f;
#pragma unroll
          for (int l = 0; l < 8; ++l) {
            if (l < (int)a.pq_len) {
              float t = rv[l] - pq[l * a.book + c];
              d       = __fmaf_rn(t, t, d);
            }
          }
          if (d < best) { best = d; code = c; }
        }
      } else {
        for (uint32_t c = 0; c < a.book; ++c) {
          float d = 0.f;
          for (uint32_t l = 0; l < a.pq_len; ++l) {
            float t = rrow[l] - pq[l * a.book + c];
            d       = __fmaf_rn(t, t, d);
          }
          if (d < best) { best = d; code = c; }
        }
      }
      ctile[lane * ct_ld + (a.pack_each_pass ? s - sub0 : s)] = (uint8_t)code;
    }
    __syncthreads();
    if (a.pack_each_pass) {  // (dc_sub is a multiple of the codes per chunk: a pass ends on a chunk boundary)
      pack_codes(a, ctile, ct_ld, sub0, (sub_end - sub0 + a.cpc - 1) / a.cpc, flat_row, tid);
      __syncthreads();
    }
  }

  if (!a.pack_each_pass) pack_codes(a, ctile, a.pq_dim, 0u, a.n_chunks, flat_row, tid);
}

// copy the old flat arrays into the new layout (lists keep their in-list positions)
__global__ void relocate_lists_kernel(const uint8_t* __restrict__ old_codes, const int64_t* __restrict__ old_ids,
                                      const uint32_t* __restrict__ old_off, const uint32_t* __restrict__ old_sizes,
                                      const uint32_t* __restrict__ new_off, uint32_t n_chunks,
                                      uint8_t* __restrict__ codes, int64_t* __restrict__ ids)
{
  const uint32_t L  = blockIdx.x;
  const uint32_t sz = old_sizes[L];
  const int64_t so = old_off[L], dn = new_off[L];
  for (uint32_t i = threadIdx.x; i < sz; i += blockDim.x) ids[dn + i] = old_ids[so + i];
  const size_t n_groups = (sz + 63) / 64;
  const size_t n16      = n_groups * n_chunks * 64;  // 16-byte words
  const uint4* s = reinterpret_cast<const uint4*>(old_codes + (size_t)(so >> 6) * n_chunks * 1024);
  uint4* d       = reinterpret_cast<uint4*>(codes + (size_t)(dn >> 6) * n_chunks * 1024);
  for (size_t i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
}

__device__ inline uint32_t get_code(const uint8_t* codes, uint32_t n_chunks, uint32_t cpc, uint32_t pq_bits,
                                    int64_t flat_row, uint32_t s)
{
  uint32_t ch = s / cpc, slot = s % cpc;
  const uint8_t* base = codes + code_chunk_addr(flat_row, n_chunks, ch);
  uint32_t bit = slot * pq_bits;
  uint32_t v   = base[bit >> 3];
  if ((bit & 7) + pq_bits > 8) v |= (uint32_t)base[(bit >> 3) + 1] << 8;
  return (v >> (bit & 7)) & ((1u << pq_bits) - 1u);
}

// contiguous little-endian bitstream per row (reference bitfield_ref_t, ivf_pq_codepacking.cuh:22-52)
__global__ void unpack_list_kernel(const uint8_t* __restrict__ codes, uint32_t n_chunks, uint32_t cpc,
                                   uint32_t pq_bits, uint32_t pq_dim, int64_t flat_row0, uint32_t n_take,
                                   uint32_t bytes_per_row, uint8_t* __restrict__ out)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_take * bytes_per_row) return;
  uint32_t r = (uint32_t)(i / bytes_per_row), byte = (uint32_t)(i % bytes_per_row);
  uint32_t v = 0;
  for (uint32_t b = 0; b < 8; ++b) {
    uint32_t pos = byte * 8 + b;
    uint32_t s   = pos / pq_bits;
    if (s >= pq_dim) break;
    uint32_t code = get_code(codes, n_chunks, cpc, pq_bits, flat_row0 + r, s);
    v |= ((code >> (pos % pq_bits)) & 1u) << b;
  }
  out[i] = (uint8_t)v;
}

uint32_t calculate_pq_dim(uint32_t dim)  // ivf_pq_index.cu:612-622
{
  if (dim >= 128) dim /= 2;
  uint32_t r = dim / 32 * 32;
  if (r > 0) return r;
  r = 1;
  while ((r << 1) <= dim) r <<= 1;
  return r;
}

void train_per_subset(resources& res, ivf_pq_index& idx, int64_t n_train, const float* trainset,
                      const uint32_t* labels, uint32_t kmeans_n_iters, uint32_t max_train_points_per_pq_code)
{
  const int64_t pq_n_rows = std::min<int64_t>((int64_t)max_train_points_per_pq_code * idx.pq_book, n_train);
  dev_buf<float> resid(res, (size_t)pq_n_rows * idx.rot_dim);
  pairwise_distance<float, float>(res, trainset, pq_n_rows, idx.dim, idx.rotation.data(), idx.rot_dim, idx.dim,
                                  idx.dim, nullptr, nullptr, M_InnerProduct, resid.data(), idx.rot_dim);
  hipLaunchKernelGGL(subtract_center_kernel, dim3(nblk(pq_n_rows * idx.rot_dim, 256)), dim3(256), 0, res.stream,
                     resid.data(), pq_n_rows, (int)idx.rot_dim, labels, idx.centers_rot.data());
  dev_buf<float> centers_tmp(res, (size_t)idx.pq_dim * idx.pq_book * idx.pq_len);
  dev_buf<uint32_t> sub_labels(res, pq_n_rows), sub_sizes(res, idx.pq_book);
  for (uint32_t s = 0; s < idx.pq_dim; ++s) {
    kmeans_build_clusters(res, resid.data() + (size_t)s * idx.pq_len, pq_n_rows, idx.rot_dim, (int)idx.pq_len,
                          (int)idx.pq_book, (int)kmeans_n_iters,
                          centers_tmp.data() + (size_t)s * idx.pq_book * idx.pq_len, sub_labels.data(),
                          sub_sizes.data());
  }
  int total = idx.pq_dim * idx.pq_len * idx.pq_book;
  hipLaunchKernelGGL(transpose_pq_centers_kernel, dim3(nblk(total, 256)), dim3(256), 0, res.stream,
                     centers_tmp.data(), idx.pq_centers.data(), (int)idx.pq_dim, (int)idx.pq_len, (int)idx.pq_book);
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ ids, int64_t cnt,
                                   int64_t dim, float* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt * dim) out[i] = src[(int64_t)ids[i / dim] * dim + i % dim];
}
__global__ void subtract_one_center_kernel(float* __restrict__ resid, int64_t n, int rot_dim,
                                           const float* __restrict__ center_rot)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * rot_dim) resid[i] -= center_rot[i % rot_dim];
}

// PER_CLUSTER codebooks (ivf_pq_build.cuh:410-497): for every list, the rotated residuals of its training rows,
// read as pq_len-dimensional points of ALL subspaces, are clustered into 2^pq_bits codes. Only the first
// max_train_points_per_pq_code * max(book, pq_dim) points of a list are used, i.e. about that many / pq_dim rows.
void train_per_cluster(resources& res, ivf_pq_index& idx, int64_t n_train, const float* trainset,
                       const uint32_t* labels, uint32_t kmeans_n_iters, uint32_t max_train_points_per_pq_code)
{
  dev_buf<uint32_t> perm(res, n_train), off(res, idx.n_lists + 1);
  group_by_label(res, labels, n_train, idx.n_lists, perm.data(), off.data());
  std::vector<uint32_t> h_off = to_host(res, off.data(), idx.n_lists + 1);
  const size_t big_enough = (size_t)max_train_points_per_pq_code * std::max<size_t>(idx.pq_book, idx.pq_dim);
  const int64_t cap_rows  = (int64_t)((big_enough + idx.pq_dim - 1) / idx.pq_dim);
  dev_buf<float> xb(res, (size_t)cap_rows * idx.dim), rx(res, (size_t)cap_rows * idx.rot_dim);
  dev_buf<float> centers_tmp(res, (size_t)idx.n_lists * idx.pq_book * idx.pq_len);
  HIP_TRY(hipMemsetAsync(centers_tmp.data(), 0, centers_tmp.bytes(), res.stream));
  dev_buf<uint32_t> sub_labels(res, (size_t)cap_rows * idx.pq_dim), sub_sizes(res, idx.pq_book);
  for (uint32_t l = 0; l < idx.n_lists; ++l) {
    const int64_t cnt = std::min<int64_t>(h_off[l + 1] - h_off[l], cap_rows);
    const int64_t pq_n_rows = (int64_t)std::min<size_t>(big_enough, (size_t)cnt * idx.pq_dim);
    // (the reference trains every non-empty list, however few points it has - ivf_pq_build.cuh:447-449 skips empty lists
    // only: fewer points than codes leave some codes on top of each other, not the list without a codebook. Rounds 1-4
    // skipped lists below 2^pq_bits points: found by the reference's own small_dims_per_cluster table, round 5)
    if (cnt == 0) continue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk(cnt * idx.dim, 256)), dim3(256), 0, res.stream, trainset,
                       perm.data() + h_off[l], cnt, (int64_t)idx.dim, xb.data());
    pairwise_distance<float, float>(res, xb.data(), cnt, idx.dim, idx.rotation.data(), idx.rot_dim, idx.dim, idx.dim,
                                    nullptr, nullptr, M_InnerProduct, rx.data(), idx.rot_dim);
    hipLaunchKernelGGL(subtract_one_center_kernel, dim3(nblk(cnt * idx.rot_dim, 256)), dim3(256), 0, res.stream,
                       rx.data(), cnt, (int)idx.rot_dim, idx.centers_rot.data() + (size_t)l * idx.rot_dim);
    kmeans_build_clusters(res, rx.data(), pq_n_rows, idx.pq_len, (int)idx.pq_len, (int)idx.pq_book,
                          (int)kmeans_n_iters, centers_tmp.data() + (size_t)l * idx.pq_book * idx.pq_len,
                          sub_labels.data(), sub_sizes.data());
  }
  const int64_t total = (int64_t)idx.n_lists * idx.pq_len * idx.pq_book;
  CUVS_EXPECTS(total < (int64_t(1) << 31), "ivf_pq: PER_CLUSTER codebooks too large");
  hipLaunchKernelGGL(transpose_pq_centers_kernel, dim3(nblk(total, 256)), dim3(256), 0, res.stream,
                     centers_tmp.data(), idx.pq_centers.data(), (int)idx.n_lists, (int)idx.pq_len, (int)idx.pq_book);
}

}  // namespace

void ivf_pq_set_centers(resources& res, ivf_pq_index& idx, const float* centers_flat)
{
  // centers_flat: device [n_lists, dim]
  idx.center_norms = dev_buf<float>::persistent(idx.n_lists);
  row_norms<float>(res, centers_flat, idx.n_lists, idx.dim, idx.dim, idx.center_norms.data(), false);
  idx.centers = dev_buf<float>::persistent((size_t)idx.n_lists * idx.dim_ext);
  hipLaunchKernelGGL(pad_centers_kernel, dim3(nblk((int64_t)idx.n_lists * idx.dim_ext, 256)), dim3(256), 0,
                     res.stream, centers_flat, idx.center_norms.data(), (int)idx.n_lists, (int)idx.dim,
                     (int)idx.dim_ext, idx.centers.data());
  idx.centers_rot = dev_buf<float>::persistent((size_t)idx.n_lists * idx.rot_dim);
  pairwise_distance<float, float>(res, centers_flat, idx.n_lists, idx.dim, idx.rotation.data(), idx.rot_dim,
                                  idx.dim, idx.dim, nullptr, nullptr, M_InnerProduct, idx.centers_rot.data(),
                                  idx.rot_dim);
}

std::unique_ptr<ivf_pq_index> ivf_pq_make_empty(resources& res, const ivf_pq_build_params& p, elem_t et, int64_t dim)
{
  CUVS_EXPECTS(p.metric == M_L2Expanded || p.metric == M_L2SqrtExpanded || p.metric == M_L2Unexpanded ||
                 p.metric == M_L2SqrtUnexpanded || p.metric == M_InnerProduct || p.metric == M_CosineExpanded,
               "ivf_pq: unsupported metric %d (L2, inner product and cosine are built)", p.metric);
  CUVS_EXPECTS(p.pq_bits >= 4 && p.pq_bits <= 8, "ivf_pq: pq_bits must be within [4, 8]");
  CUVS_EXPECTS(p.codebook_kind == 0 || p.codebook_kind == 1, "ivf_pq: invalid codebook_gen value %d", p.codebook_kind);
  auto idx           = std::make_unique<ivf_pq_index>();
  idx->codes_layout  = p.codes_layout;
  idx->metric        = p.metric;
  idx->codebook_kind = p.codebook_kind;
  idx->dtype         = et;
  idx->n_lists       = p.n_lists;
  idx->dim           = (uint32_t)dim;
  idx->dim_ext       = (uint32_t)round_up(dim + 1, 8);
  idx->pq_dim        = p.pq_dim == 0 ? calculate_pq_dim((uint32_t)dim) : p.pq_dim;
  idx->pq_bits       = p.pq_bits;
  idx->pq_book       = 1u << p.pq_bits;
  idx->pq_len        = (uint32_t)ceil_div(dim, idx->pq_dim);
  idx->rot_dim       = idx->pq_len * idx->pq_dim;
  idx->codes_per_chunk = 128 / p.pq_bits;
  idx->n_chunks        = (uint32_t)ceil_div(idx->pq_dim, idx->codes_per_chunk);
  auto rot      = make_rotation_matrix(idx->rot_dim, idx->dim, p.force_random_rotation);
  idx->rotation = dev_buf<float>::persistent(rot.size());
  copy_async(res, idx->rotation.data(), rot.data(), rot.size() * sizeof(float));
  sync(res);
  idx->pq_centers   = dev_buf<float>::persistent((size_t)(p.codebook_kind == 1 ? idx->n_lists : idx->pq_dim) * idx->pq_len *
                                                 idx->pq_book);
  HIP_TRY(hipMemsetAsync(idx->pq_centers.data(), 0, idx->pq_centers.bytes(), res.stream));
  idx->list_sizes   = dev_buf<uint32_t>::persistent(idx->n_lists);
  idx->list_offsets = dev_buf<uint32_t>::persistent(idx->n_lists + 1);
  HIP_TRY(hipMemsetAsync(idx->list_sizes.data(), 0, idx->list_sizes.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(idx->list_offsets.data(), 0, idx->list_offsets.bytes(), res.stream));
  idx->h_list_sizes.assign(idx->n_lists, 0);
  idx->h_list_offsets.assign(idx->n_lists + 1, 0);
  return idx;
}

std::unique_ptr<ivf_pq_index> ivf_pq_build(resources& res, const ivf_pq_build_params& p, const void* data,
                                           elem_t et, int64_t n, int64_t dim, bool is_host)
{
  CUVS_EXPECTS(n > 0 && dim > 0, "empty dataset");
  CUVS_EXPECTS(n >= p.n_lists, "number of rows can't be less than n_lists");
  auto idx = ivf_pq_make_empty(res, p, et, dim);

  // trainset: every `ratio`-th row (reference samples rows at random with seed 137, ivf_pq_build.cuh:1266-1319)
  const int64_t ratio   = std::max<int64_t>(1, n / std::max<int64_t>((int64_t)(p.kmeans_trainset_fraction * n), p.n_lists));
  const int64_t n_train = n / ratio;
  dev_buf<float> trainset(res, (size_t)n_train * dim);
  {
    dev_buf<uint32_t> ids(res, n_train);
    hipLaunchKernelGGL(strided_ids_kernel, dim3(nblk(n_train, 256)), dim3(256), 0, res.stream, ids.data(), n_train,
                       ratio);
    load_gather_as_float(res, data, et, is_host, dim, ids.data(), n_train, trainset.data());
    // cosine = inner product on unit-length rows (ivf_pq_build.cuh:159-166,1336-1348): rows are normalised
    // wherever they are read (training, labelling, encoding), centres after the k-means, queries at search
    if (p.metric == M_CosineExpanded) normalize_rows(res, trainset.data(), n_train, dim);
  }
  dev_buf<float> centers_flat(res, (size_t)p.n_lists * dim);
  kmeans_params kp;
  kp.n_iters = (int)p.kmeans_n_iters;
  kmeans_balanced_fit(res, trainset.data(), n_train, dim, (int)p.n_lists, kp, centers_flat.data());
  if (p.metric == M_CosineExpanded) normalize_rows(res, centers_flat.data(), p.n_lists, dim);
  dev_buf<uint32_t> labels(res, n_train);
  kmeans_predict<float>(res, trainset.data(), n_train, dim, centers_flat.data(), (int)p.n_lists, labels.data());
  ivf_pq_set_centers(res, *idx, centers_flat.data());
  if (p.codebook_kind == 1)
    train_per_cluster(res, *idx, n_train, trainset.data(), labels.data(), p.kmeans_n_iters,
                      p.max_train_points_per_pq_code);
  else
    train_per_subset(res, *idx, n_train, trainset.data(), labels.data(), p.kmeans_n_iters,
                     p.max_train_points_per_pq_code);
  trainset.release();
  labels.release();
  if (p.add_data_on_build) ivf_pq_extend(res, *idx, data, et, n, is_host, nullptr, false);
  sync(res);
  return idx;
}

__global__ void drop_foreign_labels_kernel(uint32_t* __restrict__ labels, int64_t n, uint32_t n_lists, uint32_t world,
                                           uint32_t rank, const int32_t* __restrict__ owner)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t L = labels[i];
  if ((owner != nullptr ? (uint32_t)owner[L] : L % world) != rank) labels[i] = n_lists;
}

// LDS plan of encode_kernel: dc_sub subspaces staged per pass and whether the codes of the 64 rows are packed pass by
// pass (pq_dim in the thousands - 6144-d rows at pq_len 2 hold 3072 codes per row: 192 KiB for 64 rows) or once at the end
struct encode_plan { uint32_t dc_sub; bool pack_each_pass; size_t smem; };
static encode_plan plan_encode(const ivf_pq_index& idx)
{
  CUVS_EXPECTS(idx.pq_len <= 240, "encode: pq_len %u too large", idx.pq_len);
  encode_plan p;
  p.dc_sub         = std::min<uint32_t>(idx.pq_dim, std::max<uint32_t>(1, 240 / idx.pq_len));
  p.pack_each_pass = false;
  auto bytes = [&](uint32_t dc_sub, uint32_t code_cols) {
    return ((size_t)64 * (dc_sub * idx.pq_len + 1) + 2) * sizeof(float) + 64 * sizeof(int64_t) + 64 * sizeof(uint32_t) + (size_t)64 * code_cols;
  };
  p.smem = bytes(p.dc_sub, idx.pq_dim);
  if (p.smem > 160 * 1024) {
    // a pass must end on a chunk boundary: dc_sub a multiple of the codes per 16-byte chunk
    const uint32_t cpc = idx.codes_per_chunk;
    CUVS_EXPECTS(p.dc_sub >= cpc, "encode: pq_dim %u x pq_len %u does not fit the LDS", idx.pq_dim, idx.pq_len);
    p.dc_sub         = p.dc_sub / cpc * cpc;
    p.pack_each_pass = true;
    p.smem           = bytes(p.dc_sub, p.dc_sub);
  }
  CUVS_EXPECTS(p.smem <= 160 * 1024, "encode: tile does not fit LDS");
  return p;
}

void ivf_pq_extend(resources& res, ivf_pq_index& idx, const void* data, elem_t et, int64_t n_new, bool is_host,
                   const int64_t* new_ids, bool ids_on_host)
{
  idx.shard_stats_valid = false;  // a list shard: the rows / non-empty lists of ALL ranks are exchanged again by the next search (extend is collective for a shard)
  if (n_new == 0) return;
  if (!idx.dtype_known) { idx.dtype = et; idx.dtype_known = true; }
  CUVS_EXPECTS(et == idx.dtype, "extend: vector dtype differs from the index dtype");
  CUVS_EXPECTS(new_ids != nullptr || idx.size == 0, "You must pass data indices when the index is non-empty.");
  CUVS_EXPECTS(idx.size + n_new < (int64_t(1) << 32) - 64 * (int64_t)idx.n_lists, "index too large for 32-bit row offsets");
  const int64_t dim = idx.dim;

  dev_buf<int64_t> ids_dev;
  if (new_ids && ids_on_host) {
    ids_dev = dev_buf<int64_t>(res, n_new);
    copy_async(res, ids_dev.data(), new_ids, n_new * sizeof(int64_t));
    new_ids = ids_dev.data();
  }

  // ---- 1. labels of the new rows (L2 argmin even for inner product: coarse_clustering_metric)
  dev_buf<float> centers_flat(res, (size_t)idx.n_lists * dim);
  HIP_TRY(hipMemcpy2DAsync(centers_flat.data(), dim * sizeof(float), idx.centers.data(), idx.dim_ext * sizeof(float),
                           dim * sizeof(float), idx.n_lists, hipMemcpyDeviceToDevice, res.stream));
  dev_buf<uint32_t> labels(res, n_new);
  const int64_t batch_rows = std::max<int64_t>(1024, std::min<int64_t>(n_new, (int64_t(1) << 28) / dim));
  {
    dev_buf<float> xb(res, (size_t)std::min(batch_rows, n_new) * dim);
    for (int64_t r0 = 0; r0 < n_new; r0 += batch_rows) {
      int64_t cnt = std::min(batch_rows, n_new - r0);
      load_range_as_float(res, data, et, is_host, dim, r0, cnt, xb.data());
      if (idx.metric == M_CosineExpanded) normalize_rows(res, xb.data(), cnt, dim);
      fused_l2_argmin<float>(res, xb.data(), cnt, dim, centers_flat.data(), idx.n_lists, dim,
                             idx.center_norms.data(), labels.data() + r0, nullptr);
    }
  }
  // ---- 2. group the new rows by list. A list-sharded index (ivf_pq.hpp) keeps only the rows of the lists this rank
  // owns: the others are relabelled to an extra bucket at the end of the grouping and never encoded.
  const bool sharded = idx.shard_world > 1;
  if (sharded)
    hipLaunchKernelGGL(drop_foreign_labels_kernel, dim3(nblk(n_new, 256)), dim3(256), 0, res.stream, labels.data(), n_new,
                       idx.n_lists, (uint32_t)idx.shard_world, (uint32_t)idx.shard_rank, idx.list_owner.data());
  const uint32_t n_buckets = idx.n_lists + (sharded ? 1u : 0u);
  dev_buf<uint32_t> perm(res, n_new), new_off(res, n_buckets + 1);
  group_by_label(res, labels.data(), n_new, n_buckets, perm.data(), new_off.data());
  std::vector<uint32_t> h_new_off = to_host(res, new_off.data(), n_buckets + 1);
  const int64_t n_all = n_new;
  n_new               = h_new_off[idx.n_lists];  // rows that stay on this rank (all of them when not sharded)
  (void)n_all;

  // ---- 3. new flat layout
  std::vector<uint32_t> sizes(idx.n_lists), offs(idx.n_lists + 1);
  int64_t total = 0;
  for (uint32_t L = 0; L < idx.n_lists; ++L) {
    sizes[L] = idx.h_list_sizes[L] + (h_new_off[L + 1] - h_new_off[L]);
    offs[L]  = (uint32_t)total;
    total += round_up(sizes[L], kPqGroup);
  }
  offs[idx.n_lists] = (uint32_t)total;
  auto codes        = dev_buf<uint8_t>::persistent((size_t)total * idx.n_chunks * 16);
  auto indices      = dev_buf<int64_t>::persistent((size_t)total);
  HIP_TRY(hipMemsetAsync(codes.data(), 0, codes.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(indices.data(), 0xff, indices.bytes(), res.stream));
  dev_buf<uint32_t> d_new_list_off(res, idx.n_lists + 1);
  copy_async(res, d_new_list_off.data(), offs.data(), offs.size() * sizeof(uint32_t));
  if (idx.size > 0) {
    hipLaunchKernelGGL(relocate_lists_kernel, dim3(idx.n_lists), dim3(256), 0, res.stream, idx.codes.data(),
                       idx.indices.data(), idx.list_offsets.data(), idx.list_sizes.data(), d_new_list_off.data(),
                       idx.n_chunks, codes.data(), indices.data());
  }

  // ---- 4. encode in sorted order
  {
    const int64_t eb = std::max<int64_t>(64, std::min<int64_t>(n_new, (int64_t(1) << 27) / std::max<int64_t>(dim, idx.rot_dim)) / 64 * 64);
    dev_buf<float> xb(res, (size_t)eb * dim), rx(res, (size_t)eb * idx.rot_dim);
    const encode_plan plan = plan_encode(idx);
    const uint32_t dc_sub  = plan.dc_sub;
    const size_t smem      = plan.smem;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    for (int64_t j0 = 0; j0 < n_new; j0 += eb) {
      int64_t cnt = std::min(eb, n_new - j0);
      load_gather_as_float(res, data, et, is_host, dim, perm.data() + j0, cnt, xb.data());
      if (idx.metric == M_CosineExpanded) normalize_rows(res, xb.data(), cnt, dim);
      pairwise_distance<float, float>(res, xb.data(), cnt, dim, idx.rotation.data(), idx.rot_dim, dim, dim, nullptr,
                                      nullptr, M_InnerProduct, rx.data(), idx.rot_dim);
      encode_args a;
      a.rx = rx.data(); a.perm = perm.data(); a.labels = labels.data(); a.new_off = new_off.data();
      a.old_sizes = idx.list_sizes.data(); a.list_off = d_new_list_off.data();
      a.centers_rot = idx.centers_rot.data(); a.pq_centers = idx.pq_centers.data();
      a.new_ids = new_ids; a.id_base = idx.size; a.j0 = j0; a.batch = cnt;
      a.rot_dim = idx.rot_dim; a.pq_dim = idx.pq_dim; a.pq_len = idx.pq_len; a.pq_bits = idx.pq_bits;
      a.book = idx.pq_book; a.n_chunks = idx.n_chunks; a.cpc = idx.codes_per_chunk; a.dc_sub = dc_sub;
      a.per_cluster = idx.codebook_kind == 1; a.pack_each_pass = plan.pack_each_pass;
      a.codes = codes.data(); a.indices = indices.data();
      hipLaunchKernelGGL(encode_kernel, dim3(nblk(cnt, 64)), dim3(256), smem, res.stream, a);
    }
    HIP_TRY(hipGetLastError());
  }
  sync(res);  // old arrays are released below; kernels reading them must be done
  idx.codes   = std::move(codes);
  idx.indices = std::move(indices);
  copy_async(res, idx.list_sizes.data(), sizes.data(), sizes.size() * sizeof(uint32_t));
  copy_async(res, idx.list_offsets.data(), offs.data(), offs.size() * sizeof(uint32_t));
  sync(res);
  idx.h_list_sizes   = sizes;
  idx.h_list_offsets = offs;
  idx.size += n_new;
  idx.padded_rows = total;
}

// cuvsIvfPqTransform (c/src/neighbors/ivf_pq.cpp:595-630, ivf_pq::transform): labels[n] = list of every row and
// codes[n, ceil(pq_dim * pq_bits / 8)] = its PQ code as a contiguous bitstream - the extend path's labelling and
// encoding, with the codes written row by row (identity placement) instead of into the lists.
void ivf_pq_transform(resources& res, const ivf_pq_index& idx, const void* data, elem_t et, int64_t n, uint32_t* out_labels,
                      uint8_t* out_codes)
{
  if (n == 0) return;
  CUVS_EXPECTS(n < (int64_t(1) << 31), "transform: at most 2^31 rows per call");
  const int64_t dim = idx.dim;
  dev_buf<float> centers_flat(res, (size_t)idx.n_lists * dim);
  HIP_TRY(hipMemcpy2DAsync(centers_flat.data(), dim * sizeof(float), idx.centers.data(), idx.dim_ext * sizeof(float),
                           dim * sizeof(float), idx.n_lists, hipMemcpyDeviceToDevice, res.stream));
  const int64_t eb = std::max<int64_t>(64, std::min<int64_t>(n, (int64_t(1) << 26) / std::max<int64_t>(dim, idx.rot_dim)) / 64 * 64);
  dev_buf<float> xb(res, (size_t)eb * dim), rx(res, (size_t)eb * idx.rot_dim);
  dev_buf<uint32_t> perm(res, eb), zeros(res, idx.n_lists + 1);
  dev_buf<int64_t> ids_tmp(res, eb);
  dev_buf<uint8_t> tmp_codes(res, (size_t)eb * idx.n_chunks * 16);
  HIP_TRY(hipMemsetAsync(zeros.data(), 0, zeros.bytes(), res.stream));
  {
    std::vector<uint32_t> h(eb);
    for (int64_t i = 0; i < eb; ++i) h[i] = (uint32_t)i;
    copy_async(res, perm.data(), h.data(), eb * sizeof(uint32_t));
    sync(res);
  }
  CUVS_EXPECTS(idx.pq_len <= 240, "encode: pq_len %u too large", idx.pq_len);
  const encode_plan plan = plan_encode(idx);
  const uint32_t dc_sub  = plan.dc_sub;
  const size_t smem      = plan.smem;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem));
  const uint32_t bpr = (idx.pq_dim * idx.pq_bits + 7) / 8;
  for (int64_t r0 = 0; r0 < n; r0 += eb) {
    const int64_t cnt = std::min(eb, n - r0);
    load_range_as_float(res, data, et, false, dim, r0, cnt, xb.data());
    if (idx.metric == M_CosineExpanded) normalize_rows(res, xb.data(), cnt, dim);
    fused_l2_argmin<float>(res, xb.data(), cnt, dim, centers_flat.data(), idx.n_lists, dim, idx.center_norms.data(),
                           out_labels + r0, nullptr);
    if (out_codes == nullptr) continue;  // labels only (cuvsAmdIvfPqListHistogram)
    pairwise_distance<float, float>(res, xb.data(), cnt, dim, idx.rotation.data(), idx.rot_dim, dim, dim, nullptr, nullptr,
                                    M_InnerProduct, rx.data(), idx.rot_dim);
    HIP_TRY(hipMemsetAsync(tmp_codes.data(), 0, tmp_codes.bytes(), res.stream));
    encode_args a;
    a.rx = rx.data(); a.perm = perm.data(); a.labels = out_labels + r0; a.new_off = zeros.data();
    a.old_sizes = zeros.data(); a.list_off = zeros.data();  // flat row = position in the batch
    a.centers_rot = idx.centers_rot.data(); a.pq_centers = idx.pq_centers.data();
    a.new_ids = nullptr; a.id_base = 0; a.j0 = 0; a.batch = cnt;
    a.rot_dim = idx.rot_dim; a.pq_dim = idx.pq_dim; a.pq_len = idx.pq_len; a.pq_bits = idx.pq_bits;
    a.book = idx.pq_book; a.n_chunks = idx.n_chunks; a.cpc = idx.codes_per_chunk; a.dc_sub = dc_sub;
    a.per_cluster = idx.codebook_kind == 1; a.pack_each_pass = plan.pack_each_pass;
    a.codes = tmp_codes.data(); a.indices = ids_tmp.data();
    hipLaunchKernelGGL(encode_kernel, dim3(nblk(cnt, 64)), dim3(256), smem, res.stream, a);
    hipLaunchKernelGGL(unpack_list_kernel, dim3(nblk(cnt * (int64_t)bpr, 256)), dim3(256), 0, res.stream, tmp_codes.data(),
                       idx.n_chunks, idx.codes_per_chunk, idx.pq_bits, idx.pq_dim, (int64_t)0, (uint32_t)cnt, bpr,
                       out_codes + (size_t)r0 * bpr);
    HIP_TRY(hipGetLastError());
  }
  sync(res);
}

void ivf_pq_unpack_list(resources& res, const ivf_pq_index& idx, uint32_t label, uint32_t offset, uint32_t n_take,
                        uint8_t* out)
{
  CUVS_EXPECTS(label < idx.n_lists, "Expected label to be less than number of lists in the index");
  CUVS_EXPECTS(offset + n_take <= idx.h_list_sizes[label], "unpack: range exceeds the list size");
  if (n_take == 0) return;
  uint32_t bpr  = (idx.pq_dim * idx.pq_bits + 7) / 8;
  int64_t total = (int64_t)n_take * bpr;
  hipLaunchKernelGGL(unpack_list_kernel, dim3(nblk(total, 256)), dim3(256), 0, res.stream, idx.codes.data(),
                     idx.n_chunks, idx.codes_per_chunk, idx.pq_bits, idx.pq_dim,
                     (int64_t)idx.h_list_offsets[label] + offset, n_take, bpr, out);
  HIP_TRY(hipGetLastError());
}

}  // namespace cuvs_amd
