// IVF-PQ index as laid out in MI355X HBM (internal; the C ABI sees it through cuvsIvfPqIndex.addr).
//
// Reference: cuvs::neighbors::ivf_pq::index (cpp/include/cuvs/neighbors/ivf_pq.hpp:476-675,
// cpp/src/neighbors/ivf_pq_index.cu). Differences by design:
//   * all lists live in ONE flat allocation (288 GB HBM: no per-list allocations / pointer tables);
//     list L owns rows [list_offsets[L], list_offsets[L] + list_sizes[L]) of it, offsets are multiples of 64;
//   * codes are interleaved in groups of 64 rows (one wave64 lane per row) instead of the reference's 32
//     (ivf_pq.hpp:222-224,290-297): byte address of (flat row r, 16-byte chunk c) is
//         ((r / 64 * n_chunks + c) * 64 + r % 64) * 16
//     so a wave reads 1 KiB contiguous per chunk. Chunk c holds codes [c*cpc, (c+1)*cpc) bit-packed
//     little-endian, cpc = 128 / pq_bits (16 for 8-bit codes);
//   * in-list order is ascending source row id (stable sort), i.e. deterministic, where the reference
//     allocates slots with atomicAdd (ivf_pq_process_and_fill_codes_impl.cuh:39).
// The externally visible code format (cuvsIvfPqIndexUnpackContiguousListData) is unchanged.
#pragma once
#include "common.hpp"

namespace cuvs_amd {

constexpr uint32_t kPqGroup = 64;  // rows per interleaved group == wave size

struct ivf_pq_index {
  int metric          = 0;
  int codebook_kind   = 0;  // 0 PER_SUBSPACE: one codebook per subspace; 1 PER_CLUSTER: one per list, shared by its subspaces
  elem_t dtype        = elem_t::f32;
  bool dtype_known    = true;   // false after loading a reference-format file (the reference's index is untyped)
  uint32_t n_lists = 0, dim = 0, dim_ext = 0, rot_dim = 0;
  uint32_t pq_dim = 0, pq_bits = 8, pq_len = 0, pq_book = 256;
  uint32_t n_chunks = 0;        // 16-byte chunks per encoded vector
  uint32_t codes_per_chunk = 16;
  int64_t size         = 0;     // number of indexed vectors
  int64_t padded_rows  = 0;     // rows of the flat arrays (sum of list capacities)

  dev_buf<float> centers;       // [n_lists, dim_ext]  (column `dim` = |c|^2, as in the reference)
  dev_buf<float> center_norms;  // [n_lists] canonical |c|^2
  dev_buf<float> centers_rot;   // [n_lists, rot_dim]
  dev_buf<float> rotation;      // [rot_dim, dim]
  dev_buf<float> pq_centers;    // [pq_dim, pq_len, pq_book] (PER_SUBSPACE) or [n_lists, pq_len, pq_book] (PER_CLUSTER)
  dev_buf<uint8_t> codes;       // [padded_rows / 64, n_chunks, 64, 16]
  dev_buf<int64_t> indices;     // [padded_rows] source ids
  dev_buf<uint32_t> list_sizes;   // [n_lists]
  dev_buf<uint32_t> list_offsets; // [n_lists + 1] (rows, multiples of 64)
  std::vector<uint32_t> h_list_sizes, h_list_offsets;

  // reduced-precision copies for search_params.coarse_search_dtype (ivf_pq_index.cu:640-760), built on first use:
  // fp16: centres / |c|^2 / rotation rounded to half (kept as floats); int8: trunc(x * 128) saturated, and the
  // per-centre norm term of the int8 GEMM (the y / z columns of centers_int8, ivf_pq_index.cu:674-732)
  mutable dev_buf<float> coarse_centers_h, coarse_norms_h, coarse_rot_h, coarse_centers_i8, coarse_normterm_i8, coarse_rot_i8;
  // the centres packed in the coarse type for the fp16 / int8 matrix cores (coarse_lowp.hip), built on first use
  mutable dev_buf<uint32_t> coarse_pack_h, coarse_pack_i8;

  // decode tables of the matrix-core filter (ivf_pq_scan3.hip), built on first use and rebuilt when the lists change
  struct scan3_cache {
    dev_buf<uint32_t> cb16;
    dev_buf<uint32_t> row_term;
    float sc = 1.f, cbmax = 0.f, dmax = 0.f;
    bool term_fp32 = false;  // row_term holds fp32 values (pq_filter4_kernel) instead of fp16 (hi, lo) pairs
    const void* codes_ptr = nullptr;
    const void* pq_ptr    = nullptr;
    int64_t rows = -1, size = -1;
    // codes of fewer than 8 bits: a derived copy with one byte per code, laid out like 8-bit codes (16 per chunk)
    dev_buf<uint8_t> codes8;
    const void* codes8_src = nullptr;
    int64_t codes8_rows = -1, codes8_size = -1;
    // the rows DECODED (scaled fp16, MFMA A-operand layout) for the wide filter (ivf_pq_wide.hip): rows x rot_dim x 2 bytes, made
    // when a search first takes that path and the device has the room; w_failed_*: the index state that found none
    dev_buf<uint4> rows16w;
    dev_buf<float> zeros;  // 32 zero floats (the wide filter's row terms of an inner-product search)
    dev_buf<float> cbt;  // the fp32 codebook entry-major ([subspace][256][pq_len]): the wide path's exact scores read whole entries
    const void* w_codes = nullptr;
    const void* w_pq    = nullptr;
    int64_t w_rows = -1, w_size = -1;
    const void* w_failed_codes = nullptr;
    int64_t w_failed_rows = -1, w_failed_size = -1;
  };
  mutable scan3_cache scan3;
  // largest source id held by the lists (-1: empty), computed on demand for the bitset-filter bound check
  mutable int64_t max_id = -1, max_id_rows = -1;
  mutable const void* max_id_ptr = nullptr;

  // List-sharded multi-GPU search (shard_comm.hip): every rank holds the whole model (centres, rotation, codebooks) but
  // only the lists it owns - list L belongs to rank L % shard_world. extend() drops rows of foreign lists, search()
  // scans only owned probes; the per-rank top-k lists are all-gathered and merged. shard_world == 1: not sharded.
  // list_layout of the index as its owner sees it (ivf_pq.hpp:40-90). The device arrays of this library are ALWAYS the 64-row
  // interleaved chunks the kernels read - the C ABI has no accessor for the raw list storage (only the contiguous codes of
  // cuvsIvfPqIndexUnpackContiguousListData) - so a FLAT index differs in what the reference's FLAT index differs in through
  // that ABI: its files hold [size, bytes_per_vector] list records (ivf_pq.hpp:302-338) and searching it is refused
  // (ivf_pq_search.cuh:914-916).
  int codes_layout = 1;
  int shard_rank = 0, shard_world = 1;
  void* shard_comm = nullptr;  // cuvsAmdShardComm* (not owned): all-reduce of the k-th bounds between the scan phases
  // rows / non-empty lists of ALL ranks' shards, exchanged by the first search after the communicator was attached (one
  // small all-gather): with collectives inside the search every rank must choose the same kernels, so what the choice
  // looks at (the mean list length) has to be the same number on every rank (pq3_bound_useful)
  mutable uint64_t shard_global_rows = 0, shard_global_lists = 0;
  mutable bool shard_stats_valid = false;
  // optional owner of every list (cuvsAmdIvfPqSetListOwners: lists dealt by size, cuvsAmdShardDealLists); empty: L % world
  std::vector<int32_t> h_list_owner;
  dev_buf<int32_t> list_owner;
  bool owns(uint32_t L) const
  {
    if (shard_world <= 1) return true;
    return h_list_owner.empty() ? (int)(L % (uint32_t)shard_world) == shard_rank : h_list_owner[L] == shard_rank;
  }

  static float scale(elem_t et)  // kDivisor(T) / kDivisor(float), ann_utils.cuh:134-160
  {
    return et == elem_t::u8 ? 256.0f : (et == elem_t::i8 ? 128.0f : 1.0f);
  }
};

struct ivf_pq_build_params {
  int metric = 0;
  uint32_t n_lists = 1024, kmeans_n_iters = 20;
  double kmeans_trainset_fraction = 0.5;
  uint32_t pq_bits = 8, pq_dim = 0;
  int codebook_kind = 0;
  bool force_random_rotation = false;
  bool add_data_on_build     = true;
  uint32_t max_train_points_per_pq_code = 256;
  int codes_layout = 1;  // cuvsIvfPqListLayout: 0 FLAT, 1 INTERLEAVED (ivf_pq.hpp:40-90)
};

struct ivf_pq_search_params {
  uint32_t n_probes = 20;
  int lut_dtype = 0, internal_distance_dtype = 0;  // hipDataType values: 0 = f32, 2 = f16, 8 = u8(fp8)
  int coarse_search_dtype = 0;                     // 0 = f32, 2 = f16, 3 = i8 (ivf_pq_search.cuh:171-340)
  uint32_t max_internal_batch_size = 4096;
};

// data may be a host or a device pointer (is_host)
std::unique_ptr<ivf_pq_index> ivf_pq_build(resources& res, const ivf_pq_build_params& p, const void* data,
                                           elem_t et, int64_t n, int64_t dim, bool is_host);
void ivf_pq_extend(resources& res, ivf_pq_index& idx, const void* data, elem_t et, int64_t n, bool is_host,
                   const int64_t* new_ids, bool ids_on_host);
// filter_bits: optional bitset over source ids (1 keeps the row; sample_filter.cuh bitset_filter semantics), device memory
void ivf_pq_search(resources& res, const ivf_pq_search_params& p, const ivf_pq_index& idx, const void* queries,
                   elem_t et, int64_t n_queries, int k, int64_t* neighbors, float* distances,
                   const uint32_t* filter_bits = nullptr);
// labels [n] (uint32) and contiguous bit-packed codes [n, ceil(pq_dim*pq_bits/8)] of device rows (cuvsIvfPqTransform)
void ivf_pq_transform(resources& res, const ivf_pq_index& idx, const void* data, elem_t et, int64_t n, uint32_t* out_labels,
                      uint8_t* out_codes);
// [n_take, ceil(pq_dim*pq_bits/8)] contiguous bit-packed codes of list `label` starting at `offset`
// shard_comm.hip: in-place min over all ranks of `count` order-preserving bound keys, on the stream of `res`
void shard_allreduce_min_u32(resources& res, void* comm, uint32_t* keys, size_t count);
// in-place all-gather: rank r's `count` words already sit at buf + r * count; afterwards every rank holds all blocks
void shard_allgather_inplace_u32(resources& res, void* comm, uint32_t* buf, size_t count);

void ivf_pq_unpack_list(resources& res, const ivf_pq_index& idx, uint32_t label, uint32_t offset,
                        uint32_t n_take, uint8_t* out);

}  // namespace cuvs_amd
