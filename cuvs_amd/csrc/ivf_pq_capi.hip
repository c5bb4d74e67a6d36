// cuvsIvfPq* C entry points (drop-in for c/src/neighbors/ivf_pq.cpp) over ivf_pq_build.hip / ivf_pq_search.hip.
#include "ivf_pq.hpp"
#include "ops.hpp"
#include "serialize.hpp"
#include "npy_io.hpp"

#include <cuvs/neighbors/ivf_pq.h>
#include <cuvs_amd/extensions.h>

namespace cuvs_amd {
std::unique_ptr<ivf_pq_index> ivf_pq_make_empty(resources& res, const ivf_pq_build_params& p, elem_t et, int64_t dim);
void ivf_pq_set_centers(resources& res, ivf_pq_index& idx, const float* centers_flat);
}  // namespace cuvs_amd

using namespace cuvs_amd;

namespace {

ivf_pq_index& get_index(cuvsIvfPqIndex_t index)
{
  CUVS_EXPECTS(index != nullptr && index->addr != 0, "IVF-PQ index is not built");
  return *reinterpret_cast<ivf_pq_index*>(index->addr);
}

ivf_pq_build_params to_params(const cuvsIvfPqIndexParams& p)
{
  ivf_pq_build_params b;
  b.metric                       = (int)p.metric;
  b.n_lists                      = p.n_lists;
  b.kmeans_n_iters               = p.kmeans_n_iters;
  b.kmeans_trainset_fraction     = p.kmeans_trainset_fraction;
  b.pq_bits                      = p.pq_bits;
  b.pq_dim                       = p.pq_dim;
  b.codebook_kind                = (int)p.codebook_kind;
  b.force_random_rotation        = p.force_random_rotation;
  b.add_data_on_build            = p.add_data_on_build;
  b.max_train_points_per_pq_code = p.max_train_points_per_pq_code;
  CUVS_EXPECTS(p.codes_layout == CUVS_IVF_PQ_LIST_LAYOUT_INTERLEAVED || p.codes_layout == CUVS_IVF_PQ_LIST_LAYOUT_FLAT,
               "ivf_pq: invalid codes_layout value %d", (int)p.codes_layout);
  b.codes_layout = (int)p.codes_layout;
  return b;
}

const DLDataType kF32{kDLFloat, 32, 1};

}  // namespace

extern "C" {

cuvsError_t cuvsIvfPqIndexParamsCreate(cuvsIvfPqIndexParams_t* params)
{
  return (cuvsError_t)translate_exceptions([=] {
    *params = new cuvsIvfPqIndexParams{L2Expanded, 2.0f, true, 1024, 20, 0.5, 8, 0,
                                       CUVS_IVF_PQ_CODEBOOK_GEN_PER_SUBSPACE, false, false, 256,
                                       CUVS_IVF_PQ_LIST_LAYOUT_INTERLEAVED};
  });
}
cuvsError_t cuvsIvfPqIndexParamsDestroy(cuvsIvfPqIndexParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsIvfPqSearchParamsCreate(cuvsIvfPqSearchParams_t* params)
{
  return (cuvsError_t)translate_exceptions(
    [=] { *params = new cuvsIvfPqSearchParams{20, CUDA_R_32F, CUDA_R_32F, CUDA_R_32F, 4096, 1.0}; });
}
cuvsError_t cuvsIvfPqSearchParamsDestroy(cuvsIvfPqSearchParams_t params)
{
  return (cuvsError_t)translate_exceptions([=] { delete params; });
}
cuvsError_t cuvsIvfPqIndexCreate(cuvsIvfPqIndex_t* index)
{
  return (cuvsError_t)translate_exceptions([=] { *index = new cuvsIvfPqIndex{0, DLDataType{0, 0, 0}}; });
}
cuvsError_t cuvsIvfPqIndexDestroy(cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (!index) return;
    delete reinterpret_cast<ivf_pq_index*>(index->addr);
    delete index;
  });
}

cuvsError_t cuvsIvfPqBuild(cuvsResources_t res_h, cuvsIvfPqIndexParams_t params, DLManagedTensor* dataset_tensor,
                           cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(params && dataset_tensor && index, "null argument");
    auto& ds = dataset_tensor->dl_tensor;
    CUVS_EXPECTS(ds.ndim == 2 && is_c_contiguous(ds), "dataset must be a row-major matrix");
    elem_t et    = elem_of(ds.dtype);
    bool is_host = !is_device_accessible(ds);
    auto idx     = ivf_pq_build(res, to_params(*params), dl_data(ds), et, ds.shape[0], ds.shape[1], is_host);
    delete reinterpret_cast<ivf_pq_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = ds.dtype;
  });
}

cuvsError_t cuvsIvfPqBuildPrecomputed(cuvsResources_t res_h, cuvsIvfPqIndexParams_t params, uint32_t dim,
                                      DLManagedTensor* pq_centers, DLManagedTensor* centers,
                                      DLManagedTensor* centers_rot, DLManagedTensor* rotation_matrix,
                                      cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(params && pq_centers && centers && rotation_matrix && index, "null argument");
    auto idx = ivf_pq_make_empty(res, to_params(*params), elem_t::f32, dim);
    auto& pc = pq_centers->dl_tensor;
    auto& ce = centers->dl_tensor;
    auto& ro = rotation_matrix->dl_tensor;
    CUVS_EXPECTS(dtype_is(pc.dtype, kDLFloat, 32) && dtype_is(ce.dtype, kDLFloat, 32) &&
                   dtype_is(ro.dtype, kDLFloat, 32),
                 "precomputed tensors must be float32");
    CUVS_EXPECTS(pc.ndim == 3 && pc.shape[0] == (idx->codebook_kind == 1 ? idx->n_lists : idx->pq_dim) &&
                   pc.shape[1] == idx->pq_len && pc.shape[2] == idx->pq_book,
                 "pq_centers must have shape [pq_dim, pq_len, 2^pq_bits] (PER_SUBSPACE) or [n_lists, pq_len, 2^pq_bits] "
                 "(PER_CLUSTER)");
    CUVS_EXPECTS(ro.ndim == 2 && ro.shape[0] == idx->rot_dim && ro.shape[1] == idx->dim,
                 "rotation_matrix must have shape [rot_dim, dim]");
    CUVS_EXPECTS(ce.ndim == 2 && ce.shape[0] == idx->n_lists && (ce.shape[1] == idx->dim || ce.shape[1] == idx->dim_ext),
                 "centers must have shape [n_lists, dim] or [n_lists, dim_ext]");
    copy_async(res, idx->pq_centers.data(), dl_data(pc), idx->pq_centers.bytes());
    copy_async(res, idx->rotation.data(), dl_data(ro), idx->rotation.bytes());
    dev_buf<float> flat(res, (size_t)idx->n_lists * idx->dim);
    HIP_TRY(hipMemcpy2DAsync(flat.data(), idx->dim * sizeof(float), dl_data(ce), ce.shape[1] * sizeof(float),
                             idx->dim * sizeof(float), idx->n_lists, hipMemcpyDefault, res.stream));
    ivf_pq_set_centers(res, *idx, flat.data());
    if (centers_rot != nullptr) {
      auto& cr = centers_rot->dl_tensor;
      CUVS_EXPECTS(dtype_is(cr.dtype, kDLFloat, 32) && cr.ndim == 2 && cr.shape[0] == idx->n_lists &&
                     cr.shape[1] == idx->rot_dim,
                   "centers_rot must have shape [n_lists, rot_dim]");
      copy_async(res, idx->centers_rot.data(), dl_data(cr), idx->centers_rot.bytes());
    }
    sync(res);
    delete reinterpret_cast<ivf_pq_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = kF32;
  });
}

// largest source id stored in the lists (cached per index state): the bitset of a filtered search is indexed by source id
__global__ void max_source_id_kernel(const int64_t* __restrict__ ids, const uint32_t* __restrict__ list_offsets,
                                     const uint32_t* __restrict__ list_sizes, uint32_t n_lists, unsigned long long* __restrict__ out)
{
  const uint32_t L = blockIdx.x;
  if (L >= n_lists) return;
  long long m = -1;
  for (uint32_t r = threadIdx.x; r < list_sizes[L]; r += blockDim.x) m = max(m, (long long)ids[list_offsets[L] + r]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m >= 0) atomicMax(out, (unsigned long long)m);
}
static int64_t ivf_pq_max_source_id(resources& res, const ivf_pq_index& idx)
{
  if (idx.max_id_ptr == idx.indices.data() && idx.max_id_rows == idx.size) return idx.max_id;
  int64_t m = -1;
  if (idx.size > 0) {
    dev_buf<unsigned long long> d(res, 1);
    HIP_TRY(hipMemsetAsync(d.data(), 0, sizeof(unsigned long long), res.stream));
    hipLaunchKernelGGL(max_source_id_kernel, dim3(idx.n_lists), dim3(256), 0, res.stream, idx.indices.data(), idx.list_offsets.data(),
                       idx.list_sizes.data(), idx.n_lists, d.data());
    m = (int64_t)to_host(res, d.data(), 1)[0];
  }
  idx.max_id = m; idx.max_id_rows = idx.size; idx.max_id_ptr = idx.indices.data();
  return m;
}

static cuvsError_t pq_search_entry(cuvsResources_t res_h, cuvsIvfPqSearchParams_t params, cuvsIvfPqIndex_t index_c,
                                   DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                                   DLManagedTensor* distances_tensor, cuvsFilter filter)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index_c);
    CUVS_EXPECTS(params && queries_tensor && neighbors_tensor && distances_tensor, "null argument");
    // ivf_pq_search.cuh:914-916
    CUVS_EXPECTS(idx.codes_layout == 1, "IVF-PQ search requires INTERLEAVED codes layout. FLAT layout is not supported for GPU search.");
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
    CUVS_EXPECTS(index_c->dtype.bits == 0 ||
                   (queries.dtype.code == index_c->dtype.code && queries.dtype.bits == index_c->dtype.bits),
                 "Unsupported queries DLtensor dtype: %d and bits: %d", (int)queries.dtype.code,
                 (int)queries.dtype.bits);
    CUVS_EXPECTS(queries.shape[1] == idx.dim, "queries dim %ld != index dim %u", (long)queries.shape[1], idx.dim);
    int64_t m = queries.shape[0], k = neighbors.shape[1];
    CUVS_EXPECTS(neighbors.shape[0] == m && distances.shape[0] == m && distances.shape[1] == k,
                 "neighbors/distances shape mismatch");
    const uint32_t* filter_bits = nullptr;
    if (filter.type != NO_FILTER) {  // same contract as cuvsIvfFlatSearch's prefilter (c/src/neighbors/ivf_flat.cpp)
      CUVS_EXPECTS(filter.type == BITSET && filter.addr != 0, "ivf_pq search: only BITSET filters are supported");
      auto& ft = reinterpret_cast<DLManagedTensor*>(filter.addr)->dl_tensor;
      CUVS_EXPECTS(dtype_is(ft.dtype, kDLUInt, 32) && is_device_accessible(ft), "filter must be a device uint32 tensor");
      int64_t words = 1;
      for (int i = 0; i < ft.ndim; ++i) words *= ft.shape[i];
      CUVS_EXPECTS(words * 32 >= idx.size, "bitset filter holds %ld bits, the index %ld rows", (long)(words * 32), (long)idx.size);
      // the kernels index the bitset by SOURCE id (ids given to extend() need not be 0 .. size - 1)
      const int64_t max_id = ivf_pq_max_source_id(res, idx);
      CUVS_EXPECTS(words * 32 > max_id, "bitset filter holds %ld bits, the largest source id of the index is %ld", (long)(words * 32),
                   (long)max_id);
      filter_bits = static_cast<const uint32_t*>(dl_data(ft));
    }
    ivf_pq_search_params sp;
    sp.n_probes                = params->n_probes;
    sp.lut_dtype               = (int)params->lut_dtype;
    sp.internal_distance_dtype = (int)params->internal_distance_dtype;
    sp.max_internal_batch_size = params->max_internal_batch_size;
    sp.coarse_search_dtype     = (int)params->coarse_search_dtype;
    ivf_pq_search(res, sp, idx, dl_data(queries), elem_of(queries.dtype), m, (int)k,
                  static_cast<int64_t*>(dl_data(neighbors)), static_cast<float*>(dl_data(distances)), filter_bits);
  });
}

cuvsError_t cuvsIvfPqSearch(cuvsResources_t res_h, cuvsIvfPqSearchParams_t params, cuvsIvfPqIndex_t index_c,
                            DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                            DLManagedTensor* distances_tensor)
{
  return pq_search_entry(res_h, params, index_c, queries_tensor, neighbors_tensor, distances_tensor, cuvsFilter{0, NO_FILTER});
}

// The C++ search overload with a sample filter (cpp/include/cuvs/neighbors/ivf_pq.hpp:1818-1828: bitset_filter over
// source ids) has no C entry point in the reference (c/include/cuvs/neighbors/ivf_pq.h:536-541 takes no filter); this is
// that overload with the argument convention of cuvsIvfFlatSearch (include/cuvs_amd/extensions.h).
cuvsError_t cuvsAmdIvfPqSearchFiltered(cuvsResources_t res_h, cuvsIvfPqSearchParams_t params, cuvsIvfPqIndex_t index_c,
                                       DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                                       DLManagedTensor* distances_tensor, cuvsFilter filter)
{
  return pq_search_entry(res_h, params, index_c, queries_tensor, neighbors_tensor, distances_tensor, filter);
}

// extension: index.codes_layout() of the C++ API (cpp/include/cuvs/neighbors/ivf_pq.hpp: list_layout) - 0 FLAT, 1 INTERLEAVED
cuvsError_t cuvsAmdIvfPqIndexGetCodesLayout(cuvsIvfPqIndex_t index_c, int* layout)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(layout != nullptr, "null argument");
    *layout = get_index(index_c).codes_layout;
  });
}

cuvsError_t cuvsIvfPqExtend(cuvsResources_t res_h, DLManagedTensor* new_vectors, DLManagedTensor* new_indices,
                            cuvsIvfPqIndex_t index_c)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index_c);
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
    // an untyped handle (reference-format load) takes the dtype of the first vectors (c/src/neighbors/ivf_pq.cpp:412-416)
    if (index_c->dtype.code == 0 && index_c->dtype.bits == 0) index_c->dtype = v.dtype;
    ivf_pq_extend(res, idx, dl_data(v), elem_of(v.dtype), v.shape[0], !is_device_accessible(v), ids, ids_host);
  });
}

#define GETTER(NAME, EXPR)                                                              \
  cuvsError_t NAME(cuvsIvfPqIndex_t index, int64_t* out)                                \
  {                                                                                     \
    return (cuvsError_t)translate_exceptions([=] {                                      \
      auto& idx = get_index(index);                                                     \
      CUVS_EXPECTS(out != nullptr, "null output");                                      \
      *out = (int64_t)(EXPR);                                                           \
    });                                                                                 \
  }
GETTER(cuvsIvfPqIndexGetNLists, idx.n_lists)
GETTER(cuvsIvfPqIndexGetDim, idx.dim)
GETTER(cuvsIvfPqIndexGetSize, idx.size)
GETTER(cuvsIvfPqIndexGetPqDim, idx.pq_dim)
GETTER(cuvsIvfPqIndexGetPqBits, idx.pq_bits)
GETTER(cuvsIvfPqIndexGetPqLen, idx.pq_len)
#undef GETTER

cuvsError_t cuvsIvfPqIndexGetCenters(cuvsIvfPqIndex_t index, DLManagedTensor* centers)
{
  // strided view of the first `dim` columns of the padded centers (c/src/neighbors/ivf_pq.cpp getters)
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    fill_dl_view(centers, idx.centers.data(), kF32, idx.n_lists, idx.dim, 2, 0);
    centers->dl_tensor.strides = new int64_t[2]{(int64_t)idx.dim_ext, 1};
    centers->deleter           = [](DLManagedTensor* self) {
      delete[] self->dl_tensor.shape;
      delete[] self->dl_tensor.strides;
    };
  });
}
cuvsError_t cuvsIvfPqIndexGetCentersPadded(cuvsIvfPqIndex_t index, DLManagedTensor* centers)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    fill_dl_view(centers, idx.centers.data(), kF32, idx.n_lists, idx.dim_ext, 2, 0);
  });
}
cuvsError_t cuvsIvfPqIndexGetPqCenters(cuvsIvfPqIndex_t index, DLManagedTensor* pq_centers)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    fill_dl_view(pq_centers, idx.pq_centers.data(), kF32, idx.pq_dim, idx.pq_len, 3, 0);
    delete[] pq_centers->dl_tensor.shape;
    pq_centers->dl_tensor.shape = new int64_t[3]{(int64_t)(idx.codebook_kind == 1 ? idx.n_lists : idx.pq_dim),
                                                 (int64_t)idx.pq_len, (int64_t)idx.pq_book};
  });
}
cuvsError_t cuvsIvfPqIndexGetCentersRot(cuvsIvfPqIndex_t index, DLManagedTensor* centers_rot)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    fill_dl_view(centers_rot, idx.centers_rot.data(), kF32, idx.n_lists, idx.rot_dim, 2, 0);
  });
}
cuvsError_t cuvsIvfPqIndexGetRotationMatrix(cuvsIvfPqIndex_t index, DLManagedTensor* rotation_matrix)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    fill_dl_view(rotation_matrix, idx.rotation.data(), kF32, idx.rot_dim, idx.dim, 2, 0);
  });
}
cuvsError_t cuvsIvfPqIndexGetListSizes(cuvsIvfPqIndex_t index, DLManagedTensor* list_sizes)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    fill_dl_view(list_sizes, idx.list_sizes.data(), DLDataType{kDLUInt, 32, 1}, idx.n_lists, 1, 1, 0);
  });
}

cuvsError_t cuvsIvfPqIndexUnpackContiguousListData(cuvsResources_t res_h, cuvsIvfPqIndex_t index,
                                                   DLManagedTensor* out_codes, uint32_t label, uint32_t offset)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index);
    auto& t   = out_codes->dl_tensor;
    CUVS_EXPECTS(is_device_accessible(t) && dtype_is(t.dtype, kDLUInt, 8) && t.ndim == 2 && is_c_contiguous(t),
                 "out_codes must be a device uint8 row-major matrix");
    uint32_t bpr = (idx.pq_dim * idx.pq_bits + 7) / 8;
    CUVS_EXPECTS(t.shape[1] == bpr, "out_codes must have %u columns", bpr);
    ivf_pq_unpack_list(res, idx, label, offset, (uint32_t)t.shape[0], static_cast<uint8_t*>(dl_data(t)));
  });
}

cuvsError_t cuvsIvfPqIndexGetListIndices(cuvsIvfPqIndex_t index, uint32_t label, DLManagedTensor* out_labels)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& idx = get_index(index);
    CUVS_EXPECTS(label < idx.n_lists, "Expected label to be less than number of lists in the index");
    fill_dl_view(out_labels, idx.indices.data() + idx.h_list_offsets[label], DLDataType{kDLInt, 64, 1},
                 idx.h_list_sizes[label], 1, 1, 0);
  });
}

}  // extern "C"

namespace {
constexpr int kPqRefVersion = 4;  // ivf_pq_serialize.cuh:28

void pq_write_native(resources& res, const char* filename, const ivf_pq_index& idx, DLDataType dl)
{
  file_writer w(filename, KIND_IVF_PQ);
  // (codes_layout rides in bit 8 of the codebook_kind word: files written before round 5 read as INTERLEAVED)
  w.scalar<int32_t>(idx.metric); w.scalar<int32_t>(idx.codebook_kind | (idx.codes_layout == 0 ? 0x100 : 0)); w.scalar<int32_t>((int)idx.dtype);
  const uint32_t u[] = {idx.n_lists, idx.dim, idx.dim_ext, idx.rot_dim, idx.pq_dim, idx.pq_bits, idx.pq_len,
                        idx.pq_book, idx.n_chunks, idx.codes_per_chunk};
  for (uint32_t v : u) w.scalar<uint32_t>(v);
  w.scalar<int64_t>(idx.size); w.scalar<int64_t>(idx.padded_rows);
  w.scalar<uint8_t>(dl.code); w.scalar<uint8_t>(dl.bits);
  w.device_array(res, idx.centers.data(), idx.centers.bytes());
  w.device_array(res, idx.center_norms.data(), idx.center_norms.bytes());
  w.device_array(res, idx.centers_rot.data(), idx.centers_rot.bytes());
  w.device_array(res, idx.rotation.data(), idx.rotation.bytes());
  w.device_array(res, idx.pq_centers.data(), idx.pq_centers.bytes());
  w.device_array(res, idx.list_sizes.data(), idx.list_sizes.bytes());
  w.device_array(res, idx.list_offsets.data(), idx.list_offsets.bytes());
  w.device_array(res, idx.codes.data(), idx.codes.bytes());
  w.device_array(res, idx.indices.data(), idx.indices.bytes());
}

std::unique_ptr<ivf_pq_index> pq_read_native(resources& res, const char* filename, DLDataType* dl)
{
  file_reader r(filename, KIND_IVF_PQ);
  auto idx = std::make_unique<ivf_pq_index>();
  idx->metric = r.scalar<int32_t>(); idx->codebook_kind = r.scalar<int32_t>(); idx->dtype = (elem_t)r.scalar<int32_t>();
  idx->codes_layout = (idx->codebook_kind & 0x100) ? 0 : 1;
  idx->codebook_kind &= 0xff;
  uint32_t* u[] = {&idx->n_lists, &idx->dim, &idx->dim_ext, &idx->rot_dim, &idx->pq_dim, &idx->pq_bits, &idx->pq_len,
                   &idx->pq_book, &idx->n_chunks, &idx->codes_per_chunk};
  for (uint32_t* v : u) *v = r.scalar<uint32_t>();
  idx->size = r.scalar<int64_t>(); idx->padded_rows = r.scalar<int64_t>();
  uint8_t code = r.scalar<uint8_t>(), bits = r.scalar<uint8_t>();
  idx->centers      = r.device_array<float>(res);
  idx->center_norms = r.device_array<float>(res);
  idx->centers_rot  = r.device_array<float>(res);
  idx->rotation     = r.device_array<float>(res);
  idx->pq_centers   = r.device_array<float>(res);
  idx->list_sizes   = r.device_array<uint32_t>(res);
  idx->list_offsets = r.device_array<uint32_t>(res);
  idx->codes        = r.device_array<uint8_t>(res);
  idx->indices      = r.device_array<int64_t>(res);
  idx->h_list_sizes   = to_host(res, idx->list_sizes.data(), idx->n_lists);
  idx->h_list_offsets = to_host(res, idx->list_offsets.data(), idx->n_lists + 1);
  *dl = DLDataType{code, bits, 1};
  return idx;
}

// Reference record sequence (ivf_pq_serialize.cuh:49-85, ivf_list.cuh:108-131): version, size, dim, pq_bits,
// pq_dim, conservative_memory_allocation, metric, codebook_kind, codes_layout, n_lists, pq_centers, centers
// [n_lists, dim_ext], centers_rot, rotation_matrix, list_sizes, then per list: size [, codes
// [ceil(size/32), n_chunks, 32, 16], ids [size]]. A 16-byte chunk has the same content in both libraries
// (ivf_pq_codepacking.cuh:106-137); only the row grouping differs (32 there, 64 here), so lists are regrouped
// chunk by chunk on the host. The file carries no dataset dtype (the reference's index is untyped).
void pq_write_ref(resources& res, const char* filename, const ivf_pq_index& idx)
{
  npy_writer w(filename);
  w.scalar<int32_t>(kPqRefVersion);
  w.scalar<int64_t>(idx.size);
  w.scalar<uint32_t>(idx.dim);
  w.scalar<uint32_t>(idx.pq_bits);
  w.scalar<uint32_t>(idx.pq_dim);
  w.scalar<bool>(true);  // conservative_memory_allocation
  w.scalar<int32_t>(idx.metric);
  w.scalar<int32_t>(idx.codebook_kind);
  w.scalar<int32_t>(idx.codes_layout);  // list_layout (ivf_pq.hpp:40-45): 0 FLAT, 1 INTERLEAVED
  w.scalar<uint32_t>(idx.n_lists);
  const int64_t pqc0 = idx.codebook_kind == 0 ? idx.pq_dim : idx.n_lists;
  w.device_array(res, 'f', 4, {pqc0, idx.pq_len, idx.pq_book}, idx.pq_centers.data());
  w.device_array(res, 'f', 4, {idx.n_lists, idx.dim_ext}, idx.centers.data());
  w.device_array(res, 'f', 4, {idx.n_lists, idx.rot_dim}, idx.centers_rot.data());
  w.device_array(res, 'f', 4, {idx.rot_dim, idx.dim}, idx.rotation.data());
  w.host_array<uint32_t>(idx.h_list_sizes.data(), {idx.n_lists});
  std::vector<uint8_t> ours, theirs;
  std::vector<int64_t> ids;
  const uint32_t nc = idx.n_chunks;
  for (uint32_t L = 0; L < idx.n_lists; ++L) {
    const uint32_t size = idx.h_list_sizes[L];
    w.scalar<uint32_t>(size);
    if (size == 0) continue;
    const uint32_t cap = idx.h_list_offsets[L + 1] - idx.h_list_offsets[L], g32 = (size + 31) / 32;
    ours.resize((size_t)cap * nc * 16);
    ids.resize(size);
    copy_async(res, ours.data(), idx.codes.data() + (size_t)idx.h_list_offsets[L] * nc * 16, ours.size());
    copy_async(res, ids.data(), idx.indices.data() + idx.h_list_offsets[L], (size_t)size * sizeof(int64_t));
    sync(res);
    if (idx.codes_layout == 0) {
      // FLAT list record [size, bytes_per_vector] (list_spec_flat, ivf_pq.hpp:302-338): a row's codes as one contiguous
      // little-endian bitstream; here they sit in 16-byte chunks of codes_per_chunk codes, each chunk starting at bit 0
      const uint32_t bits = idx.pq_bits, cpc = idx.codes_per_chunk, bpv = (idx.pq_dim * bits + 7) / 8;
      theirs.assign((size_t)size * bpv, 0);
      for (uint32_t r = 0; r < size; ++r) {
        uint8_t* dst = theirs.data() + (size_t)r * bpv;
        for (uint32_t j = 0; j < idx.pq_dim; ++j) {
          const uint8_t* src = ours.data() + (((size_t)(r / 64) * nc + j / cpc) * 64 + r % 64) * 16;
          uint32_t ib = (j % cpc) * bits, ob = j * bits;
          for (uint32_t b = 0; b < bits; ++b, ++ib, ++ob) dst[ob >> 3] |= (uint8_t)(((src[ib >> 3] >> (ib & 7)) & 1u) << (ob & 7));
        }
      }
      w.header('u', 1, {size, bpv});
      w.raw(theirs.data(), theirs.size());
    } else {
    theirs.assign((size_t)g32 * nc * 32 * 16, 0);
    for (uint32_t r = 0; r < size; ++r)
      for (uint32_t c = 0; c < nc; ++c)
        memcpy(theirs.data() + (((size_t)(r / 32) * nc + c) * 32 + r % 32) * 16,
               ours.data() + (((size_t)(r / 64) * nc + c) * 64 + r % 64) * 16, 16);
    w.header('u', 1, {g32, nc, 32, 16});
    w.raw(theirs.data(), theirs.size());
    }
    w.host_array<int64_t>(ids.data(), {size});
  }
  w.close();
}

std::unique_ptr<ivf_pq_index> pq_read_ref(resources& res, const char* filename)
{
  npy_reader r(filename);
  int ver = r.scalar<int32_t>();
  CUVS_EXPECTS(ver == kPqRefVersion, "serialization version mismatch %d vs. %d", ver, kPqRefVersion);
  ivf_pq_build_params p;
  int64_t n_rows  = r.scalar<int64_t>();
  uint32_t dim    = r.scalar<uint32_t>();
  p.pq_bits       = r.scalar<uint32_t>();
  p.pq_dim        = r.scalar<uint32_t>();
  (void)r.scalar<bool>();  // conservative_memory_allocation
  p.metric        = r.scalar<int32_t>();
  p.codebook_kind = r.scalar<int32_t>();
  int layout      = r.scalar<int32_t>();
  p.n_lists       = r.scalar<uint32_t>();
  CUVS_EXPECTS(layout == 0 || layout == 1, "ivf_pq::deserialize: invalid list_layout value %d", layout);
  p.codes_layout  = layout;
  CUVS_EXPECTS(p.codebook_kind == 0 || p.codebook_kind == 1, "ivf_pq::deserialize: invalid codebook_gen value %d",
               p.codebook_kind);
  CUVS_EXPECTS(dim > 0 && p.pq_dim > 0 && p.n_lists > 0 && p.n_lists <= (1u << 24), "ivf_pq::deserialize: bad header");
  auto idx = ivf_pq_make_empty(res, p, elem_t::f32, dim);
  idx->dtype_known = false;
  const int64_t pqc0 = idx->codebook_kind == 0 ? idx->pq_dim : idx->n_lists;
  idx->pq_centers  = r.device_array<float>(res, pqc0 * idx->pq_len * idx->pq_book);
  std::vector<float> h_centers = r.host_array<float>((int64_t)idx->n_lists * idx->dim_ext);
  idx->centers     = dev_buf<float>::persistent(h_centers.size());
  copy_async(res, idx->centers.data(), h_centers.data(), h_centers.size() * sizeof(float));
  std::vector<float> h_norms(idx->n_lists);
  for (uint32_t L = 0; L < idx->n_lists; ++L) h_norms[L] = h_centers[(size_t)L * idx->dim_ext + idx->dim];
  idx->center_norms = dev_buf<float>::persistent(idx->n_lists);
  copy_async(res, idx->center_norms.data(), h_norms.data(), h_norms.size() * sizeof(float));
  sync(res);
  idx->centers_rot = r.device_array<float>(res, (int64_t)idx->n_lists * idx->rot_dim);
  idx->rotation    = r.device_array<float>(res, (int64_t)idx->rot_dim * idx->dim);
  idx->h_list_sizes = r.host_array<uint32_t>(idx->n_lists);
  int64_t total = 0, live = 0;
  for (uint32_t L = 0; L < idx->n_lists; ++L) {
    idx->h_list_offsets[L] = (uint32_t)total;
    total += round_up(idx->h_list_sizes[L], 64);
    live += idx->h_list_sizes[L];
  }
  CUVS_EXPECTS(total < (int64_t(1) << 32), "ivf_pq::deserialize: index too large");
  CUVS_EXPECTS(live == n_rows, "ivf_pq::deserialize: list sizes (%ld) do not add up to the index size (%ld)",
               (long)live, (long)n_rows);
  idx->h_list_offsets[idx->n_lists] = (uint32_t)total;
  idx->size = n_rows; idx->padded_rows = total;
  copy_async(res, idx->list_sizes.data(), idx->h_list_sizes.data(), idx->n_lists * sizeof(uint32_t));
  copy_async(res, idx->list_offsets.data(), idx->h_list_offsets.data(), (idx->n_lists + 1) * sizeof(uint32_t));
  const uint32_t nc = idx->n_chunks, cpc = idx->codes_per_chunk, bits = idx->pq_bits;
  idx->codes   = dev_buf<uint8_t>::persistent((size_t)total * nc * 16);
  idx->indices = dev_buf<int64_t>::persistent((size_t)total);
  HIP_TRY(hipMemsetAsync(idx->codes.data(), 0, idx->codes.bytes(), res.stream));
  HIP_TRY(hipMemsetAsync(idx->indices.data(), 0xff, idx->indices.bytes(), res.stream));
  sync(res);
  const uint32_t bpv = (idx->pq_dim * bits + 7) / 8;
  std::vector<uint8_t> ours;
  std::vector<char> theirs;
  for (uint32_t L = 0; L < idx->n_lists; ++L) {
    const uint32_t size = r.scalar<uint32_t>();
    CUVS_EXPECTS(size == idx->h_list_sizes[L], "ivf_pq::deserialize: list %u holds %u rows, list_sizes says %u", L,
                 size, idx->h_list_sizes[L]);
    if (size == 0) continue;
    const uint32_t g32 = (size + 31) / 32, cap = (uint32_t)round_up(size, 64);
    r.array(1, layout == 1 ? (int64_t)g32 * nc * 32 * 16 : (int64_t)size * bpv, theirs);
    std::vector<int64_t> ids = r.host_array<int64_t>(size);
    ours.assign((size_t)cap * nc * 16, 0);
    if (layout == 1) {
      for (uint32_t rr = 0; rr < size; ++rr)
        for (uint32_t c = 0; c < nc; ++c)
          memcpy(ours.data() + (((size_t)(rr / 64) * nc + c) * 64 + rr % 64) * 16,
                 theirs.data() + (((size_t)(rr / 32) * nc + c) * 32 + rr % 32) * 16, 16);
    } else {
      // FLAT layout: one contiguous little-endian bitstream per row (ivf_pq.hpp:302-338) -> 16-byte chunks of
      // codes_per_chunk codes, each chunk starting at bit 0
      for (uint32_t rr = 0; rr < size; ++rr) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(theirs.data()) + (size_t)rr * bpv;
        for (uint32_t j = 0; j < idx->pq_dim; ++j) {
          uint32_t ib = j * bits, code = 0;
          for (uint32_t b = 0; b < bits; ++b, ++ib) code |= (uint32_t)((src[ib >> 3] >> (ib & 7)) & 1u) << b;
          uint8_t* dst = ours.data() + (((size_t)(rr / 64) * nc + j / cpc) * 64 + rr % 64) * 16;
          uint32_t ob  = (j % cpc) * bits;
          for (uint32_t b = 0; b < bits; ++b, ++ob) dst[ob >> 3] |= (uint8_t)(((code >> b) & 1u) << (ob & 7));
        }
      }
    }
    copy_async(res, idx->codes.data() + (size_t)idx->h_list_offsets[L] * nc * 16, ours.data(), ours.size());
    copy_async(res, idx->indices.data() + idx->h_list_offsets[L], ids.data(), (size_t)size * sizeof(int64_t));
    sync(res);
  }
  return idx;
}
}  // namespace

extern "C" {
cuvsError_t cuvsIvfPqSerialize(cuvsResources_t res_h, const char* filename, cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index);
    if (write_native_container(res)) pq_write_native(res, filename, idx, index->dtype);
    else pq_write_ref(res, filename, idx);
  });
}
cuvsError_t cuvsIvfPqDeserialize(cuvsResources_t res_h, const char* filename, cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr, "index is null");
    // a reference-format file carries no dataset dtype: the handle's dtype stays 0/0 until the first search or
    // extend names one (c/src/neighbors/ivf_pq.cpp:389-395,412-416)
    DLDataType dl{0, 0, 1};
    auto idx = is_native_container(filename) ? pq_read_native(res, filename, &dl) : pq_read_ref(res, filename);
    delete reinterpret_cast<ivf_pq_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = dl;
  });
}
cuvsError_t cuvsIvfPqTransform(cuvsResources_t res_h, cuvsIvfPqIndex_t index, DLManagedTensor* input_dataset,
                               DLManagedTensor* output_labels, DLManagedTensor* output_dataset)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index);
    CUVS_EXPECTS(input_dataset && output_labels && output_dataset, "null argument");
    auto& in = input_dataset->dl_tensor;
    auto& ol = output_labels->dl_tensor;
    auto& oc = output_dataset->dl_tensor;
    CUVS_EXPECTS(is_device_accessible(in), "input_dataset should have device compatible memory");
    CUVS_EXPECTS(is_device_accessible(ol), "output_labels should have device compatible memory");
    CUVS_EXPECTS(is_device_accessible(oc), "output_dataset should have device compatible memory");
    CUVS_EXPECTS(dtype_is(ol.dtype, kDLUInt, 32), "output_labels must have a uint32 dtype ");
    CUVS_EXPECTS(dtype_is(oc.dtype, kDLUInt, 8), "output_dataset must have a uint8 dtype");
    CUVS_EXPECTS(in.ndim == 2 && is_c_contiguous(in) && in.shape[1] == idx.dim, "input_dataset must be [n, dim] row-major");
    const uint32_t bpr = (idx.pq_dim * idx.pq_bits + 7) / 8;
    CUVS_EXPECTS(ol.shape[0] == in.shape[0] && oc.ndim == 2 && oc.shape[0] == in.shape[0] && oc.shape[1] == bpr &&
                   is_c_contiguous(oc),
                 "output_labels must be [n] and output_dataset [n, %u]", bpr);
    ivf_pq_transform(res, idx, dl_data(in), elem_of(in.dtype), in.shape[0], static_cast<uint32_t*>(dl_data(ol)),
                     static_cast<uint8_t*>(dl_data(oc)));
  });
}

}  // extern "C"
