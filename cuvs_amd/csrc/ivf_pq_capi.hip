// cuvsIvfPq* C entry points (drop-in for c/src/neighbors/ivf_pq.cpp) over ivf_pq_build.hip / ivf_pq_search.hip.
#include "ivf_pq.hpp"
#include "ops.hpp"
#include "serialize.hpp"

#include <cuvs/neighbors/ivf_pq.h>

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
  CUVS_EXPECTS(p.codes_layout == CUVS_IVF_PQ_LIST_LAYOUT_INTERLEAVED,
               "IVF-PQ search requires INTERLEAVED codes layout. FLAT layout is not supported for GPU search.");
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
    CUVS_EXPECTS(pc.ndim == 3 && pc.shape[0] == idx->pq_dim && pc.shape[1] == idx->pq_len &&
                   pc.shape[2] == idx->pq_book,
                 "pq_centers must have shape [pq_dim, pq_len, 2^pq_bits]");
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

cuvsError_t cuvsIvfPqSearch(cuvsResources_t res_h, cuvsIvfPqSearchParams_t params, cuvsIvfPqIndex_t index_c,
                            DLManagedTensor* queries_tensor, DLManagedTensor* neighbors_tensor,
                            DLManagedTensor* distances_tensor)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index_c);
    CUVS_EXPECTS(params && queries_tensor && neighbors_tensor && distances_tensor, "null argument");
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
    ivf_pq_search_params sp;
    sp.n_probes                = params->n_probes;
    sp.lut_dtype               = (int)params->lut_dtype;
    sp.internal_distance_dtype = (int)params->internal_distance_dtype;
    sp.max_internal_batch_size = params->max_internal_batch_size;
    ivf_pq_search(res, sp, idx, dl_data(queries), elem_of(queries.dtype), m, (int)k,
                  static_cast<int64_t*>(dl_data(neighbors)), static_cast<float*>(dl_data(distances)));
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
    pq_centers->dl_tensor.shape = new int64_t[3]{(int64_t)idx.pq_dim, (int64_t)idx.pq_len, (int64_t)idx.pq_book};
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

cuvsError_t cuvsIvfPqSerialize(cuvsResources_t res_h, const char* filename, cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    auto& idx = get_index(index);
    file_writer w(filename, KIND_IVF_PQ);
    w.scalar<int32_t>(idx.metric); w.scalar<int32_t>(idx.codebook_kind); w.scalar<int32_t>((int)idx.dtype);
    const uint32_t u[] = {idx.n_lists, idx.dim, idx.dim_ext, idx.rot_dim, idx.pq_dim, idx.pq_bits, idx.pq_len,
                          idx.pq_book, idx.n_chunks, idx.codes_per_chunk};
    for (uint32_t v : u) w.scalar<uint32_t>(v);
    w.scalar<int64_t>(idx.size); w.scalar<int64_t>(idx.padded_rows);
    w.scalar<uint8_t>(index->dtype.code); w.scalar<uint8_t>(index->dtype.bits);
    w.device_array(res, idx.centers.data(), idx.centers.bytes());
    w.device_array(res, idx.center_norms.data(), idx.center_norms.bytes());
    w.device_array(res, idx.centers_rot.data(), idx.centers_rot.bytes());
    w.device_array(res, idx.rotation.data(), idx.rotation.bytes());
    w.device_array(res, idx.pq_centers.data(), idx.pq_centers.bytes());
    w.device_array(res, idx.list_sizes.data(), idx.list_sizes.bytes());
    w.device_array(res, idx.list_offsets.data(), idx.list_offsets.bytes());
    w.device_array(res, idx.codes.data(), idx.codes.bytes());
    w.device_array(res, idx.indices.data(), idx.indices.bytes());
  });
}
cuvsError_t cuvsIvfPqDeserialize(cuvsResources_t res_h, const char* filename, cuvsIvfPqIndex_t index)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr, "index is null");
    file_reader r(filename, KIND_IVF_PQ);
    auto idx = std::make_unique<ivf_pq_index>();
    idx->metric = r.scalar<int32_t>(); idx->codebook_kind = r.scalar<int32_t>(); idx->dtype = (elem_t)r.scalar<int32_t>();
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
    delete reinterpret_cast<ivf_pq_index*>(index->addr);
    index->addr  = reinterpret_cast<uintptr_t>(idx.release());
    index->dtype = DLDataType{code, bits, 1};
  });
}
cuvsError_t cuvsIvfPqTransform(cuvsResources_t, cuvsIvfPqIndex_t, DLManagedTensor*, DLManagedTensor*,
                               DLManagedTensor*)
{
  return (cuvsError_t)translate_exceptions(
    [=] { CUVS_FAIL("cuvsIvfPqTransform is outside the search hot path and not built (SURVEY 8b)"); });
}

}  // extern "C"
