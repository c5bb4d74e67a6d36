// C++ surface over the C ABI: cuvs::neighbors::{brute_force, ivf_flat, ivf_pq, cagra}::build / search with the
// reference's names (cpp/include/cuvs/neighbors/{brute_force,ivf_flat,ivf_pq,cagra}.hpp). Header-only, C++17, no RAFT:
// `raft::resources` -> cuvs::resources (RAII over cuvsResources_t), `raft::device_matrix_view<T, int64_t>` ->
// cuvs::device_matrix_view<T> (data_handle(), extent(i) - the two members the reference call sites use). Parameter
// structs carry the reference's C++ field names and defaults (ivf_pq.hpp:40-236, ivf_flat.hpp:29-103,
// cagra.hpp:84-360) and are copied field by field into the C structs. Errors become cuvs::error exceptions carrying
// cuvsGetLastErrorText(), like the reference's raft::exception.
#pragma once
#include <cuvs/core/all.h>
#include <cuvs_amd/extensions.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace cuvs {

struct error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline void check(cuvsError_t e, const char* what)
{
  if (e != CUVS_SUCCESS) {
    const char* t = cuvsGetLastErrorText();
    throw error(std::string(what) + ": " + (t ? t : "unknown error"));
  }
}

class resources {
 public:
  resources() { check(cuvsResourcesCreate(&h_), "cuvsResourcesCreate"); }
  ~resources() { cuvsResourcesDestroy(h_); }
  resources(const resources&)            = delete;
  resources& operator=(const resources&) = delete;
  cuvsResources_t get() const { return h_; }
  void sync_stream() const { check(cuvsStreamSync(h_), "cuvsStreamSync"); }

 private:
  cuvsResources_t h_ = 0;
};

// row-major [rows, cols] view of memory the caller owns (device memory unless `on_host`)
template <typename T>
struct device_matrix_view {
  T* ptr         = nullptr;
  int64_t rows   = 0, cols = 0;
  bool on_host   = false;
  T* data_handle() const { return ptr; }
  int64_t extent(int i) const { return i == 0 ? rows : cols; }
};
template <typename T>
device_matrix_view<T> make_device_matrix_view(T* p, int64_t rows, int64_t cols) { return {p, rows, cols, false}; }
template <typename T>
device_matrix_view<T> make_host_matrix_view(T* p, int64_t rows, int64_t cols) { return {p, rows, cols, true}; }

namespace detail {
template <typename T>
DLDataType dl_dtype()
{
  using U = std::remove_cv_t<T>;
  if constexpr (std::is_same_v<U, float>) return {kDLFloat, 32, 1};
  else if constexpr (std::is_same_v<U, int8_t>) return {kDLInt, 8, 1};
  else if constexpr (std::is_same_v<U, uint8_t>) return {kDLUInt, 8, 1};
  else if constexpr (std::is_same_v<U, int64_t>) return {kDLInt, 64, 1};
  else if constexpr (std::is_same_v<U, uint32_t>) return {kDLUInt, 32, 1};
  else if constexpr (sizeof(U) == 2) return {kDLFloat, 16, 1};  // __half / _Float16
  else static_assert(sizeof(U) == 0, "unsupported element type");
}
// a DLManagedTensor describing a view; lives as long as the call it is passed to
template <typename T>
struct tensor {
  DLManagedTensor m{};
  int64_t shape[2];
  explicit tensor(const device_matrix_view<T>& v)
  {
    shape[0] = v.rows; shape[1] = v.cols;
    m.dl_tensor.data        = const_cast<std::remove_cv_t<T>*>(v.ptr);
    m.dl_tensor.device      = {v.on_host ? kDLCPU : kDLCUDA, 0};
    m.dl_tensor.ndim        = 2;
    m.dl_tensor.dtype       = dl_dtype<T>();
    m.dl_tensor.shape       = shape;
    m.dl_tensor.strides     = nullptr;
    m.dl_tensor.byte_offset = 0;
  }
  DLManagedTensor* get() { return &m; }
};
template <typename P, cuvsError_t (*Create)(P*), cuvsError_t (*Destroy)(P)>
struct c_params {
  P p = nullptr;
  c_params() { check(Create(&p), "params create"); }
  ~c_params() { Destroy(p); }
  c_params(const c_params&) = delete;
};
}  // namespace detail

namespace neighbors {

namespace filtering {
// cuvs::neighbors::filtering::bitset_filter (cpp/include/cuvs/neighbors/common.hpp): bit i of the device words = 1 keeps
// source row i
struct bitset_filter {
  const uint32_t* words = nullptr;  // device memory
  int64_t n_bits        = 0;
};
}  // namespace filtering

namespace brute_force {
template <typename T = float>
class index {
 public:
  index() { check(cuvsBruteForceIndexCreate(&h_), "cuvsBruteForceIndexCreate"); }
  ~index() { if (h_) cuvsBruteForceIndexDestroy(h_); }
  index(index&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  index(const index&) = delete;
  cuvsBruteForceIndex_t get() const { return h_; }

 private:
  cuvsBruteForceIndex_t h_ = nullptr;
};
// brute_force.hpp: build(res, index_params / metric, dataset) - the index VIEWS the dataset (keep it alive)
template <typename T>
index<T> build(const resources& res, device_matrix_view<const T> dataset, cuvsDistanceType metric = L2Expanded,
               float metric_arg = 2.0f)
{
  index<T> idx;
  detail::tensor<const T> d(dataset);
  check(cuvsBruteForceBuild(res.get(), d.get(), metric, metric_arg, idx.get()), "cuvsBruteForceBuild");
  return idx;
}
template <typename T>
void search(const resources& res, const index<T>& idx, device_matrix_view<const T> queries,
            device_matrix_view<int64_t> neighbors, device_matrix_view<float> distances)
{
  detail::tensor<const T> q(queries);
  detail::tensor<int64_t> n(neighbors);
  detail::tensor<float> d(distances);
  cuvsFilter none{0, NO_FILTER};
  check(cuvsBruteForceSearch(res.get(), idx.get(), q.get(), n.get(), d.get(), none), "cuvsBruteForceSearch");
}
}  // namespace brute_force

namespace ivf_flat {
struct index_params {  // ivf_flat.hpp:29-75
  cuvsDistanceType metric              = L2Expanded;
  float metric_arg                     = 2.0f;
  bool add_data_on_build               = true;
  uint32_t n_lists                     = 1024;
  uint32_t kmeans_n_iters              = 20;
  double kmeans_trainset_fraction      = 0.5;
  bool adaptive_centers                = false;
  bool conservative_memory_allocation  = false;
};
struct search_params {  // ivf_flat.hpp:77-103
  uint32_t n_probes = 20;
};
template <typename T = float>
class index {
 public:
  index() { check(cuvsIvfFlatIndexCreate(&h_), "cuvsIvfFlatIndexCreate"); }
  ~index() { if (h_) cuvsIvfFlatIndexDestroy(h_); }
  index(index&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  index(const index&) = delete;
  cuvsIvfFlatIndex_t get() const { return h_; }

 private:
  cuvsIvfFlatIndex_t h_ = nullptr;
};
template <typename T>
index<T> build(const resources& res, const index_params& p, device_matrix_view<const T> dataset)
{
  detail::c_params<cuvsIvfFlatIndexParams_t, cuvsIvfFlatIndexParamsCreate, cuvsIvfFlatIndexParamsDestroy> cp;
  *cp.p = cuvsIvfFlatIndexParams{p.metric, p.metric_arg, p.add_data_on_build, p.n_lists, p.kmeans_n_iters,
                                 p.kmeans_trainset_fraction, p.adaptive_centers, p.conservative_memory_allocation};
  index<T> idx;
  detail::tensor<const T> d(dataset);
  check(cuvsIvfFlatBuild(res.get(), cp.p, d.get(), idx.get()), "cuvsIvfFlatBuild");
  return idx;
}
template <typename T>
void search(const resources& res, const search_params& p, const index<T>& idx, device_matrix_view<const T> queries,
            device_matrix_view<int64_t> neighbors, device_matrix_view<float> distances)
{
  detail::c_params<cuvsIvfFlatSearchParams_t, cuvsIvfFlatSearchParamsCreate, cuvsIvfFlatSearchParamsDestroy> cp;
  cp.p->n_probes = p.n_probes;
  detail::tensor<const T> q(queries);
  detail::tensor<int64_t> n(neighbors);
  detail::tensor<float> d(distances);
  cuvsFilter none{0, NO_FILTER};
  check(cuvsIvfFlatSearch(res.get(), cp.p, idx.get(), q.get(), n.get(), d.get(), none), "cuvsIvfFlatSearch");
}
}  // namespace ivf_flat

namespace ivf_pq {
enum class codebook_gen { PER_SUBSPACE = 0, PER_CLUSTER = 1 };
struct index_params {  // ivf_pq.hpp:40-160
  cuvsDistanceType metric               = L2Expanded;
  float metric_arg                      = 2.0f;
  bool add_data_on_build                = true;
  uint32_t n_lists                      = 1024;
  uint32_t kmeans_n_iters               = 20;
  double kmeans_trainset_fraction       = 0.5;
  uint32_t pq_bits                      = 8;
  uint32_t pq_dim                       = 0;
  codebook_gen codebook_kind            = codebook_gen::PER_SUBSPACE;
  bool force_random_rotation            = false;
  bool conservative_memory_allocation   = false;
  uint32_t max_train_points_per_pq_code = 256;
};
struct search_params {  // ivf_pq.hpp:162-236
  uint32_t n_probes                      = 20;
  cudaDataType_t lut_dtype               = CUDA_R_32F;
  cudaDataType_t internal_distance_dtype = CUDA_R_32F;
  cudaDataType_t coarse_search_dtype     = CUDA_R_32F;
  uint32_t max_internal_batch_size       = 4096;
  double preferred_shmem_carveout        = 1.0;
};
class index {
 public:
  index() { check(cuvsIvfPqIndexCreate(&h_), "cuvsIvfPqIndexCreate"); }
  ~index() { if (h_) cuvsIvfPqIndexDestroy(h_); }
  index(index&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  index(const index&) = delete;
  cuvsIvfPqIndex_t get() const { return h_; }
  int64_t size() const { int64_t v = 0; check(cuvsIvfPqIndexGetSize(h_, &v), "cuvsIvfPqIndexGetSize"); return v; }
  int64_t n_lists() const { int64_t v = 0; check(cuvsIvfPqIndexGetNLists(h_, &v), "cuvsIvfPqIndexGetNLists"); return v; }
  int64_t pq_dim() const { int64_t v = 0; check(cuvsIvfPqIndexGetPqDim(h_, &v), "cuvsIvfPqIndexGetPqDim"); return v; }

 private:
  cuvsIvfPqIndex_t h_ = nullptr;
};
template <typename T>
index build(const resources& res, const index_params& p, device_matrix_view<const T> dataset)
{
  detail::c_params<cuvsIvfPqIndexParams_t, cuvsIvfPqIndexParamsCreate, cuvsIvfPqIndexParamsDestroy> cp;
  cp.p->metric = p.metric; cp.p->metric_arg = p.metric_arg; cp.p->add_data_on_build = p.add_data_on_build;
  cp.p->n_lists = p.n_lists; cp.p->kmeans_n_iters = p.kmeans_n_iters;
  cp.p->kmeans_trainset_fraction = p.kmeans_trainset_fraction; cp.p->pq_bits = p.pq_bits; cp.p->pq_dim = p.pq_dim;
  cp.p->codebook_kind = (cuvsIvfPqCodebookGen)p.codebook_kind; cp.p->force_random_rotation = p.force_random_rotation;
  cp.p->conservative_memory_allocation = p.conservative_memory_allocation;
  cp.p->max_train_points_per_pq_code   = p.max_train_points_per_pq_code;
  index idx;
  detail::tensor<const T> d(dataset);
  check(cuvsIvfPqBuild(res.get(), cp.p, d.get(), idx.get()), "cuvsIvfPqBuild");
  return idx;
}
template <typename T>
void extend(const resources& res, device_matrix_view<const T> new_vectors, device_matrix_view<const int64_t> new_indices,
            index* idx)
{
  detail::tensor<const T> v(new_vectors);
  detail::tensor<const int64_t> ids(new_indices);
  ids.m.dl_tensor.ndim = 1;  // [n] ids
  check(cuvsIvfPqExtend(res.get(), v.get(), ids.get(), idx->get()), "cuvsIvfPqExtend");
}
template <typename T>
void search(const resources& res, const search_params& p, const index& idx, device_matrix_view<const T> queries,
            device_matrix_view<int64_t> neighbors, device_matrix_view<float> distances)
{
  detail::c_params<cuvsIvfPqSearchParams_t, cuvsIvfPqSearchParamsCreate, cuvsIvfPqSearchParamsDestroy> cp;
  cp.p->n_probes = p.n_probes; cp.p->lut_dtype = p.lut_dtype; cp.p->internal_distance_dtype = p.internal_distance_dtype;
  cp.p->coarse_search_dtype = p.coarse_search_dtype; cp.p->max_internal_batch_size = p.max_internal_batch_size;
  cp.p->preferred_shmem_carveout = p.preferred_shmem_carveout;
  detail::tensor<const T> q(queries);
  detail::tensor<int64_t> n(neighbors);
  detail::tensor<float> d(distances);
  check(cuvsIvfPqSearch(res.get(), cp.p, idx.get(), q.get(), n.get(), d.get()), "cuvsIvfPqSearch");
}
// search with a sample filter (ivf_pq.hpp:1818-1828); the C entry point is this library's (cuvs_amd/extensions.h)
template <typename T>
void search(const resources& res, const search_params& p, const index& idx, device_matrix_view<const T> queries,
            device_matrix_view<int64_t> neighbors, device_matrix_view<float> distances,
            const filtering::bitset_filter& sample_filter)
{
  detail::c_params<cuvsIvfPqSearchParams_t, cuvsIvfPqSearchParamsCreate, cuvsIvfPqSearchParamsDestroy> cp;
  cp.p->n_probes = p.n_probes; cp.p->lut_dtype = p.lut_dtype; cp.p->internal_distance_dtype = p.internal_distance_dtype;
  cp.p->coarse_search_dtype = p.coarse_search_dtype; cp.p->max_internal_batch_size = p.max_internal_batch_size;
  cp.p->preferred_shmem_carveout = p.preferred_shmem_carveout;
  detail::tensor<const T> q(queries);
  detail::tensor<int64_t> n(neighbors);
  detail::tensor<float> d(distances);
  detail::tensor<const uint32_t> f(device_matrix_view<const uint32_t>{sample_filter.words, 1, (sample_filter.n_bits + 31) / 32, false});
  f.m.dl_tensor.ndim = 1; f.shape[0] = (sample_filter.n_bits + 31) / 32;
  cuvsFilter flt{reinterpret_cast<uintptr_t>(f.get()), BITSET};
  check(cuvsAmdIvfPqSearchFiltered(res.get(), cp.p, idx.get(), q.get(), n.get(), d.get(), flt), "cuvsAmdIvfPqSearchFiltered");
}
}  // namespace ivf_pq

namespace cagra {
struct index_params {  // cagra.hpp:84-200
  cuvsDistanceType metric          = L2Expanded;
  size_t intermediate_graph_degree = 128;
  size_t graph_degree              = 64;
  cuvsCagraGraphBuildAlgo build_algo = IVF_PQ;
  size_t nn_descent_niter          = 20;
  bool guarantee_connectivity      = false;  // cagra.hpp:193 (handle switch here: cuvsAmdCagraSetGuaranteeConnectivity)
};
enum class search_algo { SINGLE_CTA = 0, MULTI_CTA = 1, MULTI_KERNEL = 2, AUTO = 100 };
struct search_params {  // cagra.hpp:203-360
  size_t max_queries          = 0;
  size_t itopk_size           = 64;
  size_t max_iterations       = 0;
  search_algo algo            = search_algo::AUTO;
  size_t team_size            = 0;
  size_t search_width         = 1;
  size_t min_iterations       = 0;
  size_t thread_block_size    = 0;
  size_t hashmap_min_bitlen   = 0;
  float hashmap_max_fill_rate = 0.5f;
  uint32_t num_random_samplings = 1;
  uint64_t rand_xor_mask      = 0x128394;
};
template <typename T = float>
class index {
 public:
  index() { check(cuvsCagraIndexCreate(&h_), "cuvsCagraIndexCreate"); }
  ~index() { if (h_) cuvsCagraIndexDestroy(h_); }
  index(index&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  index(const index&) = delete;
  cuvsCagraIndex_t get() const { return h_; }
  int64_t size() const { int64_t v = 0; check(cuvsCagraIndexGetSize(h_, &v), "cuvsCagraIndexGetSize"); return v; }
  int64_t graph_degree() const
  {
    int64_t v = 0;
    check(cuvsCagraIndexGetGraphDegree(h_, &v), "cuvsCagraIndexGetGraphDegree");
    return v;
  }

 private:
  cuvsCagraIndex_t h_ = nullptr;
};
template <typename T>
index<T> build(const resources& res, const index_params& p, device_matrix_view<const T> dataset)
{
  detail::c_params<cuvsCagraIndexParams_t, cuvsCagraIndexParamsCreate, cuvsCagraIndexParamsDestroy> cp;
  cp.p->metric = p.metric; cp.p->intermediate_graph_degree = p.intermediate_graph_degree;
  cp.p->graph_degree = p.graph_degree; cp.p->build_algo = p.build_algo; cp.p->nn_descent_niter = p.nn_descent_niter;
  index<T> idx;
  detail::tensor<const T> d(dataset);
  check(cuvsAmdCagraSetGuaranteeConnectivity(res.get(), p.guarantee_connectivity ? 1 : 0), "cuvsAmdCagraSetGuaranteeConnectivity");
  const cuvsError_t rc = cuvsCagraBuild(res.get(), cp.p, d.get(), idx.get());
  cuvsAmdCagraSetGuaranteeConnectivity(res.get(), 0);
  check(rc, "cuvsCagraBuild");
  return idx;
}
// neighbors: uint32_t (the reference's index type) or int64_t
template <typename T, typename IdxT>
void search(const resources& res, const search_params& p, const index<T>& idx, device_matrix_view<const T> queries,
            device_matrix_view<IdxT> neighbors, device_matrix_view<float> distances)
{
  detail::c_params<cuvsCagraSearchParams_t, cuvsCagraSearchParamsCreate, cuvsCagraSearchParamsDestroy> cp;
  cp.p->max_queries = p.max_queries; cp.p->itopk_size = p.itopk_size; cp.p->max_iterations = p.max_iterations;
  cp.p->algo = (cuvsCagraSearchAlgo)p.algo; cp.p->team_size = p.team_size; cp.p->search_width = p.search_width;
  cp.p->min_iterations = p.min_iterations; cp.p->thread_block_size = p.thread_block_size;
  cp.p->hashmap_min_bitlen = p.hashmap_min_bitlen; cp.p->hashmap_max_fill_rate = p.hashmap_max_fill_rate;
  cp.p->num_random_samplings = p.num_random_samplings; cp.p->rand_xor_mask = p.rand_xor_mask;
  detail::tensor<const T> q(queries);
  detail::tensor<IdxT> n(neighbors);
  detail::tensor<float> d(distances);
  cuvsFilter none{0, NO_FILTER};
  check(cuvsCagraSearch(res.get(), cp.p, idx.get(), q.get(), n.get(), d.get(), none), "cuvsCagraSearch");
}
}  // namespace cagra

}  // namespace neighbors
}  // namespace cuvs
