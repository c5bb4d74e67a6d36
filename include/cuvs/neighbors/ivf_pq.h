/*
 * IVF-PQ entry points — drop-in for c/include/cuvs/neighbors/ivf_pq.h.
 * Struct field order and sizes are ABI (ivf_pq.h:46-205). Implemented by cuvs_amd/csrc/ivf_pq.hip.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/distance/distance.h>
#include <dlpack/dlpack.h>
#include <stdbool.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum cuvsIvfPqCodebookGen {
  CUVS_IVF_PQ_CODEBOOK_GEN_PER_SUBSPACE = 0,
  CUVS_IVF_PQ_CODEBOOK_GEN_PER_CLUSTER  = 1,
};
enum cuvsIvfPqListLayout {
  CUVS_IVF_PQ_LIST_LAYOUT_FLAT        = 0,
  CUVS_IVF_PQ_LIST_LAYOUT_INTERLEAVED = 1,
};

struct cuvsIvfPqIndexParams {
  cuvsDistanceType metric;               /* L2Expanded */
  float metric_arg;                      /* 2.0 */
  bool add_data_on_build;                /* true */
  uint32_t n_lists;                      /* 1024 */
  uint32_t kmeans_n_iters;               /* 20 */
  double kmeans_trainset_fraction;       /* 0.5 */
  uint32_t pq_bits;                      /* 8   (4..8) */
  uint32_t pq_dim;                       /* 0 = choose from dim */
  enum cuvsIvfPqCodebookGen codebook_kind;
  bool force_random_rotation;
  bool conservative_memory_allocation;
  uint32_t max_train_points_per_pq_code; /* 256 */
  enum cuvsIvfPqListLayout codes_layout; /* INTERLEAVED */
};
typedef struct cuvsIvfPqIndexParams* cuvsIvfPqIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexParamsCreate(cuvsIvfPqIndexParams_t* index_params);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexParamsDestroy(cuvsIvfPqIndexParams_t index_params);

struct cuvsIvfPqSearchParams {
  uint32_t n_probes;                       /* 20 */
  cudaDataType_t lut_dtype;                /* CUDA_R_32F | CUDA_R_16F | CUDA_R_8U */
  cudaDataType_t internal_distance_dtype;  /* CUDA_R_32F | CUDA_R_16F */
  cudaDataType_t coarse_search_dtype;      /* CUDA_R_32F | CUDA_R_16F | CUDA_R_8I */
  uint32_t max_internal_batch_size;        /* 4096 */
  double preferred_shmem_carveout;         /* ignored on CDNA4: LDS is not shared with L1 */
};
typedef struct cuvsIvfPqSearchParams* cuvsIvfPqSearchParams_t;
CUVS_EXPORT cuvsError_t cuvsIvfPqSearchParamsCreate(cuvsIvfPqSearchParams_t* params);
CUVS_EXPORT cuvsError_t cuvsIvfPqSearchParamsDestroy(cuvsIvfPqSearchParams_t params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsIvfPqIndex;
typedef cuvsIvfPqIndex* cuvsIvfPqIndex_t;
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexCreate(cuvsIvfPqIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexDestroy(cuvsIvfPqIndex_t index);

/* getters (ivf_pq.h:263-420). Tensor getters fill a caller-allocated DLManagedTensor with a
 * non-owning device view whose shape array is released by the tensor's deleter. */
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetNLists(cuvsIvfPqIndex_t index, int64_t* n_lists);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetDim(cuvsIvfPqIndex_t index, int64_t* dim);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetSize(cuvsIvfPqIndex_t index, int64_t* size);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetPqDim(cuvsIvfPqIndex_t index, int64_t* pq_dim);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetPqBits(cuvsIvfPqIndex_t index, int64_t* pq_bits);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetPqLen(cuvsIvfPqIndex_t index, int64_t* pq_len);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetCenters(cuvsIvfPqIndex_t index, DLManagedTensor* centers);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetCentersPadded(cuvsIvfPqIndex_t index, DLManagedTensor* centers);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetPqCenters(cuvsIvfPqIndex_t index, DLManagedTensor* pq_centers);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetCentersRot(cuvsIvfPqIndex_t index, DLManagedTensor* centers_rot);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetRotationMatrix(cuvsIvfPqIndex_t index,
                                                        DLManagedTensor* rotation_matrix);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetListSizes(cuvsIvfPqIndex_t index, DLManagedTensor* list_sizes);
/* out_codes: device uint8 [n_take, ceil(pq_dim*pq_bits/8)] — one contiguous bit-packed code per row,
 * starting at in-list `offset` of list `label`. This is the externally visible code format. */
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexUnpackContiguousListData(cuvsResources_t res,
                                                               cuvsIvfPqIndex_t index,
                                                               DLManagedTensor* out_codes,
                                                               uint32_t label,
                                                               uint32_t offset);
CUVS_EXPORT cuvsError_t cuvsIvfPqIndexGetListIndices(cuvsIvfPqIndex_t index,
                                                     uint32_t label,
                                                     DLManagedTensor* out_labels);

CUVS_EXPORT cuvsError_t cuvsIvfPqBuild(cuvsResources_t res,
                                       cuvsIvfPqIndexParams_t params,
                                       DLManagedTensor* dataset,
                                       cuvsIvfPqIndex_t index);
CUVS_EXPORT cuvsError_t cuvsIvfPqBuildPrecomputed(cuvsResources_t res,
                                                  cuvsIvfPqIndexParams_t params,
                                                  uint32_t dim,
                                                  DLManagedTensor* pq_centers,
                                                  DLManagedTensor* centers,
                                                  DLManagedTensor* centers_rot,
                                                  DLManagedTensor* rotation_matrix,
                                                  cuvsIvfPqIndex_t index);
/* ivf_pq.h:536-541 — no filter argument. neighbors int64 [m,k], distances fp32 [m,k]. */
CUVS_EXPORT cuvsError_t cuvsIvfPqSearch(cuvsResources_t res,
                                        cuvsIvfPqSearchParams_t search_params,
                                        cuvsIvfPqIndex_t index,
                                        DLManagedTensor* queries,
                                        DLManagedTensor* neighbors,
                                        DLManagedTensor* distances);
CUVS_EXPORT cuvsError_t cuvsIvfPqSerialize(cuvsResources_t res, const char* filename, cuvsIvfPqIndex_t index);
CUVS_EXPORT cuvsError_t cuvsIvfPqDeserialize(cuvsResources_t res, const char* filename, cuvsIvfPqIndex_t index);
CUVS_EXPORT cuvsError_t cuvsIvfPqExtend(cuvsResources_t res,
                                        DLManagedTensor* new_vectors,
                                        DLManagedTensor* new_indices,
                                        cuvsIvfPqIndex_t index);
CUVS_EXPORT cuvsError_t cuvsIvfPqTransform(cuvsResources_t res,
                                           cuvsIvfPqIndex_t index,
                                           DLManagedTensor* input_dataset,
                                           DLManagedTensor* output_labels,
                                           DLManagedTensor* output_dataset);
#ifdef __cplusplus
}
#endif
