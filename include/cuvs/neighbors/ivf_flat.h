/*
 * IVF-Flat entry points — drop-in for c/include/cuvs/neighbors/ivf_flat.h.
 * Struct field order and sizes are ABI (ivf_flat.h:29-103): callers mutate fields directly.
 * Implemented by cuvs_amd/csrc/ivf_flat.hip.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/distance/distance.h>
#include <cuvs/neighbors/common.h>
#include <dlpack/dlpack.h>
#include <stdbool.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

struct cuvsIvfFlatIndexParams {
  cuvsDistanceType metric;          /* default L2Expanded */
  float metric_arg;                 /* 2.0 */
  bool add_data_on_build;           /* true */
  uint32_t n_lists;                 /* 1024 */
  uint32_t kmeans_n_iters;          /* 20 */
  double kmeans_trainset_fraction;  /* 0.5 */
  bool adaptive_centers;            /* false */
  bool conservative_memory_allocation;
};
typedef struct cuvsIvfFlatIndexParams* cuvsIvfFlatIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexParamsCreate(cuvsIvfFlatIndexParams_t* index_params);
CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexParamsDestroy(cuvsIvfFlatIndexParams_t index_params);

struct cuvsIvfFlatSearchParams {
  uint32_t n_probes; /* 20 */
};
typedef struct cuvsIvfFlatSearchParams* cuvsIvfFlatSearchParams_t;
CUVS_EXPORT cuvsError_t cuvsIvfFlatSearchParamsCreate(cuvsIvfFlatSearchParams_t* params);
CUVS_EXPORT cuvsError_t cuvsIvfFlatSearchParamsDestroy(cuvsIvfFlatSearchParams_t params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsIvfFlatIndex;
typedef cuvsIvfFlatIndex* cuvsIvfFlatIndex_t;
CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexCreate(cuvsIvfFlatIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexDestroy(cuvsIvfFlatIndex_t index);

CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexGetNLists(cuvsIvfFlatIndex_t index, int64_t* n_lists);
CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexGetDim(cuvsIvfFlatIndex_t index, int64_t* dim);
CUVS_EXPORT cuvsError_t cuvsIvfFlatIndexGetCenters(cuvsIvfFlatIndex_t index, DLManagedTensor* centers);

/* ivf_flat.h:236 — dataset host or device, fp32/fp16/int8/uint8, row-major [n, dim] */
CUVS_EXPORT cuvsError_t cuvsIvfFlatBuild(cuvsResources_t res,
                                         cuvsIvfFlatIndexParams_t index_params,
                                         DLManagedTensor* dataset,
                                         cuvsIvfFlatIndex_t index);

/* ivf_flat.h:293 — neighbors int64 [m,k], distances fp32 [m,k], filter NO_FILTER or BITSET */
CUVS_EXPORT cuvsError_t cuvsIvfFlatSearch(cuvsResources_t res,
                                          cuvsIvfFlatSearchParams_t search_params,
                                          cuvsIvfFlatIndex_t index,
                                          DLManagedTensor* queries,
                                          DLManagedTensor* neighbors,
                                          DLManagedTensor* distances,
                                          cuvsFilter filter);

CUVS_EXPORT cuvsError_t cuvsIvfFlatSerialize(cuvsResources_t res,
                                             const char* filename,
                                             cuvsIvfFlatIndex_t index);
CUVS_EXPORT cuvsError_t cuvsIvfFlatDeserialize(cuvsResources_t res,
                                               const char* filename,
                                               cuvsIvfFlatIndex_t index);
/* ivf_flat.h:362 — new_indices may be NULL (ids continue from the current size) */
CUVS_EXPORT cuvsError_t cuvsIvfFlatExtend(cuvsResources_t res,
                                          DLManagedTensor* new_vectors,
                                          DLManagedTensor* new_indices,
                                          cuvsIvfFlatIndex_t index);
#ifdef __cplusplus
}
#endif
