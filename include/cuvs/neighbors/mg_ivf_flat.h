/*
 * Multi-GPU IVF-Flat in one process — drop-in for c/include/cuvs/neighbors/mg_ivf_flat.h (structs and handles :30-132, entry points :152-276; wrapper
 * c/src/neighbors/mg_ivf_flat.cpp, algorithm cpp/src/neighbors/mg/snmg.cuh). `res` is a cuvsMultiGpuResources handle;
 * dataset, queries, neighbors (int64) and distances (fp32) are HOST tensors, as in the reference.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/neighbors/ivf_flat.h>
#include <cuvs/neighbors/mg_common.h>
#include <dlpack/dlpack.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct cuvsMultiGpuIvfFlatIndexParams {
  cuvsIvfFlatIndexParams_t base_params;  /* owned: created and destroyed with this struct */
  cuvsMultiGpuDistributionMode mode; /* default SHARDED */
};
typedef struct cuvsMultiGpuIvfFlatIndexParams* cuvsMultiGpuIvfFlatIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatIndexParamsCreate(cuvsMultiGpuIvfFlatIndexParams_t* index_params);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatIndexParamsDestroy(cuvsMultiGpuIvfFlatIndexParams_t index_params);

struct cuvsMultiGpuIvfFlatSearchParams {
  cuvsIvfFlatSearchParams_t base_params;           /* owned */
  cuvsMultiGpuReplicatedSearchMode search_mode;  /* default LOAD_BALANCER */
  cuvsMultiGpuShardedMergeMode merge_mode;       /* default TREE_MERGE */
  int64_t n_rows_per_batch;                      /* default 1 << 20 queries */
};
typedef struct cuvsMultiGpuIvfFlatSearchParams* cuvsMultiGpuIvfFlatSearchParams_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatSearchParamsCreate(cuvsMultiGpuIvfFlatSearchParams_t* params);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatSearchParamsDestroy(cuvsMultiGpuIvfFlatSearchParams_t params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsMultiGpuIvfFlatIndex;
typedef cuvsMultiGpuIvfFlatIndex* cuvsMultiGpuIvfFlatIndex_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatIndexCreate(cuvsMultiGpuIvfFlatIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatIndexDestroy(cuvsMultiGpuIvfFlatIndex_t index);

CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatBuild(cuvsResources_t res, cuvsMultiGpuIvfFlatIndexParams_t params,
                                                 DLManagedTensor* dataset_tensor, cuvsMultiGpuIvfFlatIndex_t index);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatSearch(cuvsResources_t res, cuvsMultiGpuIvfFlatSearchParams_t params,
                                                  cuvsMultiGpuIvfFlatIndex_t index, DLManagedTensor* queries_tensor,
                                                  DLManagedTensor* neighbors_tensor, DLManagedTensor* distances_tensor);
/* new_indices_tensor may be NULL: ids continue from each shard's current size (snmg.cuh:170-246). */
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatExtend(cuvsResources_t res, cuvsMultiGpuIvfFlatIndex_t index,
                                                  DLManagedTensor* new_vectors_tensor, DLManagedTensor* new_indices_tensor);
/* One file: dtype prefix, mode, number of GPUs, then the per-GPU index streams back to back (snmg.cuh:735-757). */
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatSerialize(cuvsResources_t res, cuvsMultiGpuIvfFlatIndex_t index, const char* filename);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatDeserialize(cuvsResources_t res, const char* filename, cuvsMultiGpuIvfFlatIndex_t index);
/* Loads a single-GPU index file onto every GPU (REPLICATED), snmg.cuh:43-55. */
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfFlatDistribute(cuvsResources_t res, const char* filename, cuvsMultiGpuIvfFlatIndex_t index);

#ifdef __cplusplus
}
#endif
