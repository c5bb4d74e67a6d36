/*
 * Multi-GPU IVF-PQ in one process — drop-in for c/include/cuvs/neighbors/mg_ivf_pq.h (structs and handles :30-132, entry points :152-276; wrapper
 * c/src/neighbors/mg_ivf_pq.cpp, algorithm cpp/src/neighbors/mg/snmg.cuh). `res` is a cuvsMultiGpuResources handle;
 * dataset, queries, neighbors (int64) and distances (fp32) are HOST tensors, as in the reference.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/neighbors/ivf_pq.h>
#include <cuvs/neighbors/mg_common.h>
#include <dlpack/dlpack.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct cuvsMultiGpuIvfPqIndexParams {
  cuvsIvfPqIndexParams_t base_params;  /* owned: created and destroyed with this struct */
  cuvsMultiGpuDistributionMode mode; /* default SHARDED */
};
typedef struct cuvsMultiGpuIvfPqIndexParams* cuvsMultiGpuIvfPqIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqIndexParamsCreate(cuvsMultiGpuIvfPqIndexParams_t* index_params);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqIndexParamsDestroy(cuvsMultiGpuIvfPqIndexParams_t index_params);

struct cuvsMultiGpuIvfPqSearchParams {
  cuvsIvfPqSearchParams_t base_params;           /* owned */
  cuvsMultiGpuReplicatedSearchMode search_mode;  /* default LOAD_BALANCER */
  cuvsMultiGpuShardedMergeMode merge_mode;       /* default TREE_MERGE */
  int64_t n_rows_per_batch;                      /* default 1 << 20 queries */
};
typedef struct cuvsMultiGpuIvfPqSearchParams* cuvsMultiGpuIvfPqSearchParams_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqSearchParamsCreate(cuvsMultiGpuIvfPqSearchParams_t* params);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqSearchParamsDestroy(cuvsMultiGpuIvfPqSearchParams_t params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsMultiGpuIvfPqIndex;
typedef cuvsMultiGpuIvfPqIndex* cuvsMultiGpuIvfPqIndex_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqIndexCreate(cuvsMultiGpuIvfPqIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqIndexDestroy(cuvsMultiGpuIvfPqIndex_t index);

CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqBuild(cuvsResources_t res, cuvsMultiGpuIvfPqIndexParams_t params,
                                               DLManagedTensor* dataset_tensor, cuvsMultiGpuIvfPqIndex_t index);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqSearch(cuvsResources_t res, cuvsMultiGpuIvfPqSearchParams_t params,
                                                cuvsMultiGpuIvfPqIndex_t index, DLManagedTensor* queries_tensor,
                                                DLManagedTensor* neighbors_tensor, DLManagedTensor* distances_tensor);
/* new_indices_tensor may be NULL: ids continue from each shard's current size (snmg.cuh:170-246). */
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqExtend(cuvsResources_t res, cuvsMultiGpuIvfPqIndex_t index,
                                                DLManagedTensor* new_vectors_tensor, DLManagedTensor* new_indices_tensor);
/* One file: dtype prefix, mode, number of GPUs, then the per-GPU index streams back to back (snmg.cuh:735-757). */
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqSerialize(cuvsResources_t res, cuvsMultiGpuIvfPqIndex_t index, const char* filename);
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqDeserialize(cuvsResources_t res, const char* filename, cuvsMultiGpuIvfPqIndex_t index);
/* Loads a single-GPU index file onto every GPU (REPLICATED), snmg.cuh:43-55. */
CUVS_EXPORT cuvsError_t cuvsMultiGpuIvfPqDistribute(cuvsResources_t res, const char* filename, cuvsMultiGpuIvfPqIndex_t index);

#ifdef __cplusplus
}
#endif
