/*
 * Multi-GPU CAGRA in one process — drop-in for c/include/cuvs/neighbors/mg_cagra.h (structs and handles :30-132, entry points :152-276; wrapper
 * c/src/neighbors/mg_cagra.cpp, algorithm cpp/src/neighbors/mg/snmg.cuh). `res` is a cuvsMultiGpuResources handle;
 * dataset, queries, neighbors (int64) and distances (fp32) are HOST tensors, as in the reference.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/neighbors/cagra.h>
#include <cuvs/neighbors/mg_common.h>
#include <dlpack/dlpack.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct cuvsMultiGpuCagraIndexParams {
  cuvsCagraIndexParams_t base_params;  /* owned: created and destroyed with this struct */
  cuvsMultiGpuDistributionMode mode; /* default SHARDED */
};
typedef struct cuvsMultiGpuCagraIndexParams* cuvsMultiGpuCagraIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraIndexParamsCreate(cuvsMultiGpuCagraIndexParams_t* index_params);
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraIndexParamsDestroy(cuvsMultiGpuCagraIndexParams_t index_params);

struct cuvsMultiGpuCagraSearchParams {
  cuvsCagraSearchParams_t base_params;           /* owned */
  cuvsMultiGpuReplicatedSearchMode search_mode;  /* default LOAD_BALANCER */
  cuvsMultiGpuShardedMergeMode merge_mode;       /* default TREE_MERGE */
  int64_t n_rows_per_batch;                      /* default 1 << 20 queries */
};
typedef struct cuvsMultiGpuCagraSearchParams* cuvsMultiGpuCagraSearchParams_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraSearchParamsCreate(cuvsMultiGpuCagraSearchParams_t* params);
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraSearchParamsDestroy(cuvsMultiGpuCagraSearchParams_t params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsMultiGpuCagraIndex;
typedef cuvsMultiGpuCagraIndex* cuvsMultiGpuCagraIndex_t;
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraIndexCreate(cuvsMultiGpuCagraIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraIndexDestroy(cuvsMultiGpuCagraIndex_t index);

CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraBuild(cuvsResources_t res, cuvsMultiGpuCagraIndexParams_t params,
                                               DLManagedTensor* dataset_tensor, cuvsMultiGpuCagraIndex_t index);
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraSearch(cuvsResources_t res, cuvsMultiGpuCagraSearchParams_t params,
                                                cuvsMultiGpuCagraIndex_t index, DLManagedTensor* queries_tensor,
                                                DLManagedTensor* neighbors_tensor, DLManagedTensor* distances_tensor);
/* new_indices_tensor may be NULL: ids continue from each shard's current size (snmg.cuh:170-246). */
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraExtend(cuvsResources_t res, cuvsMultiGpuCagraIndex_t index,
                                                DLManagedTensor* new_vectors_tensor, DLManagedTensor* new_indices_tensor);
/* One file: dtype prefix, mode, number of GPUs, then the per-GPU index streams back to back (snmg.cuh:735-757). */
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraSerialize(cuvsResources_t res, cuvsMultiGpuCagraIndex_t index, const char* filename);
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraDeserialize(cuvsResources_t res, const char* filename, cuvsMultiGpuCagraIndex_t index);
/* Loads a single-GPU index file onto every GPU (REPLICATED), snmg.cuh:43-55. */
CUVS_EXPORT cuvsError_t cuvsMultiGpuCagraDistribute(cuvsResources_t res, const char* filename, cuvsMultiGpuCagraIndex_t index);

#ifdef __cplusplus
}
#endif
