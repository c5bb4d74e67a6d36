/*
 * Brute-force kNN entry points — drop-in for c/include/cuvs/neighbors/brute_force.h.
 * Implemented by cuvs_amd/csrc/brute_force.hip (fp32 MFMA distance tiles + exact select_k).
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/distance/distance.h>
#include <cuvs/neighbors/common.h>
#include <dlpack/dlpack.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* brute_force.h:28-31 — handle = {pointer to the C++ index, dataset dtype} */
typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsBruteForceIndex;
typedef cuvsBruteForceIndex* cuvsBruteForceIndex_t;

CUVS_EXPORT cuvsError_t cuvsBruteForceIndexCreate(cuvsBruteForceIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsBruteForceIndexDestroy(cuvsBruteForceIndex_t index);

/* brute_force.h:94 / c/src/neighbors/brute_force.cpp:143-179.
 * dataset: device-accessible [n, d], fp32 or fp16, C-contiguous (F-contiguous accepted and
 * copied). The index keeps a NON-OWNING view of a C-contiguous device dataset
 * (cpp/src/neighbors/brute_force.cu:66-79): the caller keeps it alive. */
CUVS_EXPORT cuvsError_t cuvsBruteForceBuild(cuvsResources_t res,
                                            DLManagedTensor* dataset,
                                            cuvsDistanceType metric,
                                            float metric_arg,
                                            cuvsBruteForceIndex_t index);

/* brute_force.h:150 / brute_force.cpp:181-231. neighbors: int64 [m,k]; distances: fp32 [m,k].
 * prefilter: NO_FILTER, BITSET or BITMAP. */
CUVS_EXPORT cuvsError_t cuvsBruteForceSearch(cuvsResources_t res,
                                             cuvsBruteForceIndex_t index,
                                             DLManagedTensor* queries,
                                             DLManagedTensor* neighbors,
                                             DLManagedTensor* distances,
                                             cuvsFilter prefilter);

CUVS_EXPORT cuvsError_t cuvsBruteForceSerialize(cuvsResources_t res,
                                                const char* filename,
                                                cuvsBruteForceIndex_t index);
CUVS_EXPORT cuvsError_t cuvsBruteForceDeserialize(cuvsResources_t res,
                                                  const char* filename,
                                                  cuvsBruteForceIndex_t index);
#ifdef __cplusplus
}
#endif
