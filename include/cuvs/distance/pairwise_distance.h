/*
 * All-pairs distances between the rows of two device matrices - drop-in for
 * c/include/cuvs/distance/pairwise_distance.h (impl c/src/distance/pairwise_distance.cpp). This is the distance GEMM
 * of the brute-force path (cuvs_amd/csrc/distance.hip: fp32 MFMA, k-ordered fma chain) exposed on its own.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/core/export.h>
#include <cuvs/distance/distance.h>
#include <dlpack/dlpack.h>
#ifdef __cplusplus
extern "C" {
#endif
/* x [m, d], y [n, d] (fp32 or fp16, same dtype, row-major, device) -> dist [m, n] fp32 row-major.
 * metric: L2Expanded / L2SqrtExpanded / L2Unexpanded / L2SqrtUnexpanded / CosineExpanded / InnerProduct. */
CUVS_EXPORT cuvsError_t cuvsPairwiseDistance(cuvsResources_t res, DLManagedTensor* x, DLManagedTensor* y,
                                             DLManagedTensor* dist, cuvsDistanceType metric, float metric_arg);
#ifdef __cplusplus
}
#endif
