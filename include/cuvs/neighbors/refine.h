/*
 * Re-rank candidate neighbours with exact distances — drop-in for c/include/cuvs/neighbors/refine.h.
 * the standard way the reference lifts IVF-PQ recall (refine_ratio in its bench grids).
 * Implemented by cuvs_amd/csrc/refine.hip (device tensors only; the host path of the reference,
 * cpp/src/neighbors/refine/refine_host.hpp, is restated in oracle/oracle.c as the CPU baseline).
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/distance/distance.h>
#include <dlpack/dlpack.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* dataset [n, dim] and queries [m, dim]: fp32 / fp16 / int8 / uint8 (same dtype); candidates int64 [m, n_cand];
 * indices int64 [m, k], distances fp32 [m, k], k <= n_cand. Candidates outside [0, n) are skipped. */
CUVS_EXPORT cuvsError_t cuvsRefine(cuvsResources_t res, DLManagedTensor* dataset, DLManagedTensor* queries,
                                   DLManagedTensor* candidates, cuvsDistanceType metric, DLManagedTensor* indices,
                                   DLManagedTensor* distances);
#ifdef __cplusplus
}
#endif
