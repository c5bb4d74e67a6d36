/* Entry points of libcuvs_c.so that the reference's C headers do not declare. Everything a binding needs for the
 * reference's API lives in <cuvs/...>; these are additions in the same conventions (cuvsError_t, DLPack tensors).
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/core/export.h>
#include <cuvs/neighbors/common.h>
#include <cuvs/neighbors/ivf_pq.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* IVF-PQ search with a pre-filter. The reference has this only in C++ - cuvs::neighbors::ivf_pq::search(..., const
 * filtering::base_filter& sample_filter), cpp/include/cuvs/neighbors/ivf_pq.hpp:1818-1828; kernel side
 * cpp/src/neighbors/ivf_pq/detail/jit_lto_kernels/compute_distances_impl.cuh:78-80 - its C entry point
 * cuvsIvfPqSearch (c/include/cuvs/neighbors/ivf_pq.h:536-541) takes no filter. Same argument convention as
 * cuvsIvfFlatSearch: filter.type NO_FILTER or BITSET, filter.addr = DLManagedTensor* of uint32 words on the device,
 * bit i = 1 keeps source id i. */
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqSearchFiltered(cuvsResources_t res, cuvsIvfPqSearchParams_t search_params,
                                                   cuvsIvfPqIndex_t index, DLManagedTensor* queries,
                                                   DLManagedTensor* neighbors, DLManagedTensor* distances,
                                                   cuvsFilter filter);

/* Measured work of the CAGRA graph walks run on `res` (the hot path's algorithmic bytes per query are n_dist * dim *
 * sizeof(T) + n_iter * graph_degree * 4 with n_dist, n_iter measured; the reference keeps per-phase clock counters,
 * cpp/src/neighbors/detail/cagra/search_single_cta_jit.cuh:91-103,425-451). enable != 0 zeroes the counters and starts
 * counting; enable == 0 stops and fills out = {rows whose distance was computed, graph rows read, walkers (waves)}. */
CUVS_EXPORT cuvsError_t cuvsAmdCagraWorkCounters(cuvsResources_t res, int enable, uint64_t out[3]);

/* cagra::index_params::guarantee_connectivity (cpp/include/cuvs/neighbors/cagra.hpp:193; the spanning-forest pass of
 * graph::optimize, cpp/src/neighbors/detail/cagra/graph_core.cuh:1186-1581,1747-1760) is C++-only in the reference: the C
 * struct cuvsCagraIndexParams has no such field. The switch is kept on the handle and applies to every cuvsCagraBuild
 * made with it. */
CUVS_EXPORT cuvsError_t cuvsAmdCagraSetGuaranteeConnectivity(cuvsResources_t res, int on);

/* cuvs::neighbors::cagra::helpers::optimize (cpp/include/cuvs/neighbors/cagra_optimize.hpp): kNN graph [n, K] uint32 ->
 * search graph [n, degree] uint32 (prune by 2-hop detours, reverse edges, optional connectivity guarantee). Either
 * tensor may live on the host or on the device. */
CUVS_EXPORT cuvsError_t cuvsAmdCagraOptimize(cuvsResources_t res, DLManagedTensor* knn_graph, DLManagedTensor* graph,
                                             int guarantee_connectivity);

/* index.codes_layout() of the reference's C++ index (cpp/include/cuvs/neighbors/ivf_pq.hpp:40-90; the C ABI sets the layout in
 * cuvsIvfPqIndexParams but has no getter): 0 = CUVS_IVF_PQ_LIST_LAYOUT_FLAT, 1 = CUVS_IVF_PQ_LIST_LAYOUT_INTERLEAVED. */
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqIndexGetCodesLayout(cuvsIvfPqIndex_t index, int* layout);

/* Measurement helpers of bench.py (no reference counterpart). cuvsAmdProfileEnable / cuvsAmdProfileCollect: HIP events
 * around the named kernels on the handle's stream (Collect sums and resets the records of `name`, returns the launch count).
 * cuvsAmdIvfPqLastFilterStats: counters of the last IVF-PQ search made by a handle created under CUVS_AMD_SCAN_DEBUG=1024
 * (behind CUVS_AMD_DEBUG_SWITCHES=1): out = {(row, query) pairs screened by the matrix-core filter, survivors re-scored,
 * 32-row subtiles decoded, work units}. */
CUVS_EXPORT int cuvsAmdDebugSwitchesCompiledIn(void); /* 0: built with -DCUVS_AMD_NO_DEBUG_SWITCHES (make PRODUCTION=1) */
CUVS_EXPORT void cuvsAmdProfileEnable(int on);
CUVS_EXPORT int cuvsAmdProfileCollect(const char* name, double* total_ms);
CUVS_EXPORT void cuvsAmdIvfPqLastFilterStats(unsigned long long out[4]);
/* the same + out[4] = (query, probe) pairs handed back to the LUT scan kernels, out[5] = candidates that went through the
 * shared overflow list */
CUVS_EXPORT void cuvsAmdIvfPqLastFilterStats6(unsigned long long out[6]);

#ifdef __cplusplus
}
#endif
