/*
 * CAGRA entry points — drop-in for c/include/cuvs/neighbors/cagra.h (struct layouts :84-152,
 * :203-245, :373-440 are ABI). Implemented by cuvs_amd/csrc/cagra.hip.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/distance/distance.h>
#include <cuvs/neighbors/common.h>
#include <cuvs/neighbors/ivf_pq.h>
#include <dlpack/dlpack.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum cuvsCagraGraphBuildAlgo {
  AUTO_SELECT            = 0,
  IVF_PQ                 = 1,
  NN_DESCENT             = 2,
  ITERATIVE_CAGRA_SEARCH = 3,
  ACE                    = 4
};
enum cuvsCagraHnswHeuristicType {
  CUVS_CAGRA_HEURISTIC_SIMILAR_SEARCH_PERFORMANCE = 0,
  CUVS_CAGRA_HEURISTIC_SAME_GRAPH_FOOTPRINT       = 1
};

struct cuvsCagraCompressionParams {
  uint32_t pq_bits;
  uint32_t pq_dim;
  uint32_t vq_n_centers;
  uint32_t kmeans_n_iters;
  double vq_kmeans_trainset_fraction;
  double pq_kmeans_trainset_fraction;
};
typedef struct cuvsCagraCompressionParams* cuvsCagraCompressionParams_t;

struct cuvsIvfPqParams {
  cuvsIvfPqIndexParams_t ivf_pq_build_params;
  cuvsIvfPqSearchParams_t ivf_pq_search_params;
  float refinement_rate;
};
typedef struct cuvsIvfPqParams* cuvsIvfPqParams_t;

struct cuvsAceParams {
  size_t npartitions;
  size_t ef_construction;
  const char* build_dir;
  bool use_disk;
  double max_host_memory_gb;
  double max_gpu_memory_gb;
};
typedef struct cuvsAceParams* cuvsAceParams_t;

struct cuvsCagraIndexParams {
  cuvsDistanceType metric;            /* L2Expanded */
  size_t intermediate_graph_degree;   /* 128 */
  size_t graph_degree;                /* 64 */
  enum cuvsCagraGraphBuildAlgo build_algo; /* IVF_PQ */
  size_t nn_descent_niter;            /* 20 */
  cuvsCagraCompressionParams_t compression;
  void* graph_build_params;           /* cuvsIvfPqParams* / cuvsAceParams* by build_algo */
};
typedef struct cuvsCagraIndexParams* cuvsCagraIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsCagraIndexParamsCreate(cuvsCagraIndexParams_t* params);
CUVS_EXPORT cuvsError_t cuvsCagraIndexParamsDestroy(cuvsCagraIndexParams_t params);
CUVS_EXPORT cuvsError_t cuvsCagraCompressionParamsCreate(cuvsCagraCompressionParams_t* params);
CUVS_EXPORT cuvsError_t cuvsCagraCompressionParamsDestroy(cuvsCagraCompressionParams_t params);
CUVS_EXPORT cuvsError_t cuvsAceParamsCreate(cuvsAceParams_t* params);
CUVS_EXPORT cuvsError_t cuvsAceParamsDestroy(cuvsAceParams_t params);
CUVS_EXPORT cuvsError_t cuvsCagraIndexParamsFromHnswParams(cuvsCagraIndexParams_t params,
                                                           int64_t n_rows,
                                                           int64_t dim,
                                                           int M,
                                                           int ef_construction,
                                                           enum cuvsCagraHnswHeuristicType heuristic,
                                                           cuvsDistanceType metric);

struct cuvsCagraExtendParams {
  uint32_t max_chunk_size;
};
typedef struct cuvsCagraExtendParams* cuvsCagraExtendParams_t;
CUVS_EXPORT cuvsError_t cuvsCagraExtendParamsCreate(cuvsCagraExtendParams_t* params);
CUVS_EXPORT cuvsError_t cuvsCagraExtendParamsDestroy(cuvsCagraExtendParams_t params);

enum cuvsCagraSearchAlgo { SINGLE_CTA = 0, MULTI_CTA = 1, MULTI_KERNEL = 2, AUTO = 100 };
enum cuvsCagraHashMode { HASH = 0, SMALL = 1, AUTO_HASH = 100 };

struct cuvsCagraSearchParams {
  size_t max_queries;        /* 0 = auto */
  size_t itopk_size;         /* 64 */
  size_t max_iterations;     /* 0 = auto */
  enum cuvsCagraSearchAlgo algo;
  size_t team_size;          /* 0 = auto (lanes cooperating on one distance) */
  size_t search_width;       /* 1 */
  size_t min_iterations;
  size_t thread_block_size;  /* 0 = auto */
  enum cuvsCagraHashMode hashmap_mode;
  size_t hashmap_min_bitlen;
  float hashmap_max_fill_rate; /* 0.5 */
  uint32_t num_random_samplings; /* 1 */
  uint64_t rand_xor_mask;        /* 0x128394 */
  bool persistent;
  float persistent_lifetime;
  float persistent_device_usage;
};
typedef struct cuvsCagraSearchParams* cuvsCagraSearchParams_t;
CUVS_EXPORT cuvsError_t cuvsCagraSearchParamsCreate(cuvsCagraSearchParams_t* params);
CUVS_EXPORT cuvsError_t cuvsCagraSearchParamsDestroy(cuvsCagraSearchParams_t params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype;
} cuvsCagraIndex;
typedef cuvsCagraIndex* cuvsCagraIndex_t;
CUVS_EXPORT cuvsError_t cuvsCagraIndexCreate(cuvsCagraIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsCagraIndexDestroy(cuvsCagraIndex_t index);
CUVS_EXPORT cuvsError_t cuvsCagraIndexGetDims(cuvsCagraIndex_t index, int64_t* dim);
CUVS_EXPORT cuvsError_t cuvsCagraIndexGetSize(cuvsCagraIndex_t index, int64_t* size);
CUVS_EXPORT cuvsError_t cuvsCagraIndexGetGraphDegree(cuvsCagraIndex_t index, int64_t* graph_degree);
CUVS_EXPORT cuvsError_t cuvsCagraIndexGetDataset(cuvsCagraIndex_t index, DLManagedTensor* dataset);
CUVS_EXPORT cuvsError_t cuvsCagraIndexGetGraph(cuvsCagraIndex_t index, DLManagedTensor* graph);

CUVS_EXPORT cuvsError_t cuvsCagraBuild(cuvsResources_t res,
                                       cuvsCagraIndexParams_t params,
                                       DLManagedTensor* dataset,
                                       cuvsCagraIndex_t index);
CUVS_EXPORT cuvsError_t cuvsCagraExtend(cuvsResources_t res,
                                        cuvsCagraExtendParams_t params,
                                        DLManagedTensor* additional_dataset,
                                        cuvsCagraIndex_t index);
/* neighbors: uint32 (or int64) [m,k]; distances fp32 [m,k]; filter NO_FILTER or BITSET
 * (c/src/neighbors/cagra.cpp:253-262,646-690). */
CUVS_EXPORT cuvsError_t cuvsCagraSearch(cuvsResources_t res,
                                        cuvsCagraSearchParams_t params,
                                        cuvsCagraIndex_t index,
                                        DLManagedTensor* queries,
                                        DLManagedTensor* neighbors,
                                        DLManagedTensor* distances,
                                        cuvsFilter filter);
CUVS_EXPORT cuvsError_t cuvsCagraSerialize(cuvsResources_t res,
                                           const char* filename,
                                           cuvsCagraIndex_t index,
                                           bool include_dataset);
CUVS_EXPORT cuvsError_t cuvsCagraSerializeToHnswlib(cuvsResources_t res,
                                                    const char* filename,
                                                    cuvsCagraIndex_t index);
CUVS_EXPORT cuvsError_t cuvsCagraDeserialize(cuvsResources_t res, const char* filename, cuvsCagraIndex_t index);
CUVS_EXPORT cuvsError_t cuvsCagraIndexFromArgs(cuvsResources_t res,
                                               cuvsDistanceType metric,
                                               DLManagedTensor* graph,
                                               DLManagedTensor* dataset,
                                               cuvsCagraIndex_t index);
CUVS_EXPORT cuvsError_t cuvsCagraMerge(cuvsResources_t res,
                                       cuvsCagraIndexParams_t params,
                                       cuvsCagraIndex_t* indices,
                                       size_t num_indices,
                                       cuvsFilter filter,
                                       cuvsCagraIndex_t output_index);
#ifdef __cplusplus
}
#endif
