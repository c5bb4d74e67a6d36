/*
 * NN-descent all-neighbours graph construction - drop-in for c/include/cuvs/neighbors/nn_descent.h
 * (struct layouts and entry points of the reference; implementation: cuvs_amd/csrc/nn_descent.hip).
 * CAGRA's default graph builder, also usable on its own.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/core/export.h>
#include <cuvs/distance/distance.h>
#include <dlpack/dlpack.h>
#include <stdbool.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic of the distance evaluation; this library always accumulates in fp32 */
typedef enum { NND_DIST_COMP_AUTO = 0, NND_DIST_COMP_FP32 = 1, NND_DIST_COMP_FP16 = 2 } cuvsNNDescentDistCompDtype;

struct cuvsNNDescentIndexParams {
  cuvsDistanceType metric;          /* L2Expanded / L2SqrtExpanded / InnerProduct / CosineExpanded */
  float metric_arg;
  size_t graph_degree;              /* 64: columns of the graph that is returned */
  size_t intermediate_graph_degree; /* 128: length of the lists the descent works on */
  size_t max_iterations;            /* 20 */
  float termination_threshold;      /* 1e-4: stop when fewer than this share of the list slots changed */
  bool return_distances;            /* true */
  cuvsNNDescentDistCompDtype dist_comp_dtype;
};
typedef struct cuvsNNDescentIndexParams* cuvsNNDescentIndexParams_t;
CUVS_EXPORT cuvsError_t cuvsNNDescentIndexParamsCreate(cuvsNNDescentIndexParams_t* index_params);
CUVS_EXPORT cuvsError_t cuvsNNDescentIndexParamsDestroy(cuvsNNDescentIndexParams_t index_params);

typedef struct {
  uintptr_t addr;
  DLDataType dtype; /* graph element type: uint32 */
} cuvsNNDescentIndex;
typedef cuvsNNDescentIndex* cuvsNNDescentIndex_t;
CUVS_EXPORT cuvsError_t cuvsNNDescentIndexCreate(cuvsNNDescentIndex_t* index);
CUVS_EXPORT cuvsError_t cuvsNNDescentIndexDestroy(cuvsNNDescentIndex_t index);

/* dataset [n, dim] fp32 / fp16 / int8 / uint8, host or device. `graph` (optional, uint32 [n, graph_degree], host or
 * device) also receives the result. */
CUVS_EXPORT cuvsError_t cuvsNNDescentBuild(cuvsResources_t res, cuvsNNDescentIndexParams_t index_params,
                                           DLManagedTensor* dataset, DLManagedTensor* graph,
                                           cuvsNNDescentIndex_t index);
/* copy out uint32 [n, graph_degree] / fp32 [n, graph_degree] (host or device destination) */
CUVS_EXPORT cuvsError_t cuvsNNDescentIndexGetGraph(cuvsResources_t res, cuvsNNDescentIndex_t index,
                                                   DLManagedTensor* graph);
CUVS_EXPORT cuvsError_t cuvsNNDescentIndexGetDistances(cuvsResources_t res, cuvsNNDescentIndex_t index,
                                                       DLManagedTensor* distances);
#ifdef __cplusplus
}
#endif
