/*
 * Single-process multi-GPU search: shared enums — drop-in for c/include/cuvs/neighbors/mg_common.h.
 * Implemented by cuvs_amd/csrc/mg.hip (one host thread per GPU of the cuvsMultiGpuResources handle).
 */
#pragma once
#include <stdint.h>

#include <cuvs/core/export.h>

#ifdef __cplusplus
extern "C" {
#endif

/* REPLICATED: every GPU holds the whole index and takes a share of the queries.
 * SHARDED: GPU r indexes rows [r * ceil(n / R), ...) and every GPU sees every query (snmg.cuh:128-166). */
typedef enum { CUVS_NEIGHBORS_MG_REPLICATED = 0, CUVS_NEIGHBORS_MG_SHARDED = 1 } cuvsMultiGpuDistributionMode;

/* REPLICATED search: batches dealt to the GPUs in turn, or the whole call on the next GPU (snmg.cuh:596-660). */
typedef enum {
  CUVS_NEIGHBORS_MG_LOAD_BALANCER = 0,
  CUVS_NEIGHBORS_MG_ROUND_ROBIN   = 1
} cuvsMultiGpuReplicatedSearchMode;

/* SHARDED search: how the reference moves the partial results between GPUs before merging them. Both values are
 * accepted; results here land on the host, so the per-GPU lists are merged there in one pass either way. */
typedef enum {
  CUVS_NEIGHBORS_MG_MERGE_ON_ROOT_RANK = 0,
  CUVS_NEIGHBORS_MG_TREE_MERGE         = 1
} cuvsMultiGpuShardedMergeMode;

#ifdef __cplusplus
}
#endif
