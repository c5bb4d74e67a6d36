/* List-sharded multi-GPU IVF-PQ search: one process per GPU, RCCL all-gather of the per-rank top-k over xGMI.
 *
 * What it replaces in the reference: cuvs::neighbors::mg sharded search - rank-local search, then the partial top-k
 * lists are exchanged with ncclSend/ncclRecv and merged (cpp/src/neighbors/mg/snmg.cuh:248-375, NCCL call sites
 * :298-340, merge knn_merge_parts.cuh:27-103). The reference shards by ROW RANGE inside one process
 * (c/include/cuvs/neighbors/mg_ivf_pq.h:152-190; built here too, mg.hip); this header adds the split the target
 * asks for - shard by IVF LIST with one global coarse quantizer - for one process per GPU:
 *   - every rank trains / loads the same model (centres, rotation, codebooks); list L belongs to rank L % world;
 *   - cuvsIvfPqExtend on a sharded index keeps only the rows that fall into the rank's lists;
 *   - cuvsIvfPqSearch on a sharded index ranks ALL centres, takes the global n_probes nearest and scans the ones it
 *     owns: over all ranks exactly the (query, probe) pairs a single GPU would scan, each once;
 *   - cuvsAmdShardAllGatherTopK: ONE ncclAllGather of the [n_queries, k] (distance, id) blocks (12 B per candidate) and
 *     an R-way merge on every rank; the result equals the single-GPU search of the whole index.
 * The communicator wraps an RCCL ncclComm_t. RCCL is loaded at run time (dlopen "librccl.so.1": the copy a host
 * framework has already loaded is reused), so the library has no link-time dependency on it.
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/core/export.h>
#include <cuvs/neighbors/ivf_pq.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUVS_AMD_SHARD_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */

typedef struct cuvsAmdShardComm* cuvsAmdShardComm_t;

/* rank 0 creates the rendezvous id and hands its 128 bytes to the other ranks by any means (file, MPI, torchrun store) */
CUVS_EXPORT cuvsError_t cuvsAmdShardCommGetUniqueId(char id[CUVS_AMD_SHARD_ID_BYTES]);
/* The same rendezvous for the HOST-STAGED transport (cuvs_amd/csrc/shm_transport.hpp): the id names a file (directory
 * CUVS_AMD_SHM_DIR, default /dev/shm) that the ranks - processes of ONE host - map; every collective then travels
 * device -> mapped file -> device. For ranks that share a device (RCCL refuses two ranks on one GPU: this is how a
 * one-GPU box runs the world > 1 code paths, tests/test_list_shard_world2_gpu.py) and for hosts without RCCL. Same
 * semantics and results as the RCCL transport, host-synchronous; a rank that waits longer than CUVS_AMD_SHM_TIMEOUT_S
 * (default 120) for its peers fails with an error instead of hanging. */
CUVS_EXPORT cuvsError_t cuvsAmdShardCommGetUniqueIdHostStaged(char id[CUVS_AMD_SHARD_ID_BYTES]);
/* collective: every rank calls it with the same id; binds the communicator to the device of `res`. The transport
 * (RCCL or host-staged) is the one the id was made for. */
CUVS_EXPORT cuvsError_t cuvsAmdShardCommCreate(cuvsResources_t res, const char id[CUVS_AMD_SHARD_ID_BYTES], int rank, int world,
                                   cuvsAmdShardComm_t* comm);
CUVS_EXPORT cuvsError_t cuvsAmdShardCommDestroy(cuvsAmdShardComm_t comm);
CUVS_EXPORT cuvsError_t cuvsAmdShardCommRank(cuvsAmdShardComm_t comm, int* rank, int* world);

/* Marks a trained, still empty index (add_data_on_build = false, or a deserialized model) as this rank's shard:
 * list L is owned by rank L % world. Call before cuvsIvfPqExtend. */
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqSetListShard(cuvsIvfPqIndex_t index, int rank, int world);

/* Lists dealt by size instead of L % world. cuvsAmdShardDealLists: greedy longest-processing-time dealing of n_lists
 * weights (rows per list) to `world` ranks - deterministic, so every rank derives the same table from the same weights.
 * cuvsAmdIvfPqListHistogram: adds, for every row of `rows` (device, [n, dim], the index dtype), one to counts[list of
 * the row] (host array of n_lists entries) - a rank calls it on ITS slice of the corpus, the launcher sums the arrays
 * over the ranks. cuvsAmdIvfPqSetListOwners: like cuvsAmdIvfPqSetListShard with an explicit owner per list (host
 * array); every rank must pass the same table. */
CUVS_EXPORT cuvsError_t cuvsAmdShardDealLists(const uint64_t* weights, uint32_t n_lists, int world, int32_t* owners);
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqListHistogram(cuvsResources_t res, cuvsIvfPqIndex_t index, DLManagedTensor* rows,
                                                  uint64_t* counts);
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqSetListOwners(cuvsIvfPqIndex_t index, const int32_t* owners, uint32_t n_lists, int rank,
                                                  int world);
/* The list of every row of `rows` (device, [n, dim], the index dtype) -> labels (device, uint32 [n]): what a rank needs to
 * keep the raw rows of ITS lists next to their codes (shard-local refinement: every rank re-ranks its own candidates
 * exactly before the all-gather, refine_ratio of the reference's bench grids; refine_device.cuh). */
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqRowLabels(cuvsResources_t res, cuvsIvfPqIndex_t index, DLManagedTensor* rows, uint32_t* labels);

/* Optional: gives the shard's searches access to the communicator. cuvsIvfPqSearch then all-reduces (min) the per-query
 * k-th bounds between its two scan phases - one ncclAllReduce of n_queries uint32 per batch - so that every rank prunes
 * with the bound of the query's globally nearest probe, whoever owns it. Results do not depend on it (bounds only
 * prune); without it a rank that owns none of a query's near lists scans its probes with a cold bound. Collective:
 * every rank must make the same searches. comm = NULL detaches. */
CUVS_EXPORT cuvsError_t cuvsAmdIvfPqSetShardComm(cuvsIvfPqIndex_t index, cuvsAmdShardComm_t comm);

/* Row-range shards (the reference's SHARDED mode, cpp/src/neighbors/mg/snmg.cuh:128-166 build, :248-375 search, any
 * index type: IVF-Flat, IVF-PQ, CAGRA, brute force): every rank builds a complete index over its rows, searches all
 * queries, translates its local row ids to global ones ON THE DEVICE - cuvsAmdShardTranslateIds: local_ids [n] uint32
 * (id_bits 32, the CAGRA output) or int64 (id_bits 64) -> global_ids [n] int64 = id + row_offset, empty slots ->
 * INT64_MAX (snmg.cuh:420-429) - and the [n_queries, k] blocks go through cuvsAmdShardAllGatherTopK: one ncclAllGather
 * instead of the reference's ncclSend/ncclRecv fan-in (:298-340) or merge tree (:439-475). */
CUVS_EXPORT cuvsError_t cuvsAmdShardTranslateIds(cuvsResources_t res, const void* local_ids, int id_bits, int64_t n,
                                                 int64_t row_offset, int64_t* global_ids);

/* Collective on the stream of `res`. local_distances [n_queries, k] float32 and local_neighbors [n_queries, k] int64
 * (device; the output of the rank's cuvsIvfPqSearch, invalid slots = FLT_MAX / INT64_MAX as the reference pads them) ->
 * distances / neighbors [n_queries, k] (device): the k best of the world * k candidates of every query, ordered by
 * (distance, rank, position); select_min = 0 for similarity metrics (inner product: larger is better). */
CUVS_EXPORT cuvsError_t cuvsAmdShardAllGatherTopK(cuvsResources_t res, cuvsAmdShardComm_t comm, const float* local_distances,
                                      const int64_t* local_neighbors, int64_t n_queries, int k, int select_min,
                                      float* distances, int64_t* neighbors);

#ifdef __cplusplus
}
#endif
