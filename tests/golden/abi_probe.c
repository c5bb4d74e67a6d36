/* Prints sizes/offsets of the C-ABI structs callers mutate directly. Compiled twice: against the reference's
 * c/include (tests/golden/gen_abi_layout.sh -> abi_layout.txt, committed) and against this repo's include/. */
#include <stddef.h>
#include <stdio.h>
#include <cuvs/neighbors/brute_force.h>
#include <cuvs/neighbors/ivf_flat.h>
#include <cuvs/neighbors/ivf_pq.h>
#include <cuvs/neighbors/cagra.h>
#include <cuvs/neighbors/nn_descent.h>
#include <cuvs/cluster/kmeans.h>
#include <cuvs/neighbors/mg_ivf_flat.h>
#include <cuvs/neighbors/mg_ivf_pq.h>
#include <cuvs/neighbors/mg_cagra.h>
#define SZ(T) printf("sizeof " #T " %zu\n", sizeof(T))
#define OFF(T, F) printf("offsetof " #T "." #F " %zu\n", offsetof(T, F))
int main(void)
{
  SZ(struct cuvsIvfFlatIndexParams);
  OFF(struct cuvsIvfFlatIndexParams, metric); OFF(struct cuvsIvfFlatIndexParams, metric_arg);
  OFF(struct cuvsIvfFlatIndexParams, add_data_on_build); OFF(struct cuvsIvfFlatIndexParams, n_lists);
  OFF(struct cuvsIvfFlatIndexParams, kmeans_n_iters); OFF(struct cuvsIvfFlatIndexParams, kmeans_trainset_fraction);
  OFF(struct cuvsIvfFlatIndexParams, adaptive_centers); OFF(struct cuvsIvfFlatIndexParams, conservative_memory_allocation);
  SZ(struct cuvsIvfFlatSearchParams);
  SZ(struct cuvsIvfPqIndexParams);
  OFF(struct cuvsIvfPqIndexParams, n_lists); OFF(struct cuvsIvfPqIndexParams, kmeans_trainset_fraction);
  OFF(struct cuvsIvfPqIndexParams, pq_bits); OFF(struct cuvsIvfPqIndexParams, pq_dim);
  OFF(struct cuvsIvfPqIndexParams, codebook_kind); OFF(struct cuvsIvfPqIndexParams, force_random_rotation);
  OFF(struct cuvsIvfPqIndexParams, conservative_memory_allocation);
  OFF(struct cuvsIvfPqIndexParams, max_train_points_per_pq_code); OFF(struct cuvsIvfPqIndexParams, codes_layout);
  SZ(struct cuvsIvfPqSearchParams);
  OFF(struct cuvsIvfPqSearchParams, n_probes); OFF(struct cuvsIvfPqSearchParams, lut_dtype);
  OFF(struct cuvsIvfPqSearchParams, internal_distance_dtype); OFF(struct cuvsIvfPqSearchParams, coarse_search_dtype);
  OFF(struct cuvsIvfPqSearchParams, max_internal_batch_size); OFF(struct cuvsIvfPqSearchParams, preferred_shmem_carveout);
  SZ(struct cuvsCagraIndexParams);
  OFF(struct cuvsCagraIndexParams, intermediate_graph_degree); OFF(struct cuvsCagraIndexParams, graph_degree);
  OFF(struct cuvsCagraIndexParams, build_algo); OFF(struct cuvsCagraIndexParams, nn_descent_niter);
  OFF(struct cuvsCagraIndexParams, compression); OFF(struct cuvsCagraIndexParams, graph_build_params);
  SZ(struct cuvsCagraSearchParams);
  OFF(struct cuvsCagraSearchParams, itopk_size); OFF(struct cuvsCagraSearchParams, max_iterations);
  OFF(struct cuvsCagraSearchParams, algo); OFF(struct cuvsCagraSearchParams, team_size);
  OFF(struct cuvsCagraSearchParams, search_width); OFF(struct cuvsCagraSearchParams, hashmap_mode);
  OFF(struct cuvsCagraSearchParams, hashmap_max_fill_rate); OFF(struct cuvsCagraSearchParams, num_random_samplings);
  OFF(struct cuvsCagraSearchParams, rand_xor_mask); OFF(struct cuvsCagraSearchParams, persistent);
  OFF(struct cuvsCagraSearchParams, persistent_lifetime); OFF(struct cuvsCagraSearchParams, persistent_device_usage);
  SZ(struct cuvsCagraCompressionParams); SZ(struct cuvsIvfPqParams); SZ(struct cuvsAceParams);
  SZ(struct cuvsNNDescentIndexParams); SZ(cuvsNNDescentIndex);
  OFF(struct cuvsNNDescentIndexParams, metric); OFF(struct cuvsNNDescentIndexParams, graph_degree);
  OFF(struct cuvsNNDescentIndexParams, intermediate_graph_degree); OFF(struct cuvsNNDescentIndexParams, max_iterations);
  OFF(struct cuvsNNDescentIndexParams, termination_threshold); OFF(struct cuvsNNDescentIndexParams, return_distances);
  OFF(struct cuvsNNDescentIndexParams, dist_comp_dtype);
  SZ(struct cuvsCagraExtendParams);
  SZ(cuvsFilter); SZ(cuvsBruteForceIndex); SZ(cuvsIvfFlatIndex); SZ(cuvsIvfPqIndex); SZ(cuvsCagraIndex);
  SZ(cuvsResources_t);
  printf("enum CUVS_ERROR %d CUVS_SUCCESS %d\n", (int)CUVS_ERROR, (int)CUVS_SUCCESS);
  printf("enum L2Expanded %d L2SqrtExpanded %d CosineExpanded %d L2Unexpanded %d InnerProduct %d\n", (int)L2Expanded,
         (int)L2SqrtExpanded, (int)CosineExpanded, (int)L2Unexpanded, (int)InnerProduct);
  printf("enum NO_FILTER %d BITSET %d BITMAP %d\n", (int)NO_FILTER, (int)BITSET, (int)BITMAP);
  printf("enum IVF_PQ %d NN_DESCENT %d SINGLE_CTA %d MULTI_CTA %d AUTO %d\n", (int)IVF_PQ, (int)NN_DESCENT,
         (int)SINGLE_CTA, (int)MULTI_CTA, (int)AUTO);
  printf("enum CUDA_R_32F %d CUDA_R_16F %d CUDA_R_8I %d CUDA_R_8U %d\n", (int)CUDA_R_32F, (int)CUDA_R_16F,
         (int)CUDA_R_8I, (int)CUDA_R_8U);
  SZ(struct cuvsKMeansParams); SZ(struct cuvsKMeansParams_v2);
  OFF(struct cuvsKMeansParams, metric); OFF(struct cuvsKMeansParams, n_clusters); OFF(struct cuvsKMeansParams, init);
  OFF(struct cuvsKMeansParams, max_iter); OFF(struct cuvsKMeansParams, tol); OFF(struct cuvsKMeansParams, n_init);
  OFF(struct cuvsKMeansParams, oversampling_factor); OFF(struct cuvsKMeansParams, batch_samples);
  OFF(struct cuvsKMeansParams, batch_centroids); OFF(struct cuvsKMeansParams, inertia_check);
  OFF(struct cuvsKMeansParams, hierarchical); OFF(struct cuvsKMeansParams, hierarchical_n_iters);
  OFF(struct cuvsKMeansParams, streaming_batch_size); OFF(struct cuvsKMeansParams, init_size);
  OFF(struct cuvsKMeansParams_v2, batch_centroids); OFF(struct cuvsKMeansParams_v2, hierarchical);
  OFF(struct cuvsKMeansParams_v2, hierarchical_n_iters); OFF(struct cuvsKMeansParams_v2, streaming_batch_size);
  OFF(struct cuvsKMeansParams_v2, init_size);
  printf("enum KMeansPlusPlus %d Random %d Array %d KMEANS %d KMEANS_BALANCED %d\n", (int)KMeansPlusPlus, (int)Random,
         (int)Array, (int)CUVS_KMEANS_TYPE_KMEANS, (int)CUVS_KMEANS_TYPE_KMEANS_BALANCED);
  SZ(struct cuvsMultiGpuIvfPqIndexParams); OFF(struct cuvsMultiGpuIvfPqIndexParams, base_params);
  OFF(struct cuvsMultiGpuIvfPqIndexParams, mode);
  SZ(struct cuvsMultiGpuIvfPqSearchParams); OFF(struct cuvsMultiGpuIvfPqSearchParams, search_mode);
  OFF(struct cuvsMultiGpuIvfPqSearchParams, merge_mode); OFF(struct cuvsMultiGpuIvfPqSearchParams, n_rows_per_batch);
  SZ(struct cuvsMultiGpuIvfFlatIndexParams); SZ(struct cuvsMultiGpuIvfFlatSearchParams);
  OFF(struct cuvsMultiGpuIvfFlatSearchParams, n_rows_per_batch);
  SZ(struct cuvsMultiGpuCagraIndexParams); SZ(struct cuvsMultiGpuCagraSearchParams);
  OFF(struct cuvsMultiGpuCagraSearchParams, n_rows_per_batch);
  SZ(cuvsMultiGpuIvfPqIndex); SZ(cuvsMultiGpuIvfFlatIndex); SZ(cuvsMultiGpuCagraIndex);
  printf("enum REPLICATED %d SHARDED %d LOAD_BALANCER %d ROUND_ROBIN %d ON_ROOT %d TREE %d\n",
         (int)CUVS_NEIGHBORS_MG_REPLICATED, (int)CUVS_NEIGHBORS_MG_SHARDED, (int)CUVS_NEIGHBORS_MG_LOAD_BALANCER,
         (int)CUVS_NEIGHBORS_MG_ROUND_ROBIN, (int)CUVS_NEIGHBORS_MG_MERGE_ON_ROOT_RANK, (int)CUVS_NEIGHBORS_MG_TREE_MERGE);
  return 0;
}
