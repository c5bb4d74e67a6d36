/* Umbrella include for the hot-path C ABI (reference: c/include/cuvs/core/all.h). */
#pragma once
#include <cuvs/core/c_config.h>
#include <cuvs/core/c_api.h>
#include <cuvs/cluster/kmeans.h>
#include <cuvs/distance/distance.h>
#include <cuvs/distance/pairwise_distance.h>
#include <cuvs/neighbors/common.h>
#include <cuvs/neighbors/brute_force.h>
#include <cuvs/neighbors/ivf_flat.h>
#include <cuvs/neighbors/ivf_pq.h>
#include <cuvs/neighbors/cagra.h>
#include <cuvs/neighbors/nn_descent.h>
#include <cuvs/neighbors/refine.h>
#ifdef CUVS_BUILD_MG_ALGOS
#include <cuvs/neighbors/mg_common.h>
#include <cuvs/neighbors/mg_ivf_flat.h>
#include <cuvs/neighbors/mg_ivf_pq.h>
#include <cuvs/neighbors/mg_cagra.h>
#endif
