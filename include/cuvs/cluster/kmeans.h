/*
 * k-means through the C ABI — drop-in for c/include/cuvs/cluster/kmeans.h (struct layouts :41-202, entry points
 * :216-414; wrapper c/src/cluster/kmeans.cpp).
 *
 * `hierarchical = true` runs the balanced hierarchical k-means that trains the IVF coarse
 * quantizers (cuvs_amd/csrc/kmeans_balanced.hip, bit-exact against oracle/oracle.c). `hierarchical = false` runs
 * Lloyd iterations with the reference's stopping rule (cpp/src/cluster/detail/kmeans.cuh:813-925,
 * kmeans_common.cuh:629-648); its seeding draws from this library's own counter-based generator, so only
 * `init = Array` is reproducible against the reference (c/tests/cluster/kmeans_c.cu:24-48 is the known answer).
 * float32 only; float64 input fails with an error text (no MFMA f64 path is built).
 */
#pragma once
#include <cuvs/core/c_api.h>
#include <cuvs/distance/distance.h>
#include <dlpack/dlpack.h>
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { KMeansPlusPlus = 0, Random = 1, Array = 2 } cuvsKMeansInitMethod;

/* Field order and sizes are ABI: callers set members directly (kmeans.h:41-128). */
struct cuvsKMeansParams {
  cuvsDistanceType metric;          /* L2Expanded (default), L2SqrtExpanded, L2Unexpanded, L2SqrtUnexpanded */
  int n_clusters;                   /* default 8 */
  cuvsKMeansInitMethod init;        /* default KMeansPlusPlus */
  int max_iter;                     /* default 300 */
  double tol;                       /* default 1e-4; relative inertia change / squared centroid shift */
  int n_init;                       /* default 1; best inertia of n_init seedings wins */
  double oversampling_factor;       /* accepted, unused: the seeding here is sequential k-means++ */
  int batch_samples;                /* accepted, unused: the E-step never materialises the distance tile */
  int batch_centroids;              /* accepted, unused */
  bool inertia_check;               /* deprecated in the reference, ignored */
  bool hierarchical;                /* true: balanced hierarchical k-means (row a8) */
  int hierarchical_n_iters;         /* default 20 */
  int64_t streaming_batch_size;     /* accepted, unused: host data is copied to HBM once (288 GB) */
  int64_t init_size;                /* accepted, unused */
};

/* The layout the reference switches to in its next ABI major (kmeans.h:130-202): no inertia_check. */
struct cuvsKMeansParams_v2 {
  cuvsDistanceType metric;
  int n_clusters;
  cuvsKMeansInitMethod init;
  int max_iter;
  double tol;
  int n_init;
  double oversampling_factor;
  int batch_samples;
  int batch_centroids;
  bool hierarchical;
  int hierarchical_n_iters;
  int64_t streaming_batch_size;
  int64_t init_size;
};

typedef struct cuvsKMeansParams* cuvsKMeansParams_t;
typedef struct cuvsKMeansParams_v2* cuvsKMeansParams_v2_t;
typedef enum { CUVS_KMEANS_TYPE_KMEANS = 0, CUVS_KMEANS_TYPE_KMEANS_BALANCED = 1 } cuvsKMeansType;

CUVS_EXPORT cuvsError_t cuvsKMeansParamsCreate(cuvsKMeansParams_t* params);
CUVS_EXPORT cuvsError_t cuvsKMeansParamsDestroy(cuvsKMeansParams_t params);
CUVS_EXPORT cuvsError_t cuvsKMeansParamsCreate_v2(cuvsKMeansParams_v2_t* params);
CUVS_EXPORT cuvsError_t cuvsKMeansParamsDestroy_v2(cuvsKMeansParams_v2_t params);

/* X [n_samples, n_features] fp32 row-major, device (or host unless hierarchical); sample_weight NULL or fp32
 * [n_samples] (not with hierarchical); centroids fp32 [n_clusters, n_features] on device: read when init = Array,
 * always written. *inertia = sum of squared distances to the closest centroid; *n_iter = iterations run. */
CUVS_EXPORT cuvsError_t cuvsKMeansFit(cuvsResources_t res, cuvsKMeansParams_t params, DLManagedTensor* X,
                                      DLManagedTensor* sample_weight, DLManagedTensor* centroids, double* inertia,
                                      int* n_iter);
CUVS_EXPORT cuvsError_t cuvsKMeansFit_v2(cuvsResources_t res, cuvsKMeansParams_v2_t params, DLManagedTensor* X,
                                         DLManagedTensor* sample_weight, DLManagedTensor* centroids,
                                         double* inertia, int* n_iter);

/* labels int32 [n_samples] on device; all tensors on device. hierarchical: *inertia = 0 like the reference. */
CUVS_EXPORT cuvsError_t cuvsKMeansPredict(cuvsResources_t res, cuvsKMeansParams_t params, DLManagedTensor* X,
                                          DLManagedTensor* sample_weight, DLManagedTensor* centroids,
                                          DLManagedTensor* labels, bool normalize_weight, double* inertia);
CUVS_EXPORT cuvsError_t cuvsKMeansPredict_v2(cuvsResources_t res, cuvsKMeansParams_v2_t params, DLManagedTensor* X,
                                             DLManagedTensor* sample_weight, DLManagedTensor* centroids,
                                             DLManagedTensor* labels, bool normalize_weight, double* inertia);

/* *cost = sum over rows of the squared distance to the closest centroid. */
CUVS_EXPORT cuvsError_t cuvsKMeansClusterCost(cuvsResources_t res, DLManagedTensor* X, DLManagedTensor* centroids,
                                              double* cost);

#ifdef __cplusplus
}
#endif
