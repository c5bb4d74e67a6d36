/* Distance metric identifiers; integer values are ABI (c/include/cuvs/distance/distance.h:15-60,
 * same as cpp/include/cuvs/distance/distance.hpp:23-68). Only L2Expanded, L2SqrtExpanded,
 * CosineExpanded, L2Unexpanded, L2SqrtUnexpanded and InnerProduct are implemented by the
 * hot path; the others are declared so bindings compile and are rejected at run time. */
#pragma once
#include <cuvs/core/export.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum {
  L2Expanded          = 0,
  L2SqrtExpanded      = 1,
  CosineExpanded      = 2,
  L1                  = 3,
  L2Unexpanded        = 4,
  L2SqrtUnexpanded    = 5,
  InnerProduct        = 6,
  Linf                = 7,
  Canberra            = 8,
  LpUnexpanded        = 9,
  CorrelationExpanded = 10,
  JaccardExpanded     = 11,
  HellingerExpanded   = 12,
  Haversine           = 13,
  BrayCurtis          = 14,
  JensenShannon       = 15,
  HammingUnexpanded   = 16,
  KLDivergence        = 17,
  RusselRaoExpanded   = 18,
  DiceExpanded        = 19,
  BitwiseHamming      = 20,
  Precomputed         = 100
} cuvsDistanceType;
#ifdef __cplusplus
}
#endif
