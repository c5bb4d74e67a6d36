/* Optional components of this build — the header the reference generates at configure time
 * (c/CMakeLists.txt:136-152). The cuvsMultiGpu* index wrappers are built (cuvs_amd/csrc/mg.hip); the hnswlib
 * bridge (CUVS_BUILD_CAGRA_HNSWLIB, <cuvs/neighbors/hnsw.h>) is not. */
#pragma once
#ifndef CUVS_BUILD_MG_ALGOS
#define CUVS_BUILD_MG_ALGOS
#endif
