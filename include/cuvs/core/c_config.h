/* Optional components of this build — the header the reference generates at configure time
 * (c/CMakeLists.txt:136-152). Neither the hnswlib bridge (CUVS_BUILD_CAGRA_HNSWLIB) nor the cuvsMultiGpu*
 * index wrappers (CUVS_BUILD_MG_ALGOS) are built, so neither macro is defined. */
#pragma once
