/* Pre-filter descriptor shared by the search entry points (c/include/cuvs/neighbors/common.h:20-33).
 * addr is a DLManagedTensor* of uint32 words: BITSET = n_rows bits (1 keeps the row),
 * BITMAP = n_queries x n_rows bits (brute force only). */
#pragma once
#include <stdint.h>
#include <cuvs/core/export.h>
#ifdef __cplusplus
extern "C" {
#endif
enum cuvsFilterType { NO_FILTER = 0, BITSET = 1, BITMAP = 2 };
typedef struct {
  uintptr_t addr;
  enum cuvsFilterType type;
} cuvsFilter;
typedef enum { MERGE_STRATEGY_PHYSICAL = 0, MERGE_STRATEGY_LOGICAL = 1 } cuvsMergeStrategy;
#ifdef __cplusplus
}
#endif
