/* Symbol visibility for the C ABI (replaces c/include/cuvs/core/export.h). */
#pragma once
#define CUVS_EXPORT __attribute__((visibility("default")))
