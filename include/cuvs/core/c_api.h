/*
 * cuvs C ABI, core part — MI355X-native implementation.
 * Drop-in for c/include/cuvs/core/c_api.h of the reference (symbols, enum values and
 * handle representation identical; `cudaStream_t` becomes `hipStream_t`, same size).
 * Every entry point returns cuvsError_t with CUVS_ERROR = 0 / CUVS_SUCCESS = 1
 * (reference c_api.h:27); the failure text is kept per thread (c/src/core/c_api.cpp:187-192).
 */
#pragma once

#include <dlpack/dlpack.h>
#include <stdbool.h>
#include <stdint.h>
#include <stddef.h>

#include <cuvs/core/export.h>

#ifdef __cplusplus
extern "C" {
#endif

/* An opaque HIP stream pointer. We do not include <hip/hip_runtime.h> here so that plain C,
 * cgo, bindgen and Panama consumers need no ROCm headers (the reference needs the same trick
 * for Rust: rust/cuvs-sys/bindgen-stubs/cuda_runtime.h:17-18). */
typedef struct ihipStream_t* cuvsStream_t;
#ifndef CUVS_AMD_NO_CUDA_ALIASES
typedef cuvsStream_t cudaStream_t; /* spelling used by the reference signatures */
/* hipDataType values equal cudaDataType_t values for the dtypes the ABI carries:
 * R_32F = 0, R_16F = 2, R_8I = 3, R_8U = 8 (/opt/rocm/include/hip/library_types.h). */
typedef enum {
  CUDA_R_32F = 0,
  CUDA_R_16F = 2,
  CUDA_R_8I  = 3,
  CUDA_R_8U  = 8
} cudaDataType_t;
#endif

typedef enum { CUVS_ERROR = 0, CUVS_SUCCESS = 1 } cuvsError_t;

/* c_api.h:33-41 */
CUVS_EXPORT const char* cuvsGetLastErrorText();
CUVS_EXPORT void cuvsSetLastErrorText(const char* error);

typedef enum {
  CUVS_LOG_LEVEL_TRACE    = 0,
  CUVS_LOG_LEVEL_DEBUG    = 1,
  CUVS_LOG_LEVEL_INFO     = 2,
  CUVS_LOG_LEVEL_WARN     = 3,
  CUVS_LOG_LEVEL_ERROR    = 4,
  CUVS_LOG_LEVEL_CRITICAL = 5,
  CUVS_LOG_LEVEL_OFF      = 6
} cuvsLogLevel_t;
CUVS_EXPORT cuvsLogLevel_t cuvsGetLogLevel();
CUVS_EXPORT void cuvsSetLogLevel(cuvsLogLevel_t);

/* c_api.h:80 — a resources handle is an integer holding a pointer */
typedef uintptr_t cuvsResources_t;

CUVS_EXPORT cuvsError_t cuvsResourcesCreate(cuvsResources_t* res);          /* c_api.h:88  */
CUVS_EXPORT cuvsError_t cuvsResourcesDestroy(cuvsResources_t res);          /* c_api.h:96  */
CUVS_EXPORT cuvsError_t cuvsStreamSet(cuvsResources_t res, cudaStream_t stream); /* :106 */
CUVS_EXPORT cuvsError_t cuvsStreamGet(cuvsResources_t res, cudaStream_t* stream);
CUVS_EXPORT cuvsError_t cuvsStreamSync(cuvsResources_t res);
CUVS_EXPORT cuvsError_t cuvsDeviceIdGet(cuvsResources_t res, int* device_id);

/* single-node multi-GPU resources (c_api.h:140-175) */
CUVS_EXPORT cuvsError_t cuvsMultiGpuResourcesCreate(cuvsResources_t* res);
CUVS_EXPORT cuvsError_t cuvsMultiGpuResourcesCreateWithDeviceIds(cuvsResources_t* res,
                                                                 DLManagedTensor* device_ids);
CUVS_EXPORT cuvsError_t cuvsMultiGpuResourcesDestroy(cuvsResources_t res);
CUVS_EXPORT cuvsError_t cuvsMultiGpuResourcesSetMemoryPool(cuvsResources_t res,
                                                           int percent_of_free_memory);

/* device / pinned allocation through the handle (reference: RMM; here: HIP stream-ordered pool) */
CUVS_EXPORT cuvsError_t cuvsRMMAlloc(cuvsResources_t res, void** ptr, size_t bytes);
CUVS_EXPORT cuvsError_t cuvsRMMFree(cuvsResources_t res, void* ptr, size_t bytes);
CUVS_EXPORT cuvsError_t cuvsRMMPoolMemoryResourceEnable(int initial_pool_size_percent,
                                                        int max_pool_size_percent,
                                                        bool managed);
CUVS_EXPORT cuvsError_t cuvsRMMMemoryResourceReset();
CUVS_EXPORT cuvsError_t cuvsRMMHostAlloc(void** ptr, size_t bytes);
CUVS_EXPORT cuvsError_t cuvsRMMHostFree(void* ptr, size_t bytes);

CUVS_EXPORT cuvsError_t cuvsVersionGet(uint16_t* major, uint16_t* minor, uint16_t* patch);

/* strided 2-D copy between any two device-accessible tensors; row slice view (c_api.h:250-275) */
CUVS_EXPORT cuvsError_t cuvsMatrixCopy(cuvsResources_t res, DLManagedTensor* src, DLManagedTensor* dst);
CUVS_EXPORT cuvsError_t cuvsMatrixSliceRows(
  cuvsResources_t res, DLManagedTensor* src, int64_t start, int64_t end, DLManagedTensor* dst);

#ifdef __cplusplus
}
#endif
