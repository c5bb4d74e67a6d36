// Shared host-side plumbing for the MI355X hot path: error convention, resources handle,
// stream-ordered device buffers and DLPack checks.
//
// Replaces what the reference gets from RAFT/RMM (raft::resources, rmm::device_uvector,
// RAFT_EXPECTS) and from c/src/core/{exceptions.hpp,detail/interop.hpp}; none of that code is
// on disk, so this is written for HIP directly: one hipStream per handle, hipMallocAsync pool.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <dlpack/dlpack.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <chrono>
#include <vector>

namespace cuvs_amd {

// ---------------------------------------------------------------- errors
struct error : public std::runtime_error {
  explicit error(const std::string& s) : std::runtime_error(s) {}
};

[[noreturn]] inline void fail(const char* file, int line, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  char out[1200];
  snprintf(out, sizeof(out), "%s (%s:%d)", buf, file, line);
  throw error(out);
}

#define CUVS_EXPECTS(cond, ...)                                              \
  do {                                                                       \
    if (!(cond)) { ::cuvs_amd::fail(__FILE__, __LINE__, __VA_ARGS__); }      \
  } while (0)
#define CUVS_FAIL(...) ::cuvs_amd::fail(__FILE__, __LINE__, __VA_ARGS__)
#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      ::cuvs_amd::fail(__FILE__, __LINE__, "HIP error %d (%s) in %s", (int)e__,              \
                       hipGetErrorString(e__), #expr);                                       \
    }                                                                                        \
  } while (0)

// thread-local last-error text (reference: c/src/core/c_api.cpp:187-192)
std::string& last_error_text();

// body wrapper used by every extern "C" entry point (reference: c/src/core/exceptions.hpp:16-32)
template <typename Fn>
int translate_exceptions(Fn&& fn)
{
  try {
    fn();
    return 1;  // CUVS_SUCCESS
  } catch (const std::exception& e) {
    last_error_text() = e.what();
    return 0;  // CUVS_ERROR
  } catch (...) {
    last_error_text() = "unknown exception";
    return 0;
  }
}

// ---------------------------------------------------------------- resources
// Test / ablation switches (CUVS_AMD_* environment variables), all behind ONE gate: they are only looked at when
// CUVS_AMD_DEBUG_SWITCHES=1 is set (core.hip). They are read ONCE, when the handle is created
// (cuvsResourcesCreate -> load_tuning_from_env), never inside a search: a drop-in library must not call getenv on its
// hot path (not thread-safe against setenv) nor change behaviour under a running caller.
struct tuning {
  int pq_head_probes    = -1;  // CUVS_AMD_PQ_HEAD_PROBES: probes per query in the cold-bounds phase (-1: default rule)
  int pq_scan2          = 1;   // CUVS_AMD_PQ_SCAN2=0: tail phase through pq_scan_kernel (comparator in the tests)
  int pq_scan3          = 1;   // CUVS_AMD_PQ_SCAN3=0: tail phase without the matrix-core filter (comparator in the tests)
  int coarse_lowp       = 1;   // CUVS_AMD_COARSE_LOWP=0: reduced-precision coarse search on the fp32 matrix cores (round 1-3; comparator)
  int pq_head_rows      = -1;  // CUVS_AMD_PQ_HEAD_ROWS: rows of a query's nearest list the head phase scores exactly for its bound (the rest of
                               // that list goes through the filter like any other probe); 0: the whole list (rounds 3-4); -1: default rule
  int pq_overlap        = 1;   // CUVS_AMD_PQ_OVERLAP=0: the IVF-PQ batch on one stream (rounds 1-4; comparator of the two-stream schedule)
  int coarse_grouped    = 1;   // CUVS_AMD_COARSE_GROUPED=0: coarse search through the plain distance matrix + select_k (rounds 1-4; comparator)
  int pq_filter4        = 1;   // CUVS_AMD_PQ_FILTER4=0: the matrix-core filter of round 3 (two waves per SIMD, 64-query units; comparator)
  int flat_scan3        = 1;   // CUVS_AMD_FLAT_SCAN3=0: IVF-Flat tail phase on the scan kernel (comparator in the tests)
  int flat_filter2      = 1;   // CUVS_AMD_FLAT_FILTER2=0: IVF-Flat's filter with 64-query units per wave (round 3's kernel) instead of 256-query units per workgroup
  int flat_bound_head   = 1;   // CUVS_AMD_FLAT_BOUND_HEAD=0: IVF-Flat's head phase scores the nearest lists exactly on the scan kernel (rounds 1-5) instead of the bound-only pass through the fp16 copy
  int pq_wide           = 1;   // CUVS_AMD_PQ_WIDE=0: IVF-PQ shapes outside pq_filter4_kernel (rot_dim > 256, odd pq_len, large k) stay on the LUT scan instead of the wide matrix-core path (ivf_pq_wide.hip)
  int pq_wide_heads     = 0;   // CUVS_AMD_PQ_WIDE_HEADS: head lists per query of that path (0: pqw_heads' rule)
  int pq_wide_blocks    = 1;   // CUVS_AMD_PQ_WIDE_BLOCKS=0: the wide path's re-score at pq_len 2 by a wave per survivor (entries from memory) instead of pq_rescore_blocks_kernel (codebook staged in LDS block by block)
  int pq3_surv_cap      = 0;   // CUVS_AMD_PQ3_SURV_CAP: survivor-list entries of the matrix-core filter (test hook: forces the hand-back path)
  int pq_qcap           = 0;   // CUVS_AMD_PQ_QCAP: survivor-queue rows of pq_scan2_kernel (test hook: forces the overflow path)
  int scan_debug        = 0;   // CUVS_AMD_SCAN_DEBUG: ablation / statistics bits of the PQ scan
  int alloc_cache       = 1;   // CUVS_AMD_ALLOC_CACHE=0: every scratch buffer goes back to the runtime's pool when it is freed
  bool shard_coarse_replicated = false;  // CUVS_AMD_SHARD_COARSE_REPLICATED
  bool bf_fused = false, bf_no_threshold = false, bf_no_fused_filter = false;  // CUVS_AMD_BF_*
  bool bf_host_flags = false;  // CUVS_AMD_BF_HOST_FLAGS=1: the overflow flags of the fused path are read back by the host (rounds 2-3; comparator)
  bool dist_old         = false;  // CUVS_AMD_DIST_OLD
  int tile_dbg          = 0;      // CUVS_AMD_TILE_DBG
  int flat_head_probes  = -1;     // CUVS_AMD_FLAT_HEAD_PROBES
  int cagra_pq_lists = 0, cagra_pq_probes = 0, cagra_kpq = 0, cagra_rank_chunk = 0, prune_dbg = 0;  // CUVS_AMD_CAGRA_*, CUVS_AMD_PRUNE_DBG
  bool cagra_auto_multi = false;  // CUVS_AMD_CAGRA_AUTO=multi
  bool native_format    = false;  // CUVS_AMD_NATIVE_FORMAT=1: *Serialize writes this library's own container
};
tuning load_tuning_from_env();

struct resources {
  tuning tune;
  int device              = 0;
  hipStream_t stream      = nullptr;  // nullptr == the legacy default stream
  bool owns_stream        = false;
  int num_cus             = 256;
  size_t lds_per_block    = 160 * 1024;
  size_t workspace_limit  = size_t(2) << 30;  // temporary distance tiles etc.
  size_t ivf_batch_limit  = size_t(8) << 30;  // scratch an IVF search may use for ONE internal batch of queries (288 GB of HBM: a
                                              // list-sharded search of 8 x 10k queries would otherwise run as six batches); the
                                              // CUVS_AMD_WORKSPACE_MB test hook sets both
  std::vector<int> mg_devices;                // multi-GPU handle: participating devices
  hipMemPool_t pool       = nullptr;          // the handle's own stream-ordered pool (scratch buffers stay cached in it)
  hipStream_t aux_stream  = nullptr;          // helper stream + events for two-stream pipelines (brute force), made on first use
  hipEvent_t aux_events[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool cagra_guarantee_connectivity = false;  // cuvsAmdCagraSetGuaranteeConnectivity (cagra.hpp:193 has no C field)
  uint32_t* host_word = nullptr;              // one pinned word for small device -> host readbacks (made on first use)
  unsigned long long* cagra_work = nullptr;   // device [3]: rows scored / graph rows read / walkers (cuvsAmdCagraWorkCounters)
  struct scratch_cache* cache = nullptr;      // freed scratch blocks kept for the next call (core.hip); shared by copies of the handle
};

inline resources* as_res(uintptr_t h)
{
  CUVS_EXPECTS(h != 0, "null cuvsResources_t");
  return reinterpret_cast<resources*>(h);
}

// Stream-ordered device allocation (the reference uses rmm::device_uvector on the handle's
// stream; hipMallocAsync gives the same semantics from the driver's pool).
// the handle's helper stream + events (two-stream pipelines: brute force's select next to the next tile, the IVF-PQ batch's
// grouping / pre-pass next to the head kernel), made on first use. The helper stream has the HIGHEST priority: its kernels
// are short and sit on the critical path behind a long kernel of the handle's stream - the workgroup dispatcher then hands
// freed slots to them first instead of to the long kernel's backlog.
void ensure_aux_stream(resources& res);
void* device_alloc(resources& res, size_t bytes);
void device_free(resources& res, void* p);
void scratch_cache_flush_all();  // every handle's kept scratch blocks back to the runtime (an allocation failed)

// Two lifetimes: scratch buffers are stream-ordered (freed on the handle's stream); buffers owned by an
// index outlive the handle that built them, so they use plain hipMalloc/hipFree ("persistent").
template <typename T>
struct dev_buf {
  resources* res = nullptr;  // nullptr => persistent allocation
  T* ptr         = nullptr;
  size_t n       = 0;
  dev_buf() = default;
  dev_buf(resources& r, size_t count) : res(&r), n(count)
  {
    ptr = count ? static_cast<T*>(device_alloc(r, count * sizeof(T))) : nullptr;
  }
  static dev_buf persistent(size_t count)
  {
    dev_buf b;
    b.n = count;
    if (count) {
      void* p = nullptr;
      if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) {  // idle scratch blocks kept by the handles may be in the way
        (void)hipGetLastError();
        scratch_cache_flush_all();
        HIP_TRY(hipMalloc(&p, count * sizeof(T)));
      }
      b.ptr = static_cast<T*>(p);
    }
    return b;
  }
  dev_buf(const dev_buf&)            = delete;
  dev_buf& operator=(const dev_buf&) = delete;
  dev_buf(dev_buf&& o) noexcept : res(o.res), ptr(o.ptr), n(o.n) { o.ptr = nullptr; o.n = 0; }
  dev_buf& operator=(dev_buf&& o) noexcept
  {
    if (this != &o) {
      release();
      res = o.res; ptr = o.ptr; n = o.n;
      o.ptr = nullptr; o.n = 0;
    }
    return *this;
  }
  ~dev_buf() { release(); }
  void release()
  {
    if (ptr) {
      if (res) device_free(*res, ptr); else (void)hipFree(ptr);
      ptr = nullptr; n = 0;
    }
  }
  T* data() const { return ptr; }
  size_t size() const { return n; }
  size_t bytes() const { return n * sizeof(T); }
};

inline void copy_async(resources& res, void* dst, const void* src, size_t bytes)
{
  if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, res.stream));
}
inline void sync(resources& res) { HIP_TRY(hipStreamSynchronize(res.stream)); }

template <typename T>
std::vector<T> to_host(resources& res, const T* d, size_t n)
{
  std::vector<T> h(n);
  copy_async(res, h.data(), d, n * sizeof(T));
  sync(res);
  return h;
}

// one device word read back through the handle's pinned word (a pageable destination makes the runtime stage the copy)
inline uint32_t read_word(resources& res, const uint32_t* d)
{
  if (res.host_word == nullptr) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&res.host_word), 64));
  HIP_TRY(hipMemcpyAsync(res.host_word, d, sizeof(uint32_t), hipMemcpyDeviceToHost, res.stream));
  sync(res);
  return *res.host_word;
}

// CUVS_AMD_SCAN_DEBUG bit 8192: host-side timeline of one search call - wall time between marks WITHOUT synchronising
// (what the host spends enqueueing each section; a section that waits for the device shows the drain time)
struct host_trace {
  bool on;
  const char* what;
  std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
  host_trace(bool enabled, const char* name) : on(enabled), what(name) { mark("begin"); }
  void mark(const char* name)
  {
    if (on) marks.emplace_back(name, std::chrono::steady_clock::now());
  }
  ~host_trace()
  {
    if (!on) return;
    mark("buffers freed");
    fprintf(stderr, "[host trace %s]", what);
    for (size_t i = 1; i < marks.size(); ++i)
      fprintf(stderr, " %s %.3f", marks[i].first, std::chrono::duration<double, std::milli>(marks[i].second - marks[i - 1].second).count());
    fprintf(stderr, " | total %.3f ms\n", std::chrono::duration<double, std::milli>(marks.back().second - marks.front().second).count());
  }
};

// ---------------------------------------------------------------- DLPack checks
// Device-accessible = what the reference accepts (kDLCUDA/kDLCUDAHost/kDLCUDAManaged,
// c/src/core/detail/interop.hpp:49-53) plus the ROCm spellings.
inline bool is_device_accessible(const DLTensor& t)
{
  switch (t.device.device_type) {
    case kDLCUDA:
    case kDLCUDAHost:
    case kDLCUDAManaged:
    case kDLROCM:
    case kDLROCMHost: return true;
    default: return false;
  }
}
inline bool is_host_accessible(const DLTensor& t)
{
  switch (t.device.device_type) {
    case kDLCPU:
    case kDLCUDAHost:
    case kDLCUDAManaged:
    case kDLROCMHost: return true;
    default: return false;
  }
}
// interop.hpp:61-75
inline bool is_c_contiguous(const DLTensor& t)
{
  if (t.strides == nullptr) return true;
  int64_t expected = 1;
  for (int i = t.ndim - 1; i >= 0; --i) {
    if (t.shape[i] != 1 && t.strides[i] != expected) return false;
    expected *= t.shape[i];
  }
  return true;
}
// interop.hpp:77-92
inline bool is_f_contiguous(const DLTensor& t)
{
  if (t.strides == nullptr) return t.ndim <= 1;
  int64_t expected = 1;
  for (int i = 0; i < t.ndim; ++i) {
    if (t.shape[i] != 1 && t.strides[i] != expected) return false;
    expected *= t.shape[i];
  }
  return true;
}
inline bool dtype_is(const DLDataType& d, uint8_t code, uint8_t bits)
{
  return d.code == code && d.bits == bits && d.lanes == 1;
}
// The reference never reads DLTensor::byte_offset (c/src/core/detail/interop.hpp builds its mdspans from `data`
// alone) and its own C tests leave the field uninitialised on the stack (c/tests/neighbors/run_ivf_pq_c.c:24-33),
// so honouring it would crash callers that work against the reference: `data` is the address, as there.
inline void* dl_data(const DLTensor& t) { return t.data; }

enum class elem_t : int { f32 = 0, f16 = 1, i8 = 2, u8 = 3 };
inline elem_t elem_of(const DLDataType& d)
{
  if (dtype_is(d, kDLFloat, 32)) return elem_t::f32;
  if (dtype_is(d, kDLFloat, 16)) return elem_t::f16;
  if (dtype_is(d, kDLInt, 8)) return elem_t::i8;
  if (dtype_is(d, kDLUInt, 8)) return elem_t::u8;
  CUVS_FAIL("Unsupported DLtensor dtype: %d and bits: %d", (int)d.code, (int)d.bits);
}
inline size_t elem_size(elem_t e) { return e == elem_t::f32 ? 4 : (e == elem_t::f16 ? 2 : 1); }

// Fill a caller-allocated DLManagedTensor with a non-owning device view
// (reference: c/src/core/detail/interop.hpp:148-172 — shape array new[]-ed, freed by deleter).
void fill_dl_view(DLManagedTensor* out, void* data, DLDataType dt, int64_t rows, int64_t cols,
                  int ndim, int device_id);

// distance metric ids (include/cuvs/distance/distance.h)
enum metric_t : int {
  M_L2Expanded = 0, M_L2SqrtExpanded = 1, M_CosineExpanded = 2, M_L2Unexpanded = 4,
  M_L2SqrtUnexpanded = 5, M_InnerProduct = 6
};
inline bool metric_supported(int m)
{
  return m == 0 || m == 1 || m == 2 || m == 4 || m == 5 || m == 6;
}
inline bool metric_is_l2(int m) { return m == 0 || m == 1 || m == 4 || m == 5; }
inline bool metric_is_sqrt(int m) { return m == 1 || m == 5; }

// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default; cuvsAmdProfileEnable(1) turns it on, cuvsAmdProfileCollect reads and resets it.
void profile_begin(resources& res, const char* name);
void profile_end(resources& res, const char* name);

// 1-D grid size for 256-thread (or smaller) workgroups. HIP silently truncates launches whose
// gridDim.x * blockDim.x reaches 2^32 threads, so refuse them here (callers batch or grid-stride).
inline unsigned grid_blocks(int64_t n_items, int per_block)
{
  int64_t b = (n_items + per_block - 1) / per_block;
  CUVS_EXPECTS(b < (int64_t(1) << 24), "launch of %ld workgroups exceeds the 2^32-thread HIP grid limit", (long)b);
  return (unsigned)(b > 0 ? b : 1);
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// n queries in batches of at most `fit`: the batch size that makes the batches EQUAL (10000 queries at fit 9900 are two batches
// of 5000, not 9900 + 100 - a remainder below 256 queries runs without a head phase on the slow kernels)
inline int64_t balanced_batch(int64_t n, int64_t fit)
{
  fit = std::max<int64_t>(1, fit);
  if (n <= fit) return std::max<int64_t>(n, 1);
  const int64_t nb = (n + fit - 1) / fit;
  return (n + nb - 1) / nb;
}
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace cuvs_amd
