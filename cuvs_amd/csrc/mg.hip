// cuvsMultiGpu{IvfFlat,IvfPq,Cagra}* — single-process multi-GPU indexes behind the reference's C entry points
// (c/include/cuvs/neighbors/mg_*.h, c/src/neighbors/mg_*.cpp; semantics of cpp/src/neighbors/mg/snmg.cuh).
//
// The reference drives its GPUs from OpenMP threads and moves the partial results of a sharded search between GPUs
// with ncclSend/ncclRecv (root fan-in :248-375 or a log2(R) tree :377-516) before a device-side k-way merge. This
// interface hands back HOST tensors, so here no byte travels between GPUs at all: one host thread per GPU
// (hipSetDevice + its own cuvsResources/stream) stages its queries, searches its shard and copies its [Q, k] block
// straight to host memory over its own PCIe link; the R blocks (12 bytes per candidate) are merged on the host in one
// pass with the shard offsets applied (translation = rows held by the lower-numbered shards, :340-347). The
// multi-process path (one process per GPU, one RCCL all_gather) is cuvs_amd/mg.py.
//
// A multi-GPU index is R complete single-GPU indexes: REPLICATED = the same rows on every GPU, queries dealt out in
// batches; SHARDED = GPU r holds rows [r * ceil(n / R), ...) and sees every query. Everything below goes through this
// library's own single-GPU C entry points, so the per-GPU work is exactly what the single-GPU tests pin.
#include "ops.hpp"
#include "ivf_pq.hpp"
#include "npy_io.hpp"
#include "mg_host.hpp"

#include <cuvs/neighbors/mg_cagra.h>
#include <cuvs/neighbors/mg_ivf_flat.h>
#include <cuvs/neighbors/mg_ivf_pq.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace cuvs_amd {
namespace {

// a failing cuvs* call leaves its text for the calling thread: turn it into an exception of this thread
void ok(cuvsError_t e, const char* what)
{
  if (e == CUVS_SUCCESS) return;
  const char* t = cuvsGetLastErrorText();
  std::string msg = t ? t : "failed";
  CUVS_FAIL("%s: %s", what, msg.c_str());
}

// non-owning DLPack descriptor of a matrix (cols >= 0) or a vector (cols < 0)
struct dl_view {
  DLManagedTensor t{};
  int64_t shape[2] = {0, 0};
  dl_view(const void* data, DLDevice dev, DLDataType dt, int64_t rows, int64_t cols)
  {
    shape[0]                = rows;
    shape[1]                = cols;
    t.dl_tensor.data        = const_cast<void*>(data);
    t.dl_tensor.device      = dev;
    t.dl_tensor.ndim        = cols < 0 ? 1 : 2;
    t.dl_tensor.dtype       = dt;
    t.dl_tensor.shape       = shape;
    t.dl_tensor.strides     = nullptr;
    t.dl_tensor.byte_offset = 0;
    t.manager_ctx           = nullptr;
    t.deleter               = nullptr;
  }
  dl_view(const dl_view&)            = delete;
  dl_view& operator=(const dl_view&) = delete;
  DLManagedTensor* ptr() { return &t; }
};

constexpr DLDataType kI64 = {kDLInt, 64, 1};
constexpr DLDataType kF32 = {kDLFloat, 32, 1};

// the single-GPU entry points of one index type
struct algo_ops {
  const char* name;
  bool keeps_rows;  // the index views the device rows it was built from (CAGRA)
  void* (*create)();
  void (*destroy)(void* index);
  void (*build)(cuvsResources_t res, void* params, DLManagedTensor* rows, void* index);
  void (*extend)(cuvsResources_t res, DLManagedTensor* rows, DLManagedTensor* ids, void* index);
  void (*search)(cuvsResources_t res, void* params, void* index, DLManagedTensor* q, DLManagedTensor* nb,
                 DLManagedTensor* d);
  void (*serialize)(cuvsResources_t res, const char* filename, void* index);
  void (*deserialize)(cuvsResources_t res, const char* filename, void* index);
  void (*info)(void* index, int64_t* size, int* metric);
};

const cuvsFilter kNoFilter = {0, NO_FILTER};

const algo_ops kIvfFlatOps = {
  "ivf_flat",
  false,
  [](void) -> void* {
    cuvsIvfFlatIndex_t i = nullptr;
    ok(cuvsIvfFlatIndexCreate(&i), "cuvsIvfFlatIndexCreate");
    return i;
  },
  [](void* i) { (void)cuvsIvfFlatIndexDestroy(static_cast<cuvsIvfFlatIndex_t>(i)); },
  [](cuvsResources_t r, void* p, DLManagedTensor* rows, void* i) {
    ok(cuvsIvfFlatBuild(r, static_cast<cuvsIvfFlatIndexParams_t>(p), rows, static_cast<cuvsIvfFlatIndex_t>(i)),
       "cuvsIvfFlatBuild");
  },
  [](cuvsResources_t r, DLManagedTensor* rows, DLManagedTensor* ids, void* i) {
    ok(cuvsIvfFlatExtend(r, rows, ids, static_cast<cuvsIvfFlatIndex_t>(i)), "cuvsIvfFlatExtend");
  },
  [](cuvsResources_t r, void* p, void* i, DLManagedTensor* q, DLManagedTensor* nb, DLManagedTensor* d) {
    ok(cuvsIvfFlatSearch(r, static_cast<cuvsIvfFlatSearchParams_t>(p), static_cast<cuvsIvfFlatIndex_t>(i), q, nb, d,
                         kNoFilter),
       "cuvsIvfFlatSearch");
  },
  [](cuvsResources_t r, const char* f, void* i) {
    ok(cuvsIvfFlatSerialize(r, f, static_cast<cuvsIvfFlatIndex_t>(i)), "cuvsIvfFlatSerialize");
  },
  [](cuvsResources_t r, const char* f, void* i) {
    ok(cuvsIvfFlatDeserialize(r, f, static_cast<cuvsIvfFlatIndex_t>(i)), "cuvsIvfFlatDeserialize");
  },
  [](void* i, int64_t* size, int* metric) {
    ivf_flat_index_info(static_cast<cuvsIvfFlatIndex_t>(i)->addr, size, metric);
  },
};

const algo_ops kIvfPqOps = {
  "ivf_pq",
  false,
  [](void) -> void* {
    cuvsIvfPqIndex_t i = nullptr;
    ok(cuvsIvfPqIndexCreate(&i), "cuvsIvfPqIndexCreate");
    return i;
  },
  [](void* i) { (void)cuvsIvfPqIndexDestroy(static_cast<cuvsIvfPqIndex_t>(i)); },
  [](cuvsResources_t r, void* p, DLManagedTensor* rows, void* i) {
    ok(cuvsIvfPqBuild(r, static_cast<cuvsIvfPqIndexParams_t>(p), rows, static_cast<cuvsIvfPqIndex_t>(i)),
       "cuvsIvfPqBuild");
  },
  [](cuvsResources_t r, DLManagedTensor* rows, DLManagedTensor* ids, void* i) {
    ok(cuvsIvfPqExtend(r, rows, ids, static_cast<cuvsIvfPqIndex_t>(i)), "cuvsIvfPqExtend");
  },
  [](cuvsResources_t r, void* p, void* i, DLManagedTensor* q, DLManagedTensor* nb, DLManagedTensor* d) {
    ok(cuvsIvfPqSearch(r, static_cast<cuvsIvfPqSearchParams_t>(p), static_cast<cuvsIvfPqIndex_t>(i), q, nb, d),
       "cuvsIvfPqSearch");
  },
  [](cuvsResources_t r, const char* f, void* i) {
    ok(cuvsIvfPqSerialize(r, f, static_cast<cuvsIvfPqIndex_t>(i)), "cuvsIvfPqSerialize");
  },
  [](cuvsResources_t r, const char* f, void* i) {
    ok(cuvsIvfPqDeserialize(r, f, static_cast<cuvsIvfPqIndex_t>(i)), "cuvsIvfPqDeserialize");
  },
  [](void* i, int64_t* size, int* metric) {
    auto addr = static_cast<cuvsIvfPqIndex_t>(i)->addr;
    CUVS_EXPECTS(addr != 0, "ivf_pq index is empty");
    auto* idx = reinterpret_cast<const ivf_pq_index*>(addr);
    *size     = idx->size;
    *metric   = idx->metric;
  },
};

const algo_ops kCagraOps = {
  "cagra",
  true,
  [](void) -> void* {
    cuvsCagraIndex_t i = nullptr;
    ok(cuvsCagraIndexCreate(&i), "cuvsCagraIndexCreate");
    return i;
  },
  [](void* i) { (void)cuvsCagraIndexDestroy(static_cast<cuvsCagraIndex_t>(i)); },
  [](cuvsResources_t r, void* p, DLManagedTensor* rows, void* i) {
    ok(cuvsCagraBuild(r, static_cast<cuvsCagraIndexParams_t>(p), rows, static_cast<cuvsCagraIndex_t>(i)),
       "cuvsCagraBuild");
  },
  [](cuvsResources_t r, DLManagedTensor* rows, DLManagedTensor* ids, void* i) {
    // like the reference's interface layer (neighbors/iface/iface.hpp: cagra::extend takes no ids): rows are numbered on
    CUVS_EXPECTS(ids == nullptr, "cagra extend numbers the new rows itself: new_indices must be NULL");
    cuvsCagraExtendParams_t ep = nullptr;
    ok(cuvsCagraExtendParamsCreate(&ep), "cuvsCagraExtendParamsCreate");
    cuvsError_t e = cuvsCagraExtend(r, ep, rows, static_cast<cuvsCagraIndex_t>(i));
    std::string text;
    if (e != CUVS_SUCCESS) {
      const char* t = cuvsGetLastErrorText();
      text          = t ? t : "failed";
    }
    (void)cuvsCagraExtendParamsDestroy(ep);
    if (e != CUVS_SUCCESS) CUVS_FAIL("cuvsCagraExtend: %s", text.c_str());
  },
  [](cuvsResources_t r, void* p, void* i, DLManagedTensor* q, DLManagedTensor* nb, DLManagedTensor* d) {
    ok(cuvsCagraSearch(r, static_cast<cuvsCagraSearchParams_t>(p), static_cast<cuvsCagraIndex_t>(i), q, nb, d,
                       kNoFilter),
       "cuvsCagraSearch");
  },
  [](cuvsResources_t r, const char* f, void* i) {
    ok(cuvsCagraSerialize(r, f, static_cast<cuvsCagraIndex_t>(i), true), "cuvsCagraSerialize");
  },
  [](cuvsResources_t r, const char* f, void* i) {
    ok(cuvsCagraDeserialize(r, f, static_cast<cuvsCagraIndex_t>(i)), "cuvsCagraDeserialize");
  },
  [](void* i, int64_t* size, int* metric) {
    cagra_index_info(static_cast<cuvsCagraIndex_t>(i)->addr, size, metric);
  },
};

struct shard {
  int device          = 0;
  cuvsResources_t res = 0;        // this shard's handle: its own stream on `device`
  void* index         = nullptr;  // cuvs{IvfFlat,IvfPq,Cagra}Index_t
  dev_buf<char> rows;             // keeps_rows: the device copy the index views
};

struct mg_index {
  const algo_ops* ops = nullptr;
  int mode            = CUVS_NEIGHBORS_MG_SHARDED;
  std::vector<shard> shards;
  std::atomic<int64_t> round_robin{0};

  ~mg_index()
  {
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& s : shards) {
      (void)hipSetDevice(s.device);
      if (s.index) ops->destroy(s.index);
      s.rows.release();
      if (s.res) (void)cuvsResourcesDestroy(s.res);
    }
    (void)hipSetDevice(cur);
  }
};

// One host thread per shard, each bound to its GPU; the first failure is rethrown on the calling thread.
template <typename F>
void for_each_shard(mg_index& idx, F&& fn)
{
  const size_t n = idx.shards.size();
  std::vector<std::string> errors(n);
  std::vector<char> failed(n, 0);
  std::vector<std::thread> threads;
  threads.reserve(n);
  for (size_t r = 0; r < n; ++r) {
    threads.emplace_back([&, r] {
      try {
        HIP_TRY(hipSetDevice(idx.shards[r].device));
        fn((int)r, idx.shards[r]);
      } catch (const std::exception& e) {
        failed[r] = 1;
        errors[r] = e.what();
      } catch (...) {
        failed[r] = 1;
        errors[r] = "unknown error";
      }
    });
  }
  for (auto& t : threads) t.join();
  for (size_t r = 0; r < n; ++r)
    if (failed[r]) CUVS_FAIL("GPU %d (rank %d of %d): %s", idx.shards[r].device, (int)r, (int)n, errors[r].c_str());
}

// same, for a single shard (the calling thread's device stays untouched)
template <typename F>
void on_shard(mg_index& idx, int r, F&& fn)
{
  std::string error;
  bool failed = false;
  std::thread t([&] {
    try {
      HIP_TRY(hipSetDevice(idx.shards[r].device));
      fn(idx.shards[r]);
    } catch (const std::exception& e) {
      failed = true;
      error  = e.what();
    } catch (...) {
      failed = true;
      error  = "unknown error";
    }
  });
  t.join();
  if (failed) CUVS_FAIL("GPU %d (rank %d): %s", idx.shards[r].device, r, error.c_str());
}

std::vector<int> devices_of(cuvsResources_t res_h)
{
  auto& res = *as_res(res_h);
  if (!res.mg_devices.empty()) return res.mg_devices;
  return {res.device};  // a plain handle: one "rank"
}

hipStream_t stream_of(const shard& s) { return as_res(s.res)->stream; }

void check_matrix(const DLTensor& t, const char* what)
{
  CUVS_EXPECTS(t.ndim == 2 && is_c_contiguous(t), "%s must be a row-major matrix", what);
  CUVS_EXPECTS(t.dtype.lanes == 1 && ((t.dtype.code == kDLFloat && (t.dtype.bits == 32 || t.dtype.bits == 16)) ||
                                      ((t.dtype.code == kDLInt || t.dtype.code == kDLUInt) && t.dtype.bits == 8)),
               "Unsupported %s DLtensor dtype: %d and bits: %d", what, (int)t.dtype.code, (int)t.dtype.bits);
}

void make_shards(mg_index& idx, const std::vector<int>& devices)
{
  idx.shards.resize(devices.size());
  for (size_t r = 0; r < devices.size(); ++r) idx.shards[r].device = devices[r];
  for_each_shard(idx, [&](int, shard& s) { ok(cuvsResourcesCreate(&s.res), "cuvsResourcesCreate"); });
}

std::unique_ptr<mg_index> mg_build(cuvsResources_t res_h, const algo_ops& ops, int mode, void* base_params,
                                   DLManagedTensor* dataset)
{
  CUVS_EXPECTS(dataset != nullptr && base_params != nullptr, "null argument");
  auto& ds = dataset->dl_tensor;
  check_matrix(ds, "dataset");
  CUVS_EXPECTS(mode == CUVS_NEIGHBORS_MG_REPLICATED || mode == CUVS_NEIGHBORS_MG_SHARDED, "unknown distribution mode %d", mode);
  auto idx  = std::make_unique<mg_index>();
  idx->ops  = &ops;
  idx->mode = mode;
  make_shards(*idx, devices_of(res_h));
  const int n_ranks     = (int)idx->shards.size();
  const int64_t n       = ds.shape[0], dim = ds.shape[1];
  const size_t row_bytes = (size_t)dim * (ds.dtype.bits / 8);
  for_each_shard(*idx, [&](int r, shard& s) {
    int64_t r0, cnt;
    rows_of_rank(mode == CUVS_NEIGHBORS_MG_SHARDED, n, r, n_ranks, &r0, &cnt);
    CUVS_EXPECTS(cnt > 0, "no rows left for this rank: %ld rows over %d GPUs", (long)n, n_ranks);
    auto rows = dev_buf<char>::persistent((size_t)cnt * row_bytes);
    HIP_TRY(hipMemcpyAsync(rows.data(), static_cast<const char*>(dl_data(ds)) + (size_t)r0 * row_bytes,
                           (size_t)cnt * row_bytes, hipMemcpyDefault, stream_of(s)));
    HIP_TRY(hipStreamSynchronize(stream_of(s)));
    dl_view v(rows.data(), DLDevice{kDLROCM, s.device}, ds.dtype, cnt, dim);
    s.index = ops.create();
    ops.build(s.res, base_params, v.ptr(), s.index);
    ok(cuvsStreamSync(s.res), "cuvsStreamSync");
    if (ops.keeps_rows) s.rows = std::move(rows);
  });
  return idx;
}

void mg_extend(mg_index& idx, DLManagedTensor* new_vectors, DLManagedTensor* new_indices)
{
  CUVS_EXPECTS(new_vectors != nullptr, "null argument");
  auto& nv = new_vectors->dl_tensor;
  check_matrix(nv, "new_vectors");
  const int64_t n = nv.shape[0], dim = nv.shape[1];
  const size_t row_bytes = (size_t)dim * (nv.dtype.bits / 8);
  const int64_t* ids     = nullptr;
  DLDevice ids_dev{kDLCPU, 0};
  if (new_indices != nullptr) {
    auto& ni = new_indices->dl_tensor;
    CUVS_EXPECTS(dtype_is(ni.dtype, kDLInt, 64) && ni.ndim == 1 && ni.shape[0] == n,
                 "new_indices must hold one int64 id per new row");
    ids     = static_cast<const int64_t*>(dl_data(ni));
    ids_dev = ni.device;
  }
  const int n_ranks = (int)idx.shards.size();
  for_each_shard(idx, [&](int r, shard& s) {
    int64_t r0, cnt;
    rows_of_rank(idx.mode == CUVS_NEIGHBORS_MG_SHARDED, n, r, n_ranks, &r0, &cnt);
    if (cnt <= 0) return;
    dev_buf<char> rows(*as_res(s.res), (size_t)cnt * row_bytes);
    HIP_TRY(hipMemcpyAsync(rows.data(), static_cast<const char*>(dl_data(nv)) + (size_t)r0 * row_bytes,
                           (size_t)cnt * row_bytes, hipMemcpyDefault, stream_of(s)));
    dl_view v(rows.data(), DLDevice{kDLROCM, s.device}, nv.dtype, cnt, dim);
    dl_view iv(ids ? ids + r0 : nullptr, ids_dev, kI64, cnt, -1);
    idx.ops->extend(s.res, v.ptr(), ids ? iv.ptr() : nullptr, s.index);
    ok(cuvsStreamSync(s.res), "cuvsStreamSync");
  });
}

// search `cnt` queries starting at row `off` on one shard; results to host (or any) memory at out_i / out_d
void search_block(shard& s, const algo_ops& ops, void* base_params, const DLTensor& q, int64_t off, int64_t cnt, int64_t k,
                  int64_t* out_i, float* out_d)
{
  const int64_t dim      = q.shape[1];
  const size_t row_bytes = (size_t)dim * (q.dtype.bits / 8);
  auto& res              = *as_res(s.res);
  dev_buf<char> dq(res, (size_t)cnt * row_bytes);
  dev_buf<int64_t> di(res, (size_t)cnt * k);
  dev_buf<float> dd(res, (size_t)cnt * k);
  HIP_TRY(hipMemcpyAsync(dq.data(), static_cast<const char*>(dl_data(q)) + (size_t)off * row_bytes,
                         (size_t)cnt * row_bytes, hipMemcpyDefault, res.stream));
  const DLDevice dev{kDLROCM, s.device};
  dl_view vq(dq.data(), dev, q.dtype, cnt, dim), vi(di.data(), dev, kI64, cnt, k), vd(dd.data(), dev, kF32, cnt, k);
  ops.search(s.res, base_params, s.index, vq.ptr(), vi.ptr(), vd.ptr());
  HIP_TRY(hipMemcpyAsync(out_i, di.data(), (size_t)cnt * k * sizeof(int64_t), hipMemcpyDefault, res.stream));
  HIP_TRY(hipMemcpyAsync(out_d, dd.data(), (size_t)cnt * k * sizeof(float), hipMemcpyDefault, res.stream));
  HIP_TRY(hipStreamSynchronize(res.stream));
}

void mg_search(mg_index& idx, void* base_params, int search_mode, int64_t n_rows_per_batch, DLManagedTensor* queries,
               DLManagedTensor* neighbors, DLManagedTensor* distances)
{
  CUVS_EXPECTS(queries && neighbors && distances && base_params, "null argument");
  auto& q  = queries->dl_tensor;
  auto& nb = neighbors->dl_tensor;
  auto& ds = distances->dl_tensor;
  check_matrix(q, "queries");
  CUVS_EXPECTS(dtype_is(nb.dtype, kDLInt, 64) && nb.ndim == 2 && is_c_contiguous(nb), "neighbors must be an int64 row-major matrix");
  CUVS_EXPECTS(dtype_is(ds.dtype, kDLFloat, 32) && ds.ndim == 2 && is_c_contiguous(ds), "distances must be a float32 row-major matrix");
  const int64_t nq = q.shape[0], k = nb.shape[1];
  CUVS_EXPECTS(nb.shape[0] == nq && ds.shape[0] == nq && ds.shape[1] == k && k > 0, "neighbors/distances must be [n_queries, k]");
  if (nq == 0) return;
  CUVS_EXPECTS(n_rows_per_batch > 0, "n_rows_per_batch must be positive");
  int64_t* out_i     = static_cast<int64_t*>(dl_data(nb));
  float* out_d       = static_cast<float*>(dl_data(ds));
  const int n_ranks  = (int)idx.shards.size();
  const auto& ops    = *idx.ops;
  if (idx.mode == CUVS_NEIGHBORS_MG_REPLICATED) {
    if (search_mode == CUVS_NEIGHBORS_MG_ROUND_ROBIN) {  // snmg.cuh:633-655: the whole call on the next GPU
      CUVS_EXPECTS(nq <= n_rows_per_batch, "In round-robin mode, n_rows must lower or equal to n_rows_per_batch");
      const int r = (int)(idx.round_robin++ % n_ranks);
      on_shard(idx, r, [&](shard& s) { search_block(s, ops, base_params, q, 0, nq, k, out_i, out_d); });
      return;
    }
    // snmg.cuh:596-632: at least one batch per GPU, batch b goes to GPU b mod R
    int64_t batch, n_batches;
    replicated_batches(nq, n_rows_per_batch, n_ranks, &batch, &n_batches);
    for_each_shard(idx, [&](int r, shard& s) {
      for (int64_t b = r; b < n_batches; b += n_ranks) {
        const int64_t off = b * batch, cnt = std::min(batch, nq - off);
        search_block(s, ops, base_params, q, off, cnt, k, out_i + off * k, out_d + off * k);
      }
    });
    return;
  }
  // SHARDED (snmg.cuh:656-720): every GPU searches every batch, the host merges - straight into the output tensors,
  // which the multi-GPU API defines as host tensors (mg_ivf_pq.h:152-190)
  CUVS_EXPECTS(is_host_accessible(nb) && is_host_accessible(ds),
               "multi-GPU sharded search: neighbors and distances must be host-accessible tensors");
  int64_t batch, n_batches;
  sharded_batches(nq, n_rows_per_batch, &batch, &n_batches);
  std::vector<int64_t> translation(n_ranks, 0);
  int metric = 0;
  {
    int64_t total = 0;
    for (int r = 0; r < n_ranks; ++r) {
      int64_t size = 0;
      ops.info(idx.shards[r].index, &size, &metric);
      translation[r] = total;
      total += size;
    }
  }
  const bool select_min = metric != M_InnerProduct;
  if (n_ranks == 1) {  // nothing to merge
    on_shard(idx, 0, [&](shard& s) {
      for (int64_t b = 0; b < n_batches; ++b) {
        const int64_t off = b * batch, cnt = std::min(batch, nq - off);
        search_block(s, ops, base_params, q, off, cnt, k, out_i + off * k, out_d + off * k);
      }
    });
    return;
  }
  std::vector<int64_t> part_i((size_t)n_ranks * batch * k);
  std::vector<float> part_d((size_t)n_ranks * batch * k);
  for (int64_t b = 0; b < n_batches; ++b) {
    const int64_t off = b * batch, cnt = std::min(batch, nq - off);
    for_each_shard(idx, [&](int r, shard& s) {
      search_block(s, ops, base_params, q, off, cnt, k, part_i.data() + (size_t)r * cnt * k, part_d.data() + (size_t)r * cnt * k);
    });
    merge_on_host(part_i.data(), part_d.data(), n_ranks, cnt, k, translation.data(), select_min, out_i + off * k, out_d + off * k);
  }
}

// File: dtype prefix (4 bytes), mode, number of ranks (numpy scalar records), then the R index streams
// (snmg.cuh:735-757). The per-index writers append through the npy_io window of the writing thread.
void mg_serialize(mg_index& idx, DLDataType dtype, const char* filename)
{
  CUVS_EXPECTS(filename != nullptr, "filename is null");
  for (auto& sh : idx.shards)
    CUVS_EXPECTS(sh.res == 0 || !write_native_container(*as_res(sh.res)),
                 "multi-GPU index files use the reference container: unset CUVS_AMD_NATIVE_FORMAT");
  {
    npy_writer w(filename);
    char prefix[4];
    elem_prefix(elem_of(dtype), prefix);
    w.raw(prefix, 4);
    w.scalar<int32_t>(idx.mode);
    w.scalar<int32_t>((int32_t)idx.shards.size());
    w.close();
  }
  for (int r = 0; r < (int)idx.shards.size(); ++r) {
    on_shard(idx, r, [&](shard& s) {
      g_npy_io        = npy_io_window{};
      g_npy_io.append = true;
      try {
        idx.ops->serialize(s.res, filename, s.index);
      } catch (...) {
        g_npy_io = npy_io_window{};
        throw;
      }
      g_npy_io = npy_io_window{};
    });
  }
}

std::unique_ptr<mg_index> mg_deserialize(cuvsResources_t res_h, const algo_ops& ops, const char* filename, DLDataType* dtype)
{
  CUVS_EXPECTS(filename != nullptr, "filename is null");
  auto idx = std::make_unique<mg_index>();
  idx->ops = &ops;
  long offset = 0;
  int n_ranks = 0;
  {
    npy_reader r(filename);
    char prefix[4];
    r.raw(prefix, 4);
    elem_t et;
    CUVS_EXPECTS(parse_elem_prefix(prefix, &et), "Unsupported dtype in file %s", filename);
    *dtype    = dl_of(et);
    idx->mode = r.scalar<int32_t>();
    n_ranks   = r.scalar<int32_t>();
    offset    = r.tell();
  }
  auto devices = devices_of(res_h);
  CUVS_EXPECTS(n_ranks == (int)devices.size(), "Serialized index has %d ranks whereas the resources handle has %d GPUs", n_ranks,
               (int)devices.size());
  CUVS_EXPECTS(idx->mode == CUVS_NEIGHBORS_MG_REPLICATED || idx->mode == CUVS_NEIGHBORS_MG_SHARDED, "%s: unknown distribution mode", filename);
  make_shards(*idx, devices);
  for (int r = 0; r < n_ranks; ++r) {
    on_shard(*idx, r, [&](shard& s) {
      s.index              = ops.create();
      g_npy_io             = npy_io_window{};
      g_npy_io.read_offset = offset;
      try {
        ops.deserialize(s.res, filename, s.index);
      } catch (...) {
        g_npy_io = npy_io_window{};
        throw;
      }
      offset   = g_npy_io.end_offset;
      g_npy_io = npy_io_window{};
      ok(cuvsStreamSync(s.res), "cuvsStreamSync");
    });
  }
  return idx;
}

// snmg.cuh:43-55: a single-GPU index file loaded onto every GPU
std::unique_ptr<mg_index> mg_distribute(cuvsResources_t res_h, const algo_ops& ops, const char* filename)
{
  CUVS_EXPECTS(filename != nullptr, "filename is null");
  auto idx  = std::make_unique<mg_index>();
  idx->ops  = &ops;
  idx->mode = CUVS_NEIGHBORS_MG_REPLICATED;
  make_shards(*idx, devices_of(res_h));
  for_each_shard(*idx, [&](int, shard& s) {
    s.index = ops.create();
    ops.deserialize(s.res, filename, s.index);
    ok(cuvsStreamSync(s.res), "cuvsStreamSync");
  });
  return idx;
}

// the dtype a single-GPU handle reports after deserialization (IVF-PQ files carry none: fp32 is assumed there, as the
// reference's C layer does for its untyped IVF-PQ index)
template <typename Handle>
DLDataType dtype_of_first(mg_index& idx)
{
  return static_cast<Handle*>(idx.shards.at(0).index)->dtype;
}

template <typename Handle>
void adopt(Handle* handle, std::unique_ptr<mg_index> idx, DLDataType dtype)
{
  CUVS_EXPECTS(handle != nullptr, "index handle is null");
  delete reinterpret_cast<mg_index*>(handle->addr);
  handle->addr  = reinterpret_cast<uintptr_t>(idx.release());
  handle->dtype = dtype;
}

template <typename Handle>
mg_index& built(Handle* handle)
{
  CUVS_EXPECTS(handle != nullptr && handle->addr != 0, "multi-GPU index is not built");
  return *reinterpret_cast<mg_index*>(handle->addr);
}

}  // namespace
}  // namespace cuvs_amd

using namespace cuvs_amd;

// The three families have the same shape (c/src/neighbors/mg_{ivf_flat,ivf_pq,cagra}.cpp): one definition each.
#define CUVS_AMD_MG_API(A, OPS)                                                                                        \
  extern "C" cuvsError_t cuvsMultiGpu##A##IndexParamsCreate(cuvsMultiGpu##A##IndexParams_t* index_params)             \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      CUVS_EXPECTS(index_params != nullptr, "null argument");                                                          \
      cuvs##A##IndexParams_t base = nullptr;                                                                           \
      ok(cuvs##A##IndexParamsCreate(&base), "base index params");                                                      \
      *index_params = new cuvsMultiGpu##A##IndexParams{base, CUVS_NEIGHBORS_MG_SHARDED};                               \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##IndexParamsDestroy(cuvsMultiGpu##A##IndexParams_t index_params)              \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      if (index_params == nullptr) return;                                                                             \
      (void)cuvs##A##IndexParamsDestroy(index_params->base_params);                                                    \
      delete index_params;                                                                                             \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##SearchParamsCreate(cuvsMultiGpu##A##SearchParams_t* params)                  \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      CUVS_EXPECTS(params != nullptr, "null argument");                                                                \
      cuvs##A##SearchParams_t base = nullptr;                                                                          \
      ok(cuvs##A##SearchParamsCreate(&base), "base search params");                                                    \
      *params = new cuvsMultiGpu##A##SearchParams{base, CUVS_NEIGHBORS_MG_LOAD_BALANCER, CUVS_NEIGHBORS_MG_TREE_MERGE, \
                                                  int64_t(1) << 20};                                                   \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##SearchParamsDestroy(cuvsMultiGpu##A##SearchParams_t params)                  \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      if (params == nullptr) return;                                                                                   \
      (void)cuvs##A##SearchParamsDestroy(params->base_params);                                                         \
      delete params;                                                                                                   \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##IndexCreate(cuvsMultiGpu##A##Index_t* index)                                 \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      CUVS_EXPECTS(index != nullptr, "null argument");                                                                 \
      *index = new cuvsMultiGpu##A##Index{};                                                                           \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##IndexDestroy(cuvsMultiGpu##A##Index_t index)                                 \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      if (index == nullptr) return;                                                                                    \
      delete reinterpret_cast<mg_index*>(index->addr);                                                                 \
      delete index;                                                                                                    \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##Build(cuvsResources_t res, cuvsMultiGpu##A##IndexParams_t params,            \
                                                DLManagedTensor* dataset_tensor, cuvsMultiGpu##A##Index_t index)       \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      CUVS_EXPECTS(params != nullptr && dataset_tensor != nullptr && index != nullptr, "null argument");               \
      auto idx = mg_build(res, OPS, (int)params->mode, params->base_params, dataset_tensor);                           \
      adopt(index, std::move(idx), dataset_tensor->dl_tensor.dtype);                                                   \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##Search(cuvsResources_t res, cuvsMultiGpu##A##SearchParams_t params,          \
                                                 cuvsMultiGpu##A##Index_t index, DLManagedTensor* queries_tensor,      \
                                                 DLManagedTensor* neighbors_tensor, DLManagedTensor* distances_tensor) \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      (void)as_res(res);                                                                                               \
      CUVS_EXPECTS(params != nullptr, "null argument");                                                                \
      mg_search(built(index), params->base_params, (int)params->search_mode, params->n_rows_per_batch, queries_tensor, \
                neighbors_tensor, distances_tensor);                                                                   \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##Extend(cuvsResources_t res, cuvsMultiGpu##A##Index_t index,                  \
                                                 DLManagedTensor* new_vectors_tensor,                                  \
                                                 DLManagedTensor* new_indices_tensor)                                  \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      (void)as_res(res);                                                                                               \
      mg_extend(built(index), new_vectors_tensor, new_indices_tensor);                                                 \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##Serialize(cuvsResources_t res, cuvsMultiGpu##A##Index_t index,               \
                                                    const char* filename)                                              \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      (void)as_res(res);                                                                                               \
      mg_serialize(built(index), index->dtype, filename);                                                              \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##Deserialize(cuvsResources_t res, const char* filename,                       \
                                                      cuvsMultiGpu##A##Index_t index)                                  \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      CUVS_EXPECTS(index != nullptr, "null argument");                                                                 \
      DLDataType dtype{};                                                                                              \
      auto idx = mg_deserialize(res, OPS, filename, &dtype);                                                           \
      adopt(index, std::move(idx), dtype);                                                                             \
    });                                                                                                                \
  }                                                                                                                    \
  extern "C" cuvsError_t cuvsMultiGpu##A##Distribute(cuvsResources_t res, const char* filename,                        \
                                                     cuvsMultiGpu##A##Index_t index)                                   \
  {                                                                                                                    \
    return (cuvsError_t)translate_exceptions([=] {                                                                     \
      CUVS_EXPECTS(index != nullptr, "null argument");                                                                 \
      auto idx         = mg_distribute(res, OPS, filename);                                                            \
      DLDataType dtype = dtype_of_first<cuvs##A##Index>(*idx);                                                         \
      adopt(index, std::move(idx), dtype);                                                                             \
    });                                                                                                                \
  }

CUVS_AMD_MG_API(IvfFlat, kIvfFlatOps)
CUVS_AMD_MG_API(IvfPq, kIvfPqOps)
CUVS_AMD_MG_API(Cagra, kCagraOps)
#undef CUVS_AMD_MG_API
