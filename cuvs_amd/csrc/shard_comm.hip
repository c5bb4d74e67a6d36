// List-sharded multi-GPU search: communicator (RCCL over xGMI, one process per GPU), the all-gather of per-rank
// top-k blocks and their merge. Interface and the reference code it stands in for: include/cuvs_amd/shard.h
// (reference: cpp/src/neighbors/mg/snmg.cuh:248-375 sharded search + :298-340 NCCL send/recv fan-in,
// cpp/src/neighbors/detail/knn_merge_parts.cuh:27-103).
//
// One ncclAllGather per batch instead of the reference's R-1 send/recv pairs into a root: xGMI is point-to-point, a
// ring all-gather of 12 * Q * k bytes per rank (1.2 MB at Q = 10k, k = 10) keeps every link busy once, and every
// rank ends up with the merged result (no broadcast afterwards).
#include "ivf_pq.hpp"
#include "ops.hpp"
#include "shm_transport.hpp"

#include <cuvs_amd/shard.h>

#include <rccl/rccl.h>

#include <atomic>
#include <cfloat>
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <string>

// Two transports behind one communicator: RCCL (one process per GPU, xGMI) and the host-staged one of
// shm_transport.hpp (processes of one host through a mapped file: ranks that SHARE a device, or hosts without RCCL).
// Which one a communicator uses is decided by the rendezvous id its ranks were created with.
struct cuvsAmdShardComm {
  ncclComm_t comm = nullptr;
  std::unique_ptr<cuvs_amd::shm_transport> shm;
  int rank = 0, world = 1, device = 0;
};

namespace cuvs_amd {
namespace {

// RCCL entry points, resolved once. dlopen by soname: if the process already holds librccl.so.1 (PyTorch loads its
// own copy), that copy is shared; otherwise the ROCm one is loaded.
struct rccl_api {
  decltype(&ncclGetUniqueId) get_unique_id       = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank     = nullptr;
  decltype(&ncclCommDestroy) comm_destroy        = nullptr;
  decltype(&ncclAllGather) all_gather            = nullptr;
  decltype(&ncclAllReduce) all_reduce            = nullptr;
  decltype(&ncclGetErrorString) get_error_string = nullptr;
};

const rccl_api& rccl()
{
  static rccl_api api;
  static std::string load_error = "missing symbols";  // dlerror() text of the failed dlopen, captured once
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h != nullptr) break;
      if (const char* err = dlerror()) load_error = err;  // (dlerror() clears the state: read it once)
    }
    if (h == nullptr) return;
    api.get_unique_id    = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    api.comm_init_rank   = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    api.comm_destroy     = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    api.all_gather       = reinterpret_cast<decltype(api.all_gather)>(dlsym(h, "ncclAllGather"));
    api.all_reduce       = reinterpret_cast<decltype(api.all_reduce)>(dlsym(h, "ncclAllReduce"));
    api.get_error_string = reinterpret_cast<decltype(api.get_error_string)>(dlsym(h, "ncclGetErrorString"));
  });
  CUVS_EXPECTS(api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_gather && api.all_reduce &&
                 api.get_error_string,
               "RCCL (librccl.so.1) could not be loaded: %s", load_error.c_str());
  return api;
}

#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r__ = (expr);                                                                           \
    if (r__ != ncclSuccess) CUVS_FAIL("RCCL error %d (%s) in %s", (int)r__, rccl().get_error_string(r__), #expr); \
  } while (0)

// The host-staged forms of the two collectives (shm_transport.hpp): the stream is drained, the rank's block goes
// device -> mapped file, every rank's block comes back file -> device. In place like the RCCL calls they stand in for:
// `buf` holds world blocks of `bytes`, this rank's at its rank offset. Host-synchronous - a functional transport for
// ranks that share a device, not a fast one.
void staged_all_gather(resources& res, shm_transport& t, void* buf, size_t bytes)
{
  char* b = static_cast<char*>(buf);
  t.begin(1, bytes);
  try {
    HIP_TRY(hipMemcpyAsync(t.own(), b + (size_t)t.rank() * bytes, bytes, hipMemcpyDeviceToHost, res.stream));
    sync(res);
    t.publish();
    for (int r = 0; r < t.world(); ++r)
      if (r != t.rank()) HIP_TRY(hipMemcpyAsync(b + (size_t)r * bytes, t.block(r), bytes, hipMemcpyHostToDevice, res.stream));
    sync(res);
  } catch (...) {
    t.mark_failed();  // (the peers are inside the same collective: they raise instead of waiting out the time limit)
    throw;
  }
  t.end();
}

void staged_all_reduce_min(resources& res, shm_transport& t, uint32_t* keys, size_t count)
{
  const size_t bytes = count * sizeof(uint32_t);
  t.begin(2, bytes);
  std::vector<uint32_t> m(count);
  try {
    HIP_TRY(hipMemcpyAsync(t.own(), keys, bytes, hipMemcpyDeviceToHost, res.stream));
    sync(res);
    t.publish();
    std::memcpy(m.data(), t.block(0), bytes);
    for (int r = 1; r < t.world(); ++r) {
      const uint32_t* o = reinterpret_cast<const uint32_t*>(t.block(r));
      for (size_t i = 0; i < count; ++i) m[i] = o[i] < m[i] ? o[i] : m[i];
    }
  } catch (...) {
    t.mark_failed();
    throw;
  }
  t.end();  // (every rank has read the blocks into its own vector)
  HIP_TRY(hipMemcpyAsync(keys, m.data(), bytes, hipMemcpyHostToDevice, res.stream));
  sync(res);
}

// one candidate = 12 bytes on the wire: a rank's block is [n_queries, k] ids (int64) followed by [n_queries, k] distances
// (fp32), padded to a multiple of 16 bytes so that the ids of every block are 8-byte aligned whatever n_queries * k is
__host__ __device__ inline size_t wire_block_bytes(int64_t n) { return ((size_t)n * 12 + 15) & ~size_t(15); }
__global__ void pack_block_kernel(const float* __restrict__ d, const int64_t* __restrict__ i, int64_t n,
                                  char* __restrict__ out)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  reinterpret_cast<int64_t*>(out)[t]               = i[t];
  reinterpret_cast<float*>(out + (size_t)n * 8)[t] = d[t];
}

// gathered [world][ ids n*8 | distances n*4 | pad ] -> per-query rows [n_queries, world * k] (rank-major inside a row)
// Slots a rank could not fill (fewer than k rows in its probed lists) carry the id INT64_MAX and the distance FLT_MAX
// (ivf_common.cuh:31 kOutOfBoundsRecord); for similarity metrics they must lose the merge, so they enter it as -FLT_MAX
// and leave it as FLT_MAX again (pad_invalid_kernel).
__global__ void regroup_kernel(const char* __restrict__ gathered, int64_t nq, int k, int world, float* __restrict__ vals,
                               int64_t* __restrict__ ids, bool select_min)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = nq * k;
  if (t >= n * world) return;
  const int64_t r = t / n, rem = t % n, q = rem / k, j = rem % k;
  const char* blk = gathered + (size_t)r * wire_block_bytes(n);
  const int64_t o = q * ((int64_t)world * k) + r * k + j;
  const int64_t id = reinterpret_cast<const int64_t*>(blk)[rem];
  const float v    = reinterpret_cast<const float*>(blk + (size_t)n * 8)[rem];
  vals[o]          = id == INT64_MAX ? (select_min ? FLT_MAX : -FLT_MAX) : v;
  ids[o]           = id;
}

// local row ids of a row-range shard -> global ids (snmg.cuh:420-429); slots without a neighbour (0xffffffff from the
// CAGRA walk, negative or INT64_MAX elsewhere) become INT64_MAX, the id the merge treats as "no candidate"
template <typename IdT>
__global__ void translate_ids_kernel(const IdT* __restrict__ in, int64_t n, int64_t row_offset, int64_t* __restrict__ out)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const IdT v = in[t];
  bool invalid;
  if constexpr (sizeof(IdT) == 4) invalid = v == (IdT)0xffffffffu;
  else                            invalid = v < 0 || v == INT64_MAX;
  out[t] = invalid ? INT64_MAX : (int64_t)v + row_offset;
}

__global__ void pad_invalid_kernel(float* __restrict__ d, const int64_t* __restrict__ i, int64_t n)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && i[t] == INT64_MAX) d[t] = FLT_MAX;
}

}  // namespace

// host/CPU-testable twin of the merge rule lives in cuvs_amd/mg.py (`merge_gathered`); the device path:
void shard_all_gather_topk(resources& res, cuvsAmdShardComm& c, const float* ld, const int64_t* li, int64_t nq, int k,
                           bool select_min, float* out_d, int64_t* out_i)
{
  CUVS_EXPECTS(nq >= 0 && k > 0, "shard all-gather: bad shape");
  if (nq == 0) return;
  const int64_t n = nq * k;
  const size_t blk = wire_block_bytes(n);
  // in place: this rank's block is packed where the all-gather leaves it (no send buffer, no local copy inside RCCL)
  dev_buf<char> recv(res, blk * c.world);
  char* send = recv.data() + (size_t)c.rank * blk;
  hipLaunchKernelGGL(pack_block_kernel, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream, ld, li, n, send);
  profile_begin(res, "shard_all_gather");
  if (c.shm) staged_all_gather(res, *c.shm, recv.data(), blk);
  else       RCCL_TRY(rccl().all_gather(send, recv.data(), blk, ncclUint8, c.comm, res.stream));
  profile_end(res, "shard_all_gather");
  dev_buf<float> vals(res, (size_t)n * c.world);
  dev_buf<int64_t> ids(res, (size_t)n * c.world);
  hipLaunchKernelGGL(regroup_kernel, dim3(grid_blocks(n * c.world, 256)), dim3(256), 0, res.stream, recv.data(), nq, k,
                     c.world, vals.data(), ids.data(), select_min);
  select_k<int64_t, int64_t>(res, vals.data(), ids.data(), nq, (int64_t)c.world * k, (int64_t)c.world * k, k, out_d, out_i,
                             select_min);
  hipLaunchKernelGGL(pad_invalid_kernel, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream, out_d, out_i, n);
  HIP_TRY(hipGetLastError());
}

void shard_allreduce_min_u32(resources& res, void* comm, uint32_t* keys, size_t count)
{
  auto* c = static_cast<cuvsAmdShardComm*>(comm);
  if (c == nullptr || count == 0) return;  // (a one-rank communicator still makes the call: that is what the tests run)
  profile_begin(res, "shard_all_reduce");
  if (c->shm) staged_all_reduce_min(res, *c->shm, keys, count);
  else        RCCL_TRY(rccl().all_reduce(keys, keys, count, ncclUint32, ncclMin, c->comm, res.stream));
  profile_end(res, "shard_all_reduce");
}

void shard_allgather_inplace_u32(resources& res, void* comm, uint32_t* buf, size_t count)
{
  auto* c = static_cast<cuvsAmdShardComm*>(comm);
  if (c == nullptr || count == 0) return;
  profile_begin(res, "shard_all_gather_probes");
  if (c->shm) staged_all_gather(res, *c->shm, buf, count * sizeof(uint32_t));
  else        RCCL_TRY(rccl().all_gather(buf + (size_t)c->rank * count, buf, count, ncclUint32, c->comm, res.stream));
  profile_end(res, "shard_all_gather_probes");
}

}  // namespace cuvs_amd

using namespace cuvs_amd;

extern "C" {

cuvsError_t cuvsAmdShardCommGetUniqueId(char id[CUVS_AMD_SHARD_ID_BYTES])
{
  return (cuvsError_t)translate_exceptions([=] {
    static_assert(CUVS_AMD_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    RCCL_TRY(rccl().get_unique_id(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  });
}

cuvsError_t cuvsAmdShardCommGetUniqueIdHostStaged(char id[CUVS_AMD_SHARD_ID_BYTES])
{
  return (cuvsError_t)translate_exceptions([=] {
    static std::atomic<unsigned> counter{0};
    const char* dir = getenv("CUVS_AMD_SHM_DIR");
    if (dir == nullptr || *dir == 0) dir = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : "/tmp";
    memset(id, 0, CUVS_AMD_SHARD_ID_BYTES);
    memcpy(id, kShmIdMagic, sizeof(kShmIdMagic));
    const auto stamp = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    const int n = snprintf(id + sizeof(kShmIdMagic), CUVS_AMD_SHARD_ID_BYTES - sizeof(kShmIdMagic), "%s/cuvsamd_comm_%ld_%llx_%u", dir,
                           (long)getpid(), stamp, counter.fetch_add(1));
    CUVS_EXPECTS(n > 0 && (size_t)n < CUVS_AMD_SHARD_ID_BYTES - sizeof(kShmIdMagic), "CUVS_AMD_SHM_DIR is too long for a %d-byte id", CUVS_AMD_SHARD_ID_BYTES);
  });
}

cuvsError_t cuvsAmdShardCommCreate(cuvsResources_t res_h, const char id[CUVS_AMD_SHARD_ID_BYTES], int rank, int world,
                                   cuvsAmdShardComm_t* comm)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(world >= 1 && rank >= 0 && rank < world, "shard comm: rank %d of %d", rank, world);
    resources& res = *as_res(res_h);
    HIP_TRY(hipSetDevice(res.device));
    auto c = std::make_unique<cuvsAmdShardComm>();
    c->rank = rank; c->world = world; c->device = res.device;
    if (memcmp(id, kShmIdMagic, sizeof(kShmIdMagic)) == 0) {  // an id of cuvsAmdShardCommGetUniqueIdHostStaged
      CUVS_EXPECTS(memchr(id + sizeof(kShmIdMagic), 0, CUVS_AMD_SHARD_ID_BYTES - sizeof(kShmIdMagic)) != nullptr, "shard comm: malformed host-staged id");
      double limit = 120.0;
      if (const char* e = getenv("CUVS_AMD_SHM_TIMEOUT_S")) limit = std::max(1.0, atof(e));
      try {
        c->shm = std::make_unique<shm_transport>(std::string(id + sizeof(kShmIdMagic)), rank, world, limit);
      } catch (const std::exception& e) { CUVS_FAIL("%s", e.what()); }
    } else {
      ncclUniqueId u;
      memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
      RCCL_TRY(rccl().comm_init_rank(&c->comm, world, u, rank));
    }
    *comm = c.release();
  });
}

cuvsError_t cuvsAmdShardCommDestroy(cuvsAmdShardComm_t comm)
{
  return (cuvsError_t)translate_exceptions([=] {
    if (comm == nullptr) return;
    if (comm->comm != nullptr) RCCL_TRY(rccl().comm_destroy(comm->comm));
    delete comm;
  });
}

cuvsError_t cuvsAmdShardCommRank(cuvsAmdShardComm_t comm, int* rank, int* world)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(comm != nullptr, "null shard communicator");
    *rank = comm->rank; *world = comm->world;
  });
}

cuvsError_t cuvsAmdIvfPqSetListShard(cuvsIvfPqIndex_t index, int rank, int world)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(index != nullptr && index->addr != 0, "IVF-PQ index is not built");
    CUVS_EXPECTS(world >= 1 && rank >= 0 && rank < world, "list shard: rank %d of %d", rank, world);
    auto& idx = *reinterpret_cast<ivf_pq_index*>(index->addr);
    CUVS_EXPECTS(idx.size == 0, "list shard: the index already holds rows (build with add_data_on_build = false)");
    idx.shard_rank = rank; idx.shard_world = world;
    idx.h_list_owner.clear();  // back to "list L on rank L % world": an owner table of an earlier SetListOwners is void
    idx.list_owner = dev_buf<int32_t>();
  });
}

// Greedy longest-processing-time dealing: lists by descending weight (ties: lower id first), each to the rank with the
// smallest load so far (ties: lower rank). Deterministic, so every rank computes the same table from the same weights.
cuvsError_t cuvsAmdShardDealLists(const uint64_t* weights, uint32_t n_lists, int world, int32_t* owners)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(weights != nullptr && owners != nullptr && world >= 1, "deal lists: bad arguments");
    std::vector<uint32_t> order(n_lists);
    for (uint32_t i = 0; i < n_lists; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weights[a] > weights[b]; });
    std::vector<uint64_t> load((size_t)world, 0);
    for (uint32_t L : order) {
      int best = 0;
      for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
      owners[L] = best;
      load[best] += weights[L];
    }
  });
}

cuvsError_t cuvsAmdIvfPqSetListOwners(cuvsIvfPqIndex_t index, const int32_t* owners, uint32_t n_lists, int rank, int world)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(index != nullptr && index->addr != 0, "IVF-PQ index is not built");
    CUVS_EXPECTS(world >= 1 && rank >= 0 && rank < world, "list shard: rank %d of %d", rank, world);
    auto& idx = *reinterpret_cast<ivf_pq_index*>(index->addr);
    CUVS_EXPECTS(idx.size == 0, "list shard: the index already holds rows (build with add_data_on_build = false)");
    CUVS_EXPECTS(owners != nullptr && n_lists == idx.n_lists, "list owners: %u entries for %u lists", n_lists, idx.n_lists);
    for (uint32_t L = 0; L < n_lists; ++L)
      CUVS_EXPECTS(owners[L] >= 0 && owners[L] < world, "list owners: list %u -> rank %d of %d", L, owners[L], world);
    idx.shard_rank = rank; idx.shard_world = world;
    idx.h_list_owner.assign(owners, owners + n_lists);
    idx.list_owner = dev_buf<int32_t>::persistent(n_lists);
    HIP_TRY(hipMemcpy(idx.list_owner.data(), owners, (size_t)n_lists * sizeof(int32_t), hipMemcpyHostToDevice));
  });
}

cuvsError_t cuvsAmdIvfPqListHistogram(cuvsResources_t res_h, cuvsIvfPqIndex_t index, DLManagedTensor* rows, uint64_t* counts)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr && index->addr != 0 && rows != nullptr && counts != nullptr, "list histogram: null argument");
    auto& idx = *reinterpret_cast<ivf_pq_index*>(index->addr);
    auto& t   = rows->dl_tensor;
    CUVS_EXPECTS(t.ndim == 2 && is_c_contiguous(t) && t.shape[1] == idx.dim && is_device_accessible(t),
                 "list histogram: rows must be a device [n, dim] row-major matrix");
    const int64_t n = t.shape[0];
    dev_buf<uint32_t> labels(res, (size_t)std::max<int64_t>(n, 1));
    ivf_pq_transform(res, idx, dl_data(t), elem_of(t.dtype), n, labels.data(), nullptr);
    auto h = to_host(res, labels.data(), (size_t)n);
    for (int64_t i = 0; i < n; ++i) counts[h[i]] += 1;
  });
}

cuvsError_t cuvsAmdIvfPqRowLabels(cuvsResources_t res_h, cuvsIvfPqIndex_t index, DLManagedTensor* rows, uint32_t* labels)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(index != nullptr && index->addr != 0 && rows != nullptr && labels != nullptr, "row labels: null argument");
    auto& idx = *reinterpret_cast<ivf_pq_index*>(index->addr);
    auto& t   = rows->dl_tensor;
    CUVS_EXPECTS(t.ndim == 2 && is_c_contiguous(t) && t.shape[1] == idx.dim && is_device_accessible(t),
                 "row labels: rows must be a device [n, dim] row-major matrix");
    if (t.shape[0] > 0) ivf_pq_transform(res, idx, dl_data(t), elem_of(t.dtype), t.shape[0], labels, nullptr);
  });
}

cuvsError_t cuvsAmdIvfPqSetShardComm(cuvsIvfPqIndex_t index, cuvsAmdShardComm_t comm)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(index != nullptr && index->addr != 0, "IVF-PQ index is not built");
    auto& idx = *reinterpret_cast<ivf_pq_index*>(index->addr);
    CUVS_EXPECTS(comm == nullptr || (comm->world == idx.shard_world && comm->rank == idx.shard_rank),
                 "shard communicator (rank %d of %d) does not match the index's list shard (rank %d of %d)",
                 comm ? comm->rank : 0, comm ? comm->world : 0, idx.shard_rank, idx.shard_world);
    idx.shard_comm = comm;
    idx.shard_stats_valid = false;  // the next search exchanges the shards' rows / list counts (collective, like the search)
  });
}

cuvsError_t cuvsAmdShardTranslateIds(cuvsResources_t res_h, const void* local_ids, int id_bits, int64_t n, int64_t row_offset,
                                     int64_t* global_ids)
{
  return (cuvsError_t)translate_exceptions([=] {
    auto& res = *as_res(res_h);
    CUVS_EXPECTS(id_bits == 32 || id_bits == 64, "translate ids: id_bits must be 32 (uint32) or 64 (int64)");
    if (n == 0) return;
    CUVS_EXPECTS(local_ids != nullptr && global_ids != nullptr, "translate ids: null argument");
    if (id_bits == 32)
      hipLaunchKernelGGL(translate_ids_kernel<uint32_t>, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream,
                         static_cast<const uint32_t*>(local_ids), n, row_offset, global_ids);
    else
      hipLaunchKernelGGL(translate_ids_kernel<int64_t>, dim3(grid_blocks(n, 256)), dim3(256), 0, res.stream,
                         static_cast<const int64_t*>(local_ids), n, row_offset, global_ids);
    HIP_TRY(hipGetLastError());
  });
}

cuvsError_t cuvsAmdShardAllGatherTopK(cuvsResources_t res_h, cuvsAmdShardComm_t comm, const float* local_distances,
                                      const int64_t* local_neighbors, int64_t n_queries, int k, int select_min,
                                      float* distances, int64_t* neighbors)
{
  return (cuvsError_t)translate_exceptions([=] {
    CUVS_EXPECTS(comm != nullptr, "null shard communicator");
    shard_all_gather_topk(*as_res(res_h), *comm, local_distances, local_neighbors, n_queries, k, select_min != 0, distances,
                          neighbors);
  });
}

}  // extern "C"
