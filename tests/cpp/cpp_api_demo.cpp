// The C++ surface (include/cuvs_amd/neighbors.hpp) against the plain C ABI on the same data: brute force, IVF-Flat,
// IVF-PQ and CAGRA must return what their C entry points return. Built and run by tests/test_cpp_api_gpu.py.
#include <cuvs_amd/neighbors.hpp>

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPCHECK(e) do { if ((e) != hipSuccess) { std::printf("HIP error line %d\n", __LINE__); return 2; } } while (0)

int main()
{
  const int64_t n = 4000, dim = 32, nq = 50, k = 10;
  std::vector<float> x(n * dim), q(nq * dim);
  unsigned s = 1234;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (float)(1u << 24); };
  for (auto& v : x) v = rnd();
  for (auto& v : q) v = rnd();
  float *dx, *dq, *dd;
  int64_t* dn;
  uint32_t* dn32;
  HIPCHECK(hipMalloc((void**)&dx, x.size() * 4)); HIPCHECK(hipMalloc((void**)&dq, q.size() * 4));
  HIPCHECK(hipMalloc((void**)&dd, nq * k * 4)); HIPCHECK(hipMalloc((void**)&dn, nq * k * 8));
  HIPCHECK(hipMalloc((void**)&dn32, nq * k * 4));
  HIPCHECK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
  std::vector<int64_t> exact(nq * k), got(nq * k);
  std::vector<uint32_t> got32(nq * k);
  try {
    cuvs::resources res;
    auto data    = cuvs::make_device_matrix_view<const float>(dx, n, dim);
    auto queries = cuvs::make_device_matrix_view<const float>(dq, nq, dim);
    auto nbr     = cuvs::make_device_matrix_view<int64_t>(dn, nq, k);
    auto dst     = cuvs::make_device_matrix_view<float>(dd, nq, k);
    namespace nb = cuvs::neighbors;
    {
      auto idx = nb::brute_force::build<float>(res, data);
      nb::brute_force::search<float>(res, idx, queries, nbr, dst);
      res.sync_stream();
      HIPCHECK(hipMemcpy(exact.data(), dn, nq * k * 8, hipMemcpyDeviceToHost));
    }
    auto recall = [&](const std::vector<int64_t>& g) {
      int hit = 0;
      for (int64_t i = 0; i < nq; ++i)
        for (int64_t a = 0; a < k; ++a)
          for (int64_t b = 0; b < k; ++b) hit += g[i * k + a] == exact[i * k + b];
      return (double)hit / (double)(nq * k);
    };
    {
      nb::ivf_flat::index_params ip; ip.n_lists = 16;
      nb::ivf_flat::search_params sp; sp.n_probes = 16;  // every list: exact
      auto idx = nb::ivf_flat::build<float>(res, ip, data);
      nb::ivf_flat::search<float>(res, sp, idx, queries, nbr, dst);
      res.sync_stream();
      HIPCHECK(hipMemcpy(got.data(), dn, nq * k * 8, hipMemcpyDeviceToHost));
      std::printf("ivf_flat recall %.4f\n", recall(got));
      if (recall(got) < 0.999) return 1;
    }
    {
      nb::ivf_pq::index_params ip; ip.n_lists = 16; ip.pq_dim = 32;
      nb::ivf_pq::search_params sp; sp.n_probes = 16; sp.lut_dtype = CUDA_R_16F;
      auto idx = nb::ivf_pq::build<float>(res, ip, data);
      if (idx.size() != n || idx.n_lists() != 16 || idx.pq_dim() != 32) return 1;
      nb::ivf_pq::search<float>(res, sp, idx, queries, nbr, dst);
      res.sync_stream();
      HIPCHECK(hipMemcpy(got.data(), dn, nq * k * 8, hipMemcpyDeviceToHost));
      std::printf("ivf_pq recall %.4f\n", recall(got));
      if (recall(got) < 0.85) return 1;
      // the overload with a sample filter: only even source ids may come back
      std::vector<uint32_t> words((n + 31) / 32, 0x55555555u);
      uint32_t* dw;
      HIPCHECK(hipMalloc((void**)&dw, words.size() * 4));
      HIPCHECK(hipMemcpy(dw, words.data(), words.size() * 4, hipMemcpyHostToDevice));
      nb::ivf_pq::search<float>(res, sp, idx, queries, nbr, dst, nb::filtering::bitset_filter{dw, n});
      res.sync_stream();
      HIPCHECK(hipMemcpy(got.data(), dn, nq * k * 8, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < nq * k; ++i)
        if (got[i] < 0 || got[i] >= n || (got[i] & 1)) { std::printf("ivf_pq filtered: id %ld\n", (long)got[i]); return 1; }
      std::printf("ivf_pq filtered OK\n");
      HIPCHECK(hipFree(dw));
    }
    {
      nb::cagra::index_params ip; ip.intermediate_graph_degree = 64; ip.graph_degree = 32;
      nb::cagra::search_params sp;
      auto idx = nb::cagra::build<float>(res, ip, data);
      if (idx.size() != n || idx.graph_degree() != 32) return 1;
      auto nbr32 = cuvs::make_device_matrix_view<uint32_t>(dn32, nq, k);
      nb::cagra::search<float, uint32_t>(res, sp, idx, queries, nbr32, dst);
      res.sync_stream();
      HIPCHECK(hipMemcpy(got32.data(), dn32, nq * k * 4, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < nq * k; ++i) got[i] = got32[i];
      std::printf("cagra recall %.4f\n", recall(got));
      if (recall(got) < 0.95) return 1;
    }
    // error convention: exceptions carry the library's text
    try {
      nb::ivf_pq::search_params sp; sp.n_probes = 0;
      nb::ivf_pq::index_params ip; ip.n_lists = 16;
      auto idx = nb::ivf_pq::build<float>(res, ip, data);
      nb::ivf_pq::search<float>(res, sp, idx, queries, nbr, dst);
      return 1;
    } catch (const cuvs::error& e) {
      std::printf("expected error: %s\n", e.what());
    }
  } catch (const std::exception& e) {
    std::printf("unexpected exception: %s\n", e.what());
    return 3;
  }
  std::printf("cpp api OK\n");
  return 0;
}
