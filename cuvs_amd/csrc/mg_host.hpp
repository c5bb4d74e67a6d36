// Host-side arithmetic of the multi-GPU layer (mg.hip): which rows a rank owns, which query batches a replica serves,
// how the per-shard result lists of a sharded search are merged. Plain C++ (no HIP), unit-tested on the CPU by
// tests/test_mg_host_cpu.py.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <utility>
#include <vector>

namespace cuvs_amd {

constexpr int64_t kNoNeighbor = INT64_MAX;  // ivf_common.cuh:31 kOutOfBoundsRecord, what the single-GPU searches emit

// rows [r0, r0 + cnt) of rank r. Replicated: everything. Sharded: ceil(n / R) rows per rank, the tail ranks get what
// is left, possibly nothing (snmg.cuh:131-150).
inline void rows_of_rank(bool sharded, int64_t n, int r, int n_ranks, int64_t* r0, int64_t* cnt)
{
  if (!sharded) {
    *r0  = 0;
    *cnt = n;
    return;
  }
  const int64_t per = (n + n_ranks - 1) / n_ranks;
  *r0               = std::min<int64_t>(n, (int64_t)r * per);
  *cnt              = std::max<int64_t>(0, std::min<int64_t>(per, n - *r0));
}

// Replicated search, load balancer (snmg.cuh:596-632): batches of at most n_rows_per_batch queries, but at least one
// batch per rank; batch b is served by rank b mod R. A single batch covers the whole call.
inline void replicated_batches(int64_t n_queries, int64_t n_rows_per_batch, int n_ranks, int64_t* batch, int64_t* n_batches)
{
  *batch     = std::max<int64_t>(1, std::min(n_rows_per_batch, (n_queries + n_ranks - 1) / n_ranks));
  *n_batches = (n_queries + *batch - 1) / *batch;
  if (*n_batches <= 1) {
    *batch     = n_queries;
    *n_batches = 1;
  }
}

// Sharded search (snmg.cuh:696-699): every rank searches every batch
inline void sharded_batches(int64_t n_queries, int64_t n_rows_per_batch, int64_t* batch, int64_t* n_batches)
{
  *batch     = std::max<int64_t>(1, n_rows_per_batch);
  *n_batches = (n_queries + *batch - 1) / *batch;
  if (*n_batches <= 1) {
    *batch     = n_queries;
    *n_batches = 1;
  }
}

// k best of the R per-shard lists of every query. part_* are [R][cnt][k]; shard-local ids move by `translation[r]` =
// rows held by the lower ranks (knn_merge_parts.cuh:27-103 with translations, snmg.cuh:340-347); entries without a
// neighbour (negative id or kNoNeighbor) are dropped; order = (distance, id), descending distance for inner product;
// missing tail entries are kNoNeighbor with the worst distance.
inline void merge_on_host(const int64_t* part_i, const float* part_d, int n_ranks, int64_t cnt, int64_t k,
                          const int64_t* translation, bool select_min, int64_t* out_i, float* out_d)
{
  std::vector<std::pair<float, int64_t>> cand;
  cand.reserve((size_t)n_ranks * k);
  auto better = [select_min](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
    if (a.first != b.first) return select_min ? a.first < b.first : a.first > b.first;
    return a.second < b.second;
  };
  for (int64_t q = 0; q < cnt; ++q) {
    cand.clear();
    for (int r = 0; r < n_ranks; ++r) {
      const size_t base = ((size_t)r * cnt + q) * k;
      for (int64_t j = 0; j < k; ++j) {
        const int64_t id = part_i[base + j];
        if (id < 0 || id == kNoNeighbor) continue;
        if (part_d[base + j] != part_d[base + j]) continue;  // NaN: not a strict weak order for partial_sort
        cand.emplace_back(part_d[base + j], id + translation[r]);
      }
    }
    const size_t keep = std::min<size_t>(cand.size(), (size_t)k);
    std::partial_sort(cand.begin(), cand.begin() + keep, cand.end(), better);
    for (size_t j = 0; j < (size_t)k; ++j) {
      out_i[q * k + j] = j < keep ? cand[j].second : kNoNeighbor;
      out_d[q * k + j] = j < keep ? cand[j].first : (select_min ? FLT_MAX : -FLT_MAX);
    }
  }
}

}  // namespace cuvs_amd
