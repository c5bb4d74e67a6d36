// The handle's scratch cache: scratch blocks a call frees are kept and handed to the next request of exactly
// that size. A search allocates the same temporaries every batch, and hipFreeAsync on a stream that has just been
// synchronised (the searches that read a flag back) costs ~80 us PER BLOCK on this runtime - 2.5 of the 6.0 ms of an
// IVF-Flat search at the C2 shape (profiles/r03_host_trace_flat.log).
//
// Ordering guarantee = that of hipFreeAsync + hipMallocAsync on one stream: a block is only re-used by work queued later
// on the stream it was used on; requests on any other stream bypass the cache.
//
// No HIP types in here (streams are opaque pointers, the runtime's allocator comes in as two callables): the bookkeeping
// is unit-tested on the CPU (tests/cpp/scratch_cache_test.cpp).
#pragma once

#include <cstddef>
#include <mutex>
#include <unordered_map>

namespace cuvs_amd {

struct scratch_cache {
  std::mutex mu;
  void* stream = nullptr;                              // the stream the kept blocks are ordered on
  bool caller_stream = false;                          // `stream` belongs to the caller (cuvsStreamSet): it may be gone by the time the
                                                       // kept blocks are given back outside a call - those sites free synchronously
  std::unordered_map<void*, size_t> live;              // blocks handed out on `stream`
  std::unordered_multimap<size_t, void*> free_blocks;  // kept blocks by exact size
  size_t cached_bytes = 0;
  size_t cap_bytes    = 0;                  // beyond it the cache is emptied
  size_t max_block    = size_t(4) << 30;    // larger blocks (build-time buffers) are never kept (4 GiB: the B-operand
                                            // buffer of a list-sharded search over 8 ranks - 80k queries x 128 probes x 256 B - is kept)

  // raw_free(stream, pointer): gives a block back to the runtime, ordered on `stream`
  template <class RawFree>
  void flush_locked(RawFree&& raw_free)
  {
    for (auto& kv : free_blocks) raw_free(stream, kv.second);
    free_blocks.clear();
    cached_bytes = 0;
  }

  template <class RawFree>
  void flush(RawFree&& raw_free)
  {
    std::lock_guard<std::mutex> lk(mu);
    flush_locked(raw_free);
  }

  // raw_alloc(bytes, failed): a block from the runtime on the caller's stream; with `failed` non-null a failure is
  // reported through it (nullptr returned), otherwise the callable falls back or throws on its own
  template <class RawAlloc, class RawFree>
  void* alloc(void* on_stream, size_t bytes, RawAlloc&& raw_alloc, RawFree&& raw_free)
  {
    std::lock_guard<std::mutex> lk(mu);
    const bool cached_path = on_stream == stream && bytes <= max_block;
    if (cached_path) {
      auto it = free_blocks.find(bytes);
      if (it != free_blocks.end()) {
        void* p = it->second;
        free_blocks.erase(it);
        cached_bytes -= bytes;
        live[p] = bytes;
        return p;
      }
    }
    bool failed = false;
    void* p = raw_alloc(bytes, cached_bytes > 0 ? &failed : nullptr);
    if (failed) {  // the kept blocks may be what is in the way
      flush_locked(raw_free);
      p = raw_alloc(bytes, nullptr);
    }
    if (cached_path) live[p] = bytes;  // (overwrites a stale entry of an address the caller released behind our back)
    return p;
  }

  template <class RawFree>
  void release(void* on_stream, void* p, RawFree&& raw_free)
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = live.find(p);
      if (it != live.end()) {
        const size_t bytes = it->second;
        live.erase(it);
        if (on_stream == stream) {
          if (cached_bytes + bytes > cap_bytes) flush_locked(raw_free);
          if (bytes <= cap_bytes) {
            free_blocks.emplace(bytes, p);
            cached_bytes += bytes;
            return;
          }
        }
      }
    }
    raw_free(on_stream, p);
  }
};

}  // namespace cuvs_amd
