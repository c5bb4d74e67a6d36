// CPU unit test of cuvs_amd/csrc/scratch_cache.hpp (the bookkeeping behind device_alloc / device_free) with a
// counting stand-in for the runtime's allocator. Built and run by tests/test_scratch_cache_cpu.py.
#include "scratch_cache.hpp"

#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

using cuvs_amd::scratch_cache;

static int g_fail = 0;
#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
  } while (0)

struct fake_runtime {
  std::set<void*> owned;       // blocks the "runtime" has handed out and not got back
  size_t n_alloc = 0, n_free = 0;
  size_t budget  = size_t(1) << 40;  // bytes the runtime is willing to hold
  size_t held    = 0;
  std::unordered_map<void*, size_t> sizes;
  std::vector<void*> free_streams;   // stream of every raw free

  void* alloc(size_t bytes, bool* failed)
  {
    if (held + bytes > budget) {
      if (failed != nullptr) { *failed = true; return nullptr; }
      std::printf("hard allocation failure\n");
      std::exit(2);
    }
    void* p = std::malloc(bytes ? 16 : 16);
    owned.insert(p);
    sizes[p] = bytes;
    held += bytes;
    ++n_alloc;
    return p;
  }
  void free(void* stream, void* p)
  {
    CHECK(owned.count(p) == 1);  // never a double free, never a foreign pointer
    owned.erase(p);
    held -= sizes[p];
    free_streams.push_back(stream);
    ++n_free;
    std::free(p);
  }
};

int main()
{
  int s1 = 0, s2 = 0;  // two "streams"
  fake_runtime rt;
  auto ra = [&](size_t n, bool* failed) { return rt.alloc(n, failed); };
  auto rf = [&](void* s, void* p) { rt.free(s, p); };

  {  // exact-size re-use, no runtime calls in the steady state
    scratch_cache c;
    c.stream = &s1; c.cap_bytes = 1 << 20;
    void* a = c.alloc(&s1, 1000, ra, rf);
    c.release(&s1, a, rf);
    CHECK(rt.n_free == 0 && c.cached_bytes == 1000);
    void* b = c.alloc(&s1, 1000, ra, rf);
    CHECK(b == a && rt.n_alloc == 1 && c.cached_bytes == 0);
    void* d = c.alloc(&s1, 1000, ra, rf);   // b is live
    CHECK(d != b && rt.n_alloc == 2);
    void* e = c.alloc(&s1, 1001, ra, rf);   // another size
    CHECK(rt.n_alloc == 3);
    c.release(&s1, d, rf); c.release(&s1, b, rf); c.release(&s1, e, rf);
    CHECK(rt.n_free == 0 && c.cached_bytes == 3001 && c.live.empty());
    const size_t before = rt.n_alloc;
    for (int it = 0; it < 100; ++it) {      // a "search": the same three buffers every call
      void* x = c.alloc(&s1, 1000, ra, rf); void* y = c.alloc(&s1, 1000, ra, rf); void* z = c.alloc(&s1, 1001, ra, rf);
      CHECK(x != y && y != z && x != z);
      c.release(&s1, z, rf); c.release(&s1, x, rf); c.release(&s1, y, rf);
    }
    CHECK(rt.n_alloc == before && rt.n_free == 0);
    c.flush(rf);
    CHECK(rt.n_free == 3 && c.cached_bytes == 0 && c.free_blocks.empty() && rt.owned.empty());
    for (void* s : rt.free_streams) CHECK(s == &s1);
  }
  rt = fake_runtime();
  {  // another stream bypasses the cache in both directions
    scratch_cache c;
    c.stream = &s1; c.cap_bytes = 1 << 20;
    void* a = c.alloc(&s1, 512, ra, rf);
    c.release(&s1, a, rf);                  // kept for s1
    void* b = c.alloc(&s2, 512, ra, rf);    // s2 does not get it
    CHECK(b != a && rt.n_alloc == 2 && c.live.count(b) == 0);
    c.release(&s2, b, rf);                  // straight back to the runtime, on s2
    CHECK(rt.n_free == 1 && rt.free_streams.back() == &s2 && c.cached_bytes == 512);
    void* d = c.alloc(&s1, 512, ra, rf);    // a block handed out on s1 ...
    CHECK(d == a);
    c.release(&s2, d, rf);                  // ... but released through a copy of the handle on s2: not kept
    CHECK(rt.n_free == 2 && c.cached_bytes == 0 && c.live.empty());
    c.flush(rf);
    CHECK(rt.owned.empty());
  }
  rt = fake_runtime();
  {  // blocks above max_block are never tracked; the size limit empties the cache; blocks above the limit are not kept
    scratch_cache c;
    c.stream = &s1; c.cap_bytes = 3000; c.max_block = 2000;
    void* big = c.alloc(&s1, 2001, ra, rf);
    CHECK(c.live.empty());
    c.release(&s1, big, rf);
    CHECK(rt.n_free == 1 && c.cached_bytes == 0);
    void* a = c.alloc(&s1, 1500, ra, rf); void* b = c.alloc(&s1, 1400, ra, rf); void* d = c.alloc(&s1, 300, ra, rf);
    c.release(&s1, a, rf); c.release(&s1, b, rf);
    CHECK(c.cached_bytes == 2900 && rt.n_free == 1);
    c.release(&s1, d, rf);                  // 3200 > 3000: a and b go back, d is kept
    CHECK(rt.n_free == 3 && c.cached_bytes == 300 && c.free_blocks.size() == 1);
    c.cap_bytes = 100;
    void* e = c.alloc(&s1, 200, ra, rf);
    c.release(&s1, e, rf);                  // 500 > 100: d goes back; e itself is above the limit: back as well
    CHECK(c.cached_bytes == 0 && c.free_blocks.empty() && rt.owned.empty());
  }
  rt = fake_runtime();
  {  // an allocation failure empties the cache and retries
    scratch_cache c;
    c.stream = &s1; c.cap_bytes = 1 << 20;
    rt.budget = 5000;
    void* a = c.alloc(&s1, 3000, ra, rf);
    c.release(&s1, a, rf);                  // kept: the runtime still holds 3000
    void* b = c.alloc(&s1, 4000, ra, rf);   // 3000 + 4000 > 5000: first attempt fails, the kept block is given back
    CHECK(b != nullptr && rt.held == 4000 && c.cached_bytes == 0 && rt.n_free == 1);
    c.release(&s1, b, rf);
    c.flush(rf);
    CHECK(rt.owned.empty());
  }
  rt = fake_runtime();
  {  // an address released behind the cache's back and re-issued with another size does not keep its stale size
    scratch_cache c;
    c.stream = &s1; c.cap_bytes = 1 << 20;
    void* a = c.alloc(&s1, 100, ra, rf);
    c.live[a] = 100;                        // (as left by a caller that freed `a` with the runtime directly)
    c.live.erase(a); c.live[a] = 100;
    // simulate: the runtime re-issues the same address for a larger request
    size_t n0 = rt.n_alloc;
    auto ra_same = [&](size_t n, bool*) { rt.sizes[a] = n; ++rt.n_alloc; return a; };
    void* b = c.alloc(&s1, 700, ra_same, rf);
    CHECK(b == a && rt.n_alloc == n0 + 1 && c.live[a] == 700);
    c.release(&s1, b, rf);
    CHECK(c.free_blocks.count(700) == 1 && c.free_blocks.count(100) == 0);
    c.flush(rf);
  }
  {  // changing the stream (cuvsStreamSet: flush, then re-target): blocks still out are kept for the new stream
    rt = fake_runtime();
    scratch_cache c;
    c.stream = &s1; c.cap_bytes = 1 << 20;
    void* a = c.alloc(&s1, 64, ra, rf); void* b = c.alloc(&s1, 64, ra, rf);
    c.release(&s1, a, rf);
    c.flush(rf);
    CHECK(rt.n_free == 1 && rt.free_streams.back() == &s1);
    c.stream = &s2;
    c.release(&s2, b, rf);                  // handed out before the change, released after it
    CHECK(c.cached_bytes == 64 && rt.n_free == 1);
    void* d = c.alloc(&s2, 64, ra, rf);
    CHECK(d == b);
    c.release(&s2, d, rf);
    c.flush(rf);
    CHECK(rt.owned.empty() && rt.free_streams.back() == &s2);
  }
  if (g_fail == 0) std::printf("scratch cache OK\n");
  return g_fail == 0 ? 0 : 1;
}
