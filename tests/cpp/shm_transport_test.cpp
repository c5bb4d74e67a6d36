// CPU unit test of the host-staged shard transport (cuvs_amd/csrc/shm_transport.hpp): three PROCESSES (fork) run
// all-gathers of growing block sizes and min all-reduces, then the two failure modes that must raise instead of hang:
// ranks in different collectives, and a rank that never arrives.
#include "shm_transport.hpp"

#include <cstdlib>
#include <vector>

#include <sys/wait.h>

using cuvs_amd::shm_transport;

static int rank_main(const std::string& path, int rank, int world)
{
  shm_transport t(path, rank, world, 20.0);
  // all-gather, sizes crossing the initial capacity twice (the file is grown between collectives)
  for (size_t bytes : {size_t(8), size_t(1000), size_t(70000), size_t(3) << 20, size_t(100)}) {
    std::vector<unsigned char> send(bytes), recv(bytes * world);
    for (size_t i = 0; i < bytes; ++i) send[i] = (unsigned char)(rank * 31 + i * 7);
    t.all_gather(send.data(), recv.data(), bytes);
    for (int r = 0; r < world; ++r)
      for (size_t i = 0; i < bytes; ++i)
        if (recv[(size_t)r * bytes + i] != (unsigned char)(r * 31 + i * 7)) {
          fprintf(stderr, "rank %d: all_gather byte %zu of rank %d wrong\n", rank, i, r);
          return 1;
        }
  }
  // all-reduce min
  for (size_t n : {size_t(1), size_t(1000), size_t(300000)}) {
    std::vector<uint32_t> k(n);
    for (size_t i = 0; i < n; ++i) k[i] = (uint32_t)((i * 2654435761u) ^ (uint32_t)(rank * 0x9e3779b9u));
    t.all_reduce_min_u32(k.data(), n);
    for (size_t i = 0; i < n; ++i) {
      uint32_t m = 0xffffffffu;
      for (int r = 0; r < world; ++r) {
        const uint32_t v = (uint32_t)((i * 2654435761u) ^ (uint32_t)(r * 0x9e3779b9u));
        m = v < m ? v : m;
      }
      if (k[i] != m) { fprintf(stderr, "rank %d: all_reduce key %zu wrong\n", rank, i); return 1; }
    }
  }
  return 0;
}

// ranks whose control flow diverged: rank 1 all-reduces while the others all-gather -> every rank raises
static int diverging_main(const std::string& path, int rank, int world)
{
  shm_transport t(path, rank, world, 20.0);
  try {
    std::vector<uint32_t> k(16, 1u), out(16 * world);
    if (rank == 1) t.all_reduce_min_u32(k.data(), 16);
    else           t.all_gather(k.data(), out.data(), 64);
  } catch (const std::exception& e) {
    return std::string(e.what()).find("shm transport") != std::string::npos ? 0 : 1;
  }
  return 1;  // must not succeed
}

// a rank that never arrives: the others give up after the time limit
static int missing_main(const std::string& path, int rank, int world)
{
  try {
    shm_transport t(path, rank, world, 1.0);
    if (rank == world - 1) return 0;  // leaves before the collective
    std::vector<uint32_t> k(4, 1u);
    t.all_reduce_min_u32(k.data(), 4);
  } catch (const std::exception& e) {
    return std::string(e.what()).find("waited") != std::string::npos || std::string(e.what()).find("peer rank failed") != std::string::npos ? 0 : 1;
  }
  return 1;
}

static int run(int (*fn)(const std::string&, int, int), const char* tag, int world)
{
  const std::string path = std::string("/tmp/cuvsamd_shmtest_") + tag + "_" + std::to_string((long)getpid());
  ::unlink(path.c_str());
  std::vector<pid_t> kids;
  for (int r = 0; r < world; ++r) {
    pid_t p = fork();
    if (p == 0) {
      int rc = 1;
      try { rc = fn(path, r, world); } catch (const std::exception& e) { fprintf(stderr, "rank %d: %s\n", r, e.what()); }
      _exit(rc);
    }
    kids.push_back(p);
  }
  int bad = 0;
  for (pid_t p : kids) {
    int st = 0;
    waitpid(p, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
  }
  ::unlink(path.c_str());
  if (bad) fprintf(stderr, "%s: %d of %d ranks failed\n", tag, bad, world);
  return bad;
}

int main()
{
  if (run(rank_main, "collectives", 3)) return 1;
  if (run(rank_main, "one_rank", 1)) return 1;
  if (run(diverging_main, "diverging", 3)) return 1;
  if (run(missing_main, "missing", 2)) return 1;
  printf("shm transport OK\n");
  return 0;
}
