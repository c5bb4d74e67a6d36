// Host-staged transport for the shard communicator: the collectives of shard_comm.hip (all-gather of equal blocks,
// all-reduce min of uint32 keys) between PROCESSES OF ONE HOST through a shared memory-mapped file.
//
// Why it exists: RCCL refuses two ranks on one device, so a box with ONE GPU can never run the product's rank-dependent
// control flow (head-bound all-reduce, probe all-gather, result all-gather + merge, list ownership) with world > 1. With
// this transport two processes share device 0 and run exactly the code an 8-GPU node runs - only the bytes travel
// device -> mapped file -> device instead of over xGMI. It is also the fallback where RCCL is not available at all.
// Reference counterpart: the reference has ONE transport (NCCL through raft::comms, cpp/src/neighbors/mg/snmg.cuh:283-341)
// and its multi-GPU tests need real devices (cpp/tests/neighbors/mg.cuh:647).
//
// This header is HIP-free (unit-tested on the CPU under ASan: tests/cpp/shm_transport_test.cpp). Layout of the file:
//   [ header: 4 KiB ][ rank 0 block | rank 1 block | ... ]   (block = capacity bytes, grown on demand)
// Every collective is: write own block -> barrier -> read all blocks -> barrier. The barrier is a generation counter in
// the header (lock-free 32-bit atomics work across processes on a MAP_SHARED mapping); a rank that waits longer than the
// time limit marks the segment failed and every rank raises instead of hanging.
#pragma once

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace cuvs_amd {

constexpr char kShmIdMagic[16] = {'C', 'U', 'V', 'S', 'A', 'M', 'D', '-', 'S', 'H', 'M', '-', 'v', '1', 0, 0};
constexpr size_t kShmHeaderBytes = 4096;
constexpr int kShmMaxRanks       = 64;

struct shm_header {
  std::atomic<uint32_t> ready;       // 1 once the creating rank has initialised the header
  std::atomic<uint32_t> arrived;     // ranks that reached the current barrier
  std::atomic<uint32_t> generation;  // completed barriers
  std::atomic<uint32_t> failed;      // a rank gave up (time limit, mismatch): every rank raises
  std::atomic<uint32_t> attached;    // ranks that mapped the file
  uint32_t world;
  uint64_t capacity;                 // bytes per rank block
  struct { uint64_t op, bytes; } want[kShmMaxRanks];  // what every rank believes the current collective is
};
static_assert(sizeof(shm_header) <= kShmHeaderBytes, "header does not fit its page");
static_assert(std::atomic<uint32_t>::is_always_lock_free, "the barrier needs lock-free 32-bit atomics");

class shm_transport {
 public:
  // `path`: a file every rank can open (default directory /dev/shm). The first rank to arrive creates it.
  shm_transport(const std::string& path, int rank, int world, double timeout_s = 120.0)
    : path_(path), rank_(rank), world_(world), timeout_s_(timeout_s)
  {
    if (world < 1 || world > kShmMaxRanks || rank < 0 || rank >= world)
      throw std::runtime_error("shm transport: rank " + std::to_string(rank) + " of " + std::to_string(world));
    bool creator = false;
    try {
      attach(creator);
    } catch (...) {  // (a constructor that throws runs no destructor: give back what was taken, and the name if it is ours)
      release();
      if (creator) ::unlink(path_.c_str());
      throw;
    }
  }
  ~shm_transport() { release(); }
  shm_transport(const shm_transport&)            = delete;
  shm_transport& operator=(const shm_transport&) = delete;

 private:
  void release()
  {
    if (data_) ::munmap(data_, mapped_);
    if (hdr_) ::munmap(hdr_, kShmHeaderBytes);
    if (fd_ >= 0) ::close(fd_);
    data_ = nullptr; hdr_ = nullptr; fd_ = -1;
  }
  void attach(bool& creator)
  {
    const std::string& path = path_;
    const int world = world_;
    fd_ = ::open(path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd_ >= 0) {
      creator = true;
      if (::ftruncate(fd_, (off_t)kShmHeaderBytes) != 0) fail_errno("ftruncate");
    } else if (errno == EEXIST) {
      // the name comes from a 128-byte id another process handed over, and the fallback directory (/tmp) is world-writable: no
      // symlinks are followed, and what is mapped must be a regular file of this user with the creator's mode
      fd_ = ::open(path.c_str(), O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0600);
      if (fd_ < 0) fail_errno("open");
      {
        struct stat st0;
        if (::fstat(fd_, &st0) != 0) fail_errno("fstat");
        if (!S_ISREG(st0.st_mode) || st0.st_uid != ::geteuid() || (st0.st_mode & 0777) != 0600)
          throw std::runtime_error("shm transport: " + path + " is not a regular 0600 file of this user - refusing to map it");
      }
      // wait until the creator has sized the header page
      const auto t0 = now();
      for (;;) {
        struct stat st;
        if (::fstat(fd_, &st) != 0) fail_errno("fstat");
        if ((size_t)st.st_size >= kShmHeaderBytes) break;
        if (elapsed(t0) > timeout_s_) throw std::runtime_error("shm transport: " + path + " was never initialised");
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
    } else {
      fail_errno("open");
    }
    hdr_ = static_cast<shm_header*>(::mmap(nullptr, kShmHeaderBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0));
    if (hdr_ == MAP_FAILED) { hdr_ = nullptr; fail_errno("mmap header"); }
    if (creator) {
      hdr_->world    = (uint32_t)world;
      hdr_->capacity = 0;
      hdr_->ready.store(1, std::memory_order_release);
    } else {
      const auto t0 = now();
      while (hdr_->ready.load(std::memory_order_acquire) != 1) {
        if (elapsed(t0) > timeout_s_) throw std::runtime_error("shm transport: header of " + path + " never became ready");
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (hdr_->world != (uint32_t)world)
        throw std::runtime_error("shm transport: " + path + " was created for another world size");
    }
    hdr_->attached.fetch_add(1, std::memory_order_acq_rel);
    barrier();  // every rank holds an open descriptor from here on:
    if (rank_ == 0) ::unlink(path.c_str());  // the name can go (no stale file after a crash), the inode lives on
  }

 public:
  int rank() const { return rank_; }
  int world() const { return world_; }

  // The rank's block for a collective of `bytes` per rank, after checking that every rank is in the same collective.
  // Call order of one collective: begin() -> fill own() -> publish() -> read block(r) of every rank -> end().
  void begin(uint64_t op, size_t bytes)
  {
    check_failed();
    hdr_->want[rank_].op    = op;
    hdr_->want[rank_].bytes = bytes;
    barrier();
    for (int r = 0; r < world_; ++r)
      if (hdr_->want[r].op != op || hdr_->want[r].bytes != bytes) {
        hdr_->failed.store(1, std::memory_order_release);
        throw std::runtime_error("shm transport: rank " + std::to_string(rank_) + " is in collective " + std::to_string(op) + " / " +
                                 std::to_string(bytes) + " bytes, rank " + std::to_string(r) + " in " +
                                 std::to_string(hdr_->want[r].op) + " / " + std::to_string(hdr_->want[r].bytes) +
                                 " (the ranks' control flow diverged)");
      }
    ensure_capacity(bytes);
  }
  char* own() { return block(rank_); }
  char* block(int r) { return data_ + (size_t)r * capacity_; }
  void publish() { barrier(); }  // every block is written
  void end() { barrier(); }      // every block is read: the next collective may overwrite
  // a rank that leaves a collective by an exception (a HIP error between begin() and end()) tells its peers: they raise at their
  // next wait instead of spinning until the time limit
  void mark_failed() noexcept { if (hdr_ != nullptr) hdr_->failed.store(1, std::memory_order_release); }

  // host-side forms (what the CPU unit test drives; shard_comm.hip stages device buffers around the same steps)
  void all_gather(const void* send, void* recv, size_t bytes)
  {
    begin(1, bytes);
    std::memcpy(own(), send, bytes);
    publish();
    for (int r = 0; r < world_; ++r) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, block(r), bytes);
    end();
  }
  void all_reduce_min_u32(uint32_t* keys, size_t count)
  {
    begin(2, count * sizeof(uint32_t));
    std::memcpy(own(), keys, count * sizeof(uint32_t));
    publish();
    for (int r = 0; r < world_; ++r) {
      const uint32_t* b = reinterpret_cast<const uint32_t*>(block(r));
      for (size_t i = 0; i < count; ++i) keys[i] = b[i] < keys[i] ? b[i] : keys[i];
    }
    end();
  }

  void barrier()
  {
    check_failed();
    const uint32_t gen = hdr_->generation.load(std::memory_order_acquire);
    if (hdr_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
      hdr_->arrived.store(0, std::memory_order_relaxed);
      hdr_->generation.store(gen + 1, std::memory_order_release);
      return;
    }
    const auto t0 = now();
    unsigned spins = 0;
    while (hdr_->generation.load(std::memory_order_acquire) == gen) {
      if (hdr_->failed.load(std::memory_order_acquire)) throw std::runtime_error("shm transport: a peer rank failed");
      if (++spins < 2000) { std::this_thread::yield(); continue; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spins & 1023) == 0 && elapsed(t0) > timeout_s_) {
        hdr_->failed.store(1, std::memory_order_release);
        throw std::runtime_error("shm transport: rank " + std::to_string(rank_) + " of " + std::to_string(world_) + " waited " +
                                 std::to_string((int)timeout_s_) + " s at a barrier (a peer never reached this collective)");
      }
    }
  }

 private:
  using clock = std::chrono::steady_clock;
  static clock::time_point now() { return clock::now(); }
  static double elapsed(clock::time_point t0) { return std::chrono::duration<double>(now() - t0).count(); }
  [[noreturn]] void fail_errno(const char* what)
  {
    throw std::runtime_error(std::string("shm transport: ") + what + " " + path_ + ": " + std::strerror(errno));
  }
  void check_failed()
  {
    if (hdr_->failed.load(std::memory_order_acquire)) throw std::runtime_error("shm transport: a peer rank failed");
  }
  // Same decision on every rank (bytes is checked to be equal): rank 0 grows the file between two barriers, then
  // every rank maps the new size.
  void ensure_capacity(size_t bytes)
  {
    if (bytes <= capacity_ && data_ != nullptr) return;
    size_t cap = capacity_ ? capacity_ : (size_t)1 << 16;
    while (cap < bytes) cap *= 2;
    if (rank_ == 0) {
      const off_t total = (off_t)(kShmHeaderBytes + cap * (size_t)world_);
      // posix_fallocate reserves the pages now: a full /dev/shm is an error here, not a SIGBUS on first touch
      const int rc = ::posix_fallocate(fd_, 0, total);
      if (rc != 0) {
        hdr_->failed.store(1, std::memory_order_release);
        throw std::runtime_error("shm transport: cannot reserve " + std::to_string((long long)total) + " bytes for " + path_ + ": " +
                                 std::strerror(rc) + " (set CUVS_AMD_SHM_DIR to a directory with more room)");
      }
      hdr_->capacity = cap;
    }
    barrier();
    if (data_) ::munmap(data_, mapped_);
    mapped_ = cap * (size_t)world_;
    data_   = static_cast<char*>(::mmap(nullptr, mapped_, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, (off_t)kShmHeaderBytes));
    if (data_ == MAP_FAILED) { data_ = nullptr; hdr_->failed.store(1, std::memory_order_release); fail_errno("mmap blocks"); }
    capacity_ = cap;
    barrier();
  }

  std::string path_;
  int rank_, world_;
  double timeout_s_;
  int fd_          = -1;
  shm_header* hdr_ = nullptr;
  char* data_      = nullptr;
  size_t mapped_ = 0, capacity_ = 0;
};

}  // namespace cuvs_amd
