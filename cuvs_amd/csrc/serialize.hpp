// Index (de)serialization container shared by the four index types.
// File = magic "CUVSAMD1", kind, version, then index-specific scalars and length-prefixed arrays.
// This is this library's own format (raw device layout, round-trips every index exactly); it is written only
// when CUVS_AMD_NATIVE_FORMAT=1. The default is the reference's numpy-record format (npy_io.hpp);
// *Deserialize recognises both by the magic.
#pragma once
#include "common.hpp"

#include <cstdio>

namespace cuvs_amd {

enum index_kind : uint32_t { KIND_BRUTE_FORCE = 1, KIND_IVF_FLAT = 2, KIND_IVF_PQ = 3, KIND_CAGRA = 4 };
constexpr uint32_t kSerialVersion = 1;

struct file_writer {
  FILE* f = nullptr;
  file_writer(const char* name, uint32_t kind)
  {
    CUVS_EXPECTS(name != nullptr, "filename is null");
    f = fopen(name, "wb");
    CUVS_EXPECTS(f != nullptr, "Cannot open file %s", name);
    try {  // a constructor that throws never runs the destructor: close the file here
      raw("CUVSAMD1", 8);
      scalar<uint32_t>(kind);
      scalar<uint32_t>(kSerialVersion);
    } catch (...) {
      fclose(f);
      f = nullptr;
      throw;
    }
  }
  ~file_writer() { if (f) fclose(f); }
  void raw(const void* p, size_t n) { CUVS_EXPECTS(fwrite(p, 1, n, f) == n, "short write"); }
  template <typename T>
  void scalar(T v) { raw(&v, sizeof(T)); }
  void device_array(resources& res, const void* d, size_t bytes)
  {
    scalar<uint64_t>(bytes);
    std::vector<char> h(bytes);
    if (bytes) {
      copy_async(res, h.data(), d, bytes);
      sync(res);
      raw(h.data(), bytes);
    }
  }
};

struct file_reader {
  FILE* f = nullptr;
  file_reader(const char* name, uint32_t kind)
  {
    CUVS_EXPECTS(name != nullptr, "filename is null");
    f = fopen(name, "rb");
    CUVS_EXPECTS(f != nullptr, "Cannot open file %s", name);
    try {  // a constructor that throws never runs the destructor: close the file here
      char magic[8];
      raw(magic, 8);
      CUVS_EXPECTS(memcmp(magic, "CUVSAMD1", 8) == 0, "%s is not a cuvs_amd index file", name);
      uint32_t k = scalar<uint32_t>(), v = scalar<uint32_t>();
      CUVS_EXPECTS(k == kind, "index kind mismatch in %s (file %u, expected %u)", name, k, kind);
      CUVS_EXPECTS(v == kSerialVersion, "serialization version mismatch: got %u, expected %u", v, kSerialVersion);
    } catch (...) {
      fclose(f);
      f = nullptr;
      throw;
    }
  }
  ~file_reader() { if (f) fclose(f); }
  void raw(void* p, size_t n) { CUVS_EXPECTS(fread(p, 1, n, f) == n, "unexpected end of file"); }
  template <typename T>
  T scalar() { T v; raw(&v, sizeof(T)); return v; }
  template <typename T>
  dev_buf<T> device_array(resources& res)
  {
    uint64_t bytes = scalar<uint64_t>();
    CUVS_EXPECTS(bytes % sizeof(T) == 0, "corrupt array length");
    auto buf = dev_buf<T>::persistent(bytes / sizeof(T));
    if (bytes) {
      std::vector<char> h(bytes);
      raw(h.data(), bytes);
      copy_async(res, buf.data(), h.data(), bytes);
      sync(res);
    }
    return buf;
  }
};

}  // namespace cuvs_amd
