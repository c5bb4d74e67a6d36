// Reader/writer for the reference's on-disk index container.
//
// The reference serializes every scalar and every mdspan through RAFT's numpy serializer
// (raft::serialize_scalar / raft::serialize_mdspan; call sites ivf_pq_serialize.cuh:49-85,
// ivf_flat_serialize.cuh:49-75, cagra_serialize.cuh:55-75, brute_force_serialize.cu:34-45,
// ivf_list.cuh:108-131): each item is a complete NumPy ".npy" v1.0 record - magic "\x93NUMPY", version 1.0,
// little-endian u16 header length, a Python-dict header {'descr', 'fortran_order', 'shape'} padded with spaces
// so that the payload starts on a 64-byte boundary, then the raw little-endian payload. Scalars are 0-d arrays.
// RAFT is not vendored in the reference tree, so the exact whitespace of its header is not pinned here; the
// writer emits standard .npy records (tests/test_serialize_format_gpu.py parses them with an independent
// Python reader; fp16 payloads use RAFT's "<e2" spelling, which numpy itself spells "<f2") and the reader
// parses the dict by key, accepts any padding, and converts scalars by (kind, itemsize) instead of assuming a
// width - so files differing in those details still load.
#pragma once
#include "common.hpp"

#include <cstdio>
#include <string>

namespace cuvs_amd {

struct npy_header {
  char kind         = 'f';
  uint32_t itemsize = 4;
  bool fortran      = false;
  std::vector<int64_t> shape;
  int64_t count() const
  {
    int64_t c = 1;
    for (auto s : shape) c *= s;
    return c;
  }
};

template <typename T>
struct npy_type;
template <> struct npy_type<float> { static constexpr char kind = 'f'; };
template <> struct npy_type<double> { static constexpr char kind = 'f'; };
template <> struct npy_type<__half> { static constexpr char kind = 'e'; };
template <> struct npy_type<int8_t> { static constexpr char kind = 'i'; };
template <> struct npy_type<int32_t> { static constexpr char kind = 'i'; };
template <> struct npy_type<int64_t> { static constexpr char kind = 'i'; };
template <> struct npy_type<uint8_t> { static constexpr char kind = 'u'; };
template <> struct npy_type<uint32_t> { static constexpr char kind = 'u'; };
template <> struct npy_type<uint64_t> { static constexpr char kind = 'u'; };
template <> struct npy_type<bool> { static constexpr char kind = 'u'; };  // is_integral, unsigned, 1 byte

inline std::string npy_descr(char kind, uint32_t itemsize)
{
  std::string s;
  s += itemsize > 1 ? '<' : '|';
  s += kind;  // half is kind 'e' ("<e2") as the reference's C layer expects (c/src/neighbors/ivf_flat.cpp:315)
  s += std::to_string(itemsize);
  return s;
}
// the 4-byte dtype prefix in front of brute-force / IVF-Flat / CAGRA files (ivf_flat_serialize.cuh:49-51:
// numpy dtype string resized to 4 chars => NUL padded), parsed by c/src/neighbors/ivf_flat.cpp:303-328
inline void elem_prefix(elem_t e, char out[4])
{
  const char* s = e == elem_t::f32 ? "<f4" : e == elem_t::f16 ? "<e2" : e == elem_t::i8 ? "|i1" : "|u1";
  memset(out, 0, 4);
  memcpy(out, s, 3);
}
inline bool parse_elem_prefix(const char p[4], elem_t* e)
{
  if ((p[0] != '<' && p[0] != '|' && p[0] != '=') || p[3] != '\0') return false;
  if ((p[1] == 'f') && p[2] == '4') { *e = elem_t::f32; return true; }
  if ((p[1] == 'e' || p[1] == 'f') && p[2] == '2') { *e = elem_t::f16; return true; }
  if (p[1] == 'i' && p[2] == '1') { *e = elem_t::i8; return true; }
  if (p[1] == 'u' && p[2] == '1') { *e = elem_t::u8; return true; }
  return false;
}
inline DLDataType dl_of(elem_t e)
{
  switch (e) {
    case elem_t::f32: return DLDataType{kDLFloat, 32, 1};
    case elem_t::f16: return DLDataType{kDLFloat, 16, 1};
    case elem_t::i8: return DLDataType{kDLInt, 8, 1};
    default: return DLDataType{kDLUInt, 8, 1};
  }
}

// A multi-GPU index file is a short header followed by the streams of its per-device indexes, back to back
// (cpp/src/neighbors/mg/snmg.cuh:735-757). The per-index writers and readers stay filename based; while mg.hip
// holds this window open (same thread), a writer appends instead of truncating and a reader starts at
// `read_offset` and reports where it stopped in `end_offset`.
struct npy_io_window {
  bool append      = false;
  long read_offset = 0;
  long end_offset  = 0;
};
inline thread_local npy_io_window g_npy_io;

class npy_writer {
 public:
  explicit npy_writer(const char* name)
  {
    CUVS_EXPECTS(name != nullptr, "filename is null");
    f_ = fopen(name, g_npy_io.append ? "ab" : "wb");
    CUVS_EXPECTS(f_ != nullptr, "Cannot open file %s", name);
  }
  ~npy_writer() { if (f_) fclose(f_); }
  npy_writer(const npy_writer&) = delete;
  void close()
  {
    int rc = fclose(f_);
    f_     = nullptr;
    CUVS_EXPECTS(rc == 0, "Error writing output");
  }
  void raw(const void* p, size_t n)
  {
    if (n) CUVS_EXPECTS(fwrite(p, 1, n, f_) == n, "short write");
  }
  void header(char kind, uint32_t itemsize, const std::vector<int64_t>& shape)
  {
    std::string d = "{'descr': '" + npy_descr(kind, itemsize) + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); i++) {
      d += std::to_string(shape[i]);
      if (shape.size() == 1) d += ",";
      else if (i + 1 < shape.size()) d += ", ";
    }
    d += "), }";
    size_t preamble = 6 + 2 + 2 + d.size() + 1;
    size_t pad      = (64 - preamble % 64) % 64;
    d.append(pad, ' ');
    d += '\n';
    CUVS_EXPECTS(d.size() < 65536, "npy header too long");
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    raw(magic, 8);
    uint16_t len = (uint16_t)d.size();
    raw(&len, 2);
    raw(d.data(), d.size());
  }
  template <typename T>
  void scalar(T v)
  {
    header(npy_type<T>::kind, sizeof(T), {});
    raw(&v, sizeof(T));
  }
  template <typename T>
  void host_array(const T* p, const std::vector<int64_t>& shape)
  {
    header(npy_type<T>::kind, sizeof(T), shape);
    int64_t c = 1;
    for (auto s : shape) c *= s;
    raw(p, (size_t)c * sizeof(T));
  }
  // device payload streamed through a bounded host bounce buffer; rows of `row_bytes` with device pitch
  void device_array(resources& res, char kind, uint32_t itemsize, const std::vector<int64_t>& shape, const void* d,
                    size_t row_bytes = 0, size_t pitch_bytes = 0)
  {
    header(kind, itemsize, shape);
    int64_t c = 1;
    for (auto s : shape) c *= s;
    size_t bytes = (size_t)c * itemsize;
    if (bytes == 0) return;
    if (row_bytes == 0) row_bytes = pitch_bytes = bytes;
    size_t rows          = bytes / row_bytes;
    size_t rows_per_step = std::max<size_t>(1, (size_t(256) << 20) / row_bytes);
    std::vector<char> h(std::min(rows, rows_per_step) * row_bytes);
    for (size_t r0 = 0; r0 < rows; r0 += rows_per_step) {
      size_t nr = std::min(rows_per_step, rows - r0);
      HIP_TRY(hipMemcpy2DAsync(h.data(), row_bytes, static_cast<const char*>(d) + r0 * pitch_bytes, pitch_bytes,
                               row_bytes, nr, hipMemcpyDefault, res.stream));
      sync(res);
      raw(h.data(), nr * row_bytes);
    }
  }

 private:
  FILE* f_ = nullptr;
};

class npy_reader {
 public:
  explicit npy_reader(const char* name) : name_(name ? name : "")
  {
    CUVS_EXPECTS(name != nullptr, "filename is null");
    f_ = fopen(name, "rb");
    CUVS_EXPECTS(f_ != nullptr, "Cannot open file %s", name);
    if (g_npy_io.read_offset > 0) CUVS_EXPECTS(fseek(f_, g_npy_io.read_offset, SEEK_SET) == 0, "Cannot seek in %s", name);
  }
  ~npy_reader()
  {
    if (f_) {
      g_npy_io.end_offset = ftell(f_);
      fclose(f_);
    }
  }
  long tell() const { return ftell(f_); }
  npy_reader(const npy_reader&) = delete;
  void raw(void* p, size_t n)
  {
    if (n) CUVS_EXPECTS(fread(p, 1, n, f_) == n, "Invalid or truncated index file %s", name_.c_str());
  }
  void rewind_to(long pos) { fseek(f_, pos, SEEK_SET); }

  npy_header header()
  {
    unsigned char pre[8];
    raw(pre, 8);
    CUVS_EXPECTS(pre[0] == 0x93 && memcmp(pre + 1, "NUMPY", 5) == 0, "%s: bad numpy record magic", name_.c_str());
    CUVS_EXPECTS(pre[6] >= 1 && pre[6] <= 3, "%s: unsupported numpy format version %d", name_.c_str(), (int)pre[6]);
    uint32_t len = 0;
    if (pre[6] == 1) {
      uint16_t l16;
      raw(&l16, 2);
      len = l16;
    } else {
      raw(&len, 4);
    }
    CUVS_EXPECTS(len > 0 && len < (1u << 20), "%s: bad numpy header length", name_.c_str());
    std::string d(len, '\0');
    raw(d.data(), len);
    npy_header h;
    // 'descr'
    std::string descr = value_of(d, "descr");
    CUVS_EXPECTS(descr.size() >= 3, "%s: bad numpy descr", name_.c_str());
    size_t q0 = descr.find_first_of("'\""), q1 = descr.find_last_of("'\"");
    CUVS_EXPECTS(q0 != std::string::npos && q1 > q0 + 2, "%s: bad numpy descr", name_.c_str());
    std::string ds = descr.substr(q0 + 1, q1 - q0 - 1);
    CUVS_EXPECTS(ds[0] != '>', "%s: big-endian payloads are not supported", name_.c_str());
    size_t o = (ds[0] == '<' || ds[0] == '|' || ds[0] == '=') ? 1 : 0;
    h.kind   = ds[o];
    if (h.kind == '?' || h.kind == 'b') h.kind = 'u';
    h.itemsize = ds.size() > o + 1 ? (uint32_t)atoi(ds.c_str() + o + 1) : 1;
    if (h.kind == 'f' && h.itemsize == 2) h.kind = 'e';
    CUVS_EXPECTS(h.itemsize >= 1 && h.itemsize <= 8, "%s: bad numpy itemsize", name_.c_str());
    h.fortran = value_of(d, "fortran_order").find("True") != std::string::npos;
    std::string sh = value_of(d, "shape");
    size_t p0 = sh.find('('), p1 = sh.find(')');
    CUVS_EXPECTS(p0 != std::string::npos && p1 != std::string::npos && p1 > p0, "%s: bad numpy shape", name_.c_str());
    const char* c = sh.c_str() + p0 + 1;
    const char* e = sh.c_str() + p1;
    while (c < e) {
      while (c < e && (*c == ' ' || *c == ',')) c++;
      if (c >= e) break;
      char* end;
      long long v = strtoll(c, &end, 10);
      CUVS_EXPECTS(end != c && v >= 0, "%s: bad numpy shape", name_.c_str());
      h.shape.push_back(v);
      c = end;
      if (c < e && *c == 'L') c++;
    }
    return h;
  }

  template <typename T>
  T scalar()
  {
    npy_header h = header();
    CUVS_EXPECTS(h.count() == 1, "%s: expected a scalar record", name_.c_str());
    unsigned char b[8] = {0};
    raw(b, h.itemsize);
    if (h.kind == 'f') {
      if (h.itemsize == 4) { float v; memcpy(&v, b, 4); return (T)v; }
      double v; memcpy(&v, b, 8); return (T)v;
    }
    if (h.kind == 'e') { __half v; memcpy(&v, b, 2); return (T)__half2float(v); }
    uint64_t u = 0;
    memcpy(&u, b, h.itemsize);
    if (h.kind == 'i' && h.itemsize < 8 && (b[h.itemsize - 1] & 0x80)) u |= ~uint64_t(0) << (8 * h.itemsize);
    return (T)(int64_t)u;
  }

  // read an array record whose element size must be `itemsize`; returns the header (shape)
  npy_header array(uint32_t itemsize, int64_t expect_count, std::vector<char>& out)
  {
    npy_header h = header();
    CUVS_EXPECTS(h.itemsize == itemsize, "%s: array element size %u, expected %u", name_.c_str(), h.itemsize, itemsize);
    CUVS_EXPECTS(!h.fortran || h.shape.size() < 2, "%s: column-major arrays are not supported", name_.c_str());
    CUVS_EXPECTS(expect_count < 0 || h.count() == expect_count, "%s: array has %ld elements, expected %ld",
                 name_.c_str(), (long)h.count(), (long)expect_count);
    out.resize((size_t)h.count() * itemsize);
    raw(out.data(), out.size());
    return h;
  }
  template <typename T>
  std::vector<T> host_array(int64_t expect_count)
  {
    std::vector<char> b;
    array(sizeof(T), expect_count, b);
    std::vector<T> v(b.size() / sizeof(T));
    if (!b.empty()) memcpy(v.data(), b.data(), b.size());
    return v;
  }
  // straight to a persistent device buffer through a bounded bounce buffer
  dev_buf<char> device_bytes(resources& res, uint32_t itemsize, int64_t expect_count, npy_header* out_h = nullptr)
  {
    npy_header h = header();
    CUVS_EXPECTS(h.itemsize == itemsize, "%s: array element size %u, expected %u", name_.c_str(), h.itemsize, itemsize);
    CUVS_EXPECTS(!h.fortran || h.shape.size() < 2, "%s: column-major arrays are not supported", name_.c_str());
    CUVS_EXPECTS(expect_count < 0 || h.count() == expect_count, "%s: array has %ld elements, expected %ld",
                 name_.c_str(), (long)h.count(), (long)expect_count);
    size_t bytes = (size_t)h.count() * itemsize, step = size_t(256) << 20;
    auto buf     = dev_buf<char>::persistent(bytes);
    std::vector<char> b(std::min(bytes, step));
    for (size_t o = 0; o < bytes; o += step) {
      size_t n = std::min(step, bytes - o);
      raw(b.data(), n);
      copy_async(res, buf.data() + o, b.data(), n);
      sync(res);
    }
    if (out_h) *out_h = h;
    return buf;
  }
  template <typename T>
  dev_buf<T> device_array(resources& res, int64_t expect_count, npy_header* out_h = nullptr)
  {
    std::vector<T> h = host_array<T>(expect_count);
    auto buf         = dev_buf<T>::persistent(h.size());
    copy_async(res, buf.data(), h.data(), h.size() * sizeof(T));
    sync(res);
    (void)out_h;
    return buf;
  }

 private:
  // text after "'key':" up to the next top-level ',' (parentheses respected)
  std::string value_of(const std::string& d, const char* key)
  {
    size_t k = d.find(std::string("'") + key + "'");
    if (k == std::string::npos) k = d.find(std::string("\"") + key + "\"");
    CUVS_EXPECTS(k != std::string::npos, "%s: numpy header lacks '%s'", name_.c_str(), key);
    size_t c = d.find(':', k);
    CUVS_EXPECTS(c != std::string::npos, "%s: malformed numpy header", name_.c_str());
    int depth = 0;
    size_t e  = c + 1;
    for (; e < d.size(); e++) {
      if (d[e] == '(') depth++;
      else if (d[e] == ')') depth--;
      else if ((d[e] == ',' || d[e] == '}') && depth <= 0) break;
    }
    return d.substr(c + 1, e - c - 1);
  }
  FILE* f_ = nullptr;
  std::string name_;
};

// true when `name` starts with this library's own container magic (serialize.hpp)
inline bool is_native_container(const char* name)
{
  CUVS_EXPECTS(name != nullptr, "filename is null");
  FILE* f = fopen(name, "rb");
  CUVS_EXPECTS(f != nullptr, "Cannot open file %s", name);
  char m[8] = {0};
  size_t n  = fread(m, 1, 8, f);
  fclose(f);
  return n == 8 && memcmp(m, "CUVSAMD1", 8) == 0;
}
// CUVS_AMD_NATIVE_FORMAT=1 makes *Serialize write this library's own container instead of the reference's
inline bool write_native_container(const resources& res) { return res.tune.native_format; }

}  // namespace cuvs_amd
