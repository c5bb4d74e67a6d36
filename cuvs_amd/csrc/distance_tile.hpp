// Tile helpers shared by the fp32-MFMA distance kernels (distance.hip, fused_knn.hip): operand staging with the
// XOR-swizzled k-major LDS layout and the distance epilogue.
#pragma once
#include "ops.hpp"
#include "device_utils.hpp"

namespace cuvs_amd {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM  = 128;
constexpr int BN  = 128;
constexpr int BK  = 16;
constexpr int LDT = 144;  // k-major row pitch: 16*k lands lanes 16-31 on banks 16-31

// ------------------------------------------------------------------ operand staging
// 4 consecutive elements of a row starting at column k0 (zero beyond dim / beyond the matrix)
template <typename T, bool VEC>
__device__ inline void load4(const T* __restrict__ base, int64_t row, int64_t nrows, int64_t ld, int64_t k0,
                             int64_t dim, float (&v)[4])
{
  v[0] = v[1] = v[2] = v[3] = 0.f;
  if (row >= nrows) return;
  const T* p = base + row * ld + k0;
  if constexpr (VEC) {
    if (k0 + 3 < dim) {
      if constexpr (sizeof(T) == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else if constexpr (sizeof(T) == 2) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        const __half* h = reinterpret_cast<const __half*>(&t);
        v[0] = __half2float(h[0]); v[1] = __half2float(h[1]);
        v[2] = __half2float(h[2]); v[3] = __half2float(h[3]);
      } else {
        uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        const T* b = reinterpret_cast<const T*>(&t);
        v[0] = (float)b[0]; v[1] = (float)b[1]; v[2] = (float)b[2]; v[3] = (float)b[3];
      }
      return;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (k0 + e < dim) v[e] = to_float(p[e]);
}

__device__ inline void stage_store(float* __restrict__ S, int r, int c, const float (&v)[4])
{
  int col = r ^ (c << 3);
#pragma unroll
  for (int e = 0; e < 4; ++e) S[(4 * c + e) * LDT + col] = v[e];
}

struct epilogue_args {
  const float* qn;
  const float* xn;
  int metric;
  float clamp_eps;
  const uint32_t* run_if = nullptr;  // device word: the launch is a no-op while it is zero (nullptr: always runs)
};

__device__ inline float finish_distance(float dot, float qn, float xn, int metric, float clamp_eps)
{
  if (metric == M_InnerProduct) return dot;
  if (metric == M_CosineExpanded) return 1.0f - dot / (qn * xn);
  float val = __fmaf_rn(-2.0f, dot, qn + xn);
  // reference self-neighbour clamp (l2_exp.cuh:36-50,113-125) + non-negativity of the fused path
  if (val * val < clamp_eps && qn == xn) val = 0.f;
  val = val > 0.f ? val : 0.f;
  if (metric == M_L2SqrtExpanded || metric == M_L2SqrtUnexpanded) val = sqrtf(val);
  return val;
}

template <typename T>
bool vec_ok(const T* p, int64_t ld, int64_t dim)
{
  size_t bytes = sizeof(T) * 4;
  return (dim % 4 == 0) && (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(p) % bytes == 0);
}

}  // namespace
}  // namespace cuvs_amd
