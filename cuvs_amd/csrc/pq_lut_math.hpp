// LUT-entry arithmetic of the IVF-PQ scan shared by the scan kernels (ivf_pq_search.hip) and the survivor re-scoring
// (ivf_pq_scan3.hip): every path that produces a score must round exactly like these.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cuvs_amd {
namespace {

// fp32 score -> fp16 LUT entry, as the reference stores it (LutT(score): the fp32-rounded score, then round to nearest
// even). The barrier keeps hipcc from folding fma + convert into v_fma_mixlo_f16, which rounds the exact product-sum
// ONCE and differs from the reference in the last half bit of some entries.
__device__ inline _Float16 to_lut_half(float v)
{
  asm volatile("" : "+v"(v));
  return (_Float16)v;
}

// lut_dtype = CUDA_R_8U / CUDA_R_8I: the reference stores LUT entries in its own 8-bit float fp_8bit<5, Signed>
// (ivf_pq_fp_8bit.cuh:32-100; unsigned for L2, sign in bit 0 for inner product - ivf_pq_search.cuh:711-728): 5 exponent
// bits (bias 15), 3 value bits, truncation on encode, half an ulp added back on decode. Here the entry is rounded
// through that type when the LUT is built and stored in the score type (the value every later add sees is the
// reference's): its float decoder (:75-88) for fp32 scores, its half decoder (:90-102, no implicit one at the
// smallest exponent, NaN/inf patterns at the largest) for fp16 scores.
template <typename AccT>
__device__ inline float fp8_round_trip(float v, bool is_signed)
{
  const float av = is_signed ? fabsf(v) : v;
  uint32_t u;
  if (av < 1.0f / 32768.0f) u = 0u;
  else if (av >= 65536.0f * 1.875f) u = 0xffu;
  else u = ((__float_as_uint(av) + (15u << 23) - 0x3f800000u) >> 20) & 0xffu;
  const bool neg = is_signed && v < 0.f;
  if (is_signed) u &= 0xfeu;
  float r;
  if constexpr (sizeof(AccT) == 2) {
    const uint16_t hb = (uint16_t)(((0x3c00u | (0x0200u >> 3)) - (15u << 10)) + (u << 7));
    r = (float)__builtin_bit_cast(_Float16, hb);
  } else {
    r = __uint_as_float(((0x3f800000u | (0x00400000u >> 3)) - (15u << 23)) + (u << 20));
  }
  return neg ? -r : r;
}

}  // namespace
}  // namespace cuvs_amd
