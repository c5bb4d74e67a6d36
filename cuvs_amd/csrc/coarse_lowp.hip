// Reduced-precision coarse search of IVF-PQ (search_params.coarse_search_dtype = CUDA_R_16F / CUDA_R_8I;
// ivf_pq_search.cuh:171-340): the query x centre products on the matrix cores OF THAT TYPE - v_mfma_f32_32x32x16_f16 /
// v_mfma_i32_32x32x32_i8 - with the reference's output arithmetic fused into the epilogue (half: half(alpha (dot - |c|^2_h
// / 2)); int8: alpha (dot + norm term)), so the nq x n_lists matrix is written once and never re-read before select_k.
// The kernel is bound by that write (10k x 16384: 655 MB): operands come straight from global memory / L2 - a wave keeps
// the A operands of its 32 queries in registers and walks over 32-centre column tiles (all centres of an index are a
// few MB: L2-resident), no LDS staging.
// int8: sum of int8 x int8 products in int32 - exact, any order. fp16: products exact in fp32, fp32 accumulation in the
// matrix core's order (the reference's cuBLAS order is not specified either).
#include "common.hpp"
#include "device_utils.hpp"
#include "ops.hpp"

#include <type_traits>

namespace cuvs_amd {

namespace {

typedef _Float16 lp_f16x8 __attribute__((ext_vector_type(8)));
typedef float lp_f32x16 __attribute__((ext_vector_type(16)));
typedef int lp_i32x4 __attribute__((ext_vector_type(4)));
typedef int lp_i32x16 __attribute__((ext_vector_type(16)));

// rows packed for the kernel: ks K steps of 32 bytes (16 halves / 32 int8), zero-padded past `cols`
template <bool I8>
__global__ void lowp_pack_kernel(const float* __restrict__ in, int64_t n, int64_t ld_in, int64_t cols, int ks,
                                 uint8_t* __restrict__ out)
{
  const int64_t per_row = (int64_t)ks * (I8 ? 32 : 16);
  const int64_t i       = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * per_row) return;
  const int64_t r = i / per_row, c = i % per_row;
  const float v   = c < cols ? in[r * ld_in + c] : 0.f;
  if constexpr (I8) {
    // static_cast<int8_t>(clamp(v * 128, -128, 127)): truncation toward zero (ivf_pq_search.cuh:205-215)
    reinterpret_cast<int8_t*>(out)[i] = (int8_t)truncf(fmaxf(-128.0f, fminf(127.0f, v * 128.0f)));
  } else {
    reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
  }
}

struct lowp_args {
  const uint4* q;  // [nq][ks][2] 16-byte K halves
  const uint4* c;  // [n][ks][2]
  int64_t nq, n;
  const float* term;  // [n] or nullptr
  float alpha;
  float* out;
  int64_t ldo;
  int ks, tiles_per_wg;
};

// KS: K steps held in registers (the packed rows hold exactly KS steps); KS == 0: any number, operands re-read per tile
template <bool I8, int KS>
__global__ __launch_bounds__(256) void coarse_lowp_kernel(const lowp_args a)
{
  constexpr int KR = KS > 0 ? KS : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int64_t q0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
  if (q0 >= a.nq) return;  // wave-uniform
  const int ks = KS > 0 ? KS : a.ks;
  const uint4* qp = a.q + ((size_t)min(q0 + li, a.nq - 1) * ks) * 2 + h;
  uint4 av[KR];
  if constexpr (KS > 0) {
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = qp[s * 2];
  }
  const int64_t n_ct = (a.n + 31) / 32;
  const int64_t ct0 = (int64_t)blockIdx.x * a.tiles_per_wg, ct1 = min(n_ct, ct0 + a.tiles_per_wg);
  auto cptr = [&](const int64_t ct) { return a.c + ((size_t)min(ct * 32 + li, a.n - 1) * ks) * 2 + h; };
  uint4 bv[2][KR];
  if constexpr (KS > 0) {
    if (ct0 < ct1) {
      const uint4* cp = cptr(ct0);
#pragma unroll
      for (int s = 0; s < KS; ++s) bv[0][s] = cp[s * 2];
    }
  }
  auto tile = [&](const int64_t ct, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    using acc_t = std::conditional_t<I8, lp_i32x16, lp_f32x16>;
    acc_t acc = {};
    if constexpr (KS > 0) {
      if (ct + 1 < ct1) {  // the next tile's operands are in flight during this tile's products and stores
        const uint4* cp = cptr(ct + 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) bv[cur ^ 1][s] = cp[s * 2];
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if constexpr (I8)
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(lp_i32x4, av[s]), __builtin_bit_cast(lp_i32x4, bv[cur][s]), acc, 0, 0, 0);
        else
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lp_f16x8, av[s]), __builtin_bit_cast(lp_f16x8, bv[cur][s]), acc, 0, 0, 0);
      }
    } else {
      const uint4* cp = cptr(ct);
#pragma unroll 4
      for (int s = 0; s < ks; ++s) {
        const uint4 x = qp[s * 2], y = cp[s * 2];
        if constexpr (I8)
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(lp_i32x4, x), __builtin_bit_cast(lp_i32x4, y), acc, 0, 0, 0);
        else
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lp_f16x8, x), __builtin_bit_cast(lp_f16x8, y), acc, 0, 0, 0);
      }
    }
    // accumulator register i of lane (li, h): query row (i & 3) + 8 (i >> 2) + 4 h of the block, centre li of the tile -
    // a half-wave writes 128 contiguous bytes of one output row
    const int64_t col = ct * 32 + li;
    if (col >= a.n) return;
    const float t = a.term != nullptr ? a.term[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = q0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= a.nq) continue;
      float v;
      if constexpr (I8) {
        v = (float)acc[i] + t;
        v = a.alpha * v;
      } else {
        v = acc[i];
        if (a.term != nullptr) v = v + -0.5f * t;
        v = (float)(_Float16)(a.alpha * v);
      }
      a.out[row * a.ldo + col] = v;
    }
  };
  for (int64_t ct = ct0; ct < ct1; ct += 2) {
    tile(ct, std::integral_constant<int, 0>{});
    if (ct + 1 < ct1) tile(ct + 1, std::integral_constant<int, 1>{});
  }
}

template <bool I8>
void launch_lowp(resources& res, const lowp_args& a)
{
  const int64_t n_ct = (a.n + 31) / 32;
  dim3 grid((unsigned)((n_ct + a.tiles_per_wg - 1) / a.tiles_per_wg), (unsigned)((a.nq + 127) / 128));
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), 0, res.stream, a); };
  switch (a.ks) {
    case 1:  go(coarse_lowp_kernel<I8, 1>); break;
    case 2:  go(coarse_lowp_kernel<I8, 2>); break;
    case 3:  go(coarse_lowp_kernel<I8, 3>); break;
    case 4:  go(coarse_lowp_kernel<I8, 4>); break;
    case 6:  go(coarse_lowp_kernel<I8, 6>); break;
    case 8:  go(coarse_lowp_kernel<I8, 8>); break;
    case 12: go(coarse_lowp_kernel<I8, 12>); break;
    case 16: go(coarse_lowp_kernel<I8, 16>); break;
    default: go(coarse_lowp_kernel<I8, 0>); break;
  }
}

}  // namespace

int coarse_lowp_ksteps(bool i8, int64_t cols) { return (int)((cols + (i8 ? 31 : 15)) / (i8 ? 32 : 16)); }

void coarse_lowp_pack(resources& res, bool i8, const float* in, int64_t n, int64_t ld_in, int64_t cols, void* out)
{
  const int ks        = coarse_lowp_ksteps(i8, cols);
  const int64_t total = n * (int64_t)ks * (i8 ? 32 : 16);
  if (total == 0) return;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (i8) hipLaunchKernelGGL(lowp_pack_kernel<true>, grid, dim3(256), 0, res.stream, in, n, ld_in, cols, ks, (uint8_t*)out);
  else    hipLaunchKernelGGL(lowp_pack_kernel<false>, grid, dim3(256), 0, res.stream, in, n, ld_in, cols, ks, (uint8_t*)out);
}

void coarse_lowp_distances(resources& res, bool i8, const void* q_pack, int64_t nq, const void* c_pack, int64_t n, int64_t cols,
                           const float* term, float alpha, float* out, int64_t ldo)
{
  if (nq == 0 || n == 0) return;
  lowp_args a{};
  a.q = (const uint4*)q_pack; a.c = (const uint4*)c_pack; a.nq = nq; a.n = n; a.term = term; a.alpha = alpha; a.out = out; a.ldo = ldo;
  a.ks = coarse_lowp_ksteps(i8, cols);
  // enough workgroups to fill the chip several times over, yet a few column tiles per wave to amortise its A operands
  const int64_t n_ct = (n + 31) / 32, row_blocks = (nq + 127) / 128;
  int64_t t = 16;
  while (t > 1 && ((n_ct + t - 1) / t) * row_blocks < 2048) t >>= 1;
  a.tiles_per_wg = (int)t;
  profile_begin(res, "coarse_lowp_kernel");
  if (i8) launch_lowp<true>(res, a); else launch_lowp<false>(res, a);
  profile_end(res, "coarse_lowp_kernel");
}

}  // namespace cuvs_amd
