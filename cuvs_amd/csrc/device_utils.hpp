// Device-side helpers written for wave64 (gfx950): order-preserving float keys, wave/block
// reductions and scans, a (key,index) bitonic sort over LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace cuvs_amd {

constexpr int kWave = 64;

// Order-preserving map float -> uint32 (ascending). NaNs sort above +inf.
__host__ __device__ inline uint32_t float_to_key(float f)
{
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(f);
#else
  memcpy(&u, &f, 4);
#endif
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float key_to_float(uint32_t k)
{
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

__device__ inline int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ inline float wave_sum(float v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ inline int wave_sum_i(int v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Inclusive scan across one wave.
__device__ inline int wave_inclusive_scan(int v)
{
  int lane = lane_id();
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    int t = __shfl_up(v, off, kWave);
    if (lane >= off) v += t;
  }
  return v;
}

// Block-wide exclusive scan of one int per thread; blockDim.x multiple of 64, <= 1024.
// `smem` needs 17 ints. Returns the exclusive prefix; *total receives the block sum.
__device__ inline int block_exclusive_scan(int v, int* smem, int* total)
{
  int lane = lane_id();
  int wid  = threadIdx.x >> 6;
  int nw   = blockDim.x >> 6;
  int inc  = wave_inclusive_scan(v);
  if (lane == kWave - 1) smem[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int w = (lane < nw) ? smem[lane] : 0;
    int s = wave_inclusive_scan(w);
    if (lane < nw) smem[lane] = s - w;  // exclusive per-wave offsets
    if (lane == nw - 1) smem[16] = s;
  }
  __syncthreads();
  int res = inc - v + smem[wid];
  *total  = smem[16];
  __syncthreads();
  return res;
}

// Bitonic sort of n (power of two) (key,idx) pairs held in LDS, ascending by (key, idx).
// All threads of the block participate; caller syncs before and gets a synced result.
template <typename IdxT>
__device__ inline void block_bitonic_sort(uint32_t* keys, IdxT* idx, int n)
{
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool up = ((lo & size) == 0);
        uint32_t ka = keys[lo], kb = keys[hi];
        IdxT ia = idx[lo], ib = idx[hi];
        bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
        if (a_gt_b == up) {
          keys[lo] = kb; keys[hi] = ka;
          idx[lo] = ib; idx[hi] = ia;
        }
      }
    }
  }
  __syncthreads();
}

// Same, but executed by ONE wave (no block barriers; LDS ops of a single wave are ordered).
template <typename IdxT>
__device__ inline void wave_bitonic_sort(uint32_t* keys, IdxT* idx, int n)
{
  int lane = lane_id();
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int t = lane; t < (n >> 1); t += kWave) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool up = ((lo & size) == 0);
        uint32_t ka = keys[lo], kb = keys[hi];
        IdxT ia = idx[lo], ib = idx[hi];
        bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
        if (a_gt_b == up) {
          keys[lo] = kb; keys[hi] = ka;
          idx[lo] = ib; idx[hi] = ia;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__host__ __device__ inline int next_pow2(int v)
{
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// element -> float mapping for the supported dataset dtypes
// (reference: cpp/src/neighbors/detail/ann_utils.cuh:134-196 `mapping<float>`)
__device__ inline float to_float(float v) { return v; }
__device__ inline float to_float(__half v) { return __half2float(v); }
__device__ inline float to_float(int8_t v) { return (float)v; }
__device__ inline float to_float(uint8_t v) { return (float)v; }

}  // namespace cuvs_amd
