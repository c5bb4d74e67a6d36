// Microbenchmark of the two ceilings of the IVF-PQ scan design on MI355X ("next round" item 2a):
//   * the LDS gather rate at the scan kernels' occupancy (one 1024-thread workgroup per CU, ~136 KiB of LDS) for the
//     access patterns the kernels use - random code rows with a common subspace (ds_read_b64 from the padded code-major
//     exact LUT, ds_read_b128 from the padded round-2 filter LUT) and the round-3 filter pattern (unpadded rows, lane l
//     looks up subspace (l + t) mod 16: bank quad == subspace, conflict-free by construction);
//   * the VALU issue rate of the accumulate instruction (v_pk_add_f16).
// Prints one JSON object: cycles per wave-instruction and per CU for every pattern (shader clock, s_memtime), the
// wall-clock time, and a self-check of the v_perm_b32 / v_alignbyte_b32 semantics the filter's addressing relies on.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_gather_bench.hip -o scripts/bin/lds_gather_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); }       \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 1024;
constexpr int kIters   = 2048;  // groups of 8 gathers per wave

__device__ inline uint32_t xorshift(uint32_t& s)
{
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return s;
}

// MODE 0: ds_read_b64  random code rows, padded (row 520 B), common subspace          (exact pass)
// MODE 1: ds_read_b128 random code rows, padded (row 272 B), common subspace          (round-2 filter)
// MODE 2: ds_read_b128 random code rows, unpadded (row 256 B), subspace (lane + t)%16 (round-3 filter)
// MODE 3: ds_read_b64  random code rows, unpadded (row 512 B), subspace (lane + t)%32 (rotated b64)
// MODE 4: ds_read_b32  random code rows (row 260 B), common subspace                  (one query per gather)
// MODE 5: no LDS: 16 v_pk_add_f16 per group (VALU issue)
template <int MODE>
__global__ __launch_bounds__(kThreads) void gather_kernel(unsigned long long* cycles, uint32_t* sink)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lane = threadIdx.x & 63u;
  // fill the LDS with something (the values do not matter)
  for (uint32_t i = threadIdx.x; i < 34816u; i += kThreads) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t st  = (blockIdx.x * kThreads + threadIdx.x) * 747796405u + 2891336453u;
  uint32_t acc0 = 0u, acc1 = 0u, acc2 = 0u, acc3 = 0u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
    const uint32_t w = xorshift(st);  // four random code bytes
    if constexpr (MODE == 5) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        asm volatile("v_pk_add_f16 %0, %0, %4\n\tv_pk_add_f16 %1, %1, %4\n\tv_pk_add_f16 %2, %2, %4\n\tv_pk_add_f16 %3, %3, %4"
                     : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(w));
      continue;
    }
    const uint32_t w2 = xorshift(st);
    const uint32_t s  = (uint32_t)(it & 15);  // common subspace of this step
    // eight gathers in flight per wave, consumed together (no VALU work beyond the address arithmetic)
    using vec_t = typename std::conditional<MODE == 1 || MODE == 2, u32x4, typename std::conditional<MODE == 4, uint32_t, u32x2>::type>::type;
    vec_t e[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint32_t code = ((b < 4 ? w : w2) >> (8 * (b & 3))) & 0xffu;
      uint32_t ad;
      if constexpr (MODE == 0)      ad = code * 520u + (s * 4 + b) * 8u;
      else if constexpr (MODE == 1) ad = code * 272u + ((s + b) & 15u) * 16u;
      else if constexpr (MODE == 2) ad = code * 256u + ((lane + s + b) & 15u) * 16u;
      else if constexpr (MODE == 3) ad = code * 512u + ((lane + s * 4 + b) & 31u) * 8u;
      else                          ad = code * 260u + (s * 4 + b) * 4u;
      e[b] = *(__attribute__((address_space(3))) const vec_t*)(uintptr_t)ad;
    }
    asm volatile("; consume %0 %1 %2 %3 %4 %5 %6 %7" ::"v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) atomicAdd(&cycles[blockIdx.x], t1 - t0);  // summed over the 16 waves of the workgroup
  if ((acc0 ^ acc1 ^ acc2 ^ acc3) == 0x12345678u) sink[0] = acc0;
}

// self-check of the address arithmetic of the rotated filter (v_alignbyte_b32, v_perm_b32)
__global__ void semantics_kernel(uint32_t* bad)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rr = lane & 15u, rb = rr & 3u;
  uint32_t st = lane * 2654435761u + 12345u;
  uint32_t w[4];
  for (int i = 0; i < 4; ++i) w[i] = xorshift(st);
  uint8_t bytes[16];
  for (int i = 0; i < 16; ++i) bytes[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
  uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  if (rr & 4u) { const uint32_t x = w0; w0 = w1; w1 = w2; w2 = w3; w3 = x; }
  if (rr & 8u) { uint32_t x = w0; w0 = w2; w2 = x; x = w1; w1 = w3; w3 = x; }
  uint32_t R[4];
  R[0] = __builtin_amdgcn_alignbyte(w1, w0, rb);
  R[1] = __builtin_amdgcn_alignbyte(w2, w1, rb);
  R[2] = __builtin_amdgcn_alignbyte(w3, w2, rb);
  R[3] = __builtin_amdgcn_alignbyte(w0, w3, rb);
  for (int t = 0; t < 16; ++t) {
    uint32_t xo = 0u;
    for (int b = 0; b < 4; ++b) xo |= (((lane + 4u * (t >> 2) + b) & 15u) << 4) << (8 * b);
    const uint32_t ad   = __builtin_amdgcn_perm(R[t >> 2], xo, 0x0c0c0400u + 0x0101u * (uint32_t)(t & 3));
    const uint32_t sub  = (rr + t) & 15u;
    const uint32_t want = ((uint32_t)bytes[sub] << 8) | (sub << 4);
    if (ad != want) atomicAdd(bad, 1u);
  }
}

template <int MODE>
void run(const char* name, int n_cus, size_t smem, unsigned long long* d_cycles, uint32_t* d_sink, bool last)
{
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best_ms = 1e30f;
  double cyc    = 0.0;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipMemset(d_cycles, 0, sizeof(unsigned long long) * n_cus));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather_kernel<MODE>, dim3(n_cus), dim3(kThreads), smem, 0, d_cycles, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best_ms) {
      best_ms = ms;
      std::vector<unsigned long long> h(n_cus);
      CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * n_cus, hipMemcpyDeviceToHost));
      double s = 0;
      for (auto v : h) s += (double)v;
      cyc = s / n_cus / 16.0;  // average cycles of a wave
    }
  }
  // wave-instructions per CU: 16 waves x kIters x 8 (gathers) [MODE 5: x 16 packed adds]
  const double per_wave = (double)kIters * (MODE == 5 ? 16.0 : 8.0);
  const double per_cu   = per_wave * 16.0;
  printf("  \"%s\": {\"cycles_per_wave_instr_per_cu\": %.3f, \"lane_ops_per_clk_per_cu\": %.2f, \"kernel_ms\": %.4f, \"ns_per_wave_instr_per_cu\": %.4f}%s\n",
         name, cyc / per_cu, 64.0 * per_cu / cyc, best_ms, best_ms * 1e6 / per_cu, last ? "" : ",");
}

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  unsigned long long* d_cycles;
  uint32_t *d_sink, *d_bad;
  CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * n_cus));
  CHECK(hipMalloc(&d_sink, 64));
  CHECK(hipMalloc(&d_bad, 4));
  CHECK(hipMemset(d_bad, 0, 4));
  hipLaunchKernelGGL(semantics_kernel, dim3(1), dim3(64), 0, 0, d_bad);
  uint32_t bad = 0;
  CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  const size_t smem = 136 * 1024 + 3 * 1024;  // the scan kernels' footprint: one workgroup per CU
  printf("{\n  \"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"perm_alignbyte_mismatches\": %u,\n", prop.gcnArchName, n_cus,
         prop.clockRate / 1000, bad);
  run<0>("b64_random_padded_common_subspace", n_cus, smem, d_cycles, d_sink, false);
  run<1>("b128_random_padded_common_subspace", n_cus, smem, d_cycles, d_sink, false);
  run<2>("b128_rotated_subspace_unpadded", n_cus, smem, d_cycles, d_sink, false);
  run<3>("b64_rotated_subspace_unpadded", n_cus, smem, d_cycles, d_sink, false);
  run<4>("b32_random_padded_common_subspace", n_cus, smem, d_cycles, d_sink, false);
  run<5>("valu_pk_add_f16", n_cus, smem, d_cycles, d_sink, true);
  printf("}\n");
  return bad == 0 ? 0 : 2;
}
