// Can ONE wave per SIMD overlap its own MFMAs with its own VALU / LDS instructions? (the question behind the 0.30 of
// pq_filter4_kernel: 1245 cycles per subtile = ~740 of MFMA + ~520 of other issue slots, as if nothing overlapped)
// A loop body of 4 independent v_mfma_f32_32x32x16_f16 (32 cycles of matrix pipe each at full rate) with N independent VALU
// instructions and M independent ds_read_b32 between them; 256 threads per workgroup (one wave per SIMD), 1 workgroup per CU;
// cycles per loop iteration from s_memtime. Serial issue would give 128 + 4 N (+ LDS); full overlap max(128, 4 N, ...).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_overlap_bench.hip -o scripts/bin/mfma_overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kIters = 16384;

template <int NV, int NL, int WAVES>  // VALU ops / LDS gathers per MFMA; waves per SIMD
__global__ __launch_bounds__(256 * WAVES) void k(unsigned long long* cycles, float* sink)
{
  __shared__ uint32_t tab[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = i * 2654435761u;
  __syncthreads();
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  uint32_t v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 17 + j;
  uint32_t addr = (threadIdx.x * 2654435761u) & 0xfffcu, g[4] = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#define STEP(C)                                                                                         \
    C = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, C, 0, 0, 0);                                       \
    _Pragma("unroll") for (int j = 0; j < NV; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(addr)); \
    _Pragma("unroll") for (int j = 0; j < NL; ++j) { uint32_t t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((addr + (uint32_t)(j * 1024 + it * 4)) & 0xfffcu)); g[j & 3] ^= t; }
    STEP(c0) STEP(c1) STEP(c2) STEP(c3)
    // (gathers stay in flight across iterations, as in a software-pipelined kernel: the counter holds 15 - wait for the oldest
    // ones only, never for all)
    if (NL > 0) { if (NL * 4 >= 8) asm volatile("s_waitcnt lgkmcnt(8)"); else asm volatile("s_waitcnt lgkmcnt(4)"); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) atomicAdd(&cycles[blockIdx.x], t1 - t0);
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  uint32_t x = 0;
  for (int j = 0; j < 8; ++j) x ^= v[j];
  if (s == 1.2345f || (x ^ g[0] ^ g[1] ^ g[2] ^ g[3]) == 0x12345678u) sink[0] = s;
}

template <int NV, int NL, int WAVES>
void run(int n_cus, unsigned long long* dc, float* ds)
{
  double cyc = 0;
  float ms = 0.f;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(dc, 0, 8 * n_cus));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<NV, NL, WAVES>), dim3(n_cus), dim3(256 * WAVES), 0, 0, dc, ds);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(n_cus);
    CHECK(hipMemcpy(h.data(), dc, 8 * n_cus, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    cyc = s / n_cus / (4.0 * WAVES) / kIters;  // s_memtime ticks per loop iteration and wave (100 MHz constant clock? see below)
  }
  // wall-clock view: MFMA rate of the whole chip and the counter's tick rate
  const double flop = (double)n_cus * 4.0 * WAVES * kIters * 4.0 * 32768.0;
  printf("  {\"valu_per_mfma\": %d, \"lds_per_mfma\": %d, \"waves_per_simd\": %d, \"ticks_per_iteration\": %.2f, \"kernel_ms\": %.4f, "
         "\"mfma_tflops\": %.1f, \"ticks_per_us\": %.0f},\n", NV, NL, WAVES, cyc, ms, flop / (ms * 1e-3) / 1e12, cyc * kIters / (ms * 1e3));
}
int main()
{
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int n = prop.multiProcessorCount;
  unsigned long long* dc; float* ds;
  CHECK(hipMalloc(&dc, 8 * n)); CHECK(hipMalloc(&ds, 64));
  printf("{\"note\": \"4 independent v_mfma_f32_32x32x16_f16 per iteration + N v_add_u32 + M ds_read_b32 (random banks) after each; ticks of __builtin_readcyclecounter per iteration\", \"rows\": [\n");
  run<0, 0, 1>(n, dc, ds); run<4, 0, 1>(n, dc, ds); run<8, 0, 1>(n, dc, ds); run<16, 0, 1>(n, dc, ds); run<32, 0, 1>(n, dc, ds);
  run<0, 1, 1>(n, dc, ds); run<0, 2, 1>(n, dc, ds); run<0, 4, 1>(n, dc, ds); run<8, 1, 1>(n, dc, ds); run<8, 2, 1>(n, dc, ds);
  run<0, 0, 4>(n, dc, ds); run<4, 1, 4>(n, dc, ds);
  run<0, 0, 2>(n, dc, ds); run<8, 0, 2>(n, dc, ds); run<16, 0, 2>(n, dc, ds); run<8, 1, 2>(n, dc, ds); run<8, 2, 2>(n, dc, ds);
  printf("  {}\n]}\n");
  return 0;
}
