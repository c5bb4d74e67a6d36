// VALU issue cost of the instructions the IVF-PQ scan kernels are made of, at the scan kernels' occupancy (one
// 1024-thread workgroup per CU = 4 waves per SIMD): cycles per wave-instruction and SIMD (shader clock, s_memtime).
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_bench.hip -o scripts/bin/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 4096;
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* cycles, uint32_t* sink)
{
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = a0 * 7 + 1, c = 0x3c003c00u;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
  typedef float f2_t __attribute__((ext_vector_type(2)));
  f2_t p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f1, f0}, p3 = {f3, f2}, pb = {1.0001f, 0.9999f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
    if constexpr (MODE == 0) { REP16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if constexpr (MODE == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(pb.x));) }
    if constexpr (MODE == 2) { REP16(asm volatile("v_pk_add_f16 %0, %0, %4\n\tv_pk_add_f16 %1, %1, %4\n\tv_pk_add_f16 %2, %2, %4\n\tv_pk_add_f16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (MODE == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
    if constexpr (MODE == 4) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
    if constexpr (MODE == 5) { REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n\tv_cvt_pkrtz_f16_f32 %1, %5, %4\n\tv_cvt_pkrtz_f16_f32 %2, %4, %4\n\tv_cvt_pkrtz_f16_f32 %3, %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1));) }
    if constexpr (MODE == 6) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n\tv_perm_b32 %1, %1, %4, %5\n\tv_perm_b32 %2, %2, %4, %5\n\tv_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(0x0c0c0400u));) }
    if constexpr (MODE == 7) { REP16(asm volatile("v_fma_mix_f32 %0, %4, 1.0, %0 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %4, 1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %4, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c));) }
    if constexpr (MODE == 8) { REP16(asm volatile("v_mul_u32_u24_sdwa %0, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_mul_u32_u24_sdwa %1, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\tv_mul_u32_u24_sdwa %2, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\tv_mul_u32_u24_sdwa %3, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(520u));) }
    if constexpr (MODE == 9) { REP16(asm volatile("v_sub_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %4\n\tv_sub_f32 %2, %2, %4\n\tv_sub_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(pb.x));) }
    if constexpr (MODE == 10) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %5, %4\n\tv_cvt_pk_f16_f32 %2, %4, %4\n\tv_cvt_pk_f16_f32 %3, %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1));) }
    if constexpr (MODE == 11) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) atomicAdd(&cycles[blockIdx.x], t1 - t0);
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u || f0 + f1 + f2 + f3 + p0.x + p1.x + p2.y + p3.y == 1.2345f) sink[0] = a0;
}
template <int MODE>
void run(const char* name, int n_cus, unsigned long long* d_cycles, uint32_t* d_sink, bool last)
{
  double cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(d_cycles, 0, 8 * n_cus));
    hipLaunchKernelGGL(k<MODE>, dim3(n_cus), dim3(1024), 0, 0, d_cycles, d_sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(n_cus);
    CHECK(hipMemcpy(h.data(), d_cycles, 8 * n_cus, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    cyc = s / n_cus / 16.0;
  }
  const double per_wave = (double)kIters * 64.0;  // wave-instructions per wave; 4 waves per SIMD
  printf("  \"%s\": {\"cycles_per_wave_instr_per_simd\": %.3f}%s\n", name, cyc / (per_wave * 4.0), last ? "" : ",");
}
int main()
{
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int n = prop.multiProcessorCount;
  unsigned long long* dc; uint32_t* ds;
  CHECK(hipMalloc(&dc, 8 * n)); CHECK(hipMalloc(&ds, 64));
  printf("{\n");
  run<0>("v_add_u32", n, dc, ds, false); run<1>("v_fma_f32", n, dc, ds, false); run<9>("v_sub_f32", n, dc, ds, false);
  run<2>("v_pk_add_f16", n, dc, ds, false); run<3>("v_pk_fma_f32", n, dc, ds, false); run<4>("v_pk_add_f32", n, dc, ds, false);
  run<5>("v_cvt_pkrtz_f16_f32", n, dc, ds, false); run<10>("v_cvt_pk_f16_f32", n, dc, ds, false); run<6>("v_perm_b32", n, dc, ds, false);
  run<7>("v_fma_mix_f32", n, dc, ds, false); run<8>("v_mul_u32_u24_sdwa", n, dc, ds, false); run<11>("v_cndmask_b32", n, dc, ds, true);
  printf("}\n");
  return 0;
}
