// Loop-shape microbenchmark for a conflict-free decode of PQ rows into MFMA A operands (the question behind round 6's filter):
// TWO waves per SIMD (512 threads, 256 registers each) share every decoded 32-row tile through an LDS ring:
//   * gather: lane (row r, K half h) reads, in instruction i, the table entry of subspace (h, r ^ i) of its row - the decode table
//     is code-major, [256 codes][64 subspaces] dwords, so the 32 lanes of a half-wave hit 32 different banks whatever the codes are
//     (a gather BY CODE into a subspace-major table - pq_filter4_kernel - lands 32 random codes in 32 banks: 2.6 x the cycles);
//     the table address is ONE v_perm_b32 (byte 1 = the code, byte 0 = a per-lane constant);
//   * the entry lands in the wrong register for an MFMA operand (slot i holds a different subspace in every row), so it goes
//     through LDS: ds_write_b32 into a [K step][lane slot][16 B] tile (address = lane constant ^ instruction constant, the slot
//     permutation makes the stores and the later ds_read_b128 conflict-free), both waves of the SIMD read the tile and multiply it
//     with THEIR half of the unit's queries (B operands: 2 groups of 32 queries = 64 registers per wave);
//   * the waves of a pair alternate as producers (even / odd tiles, ring of 2 slots), flags in LDS, no workgroup barrier.
// Reports cycles per tile and SIMD against pq_filter4_kernel's 1245 (4 query groups) at the same MFMA count per tile.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/filter5_shape_bench.hip -o scripts/bin/filter5_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) u32x4 lds_u128;

constexpr uint32_t kRing = 65536u, kFlags = 131072u, kTile = 8192u;
// lane slot of row r inside a K step's 1 KiB block: stores of instruction i (dword (r ^ i) & 3 of slot p(r), K step (r ^ i) >> 2)
// fall into 32 different banks for every i, and the 16 lanes of every ds_read_b128 group read 16 slots that differ mod 16
__constant__ uint8_t kSlotPerm[32] = {0, 2, 4, 6, 17, 19, 21, 23, 18, 20, 22, 16, 3, 5, 7, 1, 28, 30, 24, 26, 13, 15, 9, 11, 14, 8, 10, 12, 31, 25, 27, 29};

__device__ inline uint32_t lds_poll(const uint32_t addr)
{
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ inline void lds_flag(const uint32_t addr, const uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// NGW: query groups per wave (1 or 2). MODE bit 0: no decode (MFMAs + tile reads only), bit 1: no MFMAs, bit 2: conflicting gathers
// (subspace-major addressing: what a gather by code costs), bit 3: no flags (timing of the protocol)
template <int NGW, int MODE>
__global__ __launch_bounds__(512) void k5(const uint4* __restrict__ codes, const uint32_t n_tiles, unsigned long long* cycles, float* sink)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (uint32_t i = threadIdx.x; i < 16384u; i += 512u) {
    const uint32_t x = i * 2654435761u;
    reinterpret_cast<uint32_t*>(smem)[i] = 0x38003800u | (x & 0x03ff03ffu);
  }
  if (threadIdx.x < 64u) reinterpret_cast<uint32_t*>(smem + kFlags)[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, r = lane & 31u, h = lane >> 5, wave = threadIdx.x >> 6, pair = wave & 3u, role = wave >> 2;
  uint32_t cst[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    uint32_t v = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) v |= (((h * 32u + (r ^ (uint32_t)(4 * g + j))) * 4u) & 255u) << (8 * j);
    cst[g] = v;
  }
  const uint32_t slot  = h * 32u + kSlotPerm[r];
  const uint32_t ring  = kRing + pair * 2u * kTile;
  const uint32_t wbase = ring + role * kTile + (((r >> 2) << 10) | (slot << 4) | ((r & 3u) << 2));  // my tiles live in slot `role`
  const uint32_t rbase = ring + slot * 16u;
  const uint32_t f_ready = kFlags + pair * 16u, f_done = f_ready + 8u;  // [2] each

  f16x8 bop[NGW][8];
#pragma unroll
  for (int g = 0; g < NGW; ++g)
#pragma unroll
    for (int st = 0; st < 8; ++st) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bop[g][st][e] = (_Float16)(0.001f * (float)((lane * 7 + g * 13 + st * 3 + e) & 63));
      asm volatile("" : "+a"(bop[g][st]));
    }
  f32x16 acc[NGW], term = {};
  for (int g = 0; g < NGW; ++g) acc[g] = f32x16{};
  const float thr = 1e30f;
  unsigned long long hit = 0ull;
  uint32_t gat[32];
  for (int i = 0; i < 32; ++i) gat[i] = 0u;
  uint32_t nspin = 0u;
  const uint4* cp = codes + ((size_t)(blockIdx.x * 4u + pair) * 64u) * 128u + lane * 2u;  // 64 tiles per pair, wrapped
  auto load_codes = [&](const uint32_t t, uint4 (&c)[2]) {
    const uint4* p = cp + (size_t)(t & 63u) * 128u;
    c[0] = p[0]; c[1] = p[1];
  };
  auto code_word = [&](const uint4 (&c)[2], const int w) -> uint32_t {
    const uint4& q = c[w >> 2];
    return (w & 3) == 0 ? q.x : (w & 3) == 1 ? q.y : (w & 3) == 2 ? q.z : q.w;
  };
  // gather i of a tile: entry of subspace (h, r ^ i); bit 2 of MODE: subspace-major addressing with the same instruction count
  auto gather = [&](const uint4 (&c)[2], const int i) {
    uint32_t addr;
    if constexpr ((MODE & 4) != 0) addr = (__builtin_amdgcn_perm(code_word(c, i >> 2), 0u, 0x0c0c0c04u + (i & 3)) << 2) + (uint32_t)i * 1024u + h * 32768u;
    else addr = __builtin_amdgcn_perm(code_word(c, i >> 2), cst[i >> 2], 0x0c0c0400u + 0x100u * (i & 3) + (i & 3));
    gat[i] = *reinterpret_cast<lds_u32*>(addr);
  };
  auto scatter = [&](const int i) {
    const uint32_t addr = wbase ^ ((((uint32_t)i >> 2) << 10) | (((uint32_t)i & 3u) << 2));
    *reinterpret_cast<lds_u32*>(addr) = gat[i];
  };
  // prologue: role 0 produces tile 0, role 1 tile 1
  uint4 cwA[2], cwB[2];
  load_codes(role, cwA);
  if constexpr ((MODE & 1) == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) gather(cwA, i);
#pragma unroll
    for (int i = 0; i < 32; ++i) scatter(i);
  }
  lds_flag(f_ready + role * 4u, role + 1u);
  load_codes(role + 2u, cwA);
  load_codes(role + 4u, cwB);

  u32x4 av[3];
  const unsigned long long t0 = __builtin_readcyclecounter();
  // The loop is written once per role (compile time): a wave's instruction stream is straight-line code, so the compiler's waits
  // on the LDS counter are COUNTED (a branch on the role - even a uniform one - makes every wait a wait for all but one).
  auto pair_loop = [&](auto role_tag) {
    constexpr uint32_t ROLE = decltype(role_tag)::value;
    uint32_t early = 0u;  // flag value read ahead of its use
    auto flag_read = [&](const uint32_t addr) { early = *reinterpret_cast<volatile lds_u32*>(addr); };
    auto flag_wait = [&](const uint32_t addr, const uint32_t need) {
      if constexpr ((MODE & 8) != 0) return;
      if (__builtin_amdgcn_readfirstlane(early) >= need) return;
      uint32_t spins = 0u;  // (bounded: a protocol error must not hang the box)
      while (lds_poll(addr) < need && ++spins < 100000u) __builtin_amdgcn_s_sleep(1);
      if (spins >= 100000u) hit |= 1ull << 63;
      nspin += 1u;
    };
    // iteration of tile t; P = t & 1: ring slot P, produced by role P
    auto iteration = [&](auto p_tag, const uint32_t t, uint4 (&cw)[2]) {
      constexpr uint32_t P = decltype(p_tag)::value;
      constexpr bool MINE = ROLE == P;
      // screen of the previous tile
#pragma unroll
      for (int g = 0; g < NGW; ++g) {
        float m = fmaxf(fmaxf(acc[g][0], acc[g][1]), acc[g][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, acc[g][i]), acc[g][i + 1]);
        m = fmaxf(m, acc[g][15]);
        hit |= __ballot(m >= thr);
      }
      if constexpr (!MINE) flag_wait(f_ready + P * 4u, t + 1u);  // the partner's tile (flag read at K step 6 of the last iteration)
      const uint32_t rb = rbase + P * kTile;
      av[0] = *reinterpret_cast<lds_u128*>(rb);
      av[1] = *reinterpret_cast<lds_u128*>(rb + 1024u);
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        if (st + 2 < 8) av[(st + 2) % 3] = *reinterpret_cast<lds_u128*>(rb + (st + 2) * 1024u);
        const f16x8 aop = __builtin_bit_cast(f16x8, av[st % 3]);
        auto mfma = [&](const int g) {
          if constexpr ((MODE & 2) != 0) { acc[g][st] += (float)aop[0] * (float)bop[g][st][0]; }
          else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, bop[g][st], st == 0 ? term : acc[g], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        mfma(0);
        if constexpr (MINE) {  // gathers of my next tile (t + 2): four per K step
          if constexpr ((MODE & 1) == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) gather(cw, st * 4 + e);
          }
          if (st == 6) flag_read(f_ready + (P ^ 1u) * 4u);  // is the partner's tile t + 1 there?
        } else {  // stores of my tile t + 1 into the slot of my tile t - 1: K steps 2 .. 5, once the partner has read that one
          if (st == 1) flag_read(f_done + (P ^ 1u) * 4u);
          if (st == 2 && t >= 1u) flag_wait(f_done + (P ^ 1u) * 4u, t);
          if constexpr ((MODE & 1) == 0) {
            if (st >= 2 && st < 6) {
#pragma unroll
              for (int e = 0; e < 4; ++e) scatter((st - 2) * 8 + e);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NGW >= 2) mfma(1);
        if constexpr (MINE) {
          if constexpr ((MODE & 1) == 0) {
#pragma unroll
            for (int e = 2; e < 4; ++e) gather(cw, st * 4 + e);
          }
        } else {
          if constexpr ((MODE & 1) == 0) {
            if (st >= 2 && st < 6) {
#pragma unroll
              for (int e = 4; e < 8; ++e) scatter((st - 2) * 8 + e);
            }
          }
          if (st == 5) lds_flag(f_ready + (P ^ 1u) * 4u, t + 2u);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!MINE) lds_flag(f_done + P * 4u, t + 1u);  // (every read of the tile has returned: the last MFMA waited for it)
      if constexpr (MINE) load_codes(t + 6u, cw);               // the code words just decoded make room for those of tile t + 6
    };
    using P0 = std::integral_constant<uint32_t, 0>; using P1 = std::integral_constant<uint32_t, 1>;
    for (uint32_t t = 0; t < n_tiles; t += 4u) {
      iteration(P0{}, t, cwA);       // (role 0: decodes tile t + 2 from cwA; role 1: stores its tile t + 1)
      iteration(P1{}, t + 1u, cwA);  // (role 1: decodes tile t + 3 from cwA)
      iteration(P0{}, t + 2u, cwB);
      iteration(P1{}, t + 3u, cwB);
    }
  };
  if (__builtin_amdgcn_readfirstlane(role) == 0u) pair_loop(std::integral_constant<uint32_t, 0>{});
  else pair_loop(std::integral_constant<uint32_t, 1>{});
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { atomicAdd(&cycles[blockIdx.x], t1 - t0); atomicAdd(&cycles[gridDim.x + blockIdx.x], (unsigned long long)nspin); }
  float s = 0;
  for (int g = 0; g < NGW; ++g)
    for (int j = 0; j < 16; ++j) s += acc[g][j];
  uint32_t x = 0;
  for (int j = 0; j < 32; ++j) x ^= gat[j];
  if (s == 1.2345f || x == 0x12345678u || hit != 0ull) sink[0] = s;
}

template <int NGW, int MODE>
void run(const char* what, int n_cus, const uint4* codes, unsigned long long* dc, float* ds)
{
  const uint32_t n_tiles = 4096;
  double cyc = 0, spins = 0;
  float ms = 0.f;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t smem = kFlags + 256;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k5<NGW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(dc, 0, 16 * n_cus));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k5<NGW, MODE>), dim3(n_cus), dim3(512), smem, 0, codes, n_tiles, dc, ds);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hbuf(2 * n_cus);
    CHECK(hipMemcpy(hbuf.data(), dc, 16 * n_cus, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < n_cus; ++i) s += (double)hbuf[i];
    spins = 0; for (int i = 0; i < n_cus; ++i) spins += (double)hbuf[n_cus + i];
    cyc = s / n_cus / 8.0 / n_tiles;  // ticks per tile and wave = per tile and SIMD pair
  }
  // a SIMD's two waves multiply every tile of the pair with 2 NGW query groups: 16 NGW MFMAs of 32 cycles per tile and SIMD
  printf("  {\"what\": \"%s\", \"groups_per_wave\": %d, \"mode\": %d, \"ticks_per_tile_and_simd\": %.1f, \"mfma_cycles_per_tile_and_simd\": %d, "
         "\"kernel_ms\": %.3f, \"ticks_per_us\": %.0f, \"flag_spins_per_tile_and_wave\": %.3f},\n", what, NGW, MODE, cyc, 16 * NGW * 32, ms, cyc * n_tiles / (ms * 1e3), spins / n_cus / 8.0 / n_tiles);
  fflush(stdout);
}
int main()
{
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int n = prop.multiProcessorCount;
  unsigned long long* dc; float* ds; uint4* codes;
  CHECK(hipMalloc(&dc, 16 * n)); CHECK(hipMalloc(&ds, 64));
  const size_t nb = (size_t)n * 4 * 64 * 128 * 16;
  CHECK(hipMalloc(&codes, nb));
  std::vector<uint32_t> hc(nb / 4);
  uint32_t x = 12345u;
  for (auto& v : hc) { x = x * 1664525u + 1013904223u; v = x ^ (x >> 13); }
  CHECK(hipMemcpy(codes, hc.data(), nb, hipMemcpyHostToDevice));
  printf("{\"note\": \"two waves per SIMD sharing conflict-free decoded tiles through an LDS ring; ticks of __builtin_readcyclecounter\", \"rows\": [\n");
  run<2, 0>("full: 4 query groups per tile", n, codes, dc, ds);
  run<1, 0>("full: 2 query groups per tile", n, codes, dc, ds);
  run<2, 1>("no decode (tile reads + MFMAs + screens)", n, codes, dc, ds);
  run<2, 2>("no MFMAs", n, codes, dc, ds);
  run<2, 4>("gathers by code into a subspace-major table (bank conflicts)", n, codes, dc, ds);
  run<2, 8>("no flags (protocol cost)", n, codes, dc, ds);
  run<1, 1>("no decode, 2 groups", n, codes, dc, ds);
  printf("  {}\n]}\n");
  return 0;
}
