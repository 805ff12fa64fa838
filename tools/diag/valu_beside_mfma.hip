// Diagnostic (not part of the product): how long do ordinary instructions of one wave take while the OTHER wave on
// the same SIMD streams fp32 MFMAs?  512-thread workgroups, one per CU: waves 0..3 run the probe (32 independent
// instructions of one class, then a dependent v_readfirstlane that waits for all results), waves 4..7 either idle
// (partner 0) or stream back-to-back v_mfma_f32_32x32x2_f32 (partner 1: bare stream, 12 independent accumulators;
// partner 2: the same stream with two ds_read_b32 + a wait per 12 MFMAs, like a GEMM loop).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/valu_beside_mfma tools/diag/valu_beside_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int CLS>
__global__ __launch_bounds__(512, 1) void k(unsigned long long* out, int iters, int partner, float* sink) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (wave >= 4) {
    if (!partner) return;
    f32x16 acc[12] = {};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    for (int it = 0; it < iters * 24; ++it) {
      if (partner == 2) { a += lds[(threadIdx.x + it) & 4095]; b += lds[(threadIdx.x * 3 + it) & 4095]; }
#pragma unroll
      for (int r = 0; r < 12; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 12; ++r) s += acc[r][0];
    if (s == 123.0f) sink[0] = 1.0f;
    return;
  }
  unsigned long long total = 0;
  unsigned x = lane + 1, y = lane * 3 + 7, r0 = 0;
  unsigned long long z = lane;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (CLS == 0) { REP32(asm volatile("v_add_u32 %0, %1, %2" : "=v"(r0) : "v"(x), "v"(y));) }
    if constexpr (CLS == 1) { REP32(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r0) : "v"(x), "v"(y));) }
    if constexpr (CLS == 2) { REP32(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(z) : "v"(x), "v"(y) : "vcc");) }
    if constexpr (CLS == 3) { REP32(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(z) : "v"(z));) }
    if constexpr (CLS == 4) { REP32(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r0) : "v"(x), "v"(y) : "vcc");) }
    if constexpr (CLS == 5) { REP32(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r0) : "v"(x));) }
    if constexpr (CLS == 6) { REP32(asm volatile("s_add_u32 %0, %0, 1" : "+s"(r0) :: "scc");) }
    if constexpr (CLS == 7) { REP32(asm volatile("s_mul_i32 %0, %0, 3" : "+s"(r0));) }
    if constexpr (CLS == 8) { REP32(asm volatile("v_cmp_le_u32 vcc, %0, %1\n\ts_and_b64 %2, vcc, exec" :: "v"(x), "v"(y), "s"(0ull) : "vcc");) }
    if constexpr (CLS == 9) { REP32(asm volatile("v_rcp_iflag_f32 %0, %1" : "=v"(r0) : "v"(x));) }
    if constexpr (CLS == 10) { REP32(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(x), "v"(y));) }
    if constexpr (CLS == 11) { REP32(asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(x));) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    unsigned fin = __builtin_amdgcn_readfirstlane(r0 + (unsigned)z);
    asm volatile("" :: "s"(fin));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    total += t1 - t0;
    x += fin & 1;
    __builtin_amdgcn_s_sleep(4);
  }
  if (lane == 0) out[blockIdx.x * 4 + wave] = total;
}

int main() {
  const int grid = 256, iters = 200;
  unsigned long long* out; hipMalloc(&out, grid * 4 * 8);
  float* sink; hipMalloc(&sink, 64);
  const char* names[12] = {"v_add_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_lshl_add_u64", "v_cndmask_b32", "v_readlane_b32", "s_add_u32",
                           "s_mul_i32", "v_cmp + s_and_b64", "v_rcp_iflag_f32", "v_fma_f32", "ds_read_b32"};
  for (int c = 0; c < 12; ++c) {
    double med[3];
    for (int partner = 0; partner < 3; ++partner) {
      hipMemset(out, 0, grid * 4 * 8);
#define L(C) case C: k<C><<<grid, 512>>>(out, iters, partner, sink); break;
      switch (c) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) }
#undef L
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(grid * 4);
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> v; for (auto t : h) if (t) v.push_back((double)t / iters);
      std::sort(v.begin(), v.end());
      med[partner] = v.empty() ? 0 : v[v.size() / 2];
    }
    printf("%-18s 32 instructions + result wait: alone %6.0f cyc (%5.1f each)   beside a bare MFMA stream %7.0f (%6.1f each)   beside MFMAs + LDS reads %7.0f (%6.1f each)\n",
           names[c], med[0], med[0] / 32, med[1], med[1] / 32, med[2], med[2] / 32);
  }
  return 0;
}
