// Diagnostic (not part of the product): how many cycles a wave64 fp32 vector instruction occupies its SIMD on MI355X --
// the roof the per-pixel chain kernels (k_chain_bwd_static: 722 vector instructions per wave and pixel) are priced against.
// The micro-architecture guide's constants table says v_fma_f32 wave64 = 2 cycles with >= 2 waves per SIMD; DESIGN.md
// (round 2) assumed 4.  One SIMD-resident loop of independent v_fma_f32 / v_pk_fma_f32 / v_add_f32 / v_cndmask chains,
// at 1, 2, 3, 4 waves per SIMD (blocks of 64 threads, occupancy set by the launch: one block per SIMD slot).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue tools/diag/valu_issue.hip && /tmp/valu_issue
// Output: cycles per wave-instruction per SIMD = (kernel cycles of a wave) / (instructions per wave) / (waves per SIMD),
// and the event-timed issue rate (G wave-instructions / s / SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kChains = 16;     // independent dependency chains per lane (latency fully hidden within one wave)
constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(64) void k_issue(const float* in, float* out, unsigned long long* cyc) {
  float a[kChains], b = in[threadIdx.x & 63], c = in[(threadIdx.x & 63) + 64];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[kChains / 2], pb = {b, c}, pc = {c, b};
#pragma unroll
  for (int i = 0; i < kChains; ++i) a[i] = in[i + threadIdx.x];
#pragma unroll
  for (int i = 0; i < kChains / 2; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < kChains; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < kChains / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < kChains; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < kChains; ++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    } else {
#pragma unroll
      for (int i = 0; i < kChains; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < kChains; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < kChains / 2; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int insts_per_iter, float flop_per_inst, const float* in, float* out, unsigned long long* cyc) {
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = 256 * 4 * wps;                       // 256 CUs x 4 SIMDs x wps waves
    k_issue<MODE><<<blocks, 64>>>(in, out, cyc);            // warm-up
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_issue<MODE><<<blocks, 64>>>(in, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double insts = (double)kIters * insts_per_iter;   // wave-instructions per wave
    const double cyc_per = (double)h[blocks / 2] / insts / wps;      // shader cycles (s_memtime) per wave-instruction the SIMD issued
    const double tflops = (double)blocks * insts * 64 * flop_per_inst / (ms * 1e-3) / 1e12;
    printf("%-14s waves/SIMD %d: %8.3f ms  median cycles/wave-inst/SIMD %.3f  -> %.1f TFLOP/s (%.2f G wave-inst/s/SIMD)\n",
           name, wps, ms, cyc_per, tflops, (double)blocks * insts / (ms * 1e-3) / 1e9 / 1024.0);
  }
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 16 * 64 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  std::vector<float> h(4096, 0.999f);
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  printf("(G wave-inst/s/SIMD = issue rate of one SIMD; at a 2.4 GHz clock 2-cycle issue = 1.2, 4-cycle issue = 0.6)\n");
  run<0>("v_fma_f32", kChains, 2.0f, in, out, cyc);
  run<1>("v_pk_fma_f32", kChains / 2, 4.0f, in, out, cyc);
  run<2>("v_add_f32", kChains, 1.0f, in, out, cyc);
  run<4>("v_mul_f32", kChains, 1.0f, in, out, cyc);
  run<3>("v_cmp+cndmask", 2 * kChains, 0.0f, in, out, cyc);
  return 0;
}
