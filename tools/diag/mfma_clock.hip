// Diagnostic (not part of the product): the clock an MI355X holds under a dense fp32 MFMA loop on random data, and
// the FLOP rate that clock allows -- the practical ceiling for the encoder's fp32 convolutions.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_clock tools/diag/mfma_clock.hip && /tmp/mfma_clock
// In-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma_loop(const float* in, float* out, unsigned long long* stamps, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float a0 = in[tid], b0 = in[tid + 1], a1 = in[tid + 2], b1 = in[tid + 3];
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c[0], 0, 0, 0);
    c[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c[1], 0, 0, 0);
    c[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c[2], 0, 0, 0);
    c[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c[3], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
  out[tid] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks = 256 * 2, iters = 200000;           // 2 workgroups (8 waves) per CU: 2 waves per SIMD
  const size_t n = (size_t)blocks * 256 + 4;
  std::vector<float> h(n);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&st, blocks * 16);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    k_mfma_loop<<<blocks, 256>>>(in, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(2 * blocks);
    hipMemcpy(s.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int b = 0; b < blocks; ++b) clk.push_back((double)s[2 * b] / (double)s[2 * b + 1] * 0.1);   // GHz
    std::sort(clk.begin(), clk.end());
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4.0 * 32 * 32 * 2 * 2;
    printf("rep %d: %.1f ms, %.1f TFLOP/s, in-kernel clock median %.3f GHz (min %.3f max %.3f) -> peak at that clock %.1f TFLOP/s\n",
           rep, ms, flop / ms / 1e9, clk[blocks / 2], clk.front(), clk.back(), 1024 * 64.0 * clk[blocks / 2] / 1e3);
  }
  return 0;
}
