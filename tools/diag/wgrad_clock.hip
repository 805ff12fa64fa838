// Diagnostic (not part of the product): in-kernel clock and matrix-pipe occupancy of k_conv3x3_wgrad on one layer.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Iinclude -o /tmp/wgrad_clock tools/diag/wgrad_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
namespace t2o { int set_error(int c, const char*) { return c; } }
#include "../../t2onet_amd/csrc/t2o_conv.hip"

int main(int argc, char** argv) {
  const int N = 64, C = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 32, W = H;
  const size_t act = (size_t)N * H * W * C;
  std::vector<float> h(act);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
  float *x, *dy, *ws; unsigned long long* st;
  const WgradPlan p = wgrad_plan(N, H, W, C, C);
  hipMalloc(&x, act * 4); hipMalloc(&dy, act * 4); hipMalloc(&ws, (size_t)p.splits * C * 9 * C * 4);
  hipMemcpy(x, h.data(), act * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, h.data(), act * 4, hipMemcpyHostToDevice);
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = ws; a.N = N; a.H = H; a.W = W; a.Ci = C; a.Co = C;
  a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.splits = p.splits; a.stages_per_split = p.stages_per_split;
  a.total_stages = p.total_stages;
  const int units = p.splits * p.tiles_m * p.tiles_n;
  const unsigned grid = ((units + 7) / 8) * 72;
  hipMalloc(&st, grid * 16);
  hipMemset(st, 0, grid * 16);
  a.stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 200; ++rep) {                       // ~40 ms of back-to-back launches before the reading that counts
    if (rep == 199) hipEventRecord(e0);
    if (C % 128 == 0) k_conv3x3_wgrad<128, 128, 32, 2><<<grid, kConvThreads>>>(a);
    else k_conv3x3_wgrad<64, 64, 32, 2><<<grid, kConvThreads>>>(a);
    if (rep == 199) hipEventRecord(e1);
  }
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(2 * grid);
  hipMemcpy(s.data(), st, grid * 16, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (unsigned b = 0; b < grid; ++b) if (s[2 * b + 1]) { clk.push_back((double)s[2 * b] / s[2 * b + 1] * 0.1); cyc.push_back((double)s[2 * b]); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double mfma_per_wave = (double)p.stages_per_split * (32 / 2) * (C % 128 == 0 ? 4 : 1);
  printf("C=%d %dx%d: kernel %.1f us, %zu workgroups, in-kernel clock median %.3f GHz (min %.3f max %.3f); workgroup main-loop cycles median %.0f "
         "(max %.0f); MFMA cycles per wave %.0f -> one wave alone would keep its SIMD's pipe %.0f %% busy, two co-resident waves up to 2x that\n",
         C, H, W, ms * 1e3, clk.size(), clk[clk.size() / 2], clk.front(), clk.back(), cyc[cyc.size() / 2], cyc.back(), mfma_per_wave * 64,
         100.0 * mfma_per_wave * 64 / cyc[cyc.size() / 2]);
  return 0;
}
