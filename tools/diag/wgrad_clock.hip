// Diagnostic (not part of the product): in-kernel clock and matrix-pipe occupancy of k_conv3x3_wgrad on one layer.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_CONV_DIAG -Iinclude -o /tmp/wgrad_clock tools/diag/wgrad_clock.hip
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
  const unsigned grid = ((units + 7) / 8) * 24;
  const size_t stamp_bytes = (size_t)grid * 16 + (size_t)grid * 18 * 8;
  hipMalloc(&st, stamp_bytes);
  hipMemset(st, 0, stamp_bytes);
  a.stamps = st;
  float* zp; hipMalloc(&zp, p.zero_bytes); hipMemset(zp, 0, p.zero_bytes); a.zero = zp;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 200; ++rep) {                       // ~40 ms of back-to-back launches before the reading that counts
    if (rep == 199) hipEventRecord(e0);
    if (p.tm == 128 && p.tn == 128) k_conv3x3_wgrad<128, 128, 2, 1><<<grid, kConvThreads>>>(a);
    else if (p.tm == 128) k_conv3x3_wgrad<128, 64, 2, 1><<<grid, kConvThreads>>>(a);
    else k_conv3x3_wgrad<64, 64, 2, 1><<<grid, kConvThreads>>>(a);
    if (rep == 199) hipEventRecord(e1);
  }
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(2 * grid + (size_t)grid * 18);
  hipMemcpy(s.data(), st, stamp_bytes, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (unsigned b = 0; b < grid; ++b) if (s[2 * b + 1]) { clk.push_back((double)s[2 * b] / s[2 * b + 1] * 0.1); cyc.push_back((double)s[2 * b]); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double mfma_per_wave = (double)p.stages_per_split * (32 / 2) * 3 * (p.tm / 64) * (p.tn / 64);
  printf("C=%d %dx%d: kernel %.1f us, %zu workgroups, in-kernel clock median %.3f GHz (min %.3f max %.3f); workgroup main-loop cycles median %.0f "
         "(max %.0f); MFMA cycles per wave %.0f -> one wave alone would keep its SIMD's pipe %.0f %% busy, two co-resident waves up to 2x that\n",
         C, H, W, ms * 1e3, clk.size(), clk[clk.size() / 2], clk.front(), clk.back(), cyc[cyc.size() / 2], cyc.back(), mfma_per_wave * 64,
         100.0 * mfma_per_wave * 64 / cyc[cyc.size() / 2]);
  {
    printf("  workgroup main-loop cycles, quantiles 0/10/25/50/75/90/100 %%:");
    const double qs[7] = {0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0};
    for (double q : qs) printf(" %.0f", cyc[(size_t)(q * (cyc.size() - 1))]);
    printf("\n  mean by XCD (block %% 8):");
    for (int x = 0; x < 8; ++x) { double t = 0; int n = 0; for (unsigned b = x; b < grid; b += 8) if (s[2 * b + 1]) { t += (double)s[2 * b]; ++n; } printf(" %.0f(%d)", n ? t / n : 0.0, n); }
    printf("\n  mean by kernel row:");
    for (int kh = 0; kh < 3; ++kh) { double t = 0; int n = 0; for (unsigned b = 0; b < grid; ++b) if ((b % 24) / 8 == (unsigned)kh && s[2 * b + 1]) { t += (double)s[2 * b]; ++n; } printf(" %.0f(%d)", n ? t / n : 0.0, n); }
    printf("\n  by dispatch order, mean of 32 consecutive blocks:");
    for (unsigned b0 = 0; b0 < grid; b0 += 32) { double t = 0; int n = 0; for (unsigned b = b0; b < b0 + 32 && b < grid; ++b) if (s[2 * b + 1]) { t += (double)s[2 * b]; ++n; } printf(" %.0f", n ? t / n / 1000 : 0.0); }
    printf(" (kcyc)\n");
  }
  // cycles per k-pair position, averaged over the stages of each workgroup (wave 0), median over workgroups
  printf("  cycles per k-pair position (12 or 3 MFMAs of 64 pipe cycles; x2 with a co-resident wave), mean over the workgroup's stages:");
  double total = 0;
  for (int kk = 0; kk < 16; ++kk) {
    std::vector<double> d;
    for (unsigned b = 0; b < grid; ++b) { const unsigned long long* q = &s[2 * grid + (size_t)b * 18]; if (q[16]) d.push_back((double)q[kk] / (double)q[16]); }
    if (d.empty()) continue;
    std::sort(d.begin(), d.end());
    printf(" %.0f", d[d.size() / 2]);
    total += d[d.size() / 2];
  }
  {
    std::vector<double> d, ad;
    for (unsigned b = 0; b < grid; ++b) { const unsigned long long* q = &s[2 * grid + (size_t)b * 18]; if (q[16]) { d.push_back((double)(q[17] & 0xffffffffull) / (double)q[16]); ad.push_back((double)(q[17] >> 32) / (double)q[16]); } }
    std::sort(d.begin(), d.end()); std::sort(ad.begin(), ad.end());
    printf("  + DMA phase %.0f", d.empty() ? 0.0 : d[d.size() / 2]);
    total += d.empty() ? 0.0 : d[d.size() / 2];
  }
  printf("  (sum %.0f)\n", total);
  return 0;
}
