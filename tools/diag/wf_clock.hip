// Diagnostic (not part of the product): where a wave of k_wino_fused spends its cycles.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_WF_DIAG -Iinclude -o /tmp/wf_clock tools/diag/wf_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
namespace t2o { int set_error(int c, const char*) { return c; } }
#include "../../t2onet_amd/csrc/t2o_wino_fused.hip"

int main(int argc, char** argv) {
  const int N = 64, C = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 64, W = H;
  const size_t act = (size_t)N * H * W * C, un = (size_t)16 * C * C;
  std::vector<float> h(act), hu(un);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
  for (auto& v : hu) v = ((float)rand() / RAND_MAX * 2.0f - 1.0f) * 0.05f;
  float *x, *u, *y, *zero; unsigned long long* st;
  hipMalloc(&x, act * 4); hipMalloc(&y, act * 4); hipMalloc(&u, un * 4);
  hipMalloc(&zero, 4096); hipMemset(zero, 0, 4096);
  hipMemcpy(x, h.data(), act * 4, hipMemcpyHostToDevice);
  hipMemcpy(u, hu.data(), un * 4, hipMemcpyHostToDevice);
  WfArgs a = {};
  a.x = x; a.uc = u; a.y = y; a.zero = zero; a.N = N; a.H = H; a.W = W; a.Ci = C; a.Co = C;
  a.blocks = N * (H / 16) * (W / 16); a.tiles_n = C / 64;
  const unsigned grid = ((a.blocks + 7) / 8) * 8 * a.tiles_n;
  hipMalloc(&st, (size_t)grid * 4 * 128); hipMemset(st, 0, (size_t)grid * 4 * 128);
  a.stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 50; ++rep) {
    if (rep == 49) hipEventRecord(e0);
    k_wino_fused<0><<<grid, kWfThreads>>>(a);
    if (rep == 49) hipEventRecord(e1);
  }
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s((size_t)grid * 4 * 16);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v[12];
  for (size_t i = 0; i < (size_t)grid * 4; ++i) {
    if (!s[i * 16 + 1]) continue;
    for (int k = 0; k < 12; ++k) v[k].push_back((double)s[i * 16 + k]);
  }
  auto med = [](std::vector<double>& q) { std::sort(q.begin(), q.end()); return q[q.size() / 2]; };
  const int chunks = C / 8;
  printf("C=%d %dx%d: kernel %.1f us, %u workgroups, %d chunks (shader cycles)\n", C, H, W, ms * 1e3, grid, chunks);
  printf("  per wave (median): prologue %.0f, loop %.0f (= %.0f per chunk; its 64 MFMAs alone: 4096), epilogue issue %.0f cycles\n",
         med(v[0]), med(v[1]), med(v[1]) / chunks, med(v[2]));
  const char* what[8] = {"8 U pieces", "transform rows", "transform columns", "transform stores", "3 x pieces", "-", "-", "barrier, reads"};
  for (int k = 0; k < 8; ++k) printf("  plane pair %d (8 MFMAs = 512 cycles) + %-18s %6.0f cycles per chunk\n", k, what[k], med(v[3 + k]) / chunks);
  printf("  wait for the DMA + barrier %6.0f cycles per chunk\n", med(v[11]) / chunks);
  {   // workgroup lifetimes on the chip-wide 100 MHz clock: entry of its first instruction .. its last stamp; and the whole launch
    std::vector<double> life, mhz, pre; unsigned long long lo = ~0ull, hi = 0;
    for (size_t i = 0; i < (size_t)grid * 4; ++i) {
      if (!s[i * 16 + 1] || !s[i * 16 + 13]) continue;
      life.push_back((double)(s[i * 16 + 13] - s[i * 16 + 12]) * 0.01);
      if (s[i * 16 + 15] > s[i * 16 + 14]) mhz.push_back((double)s[i * 16 + 1] / ((double)(s[i * 16 + 15] - s[i * 16 + 14]) * 0.01));
      pre.push_back((double)(s[i * 16 + 14] - s[i * 16 + 12]) * 0.01);
      lo = std::min(lo, s[i * 16 + 12]); hi = std::max(hi, s[i * 16 + 13]);
    }
    const double rounds = (double)grid / 256.0;
    printf("  shader clock over the loop: %.0f MHz; entry .. loop start %.2f us\n", med(mhz), med(pre));
    printf("  wave lifetime (entry .. last stamp): median %.2f us; first entry .. last exit %.1f us = %.2f us per round of 256 workgroups\n",
           med(life), (double)(hi - lo) * 0.01, (double)(hi - lo) * 0.01 / rounds);
  }
  return 0;
}
