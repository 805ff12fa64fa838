// Diagnostic (not part of the product): where a workgroup of the STRIDE-2 forward k_conv3x3_fwd<1, 2> spends its cycles
// (argv: Ci Co Hout -- the stage entries: 64 64 64 | 64 128 32 | 128 256 16 | 256 512 8).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_CONV_DIAG -Iinclude -o /tmp/fwd_s2_clock tools/diag/fwd_s2_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
namespace t2o { int set_error(int c, const char*) { return c; } }
#include "../../t2onet_amd/csrc/t2o_conv.hip"

int main(int argc, char** argv) {
  const int N = 64, C = argc > 1 ? atoi(argv[1]) : 64, Co = argc > 2 ? atoi(argv[2]) : 64, H = argc > 3 ? atoi(argv[3]) : 64, W = H;
  const size_t act = (size_t)N * 4 * H * W * C, outn = (size_t)N * H * W * Co, wn = (size_t)Co * 9 * C;
  std::vector<float> h(act), hw(wn);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2.0f - 1.0f) * 0.05f;
  float *x, *w, *y, *zero; unsigned long long* st;
  hipMalloc(&x, act * 4); hipMalloc(&y, outn * 4); hipMalloc(&w, wn * 4);
  hipMalloc(&zero, fwd_zero_bytes(2 * C)); hipMemset(zero, 0, fwd_zero_bytes(2 * C));
  hipMemcpy(x, h.data(), act * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice);
  FwdArgs a = {};
  a.x = x; a.w = w; a.y = y; a.zero = zero; a.N = N; a.H = H; a.W = W; a.Ci = C; a.Co = Co;
  const int P = N * H * W;
  const int bm = 1;
  a.tiles_p = (P + 128 * bm - 1) / (128 * bm); a.tiles_n = Co / kFwdCo;
  const unsigned grid = ((a.tiles_p + 7) / 8) * 8 * a.tiles_n;
  hipMalloc(&st, (size_t)grid * 64); hipMemset(st, 0, (size_t)grid * 64);
  a.stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 100; ++rep) {
    if (rep == 99) hipEventRecord(e0);
    k_conv3x3_fwd<1, 2><<<grid, kFwdThreads>>>(a);
    if (rep == 99) hipEventRecord(e1);
  }
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s((size_t)grid * 8);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0;
  std::vector<double> pro, loop, epi, g[4], starts, ends;
  const int stages = 3 * C / 32;
  for (unsigned b = 0; b < grid; ++b) {
    const unsigned long long* q = &s[(size_t)b * 8];
    if (!q[2]) continue;
    tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[6]); starts.push_back((double)q[0]); ends.push_back((double)q[6]);
    pro.push_back((double)q[1]); loop.push_back((double)q[2]); epi.push_back((double)q[7]);
    for (int k = 0; k < 3; ++k) g[k].push_back((double)q[3 + k] / stages);
    g[3].push_back(0);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const double mfma_wave = (double)stages * 48 * bm * 64;
  std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
  auto at = [&](std::vector<double>& v, double q) { return (v[(size_t)(q * (v.size() - 1))] - (double)tmin) * 0.01; };
  printf("stride 2, Ci=%d Co=%d out %dx%d: kernel %.1f us (%.1f TF/s), %u workgroups (%zu ran), %d stages\n", C, Co, H, W, ms * 1e3, 2.0 * 9 * C * Co * N * H * W / ms / 1e9, grid, pro.size(), stages);
  printf("  workgroup starts after the first one (us): 10%% %.1f  50%% %.1f  90%% %.1f  last %.1f;  ends: first %.1f  10%% %.1f  50%% %.1f  90%% %.1f  last %.1f\n",
         at(starts, 0.1), at(starts, 0.5), at(starts, 0.9), at(starts, 1.0), at(ends, 0.0), at(ends, 0.1), at(ends, 0.5), at(ends, 0.9), at(ends, 1.0));
  printf("  per workgroup (median): prologue %.0f, main loop %.0f (MFMA cycles per wave %.0f: two waves per SIMD -> %.0f %% of the loop), epilogue issue %.0f cycles\n",
         med(pro), med(loop), mfma_wave, 100.0 * 2 * mfma_wave / med(loop), med(epi));
  printf("  cycles per group position (24 MFMAs = 1536 pipe cycles; x2 waves = 3072): %.0f %.0f %.0f %.0f\n", med(g[0]), med(g[1]), med(g[2]), med(g[3]));
  return 0;
}
