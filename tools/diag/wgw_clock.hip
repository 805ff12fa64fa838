// Diagnostic (not part of the product): where a wave of k_wino_wgrad spends its cycles, and what each part of a step costs.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_WGW_DIAG [-DT2O_WGW_NO_LOADS | -DT2O_WGW_NO_XFORM |
//         -DT2O_WGW_NO_STORES] -Iinclude -o /tmp/wgw_clock tools/diag/wgw_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
namespace t2o { int set_error(int c, const char*) { return c; } }
extern "C" int t2o_wino_dw_transform(const float*, float*, int, int, int, int, void*) { return 0; }
#include "../../t2onet_amd/csrc/t2o_wino_wgrad.hip"

int main(int argc, char** argv) {
  const int N = argc > 3 ? atoi(argv[3]) : 320, C = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 64, W = H;
  const size_t act = (size_t)N * H * W * C;
  float *x, *dy, *zero, *part; unsigned long long* st;
  hipMalloc(&x, act * 4); hipMalloc(&dy, act * 4);
  hipMemset(x, 0, act * 4); hipMemset(dy, 0, act * 4);
  hipMalloc(&zero, 65536); hipMemset(zero, 0, 65536);
  WgwArgs a = {};
  a.x = x; a.dy = dy; a.zero = zero; a.n_img = N; a.H = H; a.W = W; a.Ci = C; a.Co = C;
  a.steps_total = N * (H / 2) * (W / 16);
  a.splits = wgw_splits(N, H, W, C, C);
  a.tiles_ci = C / 64; a.combos = (C / 64) * (C / 64);
  hipMalloc(&part, (size_t)a.splits * 16 * C * C * 4);
  a.part = part;
  const unsigned grid = ((a.splits + 7) / 8) * 8 * a.combos;
  hipMalloc(&st, (size_t)grid * 4 * 32 * 8); hipMemset(st, 0, (size_t)grid * 4 * 32 * 8);
  a.stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    if (rep == 5) hipEventRecord(e0);
    k_wino_wgrad<<<grid, kWgThreads>>>(a);
    if (rep == 5) hipEventRecord(e1);
  }
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s((size_t)grid * 4 * 32);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> loop, pl[16];
  for (size_t i = 0; i < (size_t)grid * 4; ++i) {
    if (!s[i * 32 + 1]) continue;
    const double steps = (double)s[i * 32 + 1];
    loop.push_back((double)s[i * 32] / steps);
    for (int k = 0; k < 16; ++k) pl[k].push_back((double)s[i * 32 + 2 + k] / steps);
  }
  auto med = [](std::vector<double>& q) { std::sort(q.begin(), q.end()); return q[q.size() / 2]; };
  printf("C=%d %dx%d N=%d: kernel %.1f us, %u workgroups, %d splits (100 MHz s_memtime ticks are scaled by the tool's reader: raw counts below)\n", C, H, W, N, ms * 1e3, grid, a.splits);
  printf("  per step (median over waves): %.0f ticks; planes:", med(loop));
  for (int k = 0; k < 16; ++k) printf(" %.0f", med(pl[k]));
  printf("\n");
  return 0;
}
