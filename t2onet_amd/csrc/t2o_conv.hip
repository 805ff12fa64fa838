// t2o_conv.hip -- fp32 convolutions of the actor's image encoder (models/actor_resnet.py:27-44: the 3x3,
// stride-1, padding-1 convolutions of the BasicBlocks, 12 of the encoder's 21 convolutions and ~80 % of its
// FLOPs) as implicit GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per product,
// 64 FLOP/clk/SIMD = the chip's 157 TFLOP/s fp32 peak).  This is the one dense contraction of the hot path -- the
// only place MFMA belongs (everything else is per-pixel, HBM- or VALU-bound work).
//
// Layout: activations NHWC ((N,H,W,C) rows of C contiguous channels = torch.channels_last), weights
// (Co, 3, 3, Ci) (= channels_last storage of a (Co,Ci,3,3) weight).
//
// WEIGHT GRADIENT   dw[co][tap][ci] = sum_p dy[p][co] * x[p + shift(tap)][ci]      (p over all N*H*W pixels)
//   One GEMM per tap: M = Co, N = Ci, K = pixels.  Both operands are stored pixel-major with the channel
//   contiguous, which IS the MFMA operand order (lane l supplies A[m = l % 32][k = l / 32]: for one pixel k the 32
//   lanes read 32 consecutive channels), so tiles go global -> LDS as plain 16-byte row copies and fragments are
//   conflict-free ds_read_b32 -- no transposes anywhere.
//   Workgroup = 256 threads = 2 x 2 waves, tile TM x TN channels (128 x 128: 2 x 2 MFMA blocks per wave, 64
//   accumulator registers; 64 x 64 for the 64-channel stage), K consumed in stages of 32 pixels through two LDS
//   buffers (the next stage's global loads are issued before the current stage's MFMAs).
//   K is split across workgroups (a stage-aligned pixel range each); every workgroup writes its TM x TN tile of
//   its own partial (split, Co, 9, Ci) array and a second kernel adds the partials in split order: deterministic,
//   no float atomics.  Workgroups of one pixel range (9 taps x channel tiles: they re-read the same dy / x rows) get
//   block indices congruent mod 8, i.e. land on the same XCD and share its L2.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kConvThreads = 256;

struct WgradArgs {
  const float* x;      // (N,H,W,Ci)
  const float* dy;     // (N,H,W,Co)
  float* partial;      // (splits, Co, 9, Ci)
  int N, H, W, Ci, Co;
  int tiles_m, tiles_n;       // channel tiles
  int splits, stages_per_split, total_stages;
  unsigned long long* stamps;   // diagnostic builds only (tools/diag/wgrad_clock.hip): per-workgroup clock stamps; null in the product
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int TM, int TN, int kStagePix, int kMinWaves>
__global__ __launch_bounds__(kConvThreads, kMinWaves) void k_conv3x3_wgrad(WgradArgs a) {
  constexpr int BM = TM / 64, BN = TN / 64;          // MFMA blocks per wave along m / n (wave tile = TM/2 x TN/2)
  constexpr int RA = TM / 4, RB = TN / 4;            // float4 per tile row
  constexpr int PA = kStagePix * RA / kConvThreads;  // float4 loads per thread per stage (dy / x)
  constexpr int PB = kStagePix * RB / kConvThreads;
  __shared__ __attribute__((aligned(16))) float As[2][kStagePix][TM];
  __shared__ __attribute__((aligned(16))) float Bs[2][kStagePix][TN];

  // block index -> (split, channel tile, tap).  The 9 taps of one (split, tile) re-read the same dy / x tile rows:
  // they get block indices congruent mod 8 (same XCD, shared L2) and adjacent in dispatch order; consecutive
  // (split, tile) units go to consecutive XCDs, so every XCD gets the same number of workgroups.
  const int units = a.splits * a.tiles_m * a.tiles_n;
  const int b = blockIdx.x;
  const int unit = (b / 72) * 8 + b % 8;
  const int tap = (b % 72) / 8;
  if (unit >= units) return;
  const int tiles = a.tiles_m * a.tiles_n;
  const int split = unit / tiles, tile = unit % tiles;
  const int m0 = (tile / a.tiles_n) * TM, n0 = (tile % a.tiles_n) * TN;
  const int dh = tap / 3 - 1, dw = tap % 3 - 1;

  unsigned long long t0 = 0, r0 = 0;
  if (a.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  const int P = a.N * a.H * a.W;
  const int s0 = split * a.stages_per_split;
  int s1 = s0 + a.stages_per_split;
  if (s1 > a.total_stages) s1 = a.total_stages;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lk = lane >> 5;

  f32x16 acc[BM][BN];
#pragma unroll
  for (int i = 0; i < BM; ++i)
#pragma unroll
    for (int jn = 0; jn < BN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;

  float4 ra[2][PA], rb[2][PB];                     // two register sets: the global loads run TWO stages ahead
  // per load slot: the pixel's (h, w), advanced by one stage (32 pixels) at a time -- no division in the loop
  int ph[PB], pw[PB];
  {
    const int pbase = s0 * kStagePix;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int p = pbase + (tid + i * kConvThreads) / RB;
      pw[i] = p % a.W;
      ph[i] = (p / a.W) % a.H;
    }
  }
  const int adv_w = kStagePix % a.W, adv_h = kStagePix / a.W;
  auto load_stage = [&](int st, int set) {
    const int pbase = st * kStagePix;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int idx = tid + i * kConvThreads;
      const int row = idx / RA, c4 = idx % RA;
      const int p = pbase + row;
      ra[set][i] = p < P ? ldg4(a.dy + (size_t)p * a.Co + m0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int idx = tid + i * kConvThreads;
      const int row = idx / RB, c4 = idx % RB;
      const int p = pbase + row;
      const bool ok = p < P && (unsigned)(ph[i] + dh) < (unsigned)a.H && (unsigned)(pw[i] + dw) < (unsigned)a.W;
      rb[set][i] = ok ? ldg4(a.x + (size_t)(p + dh * a.W + dw) * a.Ci + n0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      pw[i] += adv_w;                                   // -> the same slot's pixel of the next stage
      const int carry = pw[i] >= a.W ? 1 : 0;
      pw[i] -= carry * a.W;
      ph[i] += adv_h + carry;
      while (ph[i] >= a.H) ph[i] -= a.H;
    }
  };
  auto store_stage = [&](int buf, int set) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int idx = tid + i * kConvThreads;
      *reinterpret_cast<float4*>(&As[buf][idx / RA][(idx % RA) * 4]) = ra[set][i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int idx = tid + i * kConvThreads;
      *reinterpret_cast<float4*>(&Bs[buf][idx / RB][(idx % RB) * 4]) = rb[set][i];
    }
  };

  // One continuous MFMA stream across stages, global loads two stages ahead.  Stage st (LDS buffer q = parity of
  // st - s0, kStagePix / 2 k-pairs, 4 MFMAs of 64 cycles each per pair):
  //   k-pair 0        the loads of stage st+2 are issued into register set q (its previous content, stage st, went to
  //                   LDS during stage st-1): they have a whole stage (~10k cycles) to land, HBM latency never shows
  //   k-pairs kS..    one 16-byte LDS store per k-pair of stage st+1 (register set q^1, loaded during stage st-1)
  //                   into the OTHER buffer
  //   k-pair  last-1  barrier: every wave's stores are done -- it does not have to wait for anybody's reads of the
  //                   current buffer, and the matrix pipe still holds 256 cycles of this wave's MFMAs to cover the skew
  //   k-pair  last    its fragment prefetch already reads k-pair 0 of stage st+1, so the first MFMAs after the stage
  //                   boundary find their operands in registers: no bubble at the boundary
  constexpr int KP = kStagePix / 2;                       // k-pairs per stage
  constexpr int NST = PA + PB;                            // LDS stores per thread per stage
  constexpr int kS = 1;                                   // first k-pair that carries a store
  static_assert(kS + NST <= KP - 1, "stage too short for its stores");
  float fa[2][BM], fb[2][BN];
  auto read_frags = [&](int buf, int kk, int slot) {
    const int k = kk * 2 + lk;
#pragma unroll
    for (int i = 0; i < BM; ++i) fa[slot][i] = As[buf][k][wm * (TM / 2) + i * 32 + lr];
#pragma unroll
    for (int jn = 0; jn < BN; ++jn) fb[slot][jn] = Bs[buf][k][wn * (TN / 2) + jn * 32 + lr];
  };
  // one stage; Q (compile time) = LDS buffer and register-set parity of this stage
  auto stage = [&](auto Qc, int st) {
    constexpr int Q = decltype(Qc)::value;
    const bool has_next = st + 1 < s1;
    if (st + 2 < s1) load_stage(st + 2, Q);
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < KP) read_frags(Q, kk + 1, nxt);
      else if (has_next) read_frags(Q ^ 1, 0, nxt);       // (after the barrier of k-pair KP-2)
      if (has_next && kk >= kS && kk < kS + NST) {
        const int i = kk - kS;
        if (i < PA) {
          const int idx = tid + i * kConvThreads;
          *reinterpret_cast<float4*>(&As[Q ^ 1][idx / RA][(idx % RA) * 4]) = ra[Q ^ 1][i < PA ? i : 0];
        } else {
          const int idx = tid + (i - PA) * kConvThreads;
          *reinterpret_cast<float4*>(&Bs[Q ^ 1][idx / RB][(idx % RB) * 4]) = rb[Q ^ 1][i >= PA ? i - PA : 0];
        }
      }
      __builtin_amdgcn_sched_barrier(0);                  // keep reads / stores above the MFMAs (the scheduler sinks them)
#pragma unroll
      for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int jn = 0; jn < BN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][jn], acc[i][jn], 0, 0, 0);
      if (kk == KP - 2) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (s0 < s1) {
    load_stage(s0, 0);
    store_stage(0, 0);
    if (s0 + 1 < s1) load_stage(s0 + 1, 1);
  }
  __syncthreads();
  if (s0 < s1) read_frags(0, 0, 0);
  for (int st = s0; st < s1; st += 2) {
    stage(std::integral_constant<int, 0>{}, st);
    if (st + 1 < s1) stage(std::integral_constant<int, 1>{}, st + 1);
  }

  if (a.stamps && threadIdx.x == 0) {
    a.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  // C/D layout: column (n) = lane % 32, row (m) = (reg % 4) + 8 * (reg / 4) + 4 * (lane / 32)
  float* out = a.partial + (size_t)split * a.Co * 9 * a.Ci;
#pragma unroll
  for (int i = 0; i < BM; ++i)
#pragma unroll
    for (int jn = 0; jn < BN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int n = n0 + wn * (TN / 2) + jn * 32 + lr;
        out[((size_t)m * 9 + tap) * a.Ci + n] = acc[i][jn][r];
      }
}

// dw[i] = sum over splits of partial[s][i], in split order; 4 floats per thread
__global__ __launch_bounds__(kConvThreads) void k_conv_wgrad_reduce(const float* partial, float* dw, size_t n4, int splits, size_t stride) {
  const size_t i = (size_t)blockIdx.x * kConvThreads + threadIdx.x;
  if (i >= n4) return;
  float4 s = ldg4(partial + 4 * i);
  for (int k = 1; k < splits; ++k) {
    const float4 v = ldg4(partial + (size_t)k * stride + 4 * i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(dw + 4 * i) = s;
}

struct WgradPlan { int tm, tn, tiles_m, tiles_n, splits, stages_per_split, total_stages, stage_pix; };

int conv_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

bool wgrad_supported(int N, int H, int W, int Ci, int Co) {
  return N > 0 && H > 0 && W > 0 && Ci >= 64 && Co >= 64 && Ci % 64 == 0 && Co % 64 == 0 &&
         (size_t)N * H * W < ((size_t)1 << 30);
}

WgradPlan wgrad_plan(int N, int H, int W, int Ci, int Co) {
  WgradPlan p;
  p.tm = (Co % 128 == 0) ? 128 : 64;
  p.tn = (Ci % 128 == 0) ? 128 : 64;
  p.tiles_m = Co / p.tm;
  p.tiles_n = Ci / p.tn;
  const int P = N * H * W;
  static const int stage_pix = conv_env("T2O_WGRAD_STAGE", 32);          // 16: three workgroups per CU (A/B runs)
  p.stage_pix = stage_pix == 16 ? 16 : 32;
  p.total_stages = (P + p.stage_pix - 1) / p.stage_pix;
  const int group = 9 * p.tiles_m * p.tiles_n;
  // ONE round of workgroups: 2 are resident per CU (LDS), 64 slots per XCD; a second, partly filled round would
  // cost a whole round's time.  (split, tile) units are dealt to the 8 XCDs in turn, 9 workgroups (taps) each:
  // 7 units per XCD = 63 slots.  At least 8 stages of K per workgroup.
  static const int units = conv_env("T2O_WGRAD_UNITS", 0);
  int splits = (units > 0 ? units : p.stage_pix == 16 ? 80 : 56) / (p.tiles_m * p.tiles_n);
  (void)group;
  if (splits > p.total_stages / 8) splits = p.total_stages / 8;
  if (splits < 1) splits = 1;
  p.stages_per_split = (p.total_stages + splits - 1) / splits;
  p.splits = (p.total_stages + p.stages_per_split - 1) / p.stages_per_split;
  return p;
}

}  // namespace

extern "C" {

size_t t2o_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
  if (!wgrad_supported(N, H, W, Ci, Co)) return 0;
  const WgradPlan p = wgrad_plan(N, H, W, Ci, Co);
  return sizeof(float) * (size_t)p.splits * Co * 9 * Ci;
}

int t2o_conv3x3_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, void* stream) {
  if (!x || !dy || !dw) return set_error(T2O_EINVAL, "conv3x3_wgrad: null pointer");
  if (!wgrad_supported(N, H, W, Ci, Co))
    return set_error(T2O_EUNSUPPORTED, "conv3x3_wgrad: channel counts must be multiples of 64 (>= 64)");
  if (!workspace || workspace_bytes < t2o_conv3x3_wgrad_workspace_bytes(N, H, W, Ci, Co))
    return set_error(T2O_EWORKSPACE, "conv3x3_wgrad: workspace too small");
  const WgradPlan p = wgrad_plan(N, H, W, Ci, Co);
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = (float*)workspace;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n;
  a.splits = p.splits; a.stages_per_split = p.stages_per_split; a.total_stages = p.total_stages;
  a.stamps = nullptr;
  const int units = p.splits * p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(((units + 7) / 8) * 72);
  hipStream_t st = (hipStream_t)stream;
#define T2O_WGRAD_LAUNCH(TM_, TN_)                                                                     \
  do {                                                                                                 \
    if (p.stage_pix == 16) k_conv3x3_wgrad<TM_, TN_, 16, 3><<<grid, kConvThreads, 0, st>>>(a);           \
    else k_conv3x3_wgrad<TM_, TN_, 32, 2><<<grid, kConvThreads, 0, st>>>(a);                              \
  } while (0)
  if (p.tm == 128 && p.tn == 128) T2O_WGRAD_LAUNCH(128, 128);
  else if (p.tm == 128) T2O_WGRAD_LAUNCH(128, 64);
  else if (p.tn == 128) T2O_WGRAD_LAUNCH(64, 128);
  else T2O_WGRAD_LAUNCH(64, 64);
#undef T2O_WGRAD_LAUNCH
  const size_t n = (size_t)Co * 9 * Ci, n4 = n / 4;
  k_conv_wgrad_reduce<<<(unsigned)((n4 + kConvThreads - 1) / kConvThreads), kConvThreads, 0, st>>>(
      (const float*)workspace, dw, n4, p.splits, n);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv3x3_wgrad launch failed");
}

}  // extern "C"
