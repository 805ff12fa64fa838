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

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kConvThreads = 256;
constexpr int kStagePix = 32;                 // K per stage

struct WgradArgs {
  const float* x;      // (N,H,W,Ci)
  const float* dy;     // (N,H,W,Co)
  float* partial;      // (splits, Co, 9, Ci)
  int N, H, W, Ci, Co;
  int tiles_m, tiles_n;       // channel tiles
  int splits, stages_per_split, total_stages;
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int TM, int TN>
__global__ __launch_bounds__(kConvThreads, 2) void k_conv3x3_wgrad(WgradArgs a) {
  constexpr int BM = TM / 64, BN = TN / 64;          // MFMA blocks per wave along m / n (wave tile = TM/2 x TN/2)
  constexpr int RA = TM / 4, RB = TN / 4;            // float4 per tile row
  constexpr int PA = kStagePix * RA / kConvThreads;  // float4 loads per thread per stage (dy / x)
  constexpr int PB = kStagePix * RB / kConvThreads;
  __shared__ __attribute__((aligned(16))) float As[2][kStagePix][TM];
  __shared__ __attribute__((aligned(16))) float Bs[2][kStagePix][TN];

  // block index -> (split, tap, channel tile); blocks of one split are congruent mod 8 (same XCD)
  const int group = 9 * a.tiles_m * a.tiles_n;
  const int b = blockIdx.x;
  const int chunk = b / (8 * group), within = b % (8 * group);
  const int split = chunk * 8 + within % 8;
  const int j = within / 8;
  if (split >= a.splits) return;
  const int tap = j % 9, tile = j / 9;
  const int m0 = (tile / a.tiles_n) * TM, n0 = (tile % a.tiles_n) * TN;
  const int dh = tap / 3 - 1, dw = tap % 3 - 1;

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

  float4 ra[PA], rb[PB];
  auto load_stage = [&](int st) {
    const int pbase = st * kStagePix;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int idx = tid + i * kConvThreads;
      const int row = idx / RA, c4 = idx % RA;
      const int p = pbase + row;
      ra[i] = p < P ? ldg4(a.dy + (size_t)p * a.Co + m0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int idx = tid + i * kConvThreads;
      const int row = idx / RB, c4 = idx % RB;
      const int p = pbase + row;
      const int w = p % a.W, h = (p / a.W) % a.H;
      const bool ok = p < P && (unsigned)(h + dh) < (unsigned)a.H && (unsigned)(w + dw) < (unsigned)a.W;
      rb[i] = ok ? ldg4(a.x + (size_t)(p + dh * a.W + dw) * a.Ci + n0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int idx = tid + i * kConvThreads;
      *reinterpret_cast<float4*>(&As[buf][idx / RA][(idx % RA) * 4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int idx = tid + i * kConvThreads;
      *reinterpret_cast<float4*>(&Bs[buf][idx / RB][(idx % RB) * 4]) = rb[i];
    }
  };

  if (s0 < s1) {
    load_stage(s0);
    store_stage(0);
  }
  __syncthreads();
  for (int st = s0; st < s1; ++st) {
    const int buf = (st - s0) & 1;
    if (st + 1 < s1) load_stage(st + 1);              // global loads in flight under this stage's MFMAs
#pragma unroll
    for (int kk = 0; kk < kStagePix / 2; ++kk) {
      const int k = kk * 2 + lk;
      float fa[BM], fb[BN];
#pragma unroll
      for (int i = 0; i < BM; ++i) fa[i] = As[buf][k][wm * (TM / 2) + i * 32 + lr];
#pragma unroll
      for (int jn = 0; jn < BN; ++jn) fb[jn] = Bs[buf][k][wn * (TN / 2) + jn * 32 + lr];
#pragma unroll
      for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int jn = 0; jn < BN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
    }
    if (st + 1 < s1) store_stage(buf ^ 1);
    __syncthreads();
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

struct WgradPlan { int tm, tn, tiles_m, tiles_n, splits, stages_per_split, total_stages; };

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
  p.total_stages = (P + kStagePix - 1) / kStagePix;
  const int group = 9 * p.tiles_m * p.tiles_n;
  // ~4 workgroups per CU in total (2 resident per CU, two rounds), at least 8 stages of K per workgroup
  int splits = (1024 + group - 1) / group;
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
  const int group = 9 * p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(((p.splits + 7) / 8) * 8 * group);
  hipStream_t st = (hipStream_t)stream;
  if (p.tm == 128 && p.tn == 128) k_conv3x3_wgrad<128, 128><<<grid, kConvThreads, 0, st>>>(a);
  else if (p.tm == 128) k_conv3x3_wgrad<128, 64><<<grid, kConvThreads, 0, st>>>(a);
  else if (p.tn == 128) k_conv3x3_wgrad<64, 128><<<grid, kConvThreads, 0, st>>>(a);
  else k_conv3x3_wgrad<64, 64><<<grid, kConvThreads, 0, st>>>(a);
  const size_t n = (size_t)Co * 9 * Ci, n4 = n / 4;
  k_conv_wgrad_reduce<<<(unsigned)((n4 + kConvThreads - 1) / kConvThreads), kConvThreads, 0, st>>>(
      (const float*)workspace, dw, n4, p.splits, n);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv3x3_wgrad launch failed");
}

}  // extern "C"
