// t2o_conv_generic.hip -- the encoder's 3x3 convolutions (models/actor_resnet.py:27-36, padding 1, stride 1 or 2) for
// ANY image size, on the fp32 matrix cores: forward, data gradient, weight gradient.
//
// The fast kernels (t2o_conv.hip) move whole 1 KiB pieces by LDS-DMA and therefore want the image width to be a multiple
// of 8 (forward / data gradient) or 4 (weight gradient) and even sizes under stride 2 -- true for every stage of a
// 256 x 256 image.  At the reference's training size (128 x 128: the last stage is 4 x 4) and at full-resolution inference
// (short side 600: 300 -> 150 -> 75 -> 38 -> 19) they do not apply, and those layers went to the library (MIOpen).  These
// kernels take them instead: implicit GEMMs whose A rows are GATHERED per (output pixel, tap) with a per-row validity
// (zero padding, image borders, stride-2 parity), staged through registers into LDS -- the tiling of t2o_conv1x1.hip
// with nine taps.  They reach roughly half the fast kernels' rate; what they buy is coverage: with them no convolution of
// the encoder is a library call at any image size, every gradient is deterministic, and the one-node trunk (encoder.py)
// runs for every even-sized training image.
//   mode FWD   y[n][oh][ow][co]  = sum_{kh,kw,ci} x[n][oh*s+kh-1][ow*s+kw-1][ci] w[co][kh][kw][ci]
//   mode DG1   dx = the same sum over dy with the tap-mirrored transposed weight (stride 1)
//   mode DG2   dx[n][i][j][ci]   = sum over the taps with (i+1-kh), (j+1-kw) even and inside the dy grid of
//                                  dy[n][(i+1-kh)/2][(j+1-kw)/2][co] wt[ci][kh][kw][co]                 (stride 2)
//   WGRAD      dw[co][kh][kw][ci] = sum_{n,oh,ow} dy[n][oh][ow][co] x[n][oh*s+kh-1][ow*s+kw-1][ci]     (split-K, fixed order)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "t2onet_hip.h"

namespace t2o {
int set_error(int code, const char* msg);
void launch_wgrad_reduce(const float* partial, float* dw, size_t n, int splits, int accumulate, hipStream_t st);   // t2o_conv.hip
}
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct GArgs {
  const float* A;        // gathered operand: x (FWD), dy (DG1, DG2): (N, Ha, Wa, K)
  const float* B;        // (Ncols, 9, K): w for FWD, the transformed weight for DG1 / DG2
  const float* addend;   // null, or (N, Hc, Wc, Ncols) added to C (DG1: the identity shortcut's gradient)
  float* C;              // (N, Hc, Wc, Ncols)
  int N, Ha, Wa, Hc, Wc; // A grid, C grid
  int K, Ncols;
  int mode;              // 0 FWD stride 1, 1 FWD stride 2, 2 DG2 (DG1 = FWD stride 1 on dy)
  int tiles_m, tiles_n;
};

// A row (in pixels of the A grid) that output pixel (n, oh, ow) reads for tap (kh, kw), and whether it exists (n < 0: a row past
// the batch).  Branch-free -- the mode is a template argument, the tests are bit operations: with run-time branches here the
// chunk loop was 169 basic blocks and every load sat behind one (round 6).
template <int MODE>
__device__ __forceinline__ long long a_row(const GArgs& g, int n, int oh, int ow, int kh, int kw, bool& ok) {
  int ih, iw, bad = n;                                     // (bad < 0 <=> the row does not exist)
  if constexpr (MODE == 2) {
    const int a = oh + 1 - kh, b = ow + 1 - kw;
    bad |= a | b | -((a | b) & 1);                         // negative, or odd
    ih = a >> 1; iw = b >> 1;
  } else {
    constexpr int s = MODE == 1 ? 2 : 1;
    ih = oh * s + kh - 1; iw = ow * s + kw - 1;
  }
  bad |= ih | iw | (g.Ha - 1 - ih) | (g.Wa - 1 - iw);
  ok = bad >= 0;
  const long long row = ((long long)n * g.Ha + ih) * g.Wa + iw;
  return ok ? row : 0;
}

// TM output pixels x 64 output channels per workgroup, K = (tap, 32 channels) chunks; see k_sc_gemm (t2o_conv1x1.hip) for
// the LDS layout (16-byte chunks swizzled by row, one conflict-free ds_read_b128 = four k-steps of a lane).  TM = 128: a wave owns
// 32 pixels x 64 channels (two MFMA blocks); TM = 64: 32 x 32 (one block) -- twice the workgroups where 128-pixel tiles leave CUs
// idle (the 256 -> 512 stride-2 layer of a 128 x 128 image: 64 of them for 256 CUs, round 6).  The same sums in the same order.
template <int MODE, int TM>
__global__ __launch_bounds__(kThreads) void k_gconv(GArgs g) {
  constexpr int RJ = TM / 32, NB = TM / 64;               // A pieces per thread; MFMA blocks per wave
  __shared__ float4 As[2][TM * 8];
  __shared__ float4 Bs[2][64 * 8];
  const int b = blockIdx.x;
  const int xcd = b % 8, k8 = b / 8;
  const int rt = (k8 / g.tiles_n) * 8 + xcd, ct = k8 % g.tiles_n;
  if (rt >= g.tiles_m) return;
  const int Q = g.N * g.Hc * g.Wc;
  const int q0 = rt * TM, n0 = ct * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, lh = lane >> 5;
  const int wrow = TM == 128 ? wave : (wave >> 1), wcol = TM == 128 ? 0 : (wave & 1);     // this wave's 32-pixel row block, first column block
  const int lrow = tid >> 3, lc = tid & 7;
  const int lswz = (lrow >> 1) & 7;
  int pn[RJ], poh[RJ], pow_[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const int q = q0 + lrow + 32 * j;
    if (q < Q) {
      const int hw = g.Hc * g.Wc;
      pn[j] = q / hw;
      const int r = q - pn[j] * hw;
      poh[j] = r / g.Wc;
      pow_[j] = r - poh[j] * g.Wc;
    } else {
      pn[j] = -1; poh[j] = 0; pow_[j] = 0;
    }
  }
  const size_t brow0 = (size_t)(n0 + lrow) * 9 * g.K + lc * 4, brow1 = brow0 + (size_t)32 * 9 * g.K;
  const int chunks = g.K / 32, nk = 9 * chunks;
  // Two chunks travel in registers (sets 0 / 1) while a third is multiplied out of LDS: with ONE in flight the loop was a chain of
  // global round trips -- 1.7 us per chunk, 125 us for the 72 chunks of the 256 -> 512 stride-2 layer of a 128 x 128 image whose
  // matrix work is 15 us (round 6, profiles/r06_step128_kernel_stats.csv: 10 launches, 1.95 ms of an 18 ms step).
  float4 ra[2][RJ], rb[2][2];
  auto gload = [&](auto sc, int kc) {
    constexpr int S = decltype(sc)::value;
    const int tap = kc / chunks, cc = kc - tap * chunks;
    const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
      // (the load is ISSUED for every lane -- row 0 stands in for a padding / parity / out-of-batch row -- and the value
      // selected afterwards: a load under a branch makes the compiler wait for everything outstanding at the next use,
      // and the two chunks in flight became one again)
      bool ok;
      const long long r = a_row<MODE>(g, pn[j], poh[j], pow_[j], kh, kw, ok);
      const float4 v = ld4(g.A + (size_t)r * g.K + cc * 32 + lc * 4);
      ra[S][j] = ok ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    rb[S][0] = ld4(g.B + brow0 + (size_t)tap * g.K + cc * 32);
    rb[S][1] = ld4(g.B + brow1 + (size_t)tap * g.K + cc * 32);
  };
  auto sstore = [&](auto sc, int buf) {
    constexpr int S = decltype(sc)::value;
#pragma unroll
    for (int j = 0; j < RJ; ++j) As[buf][(lrow + 32 * j) * 8 + (lc ^ lswz)] = ra[S][j];
#pragma unroll
    for (int j = 0; j < 2; ++j) Bs[buf][(lrow + 32 * j) * 8 + (lc ^ lswz)] = rb[S][j];
  };
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const int frow = wrow * 32 + ln, fswz = (ln >> 1) & 7;
  auto compute = [&](int buf) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int pos = (2 * gq + lh) ^ fswz;
      const float4 a = As[buf][frow * 8 + pos];
      const float av[4] = {a.x, a.y, a.z, a.w};
      float bv[NB][4];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float4 b = Bs[buf][((wcol + nb) * 32 + ln) * 8 + pos];
        bv[nb][0] = b.x; bv[nb][1] = b.y; bv[nb][2] = b.z; bv[nb][3] = b.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[nb][s], acc[nb], 0, 0, 0);
    }
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  gload(S0, 0);
  sstore(S0, 0);
  __syncthreads();
  if (nk > 1) gload(S1, 1);
  int kc = 0;
  // steady state: loads issued on every path, so the wait in front of a store is for the older register set only (behind a
  // conditional load the compiler waits for everything outstanding)
  for (; kc + 3 < nk; kc += 2) {
    gload(S0, kc + 2);
    compute(0);
    sstore(S1, 1);
    __syncthreads();
    gload(S1, kc + 3);
    compute(1);
    sstore(S0, 0);
    __syncthreads();
  }
  for (; kc < nk; kc += 2) {
    // LDS buffer 0 holds chunk kc, register set 1 chunk kc + 1
    if (kc + 2 < nk) gload(S0, kc + 2);
    compute(0);
    if (kc + 1 < nk) sstore(S1, 1);
    __syncthreads();
    if (kc + 1 >= nk) break;
    // LDS buffer 1 holds chunk kc + 1, register set 0 chunk kc + 2
    if (kc + 3 < nk) gload(S1, kc + 3);
    compute(1);
    if (kc + 2 < nk) sstore(S0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + wrow * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (q < Q) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const size_t o = (size_t)q * g.Ncols + n0 + (wcol + nb) * 32 + ln;
        g.C[o] = g.addend ? acc[nb][r] + g.addend[o] : acc[nb][r];
      }
    }
  }
}

struct GWArgs {
  const float* x;        // (N, Hi, Wi, Ci)
  const float* dy;       // (N, Ho, Wo, Co)
  float* partial;        // (splits, Co, 9, Ci)
  int N, Hi, Wi, Ho, Wo, Ci, Co, stride;
  int tiles_m, tiles_n, q_per_split;
};

// one workgroup = (pixel range, 64 x 64 tile of (co, ci), tap): see k_sc_wgrad (t2o_conv1x1.hip)
__global__ __launch_bounds__(kThreads) void k_gconv_wgrad(GWArgs g) {
  __shared__ float4 Ds[2][32 * 16];
  __shared__ float4 Xs[2][32 * 16];
  const int tiles = g.tiles_m * g.tiles_n;
  const int tap = blockIdx.x % 9, unit = blockIdx.x / 9;
  const int split = unit / tiles, tile = unit % tiles;
  const int kh = tap / 3, kw = tap - 3 * kh;
  const int m0 = (tile / g.tiles_n) * 64, n0 = (tile % g.tiles_n) * 64;
  const int Q = g.N * g.Ho * g.Wo;
  const int qa = split * g.q_per_split, qb = min(qa + g.q_per_split, Q);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 4, lc = tid & 15;
  float4 rd[2], rx[2];
  auto gload = [&](int q0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = q0 + lrow + 16 * j;
      rd[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      rx[j] = rd[j];
      if (q < qb) {
        const int hw = g.Ho * g.Wo;
        const int n = q / hw, r = q - n * hw;
        const int oh = r / g.Wo, ow = r - oh * g.Wo;
        const int ih = oh * g.stride + kh - 1, iw = ow * g.stride + kw - 1;
        if (ih >= 0 && iw >= 0 && ih < g.Hi && iw < g.Wi) {
          rd[j] = ld4(g.dy + (size_t)q * g.Co + m0 + lc * 4);
          rx[j] = ld4(g.x + (((size_t)n * g.Hi + ih) * g.Wi + iw) * g.Ci + n0 + lc * 4);
        }
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      Ds[buf][(lrow + 16 * j) * 16 + lc] = rd[j];
      Xs[buf][(lrow + 16 * j) * 16 + lc] = rx[j];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int stages = (qb - qa + 31) / 32;
  if (stages > 0) {
    gload(qa);
    sstore(0);
  }
  __syncthreads();
  for (int st = 0; st < stages; ++st) {
    const int buf = st & 1;
    if (st + 1 < stages) gload(qa + (st + 1) * 32);
    const float* dsm = reinterpret_cast<const float*>(&Ds[buf][0]) + wm * 32 + ln;
    const float* xsm = reinterpret_cast<const float*>(&Xs[buf][0]) + wn * 32 + ln;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dsm[(2 * kk + lh) * 64], xsm[(2 * kk + lh) * 64], acc, 0, 0, 0);
    if (st + 1 < stages) sstore(buf ^ 1);
    __syncthreads();
  }
  float* out = g.partial + (((size_t)split * g.Co + m0 + wm * 32) * 9 + tap) * g.Ci + n0 + wn * 32 + ln;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * 9 * g.Ci] = acc[r];
}

bool shape_ok(int N, int H, int W, int Ci, int Co) {
  return N > 0 && H > 0 && W > 0 && Ci >= 32 && Ci % 32 == 0 && Co >= 64 && Co % 64 == 0 && (size_t)N * H * W < ((size_t)1 << 30);
}
bool misaligned16(const void* a, const void* b, const void* c, const void* d = nullptr) {
  return ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(c) | reinterpret_cast<size_t>(d)) & 15) != 0;
}

int launch(const float* A, const float* B, const float* addend, float* C, int N, int Ha, int Wa, int Hc, int Wc, int K, int Ncols,
           int mode, hipStream_t st) {
  GArgs g;
  g.A = A; g.B = B; g.addend = addend; g.C = C; g.N = N; g.Ha = Ha; g.Wa = Wa; g.Hc = Hc; g.Wc = Wc; g.K = K; g.Ncols = Ncols; g.mode = mode;
  g.tiles_n = Ncols / 64;
  const int Q = N * Hc * Wc;
  const bool small = (long long)((Q + 127) / 128) * g.tiles_n < 256;       // 128-pixel tiles would leave CUs idle: 64-pixel ones
  g.tiles_m = small ? (Q + 63) / 64 : (Q + 127) / 128;
  const unsigned grid = (unsigned)(((g.tiles_m + 7) / 8) * 8 * g.tiles_n);
#define T2O_GCONV(M)                                                        \
  do {                                                                      \
    if (small) k_gconv<M, 64><<<grid, kThreads, 0, st>>>(g);                \
    else k_gconv<M, 128><<<grid, kThreads, 0, st>>>(g);                     \
  } while (0)
  if (g.mode == 2) T2O_GCONV(2);
  else if (g.mode == 1) T2O_GCONV(1);
  else T2O_GCONV(0);
#undef T2O_GCONV
  return hipGetLastError() == hipSuccess ? T2O_OK : T2O_ELAUNCH;
}

struct WPlan { int tiles_m, tiles_n, splits, q_per_split; };
WPlan wplan(int Q, int Ci, int Co) {
  WPlan p;
  p.tiles_m = Co / 64; p.tiles_n = Ci / 64;
  int splits = 512 / (9 * p.tiles_m * p.tiles_n);
  if (splits > (Q + 63) / 64) splits = (Q + 63) / 64;
  if (splits < 1) splits = 1;
  p.q_per_split = (((Q + splits - 1) / splits) + 31) / 32 * 32;
  p.splits = (Q + p.q_per_split - 1) / p.q_per_split;
  return p;
}

}  // namespace

extern "C" {

int t2o_conv3x3_any_fwd_nhwc(const float* x, const float* w, float* y, int N, int H, int W, int Ci, int Co, int stride, void* stream) {
  if (!x || !w || !y || misaligned16(x, w, y)) return set_error(T2O_EINVAL, "conv3x3_any_fwd: null or not 16-byte aligned pointer");
  if (!shape_ok(N, H, W, Ci, Co) || (stride != 1 && stride != 2))
    return set_error(T2O_EUNSUPPORTED, "conv3x3_any_fwd: Ci must be a multiple of 32, Co of 64, stride 1 or 2");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int rc = launch(x, w, nullptr, y, N, H, W, Ho, Wo, Ci, Co, stride == 2 ? 1 : 0, (hipStream_t)stream);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_any_fwd launch failed");
}

int t2o_conv3x3_any_dgrad_nhwc(const float* dy, const float* wt, const float* addend, float* dx, int N, int H, int W, int Ci, int Co,
                               int stride, void* stream) {
  if (!dy || !wt || !dx || misaligned16(dy, wt, dx, addend)) return set_error(T2O_EINVAL, "conv3x3_any_dgrad: null or not 16-byte aligned pointer");
  if (!shape_ok(N, H, W, Co, Ci) || (stride != 1 && stride != 2))
    return set_error(T2O_EUNSUPPORTED, "conv3x3_any_dgrad: Co must be a multiple of 32, Ci of 64, stride 1 or 2");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int rc = launch(dy, wt, addend, dx, N, Ho, Wo, H, W, Co, Ci, stride == 2 ? 2 : 0, (hipStream_t)stream);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_any_dgrad launch failed");
}

size_t t2o_conv3x3_any_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co, int stride) {
  if (!shape_ok(N, H, W, Ci, Co) || Ci % 64 != 0 || (stride != 1 && stride != 2)) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const WPlan p = wplan(N * Ho * Wo, Ci, Co);
  return sizeof(float) * (size_t)p.splits * Co * 9 * Ci;
}

int t2o_conv3x3_any_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                               int N, int H, int W, int Ci, int Co, int stride, int accumulate, void* stream) {
  if (!x || !dy || !dw || misaligned16(x, dy, dw)) return set_error(T2O_EINVAL, "conv3x3_any_wgrad: null or not 16-byte aligned pointer");
  const size_t need = t2o_conv3x3_any_wgrad_workspace_bytes(N, H, W, Ci, Co, stride);
  if (need == 0) return set_error(T2O_EUNSUPPORTED, "conv3x3_any_wgrad: channel counts must be multiples of 64, stride 1 or 2");
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "conv3x3_any_wgrad: workspace too small");
  GWArgs g;
  g.x = x; g.dy = dy; g.partial = (float*)workspace;
  g.N = N; g.Hi = H; g.Wi = W; g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1; g.Ci = Ci; g.Co = Co; g.stride = stride;
  const WPlan p = wplan(N * g.Ho * g.Wo, Ci, Co);
  g.tiles_m = p.tiles_m; g.tiles_n = p.tiles_n; g.q_per_split = p.q_per_split;
  hipStream_t st = (hipStream_t)stream;
  k_gconv_wgrad<<<(unsigned)(p.splits * p.tiles_m * p.tiles_n * 9), kThreads, 0, st>>>(g);
  t2o::launch_wgrad_reduce(g.partial, dw, (size_t)Co * 9 * Ci, p.splits, accumulate, st);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv3x3_any_wgrad launch failed");
}

}  // extern "C"
