// t2o_conv1x1.hip -- the 1x1 stride-2 shortcut convolutions of the image encoder's stages
// (models/actor_resnet.py:33-36: nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride, bias=False) in front of a
// batch norm), forward, data gradient and weight gradient on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// With NHWC activations these are plain GEMMs over GATHERED rows: output pixel q = (n, a, b) of the (N, Ho, Wo) grid
// reads input pixel (n, 2a, 2b), a row of Ci contiguous floats.
//   forward        Y[q][co]        = sum_ci X[row(q)][ci] W[co][ci]         M = pixels, N = Co, K = Ci   (A rows gathered)
//   data gradient  dX[row(q)][ci] += sum_co dY[q][co] Wt[ci][co]            M = pixels, N = Ci, K = Co   (C rows scattered, +=)
//   weight gradient dW[co][ci]     = sum_q dY[q][co] X[row(q)][ci]          M = Co, N = Ci, K = pixels   (split-K, fixed-order sum)
// They are small (1-2 GFLOP per layer at bs=64 256x256) and bound by the memory system (the 64-channel stage reads
// 67 MB and writes 67 MB for 2 GFLOP), so the kernels are plain register-staged, double-buffered LDS tilings -- the
// LDS-DMA / scalar-only machinery of the 3x3 kernels (t2o_conv.hip) would buy nothing here.  What they replace are
// three library calls per layer whose weight gradient adds atomically into a buffer a memset node has to clear first:
// the last non-deterministic gradients and the last memset nodes of the encoder's graphs.
#include <hip/hip_runtime.h>

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

struct RowMap { int Ho, Wo, H, W; };
// pixel index of (n, 2a, 2b) in the (N, H, W) grid for q = (n, a, b) of the (N, Ho, Wo) grid
__device__ __forceinline__ size_t strided_row(int q, const RowMap& m) {
  const int hw = m.Ho * m.Wo;
  const int n = q / hw, r = q - n * hw;
  const int a = r / m.Wo, b = r - a * m.Wo;
  return ((size_t)n * m.H + 2 * a) * m.W + 2 * b;
}

struct GemmArgs {
  const float* A;      // rows of K floats (gather: row q at strided_row(q))
  const float* B;      // (Ncols, K)
  float* C;            // rows of Ncols floats (scatter: row q at strided_row(q), added to)
  int Q, K, Ncols;
  RowMap map;
  int gather, scatter;
  int tiles_m, tiles_n;
};

// C[q][n] (+)= sum_k A[q][k] B[n][k].  Workgroup = 4 waves = 128 rows x 64 columns; wave = 32 rows x 64 columns (two
// MFMA blocks).  K in chunks of 32 floats: rows of 128 bytes stored as 8 chunks of 16 bytes with chunk c of row r at
// position c ^ ((r >> 1) & 7), so that lane l of a fragment reads chunk 2g + l / 32 of row l % 32 with one conflict-free
// ds_read_b128 = four k-steps of that lane (the same scheme as k_conv3x3_fwd; both operands use the same k order).
// The next chunk travels global -> registers during a chunk's MFMAs.
__global__ __launch_bounds__(kThreads) void k_sc_gemm(GemmArgs g) {
  __shared__ float4 As[2][128 * 8];
  __shared__ float4 Bs[2][64 * 8];
  const int b = blockIdx.x;
  const int xcd = b % 8, k8 = b / 8;                       // the column tiles of one row tile share an XCD (its L2 holds the rows)
  const int rt = (k8 / g.tiles_n) * 8 + xcd, ct = k8 % g.tiles_n;
  if (rt >= g.tiles_m) return;
  const int q0 = rt * 128, n0 = ct * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, lh = lane >> 5;

  const int lrow = tid >> 3, lc = tid & 7;                 // loader: rows lrow + 32 j, 16-byte chunk lc
  const int lswz = (lrow >> 1) & 7;
  size_t arow[4], brow[2];
  bool aok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = q0 + lrow + 32 * j;
    aok[j] = q < g.Q;
    arow[j] = aok[j] ? (g.gather ? strided_row(q, g.map) : (size_t)q) * g.K + lc * 4 : 0;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) brow[j] = (size_t)(n0 + lrow + 32 * j) * g.K + lc * 4;
  float4 ra[4], rb[2];
  auto gload = [&](int kc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = aok[j] ? ld4(g.A + arow[j] + kc * 32) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < 2; ++j) rb[j] = ld4(g.B + brow[j] + kc * 32);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) As[buf][(lrow + 32 * j) * 8 + (lc ^ lswz)] = ra[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) Bs[buf][(lrow + 32 * j) * 8 + (lc ^ lswz)] = rb[j];
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const int frow = wave * 32 + ln, fswz = (ln >> 1) & 7;   // (wave * 32 and 32 j do not change the swizzle)

  const int nk = g.K / 32;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) gload(kc + 1);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int pos = (2 * gq + lh) ^ fswz;
      const float4 a = As[buf][frow * 8 + pos];
      const float4 b0 = Bs[buf][ln * 8 + pos], b1 = Bs[buf][(32 + ln) * 8 + pos];
      const float av[4] = {a.x, a.y, a.z, a.w}, bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv0[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv1[s], acc[1], 0, 0, 0);
      }
    }
    if (kc + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // C/D layout: column = lane % 32, row = (reg % 4) + 8 * (reg / 4) + 4 * (lane / 32)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (q < g.Q) {
      float* dst = g.C + (g.scatter ? strided_row(q, g.map) : (size_t)q) * g.Ncols + n0 + ln;
      if (g.scatter) { dst[0] += acc[0][r]; dst[32] += acc[1][r]; }
      else { dst[0] = acc[0][r]; dst[32] = acc[1][r]; }
    }
  }
}

struct WgArgs {
  const float* x;      // (N,H,W,Ci)
  const float* dy;     // (Q, Co)
  float* partial;      // (splits, Co, Ci)
  int Q, Ci, Co;
  RowMap map;
  int tiles_m, tiles_n, q_per_split;
};

// One workgroup = one 64 (co) x 64 (ci) tile of dW over a range of pixels; 2 x 2 waves, one MFMA block each.  NHWC rows
// ARE the MFMA operand order (for one pixel k the 32 lanes read 32 consecutive channels): tiles are [pixel][64 channels]
// in LDS, fragments conflict-free ds_read_b32.  32 pixels per stage, double-buffered through registers.
__global__ __launch_bounds__(kThreads) void k_sc_wgrad(WgArgs g) {
  __shared__ float4 Ds[2][32 * 16];
  __shared__ float4 Xs[2][32 * 16];
  const int tiles = g.tiles_m * g.tiles_n;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int m0 = (tile / g.tiles_n) * 64, n0 = (tile % g.tiles_n) * 64;
  const int qa = split * g.q_per_split, qb = min(qa + g.q_per_split, g.Q);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 4, lc = tid & 15;                // loader: pixel rows lrow, lrow + 16; channel quad lc
  float4 rd[2], rx[2];
  auto gload = [&](int q0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = q0 + lrow + 16 * j;
      const bool ok = q < qb;
      rd[j] = ok ? ld4(g.dy + (size_t)q * g.Co + m0 + lc * 4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      rx[j] = ok ? ld4(g.x + strided_row(q, g.map) * g.Ci + n0 + lc * 4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
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
  float* out = g.partial + ((size_t)split * g.Co + m0 + wm * 32) * g.Ci + n0 + wn * 32 + ln;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * g.Ci] = acc[r];
}

bool misaligned16(const void* a, const void* b, const void* c) {
  return ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(c)) & 15) != 0;
}

bool sc_supported(int N, int H, int W, int Ci, int Co) {
  return N > 0 && H > 0 && W > 0 && Ci >= 64 && Co >= 64 && Ci % 64 == 0 && Co % 64 == 0 && (size_t)N * H * W < ((size_t)1 << 30);
}

RowMap row_map(int H, int W) {
  RowMap m;
  m.H = H; m.W = W; m.Ho = (H + 1) / 2; m.Wo = (W + 1) / 2;
  return m;
}

int launch_gemm(const float* A, const float* B, float* C, int Q, int K, int Ncols, const RowMap& map, int gather, int scatter,
                hipStream_t st) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.Q = Q; g.K = K; g.Ncols = Ncols; g.map = map; g.gather = gather; g.scatter = scatter;
  g.tiles_m = (Q + 127) / 128; g.tiles_n = Ncols / 64;
  const unsigned grid = (unsigned)(((g.tiles_m + 7) / 8) * 8 * g.tiles_n);
  k_sc_gemm<<<grid, kThreads, 0, st>>>(g);
  return hipGetLastError() == hipSuccess ? T2O_OK : T2O_ELAUNCH;
}

struct WgPlan { int tiles_m, tiles_n, splits, q_per_split; };
WgPlan wg_plan(int Q, int Ci, int Co) {
  WgPlan p;
  p.tiles_m = Co / 64; p.tiles_n = Ci / 64;
  int splits = 512 / (p.tiles_m * p.tiles_n);              // ~2 workgroups per CU
  if (splits > (Q + 63) / 64) splits = (Q + 63) / 64;      // at least two stages of K each
  if (splits < 1) splits = 1;
  p.q_per_split = (((Q + splits - 1) / splits) + 31) / 32 * 32;
  p.splits = (Q + p.q_per_split - 1) / p.q_per_split;
  return p;
}

}  // namespace

extern "C" {

int t2o_conv1x1s2_fwd_nhwc(const float* x, const float* w, float* y, int N, int H, int W, int Ci, int Co, void* stream) {
  if (!x || !w || !y || misaligned16(x, w, y)) return set_error(T2O_EINVAL, "conv1x1s2_fwd: null or not 16-byte aligned pointer");
  if (!sc_supported(N, H, W, Ci, Co)) return set_error(T2O_EUNSUPPORTED, "conv1x1s2_fwd: channel counts must be multiples of 64");
  const RowMap m = row_map(H, W);
  const int rc = launch_gemm(x, w, y, N * m.Ho * m.Wo, Ci, Co, m, 1, 0, (hipStream_t)stream);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv1x1s2_fwd launch failed");
}

int t2o_conv1x1s2_dgrad_acc_nhwc(const float* dy, const float* wt, float* dx, int N, int H, int W, int Ci, int Co, void* stream) {
  if (!dy || !wt || !dx || misaligned16(dy, wt, dx)) return set_error(T2O_EINVAL, "conv1x1s2_dgrad_acc: null or not 16-byte aligned pointer");
  if (!sc_supported(N, H, W, Ci, Co)) return set_error(T2O_EUNSUPPORTED, "conv1x1s2_dgrad_acc: channel counts must be multiples of 64");
  const RowMap m = row_map(H, W);
  const int rc = launch_gemm(dy, wt, dx, N * m.Ho * m.Wo, Co, Ci, m, 0, 1, (hipStream_t)stream);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv1x1s2_dgrad_acc launch failed");
}

size_t t2o_conv1x1s2_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
  if (!sc_supported(N, H, W, Ci, Co)) return 0;
  const RowMap m = row_map(H, W);
  const WgPlan p = wg_plan(N * m.Ho * m.Wo, Ci, Co);
  return sizeof(float) * (size_t)p.splits * Co * Ci;
}

int t2o_conv1x1s2_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                             int N, int H, int W, int Ci, int Co, int accumulate, void* stream) {
  if (!x || !dy || !dw || misaligned16(x, dy, dw)) return set_error(T2O_EINVAL, "conv1x1s2_wgrad: null or not 16-byte aligned pointer");
  const size_t need = t2o_conv1x1s2_wgrad_workspace_bytes(N, H, W, Ci, Co);
  if (need == 0) return set_error(T2O_EUNSUPPORTED, "conv1x1s2_wgrad: channel counts must be multiples of 64");
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "conv1x1s2_wgrad: workspace too small");
  const RowMap m = row_map(H, W);
  WgArgs g;
  g.x = x; g.dy = dy; g.partial = (float*)workspace;
  g.Q = N * m.Ho * m.Wo; g.Ci = Ci; g.Co = Co; g.map = m;
  const WgPlan p = wg_plan(g.Q, Ci, Co);
  g.tiles_m = p.tiles_m; g.tiles_n = p.tiles_n; g.q_per_split = p.q_per_split;
  hipStream_t st = (hipStream_t)stream;
  k_sc_wgrad<<<(unsigned)(p.splits * p.tiles_m * p.tiles_n), kThreads, 0, st>>>(g);
  t2o::launch_wgrad_reduce(g.partial, dw, (size_t)Co * Ci, p.splits, accumulate, st);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv1x1s2_wgrad launch failed");
}

}  // extern "C"
