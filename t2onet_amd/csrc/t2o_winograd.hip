// t2o_winograd.hip -- Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the image encoder's deep stages
// (models/actor_resnet.py:24-44 BasicBlock, 256- and 512-channel layers: 16x16 and 8x8 maps at bs = 64).
//
//   y = conv2d(x, w, None, 1, 1) on NHWC activations, H and W even.  A tile t = (n, th, tw) is the 2 x 2 output block
//   at (2 th, 2 tw); it reads the 4 x 4 input patch at rows 2 th - 1 .. 2 th + 2, columns 2 tw - 1 .. 2 tw + 2.
//       V[xi][t][ci]  = (B^T d B)[xi]           xi = 4 r + c, the 16 positions of the transformed patch
//       U[xi][co][ci] = (G g G^T)[xi]
//       M[xi][t][co]  = sum_ci V[xi][t][ci] * U[xi][co][ci]        16 GEMMs  (T x Ci) x (Ci x Co): t2o_conv.hip k_gemm_nt
//       y tile        = A^T M A
//   36 multiplies per output pixel and channel pair become 16: the direct kernel needs 19.3 GFLOP per layer and runs at
//   115-120 TFLOP/s (160 us); the 16 GEMMs are 8.6 GFLOP.  What the transforms move decides where it pays: V and M are
//   4 x the activation, so the two transform passes cost 10 x the activation in HBM traffic -- 84 MB at the 512-channel
//   stage (8 MB activations), 168 MB at the 256-channel stage, 1.3 GB at the 64-channel stage (where it loses).
//   The data gradient of these layers is the same convolution on dy with the mirrored, transposed filter.
//
//   The transform kernels: one thread per (tile, channel quad): 16 (input) / 16 (output) 16-byte accesses, consecutive
//   threads = consecutive channel quads, so every wave access is C * 4 contiguous bytes (1-2 KiB).  The output transform
//   carries the direct kernels' epilogue options: an addend (the gradient through the identity shortcut) and the
//   per-channel sum / sum of squares of y for the batch norm that follows (t2o_bn_relu_nhwc_fwd_partials), reduced in a
//   fixed order: thread-sequential over its tiles, then over the workgroup's tile slots, one partial row per workgroup.
#include <hip/hip_runtime.h>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 512;            // workgroups (= statistics partial rows) of the output transform

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// x (N,H,W,C) -> V (16, Tpad, C), T = N * H/2 * W/2 tiles, rows T .. Tpad - 1 zero (the weight gradient's GEMM runs over
// whole 64-row stages of the tile index)
// (PR = rows per plane of V: Tpad for a tensor of its own, more when V is a row range of a larger (16, PR, C) arena)
__global__ __launch_bounds__(kThreads) void k_wino_input(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W, int C, int Tpad, int PR) {
  const int q = C >> 2, TH = H >> 1, TW = W >> 1;
  const size_t T = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= (size_t)Tpad * q) return;
  const size_t t = idx / q;
  const int cq = (int)(idx - t * q);
  if (t >= T) {
    float* o = V + t * C + 4 * cq;
#pragma unroll
    for (int k = 0; k < 16; ++k) st4(o + (size_t)k * PR * C, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    return;
  }
  const int n = (int)(t / (TH * TW)), rem = (int)(t - (size_t)n * TH * TW), th = rem / TW, tw = rem - th * TW;
  const int h0 = 2 * th - 1, w0 = 2 * tw - 1;
  float4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int h = h0 + i, w = w0 + j;
      d[i][j] = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ld4(x + (((size_t)n * H + h) * W + w) * C + 4 * cq)
                                                                         : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3), then the same along the columns
  float4 r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = sub4(d[0][j], d[2][j]);
    r[1][j] = add4(d[1][j], d[2][j]);
    r[2][j] = sub4(d[2][j], d[1][j]);
    r[3][j] = sub4(d[1][j], d[3][j]);
  }
  const size_t plane = (size_t)PR * C;
  float* o = V + t * C + 4 * cq;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st4(o + (size_t)(4 * i + 0) * plane, sub4(r[i][0], r[i][2]));
    st4(o + (size_t)(4 * i + 1) * plane, add4(r[i][1], r[i][2]));
    st4(o + (size_t)(4 * i + 2) * plane, sub4(r[i][2], r[i][1]));
    st4(o + (size_t)(4 * i + 3) * plane, sub4(r[i][1], r[i][3]));
  }
}

// M (16, T, C) -> y (N,H,W,C) (+ addend), optional statistics rows (gridDim.x, 2, C)
template <bool kStats>
__global__ __launch_bounds__(kThreads) void k_wino_output(const float* __restrict__ Mx, const float* __restrict__ addend, float* __restrict__ y,
                                                          float* __restrict__ stats, int N, int H, int W, int C) {
  const int q = C >> 2, R = kThreads / q, TH = H >> 1, TW = W >> 1;
  const int cq = threadIdx.x % q, slot = threadIdx.x / q;
  const size_t T = (size_t)N * TH * TW, plane = T * C;
  float4 s1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s2 = s1;
  for (size_t t = (size_t)blockIdx.x * R + slot; t < T; t += (size_t)gridDim.x * R) {
    const float* mp = Mx + t * C + 4 * cq;
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = ld4(mp + (size_t)(4 * i + j) * plane);
    // A^T m: rows (m0 + m1 + m2, m1 - m2 - m3), then along the columns
    float4 r[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[0][j] = add4(add4(m[0][j], m[1][j]), m[2][j]);
      r[1][j] = sub4(sub4(m[1][j], m[2][j]), m[3][j]);
    }
    const int n = (int)(t / (TH * TW)), rem = (int)(t - (size_t)n * TH * TW), th = rem / TW, tw = rem - th * TW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 o0 = add4(add4(r[i][0], r[i][1]), r[i][2]);
      float4 o1 = sub4(sub4(r[i][1], r[i][2]), r[i][3]);
      const size_t off = (((size_t)n * H + 2 * th + i) * W + 2 * tw) * C + 4 * cq;
      if (addend) { o0 = add4(o0, ld4(addend + off)); o1 = add4(o1, ld4(addend + off + C)); }
      st4(y + off, o0);
      st4(y + off + C, o1);
      if (kStats) {
        s1 = add4(s1, add4(o0, o1));
        s2.x += o0.x * o0.x + o1.x * o1.x; s2.y += o0.y * o0.y + o1.y * o1.y;
        s2.z += o0.z * o0.z + o1.z * o1.z; s2.w += o0.w * o0.w + o1.w * o1.w;
      }
    }
  }
  if (kStats) {
    __shared__ float4 sm[2][kThreads];
    sm[0][threadIdx.x] = s1;
    sm[1][threadIdx.x] = s2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (not __syncthreads(): no need to wait for the y stores)
    __builtin_amdgcn_s_barrier();
    if ((int)threadIdx.x < q) {
      float4 a = sm[0][threadIdx.x], b = sm[1][threadIdx.x];
      for (int k = threadIdx.x + q; k < kThreads; k += q) { a = add4(a, sm[0][k]); b = add4(b, sm[1][k]); }      // fixed order
      float* dst = stats + (size_t)blockIdx.x * 2 * C;
      st4(dst + 4 * threadIdx.x, a);
      st4(dst + C + 4 * threadIdx.x, b);
    }
  }
}

// w (Cn, 3, 3, Ck) -> U (16, Cn, Ck) = G g G^T per (n, k) pair; G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(kThreads) void k_wino_weight(const float* __restrict__ w, float* __restrict__ U, int Cn, int Ck) {
  const size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x, total = (size_t)Cn * Ck;
  if (idx >= total) return;
  const int n = (int)(idx / Ck), k = (int)(idx - (size_t)n * Ck);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)n * 9 + a * 3 + b) * Ck + k];
  float r[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    r[0][b] = g[0][b];
    r[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
    r[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
    r[3][b] = g[2][b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    U[(size_t)(4 * a + 0) * total + idx] = r[a][0];
    U[(size_t)(4 * a + 1) * total + idx] = 0.5f * ((r[a][0] + r[a][2]) + r[a][1]);
    U[(size_t)(4 * a + 2) * total + idx] = 0.5f * ((r[a][0] + r[a][2]) - r[a][1]);
    U[(size_t)(4 * a + 3) * total + idx] = r[a][2];
  }
}

// WEIGHT GRADIENT in the transformed domain: dg = G^T [ sum_t (A dY A^T)[t] (x) (B^T d B)[t] ] G per (co, ci):
//   Ad (16, T, Co) = A dY A^T of the 2 x 2 output-gradient tiles (A = [[1,0],[1,1],[1,-1],[0,-1]])
//   dU[xi] (Co, Ci) = Ad[xi]^T * V[xi]       16 GEMMs with K = T, V = the forward's transformed input (kept, or redone)
//   dw (Co,3,3,Ci) (+)= G^T dU G
__global__ __launch_bounds__(kThreads) void k_wino_dy(const float* __restrict__ dy, float* __restrict__ Ad, int N, int H, int W, int C, int Tpad, int PR) {
  const int q = C >> 2, TH = H >> 1, TW = W >> 1;
  const size_t T = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= (size_t)Tpad * q) return;
  const size_t t = idx / q;
  const int cq = (int)(idx - t * q);
  if (t >= T) {                                            // zero rows up to Tpad (as k_wino_input)
    float* o = Ad + t * C + 4 * cq;
#pragma unroll
    for (int k = 0; k < 16; ++k) st4(o + (size_t)k * PR * C, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    return;
  }
  const int n = (int)(t / (TH * TW)), rem = (int)(t - (size_t)n * TH * TW), th = rem / TW, tw = rem - th * TW;
  const float* p = dy + (((size_t)n * H + 2 * th) * W + 2 * tw) * C + 4 * cq;
  const float4 d00 = ld4(p), d01 = ld4(p + C), d10 = ld4(p + (size_t)W * C), d11 = ld4(p + (size_t)W * C + C);
  // A d: rows (d0, d0 + d1, d0 - d1, -d1)
  const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const float4 r[4][2] = {{d00, d01}, {add4(d00, d10), add4(d01, d11)}, {sub4(d00, d10), sub4(d01, d11)}, {sub4(zero, d10), sub4(zero, d11)}};
  const size_t plane = (size_t)PR * C;
  float* o = Ad + t * C + 4 * cq;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st4(o + (size_t)(4 * i + 0) * plane, r[i][0]);
    st4(o + (size_t)(4 * i + 1) * plane, add4(r[i][0], r[i][1]));
    st4(o + (size_t)(4 * i + 2) * plane, sub4(r[i][0], r[i][1]));
    st4(o + (size_t)(4 * i + 3) * plane, sub4(zero, r[i][1]));
  }
}

// Both transforms of an output gradient in one pass over dy: V = B^T d B of the 4 x 4 patches (the data gradient's GEMM
// operand) and Ad = A dY A^T of their inner 2 x 2 blocks (the weight gradient's); rows T .. Tpad - 1 of both are zero.
__global__ __launch_bounds__(kThreads) void k_wino_dy_both(const float* __restrict__ dy, float* __restrict__ V, float* __restrict__ Ad,
                                                           int N, int H, int W, int C, int Tpad, int PRa) {
  const int q = C >> 2, TH = H >> 1, TW = W >> 1;
  const size_t T = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= (size_t)Tpad * q) return;
  const size_t t = idx / q;
  const int cq = (int)(idx - t * q);
  const size_t plane = (size_t)Tpad * C, plane_a = (size_t)PRa * C;
  float* ov = V + t * C + 4 * cq;
  float* oa = Ad + t * C + 4 * cq;
  const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (t >= T) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { st4(ov + (size_t)k * plane, zero); st4(oa + (size_t)k * plane_a, zero); }
    return;
  }
  const int n = (int)(t / (TH * TW)), rem = (int)(t - (size_t)n * TH * TW), th = rem / TW, tw = rem - th * TW;
  const int h0 = 2 * th - 1, w0 = 2 * tw - 1;
  float4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int h = h0 + i, w = w0 + j;
      d[i][j] = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ld4(dy + (((size_t)n * H + h) * W + w) * C + 4 * cq) : zero;
    }
  {
    const float4 r[4][2] = {{d[1][1], d[1][2]}, {add4(d[1][1], d[2][1]), add4(d[1][2], d[2][2])},
                            {sub4(d[1][1], d[2][1]), sub4(d[1][2], d[2][2])}, {sub4(zero, d[2][1]), sub4(zero, d[2][2])}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st4(oa + (size_t)(4 * i + 0) * plane_a, r[i][0]);
      st4(oa + (size_t)(4 * i + 1) * plane_a, add4(r[i][0], r[i][1]));
      st4(oa + (size_t)(4 * i + 2) * plane_a, sub4(r[i][0], r[i][1]));
      st4(oa + (size_t)(4 * i + 3) * plane_a, sub4(zero, r[i][1]));
    }
  }
  float4 r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = sub4(d[0][j], d[2][j]);
    r[1][j] = add4(d[1][j], d[2][j]);
    r[2][j] = sub4(d[2][j], d[1][j]);
    r[3][j] = sub4(d[1][j], d[3][j]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st4(ov + (size_t)(4 * i + 0) * plane, sub4(r[i][0], r[i][2]));
    st4(ov + (size_t)(4 * i + 1) * plane, add4(r[i][1], r[i][2]));
    st4(ov + (size_t)(4 * i + 2) * plane, sub4(r[i][2], r[i][1]));
    st4(ov + (size_t)(4 * i + 3) * plane, sub4(r[i][1], r[i][3]));
  }
}

// dU (16, Co, Ci) -> dw (Co,3,3,Ci) = G^T dU G (acc: added to dw)
__global__ __launch_bounds__(kThreads) void k_wino_dw(const float* __restrict__ dU, float* __restrict__ dw, int Co, int Ci, int acc, int splits) {
  const size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x, total = (size_t)Co * Ci;
  if (idx >= total) return;
  const int co = (int)(idx / Ci), ci = (int)(idx - (size_t)co * Ci);
  float u[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v = dU[(size_t)(4 * a + b) * total + idx];
      for (int sp = 1; sp < splits; ++sp) v += dU[((size_t)sp * 16 + 4 * a + b) * total + idx];       // the split-K pieces, in order
      u[a][b] = v;
    }
  // G^T u: rows (u0 + (u1 + u2)/2, (u1 - u2)/2, (u1 + u2)/2 + u3)
  float r[3][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const float h = 0.5f * (u[1][b] + u[2][b]);
    r[0][b] = u[0][b] + h;
    r[1][b] = 0.5f * (u[1][b] - u[2][b]);
    r[2][b] = h + u[3][b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float h = 0.5f * (r[a][1] + r[a][2]);
    const float g0 = r[a][0] + h, g1 = 0.5f * (r[a][1] - r[a][2]), g2 = h + r[a][3];
    float* o = dw + ((size_t)co * 9 + a * 3) * Ci + ci;
    if (acc) { o[0] += g0; o[Ci] += g1; o[2 * (size_t)Ci] += g2; }
    else { o[0] = g0; o[Ci] = g1; o[2 * (size_t)Ci] = g2; }
  }
}

// the filter transforms of several layers in ONE launch (once per optimiser step: 6 forward + 6 data-gradient banks)
constexpr int kMaxUJobs = 32;
struct UJobs {
  const float* w[kMaxUJobs];
  float* U[kMaxUJobs];
  int Cn[kMaxUJobs], Ck[kMaxUJobs];
  unsigned first[kMaxUJobs + 1];
  int n;
  int chunked;      // U in the on-chip kernel's chunk-major layout (Ck/8, 16, Cn, 8) instead of (16, Cn, Ck)
};

__global__ __launch_bounds__(kThreads) void k_wino_weight_batch(UJobs jb) {
  int j = 0;
  while (j + 1 < jb.n && blockIdx.x >= jb.first[j + 1]) ++j;
  const size_t idx = (size_t)(blockIdx.x - jb.first[j]) * kThreads + threadIdx.x, total = (size_t)jb.Cn[j] * jb.Ck[j];
  if (idx >= total) return;
  const int Ck = jb.Ck[j];
  const int n = (int)(idx / Ck), k = (int)(idx - (size_t)n * Ck);
  const float* w = jb.w[j];
  float* U = jb.U[j];
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)n * 9 + a * 3 + b) * Ck + k];
  float r[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    r[0][b] = g[0][b];
    r[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
    r[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
    r[3][b] = g[2][b];
  }
  // element (xi, n, k): plane-major xi * Cn * Ck + n * Ck + k, or chunk-major ((k / 8 * 16 + xi) * Cn + n) * 8 + k % 8
  const size_t step = jb.chunked ? (size_t)jb.Cn[j] * 8 : total;
  const size_t base = jb.chunked ? ((size_t)(k >> 3) * 16 * jb.Cn[j] + n) * 8 + (k & 7) : idx;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    U[(size_t)(4 * a + 0) * step + base] = r[a][0];
    U[(size_t)(4 * a + 1) * step + base] = 0.5f * ((r[a][0] + r[a][2]) + r[a][1]);
    U[(size_t)(4 * a + 2) * step + base] = 0.5f * ((r[a][0] + r[a][2]) - r[a][1]);
    U[(size_t)(4 * a + 3) * step + base] = r[a][2];
  }
}

bool wino_shape_ok(int N, int H, int W, int C) {
  return N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C >= 4 && C <= 1024 && (C & (C - 1)) == 0 &&
         (size_t)N * H * W * C < ((size_t)1 << 40);
}

int output_blocks(int N, int H, int W, int C) {
  const int R = kThreads / (C >> 2);
  const size_t T = (size_t)N * (H / 2) * (W / 2);
  size_t blocks = (T + R - 1) / R;
  if (blocks > (size_t)kMaxBlocks) blocks = kMaxBlocks;
  return (int)blocks;
}

}  // namespace

extern "C" {

int t2o_wino_weight_transform(const float* w, float* U, int Cn, int Ck, void* stream) {
  if (!w || !U || Cn <= 0 || Ck <= 0) return set_error(T2O_EINVAL, "wino_weight_transform: null pointer or bad shape");
  const size_t total = (size_t)Cn * Ck;
  k_wino_weight<<<(unsigned)((total + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(w, U, Cn, Ck);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_weight_transform: launch failed");
}

static int wino_weight_batch(const float* const* w, float* const* U, const int* Cn, const int* Ck, int n, int chunked, void* stream);

int t2o_wino_weight_transform_batch(const float* const* w, float* const* U, const int* Cn, const int* Ck, int n, void* stream) {
  return wino_weight_batch(w, U, Cn, Ck, n, 0, stream);
}

int t2o_wino_weight_transform_chunked_batch(const float* const* w, float* const* Uc, const int* Cn, const int* Ck, int n, void* stream) {
  for (int j = 0; Ck && j < n && j < kMaxUJobs; ++j)
    if (Ck[j] % 8 != 0) return set_error(T2O_EINVAL, "wino_weight_transform_chunked_batch: Ck must be a multiple of 8");
  return wino_weight_batch(w, Uc, Cn, Ck, n, 1, stream);
}

static int wino_weight_batch(const float* const* w, float* const* U, const int* Cn, const int* Ck, int n, int chunked, void* stream) {
  if (!w || !U || !Cn || !Ck || n < 1 || n > kMaxUJobs) return set_error(T2O_EINVAL, "wino_weight_transform_batch: null pointer or more than 32 jobs");
  UJobs jb = {};
  jb.n = n;
  jb.chunked = chunked;
  unsigned total = 0;
  for (int j = 0; j < n; ++j) {
    if (!w[j] || !U[j] || Cn[j] <= 0 || Ck[j] <= 0) return set_error(T2O_EINVAL, "wino_weight_transform_batch: null pointer or bad shape");
    jb.w[j] = w[j]; jb.U[j] = U[j]; jb.Cn[j] = Cn[j]; jb.Ck[j] = Ck[j];
    jb.first[j] = total;
    total += (unsigned)(((size_t)Cn[j] * Ck[j] + kThreads - 1) / kThreads);
  }
  jb.first[n] = total;
  k_wino_weight_batch<<<total, kThreads, 0, (hipStream_t)stream>>>(jb);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_weight_transform_batch: launch failed");
}

int t2o_wino_padded_tiles(int N, int H, int W) {
  if (N <= 0 || H < 2 || W < 2 || H % 2 || W % 2) return 0;
  const long long T = (long long)N * (H / 2) * (W / 2);
  const long long Tp = (T + 255) / 256 * 256;
  return Tp < ((long long)1 << 31) ? (int)Tp : 0;
}

int t2o_wino_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream) {
  return t2o_wino_input_transform_ld(x, V, N, H, W, C, 0, stream);
}

int t2o_wino_input_transform_ld(const float* x, float* V, int N, int H, int W, int C, int plane_rows, void* stream) {
  if (!x || !V || !wino_shape_ok(N, H, W, C)) return set_error(T2O_EINVAL, "wino_input_transform: null pointer or bad shape (H, W even; C a power of two in [4, 1024])");
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(V)) & 15) return set_error(T2O_EINVAL, "wino_input_transform: tensors must be 16-byte aligned");
  const int Tpad = t2o_wino_padded_tiles(N, H, W);
  if (Tpad <= 0) return set_error(T2O_EUNSUPPORTED, "wino_input_transform: too many tiles");
  if (plane_rows != 0 && plane_rows < Tpad) return set_error(T2O_EINVAL, "wino_input_transform: plane_rows smaller than the padded tile count");
  const size_t work = (size_t)Tpad * (C / 4);
  k_wino_input<<<(unsigned)((work + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(x, V, N, H, W, C, Tpad, plane_rows ? plane_rows : Tpad);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_input_transform: launch failed");
}

int t2o_wino_dy_transform(const float* dy, float* Ad, int N, int H, int W, int C, void* stream) {
  if (!dy || !Ad || !wino_shape_ok(N, H, W, C)) return set_error(T2O_EINVAL, "wino_dy_transform: null pointer or bad shape (H, W even; C a power of two in [4, 1024])");
  if ((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(Ad)) & 15) return set_error(T2O_EINVAL, "wino_dy_transform: tensors must be 16-byte aligned");
  const int Tpad = t2o_wino_padded_tiles(N, H, W);
  if (Tpad <= 0) return set_error(T2O_EUNSUPPORTED, "wino_dy_transform: too many tiles");
  const size_t work = (size_t)Tpad * (C / 4);
  k_wino_dy<<<(unsigned)((work + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(dy, Ad, N, H, W, C, Tpad, Tpad);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_dy_transform: launch failed");
}

int t2o_wino_dy_transform_ld(const float* dy, float* Ad, int N, int H, int W, int C, int plane_rows, void* stream) {
  if (!dy || !Ad || !wino_shape_ok(N, H, W, C)) return set_error(T2O_EINVAL, "wino_dy_transform: null pointer or bad shape (H, W even; C a power of two in [4, 1024])");
  if ((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(Ad)) & 15) return set_error(T2O_EINVAL, "wino_dy_transform: tensors must be 16-byte aligned");
  const int Tpad = t2o_wino_padded_tiles(N, H, W);
  if (Tpad <= 0) return set_error(T2O_EUNSUPPORTED, "wino_dy_transform: too many tiles");
  if (plane_rows != 0 && plane_rows < Tpad) return set_error(T2O_EINVAL, "wino_dy_transform: plane_rows smaller than the padded tile count");
  const size_t work = (size_t)Tpad * (C / 4);
  k_wino_dy<<<(unsigned)((work + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(dy, Ad, N, H, W, C, Tpad, plane_rows ? plane_rows : Tpad);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_dy_transform: launch failed");
}

int t2o_wino_dy_transforms(const float* dy, float* V, float* Ad, int N, int H, int W, int C, void* stream) {
  return t2o_wino_dy_transforms_ld(dy, V, Ad, N, H, W, C, 0, stream);
}

int t2o_wino_dy_transforms_ld(const float* dy, float* V, float* Ad, int N, int H, int W, int C, int ad_plane_rows, void* stream) {
  if (!dy || !V || !Ad || !wino_shape_ok(N, H, W, C)) return set_error(T2O_EINVAL, "wino_dy_transforms: null pointer or bad shape (H, W even; C a power of two in [4, 1024])");
  if ((reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(V) | reinterpret_cast<size_t>(Ad)) & 15) return set_error(T2O_EINVAL, "wino_dy_transforms: tensors must be 16-byte aligned");
  const int Tpad = t2o_wino_padded_tiles(N, H, W);
  if (Tpad <= 0) return set_error(T2O_EUNSUPPORTED, "wino_dy_transforms: too many tiles");
  if (ad_plane_rows != 0 && ad_plane_rows < Tpad) return set_error(T2O_EINVAL, "wino_dy_transforms: ad_plane_rows smaller than the padded tile count");
  const size_t work = (size_t)Tpad * (C / 4);
  k_wino_dy_both<<<(unsigned)((work + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(dy, V, Ad, N, H, W, C, Tpad, ad_plane_rows ? ad_plane_rows : Tpad);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_dy_transforms: launch failed");
}

int t2o_wino_dw_transform(const float* dU, float* dw, int Co, int Ci, int splits, int accumulate, void* stream) {
  if (!dU || !dw || Co <= 0 || Ci <= 0 || splits < 1) return set_error(T2O_EINVAL, "wino_dw_transform: null pointer or bad shape");
  const size_t total = (size_t)Co * Ci;
  k_wino_dw<<<(unsigned)((total + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(dU, dw, Co, Ci, accumulate ? 1 : 0, splits);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_dw_transform: launch failed");
}

int t2o_wino_stats_rows(int N, int H, int W, int C) { return wino_shape_ok(N, H, W, C) ? output_blocks(N, H, W, C) : 0; }

int t2o_wino_output_transform(const float* M, const float* addend, float* y, float* stats, int N, int H, int W, int C, void* stream) {
  if (!M || !y || !wino_shape_ok(N, H, W, C)) return set_error(T2O_EINVAL, "wino_output_transform: null pointer or bad shape (H, W even; C a power of two in [4, 1024])");
  if ((reinterpret_cast<size_t>(M) | reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(addend) | reinterpret_cast<size_t>(stats)) & 15)
    return set_error(T2O_EINVAL, "wino_output_transform: tensors must be 16-byte aligned");
  const unsigned grid = (unsigned)output_blocks(N, H, W, C);
  if (stats) k_wino_output<true><<<grid, kThreads, 0, (hipStream_t)stream>>>(M, addend, y, stats, N, H, W, C);
  else k_wino_output<false><<<grid, kThreads, 0, (hipStream_t)stream>>>(M, addend, y, nullptr, N, H, W, C);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_output_transform: launch failed");
}

}  // extern "C"
