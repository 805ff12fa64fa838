// t2o_norm.hip -- training-mode BatchNorm2d fused with the residual add and the ReLU that follow it in
// the actor's image encoder (models/actor_resnet.py:38-44 BasicBlock.forward, :99 stem, i.e.
//     out = relu(bn(x))            and            out = relu(bn(x) + shortcut)
// ).  The convolutions stay in MIOpen; these are the HBM-bound passes between them.  NCHW fp32.
//
//   forward   k_bn_stats     per-(channel, split) sum / sum of squares           reads x
//             k_bn_finalize  mean, biased var -> invstd, scale/shift, running stats (momentum, unbiased)
//             k_bn_apply     y = max(x * scale_c + shift_c (+ res), 0)           reads x (res), writes y
//   backward  k_bn_bwd_sums  g = dy * [y > 0]; per-(channel, split) sum g, sum g * xhat   reads x, dy (y)
//             k_bn_bwd_finalize  dgamma, dbeta, per-channel coefficients
//             k_bn_bwd_apply dx = a_c * (g - mean_g - xhat * mean_gxhat) (, dres = g)     reads x, dy (y), writes dx
// Without a residual the ReLU mask is recomputed from x (the same expression the forward evaluated), so
// the backward reads x and dy only: 3 + 5 tensor passes for a layer instead of the 5 + 9 of the separate
// batch-norm, ReLU and threshold kernels.  Sums: fp32 per thread (<= a few thousand terms), fp64 across
// workgroups; fixed order => reproducible.
#include <hip/hip_runtime.h>

#include "t2onet_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;          // independent 16-byte loads per thread before their first use

struct BnArgs {
  const float* x;        // (N,C,HW) input of the batch norm
  const float* res;      // residual added before the ReLU, or null
  const float* dy;       // backward: gradient w.r.t. the fused output
  const float* y;        // backward with residual: the fused output (ReLU mask)
  float* out;            // forward: y;  backward: dx
  float* dres;           // backward: gradient w.r.t. res (= g), or null
  const float* weight;   // gamma (C)
  const float* bias;     // beta (C)
  float* running_mean;   // (C) or null
  float* running_var;    // (C) or null
  float* save_mean;      // (C) batch mean
  float* save_invstd;    // (C) 1 / sqrt(biased var + eps)
  float* coef;           // (4,C) scratch: forward scale, shift; backward a, mean_g, mean_gxhat (rows reused)
  float* dweight;        // (C)
  float* dbias;          // (C)
  double* partials;      // (C, splits, 2)
  int N, C, HW, splits;
  float eps, momentum;
  int relu;              // NHWC kernels: 0 = plain batch norm (the shortcut branch, no activation); the NCHW kernels always apply it
  int acc;               // backward: dweight / dbias are ADDED to (a trainer's persistent, pre-zeroed gradient buffers)
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// two running sums of this block -> partials[(c * splits + s) * 2 + {0,1}] as doubles
__device__ __forceinline__ void block_sum2_store(float a, float b, double* dst) {
  __shared__ float sa[kThreads / 64], sb[kThreads / 64];
  a = wave_sum_f(a);
  b = wave_sum_f(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sa[wave] = a; sb[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dst[0] = ((double)sa[0] + (double)sa[1]) + ((double)sa[2] + (double)sa[3]);
    dst[1] = ((double)sb[0] + (double)sb[1]) + ((double)sb[2] + (double)sb[3]);
  }
}

// channel c, split s: planes n = s, s + splits, ...; a plane is HW contiguous floats
template <int V>
__global__ __launch_bounds__(kThreads) void k_bn_stats(BnArgs a) {
  const int c = blockIdx.x / a.splits, s = blockIdx.x % a.splits;
  float sum = 0.0f, sq = 0.0f;
  for (int n = s; n < a.N; n += a.splits) {
    const float* p = a.x + ((size_t)n * a.C + c) * a.HW;
    if (V == 4) {
      for (int i0 = threadIdx.x * 4; i0 < a.HW; i0 += kThreads * 4 * kUnroll) {
        float4 v[kUnroll];
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {               // independent loads first: kUnroll x 16 B in flight per thread
          const int i = i0 + k * kThreads * 4;
          v[k] = i < a.HW ? *reinterpret_cast<const float4*>(p + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
          sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
          sq += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
      }
    } else {
      for (int i = threadIdx.x; i < a.HW; i += kThreads) { const float v = p[i]; sum += v; sq += v * v; }
    }
  }
  block_sum2_store(sum, sq, a.partials + ((size_t)c * a.splits + s) * 2);
}

// per-channel epilogue of the statistics pass: mean, biased variance -> invstd, scale / shift, running statistics
__device__ __forceinline__ void bn_finalize_channel(const BnArgs& a, int c, double sum, double sq) {
  const double m = (double)a.N * a.HW;
  const double mean = sum / m;
  double var = sq / m - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
  a.save_mean[c] = (float)mean;
  a.save_invstd[c] = invstd;
  const float scale = a.weight[c] * invstd;
  a.coef[c] = scale;
  a.coef[a.C + c] = a.bias[c] - (float)mean * scale;
  if (a.running_mean) {
    const double unbiased = m > 1.0 ? var * m / (m - 1.0) : var;
    a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
    a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(kThreads) void k_bn_finalize(BnArgs a) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= a.C) return;
  double sum = 0.0, sq = 0.0;
  for (int s = 0; s < a.splits; ++s) { sum += a.partials[((size_t)c * a.splits + s) * 2]; sq += a.partials[((size_t)c * a.splits + s) * 2 + 1]; }
  bn_finalize_channel(a, c, sum, sq);
}

// channel of flat element e (32-bit arithmetic when the tensor has < 2^32 elements: the 64-bit divide is
// ~4x the instructions)
__device__ __forceinline__ int channel_of(size_t e, const BnArgs& a, bool small) {
  return small ? (int)(((unsigned)e / (unsigned)a.HW) % (unsigned)a.C) : (int)((e / (size_t)a.HW) % (size_t)a.C);
}

// flat over all N*C*HW elements: a workgroup takes spans of kThreads * V * kUnroll consecutive elements,
// a thread kUnroll groups of V (loads first, then arithmetic, then stores)
template <int V, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void k_bn_apply(BnArgs a, size_t total) {
  const bool small = total < ((size_t)1 << 32);
  const size_t span = (size_t)kThreads * V * kUnroll, stride = (size_t)gridDim.x * span;
  for (size_t e0 = (size_t)blockIdx.x * span + (size_t)threadIdx.x * V; e0 < total; e0 += stride) {
    if (V == 4) {
      float4 v[kUnroll], r[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const size_t e = e0 + (size_t)k * kThreads * 4;
        const bool in = e < total;
        v[k] = in ? *reinterpret_cast<const float4*>(a.x + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        r[k] = (HAS_RES && in) ? *reinterpret_cast<const float4*>(a.res + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const size_t e = e0 + (size_t)k * kThreads * 4;
        if (e >= total) break;
        const int c = channel_of(e, a, small);
        const float sc = a.coef[c], sh = a.coef[a.C + c];
        float4 o = make_float4(v[k].x * sc + sh, v[k].y * sc + sh, v[k].z * sc + sh, v[k].w * sc + sh);
        if (HAS_RES) { o.x += r[k].x; o.y += r[k].y; o.z += r[k].z; o.w += r[k].w; }
        o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); o.z = fmaxf(o.z, 0.0f); o.w = fmaxf(o.w, 0.0f);
        *reinterpret_cast<float4*>(a.out + e) = o;
      }
    } else {
      for (int k = 0; k < kUnroll; ++k) {
        const size_t e = e0 + (size_t)k * kThreads;
        if (e >= total) break;
        const int c = channel_of(e, a, small);
        float o = a.x[e] * a.coef[c] + a.coef[a.C + c];
        if (HAS_RES) o += a.res[e];
        a.out[e] = fmaxf(o, 0.0f);
      }
    }
  }
}

// ReLU-gated gradient of one element: with a residual the mask comes from the saved output, without one
// it is recomputed from x exactly as the forward evaluated it
template <bool HAS_RES>
__device__ __forceinline__ float gated(float dy, float x, float y, float sc, float sh) {
  const bool pass = HAS_RES ? (y > 0.0f) : (x * sc + sh > 0.0f);
  return pass ? dy : 0.0f;
}

// the same with the activation optional (NHWC kernels; relu == 0: the gradient passes unchanged)
template <bool HAS_RES>
__device__ __forceinline__ float gated_opt(int relu, float dy, float x, float y, float sc, float sh) {
  return relu ? gated<HAS_RES>(dy, x, y, sc, sh) : dy;
}

template <int V, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_sums(BnArgs a) {
  const int c = blockIdx.x / a.splits, s = blockIdx.x % a.splits;
  const float mean = a.save_mean[c], invstd = a.save_invstd[c];
  const float sc = a.weight[c] * invstd, sh = a.bias[c] - mean * sc;
  float sg = 0.0f, sgx = 0.0f;
  for (int n = s; n < a.N; n += a.splits) {
    const size_t base = ((size_t)n * a.C + c) * a.HW;
    if (V == 4) {
      for (int i0 = threadIdx.x * 4; i0 < a.HW; i0 += kThreads * 4 * 2) {
       float4 xq[2], dq[2], yq[2];
#pragma unroll
       for (int k = 0; k < 2; ++k) {
        const int i = i0 + k * kThreads * 4;
        const bool in = i < a.HW;
        xq[k] = in ? *reinterpret_cast<const float4*>(a.x + base + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        dq[k] = in ? *reinterpret_cast<const float4*>(a.dy + base + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        yq[k] = (HAS_RES && in) ? *reinterpret_cast<const float4*>(a.y + base + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
       }
#pragma unroll
       for (int k = 0; k < 2; ++k) {
        const float4 xv = xq[k], dv = dq[k], yv = yq[k];
        const float g0 = gated<HAS_RES>(dv.x, xv.x, yv.x, sc, sh), g1 = gated<HAS_RES>(dv.y, xv.y, yv.y, sc, sh);
        const float g2 = gated<HAS_RES>(dv.z, xv.z, yv.z, sc, sh), g3 = gated<HAS_RES>(dv.w, xv.w, yv.w, sc, sh);
        sg += (g0 + g1) + (g2 + g3);
        sgx += (g0 * ((xv.x - mean) * invstd) + g1 * ((xv.y - mean) * invstd)) +
               (g2 * ((xv.z - mean) * invstd) + g3 * ((xv.w - mean) * invstd));
       }
      }
    } else {
      for (int i = threadIdx.x; i < a.HW; i += kThreads) {
        const float xv = a.x[base + i];
        const float g = gated<HAS_RES>(a.dy[base + i], xv, HAS_RES ? a.y[base + i] : 0.0f, sc, sh);
        sg += g;
        sgx += g * ((xv - mean) * invstd);
      }
    }
  }
  block_sum2_store(sg, sgx, a.partials + ((size_t)c * a.splits + s) * 2);
}

// per-channel epilogue of the backward sums: dgamma, dbeta and the coefficients of the apply pass
__device__ __forceinline__ void bn_bwd_finalize_channel(const BnArgs& a, int c, double sg, double sgx) {
  const double m = (double)a.N * a.HW;
  if (a.dbias) a.dbias[c] = a.acc ? a.dbias[c] + (float)sg : (float)sg;
  if (a.dweight) a.dweight[c] = a.acc ? a.dweight[c] + (float)sgx : (float)sgx;
  a.coef[c] = a.weight[c] * a.save_invstd[c];     // a_c
  a.coef[a.C + c] = (float)(sg / m);             // mean of g
  a.coef[2 * a.C + c] = (float)(sgx / m);        // mean of g * xhat
}

__global__ __launch_bounds__(kThreads) void k_bn_bwd_finalize(BnArgs a) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= a.C) return;
  double sg = 0.0, sgx = 0.0;
  for (int s = 0; s < a.splits; ++s) { sg += a.partials[((size_t)c * a.splits + s) * 2]; sgx += a.partials[((size_t)c * a.splits + s) * 2 + 1]; }
  bn_bwd_finalize_channel(a, c, sg, sgx);
}

template <int V, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(BnArgs a, size_t total) {
  constexpr int U = 2;
  const bool small = total < ((size_t)1 << 32);
  const size_t span = (size_t)kThreads * V * U, stride = (size_t)gridDim.x * span;
  for (size_t e0 = (size_t)blockIdx.x * span + (size_t)threadIdx.x * V; e0 < total; e0 += stride) {
    if (V == 4) {
      float4 xq[U], dq[U], yq[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const size_t e = e0 + (size_t)k * kThreads * 4;
        const bool in = e < total;
        xq[k] = in ? *reinterpret_cast<const float4*>(a.x + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        dq[k] = in ? *reinterpret_cast<const float4*>(a.dy + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        yq[k] = (HAS_RES && in) ? *reinterpret_cast<const float4*>(a.y + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const size_t e = e0 + (size_t)k * kThreads * 4;
        if (e >= total) break;
        const int c = channel_of(e, a, small);
        const float mean = a.save_mean[c], invstd = a.save_invstd[c];
        const float ac = a.coef[c], mg = a.coef[a.C + c], mgx = a.coef[2 * a.C + c];
        const float sc = ac, sh = a.bias[c] - mean * sc;
        const float4 xv = xq[k], dv = dq[k], yv = yq[k];
        float4 g;
        g.x = gated<HAS_RES>(dv.x, xv.x, yv.x, sc, sh); g.y = gated<HAS_RES>(dv.y, xv.y, yv.y, sc, sh);
        g.z = gated<HAS_RES>(dv.z, xv.z, yv.z, sc, sh); g.w = gated<HAS_RES>(dv.w, xv.w, yv.w, sc, sh);
        float4 o;
        o.x = ac * ((g.x - mg) - ((xv.x - mean) * invstd) * mgx);
        o.y = ac * ((g.y - mg) - ((xv.y - mean) * invstd) * mgx);
        o.z = ac * ((g.z - mg) - ((xv.z - mean) * invstd) * mgx);
        o.w = ac * ((g.w - mg) - ((xv.w - mean) * invstd) * mgx);
        *reinterpret_cast<float4*>(a.out + e) = o;
        if (HAS_RES && a.dres) *reinterpret_cast<float4*>(a.dres + e) = g;
      }
    } else {
      for (int k = 0; k < U; ++k) {
        const size_t e = e0 + (size_t)k * kThreads;
        if (e >= total) break;
        const int c = channel_of(e, a, small);
        const float mean = a.save_mean[c], invstd = a.save_invstd[c];
        const float ac = a.coef[c], mg = a.coef[a.C + c], mgx = a.coef[2 * a.C + c];
        const float sc = ac, sh = a.bias[c] - mean * sc;
        const float xv = a.x[e];
        const float g = gated<HAS_RES>(a.dy[e], xv, HAS_RES ? a.y[e] : 0.0f, sc, sh);
        a.out[e] = ac * ((g - mg) - ((xv - mean) * invstd) * mgx);
        if (HAS_RES && a.dres) a.dres[e] = g;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// NHWC (channels-last) forms: x is (M, C) with M = N*H*W rows of C contiguous channels -- the layout MIOpen's
// fp32 implicit-GEMM kernels run in natively, so the encoder needs no layout transposes around its convolutions.
// Thread t of a 256-thread workgroup owns the channel quad (t % (C/4)) for the whole kernel: its scale / shift /
// mean coefficients are loaded ONCE into registers (no per-element channel arithmetic at all), every access is
// a 16-byte load of 4 consecutive channels, a wave reads 1 KiB of consecutive memory.  C must be a power of two
// in [4, 1024] (the encoder: 64 / 128 / 256 / 512).
//   sums kernels   workgroup b reduces rows [b * rows_per_block, ...): per-thread fp32, LDS across the R = 1024/C
//                  row slots of the workgroup, one (2, C) fp32 row of partials per workgroup
//   finalize       fp64 over the workgroups' partials in a fixed order, then exactly the NCHW finalize arithmetic
//   apply kernels  flat over the float4s, kUnroll independent loads per thread
// ---------------------------------------------------------------------------------------------------------
constexpr int kNhwcMaxBlocks = 1024;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }

// reduce the workgroup's per-thread quads over its R row slots and store one partial row: dst[0..C) and dst[C..2C)
__device__ __forceinline__ void nhwc_block_store(float4 s1, float4 s2, int q, float* dst, int C) {
  __shared__ float4 sm[2][kThreads];
  sm[0][threadIdx.x] = s1;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  if ((int)threadIdx.x < q) {
    float4 a = sm[0][threadIdx.x], b = sm[1][threadIdx.x];
    for (int t = threadIdx.x + q; t < kThreads; t += q) {          // fixed order
      const float4 u = sm[0][t], v = sm[1][t];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    *reinterpret_cast<float4*>(dst + 4 * threadIdx.x) = a;
    *reinterpret_cast<float4*>(dst + C + 4 * threadIdx.x) = b;
  }
}

__global__ __launch_bounds__(kThreads) void k_bn_nhwc_stats(BnArgs a, int M, int rows_per_block, float* partial) {
  const int q = a.C >> 2, cq = threadIdx.x % q, r = threadIdx.x / q, R = kThreads / q;
  const int row0 = blockIdx.x * rows_per_block, row1 = min(row0 + rows_per_block, M);
  const float* base = a.x + 4 * cq;
  float4 s = f4(0.0f), ss = f4(0.0f);
  for (int row = row0 + r; row < row1; row += R * kUnroll) {
    float4 v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int rr = row + k * R;
      v[k] = rr < row1 ? ld4(base + (size_t)rr * a.C) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w;
      ss.x += v[k].x * v[k].x; ss.y += v[k].y * v[k].y; ss.z += v[k].z * v[k].z; ss.w += v[k].w * v[k].w;
    }
  }
  nhwc_block_store(s, ss, q, partial + (size_t)blockIdx.x * 2 * a.C, a.C);
}

// partial (nblk, 2, C) fp32 -> per-channel fp64 totals in a fixed order, then the channel epilogue.  A workgroup
// takes 4 channels: thread = (slice s of 64, channel quad), each slice adds the workgroups b = s, s + 64, ...
// (one 16-byte load per partial row), slices are combined through LDS by the first 4 threads.
// (stride = floats per partial row, the second quantity at column offset off1: (2C, C) for the plain forms, (3C, C | 2C) for the
// two batch norms of a shortcut block, which share the first quantity; block = which 4 channels)
template <bool BWD>
__device__ __forceinline__ void bn_nhwc_finalize_body(const BnArgs& a, const float* partial, int nblk, int stride, int off1, int block) {
  __shared__ double sm[2][kThreads / 64][4];
  const int sl = threadIdx.x >> 2, j = threadIdx.x & 3;
  const int c = block * 4 + j;
  // the channel epilogue's own operands travel while the partial rows are being added (the kernel is one dependent chain:
  // every microsecond of latency taken out of it is a microsecond of an otherwise idle GPU)
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
  if (threadIdx.x < 4) {
    p0 = a.weight[c];
    if (BWD) {
      p1 = a.save_invstd[c];
      if (a.acc) { p2 = a.dbias ? a.dbias[c] : 0.0f; p3 = a.dweight ? a.dweight[c] : 0.0f; }
    } else {
      p1 = a.bias[c];
      if (a.running_mean) { p2 = a.running_mean[c]; p3 = a.running_var[c]; }
    }
  }
  double s0 = 0.0, s1 = 0.0;
  // (measured: issuing a slice's 16 row loads 8 at a time made this kernel SLOWER, 8.4 -> 12.5 us -- it is a 16-workgroup
  // launch whose time is the launch itself plus one dependent chain load -> reduce -> coefficient loads -> stores)
  for (int b = sl; b < nblk; b += 64) {
    s0 += (double)partial[(size_t)b * stride + c];
    s1 += (double)partial[(size_t)b * stride + off1 + c];
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {                 // the 16 slices of this wave (lane = slice * 4 + channel)
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 4) { sm[0][wave][lane] = s0; sm[1][wave][lane] = s1; }
  __syncthreads();
  if (threadIdx.x < 4) {
    s0 = (sm[0][0][j] + sm[0][1][j]) + (sm[0][2][j] + sm[0][3][j]);
    s1 = (sm[1][0][j] + sm[1][1][j]) + (sm[1][2][j] + sm[1][3][j]);
    const double m = (double)a.N * a.HW;
    if (BWD) {                                         // bn_bwd_finalize_channel on the prefetched operands
      if (a.dbias) a.dbias[c] = a.acc ? p2 + (float)s0 : (float)s0;
      if (a.dweight) a.dweight[c] = a.acc ? p3 + (float)s1 : (float)s1;
      a.coef[c] = p0 * p1;
      a.coef[a.C + c] = (float)(s0 / m);
      a.coef[2 * a.C + c] = (float)(s1 / m);
    } else {                                           // bn_finalize_channel on the prefetched operands
      const double mean = s0 / m;
      double var = s1 / m - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
      a.save_mean[c] = (float)mean;
      a.save_invstd[c] = invstd;
      const float scale = p0 * invstd;
      a.coef[c] = scale;
      a.coef[a.C + c] = p1 - (float)mean * scale;
      if (a.running_mean) {
        const double unbiased = m > 1.0 ? var * m / (m - 1.0) : var;
        a.running_mean[c] = (1.0f - a.momentum) * p2 + a.momentum * (float)mean;
        a.running_var[c] = (1.0f - a.momentum) * p3 + a.momentum * (float)unbiased;
      }
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(kThreads) void k_bn_nhwc_finalize(BnArgs a, const float* partial, int nblk) {
  bn_nhwc_finalize_body<BWD>(a, partial, nblk, 2 * a.C, a.C, blockIdx.x);
}

// both batch norms of a shortcut block in ONE launch (2 x C/4 workgroups): the main branch's from (partial0, n0, stride0,
// off0), the shortcut's from (partial1, n1, stride1, off1)
template <bool BWD>
__global__ __launch_bounds__(kThreads) void k_bn_nhwc_finalize_pair(BnArgs a0, const float* partial0, int n0, int stride0, int off0,
                                                                    BnArgs a1, const float* partial1, int n1, int stride1, int off1) {
  const int per = a0.C >> 2;
  if ((int)blockIdx.x < per) bn_nhwc_finalize_body<BWD>(a0, partial0, n0, stride0, off0, blockIdx.x);
  else bn_nhwc_finalize_body<BWD>(a1, partial1, n1, stride1, off1, blockIdx.x - per);
}

template <bool HAS_RES>
__global__ __launch_bounds__(kThreads) void k_bn_nhwc_apply(BnArgs a, size_t total4) {
  const int q = a.C >> 2, cq = threadIdx.x % q;
  const float4 sc = ld4(a.coef + 4 * cq), sh = ld4(a.coef + a.C + 4 * cq);
  const size_t span = (size_t)kThreads * kUnroll, stride = (size_t)gridDim.x * span;
  for (size_t i0 = (size_t)blockIdx.x * span + threadIdx.x; i0 < total4; i0 += stride) {
    float4 v[kUnroll], rs[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      const bool in = i < total4;
      v[k] = in ? ld4(a.x + 4 * i) : f4(0.0f);
      rs[k] = (HAS_RES && in) ? ld4(a.res + 4 * i) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      if (i >= total4) break;
      float4 o = make_float4(v[k].x * sc.x + sh.x, v[k].y * sc.y + sh.y, v[k].z * sc.z + sh.z, v[k].w * sc.w + sh.w);
      if (HAS_RES) { o.x += rs[k].x; o.y += rs[k].y; o.z += rs[k].z; o.w += rs[k].w; }
      if (a.relu) { o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); o.z = fmaxf(o.z, 0.0f); o.w = fmaxf(o.w, 0.0f); }
      *reinterpret_cast<float4*>(a.out + 4 * i) = o;
    }
  }
}

template <bool HAS_RES>
__global__ __launch_bounds__(kThreads) void k_bn_nhwc_bwd_sums(BnArgs a, int M, int rows_per_block, float* partial) {
  constexpr int U = 2;
  const int q = a.C >> 2, cq = threadIdx.x % q, r = threadIdx.x / q, R = kThreads / q;
  const int row0 = blockIdx.x * rows_per_block, row1 = min(row0 + rows_per_block, M);
  const float4 mean = ld4(a.save_mean + 4 * cq), invstd = ld4(a.save_invstd + 4 * cq);
  const float4 w = ld4(a.weight + 4 * cq), bi = ld4(a.bias + 4 * cq);
  const float4 sc = make_float4(w.x * invstd.x, w.y * invstd.y, w.z * invstd.z, w.w * invstd.w);
  const float4 sh = make_float4(bi.x - mean.x * sc.x, bi.y - mean.y * sc.y, bi.z - mean.z * sc.z, bi.w - mean.w * sc.w);
  float4 sg = f4(0.0f), sgx = f4(0.0f);
  for (int row = row0 + r; row < row1; row += R * U) {
    float4 xq[U], dq[U], yq[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int rr = row + k * R;
      const bool in = rr < row1;
      const size_t off = (size_t)rr * a.C + 4 * cq;
      xq[k] = in ? ld4(a.x + off) : f4(0.0f);
      dq[k] = in ? ld4(a.dy + off) : f4(0.0f);
      yq[k] = (HAS_RES && in) ? ld4(a.y + off) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const float g0 = gated_opt<HAS_RES>(a.relu, dq[k].x, xq[k].x, yq[k].x, sc.x, sh.x), g1 = gated_opt<HAS_RES>(a.relu, dq[k].y, xq[k].y, yq[k].y, sc.y, sh.y);
      const float g2 = gated_opt<HAS_RES>(a.relu, dq[k].z, xq[k].z, yq[k].z, sc.z, sh.z), g3 = gated_opt<HAS_RES>(a.relu, dq[k].w, xq[k].w, yq[k].w, sc.w, sh.w);
      sg.x += g0; sg.y += g1; sg.z += g2; sg.w += g3;
      sgx.x += g0 * ((xq[k].x - mean.x) * invstd.x); sgx.y += g1 * ((xq[k].y - mean.y) * invstd.y);
      sgx.z += g2 * ((xq[k].z - mean.z) * invstd.z); sgx.w += g3 * ((xq[k].w - mean.w) * invstd.w);
    }
  }
  nhwc_block_store(sg, sgx, q, partial + (size_t)blockIdx.x * 2 * a.C, a.C);
}

template <bool HAS_RES>
__global__ __launch_bounds__(kThreads) void k_bn_nhwc_bwd_apply(BnArgs a, size_t total4) {
  constexpr int U = 2;
  const int q = a.C >> 2, cq = threadIdx.x % q;
  const float4 mean = ld4(a.save_mean + 4 * cq), invstd = ld4(a.save_invstd + 4 * cq), bi = ld4(a.bias + 4 * cq);
  const float4 ac = ld4(a.coef + 4 * cq), mg = ld4(a.coef + a.C + 4 * cq), mgx = ld4(a.coef + 2 * a.C + 4 * cq);
  const float4 sh = make_float4(bi.x - mean.x * ac.x, bi.y - mean.y * ac.y, bi.z - mean.z * ac.z, bi.w - mean.w * ac.w);
  const size_t span = (size_t)kThreads * U, stride = (size_t)gridDim.x * span;
  for (size_t i0 = (size_t)blockIdx.x * span + threadIdx.x; i0 < total4; i0 += stride) {
    float4 xq[U], dq[U], yq[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      const bool in = i < total4;
      xq[k] = in ? ld4(a.x + 4 * i) : f4(0.0f);
      dq[k] = in ? ld4(a.dy + 4 * i) : f4(0.0f);
      yq[k] = (HAS_RES && in) ? ld4(a.y + 4 * i) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      if (i >= total4) break;
      float4 g;
      g.x = gated_opt<HAS_RES>(a.relu, dq[k].x, xq[k].x, yq[k].x, ac.x, sh.x); g.y = gated_opt<HAS_RES>(a.relu, dq[k].y, xq[k].y, yq[k].y, ac.y, sh.y);
      g.z = gated_opt<HAS_RES>(a.relu, dq[k].z, xq[k].z, yq[k].z, ac.z, sh.z); g.w = gated_opt<HAS_RES>(a.relu, dq[k].w, xq[k].w, yq[k].w, ac.w, sh.w);
      float4 o;
      o.x = ac.x * ((g.x - mg.x) - ((xq[k].x - mean.x) * invstd.x) * mgx.x);
      o.y = ac.y * ((g.y - mg.y) - ((xq[k].y - mean.y) * invstd.y) * mgx.y);
      o.z = ac.z * ((g.z - mg.z) - ((xq[k].z - mean.z) * invstd.z) * mgx.z);
      o.w = ac.w * ((g.w - mg.w) - ((xq[k].w - mean.w) * invstd.w) * mgx.w);
      *reinterpret_cast<float4*>(a.out + 4 * i) = o;
      if (HAS_RES && a.dres) *reinterpret_cast<float4*>(a.dres + 4 * i) = g;
    }
  }
}

// ---- the two batch norms of a shortcut block as one pass each way (models/actor_resnet.py:33-36, 42-44) ---------------
//   out = relu(bn2(y2) + bn_s(ys))           ys = the 1x1 stride-2 shortcut convolution's output
// Separately that is: apply(bn_s) writes sc, apply(bn2, res = sc) re-reads it; backward: sums / finalize / apply for bn2
// (writing the gated gradient g for the shortcut), then sums / finalize / apply again for bn_s over the same g.  Both
// branches see the SAME gated gradient g = dout * [out > 0], so one sums pass yields sum g, sum g xhat2, sum g xhat_s and one
// apply pass writes both input gradients: 3 launches instead of 5 forward, 3 instead of 6 backward, sc and g never stored.
// Same arithmetic in the same order as the separate kernels (bit-identical results).
struct BnDual {
  const float* xs;         // shortcut branch input (M, C)
  const float* coef_s;     // forward: (2, C) scale, shift of bn_s;  backward: (3, C) a, mean_g, mean_gxhat of bn_s
  const float* mean_s;     // backward: batch statistics / parameters of bn_s
  const float* invstd_s;
  float* out_s;            // backward: gradient w.r.t. xs
};

__global__ __launch_bounds__(kThreads) void k_bn_nhwc_apply_dual(BnArgs a, BnDual d, size_t total4) {
  const int q = a.C >> 2, cq = threadIdx.x % q;
  const float4 sc = ld4(a.coef + 4 * cq), sh = ld4(a.coef + a.C + 4 * cq);
  const float4 scs = ld4(d.coef_s + 4 * cq), shs = ld4(d.coef_s + a.C + 4 * cq);
  const size_t span = (size_t)kThreads * kUnroll, stride = (size_t)gridDim.x * span;
  for (size_t i0 = (size_t)blockIdx.x * span + threadIdx.x; i0 < total4; i0 += stride) {
    float4 v[kUnroll], rs[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      const bool in = i < total4;
      v[k] = in ? ld4(a.x + 4 * i) : f4(0.0f);
      rs[k] = in ? ld4(d.xs + 4 * i) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      if (i >= total4) break;
      const float4 r = make_float4(rs[k].x * scs.x + shs.x, rs[k].y * scs.y + shs.y, rs[k].z * scs.z + shs.z, rs[k].w * scs.w + shs.w);
      float4 o = make_float4(v[k].x * sc.x + sh.x, v[k].y * sc.y + sh.y, v[k].z * sc.z + sh.z, v[k].w * sc.w + sh.w);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); o.z = fmaxf(o.z, 0.0f); o.w = fmaxf(o.w, 0.0f);
      *reinterpret_cast<float4*>(a.out + 4 * i) = o;
    }
  }
}

// three running sums of the workgroup's threads -> one partial row (3, C)
__device__ __forceinline__ void nhwc_block_store3(float4 s1, float4 s2, float4 s3, int q, float* dst, int C) {
  __shared__ float4 sm[3][kThreads];
  sm[0][threadIdx.x] = s1;
  sm[1][threadIdx.x] = s2;
  sm[2][threadIdx.x] = s3;
  __syncthreads();
  if ((int)threadIdx.x < q) {
    float4 a = sm[0][threadIdx.x], b = sm[1][threadIdx.x], c = sm[2][threadIdx.x];
    for (int t = threadIdx.x + q; t < kThreads; t += q) {          // fixed order
      const float4 u = sm[0][t], v = sm[1][t], w = sm[2][t];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      c.x += w.x; c.y += w.y; c.z += w.z; c.w += w.w;
    }
    *reinterpret_cast<float4*>(dst + 4 * threadIdx.x) = a;
    *reinterpret_cast<float4*>(dst + C + 4 * threadIdx.x) = b;
    *reinterpret_cast<float4*>(dst + 2 * C + 4 * threadIdx.x) = c;
  }
}

__global__ __launch_bounds__(kThreads) void k_bn_nhwc_bwd_sums_dual(BnArgs a, BnDual d, int M, int rows_per_block, float* partial) {
  constexpr int U = 2;
  const int q = a.C >> 2, cq = threadIdx.x % q, r = threadIdx.x / q, R = kThreads / q;
  const int row0 = blockIdx.x * rows_per_block, row1 = min(row0 + rows_per_block, M);
  const float4 mean = ld4(a.save_mean + 4 * cq), invstd = ld4(a.save_invstd + 4 * cq);
  const float4 means = ld4(d.mean_s + 4 * cq), invstds = ld4(d.invstd_s + 4 * cq);
  float4 sg = f4(0.0f), sgx = f4(0.0f), sgs = f4(0.0f);
  for (int row = row0 + r; row < row1; row += R * U) {
    float4 xq[U], xsq[U], dq[U], yq[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int rr = row + k * R;
      const bool in = rr < row1;
      const size_t off = (size_t)rr * a.C + 4 * cq;
      xq[k] = in ? ld4(a.x + off) : f4(0.0f);
      xsq[k] = in ? ld4(d.xs + off) : f4(0.0f);
      dq[k] = in ? ld4(a.dy + off) : f4(0.0f);
      yq[k] = in ? ld4(a.y + off) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const float g0 = yq[k].x > 0.0f ? dq[k].x : 0.0f, g1 = yq[k].y > 0.0f ? dq[k].y : 0.0f;
      const float g2 = yq[k].z > 0.0f ? dq[k].z : 0.0f, g3 = yq[k].w > 0.0f ? dq[k].w : 0.0f;
      sg.x += g0; sg.y += g1; sg.z += g2; sg.w += g3;
      sgx.x += g0 * ((xq[k].x - mean.x) * invstd.x); sgx.y += g1 * ((xq[k].y - mean.y) * invstd.y);
      sgx.z += g2 * ((xq[k].z - mean.z) * invstd.z); sgx.w += g3 * ((xq[k].w - mean.w) * invstd.w);
      sgs.x += g0 * ((xsq[k].x - means.x) * invstds.x); sgs.y += g1 * ((xsq[k].y - means.y) * invstds.y);
      sgs.z += g2 * ((xsq[k].z - means.z) * invstds.z); sgs.w += g3 * ((xsq[k].w - means.w) * invstds.w);
    }
  }
  nhwc_block_store3(sg, sgx, sgs, q, partial + (size_t)blockIdx.x * 3 * a.C, a.C);
}

__global__ __launch_bounds__(kThreads) void k_bn_nhwc_bwd_apply_dual(BnArgs a, BnDual d, size_t total4) {
  constexpr int U = 2;
  const int q = a.C >> 2, cq = threadIdx.x % q;
  const float4 mean = ld4(a.save_mean + 4 * cq), invstd = ld4(a.save_invstd + 4 * cq);
  const float4 ac = ld4(a.coef + 4 * cq), mg = ld4(a.coef + a.C + 4 * cq), mgx = ld4(a.coef + 2 * a.C + 4 * cq);
  const float4 means = ld4(d.mean_s + 4 * cq), invstds = ld4(d.invstd_s + 4 * cq);
  const float4 acs = ld4(d.coef_s + 4 * cq), mgs = ld4(d.coef_s + a.C + 4 * cq), mgxs = ld4(d.coef_s + 2 * a.C + 4 * cq);
  const size_t span = (size_t)kThreads * U, stride = (size_t)gridDim.x * span;
  for (size_t i0 = (size_t)blockIdx.x * span + threadIdx.x; i0 < total4; i0 += stride) {
    float4 xq[U], xsq[U], dq[U], yq[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      const bool in = i < total4;
      xq[k] = in ? ld4(a.x + 4 * i) : f4(0.0f);
      xsq[k] = in ? ld4(d.xs + 4 * i) : f4(0.0f);
      dq[k] = in ? ld4(a.dy + 4 * i) : f4(0.0f);
      yq[k] = in ? ld4(a.y + 4 * i) : f4(0.0f);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const size_t i = i0 + (size_t)k * kThreads;
      if (i >= total4) break;
      float4 g;
      g.x = yq[k].x > 0.0f ? dq[k].x : 0.0f; g.y = yq[k].y > 0.0f ? dq[k].y : 0.0f;
      g.z = yq[k].z > 0.0f ? dq[k].z : 0.0f; g.w = yq[k].w > 0.0f ? dq[k].w : 0.0f;
      float4 o, os;
      o.x = ac.x * ((g.x - mg.x) - ((xq[k].x - mean.x) * invstd.x) * mgx.x);
      o.y = ac.y * ((g.y - mg.y) - ((xq[k].y - mean.y) * invstd.y) * mgx.y);
      o.z = ac.z * ((g.z - mg.z) - ((xq[k].z - mean.z) * invstd.z) * mgx.z);
      o.w = ac.w * ((g.w - mg.w) - ((xq[k].w - mean.w) * invstd.w) * mgx.w);
      os.x = acs.x * ((g.x - mgs.x) - ((xsq[k].x - means.x) * invstds.x) * mgxs.x);
      os.y = acs.y * ((g.y - mgs.y) - ((xsq[k].y - means.y) * invstds.y) * mgxs.y);
      os.z = acs.z * ((g.z - mgs.z) - ((xsq[k].z - means.z) * invstds.z) * mgxs.z);
      os.w = acs.w * ((g.w - mgs.w) - ((xsq[k].w - means.w) * invstds.w) * mgxs.w);
      *reinterpret_cast<float4*>(a.out + 4 * i) = o;
      *reinterpret_cast<float4*>(d.out_s + 4 * i) = os;
    }
  }
}

bool nhwc_channels_ok(int C) { return C >= 4 && C <= 1024 && (C & (C - 1)) == 0; }
// rows per workgroup (a multiple of the workgroup's R = 1024 / C row slots x the unroll) and the workgroup count
void nhwc_partition(int M, int C, int unroll, int* rows_per_block, int* nblk) {
  const int R = kThreads / (C >> 2), step = R * unroll;
  // at most kNhwcMaxBlocks workgroups, and at least 16 Ki elements for each (the finalize kernel reads every partial row: with
  // 1024 rows of 512 channels it read 4 MB to finish an 8 MB tensor -- 8.4 us; with 128 rows 5 us)
  long long maxb = (long long)M * C / 16384;
  if (maxb > kNhwcMaxBlocks) maxb = kNhwcMaxBlocks;
  if (maxb < 64) maxb = 64;
  int rpb = (int)((M + maxb - 1) / maxb);
  rpb = ((rpb + step - 1) / step) * step;
  *rows_per_block = rpb;
  *nblk = (M + rpb - 1) / rpb;
}

int bn_splits(int N, int C) {
  int s = 4096 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  if (s > N) s = N;
  return s;
}
unsigned flat_grid(size_t total, int V, int unroll) {
  const size_t span = (size_t)kThreads * V * unroll;
  size_t blocks = (total + span - 1) / span;
  const size_t cap = 256 * 32;                      // grid-stride above 32 workgroups per CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
bool bn_check(const void* x, int N, int C, int HW) { return x && N > 0 && C > 0 && HW > 0 && (size_t)N * C * HW < ((size_t)1 << 40); }

}  // namespace

namespace t2o { int set_error(int code, const char* msg); }   // t2o_kernels.hip: thread-local text behind t2o_last_error()
using t2o::set_error;

extern "C" {

size_t t2o_bn_workspace_bytes(int N, int C) {
  return sizeof(double) * 2 * (size_t)C * bn_splits(N, C) + sizeof(float) * 4 * (size_t)C;
}

int t2o_bn_relu_fwd(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                    float* running_var, float* save_mean, float* save_invstd, float* out, float momentum, float eps,
                    void* workspace, size_t workspace_bytes, int N, int C, int HW, void* stream) {
  if (!bn_check(x, N, C, HW) || !weight || !bias || !save_mean || !save_invstd || !out)
    return set_error(T2O_EINVAL, "bn_relu_fwd: null pointer or bad shape");
  if ((running_mean == nullptr) != (running_var == nullptr))
    return set_error(T2O_EINVAL, "bn_relu_fwd: running_mean and running_var must both be given or both be null");
  if (!workspace || workspace_bytes < t2o_bn_workspace_bytes(N, C)) return set_error(T2O_EWORKSPACE, "bn_relu_fwd: workspace too small");
  BnArgs a = {};
  a.x = x; a.res = res; a.out = out; a.weight = weight; a.bias = bias;
  a.running_mean = running_mean; a.running_var = running_var; a.save_mean = save_mean; a.save_invstd = save_invstd;
  a.N = N; a.C = C; a.HW = HW; a.splits = bn_splits(N, C); a.eps = eps; a.momentum = momentum;
  a.partials = (double*)workspace;
  a.coef = (float*)((char*)workspace + sizeof(double) * 2 * (size_t)C * a.splits);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * C * HW;
  const bool v4 = HW % 4 == 0;
  if (v4) k_bn_stats<4><<<C * a.splits, kThreads, 0, st>>>(a); else k_bn_stats<1><<<C * a.splits, kThreads, 0, st>>>(a);
  k_bn_finalize<<<(C + kThreads - 1) / kThreads, kThreads, 0, st>>>(a);
  if (v4) { if (res) k_bn_apply<4, true><<<flat_grid(total, 4, kUnroll), kThreads, 0, st>>>(a, total); else k_bn_apply<4, false><<<flat_grid(total, 4, kUnroll), kThreads, 0, st>>>(a, total); }
  else    { if (res) k_bn_apply<1, true><<<flat_grid(total, 1, kUnroll), kThreads, 0, st>>>(a, total); else k_bn_apply<1, false><<<flat_grid(total, 1, kUnroll), kThreads, 0, st>>>(a, total); }
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

int t2o_bn_relu_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* bias,
                    const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dweight,
                    float* dbias, int has_res, void* workspace, size_t workspace_bytes, int N, int C, int HW,
                    void* stream) {
  if (!bn_check(x, N, C, HW) || !dy || !weight || !bias || !save_mean || !save_invstd || !dx)
    return set_error(T2O_EINVAL, "bn_relu_bwd: null pointer or bad shape");
  if (has_res && !y) return set_error(T2O_EINVAL, "bn_relu_bwd: y is needed when a residual was added");
  if (!workspace || workspace_bytes < t2o_bn_workspace_bytes(N, C)) return set_error(T2O_EWORKSPACE, "bn_relu_bwd: workspace too small");
  BnArgs a = {};
  a.x = x; a.y = y; a.dy = dy; a.out = dx; a.dres = dres; a.weight = weight; a.bias = bias;
  a.save_mean = const_cast<float*>(save_mean); a.save_invstd = const_cast<float*>(save_invstd);
  a.dweight = dweight; a.dbias = dbias;
  a.N = N; a.C = C; a.HW = HW; a.splits = bn_splits(N, C);
  a.partials = (double*)workspace;
  a.coef = (float*)((char*)workspace + sizeof(double) * 2 * (size_t)C * a.splits);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * C * HW;
  const bool v4 = HW % 4 == 0;
  const unsigned g1 = C * a.splits;
  if (v4) { if (has_res) k_bn_bwd_sums<4, true><<<g1, kThreads, 0, st>>>(a); else k_bn_bwd_sums<4, false><<<g1, kThreads, 0, st>>>(a); }
  else    { if (has_res) k_bn_bwd_sums<1, true><<<g1, kThreads, 0, st>>>(a); else k_bn_bwd_sums<1, false><<<g1, kThreads, 0, st>>>(a); }
  k_bn_bwd_finalize<<<(C + kThreads - 1) / kThreads, kThreads, 0, st>>>(a);
  if (v4) { if (has_res) k_bn_bwd_apply<4, true><<<flat_grid(total, 4, 2), kThreads, 0, st>>>(a, total); else k_bn_bwd_apply<4, false><<<flat_grid(total, 4, 2), kThreads, 0, st>>>(a, total); }
  else    { if (has_res) k_bn_bwd_apply<1, true><<<flat_grid(total, 1, 2), kThreads, 0, st>>>(a, total); else k_bn_bwd_apply<1, false><<<flat_grid(total, 1, 2), kThreads, 0, st>>>(a, total); }
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

size_t t2o_bn_nhwc_workspace_bytes(int M, int C) {
  (void)M;
  return sizeof(double) * 2 * (size_t)C + sizeof(float) * 4 * (size_t)C + sizeof(float) * 2 * (size_t)C * kNhwcMaxBlocks;   // (the fp64 slot is unused now, kept for alignment)
}

int t2o_bn_relu_nhwc_fwd(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                         float* running_var, float* save_mean, float* save_invstd, float* out, float momentum, float eps,
                         int relu, void* workspace, size_t workspace_bytes, int M, int C, void* stream) {
  if (!x || M <= 0 || !nhwc_channels_ok(C) || !weight || !bias || !save_mean || !save_invstd || !out)
    return set_error(T2O_EINVAL, "bn_relu_nhwc_fwd: null pointer or bad shape (C must be a power of two in [4, 1024])");
  if ((running_mean == nullptr) != (running_var == nullptr))
    return set_error(T2O_EINVAL, "bn_relu_nhwc_fwd: running_mean and running_var must both be given or both be null");
  if (!workspace || workspace_bytes < t2o_bn_nhwc_workspace_bytes(M, C)) return set_error(T2O_EWORKSPACE, "bn_relu_nhwc_fwd: workspace too small");
  BnArgs a = {};
  a.x = x; a.res = res; a.out = out; a.weight = weight; a.bias = bias;
  a.running_mean = running_mean; a.running_var = running_var; a.save_mean = save_mean; a.save_invstd = save_invstd;
  a.N = M; a.C = C; a.HW = 1; a.splits = 1; a.eps = eps; a.momentum = momentum;     // finalize: m = N * HW = M
  a.relu = relu;
  a.partials = (double*)workspace;
  a.coef = (float*)((char*)workspace + sizeof(double) * 2 * (size_t)C);
  float* partial = a.coef + 4 * (size_t)C;
  hipStream_t st = (hipStream_t)stream;
  int rpb, nblk;
  nhwc_partition(M, C, kUnroll, &rpb, &nblk);
  k_bn_nhwc_stats<<<nblk, kThreads, 0, st>>>(a, M, rpb, partial);
  k_bn_nhwc_finalize<false><<<C / 4, kThreads, 0, st>>>(a, partial, nblk);
  const size_t total4 = (size_t)M * (C >> 2);
  const unsigned grid = flat_grid(total4, 1, kUnroll);
  if (res) k_bn_nhwc_apply<true><<<grid, kThreads, 0, st>>>(a, total4); else k_bn_nhwc_apply<false><<<grid, kThreads, 0, st>>>(a, total4);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

int t2o_bn_relu_nhwc_fwd_partials(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                                  float* running_var, float* save_mean, float* save_invstd, float* out, float momentum,
                                  float eps, int relu, const float* partial, int partial_rows, void* workspace,
                                  size_t workspace_bytes, int M, int C, void* stream) {
  if (!x || M <= 0 || !nhwc_channels_ok(C) || !weight || !bias || !save_mean || !save_invstd || !out)
    return set_error(T2O_EINVAL, "bn_relu_nhwc_fwd_partials: null pointer or bad shape (C must be a power of two in [4, 1024])");
  if (!partial || partial_rows <= 0) return set_error(T2O_EINVAL, "bn_relu_nhwc_fwd_partials: no partial sums");
  if ((running_mean == nullptr) != (running_var == nullptr))
    return set_error(T2O_EINVAL, "bn_relu_nhwc_fwd_partials: running_mean and running_var must both be given or both be null");
  if (!workspace || workspace_bytes < t2o_bn_nhwc_workspace_bytes(M, C)) return set_error(T2O_EWORKSPACE, "bn_relu_nhwc_fwd_partials: workspace too small");
  BnArgs a = {};
  a.x = x; a.res = res; a.out = out; a.weight = weight; a.bias = bias;
  a.running_mean = running_mean; a.running_var = running_var; a.save_mean = save_mean; a.save_invstd = save_invstd;
  a.N = M; a.C = C; a.HW = 1; a.splits = 1; a.eps = eps; a.momentum = momentum;
  a.relu = relu;
  a.partials = (double*)workspace;
  a.coef = (float*)((char*)workspace + sizeof(double) * 2 * (size_t)C);
  hipStream_t st = (hipStream_t)stream;
  k_bn_nhwc_finalize<false><<<C / 4, kThreads, 0, st>>>(a, partial, partial_rows);
  const size_t total4 = (size_t)M * (C >> 2);
  const unsigned grid = flat_grid(total4, 1, kUnroll);
  if (res) k_bn_nhwc_apply<true><<<grid, kThreads, 0, st>>>(a, total4); else k_bn_nhwc_apply<false><<<grid, kThreads, 0, st>>>(a, total4);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

int t2o_bn_relu_nhwc_bwd_acc(const float* x, const float* y, const float* dy, const float* weight, const float* bias,
                             const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dweight,
                             float* dbias, int has_res, int relu, int accumulate, void* workspace, size_t workspace_bytes,
                             int M, int C, void* stream) {
  if (!x || M <= 0 || !nhwc_channels_ok(C) || !dy || !weight || !bias || !save_mean || !save_invstd || !dx)
    return set_error(T2O_EINVAL, "bn_relu_nhwc_bwd: null pointer or bad shape (C must be a power of two in [4, 1024])");
  if (has_res && relu && !y) return set_error(T2O_EINVAL, "bn_relu_nhwc_bwd: y is needed when a residual was added");
  if (!workspace || workspace_bytes < t2o_bn_nhwc_workspace_bytes(M, C)) return set_error(T2O_EWORKSPACE, "bn_relu_nhwc_bwd: workspace too small");
  BnArgs a = {};
  a.x = x; a.y = y; a.dy = dy; a.out = dx; a.dres = dres; a.weight = weight; a.bias = bias;
  a.save_mean = const_cast<float*>(save_mean); a.save_invstd = const_cast<float*>(save_invstd);
  a.dweight = dweight; a.dbias = dbias;
  a.N = M; a.C = C; a.HW = 1; a.splits = 1; a.relu = relu; a.acc = accumulate ? 1 : 0;
  a.partials = (double*)workspace;
  a.coef = (float*)((char*)workspace + sizeof(double) * 2 * (size_t)C);
  float* partial = a.coef + 4 * (size_t)C;
  hipStream_t st = (hipStream_t)stream;
  int rpb, nblk;
  nhwc_partition(M, C, 2, &rpb, &nblk);
  if (has_res) k_bn_nhwc_bwd_sums<true><<<nblk, kThreads, 0, st>>>(a, M, rpb, partial); else k_bn_nhwc_bwd_sums<false><<<nblk, kThreads, 0, st>>>(a, M, rpb, partial);
  k_bn_nhwc_finalize<true><<<C / 4, kThreads, 0, st>>>(a, partial, nblk);
  const size_t total4 = (size_t)M * (C >> 2);
  const unsigned grid = flat_grid(total4, 1, 2);
  if (has_res) k_bn_nhwc_bwd_apply<true><<<grid, kThreads, 0, st>>>(a, total4); else k_bn_nhwc_bwd_apply<false><<<grid, kThreads, 0, st>>>(a, total4);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

// The same backward (no residual) when the producer of dy already left the sums: partial (partial_rows, 2, C) = per row block
// the channels' sums of the gated gradient and of gated gradient * xhat (t2o_conv3x3_dgrad_pre_bnsums_nhwc's epilogue) -- the
// sums pass over dy and x is not launched.
int t2o_bn_relu_nhwc_bwd_partials_acc(const float* x, const float* dy, const float* weight, const float* bias,
                                      const float* save_mean, const float* save_invstd, float* dx, float* dweight, float* dbias,
                                      int relu, int accumulate, const float* partial, int partial_rows, void* workspace,
                                      size_t workspace_bytes, int M, int C, void* stream) {
  if (!x || M <= 0 || !nhwc_channels_ok(C) || !dy || !weight || !bias || !save_mean || !save_invstd || !dx)
    return set_error(T2O_EINVAL, "bn_relu_nhwc_bwd_partials: null pointer or bad shape (C must be a power of two in [4, 1024])");
  if (!partial || partial_rows <= 0) return set_error(T2O_EINVAL, "bn_relu_nhwc_bwd_partials: no partial sums");
  if (!workspace || workspace_bytes < t2o_bn_nhwc_workspace_bytes(M, C)) return set_error(T2O_EWORKSPACE, "bn_relu_nhwc_bwd_partials: workspace too small");
  BnArgs a = {};
  a.x = x; a.dy = dy; a.out = dx; a.weight = weight; a.bias = bias;
  a.save_mean = const_cast<float*>(save_mean); a.save_invstd = const_cast<float*>(save_invstd);
  a.dweight = dweight; a.dbias = dbias;
  a.N = M; a.C = C; a.HW = 1; a.splits = 1; a.relu = relu; a.acc = accumulate ? 1 : 0;
  a.partials = (double*)workspace;
  a.coef = (float*)((char*)workspace + sizeof(double) * 2 * (size_t)C);
  hipStream_t st = (hipStream_t)stream;
  k_bn_nhwc_finalize<true><<<C / 4, kThreads, 0, st>>>(a, partial, partial_rows);
  const size_t total4 = (size_t)M * (C >> 2);
  k_bn_nhwc_bwd_apply<false><<<flat_grid(total4, 1, 2), kThreads, 0, st>>>(a, total4);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

int t2o_bn_relu_nhwc_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* bias,
                         const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dweight,
                         float* dbias, int has_res, int relu, void* workspace, size_t workspace_bytes, int M, int C,
                         void* stream) {
  return t2o_bn_relu_nhwc_bwd_acc(x, y, dy, weight, bias, save_mean, save_invstd, dx, dres, dweight, dbias, has_res, relu, 0,
                                  workspace, workspace_bytes, M, C, stream);
}

size_t t2o_bn_dual_nhwc_workspace_bytes(int M, int C) {
  (void)M;
  // [coef main (4, C)][coef shortcut (4, C)][partials: shortcut statistics (2, C) rows | backward (3, C) rows]
  return sizeof(float) * 8 * (size_t)C + sizeof(float) * 3 * (size_t)C * kNhwcMaxBlocks + sizeof(float) * 2 * (size_t)C * kNhwcMaxBlocks;
}

int t2o_bn_dual_relu_nhwc_fwd(const float* x, const float* partial, int partial_rows, const float* xs,
                              const float* weight, const float* bias, float* running_mean, float* running_var, float* save_mean,
                              float* save_invstd, const float* weight_s, const float* bias_s, float* running_mean_s,
                              float* running_var_s, float* save_mean_s, float* save_invstd_s, float* out, float momentum, float eps,
                              float momentum_s, float eps_s, void* workspace, size_t workspace_bytes, int M, int C, void* stream) {
  if (!x || !xs || M <= 0 || !nhwc_channels_ok(C) || !weight || !bias || !weight_s || !bias_s || !save_mean || !save_invstd || !save_mean_s ||
      !save_invstd_s || !out)
    return set_error(T2O_EINVAL, "bn_dual_relu_nhwc_fwd: null pointer or bad shape (C must be a power of two in [4, 1024])");
  if ((running_mean == nullptr) != (running_var == nullptr) || (running_mean_s == nullptr) != (running_var_s == nullptr))
    return set_error(T2O_EINVAL, "bn_dual_relu_nhwc_fwd: running_mean and running_var must both be given or both be null");
  if (partial && partial_rows <= 0) return set_error(T2O_EINVAL, "bn_dual_relu_nhwc_fwd: partial_rows");
  if (!workspace || workspace_bytes < t2o_bn_dual_nhwc_workspace_bytes(M, C)) return set_error(T2O_EWORKSPACE, "bn_dual_relu_nhwc_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* coef0 = (float*)workspace;
  float* coef1 = coef0 + 4 * (size_t)C;
  float* part_s = coef1 + 4 * (size_t)C;                        // (rows, 2, C) statistics of the shortcut branch
  float* part_m = part_s + 2 * (size_t)C * kNhwcMaxBlocks;      // (rows, 2, C) of the main branch when its producer left none
  BnArgs a = {}, b = {};
  a.x = x; a.out = out; a.weight = weight; a.bias = bias; a.running_mean = running_mean; a.running_var = running_var;
  a.save_mean = save_mean; a.save_invstd = save_invstd; a.N = M; a.C = C; a.HW = 1; a.splits = 1; a.eps = eps; a.momentum = momentum;
  a.relu = 1; a.coef = coef0;
  b.x = xs; b.weight = weight_s; b.bias = bias_s; b.running_mean = running_mean_s; b.running_var = running_var_s;
  b.save_mean = save_mean_s; b.save_invstd = save_invstd_s; b.N = M; b.C = C; b.HW = 1; b.splits = 1; b.eps = eps_s; b.momentum = momentum_s;
  b.relu = 0; b.coef = coef1;
  int rpb, nblk;
  nhwc_partition(M, C, kUnroll, &rpb, &nblk);
  k_bn_nhwc_stats<<<nblk, kThreads, 0, st>>>(b, M, rpb, part_s);
  if (!partial) {
    k_bn_nhwc_stats<<<nblk, kThreads, 0, st>>>(a, M, rpb, part_m);
    partial = part_m;
    partial_rows = nblk;
  }
  k_bn_nhwc_finalize_pair<false><<<2 * (C / 4), kThreads, 0, st>>>(a, partial, partial_rows, 2 * C, C, b, part_s, nblk, 2 * C, C);
  BnDual d = {};
  d.xs = xs; d.coef_s = coef1;
  const size_t total4 = (size_t)M * (C >> 2);
  k_bn_nhwc_apply_dual<<<flat_grid(total4, 1, kUnroll), kThreads, 0, st>>>(a, d, total4);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

int t2o_bn_dual_relu_nhwc_bwd_acc(const float* x, const float* xs, const float* y, const float* dy, const float* weight,
                                  const float* bias, const float* save_mean, const float* save_invstd, const float* weight_s,
                                  const float* bias_s, const float* save_mean_s, const float* save_invstd_s, float* dx, float* dxs,
                                  float* dweight, float* dbias, float* dweight_s, float* dbias_s, int accumulate, void* workspace,
                                  size_t workspace_bytes, int M, int C, void* stream) {
  if (!x || !xs || !y || !dy || M <= 0 || !nhwc_channels_ok(C) || !weight || !bias || !weight_s || !bias_s || !save_mean || !save_invstd ||
      !save_mean_s || !save_invstd_s || !dx || !dxs)
    return set_error(T2O_EINVAL, "bn_dual_relu_nhwc_bwd: null pointer or bad shape (C must be a power of two in [4, 1024])");
  if (!workspace || workspace_bytes < t2o_bn_dual_nhwc_workspace_bytes(M, C)) return set_error(T2O_EWORKSPACE, "bn_dual_relu_nhwc_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* coef0 = (float*)workspace;
  float* coef1 = coef0 + 4 * (size_t)C;
  float* partial = coef1 + 4 * (size_t)C;                       // (rows, 3, C)
  BnArgs a = {}, b = {};
  a.x = x; a.y = y; a.dy = dy; a.out = dx; a.weight = weight; a.bias = bias;
  a.save_mean = const_cast<float*>(save_mean); a.save_invstd = const_cast<float*>(save_invstd);
  a.dweight = dweight; a.dbias = dbias; a.N = M; a.C = C; a.HW = 1; a.splits = 1; a.relu = 1; a.acc = accumulate ? 1 : 0; a.coef = coef0;
  b.x = xs; b.weight = weight_s; b.bias = bias_s;
  b.save_mean = const_cast<float*>(save_mean_s); b.save_invstd = const_cast<float*>(save_invstd_s);
  b.dweight = dweight_s; b.dbias = dbias_s; b.N = M; b.C = C; b.HW = 1; b.splits = 1; b.relu = 0; b.acc = accumulate ? 1 : 0; b.coef = coef1;
  BnDual d = {};
  d.xs = xs; d.coef_s = coef1; d.mean_s = save_mean_s; d.invstd_s = save_invstd_s; d.out_s = dxs;
  int rpb, nblk;
  nhwc_partition(M, C, 2, &rpb, &nblk);
  k_bn_nhwc_bwd_sums_dual<<<nblk, kThreads, 0, st>>>(a, d, M, rpb, partial);
  k_bn_nhwc_finalize_pair<true><<<2 * (C / 4), kThreads, 0, st>>>(a, partial, nblk, 3 * C, C, b, partial, nblk, 3 * C, 2 * C);
  const size_t total4 = (size_t)M * (C >> 2);
  k_bn_nhwc_bwd_apply_dual<<<flat_grid(total4, 1, 2), kThreads, 0, st>>>(a, d, total4);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "batch-norm kernel launch failed");
}

}  // extern "C"
