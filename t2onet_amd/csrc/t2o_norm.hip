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

__global__ __launch_bounds__(kThreads) void k_bn_finalize(BnArgs a) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= a.C) return;
  double sum = 0.0, sq = 0.0;
  for (int s = 0; s < a.splits; ++s) { sum += a.partials[((size_t)c * a.splits + s) * 2]; sq += a.partials[((size_t)c * a.splits + s) * 2 + 1]; }
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

__global__ __launch_bounds__(kThreads) void k_bn_bwd_finalize(BnArgs a) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= a.C) return;
  double sg = 0.0, sgx = 0.0;
  for (int s = 0; s < a.splits; ++s) { sg += a.partials[((size_t)c * a.splits + s) * 2]; sgx += a.partials[((size_t)c * a.splits + s) * 2 + 1]; }
  const double m = (double)a.N * a.HW;
  if (a.dbias) a.dbias[c] = (float)sg;
  if (a.dweight) a.dweight[c] = (float)sgx;
  a.coef[c] = a.weight[c] * a.save_invstd[c];     // a_c
  a.coef[a.C + c] = (float)(sg / m);             // mean of g
  a.coef[2 * a.C + c] = (float)(sgx / m);        // mean of g * xhat
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

}  // extern "C"
