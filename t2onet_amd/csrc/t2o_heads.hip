// t2o_heads.hip -- the operators' parameter heads for a batch whose samples use DIFFERENT operators
// (models/operators.py:73-88 extract_parameters = op_param_regressor(fc2(LeakyReLU(fc1(features)))), called per
// operator group by models/actor.py:244-255).  The reference runs 2 small GEMMs + ~5 elementwise kernels per group
// (and as many again, twice, in the backward); the batched PyTorch form of that costs ~150 launches per decoder step.
// Here: one forward launch (one workgroup per sample evaluates ITS operator's head) and two backward launches.
//
//   hidden_b = lrelu(W1[op_b] ctx_b + b1[op_b])      W1 (512,512), slope 0.01
//   raw_b    = W2[op_b] hidden_b + b2[op_b]          W2 (n_op, 512), n_op in {1, 8, 24}
//   param_b  = regressor_{op_b}(raw_b), zero-padded to 24 columns
// Dot products: a wave per output row, lanes stride the 512 columns with 16-byte loads, shuffle reduction (fixed
// order).  Weight gradients: one workgroup per (operator, 64x64 tile) sums the outer products of that operator's
// samples in batch order -- deterministic, no atomics; heads no sample selected get exact zeros.
#include <hip/hip_runtime.h>

#include <cmath>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

constexpr int kD = 512;             // feature width = operator_fc_dim (2 * hidden_size)
constexpr int kOps = 8;
constexpr int kPad = 24;
constexpr int kHT = 256;

struct HeadArgs {
  const float* w1[kOps];   // (512,512) row = output unit
  const float* b1[kOps];   // (512)
  const float* w2[kOps];   // (n,512)
  const float* b2[kOps];   // (n)
  float* gw1[kOps];
  float* gb1[kOps];
  float* gw2[kOps];
  float* gb2[kOps];
  const int* op_id;        // (B) executor index, < 0 or 4: no head (zeros)
  const float* ctx;        // (B,512)
  float* hidden;           // (B,512) saved for the backward
  float* raw;              // (B,24)
  float* param;            // (B,24)
  const float* gparam;     // (B,24)
  float* dpre;             // (B,512) backward scratch: gradient w.r.t. fc1's pre-activation
  float* gctx;             // (B,512)
  int B;
  float brightness_range, sat_lo, sat_hi, sharpness_range;
};

__device__ __forceinline__ int n_params(int op) { return op == 3 ? 24 : op == 5 ? 8 : 1; }
__device__ __forceinline__ bool has_head(int op) { return op >= 0 && op < kOps && op != 4; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dot(row (512 floats in global memory), vec (512 floats in LDS)) by one wave
__device__ __forceinline__ float row_dot(const float* row, const float* vec, int lane) {
  const float4 a0 = *reinterpret_cast<const float4*>(row + 4 * lane), a1 = *reinterpret_cast<const float4*>(row + 256 + 4 * lane);
  const float4 v0 = *reinterpret_cast<const float4*>(vec + 4 * lane), v1 = *reinterpret_cast<const float4*>(vec + 256 + 4 * lane);
  float s = (a0.x * v0.x + a0.y * v0.y) + (a0.z * v0.z + a0.w * v0.w);
  s += (a1.x * v1.x + a1.y * v1.y) + (a1.z * v1.z + a1.w * v1.w);
  return wave_sum(s);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// op_param_regressor of each operator (models/operators.py:266-269, :232, :464, :343, :501; curves: identity)
__device__ __forceinline__ float regress(const HeadArgs& a, int op, float f) {
  switch (op) {
    case 0: return (tanhf(f) * 0.5f + 0.5f) * (2.0f * a.brightness_range) + (-a.brightness_range);   // tanh_range(-r, r, initial=0): bias 0
    case 1: return tanhf(f);
    case 2: return tanhf(fmaxf(f, 0.0f)) * a.sat_hi + tanhf(fmaxf(-f, 0.0f)) * a.sat_lo;
    case 6: return sigmoidf_(f) * a.sharpness_range;
    case 7: return sigmoidf_(f);
    default: return f;
  }
}
__device__ __forceinline__ float regress_grad(const HeadArgs& a, int op, float f) {
  switch (op) {
    case 0: { const float t = tanhf(f); return (1.0f - t * t) * 0.5f * (2.0f * a.brightness_range); }
    case 1: { const float t = tanhf(f); return 1.0f - t * t; }
    case 2: {
      if (f > 0.0f) { const float t = tanhf(f); return (1.0f - t * t) * a.sat_hi; }
      if (f < 0.0f) { const float t = tanhf(-f); return -(1.0f - t * t) * a.sat_lo; }
      return 0.0f;                                                      // relu'(0) = 0 on both branches
    }
    case 6: { const float s = sigmoidf_(f); return s * (1.0f - s) * a.sharpness_range; }
    case 7: { const float s = sigmoidf_(f); return s * (1.0f - s); }
    default: return 1.0f;
  }
}

__global__ __launch_bounds__(kHT) void k_heads_fwd(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float ctx[kD];
  __shared__ __attribute__((aligned(16))) float hid[kD];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int op = a.op_id[b];
  if (!has_head(op)) {
    if (tid < kPad) { a.param[(size_t)b * kPad + tid] = 0.0f; a.raw[(size_t)b * kPad + tid] = 0.0f; }
    for (int i = tid; i < kD; i += kHT) a.hidden[(size_t)b * kD + i] = 0.0f;
    return;
  }
  for (int i = tid; i < kD; i += kHT) ctx[i] = a.ctx[(size_t)b * kD + i];
  __syncthreads();
  const float* w1 = a.w1[op];
  for (int j = wave; j < kD; j += kHT / 64) {
    const float s = row_dot(w1 + (size_t)j * kD, ctx, lane) + a.b1[op][j];
    if (lane == 0) hid[j] = s > 0.0f ? s : 0.01f * s;
  }
  __syncthreads();
  for (int i = tid; i < kD; i += kHT) a.hidden[(size_t)b * kD + i] = hid[i];
  const int n = n_params(op);
  for (int r = wave; r < kPad; r += kHT / 64) {
    float f = 0.0f, p = 0.0f;
    if (r < n) {
      f = row_dot(a.w2[op] + (size_t)r * kD, hid, lane) + a.b2[op][r];
      p = regress(a, op, f);
    }
    if (lane == 0) { a.raw[(size_t)b * kPad + r] = f; a.param[(size_t)b * kPad + r] = p; }
  }
}

// per sample: dpre_b (gradient at fc1's pre-activation) and gctx_b = W1^T dpre_b
__global__ __launch_bounds__(kHT) void k_heads_bwd_sample(HeadArgs a) {
  __shared__ float df[kPad];
  __shared__ __attribute__((aligned(16))) float dp[kD];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int op = a.op_id[b];
  if (!has_head(op)) {
    for (int i = tid; i < kD; i += kHT) { a.dpre[(size_t)b * kD + i] = 0.0f; a.gctx[(size_t)b * kD + i] = 0.0f; }
    return;
  }
  const int n = n_params(op);
  if (tid < kPad) df[tid] = tid < n ? a.gparam[(size_t)b * kPad + tid] * regress_grad(a, op, a.raw[(size_t)b * kPad + tid]) : 0.0f;
  __syncthreads();
  for (int i = tid; i < kD; i += kHT) {                        // dh_i = sum_r W2[r][i] df_r ; LeakyReLU' from the output's sign
    float s = 0.0f;
    for (int r = 0; r < n; ++r) s += a.w2[op][(size_t)r * kD + i] * df[r];
    const float h = a.hidden[(size_t)b * kD + i];
    const float d = h > 0.0f ? s : 0.01f * s;
    dp[i] = d;
    a.dpre[(size_t)b * kD + i] = d;
  }
  __syncthreads();
  const float* w1 = a.w1[op];
  for (int i = tid; i < kD; i += kHT) {                        // gctx_i = sum_j W1[j][i] dpre_j (threads walk a row: coalesced)
    float s = 0.0f;
#pragma unroll 8
    for (int j = 0; j < kD; ++j) s += w1[(size_t)j * kD + i] * dp[j];
    a.gctx[(size_t)b * kD + i] = s;
  }
}

// weight gradients.  blockIdx.y = operator; blockIdx.x < 64: a 64x64 tile of gW1 (+ gb1 on the first tile column);
// blockIdx.x == 64: gW2 and gb2.  Samples are visited in batch order: deterministic sums.
__global__ __launch_bounds__(kHT) void k_heads_bwd_weights(HeadArgs a) {
  const int op = blockIdx.y, tid = threadIdx.x;
  if (op == 4) return;
  if (blockIdx.x < 64) {
    const int tr = (blockIdx.x / 8) * 64, tc = (blockIdx.x % 8) * 64;     // rows = output units (dpre), cols = inputs (ctx)
    const int r0 = tr + (tid / 16) * 4, c0 = tc + (tid % 16) * 4;
    float acc[4][4] = {};
    float accb[4] = {};
    for (int b = 0; b < a.B; ++b) {
      if (a.op_id[b] != op) continue;                                      // uniform branch
      const float4 d = *reinterpret_cast<const float4*>(a.dpre + (size_t)b * kD + r0);
      const float4 c = *reinterpret_cast<const float4*>(a.ctx + (size_t)b * kD + c0);
      const float dv[4] = {d.x, d.y, d.z, d.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        accb[i] += dv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += dv[i] * cv[j];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(a.gw1[op] + (size_t)(r0 + i) * kD + c0) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    if (tc == 0 && tid % 16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a.gb1[op][r0 + i] = accb[i];
    }
  } else {
    const int n = n_params(op);
    for (int e = tid; e < n * kD; e += kHT) {
      const int r = e / kD, i = e % kD;
      float s = 0.0f;
      for (int b = 0; b < a.B; ++b) {
        if (a.op_id[b] != op) continue;
        const float f = a.raw[(size_t)b * kPad + r];
        s += a.gparam[(size_t)b * kPad + r] * regress_grad(a, op, f) * a.hidden[(size_t)b * kD + i];
      }
      a.gw2[op][e] = s;
    }
    if (tid < n) {
      float s = 0.0f;
      for (int b = 0; b < a.B; ++b) {
        if (a.op_id[b] != op) continue;
        s += a.gparam[(size_t)b * kPad + tid] * regress_grad(a, op, a.raw[(size_t)b * kPad + tid]);
      }
      a.gb2[op][tid] = s;
    }
  }
}

bool fill(HeadArgs& a, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2) {
  for (int k = 0; k < kOps; ++k) {
    a.w1[k] = w1[k]; a.b1[k] = b1[k]; a.w2[k] = w2[k]; a.b2[k] = b2[k];
    if (k != 4 && (!w1[k] || !b1[k] || !w2[k] || !b2[k])) return false;
  }
  return true;
}

}  // namespace

extern "C" {

int t2o_param_heads_fwd(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, float* hidden, float* raw, float* param,
                        float brightness_range, float sat_lo, float sat_hi, float sharpness_range, int B, int D, void* stream) {
  if (!op_id || !ctx || !w1 || !b1 || !w2 || !b2 || !hidden || !raw || !param) return set_error(T2O_EINVAL, "param_heads_fwd: null pointer");
  if (B <= 0 || D != kD) return set_error(T2O_EINVAL, "param_heads_fwd: B must be positive and the feature width 512");
  HeadArgs a = {};
  if (!fill(a, w1, b1, w2, b2)) return set_error(T2O_EINVAL, "param_heads_fwd: a head's weight pointer is null");
  a.op_id = op_id; a.ctx = ctx; a.hidden = hidden; a.raw = raw; a.param = param; a.B = B;
  a.brightness_range = brightness_range; a.sat_lo = sat_lo; a.sat_hi = sat_hi; a.sharpness_range = sharpness_range;
  k_heads_fwd<<<B, kHT, 0, (hipStream_t)stream>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "param_heads_fwd launch failed");
}

int t2o_param_heads_bwd(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, const float* hidden, const float* raw,
                        const float* gparam, float* gctx, float* dpre, float* const* gw1, float* const* gb1,
                        float* const* gw2, float* const* gb2, float brightness_range, float sat_lo, float sat_hi,
                        float sharpness_range, int B, int D, void* stream) {
  if (!op_id || !ctx || !w1 || !b1 || !w2 || !b2 || !hidden || !raw || !gparam || !gctx || !dpre || !gw1 || !gb1 || !gw2 || !gb2)
    return set_error(T2O_EINVAL, "param_heads_bwd: null pointer");
  if (B <= 0 || D != kD) return set_error(T2O_EINVAL, "param_heads_bwd: B must be positive and the feature width 512");
  HeadArgs a = {};
  if (!fill(a, w1, b1, w2, b2)) return set_error(T2O_EINVAL, "param_heads_bwd: a head's weight pointer is null");
  for (int k = 0; k < kOps; ++k) {
    a.gw1[k] = gw1[k]; a.gb1[k] = gb1[k]; a.gw2[k] = gw2[k]; a.gb2[k] = gb2[k];
    if (k != 4 && (!gw1[k] || !gb1[k] || !gw2[k] || !gb2[k])) return set_error(T2O_EINVAL, "param_heads_bwd: a gradient pointer is null");
  }
  a.op_id = op_id; a.ctx = ctx; a.hidden = const_cast<float*>(hidden); a.raw = const_cast<float*>(raw);
  a.gparam = gparam; a.gctx = gctx; a.dpre = dpre; a.B = B;
  a.brightness_range = brightness_range; a.sat_lo = sat_lo; a.sat_hi = sat_hi; a.sharpness_range = sharpness_range;
  hipStream_t st = (hipStream_t)stream;
  k_heads_bwd_sample<<<B, kHT, 0, st>>>(a);
  k_heads_bwd_weights<<<dim3(65, kOps), kHT, 0, st>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "param_heads_bwd launch failed");
}

}  // extern "C"
