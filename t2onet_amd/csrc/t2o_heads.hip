// t2o_heads.hip -- the operators' parameter heads for a batch whose samples use DIFFERENT operators
// (models/operators.py:73-88 extract_parameters = op_param_regressor(fc2(LeakyReLU(fc1(features)))), called per
// operator group by models/actor.py:244-255).  The reference runs 2 small GEMMs + ~5 elementwise kernels per group
// (and as many again, twice, in the backward); the batched PyTorch form of that costs ~150 launches per decoder step.
// Here: two forward launches (each sample evaluates ITS operator's head: fc1 dealt over 8 workgroups per sample, then
// fc2 + regressor) and two backward launches.
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
  int accumulate;          // backward: weight gradients are ADDED to gw1 / gb1 / gw2 / gb2 (each element has one writer)
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

constexpr int kSlices = 8;          // one sample's fc1 rows (forward) / W1 columns (backward) are dealt to this many workgroups
constexpr int kSliceW = kD / kSlices;

// fc1 + LeakyReLU.  blockIdx.x = sample, blockIdx.y = 64-row slice of its operator's W1: 8 x B workgroups stream
// the 1 MB weight of each selected head instead of B (64 workgroups leave three quarters of the chip idle).
__global__ __launch_bounds__(kHT) void k_heads_fc1(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float ctx[kD];
  const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int op = a.op_id[b];
  float* hid = a.hidden + (size_t)b * kD + sl * kSliceW;
  if (!has_head(op)) {
    if (tid < kSliceW) hid[tid] = 0.0f;
    if (sl == 0 && tid < kPad) { a.param[(size_t)b * kPad + tid] = 0.0f; a.raw[(size_t)b * kPad + tid] = 0.0f; }
    return;
  }
  for (int i = tid; i < kD; i += kHT) ctx[i] = a.ctx[(size_t)b * kD + i];
  __syncthreads();
  constexpr int kRows = kSliceW / (kHT / 64);                  // rows per wave
  const int j0 = sl * kSliceW + wave * kRows;
  const float* w1 = a.w1[op] + (size_t)j0 * kD;
  // a wave's 16 rows: all 32 weight loads and the 16 bias loads in flight before the first reduction (one memory round trip,
  // not four); the arithmetic of every row is row_dot's
  const float4 v0 = *reinterpret_cast<const float4*>(ctx + 4 * lane), v1 = *reinterpret_cast<const float4*>(ctx + 256 + 4 * lane);
  float4 a0[kRows], a1[kRows];
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    a0[k] = *reinterpret_cast<const float4*>(w1 + (size_t)k * kD + 4 * lane);
    a1[k] = *reinterpret_cast<const float4*>(w1 + (size_t)k * kD + 256 + 4 * lane);
  }
  const float bias = lane < kRows ? a.b1[op][j0 + lane] : 0.0f;
  float mine = 0.0f;
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    float s = (a0[k].x * v0.x + a0[k].y * v0.y) + (a0[k].z * v0.z + a0[k].w * v0.w);
    s += (a1[k].x * v1.x + a1[k].y * v1.y) + (a1[k].z * v1.z + a1[k].w * v1.w);
    s = wave_sum(s);
    if (lane == k) mine = s;
  }
  if (lane < kRows) {
    const float s = mine + bias;
    hid[wave * kRows + lane] = s > 0.0f ? s : 0.01f * s;
  }
}

// fc2 + the operator's parameter regressor.  One workgroup per sample.
__global__ __launch_bounds__(kHT) void k_heads_fc2(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float hid[kD];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int op = a.op_id[b];
  if (!has_head(op)) return;                                   // (k_heads_fc1 wrote the zeros)
  for (int i = tid; i < kD; i += kHT) hid[i] = a.hidden[(size_t)b * kD + i];
  __syncthreads();
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

// per sample: dpre_b (gradient at fc1's pre-activation) and gctx_b = W1^T dpre_b.  blockIdx.x = sample, blockIdx.y =
// 64-column slice of gctx; every slice recomputes the sample's dpre (512 x n multiply-adds) rather than wait for it.
__global__ __launch_bounds__(kHT) void k_heads_bwd_sample(HeadArgs a) {
  __shared__ float df[kPad];
  __shared__ __attribute__((aligned(16))) float dp[kD];
  __shared__ __attribute__((aligned(16))) float red[16][kSliceW];
  const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const int op = a.op_id[b];
  const size_t slice = (size_t)b * kD + sl * kSliceW;
  if (!has_head(op)) {
    if (tid < kSliceW) { a.dpre[slice + tid] = 0.0f; a.gctx[slice + tid] = 0.0f; }
    return;
  }
  const int n = n_params(op);
  if (tid < kPad) df[tid] = tid < n ? a.gparam[(size_t)b * kPad + tid] * regress_grad(a, op, a.raw[(size_t)b * kPad + tid]) : 0.0f;
  __syncthreads();
  for (int i = tid; i < kD; i += kHT) {                        // dh_i = sum_r W2[r][i] df_r ; LeakyReLU' from the output's sign
    float s = 0.0f;
    for (int r = 0; r < n; ++r) s += a.w2[op][(size_t)r * kD + i] * df[r];
    const float h = a.hidden[(size_t)b * kD + i];
    dp[i] = h > 0.0f ? s : 0.01f * s;
  }
  __syncthreads();
  if (tid < kSliceW) a.dpre[slice + tid] = dp[sl * kSliceW + tid];
  // gctx_i = sum_j W1[j][i] dpre_j over this slice's columns: 16 lanes x float4 span the 64 columns, the 16 lane
  // groups take 32 rows each, partial sums meet in LDS in a fixed order
  const int cq = tid & 15, rp = tid >> 4;
  const float* w1 = a.w1[op] + (size_t)(rp * 32) * kD + sl * kSliceW + 4 * cq;
  float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 8
  for (int j = 0; j < 32; ++j) {
    const float4 w = *reinterpret_cast<const float4*>(w1 + (size_t)j * kD);
    const float d = dp[rp * 32 + j];
    s.x += w.x * d; s.y += w.y * d; s.z += w.z * d; s.w += w.w * d;
  }
  *reinterpret_cast<float4*>(&red[rp][4 * cq]) = s;
  __syncthreads();
  if (tid < kSliceW) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tid];
    a.gctx[slice + tid] = t;
  }
}

// weight gradients.  blockIdx.y = operator; blockIdx.x < 64: a 64x64 tile of gW1 (+ gb1 on the first tile column);
// blockIdx.x >= 64: a 64-column slice of gW2 (+ gb2 on the first).  Samples are visited in batch order:
// deterministic sums.  The operator's samples are found 64 at a time with one ballot.
__global__ __launch_bounds__(kHT) void k_heads_bwd_weights(HeadArgs a) {
  const int op = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  if (op == 4) return;
  if (blockIdx.x < 64) {
    const int tr = (blockIdx.x / 8) * 64, tc = (blockIdx.x % 8) * 64;     // rows = output units (dpre), cols = inputs (ctx)
    const int r0 = tr + (tid / 16) * 4, c0 = tc + (tid % 16) * 4;
    float acc[4][4] = {};
    float accb[4] = {};
    for (int b0 = 0; b0 < a.B; b0 += 64) {
      unsigned long long m = __ballot(b0 + lane < a.B && a.op_id[min(b0 + lane, a.B - 1)] == op);
      while (m) {                                                          // uniform
        const int b = b0 + __ffsll((long long)m) - 1;
        m &= m - 1;
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
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4* dst = reinterpret_cast<float4*>(a.gw1[op] + (size_t)(r0 + i) * kD + c0);
      float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      if (a.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *dst = v;
    }
    if (tc == 0 && tid % 16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a.gb1[op][r0 + i] = accb[i] + (a.accumulate ? a.gb1[op][r0 + i] : 0.0f);
    }
  } else {
    __shared__ float df[64][kPad];                                         // gparam * regressor' of 64 samples (0: not this operator's)
    const int sl = blockIdx.x - 64, n = n_params(op);
    const int i = sl * kSliceW + lane, ph = tid >> 6;                      // thread: column i, rows ph, ph + 4, ...
    float acc[kPad / 4] = {};
    float accb = 0.0f;
    for (int b0 = 0; b0 < a.B; b0 += 64) {
      const unsigned long long m0 = __ballot(b0 + lane < a.B && a.op_id[min(b0 + lane, a.B - 1)] == op);
      if (m0 == 0) continue;                                               // uniform
      __syncthreads();
      for (int e = tid; e < 64 * kPad; e += kHT) {
        const int bl = e / kPad, r = e % kPad, b = b0 + bl;
        float v = 0.0f;
        if (((m0 >> bl) & 1) && r < n) v = a.gparam[(size_t)b * kPad + r] * regress_grad(a, op, a.raw[(size_t)b * kPad + r]);
        df[bl][r] = v;
      }
      __syncthreads();
      unsigned long long m = m0;
      while (m) {
        const int bl = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float h = a.hidden[(size_t)(b0 + bl) * kD + i];
#pragma unroll
        for (int k = 0; k < kPad / 4; ++k) acc[k] += df[bl][ph + 4 * k] * h;
        if (sl == 0 && tid < kPad) accb += df[bl][tid];
      }
    }
#pragma unroll
    for (int k = 0; k < kPad / 4; ++k)
      if (ph + 4 * k < n) {
        float* dst = a.gw2[op] + (size_t)(ph + 4 * k) * kD + i;
        *dst = acc[k] + (a.accumulate ? *dst : 0.0f);
      }
    if (sl == 0 && tid < n) a.gb2[op][tid] = accb + (a.accumulate ? a.gb2[op][tid] : 0.0f);
  }
}

bool fill(HeadArgs& a, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2) {
  for (int k = 0; k < kOps; ++k) {
    a.w1[k] = w1[k]; a.b1[k] = b1[k]; a.w2[k] = w2[k]; a.b2[k] = b2[k];
    if (k != 4 && (!w1[k] || !b1[k] || !w2[k] || !b2[k])) return false;
    // the weight matrices are read with 16-byte accesses (row_dot: 512-float rows)
    if (k != 4 && ((reinterpret_cast<size_t>(w1[k]) | reinterpret_cast<size_t>(w2[k])) & 15) != 0) return false;
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
  if (!fill(a, w1, b1, w2, b2)) return set_error(T2O_EINVAL, "param_heads_fwd: a head's weight pointer is null or its matrix not 16-byte aligned");
  a.op_id = op_id; a.ctx = ctx; a.hidden = hidden; a.raw = raw; a.param = param; a.B = B;
  a.brightness_range = brightness_range; a.sat_lo = sat_lo; a.sat_hi = sat_hi; a.sharpness_range = sharpness_range;
  k_heads_fc1<<<dim3(B, kSlices), kHT, 0, (hipStream_t)stream>>>(a);
  k_heads_fc2<<<B, kHT, 0, (hipStream_t)stream>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "param_heads_fwd launch failed");
}

int t2o_param_heads_bwd_acc(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, const float* hidden, const float* raw,
                        const float* gparam, float* gctx, float* dpre, float* const* gw1, float* const* gb1,
                        float* const* gw2, float* const* gb2, float brightness_range, float sat_lo, float sat_hi,
                        float sharpness_range, int B, int D, int accumulate, void* stream) {
  if (!op_id || !ctx || !w1 || !b1 || !w2 || !b2 || !hidden || !raw || !gparam || !gctx || !dpre || !gw1 || !gb1 || !gw2 || !gb2)
    return set_error(T2O_EINVAL, "param_heads_bwd: null pointer");
  if (B <= 0 || D != kD) return set_error(T2O_EINVAL, "param_heads_bwd: B must be positive and the feature width 512");
  HeadArgs a = {};
  if (!fill(a, w1, b1, w2, b2)) return set_error(T2O_EINVAL, "param_heads_bwd: a head's weight pointer is null or its matrix not 16-byte aligned");
  for (int k = 0; k < kOps; ++k) {
    a.gw1[k] = gw1[k]; a.gb1[k] = gb1[k]; a.gw2[k] = gw2[k]; a.gb2[k] = gb2[k];
    if (k != 4 && (!gw1[k] || !gb1[k] || !gw2[k] || !gb2[k])) return set_error(T2O_EINVAL, "param_heads_bwd: a gradient pointer is null");
    if (k != 4 && ((reinterpret_cast<size_t>(gw1[k]) | reinterpret_cast<size_t>(gw2[k])) & 15) != 0)
      return set_error(T2O_EINVAL, "param_heads_bwd: a weight-gradient matrix is not 16-byte aligned");
  }
  a.op_id = op_id; a.ctx = ctx; a.hidden = const_cast<float*>(hidden); a.raw = const_cast<float*>(raw);
  a.gparam = gparam; a.gctx = gctx; a.dpre = dpre; a.B = B; a.accumulate = accumulate ? 1 : 0;
  a.brightness_range = brightness_range; a.sat_lo = sat_lo; a.sat_hi = sat_hi; a.sharpness_range = sharpness_range;
  hipStream_t st = (hipStream_t)stream;
  k_heads_bwd_sample<<<dim3(B, kSlices), kHT, 0, st>>>(a);
  k_heads_bwd_weights<<<dim3(64 + kSlices, kOps), kHT, 0, st>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "param_heads_bwd launch failed");
}

int t2o_param_heads_bwd(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, const float* hidden, const float* raw,
                        const float* gparam, float* gctx, float* dpre, float* const* gw1, float* const* gb1,
                        float* const* gw2, float* const* gb2, float brightness_range, float sat_lo, float sat_hi,
                        float sharpness_range, int B, int D, void* stream) {
  return t2o_param_heads_bwd_acc(op_id, ctx, w1, b1, w2, b2, hidden, raw, gparam, gctx, dpre, gw1, gb1, gw2, gb2, brightness_range, sat_lo,
                                 sat_hi, sharpness_range, B, D, 0, stream);
}

}  // extern "C"
