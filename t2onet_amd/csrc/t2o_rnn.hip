// t2o_rnn.hip -- the request encoder's LSTM layers (models/lang_encoder.py:70-113: a 2-layer bidirectional LSTM over
// packed, i.e. per-sample-length, token sequences), one launch per time step for BOTH directions, forward and backward.
//
// Why own kernels: the library's sequence entry point (MIOpen) is ~450 small launches per train step (4.5 ms of GPU time
// for a 17-step B=64 batch, launch-latency bound), needs the lengths on the host and a length-sorted batch; unrolled
// with framework operators it is ~2,000 launches.  Here a layer is 1 input GEMM (all steps and both directions, library)
// + L step kernels, shapes are static for a given L (hipGraph-capturable), lengths stay on the device.
//
// One step, forward (time t = s for direction 0, L-1-s for direction 1; state = the previous processed time's):
//     gates = gi[b][t][d] + b_ih[d] + b_hh[d] + h_prev[d][b] . W_hh[d]^T          (i, f, g, o: torch.nn.LSTM's order)
//     c2 = sigmoid(f) * c_prev + sigmoid(i) * tanh(g);   h2 = sigmoid(o) * tanh(c2)
//     valid = t < len[b]:  state := valid ? (h2, c2) : (h_prev, c_prev);   out[b][t][d] = valid ? h2 : 0
//   which is exactly what pack_padded_sequence -> LSTM -> pad_packed_sequence computes: a sample's forward state stops
//   at its last token, its reverse state starts (from zero) at its last token, outputs are zero at pads.
// Work split: workgroup = (64 hidden units, 8 samples, direction); thread = (hidden unit j, sample pair): the four gate
// columns of unit j are four coalesced 256-byte rows of the TRANSPOSED recurrent weight per k, h_prev of the 8 samples is
// staged in LDS (wave-uniform reads: broadcasts).  Backward: the same split; the recurrent term of dh (previous
// processed step's gate gradients times W_hh, a reduction over all 4H gate columns) is the kernel's prologue, so one
// launch per step suffices there too.  Weight / input gradients are library GEMMs over all steps at once (host side).
#include <hip/hip_runtime.h>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

constexpr int kThreads = 256;
constexpr int kTB = 8;             // samples per workgroup
constexpr int kTH = 64;            // hidden units per workgroup

struct LstmArgs {
  const float* gi;        // (B, L, D*4H) input-gate pre-activations (x W_ih^T), no bias
  const float* whh_t;     // (D, H, H, 4): whh_t[d][k][j][g] = W_hh[d][g*H + j][k]   (forward: 16-byte loads of a unit's 4 gates)
  const float* whh;       // (D, H, H, 4): whh[d][c4][k][q] = W_hh[d][4*c4 + q][k]     (backward: 4 gate columns per load)
  const float* b_ih;      // (D, 4H) or null
  const float* b_hh;      // (D, 4H) or null
  const long long* len;   // (B) valid lengths
  float* out;             // (B, L, D*H)
  float* hnew;            // (L, D, B, H) state after processing time t
  float* cnew;            // (L, D, B, H)
  float* gates;           // (L, D, B, 4H) post-activation i, f, g, o
  // backward
  const float* dout;      // (B, L, D*H) or null
  const float* dhn;       // (D, B, H) gradient of the final hidden state, or null
  const float* dcn;       // (D, B, H) or null
  float* dgates;          // (L, D, B, 4H) pre-activation gate gradients
  float* carry_h;         // (D, B, H) gradient of the state that passes a masked step unchanged
  float* dc;              // (D, B, H) running cell gradient
  int B, L, H, D;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Forward step.  The recurrent product is split over the workgroup's 4 waves along k (each wave a quarter of H for all
// 8 samples: 32 accumulators per lane, its weight column block as 16-byte loads -- whh_t is stored (D, H(k), H(j), 4
// gates)), partial sums meet in LDS, then thread (j, sample pair) applies the gates.  (First version: every thread the
// whole k range for 2 samples, scalar loads: 34 us per launch, latency-bound on 64 dependent load batches.)
__global__ __launch_bounds__(kThreads) void k_lstm_fwd_step(LstmArgs a, int s) {
  extern __shared__ float smem[];
  const int H = a.H, G = 4 * a.H;
  float* hs = smem;                                           // [kTB][H]
  float* red = smem + kTB * H;                                // [4 waves][kTB][4 gates][64]
  const int d = blockIdx.z, b0 = blockIdx.y * kTB, j0 = blockIdx.x * kTH;
  const int tid = threadIdx.x, j = tid & 63, wv = tid >> 6;
  const int t = d == 0 ? s : a.L - 1 - s, tp = d == 0 ? t - 1 : t + 1;     // tp: the time processed before t
  const bool first = s == 0;
  const int tpc = first ? t : tp;                             // (no previous time at the first step: never read)
  const float* hprev = a.hnew + ((size_t)tpc * a.D + d) * a.B * H;
  const float* cprev = a.cnew + ((size_t)tpc * a.D + d) * a.B * H;
  if (!first) {
    for (int i = tid; i < kTB * H; i += kThreads) {
      const int bb = b0 + i / H;
      hs[i] = bb < a.B ? hprev[(size_t)bb * H + (i % H)] : 0.0f;
    }
    __syncthreads();
    float acc[kTB][4];
#pragma unroll
    for (int q = 0; q < kTB; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[q][g] = 0.0f;
    const int kq = H / 4, k0 = wv * kq;
    const float4* w = reinterpret_cast<const float4*>(a.whh_t) + ((size_t)d * H + k0) * H + j0 + j;      // [k][j] float4 of gates
    const float* hk = hs + k0;
#pragma unroll 8
    for (int k = 0; k < kq; ++k) {
      const float4 wv4 = w[(size_t)k * H];
#pragma unroll
      for (int q = 0; q < kTB; ++q) {
        const float x = hk[q * H + k];
        acc[q][0] = fmaf(x, wv4.x, acc[q][0]); acc[q][1] = fmaf(x, wv4.y, acc[q][1]);
        acc[q][2] = fmaf(x, wv4.z, acc[q][2]); acc[q][3] = fmaf(x, wv4.w, acc[q][3]);
      }
    }
#pragma unroll
    for (int q = 0; q < kTB; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) red[((wv * kTB + q) * 4 + g) * 64 + j] = acc[q][g];
    __syncthreads();
  }
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int q = 2 * wv + qq, b = b0 + q;
    if (b >= a.B) continue;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = g * H + j0 + j;
      float v = a.gi[((size_t)b * a.L + t) * a.D * G + (size_t)d * G + col];
      if (a.b_ih) v += a.b_ih[d * G + col];
      if (a.b_hh) v += a.b_hh[d * G + col];
      if (!first) v += (red[((0 * kTB + q) * 4 + g) * 64 + j] + red[((1 * kTB + q) * 4 + g) * 64 + j]) +
                       (red[((2 * kTB + q) * 4 + g) * 64 + j] + red[((3 * kTB + q) * 4 + g) * 64 + j]);
      pre[g] = v;
    }
    const float ig = sigm(pre[0]), fg = sigm(pre[1]), gg = tanhf(pre[2]), og = sigm(pre[3]);
    const float cp = first ? 0.0f : cprev[(size_t)b * H + j0 + j];
    const float hp = first ? 0.0f : hs[q * H + j0 + j];
    const float c2 = fg * cp + ig * gg, h2 = og * tanhf(c2);
    const bool valid = (long long)t < a.len[b];
    const size_t so = (((size_t)t * a.D + d) * a.B + b) * H + j0 + j;
    a.hnew[so] = valid ? h2 : hp;
    a.cnew[so] = valid ? c2 : cp;
    a.out[((size_t)b * a.L + t) * a.D * H + (size_t)d * H + j0 + j] = valid ? h2 : 0.0f;
    float* gs = a.gates + (((size_t)t * a.D + d) * a.B + b) * G + j0 + j;
    gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
  }
}

// One backward step (s counts DOWN from L-1: the reverse of the forward's processing order).  Prologue: the recurrent
// part of dh, sum over the 4H gate columns of the gradients of the step processed before (in backward order) times
// W_hh -- again split over the 4 waves (a quarter of the columns each, all 8 samples), a.whh stored (D, 4H/4, H(k), 4
// columns) for 16-byte loads.
__global__ __launch_bounds__(kThreads) void k_lstm_bwd_step(LstmArgs a, int s) {
  extern __shared__ float smem[];
  const int H = a.H, G = 4 * a.H;
  float* dgs = smem;                                          // [kTB][G]: the next step's gate gradients of this batch tile
  float* red = smem + kTB * G;                                // [4 waves][kTB][64]
  const int d = blockIdx.z, b0 = blockIdx.y * kTB, j0 = blockIdx.x * kTH;
  const int tid = threadIdx.x, j = tid & 63, wv = tid >> 6;
  const int t = d == 0 ? s : a.L - 1 - s;
  const int tn = d == 0 ? t + 1 : t - 1;                      // the time processed AFTER t in the forward (its gradients exist already)
  const int tp = d == 0 ? t - 1 : t + 1;                      // the time processed BEFORE t (c_prev)
  const bool last = s == a.L - 1, first = s == 0;
  const int tnc = last ? t : tn, tpc = first ? t : tp;
  if (!last) {
    const float* dgn = a.dgates + ((size_t)tnc * a.D + d) * a.B * G;
    for (int i = tid; i < kTB * G; i += kThreads) {
      const int bb = b0 + i / G;
      dgs[i] = bb < a.B ? dgn[(size_t)bb * G + (i % G)] : 0.0f;
    }
    __syncthreads();
    float rec[kTB];
#pragma unroll
    for (int q = 0; q < kTB; ++q) rec[q] = 0.0f;
    const int cq = G / 16, c0 = wv * cq;                      // this wave's column quads [c0, c0 + cq)
    const float4* w = reinterpret_cast<const float4*>(a.whh) + ((size_t)d * (G / 4) + c0) * H + j0 + j;   // [c4][k] float4 of 4 columns
#pragma unroll 8
    for (int c = 0; c < cq; ++c) {
      const float4 wv4 = w[(size_t)c * H];
#pragma unroll
      for (int q = 0; q < kTB; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(dgs + q * G + 4 * (c0 + c));
        rec[q] = fmaf(g4.x, wv4.x, rec[q]); rec[q] = fmaf(g4.y, wv4.y, rec[q]);
        rec[q] = fmaf(g4.z, wv4.z, rec[q]); rec[q] = fmaf(g4.w, wv4.w, rec[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < kTB; ++q) red[(wv * kTB + q) * 64 + j] = rec[q];
    __syncthreads();
  }
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int q = 2 * wv + qq, b = b0 + q;
    if (b >= a.B) continue;
    const size_t si = ((size_t)d * a.B + b) * H + j0 + j;
    float dh_state, dc_state;
    if (last) {
      dh_state = a.dhn ? a.dhn[si] : 0.0f;
      dc_state = a.dcn ? a.dcn[si] : 0.0f;
    } else {
      dh_state = a.carry_h[si] + ((red[(0 * kTB + q) * 64 + j] + red[(1 * kTB + q) * 64 + j]) + (red[(2 * kTB + q) * 64 + j] + red[(3 * kTB + q) * 64 + j]));
      dc_state = a.dc[si];
    }
    const bool valid = (long long)t < a.len[b];
    float* dg = a.dgates + (((size_t)t * a.D + d) * a.B + b) * G + j0 + j;
    if (valid) {
      const float* gs = a.gates + (((size_t)t * a.D + d) * a.B + b) * G + j0 + j;
      const float ig = gs[0], fg = gs[H], gg = gs[2 * H], og = gs[3 * H];
      const float cp = first ? 0.0f : a.cnew[(((size_t)tpc * a.D + d) * a.B + b) * H + j0 + j];
      const float c2 = fg * cp + ig * gg, tc = tanhf(c2);
      const float dh = dh_state + (a.dout ? a.dout[((size_t)b * a.L + t) * a.D * H + (size_t)d * H + j0 + j] : 0.0f);
      const float dc = dc_state + dh * og * (1.0f - tc * tc);
      dg[0] = dc * gg * ig * (1.0f - ig);
      dg[H] = dc * cp * fg * (1.0f - fg);
      dg[2 * H] = dc * ig * (1.0f - gg * gg);
      dg[3 * H] = dh * tc * og * (1.0f - og);
      a.dc[si] = dc * fg;
      a.carry_h[si] = 0.0f;
    } else {
      dg[0] = 0.0f; dg[H] = 0.0f; dg[2 * H] = 0.0f; dg[3 * H] = 0.0f;
      a.dc[si] = dc_state;
      a.carry_h[si] = dh_state;
    }
  }
}

bool lstm_shape_ok(int B, int L, int H, int D) {
  return B > 0 && L > 0 && (D == 1 || D == 2) && H >= kTH && H % kTH == 0 && H <= 256 && (size_t)B * L * D * 4 * H < ((size_t)1 << 31);
}

}  // namespace

extern "C" {

int t2o_lstm_layer_fwd(const float* gi, const float* whh_t, const float* b_ih, const float* b_hh, const long long* len,
                       float* out, float* hnew, float* cnew, float* gates, int B, int L, int H, int D, void* stream) {
  if (!gi || !whh_t || !len || !out || !hnew || !cnew || !gates) return set_error(T2O_EINVAL, "lstm_layer_fwd: null pointer");
  if (!lstm_shape_ok(B, L, H, D)) return set_error(T2O_EUNSUPPORTED, "lstm_layer_fwd: H must be a multiple of 64 (<= 256: the gate-gradient tile of a step lives in LDS), 1 or 2 directions");
  LstmArgs a = {};
  a.gi = gi; a.whh_t = whh_t; a.b_ih = b_ih; a.b_hh = b_hh; a.len = len; a.out = out; a.hnew = hnew; a.cnew = cnew; a.gates = gates;
  a.B = B; a.L = L; a.H = H; a.D = D;
  const dim3 grid((unsigned)(H / kTH), (unsigned)((B + kTB - 1) / kTB), (unsigned)D);
  hipStream_t st = (hipStream_t)stream;
  for (int s = 0; s < L; ++s) k_lstm_fwd_step<<<grid, kThreads, sizeof(float) * (kTB * H + 4 * kTB * 4 * 64), st>>>(a, s);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "lstm_layer_fwd launch failed");
}

int t2o_lstm_layer_bwd(const float* whh, const long long* len, const float* cnew, const float* gates, const float* dout,
                       const float* dhn, const float* dcn, float* dgates, float* carry_h, float* dc,
                       int B, int L, int H, int D, void* stream) {
  if (!whh || !len || !cnew || !gates || !dgates || !carry_h || !dc) return set_error(T2O_EINVAL, "lstm_layer_bwd: null pointer");
  if (!lstm_shape_ok(B, L, H, D)) return set_error(T2O_EUNSUPPORTED, "lstm_layer_bwd: H must be a multiple of 64 (<= 256: the gate-gradient tile of a step lives in LDS), 1 or 2 directions");
  LstmArgs a = {};
  a.whh = whh; a.len = len; a.cnew = const_cast<float*>(cnew); a.gates = const_cast<float*>(gates); a.dout = dout; a.dhn = dhn; a.dcn = dcn;
  a.dgates = dgates; a.carry_h = carry_h; a.dc = dc;
  a.B = B; a.L = L; a.H = H; a.D = D;
  const dim3 grid((unsigned)(H / kTH), (unsigned)((B + kTB - 1) / kTB), (unsigned)D);
  hipStream_t st = (hipStream_t)stream;
  for (int s = L - 1; s >= 0; --s) k_lstm_bwd_step<<<grid, kThreads, sizeof(float) * (kTB * 4 * H + 4 * kTB * 64), st>>>(a, s);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "lstm_layer_bwd launch failed");
}

}  // extern "C"
