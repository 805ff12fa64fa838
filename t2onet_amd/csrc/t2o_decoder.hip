// t2o_decoder.hip -- one decoding step of the actor as a handful of launches (models/action_decoder.py:38-64
// Decoder.forward_step, models/attention.py:41-42 the output projection, models/actor.py:50,143,216 relu(bn1(fc(.)))).
//
// Everything here is a product with M = batch (<= 64 rows per tile) against a weight that lives in L2 / MALL (23 MB for
// the whole decoder): 0.6 GFLOP per step, i.e. pure latency.  As framework calls a step was ~35 forward and ~70 backward
// launches of 4-13 us each (library GEMMs with M = 64, gate / activation / concat / accumulate kernels) with the host
// unable to enqueue them as fast as they retire; here it is 6 forward + 9 backward launches (+ the attention core of
// t2o_kernels.hip), every pointwise stage living in the prologue or epilogue of the product next to it:
//
//   forward   k_dec_feat     feat = relu(bn1(fc(pooled)))          a workgroup owns 16 feature columns for ALL rows, so
//                                                                  BatchNorm1d's batch statistics are workgroup-local
//             k_dec_gemm     step_in = [emb[prev_op] | relu(vis_linear(feat))]
//             k_dec_lstm x2  gates = [x | h] . [W_ih | W_hh]^T + b, cell in the epilogue (a workgroup owns the 4 gates of
//                            4 hidden units)
//             (t2o_attn_fwd) mix, attn
//             k_dec_gemm     ctx = tanh(linear_out([mix | q]))     two K segments instead of a concat
//             k_dec_logits   log_softmax(out_linear(ctx))
//   backward  the same products against the untransposed weights ("TN": the lane's weight column is walked down its
//             rows), tanh' / relu' as the A operand's prologue, the LSTM cell's and the batch norm's pointwise
//             backward as two small kernels each.  Only DATA gradients are on this path: every dY is left in caller
//             storage together with its X, and the weight gradients are ONE product per weight over all steps of a
//             train step (K = steps x batch; host side, t2onet_amd/decoder_tape.py).
//
// The product itself: fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 multiply-adds) on 16 x 16 output tiles.
// What binds these kernels is neither arithmetic nor bandwidth but MEMORY-LEVEL PARALLELISM: the operands arrive from
// L2 / MALL / HBM (the encoder passes between two decoder steps sweep the caches) with 1-2 us of latency, a CU pulls
// ~60 GB/s, and the first version (4 waves per workgroup, 64 rows x 16 columns, one 32-wide K chunk prefetched: 40 KB
// in flight per CU, 64-128 workgroups on 256 CUs) took 16-28 us per product.  Now a workgroup is 16 WAVES that split K
// in 32-wide chunks (wave w takes chunks w, w+16, ...: a lane reads whole 16-byte pieces of a 128-byte line of its
// row) and keep up to 4 chunks of fragments in flight each, row tiles of 16 go to separate workgroups (grid.y) so
// that every CU has work, and the 16 partial tiles meet in LDS and are added in wave order (deterministic).
#include <hip/hip_runtime.h>

#include <cmath>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kW = 16;              // waves per workgroup: the K split
constexpr int kT = 64 * kW;

enum { A_PLAIN = 0, A_TANH_BWD = 1, A_RELU_BWD = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct GSeg {              // one K segment of a product: rows of A
  const float* x;          // x[row * ldx + k]
  const float* x2;         // second source of the prologue forms (tanh': the tanh output; relu': the relu output)
  float* a_store;          // nullable: workgroup column 0 stores the A values it forms (a_store[row * lds + k])
  int ldx, ldx2, lds, K;
};
struct GGroup {            // a range of output columns with its own weights and destination
  const float* w[2];       // per segment.  NT: w[n * ldw + k];  TN: w[k * ldw + n]
  const float* bias[2];    // nullable, added in the epilogue
  float* out;              // out[row * ldo + n]
  int ldw[2];
  int ldo, N, tiles;       // tiles = ceil(N / 16)
};
struct GemmArgs {
  GSeg seg[2];
  GGroup grp[2];
  int nseg, ngrp, M;
  // optional row gather riding along (the operator embedding): gdst[row * ldg + e] = emb[idx[row] * E + e]
  const float* emb;
  const long long* idx;
  long long* idx_copy;     // nullable
  float* gdst;
  int E, ldg, V;
};

template <int RB>
struct Frag {
  f32x4 a[RB][2];
  f32x4 b[2];
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// fragments of chunk c (32 k's) for this lane: A rows arow[rb] (clamped), weight row / column nrow (clamped).
// c is wave-uniform, so the segment's fields are scalar selects (no indexed copy of the kernel arguments).
template <int RB, bool TN, int AMODE>
__device__ __forceinline__ void load_frag(Frag<RB>& f, const GSeg& s0, const GSeg& s1, int C0, const float* w0, const float* w1, int ldw0,
                                          int ldw1, int c, const int* arow, const bool* astore, int nrow, int kg) {
  const bool s = c >= C0;
  const int cc = s ? c - C0 : c;
  const float* x = s ? s1.x : s0.x;
  const float* x2 = s ? s1.x2 : s0.x2;
  float* ast = s ? s1.a_store : s0.a_store;
  const int ldx = s ? s1.ldx : s0.ldx, ldx2 = s ? s1.ldx2 : s0.ldx2, lds = s ? s1.lds : s0.lds, K = s ? s1.K : s0.K;
  const float* ws = s ? w1 : w0;
  const int lw = s ? ldw1 : ldw0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kk = cc * 32 + h * 16 + 4 * kg;
    const bool valid = kk < K;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      f32x4 v = zero;
      if (valid) {
        v = ld4(x + (size_t)arow[rb] * ldx + kk);
        if (AMODE == A_TANH_BWD) {
          const f32x4 t = ld4(x2 + (size_t)arow[rb] * ldx2 + kk);
          v = v * (1.0f - t * t);
        } else if (AMODE == A_RELU_BWD) {
          const f32x4 t = ld4(x2 + (size_t)arow[rb] * ldx2 + kk);
          v.x = t.x > 0.0f ? v.x : 0.0f; v.y = t.y > 0.0f ? v.y : 0.0f; v.z = t.z > 0.0f ? v.z : 0.0f; v.w = t.w > 0.0f ? v.w : 0.0f;
        }
        if (ast != nullptr && astore[rb]) *reinterpret_cast<f32x4*>(ast + (size_t)arow[rb] * lds + kk) = v;
      }
      f.a[rb][h] = v;
    }
    f32x4 b = zero;
    if (valid) {
      if (TN) {
        const float* p = ws + (size_t)kk * lw + nrow;
        b.x = p[0]; b.y = p[lw]; b.z = p[2 * (size_t)lw]; b.w = p[3 * (size_t)lw];
      } else {
        b = ld4(ws + (size_t)nrow * lw + kk);
      }
    }
    f.b[h] = b;
  }
}

template <int RB>
__device__ __forceinline__ void mma_frag(f32x4* acc, const Frag<RB>& f) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[rb][h][i], f.b[h][i], acc[rb], 0, 0, 0);
}

// The workgroup's (16 RB) x 16 tile of  sum_seg A_seg . W_seg  as kW per-wave partial tiles in LDS:
// red[wave][row * 16 + col].  Ends with a barrier.  nseg == 1: s1 is never read.  PD = chunks in flight per wave.
template <int RB, bool TN, int AMODE, int PD>
__device__ __forceinline__ void gemm_core(const GSeg& s0, const GSeg& s1, int nseg, const float* w0, const float* w1, int ldw0, int ldw1,
                                          int nrow, int row0, int M, bool store_a, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int arow[RB];
  bool astore[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int r = row0 + rb * 16 + li;
    arow[rb] = r < M ? r : M - 1;
    astore[rb] = store_a && r < M;
  }
  const int C0 = (s0.K + 31) >> 5;
  const int C = C0 + (nseg > 1 ? (s1.K + 31) >> 5 : 0);
  f32x4 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  Frag<RB> f[PD];
#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (wave + p * kW < C) load_frag<RB, TN, AMODE>(f[p], s0, s1, C0, w0, w1, ldw0, ldw1, wave + p * kW, arow, astore, nrow, kg);
  for (int c = wave; c < C; c += PD * kW) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      if (c + p * kW < C) {
        mma_frag<RB>(acc, f[p]);
        if (c + (p + PD) * kW < C) load_frag<RB, TN, AMODE>(f[p], s0, s1, C0, w0, w1, ldw0, ldw1, c + (p + PD) * kW, arow, astore, nrow, kg);
      }
    }
  }
  float* mine = red + wave * (RB * 256);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[(rb * 16 + 4 * kg + r) * 16 + li] = acc[rb][r];
  __syncthreads();
}

template <int RB>
__device__ __forceinline__ float red_sum(const float* red, int e) {
  float s = red[e];
#pragma unroll
  for (int w = 1; w < kW; ++w) s += red[w * (RB * 256) + e];
  return s;
}

// ---- generic product with bias / activation epilogue --------------------------------------------------------------
template <bool TN, int AMODE, int ACT>
__global__ __launch_bounds__(kT) void k_dec_gemm(GemmArgs a) {
  constexpr int RB = 1;
  __shared__ float red[kW * RB * 256];
  const int tid = threadIdx.x, li = tid & 15;
  int bx = blockIdx.x;
  const bool g1 = a.ngrp > 1 && bx >= a.grp[0].tiles;       // (block-uniform: scalar selects of the group's fields)
  if (g1) bx -= a.grp[0].tiles;
  const float* w0 = g1 ? a.grp[1].w[0] : a.grp[0].w[0];
  const float* w1 = g1 ? a.grp[1].w[1] : a.grp[0].w[1];
  const float* b0 = g1 ? a.grp[1].bias[0] : a.grp[0].bias[0];
  const float* b1 = g1 ? a.grp[1].bias[1] : a.grp[0].bias[1];
  float* out = g1 ? a.grp[1].out : a.grp[0].out;
  const int ldw0 = g1 ? a.grp[1].ldw[0] : a.grp[0].ldw[0], ldw1 = g1 ? a.grp[1].ldw[1] : a.grp[0].ldw[1];
  const int ldo = g1 ? a.grp[1].ldo : a.grp[0].ldo, N = g1 ? a.grp[1].N : a.grp[0].N;
  const int col0 = bx * 16, row0 = blockIdx.y * (16 * RB);
  if (a.emb != nullptr) {                                    // the embedding rows, dealt over the whole grid
    const int q = a.E >> 2, total = a.M * q;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * kT + tid; i < total; i += gridDim.x * gridDim.y * kT) {
      const int r = i / q, e = (i - r * q) * 4;
      long long t = a.idx[r];
      t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
      *reinterpret_cast<f32x4*>(a.gdst + (size_t)r * a.ldg + e) = ld4(a.emb + (size_t)t * a.E + e);
    }
    if (a.idx_copy != nullptr && blockIdx.x == 0 && blockIdx.y == 0)
      for (int r = tid; r < a.M; r += kT) a.idx_copy[r] = a.idx[r];
  }
  const int n = col0 + li;
  gemm_core<RB, TN, AMODE, 4>(a.seg[0], a.seg[1], a.nseg, w0, w1, ldw0, ldw1, n < N ? n : N - 1, row0, a.M, blockIdx.x == 0, red);
  for (int e = tid; e < RB * 256; e += kT) {
    const int row = row0 + (e >> 4), col = col0 + (e & 15);
    if (row < a.M && col < N) {
      float v = red_sum<RB>(red, e);
      if (b0) v += b0[col];
      if (b1) v += b1[col];
      if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
      if (ACT == ACT_TANH) v = tanhf(v);
      out[(size_t)row * ldo + col] = v;
    }
  }
}

// ---- LSTM cell: gates = [x | h] . [W_ih | W_hh]^T + b_ih + b_hh (torch gate order i, f, g, o), cell in the epilogue ----
struct LstmArgs {
  GSeg seg[2];                    // x (K = input width), h_prev (K = H)
  const float* w[2];              // W_ih (4H, K0), W_hh (4H, H)
  int ldw[2];
  const float *b_ih, *b_hh;       // (4H) nullable
  const float* c_prev;            // (M, H)
  float *h_out, *c_out, *gates;   // (M, H), (M, H), (M, 4H) post-activation
  int M, H;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(kT) void k_dec_lstm(LstmArgs a) {
  constexpr int RB = 1;
  __shared__ float red[kW * RB * 256];
  const int tid = threadIdx.x, li = tid & 15;
  const int u0 = blockIdx.x * 4, row0 = blockIdx.y * (16 * RB), H = a.H;
  // column j of the tile = gate (j >> 2) of hidden unit u0 + (j & 3)
  gemm_core<RB, false, A_PLAIN, 4>(a.seg[0], a.seg[1], 2, a.w[0], a.w[1], a.ldw[0], a.ldw[1], (li >> 2) * H + u0 + (li & 3), row0, a.M,
                                   blockIdx.x == 0, red);
  for (int t = tid; t < RB * 64; t += kT) {
    const int r = t >> 2, u = t & 3, row = row0 + r, unit = u0 + u;
    if (row >= a.M) continue;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = red_sum<RB>(red, r * 16 + g * 4 + u);
      if (a.b_ih) v += a.b_ih[g * H + unit];
      if (a.b_hh) v += a.b_hh[g * H + unit];
      pre[g] = v;
    }
    const float ig = sigm(pre[0]), fg = sigm(pre[1]), gg = tanhf(pre[2]), og = sigm(pre[3]);
    const float cy = fg * a.c_prev[(size_t)row * H + unit] + ig * gg;
    const float hy = og * tanhf(cy);
    a.c_out[(size_t)row * H + unit] = cy;
    a.h_out[(size_t)row * H + unit] = hy;
    float* gs = a.gates + (size_t)row * 4 * H + unit;
    gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
  }
}

// pointwise backward of the cell: dh = dh_a + dh_b + dh_c (each nullable), dc_next nullable
struct LstmBwdArgs {
  const float *dh_a, *dh_b, *dh_c, *dc_next;
  int ld_a, ld_b, ld_c;                      // row strides of the dh sources (dc_next, states: H)
  const float *gates, *c_prev, *c_new;
  float *dgates, *dc_prev;                   // (M, 4H) pre-activation gate gradients, (M, H)
  int M, H;
};

__global__ __launch_bounds__(256) void k_dec_lstm_bwd(LstmBwdArgs a) {
  const int H = a.H, q = H >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.M * q) return;
  const int row = i / q, u = (i - row * q) * 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 dh = zero;
  if (a.dh_a) dh += ld4(a.dh_a + (size_t)row * a.ld_a + u);
  if (a.dh_b) dh += ld4(a.dh_b + (size_t)row * a.ld_b + u);
  if (a.dh_c) dh += ld4(a.dh_c + (size_t)row * a.ld_c + u);
  const f32x4 dcn = a.dc_next ? ld4(a.dc_next + (size_t)row * H + u) : zero;
  const float* gs = a.gates + (size_t)row * 4 * H + u;
  const f32x4 ig = ld4(gs), fg = ld4(gs + H), gg = ld4(gs + 2 * H), og = ld4(gs + 3 * H);
  const f32x4 cp = ld4(a.c_prev + (size_t)row * H + u), cn = ld4(a.c_new + (size_t)row * H + u);
  f32x4 tc;
  tc.x = tanhf(cn.x); tc.y = tanhf(cn.y); tc.z = tanhf(cn.z); tc.w = tanhf(cn.w);
  const f32x4 d_o = dh * tc * og * (1.0f - og);
  const f32x4 dc = dcn + dh * og * (1.0f - tc * tc);
  const f32x4 d_i = dc * gg * ig * (1.0f - ig);
  const f32x4 d_f = dc * cp * fg * (1.0f - fg);
  const f32x4 d_g = dc * ig * (1.0f - gg * gg);
  float* dg = a.dgates + (size_t)row * 4 * H + u;
  *reinterpret_cast<f32x4*>(dg) = d_i;
  *reinterpret_cast<f32x4*>(dg + H) = d_f;
  *reinterpret_cast<f32x4*>(dg + 2 * H) = d_g;
  *reinterpret_cast<f32x4*>(dg + 3 * H) = d_o;
  *reinterpret_cast<f32x4*>(a.dc_prev + (size_t)row * H + u) = dc * fg;
}

// ---- feature head: relu(bn1(fc(pooled))), models/actor.py:50 (training: batch statistics over the M rows) ----------
struct FeatArgs {
  GSeg seg;                                  // pooled (M, K)
  const float *w, *b;                        // fc (D, K), (D)
  const float *bn_w, *bn_b;                  // (D) nullable (affine=False)
  float *rmean, *rvar;                       // (D) nullable; updated in training mode
  long long* nbt;                            // nullable; += 1 in training mode
  float *fc_out, *stats, *feat;              // (M, D), (2, D) = mean, invstd, (M, D)
  float momentum, eps;
  int training, M, D;
};

template <int RB>
__global__ __launch_bounds__(kT) void k_dec_feat(FeatArgs a) {
  __shared__ float red[kW * RB * 256];
  __shared__ float part[16][16];
  const int tid = threadIdx.x, li = tid & 15;
  const int col0 = blockIdx.x * 16, M = a.M;
  const int n = col0 + li;
  gemm_core<RB, false, A_PLAIN, 2>(a.seg, a.seg, 1, a.w, nullptr, a.seg.K, 0, n < a.D ? n : a.D - 1, 0, M, blockIdx.x == 0, red);
  // the tile (+ bias) into red[0 .. RB*256): element e = row * 16 + col; threads 0 .. RB*256-1 own one element each
  const bool owner = tid < RB * 256;
  const int row = tid >> 4, c = tid & 15, col = col0 + c;
  float x = 0.0f;
  if (owner) {
    x = red_sum<RB>(red, tid);
    if (a.b && col < a.D) x += a.b[col];
    if (row < M && col < a.D) a.fc_out[(size_t)row * a.D + col] = x;
  }
  __syncthreads();
  if (owner) red[tid] = x;
  __syncthreads();
  // column statistics over the M rows: 16 row groups x 16 columns of partial sums, two passes (mean, then variance)
  float mean = 0.0f, invstd = 1.0f;
  if (a.training) {
    if (tid < 256) {
      float sacc = 0.0f;
      for (int r = tid >> 4; r < M; r += 16) sacc += red[r * 16 + c];
      part[tid >> 4][c] = sacc;
    }
    __syncthreads();
    float sum = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) sum += part[g][c];
    mean = sum / (float)M;
    __syncthreads();
    if (tid < 256) {
      float vacc = 0.0f;
      for (int r = tid >> 4; r < M; r += 16) { const float d = red[r * 16 + c] - mean; vacc += d * d; }
      part[tid >> 4][c] = vacc;
    }
    __syncthreads();
    float v = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += part[g][c];
    invstd = 1.0f / sqrtf(v / (float)M + a.eps);
    if (tid < 16 && col < a.D) {
      if (a.rmean) a.rmean[col] = (1.0f - a.momentum) * a.rmean[col] + a.momentum * mean;
      if (a.rvar) a.rvar[col] = (1.0f - a.momentum) * a.rvar[col] + a.momentum * (v / (float)(M > 1 ? M - 1 : 1));
    }
  } else if (col < a.D) {
    mean = a.rmean[col];
    invstd = 1.0f / sqrtf(a.rvar[col] + a.eps);
  }
  if (tid < 16 && col < a.D) { a.stats[col] = mean; a.stats[a.D + col] = invstd; }
  if (blockIdx.x == 0 && tid == 0 && a.training && a.nbt) *a.nbt += 1;
  if (owner && row < M && col < a.D) {
    float y = (x - mean) * invstd;
    if (a.bn_w) y *= a.bn_w[col];
    if (a.bn_b) y += a.bn_b[col];
    a.feat[(size_t)row * a.D + col] = fmaxf(y, 0.0f);
  }
}

// backward of relu(bn(x)) over the batch.  Workgroup = 16 columns; thread = (column, row group g of 16): its rows g,
// g+16, ... (at most 4 for M <= 64) stay in registers between the sums and the result -- one memory round trip.
struct FeatBwdArgs {
  const float *g_feat, *feat, *fc_out, *stats, *bn_w;
  float *d_fc, *d_bn;                        // (M, D), (2, D) = d weight, d bias
  int training, M, D;
};

__global__ __launch_bounds__(256) void k_dec_feat_bwd(FeatBwdArgs a) {
  __shared__ float sg[16][16], sgx[16][16];
  const int tid = threadIdx.x, c = tid & 15, rg = tid >> 4, col = blockIdx.x * 16 + c, D = a.D, M = a.M;
  const bool ok = col < D;
  const float mean = ok ? a.stats[col] : 0.0f, invstd = ok ? a.stats[D + col] : 0.0f;
  float g[4], xh[4];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = rg + 16 * j;
    g[j] = 0.0f; xh[j] = 0.0f;
    if (ok && r < M) {
      const size_t i = (size_t)r * D + col;
      g[j] = a.feat[i] > 0.0f ? a.g_feat[i] : 0.0f;
      xh[j] = (a.fc_out[i] - mean) * invstd;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1 += g[j]; s2 += g[j] * xh[j]; }
  sg[rg][c] = s1;
  sgx[rg][c] = s2;
  __syncthreads();
  float sum_g = 0.0f, sum_gx = 0.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { sum_g += sg[q][c]; sum_gx += sgx[q][c]; }
  if (!ok) return;
  if (rg == 0) { a.d_bn[col] = sum_gx; a.d_bn[D + col] = sum_g; }
  const float w = a.bn_w ? a.bn_w[col] : 1.0f;
  const float inv_m = 1.0f / (float)M;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = rg + 16 * j;
    if (r < M) {
      const float d = a.training ? (g[j] - sum_g * inv_m - xh[j] * (sum_gx * inv_m)) * invstd * w : g[j] * invstd * w;
      a.d_fc[(size_t)r * D + col] = d;
    }
  }
}

// ---- operator scores: log_softmax(out_linear(ctx)), V <= 16 ---------------------------------------------------------
struct LogitArgs {
  GSeg seg;                                  // ctx (M, D)
  const float *w, *b;                        // (V, D), (V) nullable
  float* logp;                               // (M, V)
  int M, V;
};

__global__ __launch_bounds__(kT) void k_dec_logits(LogitArgs a) {
  constexpr int RB = 1;
  __shared__ float red[kW * RB * 256];
  const int tid = threadIdx.x, li = tid & 15, row0 = blockIdx.y * (16 * RB);
  gemm_core<RB, false, A_PLAIN, 2>(a.seg, a.seg, 1, a.w, nullptr, a.seg.K, 0, li < a.V ? li : a.V - 1, row0, a.M, false, red);
  if (tid < RB * 16 && row0 + tid < a.M) {
    float s[16];
    float mx = -INFINITY;
    for (int v = 0; v < a.V; ++v) {
      s[v] = red_sum<RB>(red, tid * 16 + v) + (a.b ? a.b[v] : 0.0f);
      mx = fmaxf(mx, s[v]);
    }
    float sum = 0.0f;
    for (int v = 0; v < a.V; ++v) sum += expf(s[v] - mx);
    const float lse = logf(sum);
    for (int v = 0; v < a.V; ++v) a.logp[(size_t)(row0 + tid) * a.V + v] = s[v] - mx - lse;
  }
}

// backward: dlogits = g - softmax * sum(g);  dctx = g_ctx (nullable) + dlogits . W_out.   One workgroup per row.
struct LogitBwdArgs {
  const float *g_logp, *logp, *w, *g_ctx;
  float *d_logits, *d_ctx;
  int M, V, D;
};

__global__ __launch_bounds__(256) void k_dec_logits_bwd(LogitBwdArgs a) {
  __shared__ float dl[16];
  const int row = blockIdx.x, tid = threadIdx.x;
  if (tid < a.V) {
    float sum = 0.0f;
    for (int v = 0; v < a.V; ++v) sum += a.g_logp[(size_t)row * a.V + v];
    const float d = a.g_logp[(size_t)row * a.V + tid] - expf(a.logp[(size_t)row * a.V + tid]) * sum;
    dl[tid] = d;
    a.d_logits[(size_t)row * a.V + tid] = d;
  }
  __syncthreads();
  for (int col = tid; col < a.D; col += 256) {
    float s = a.g_ctx ? a.g_ctx[(size_t)row * a.D + col] : 0.0f;
    for (int v = 0; v < a.V; ++v) s += dl[v] * a.w[(size_t)v * a.D + col];
    a.d_ctx[(size_t)row * a.D + col] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------- host side
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    static thread_local char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return set_error(T2O_ELAUNCH, buf);
  }
  return T2O_OK;
}

inline int rb_for(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }         // the feature head: all rows in one workgroup
inline unsigned row_tiles(int M) { return (unsigned)((M + 15) / 16); }
inline bool al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

GSeg seg_of(const float* x, int ldx, int K, const float* x2 = nullptr, int ldx2 = 0, float* a_store = nullptr, int lds = 0) {
  GSeg s;
  s.x = x; s.x2 = x2; s.a_store = a_store; s.ldx = ldx; s.ldx2 = ldx2; s.lds = lds; s.K = K;
  return s;
}

void clear(GemmArgs& g) {
  g = GemmArgs();
  g.nseg = 1; g.ngrp = 1;
}

void set_group(GGroup& g, const float* w0, int ldw0, const float* w1, int ldw1, const float* b0, const float* b1, float* out, int ldo, int N) {
  g.w[0] = w0; g.w[1] = w1; g.ldw[0] = ldw0; g.ldw[1] = ldw1; g.bias[0] = b0; g.bias[1] = b1; g.out = out; g.ldo = ldo; g.N = N;
  g.tiles = (N + 15) / 16;
}

template <bool TN, int AMODE, int ACT>
void launch_gemm(const GemmArgs& g, hipStream_t st) {
  const dim3 grid((unsigned)(g.grp[0].tiles + (g.ngrp > 1 ? g.grp[1].tiles : 0)), row_tiles(g.M));
  k_dec_gemm<TN, AMODE, ACT><<<grid, kT, 0, st>>>(g);
}

void launch_lstm(const LstmArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(a.H / 4), row_tiles(a.M));
  k_dec_lstm<<<grid, kT, 0, st>>>(a);
}

}  // namespace

extern "C" {

int t2o_image_feature_fwd(const t2o_image_feature_t* p, void* stream) {
  if (!p) return set_error(T2O_EINVAL, "image_feature_fwd: null argument block");
  if (!p->fc_w || !p->pooled || !p->fc_out || !p->stats || !p->feat) return set_error(T2O_EINVAL, "image_feature_fwd: null pointer");
  if (p->B <= 0 || p->B > 64 || p->K <= 0 || p->K % 4 != 0 || p->D <= 0)
    return set_error(T2O_EINVAL, "image_feature: need 1 <= B <= 64 (one workgroup holds a feature column of the whole batch), K % 4 == 0");
  if (p->training && p->B < 2) return set_error(T2O_EINVAL, "image_feature: batch statistics need more than one row");
  if (!p->training && (!p->running_mean || !p->running_var)) return set_error(T2O_EINVAL, "image_feature: evaluation mode needs the running statistics");
  if (!al16(p->pooled) || !al16(p->fc_w) || (p->pooled_copy && !al16(p->pooled_copy)))
    return set_error(T2O_EINVAL, "image_feature: rows must be 16-byte aligned");
  FeatArgs a;
  a.seg = seg_of(p->pooled, p->K, p->K, nullptr, 0, p->pooled_copy, p->K);
  a.w = p->fc_w; a.b = p->fc_b; a.bn_w = p->bn_w; a.bn_b = p->bn_b;
  a.rmean = p->running_mean; a.rvar = p->running_var; a.nbt = p->num_batches_tracked;
  a.fc_out = p->fc_out; a.stats = p->stats; a.feat = p->feat;
  a.momentum = p->momentum; a.eps = p->eps; a.training = p->training; a.M = p->B; a.D = p->D;
  const hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((p->D + 15) / 16);
  const int rb = rb_for(p->B);
  if (rb == 1) k_dec_feat<1><<<grid, kT, 0, st>>>(a);
  else if (rb == 2) k_dec_feat<2><<<grid, kT, 0, st>>>(a);
  else k_dec_feat<4><<<grid, kT, 0, st>>>(a);
  return check_launch("image_feature_fwd");
}

int t2o_image_feature_bwd(const t2o_image_feature_t* p, void* stream) {
  if (!p) return set_error(T2O_EINVAL, "image_feature_bwd: null argument block");
  if (!p->fc_w || !p->g_feat || !p->feat || !p->fc_out || !p->stats || !p->d_fc || !p->d_bn)
    return set_error(T2O_EINVAL, "image_feature_bwd: null pointer");
  if (p->B <= 0 || p->B > 64 || p->K <= 0 || p->K % 4 != 0 || p->D <= 0 || p->D % 4 != 0)
    return set_error(T2O_EINVAL, "image_feature: need 1 <= B <= 64, K % 4 == 0, D % 4 == 0");
  if (!al16(p->d_fc)) return set_error(T2O_EINVAL, "image_feature: rows must be 16-byte aligned");
  const hipStream_t st = (hipStream_t)stream;
  FeatBwdArgs b;
  b.g_feat = p->g_feat; b.feat = p->feat; b.fc_out = p->fc_out; b.stats = p->stats; b.bn_w = p->bn_w;
  b.d_fc = p->d_fc; b.d_bn = p->d_bn; b.training = p->training; b.M = p->B; b.D = p->D;
  k_dec_feat_bwd<<<(unsigned)((p->D + 15) / 16), 256, 0, st>>>(b);
  if (p->d_pooled) {                                         // d pooled = d fc . W_fc
    GemmArgs g;
    clear(g);
    g.M = p->B;
    g.seg[0] = seg_of(p->d_fc, p->D, p->D);
    set_group(g.grp[0], p->fc_w, p->K, nullptr, 0, nullptr, nullptr, p->d_pooled, p->K, p->K);
    launch_gemm<true, A_PLAIN, ACT_NONE>(g, st);
  }
  return check_launch("image_feature_bwd");
}

static int step_shape_ok(const t2o_decoder_step_t* p, const char* who) {
  if (!p) return set_error(T2O_EINVAL, "decoder_step: null argument block");
  if (p->B <= 0 || p->D <= 0 || p->D % 64 != 0 || p->D > 1024 || p->E <= 0 || p->E % 4 != 0 || p->V <= 0 || p->V > 16 || p->L <= 0 || p->L > 64) {
    static thread_local char buf[160];
    snprintf(buf, sizeof buf, "%s: need B >= 1, D %% 64 == 0, D <= 1024, E %% 4 == 0, 1 <= V <= 16, 1 <= L <= 64", who);
    return set_error(T2O_EINVAL, buf);
  }
  return T2O_OK;
}

int t2o_decoder_step_fwd(const t2o_decoder_step_t* p, void* stream) {
  if (const int rc = step_shape_ok(p, "decoder_step_fwd")) return rc;
  if (!p->emb || !p->vis_w || !p->w_ih0 || !p->w_hh0 || !p->w_ih1 || !p->w_hh1 || !p->lo_w || !p->out_w || !p->prev_op || !p->feat ||
      !p->h0 || !p->c0 || !p->h1 || !p->c1 || !p->enc || !p->step_in || !p->gates0 || !p->gates1 || !p->h0n || !p->c0n || !p->h1n ||
      !p->c1n || !p->attn || !p->mix || !p->ctx || !p->logp)
    return set_error(T2O_EINVAL, "decoder_step_fwd: null pointer");
  const int B = p->B, D = p->D, E = p->E, X = E + D;
  const hipStream_t st = (hipStream_t)stream;
  GemmArgs g;
  // step_in = [emb[prev_op] | relu(vis_linear(feat))]
  clear(g);
  g.M = B;
  g.seg[0] = seg_of(p->feat, D, D, nullptr, 0, p->feat_copy, D);
  set_group(g.grp[0], p->vis_w, D, nullptr, 0, p->vis_b, nullptr, p->step_in + E, X, D);
  g.emb = p->emb; g.idx = p->prev_op; g.idx_copy = p->prev_op_copy; g.gdst = p->step_in; g.E = E; g.ldg = X; g.V = p->V;
  launch_gemm<false, A_PLAIN, ACT_RELU>(g, st);
  // layer 0, layer 1
  LstmArgs l;
  l.seg[0] = seg_of(p->step_in, X, X);
  l.seg[1] = seg_of(p->h0, D, D, nullptr, 0, p->hp0, D);
  l.w[0] = p->w_ih0; l.w[1] = p->w_hh0; l.ldw[0] = X; l.ldw[1] = D;
  l.b_ih = p->b_ih0; l.b_hh = p->b_hh0; l.c_prev = p->c0; l.h_out = p->h0n; l.c_out = p->c0n; l.gates = p->gates0; l.M = B; l.H = D;
  launch_lstm(l, st);
  l.seg[0] = seg_of(p->h0n, D, D);
  l.seg[1] = seg_of(p->h1, D, D, nullptr, 0, p->hp1, D);
  l.w[0] = p->w_ih1; l.w[1] = p->w_hh1; l.ldw[0] = D; l.ldw[1] = D;
  l.b_ih = p->b_ih1; l.b_hh = p->b_hh1; l.c_prev = p->c1; l.h_out = p->h1n; l.c_out = p->c1n; l.gates = p->gates1;
  launch_lstm(l, st);
  if (const int rc = check_launch("decoder_step_fwd (cells)")) return rc;
  // attention core over the request encoding, then ctx = tanh(linear_out([mix | q]))
  if (const int rc = t2o_attn_fwd(p->h1n, p->enc, p->attn, p->mix, B, p->L, D, stream)) return rc;
  clear(g);
  g.M = B; g.nseg = 2;
  g.seg[0] = seg_of(p->mix, D, D);
  g.seg[1] = seg_of(p->h1n, D, D);
  set_group(g.grp[0], p->lo_w, 2 * D, p->lo_w + D, 2 * D, p->lo_b, nullptr, p->ctx, D, D);
  launch_gemm<false, A_PLAIN, ACT_TANH>(g, st);
  // operator scores
  LogitArgs q;
  q.seg = seg_of(p->ctx, D, D);
  q.w = p->out_w; q.b = p->out_b; q.logp = p->logp; q.M = B; q.V = p->V;
  k_dec_logits<<<dim3(1, row_tiles(B)), kT, 0, st>>>(q);
  return check_launch("decoder_step_fwd");
}

int t2o_decoder_step_bwd(const t2o_decoder_step_t* p, void* stream) {
  if (const int rc = step_shape_ok(p, "decoder_step_bwd")) return rc;
  if (!p->vis_w || !p->w_ih0 || !p->w_hh0 || !p->w_ih1 || !p->w_hh1 || !p->lo_w || !p->out_w || !p->enc || !p->step_in || !p->gates0 ||
      !p->gates1 || !p->h0n || !p->c0n || !p->h1n || !p->c1n || !p->c0 || !p->c1 || !p->attn || !p->ctx || !p->logp || !p->d_lin ||
      !p->d_gates1 || !p->d_gates0 || !p->d_step_in || !p->d_vis || !p->d_mix || !p->d_qa || !p->d_q || !p->d_x1 || !p->d_ctx || !p->d_feat ||
      !p->d_h0 || !p->d_c0 || !p->d_h1 || !p->d_c1 || !p->d_enc || !p->d_logits)
    return set_error(T2O_EINVAL, "decoder_step_bwd: null pointer");
  if (!p->g_ctx && !p->g_logp) return set_error(T2O_EINVAL, "decoder_step_bwd: neither g_ctx nor g_logp (pass zeros for a step whose context carries no gradient)");
  const int B = p->B, D = p->D, E = p->E, X = E + D;
  const hipStream_t st = (hipStream_t)stream;
  // d ctx (+ the scores' part)
  const float* dctx = p->g_ctx;
  if (p->g_logp) {
    LogitBwdArgs q;
    q.g_logp = p->g_logp; q.logp = p->logp; q.w = p->out_w; q.g_ctx = p->g_ctx; q.d_logits = p->d_logits; q.d_ctx = p->d_ctx;
    q.M = B; q.V = p->V; q.D = D;
    k_dec_logits_bwd<<<(unsigned)B, 256, 0, st>>>(q);
    dctx = p->d_ctx;
  }
  // d lin = d ctx * (1 - ctx^2) (kept for the weight gradient);  d mix, d q (its part) = d lin . W_lo
  GemmArgs g;
  clear(g);
  g.M = B;
  g.seg[0] = seg_of(dctx, D, D, p->ctx, D, p->d_lin, D);
  g.ngrp = 2;
  set_group(g.grp[0], p->lo_w, 2 * D, nullptr, 0, nullptr, nullptr, p->d_mix, D, D);
  set_group(g.grp[1], p->lo_w + D, 2 * D, nullptr, 0, nullptr, nullptr, p->d_qa, D, D);
  launch_gemm<true, A_TANH_BWD, ACT_NONE>(g, st);
  if (const int rc = check_launch("decoder_step_bwd (output projection)")) return rc;
  // attention core: d mix -> d q (its part), d enc
  if (const int rc = t2o_attn_bwd(p->h1n, p->enc, p->attn, p->d_mix, nullptr, p->d_q, p->d_enc, B, p->L, D, stream)) return rc;
  // layer 1: d h1n = g_h1n + d q (attention) + d q (output projection)
  LstmBwdArgs c;
  c.dh_a = p->g_h1n; c.ld_a = D; c.dh_b = p->d_q; c.ld_b = D; c.dh_c = p->d_qa; c.ld_c = D; c.dc_next = p->g_c1n;
  c.gates = p->gates1; c.c_prev = p->c1; c.c_new = p->c1n; c.dgates = p->d_gates1; c.dc_prev = p->d_c1; c.M = B; c.H = D;
  const unsigned pw = (unsigned)((B * (D / 4) + 255) / 256);
  k_dec_lstm_bwd<<<pw, 256, 0, st>>>(c);
  clear(g);
  g.M = B; g.ngrp = 2;
  g.seg[0] = seg_of(p->d_gates1, 4 * D, 4 * D);
  set_group(g.grp[0], p->w_ih1, D, nullptr, 0, nullptr, nullptr, p->d_x1, D, D);
  set_group(g.grp[1], p->w_hh1, D, nullptr, 0, nullptr, nullptr, p->d_h1, D, D);
  launch_gemm<true, A_PLAIN, ACT_NONE>(g, st);
  // layer 0: d h0n = g_h0n + d x1
  c.dh_a = p->g_h0n; c.ld_a = D; c.dh_b = p->d_x1; c.ld_b = D; c.dh_c = nullptr; c.ld_c = 0; c.dc_next = p->g_c0n;
  c.gates = p->gates0; c.c_prev = p->c0; c.c_new = p->c0n; c.dgates = p->d_gates0; c.dc_prev = p->d_c0;
  k_dec_lstm_bwd<<<pw, 256, 0, st>>>(c);
  clear(g);
  g.M = B; g.ngrp = 2;
  g.seg[0] = seg_of(p->d_gates0, 4 * D, 4 * D);
  set_group(g.grp[0], p->w_ih0, X, nullptr, 0, nullptr, nullptr, p->d_step_in, X, X);
  set_group(g.grp[1], p->w_hh0, D, nullptr, 0, nullptr, nullptr, p->d_h0, D, D);
  launch_gemm<true, A_PLAIN, ACT_NONE>(g, st);
  // d vis = d step_in[:, E:] * (vis > 0) (kept);  d feat = d vis . W_vis
  clear(g);
  g.M = B;
  g.seg[0] = seg_of(p->d_step_in + E, X, D, p->step_in + E, X, p->d_vis, D);
  set_group(g.grp[0], p->vis_w, D, nullptr, 0, nullptr, nullptr, p->d_feat, D, D);
  launch_gemm<true, A_RELU_BWD, ACT_NONE>(g, st);
  return check_launch("decoder_step_bwd");
}

}  // extern "C"
