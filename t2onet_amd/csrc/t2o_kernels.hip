// libt2onet_hip: gfx950 kernels + C ABI (include/t2onet_hip.h) for the T2ONet
// executor/operator hot path.  Written for CDNA4 only: wave64, 256-thread
// workgroups, 16-byte coalesced global accesses, LDS tiles for the stencil,
// wave-shuffle + LDS reductions, deterministic two-stage parameter-gradient sums
// (no float atomics), XCD-aware workgroup -> tile mapping.
//
// All per-thread arithmetic lives in t2o_pixel_math.h / t2o_block_programs.h;
// this file is launch geometry, barriers, reductions and argument checking.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/t2onet_hip.h"
#include "t2o_block_programs.h"
#include "t2o_chain_kernels.h"
#include "t2o_jit.h"

using namespace t2o;

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

}  // namespace
namespace t2o { int set_error(int code, const char* msg) { return fail(code, msg); } }   // for t2o_norm.hip
namespace {

// ------------------------------------------------------------------ pointwise kernels
// grid: 1-D, B * nblk workgroups (XCD-remapped); each handles `iters` x 256 pixel groups of one sample.
template <int OP, int V, bool MASKED, bool L1>
__global__ __launch_bounds__(kThreads) void k_point_fwd(OpArgs a, int nblk) {
  int b, blk;
  wg_coords(nblk, b, blk);
  const int op = (OP == OP_DYNAMIC) ? a.op_id[b] : OP;
  if (OP == OP_DYNAMIC && op == OP_SHARPNESS) return;      // handled by the stencil kernel
  float l1 = 0.0f;
  if (OP == OP_DYNAMIC) {
    // one specialised body per operator: the kernel's register allocation is the widest
    // branch, not the union of all of them
    switch (op) {
#define T2O_CASE(K) case K: l1 = pointwise_fwd_thread<V, MASKED, L1>(a, K, b, blk, threadIdx.x); break;
      T2O_CASE(OP_BRIGHTNESS) T2O_CASE(OP_CONTRAST) T2O_CASE(OP_SATURATION) T2O_CASE(OP_COLOR)
      T2O_CASE(OP_TONE) T2O_CASE(OP_WHITE)
#undef T2O_CASE
      default: l1 = pointwise_fwd_thread<V, MASKED, L1>(a, OP_IDENTITY, b, blk, threadIdx.x); break;
    }
  } else {
    l1 = pointwise_fwd_thread<V, MASKED, L1>(a, OP, b, blk, threadIdx.x);
  }
  if (L1) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk_max + blk);
}

template <int OP, int V, bool MASKED, bool L1>
__global__ __launch_bounds__(kThreads) void k_point_bwd(OpArgs a, int nblk) {
  int b, blk;
  wg_coords(nblk, b, blk);
  const int op = (OP == OP_DYNAMIC) ? a.op_id[b] : OP;
  if (OP == OP_DYNAMIC && op == OP_SHARPNESS) return;
  float red[kRedSlots];
#pragma unroll
  for (int i = 0; i < kRedSlots; ++i) red[i] = 0.0f;
  __shared__ float tab[kTabStride];             // curve lookup table of this sample (curve operators only)
  if (OP == OP_COLOR || OP == OP_TONE || OP == OP_DYNAMIC) {
    if ((op == OP_COLOR || op == OP_TONE) && threadIdx.x == 0)
      curve_table_build(a.param + (size_t)b * a.param_stride, op == OP_COLOR, tab);
    __syncthreads();
  }
  if (OP == OP_DYNAMIC) {
    switch (op) {
#define T2O_CASE(K) case K: pointwise_bwd_thread<V, MASKED, L1>(a, K, b, blk, threadIdx.x, red, tab); break;
      T2O_CASE(OP_BRIGHTNESS) T2O_CASE(OP_CONTRAST) T2O_CASE(OP_SATURATION) T2O_CASE(OP_COLOR)
      T2O_CASE(OP_TONE) T2O_CASE(OP_WHITE)
#undef T2O_CASE
      default: pointwise_bwd_thread<V, MASKED, L1>(a, OP_IDENTITY, b, blk, threadIdx.x, red, tab); break;
    }
  } else {
    pointwise_bwd_thread<V, MASKED, L1>(a, OP, b, blk, threadIdx.x, red, tab);
  }
  if (op == OP_IDENTITY || op == OP_WHITE) return;        // no parameter gradient (uniform per block)
  block_reduce_store(red, nred_of(op), a.partials + ((size_t)b * a.nblk_max + blk) * kRedSlots);
}

// ------------------------------------------------------------------ sharpness kernels
// grid: 1-D, B * tiles workgroups; dynamic LDS.
template <bool DYN, int V>
__global__ __launch_bounds__(kThreads) void k_sharp_fwd(OpArgs a, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int b, tile;
  wg_coords(tiles, b, tile);
  if (DYN && a.op_id[b] != OP_SHARPNESS) return;
  sharp_fwd_phase_load<V>(a, b, tile, threadIdx.x, lds);
  __syncthreads();
  const float l1 = sharp_fwd_phase_compute<V>(a, b, tile, threadIdx.x, lds);
  if (a.target) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk_max + tile);
}

// ---- stencil kernels without LDS (W % 4 == 0) ----
// A thread owns a column strip of kStripRows rows x one aligned quad; a wave = 64 consecutive quads
// (a 256-pixel segment) x kStripRows rows; the workgroup = 4 strips stacked (16 rows).  Every row a strip
// needs is loaded up front with independent 16-byte loads, rows above / below are simply more registers,
// left / right neighbours come from the adjacent lanes by wave-wide DPP shifts (wave_shr:1 / wave_shl:1);
// only lane 0 / lane 63 of a segment that does not touch the image border read their outside column
// from global memory.  No LDS staging pass, no barrier, no item arithmetic: these kernels stream.
// (The LDS-tile kernels above remain for W % 4 != 0.  Measured at bs=64
// 256x256 inside the benchmark step: backward 48.6 -> 31.0 us.  The tile kernels were issue-bound, ~170
// vector + ~90 scalar instructions per pixel mostly for window addressing and predication, and their
// load / compute / store phases added up exactly, with or without a register prefetch of the next tile.)
// blocks per sample = ceil(H / 16) * ceil(W / 256).

__device__ __forceinline__ float dpp_wave_shr1(float v) {   // lane i <- lane i-1 (lane 0 <- 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_wave_shl1(float v) {   // lane i <- lane i+1 (lane 63 <- 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// WIDE = the image is wider than one 256-pixel segment (outside-column code compiled in).
template <bool DYN, bool WIDE, int ROWS>
__global__ __launch_bounds__(kThreads) void k_sharp_fwd_strip(OpArgs a, int nblk, int nseg) {
  int b, blk;
  wg_coords(nblk, b, blk);
  if (DYN && a.op_id[b] != OP_SHARPNESS) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = (blk / nseg) * (4 * ROWS) + wave * ROWS, gx0 = (blk % nseg) * 256 + 4 * lane;
  const bool col_live = gx0 < a.W;
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const float p = a.param[(size_t)b * a.param_stride];
  const bool ext_l = WIDE && lane == 0 && gx0 > 0 && col_live, ext_r = WIDE && lane == 63 && gx0 + 4 < a.W;
  float l1 = 0.0f;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const float* xp = a.img + ((size_t)b * 3 + c) * hw;
    float xr[ROWS + 2][4], tg[ROWS][4];
#pragma unroll
    for (int k = 0; k < ROWS + 2; ++k) {
      const int y = y0 - 1 + k;
      xr[k][0] = xr[k][1] = xr[k][2] = xr[k][3] = 0.0f;
      if (col_live && y >= 0 && y < a.H) load_vec<4>(xp + (unsigned)y * (unsigned)a.W + (unsigned)gx0, xr[k]);
    }
    if (a.target) {
#pragma unroll
      for (int k = 0; k < ROWS; ++k) {
        const int y = y0 + k;
        tg[k][0] = tg[k][1] = tg[k][2] = tg[k][3] = 0.0f;
        if (col_live && y < a.H) load_vec<4>(a.target + ((size_t)b * 3 + c) * hw + (unsigned)y * (unsigned)a.W + (unsigned)gx0, tg[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      const int y = y0 + k;
      const bool live = col_live && y < a.H;
      const float* ce = xr[k + 1];
      float L = dpp_wave_shr1(ce[3]), R = dpp_wave_shl1(ce[0]);
      if (ext_l && live) L = xp[(unsigned)y * (unsigned)a.W + (unsigned)gx0 - 1];
      if (ext_r && live) R = xp[(unsigned)y * (unsigned)a.W + (unsigned)gx0 + 4];
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float left = i == 0 ? L : ce[i > 0 ? i - 1 : 0], right = i == 3 ? R : ce[i < 3 ? i + 1 : 3];
        o[i] = ce[i] + p * sharp_delta(ce[i], xr[k][i], left, right, xr[k + 2][i]);
      }
      if (!live) continue;
      const unsigned off = (unsigned)y * (unsigned)a.W + (unsigned)gx0;
      if (a.mask_ch) {
        float m[4];
        load_vec<4>(a.mask + ((size_t)b * a.mask_ch + (a.mask_ch == 3 ? c : 0)) * hw + off, m);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = blend(o[i], ce[i], m[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = clamp01(o[i]);
      store_vec<4>(a.out + ((size_t)b * 3 + c) * hw + off, o);
      if (a.target) {
#pragma unroll
        for (int i = 0; i < 4; ++i) l1 += fabsf(o[i] - tg[k][i]);
      }
    }
  }
  if (a.target) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk_max + blk);
}

// Backward, strip layout as above: x rows y0-2..y0+5 and gradient (or L1 target) rows y0-1..y0+4
// are loaded up front (14 independent 16-byte loads per plane), dz is computed once per window row in
// registers (in place of the gradient rows), then the symmetric stencil is applied to it.
// WIDE (W > 256): segments OVERLAP by one quad on each side -- lanes 0 and 63 only compute the dz column
// their neighbours need (their own outer columns may be wrong and are never used) and store nothing, so a
// wave outputs 62 quads = kStripWideCols pixels and no lane ever needs a dz from outside its wave.

// MASKED: out = clamp(blend(x + p * Lap(x), x, m)): with do = dz * m the input gradient is
// dz * (1 - m) + do + p * Lap(do) and d loss / d p sums do * Lap(x); the mask rows ride along in registers.
// VAL (fused-L1 calls only): the launch also emits |out - target| partial sums (a.loss_partials) and, when a.out is set,
// the output image -- a value-and-gradient call needs no forward launch.  A separate instantiation: the plain kernels keep
// their register count (119 / 173 VGPRs; +29 with the value outputs).
template <bool DYN, bool WIDE, bool MASKED, bool VAL = false>
__global__ __launch_bounds__(kThreads) void k_sharp_bwd_strip(OpArgs a, int nblk, int nseg) {
  int b, blk;
  wg_coords(nblk, b, blk);
  if (DYN && a.op_id[b] != OP_SHARPNESS) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = (blk / nseg) * (4 * kStripRows) + wave * kStripRows;
  const int gx0 = WIDE ? (blk % nseg) * kStripWideCols - 4 + 4 * lane : 4 * lane;
  const bool col_live = gx0 >= 0 && gx0 < a.W;
  const bool own = !WIDE || (lane >= 1 && lane <= 62);      // this lane's quad is output (not an overlap lane)
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const float p = a.param[(size_t)b * a.param_stride];
  const float gs = a.target ? a.gloss[0] * a.inv_n : 0.0f;
  float red0 = 0.0f;
  float l1 = 0.0f;                  // L1 form: sum |out - target| over this thread's output pixels (value-and-gradient calls)
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const float* xp = a.img + ((size_t)b * 3 + c) * hw;
    const float* gp = (a.target ? a.target : a.gout) + ((size_t)b * 3 + c) * hw;
    const float* mp = MASKED ? a.mask + ((size_t)b * a.mask_ch + (a.mask_ch == 3 ? c : 0)) * hw : nullptr;
    float xr[kStripRows + 4][4], g[kStripRows + 2][4], mk[MASKED ? kStripRows + 2 : 1][4], pass[MASKED ? kStripRows : 1][4];
#pragma unroll
    for (int k = 0; k < kStripRows + 4; ++k) {
      const int y = y0 - 2 + k;
      xr[k][0] = xr[k][1] = xr[k][2] = xr[k][3] = 0.0f;
      if (col_live && y >= 0 && y < a.H) load_vec<4>(xp + (unsigned)y * (unsigned)a.W + (unsigned)gx0, xr[k]);
    }
#pragma unroll
    for (int k = 0; k < kStripRows + 2; ++k) {
      const int y = y0 - 1 + k;
      g[k][0] = g[k][1] = g[k][2] = g[k][3] = 0.0f;
      if (MASKED) mk[k][0] = mk[k][1] = mk[k][2] = mk[k][3] = 0.0f;
      if (col_live && y >= 0 && y < a.H) {
        load_vec<4>(gp + (unsigned)y * (unsigned)a.W + (unsigned)gx0, g[k]);
        if (MASKED) load_vec<4>(mp + (unsigned)y * (unsigned)a.W + (unsigned)gx0, mk[k]);
      }
    }
    // dz for window rows y0-1 .. y0+kStripRows (in place of g)
#pragma unroll
    for (int k = 0; k < kStripRows + 2; ++k) {
      const int y = y0 - 1 + k;
      const bool in = col_live && y >= 0 && y < a.H;
      const float* ce = xr[k + 1];
      const float L = dpp_wave_shr1(ce[3]), R = dpp_wave_shl1(ce[0]);
      const bool mine = VAL && k >= 1 && k <= kStripRows && own && in;        // an output pixel of this thread
      float oz[VAL ? 4 : 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float left = i == 0 ? L : ce[i > 0 ? i - 1 : 0], right = i == 3 ? R : ce[i < 3 ? i + 1 : 3];
        const float d = sharp_delta(ce[i], xr[k][i], left, right, xr[k + 2][i]);
        const float m = MASKED ? mk[k][i] : 1.0f;
        float z = ce[i] + p * d;
        if (MASKED) z = blend(z, ce[i], m);
        if constexpr (VAL) {
          oz[i] = clamp01(z);
          if (mine) l1 += fabsf(oz[i] - g[k][i]);
        }
        const float gz = a.target ? sign_of(clamp01(z) - g[k][i]) * gs : g[k][i];
        const float dz = (in && z >= 0.0f && z <= 1.0f) ? gz : 0.0f;
        g[k][i] = MASKED ? dz * m : dz;                                   // do
        if (MASKED && k >= 1 && k <= kStripRows) pass[k - 1][i] = dz * (1.0f - m);
        if (k >= 1 && k <= kStripRows && own) red0 += dz * m * d;
      }
      if constexpr (VAL) {
        if (a.out && mine) store_vec<4>(a.out + ((size_t)b * 3 + c) * hw + (unsigned)y * (unsigned)a.W + (unsigned)gx0, oz);
      }
    }
    // gimg rows y0 .. y0+kStripRows-1
#pragma unroll
    for (int k = 1; k <= kStripRows; ++k) {
      const int y = y0 - 1 + k;
      const float* ce = g[k];
      const float L = dpp_wave_shr1(ce[3]), R = dpp_wave_shl1(ce[0]);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float left = i == 0 ? L : ce[i > 0 ? i - 1 : 0], right = i == 3 ? R : ce[i < 3 ? i + 1 : 3];
        o[i] = ce[i] + p * sharp_delta(ce[i], g[k - 1][i], left, right, g[k + 1][i]);
        if (MASKED) o[i] = pass[k - 1][i] + o[i];
      }
      if (a.gimg && own && col_live && y < a.H)
        store_vec<4>(a.gimg + ((size_t)b * 3 + c) * hw + (unsigned)y * (unsigned)a.W + (unsigned)gx0, o);
    }
  }
  block_reduce_store1(red0, a.partials + ((size_t)b * a.nblk_max + blk) * kRedSlots);
  if constexpr (VAL) {
    __syncthreads();                                           // (block_reduce_store1's staging cells are reused)
    block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk_max + blk);
  }
}

template <bool DYN, int V>
__global__ __launch_bounds__(kThreads) void k_sharp_bwd(OpArgs a, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int b, tile;
  wg_coords(tiles, b, tile);
  if (DYN && a.op_id[b] != OP_SHARPNESS) return;
  sharp_bwd_phase_load<V>(a, b, tile, threadIdx.x, lds);
  __syncthreads();
  float red0 = 0.0f;
  sharp_bwd_phase_dz<V>(a, b, tile, threadIdx.x, lds, red0);
  __syncthreads();
  sharp_bwd_phase_out<V>(a, b, tile, threadIdx.x, lds);
  block_reduce_store1(red0, a.partials + ((size_t)b * a.nblk_max + tile) * kRedSlots);
}

template <int V, bool L1>
__global__ __launch_bounds__(kThreads) void k_chain_fwd(ChainArgs a) {
  __shared__ float tab[kMaxChain * kTabStride];
  int b, blk;
  wg_coords(a.nblk, b, blk);
  if ((int)threadIdx.x < a.K) chain_build_table(a, b, threadIdx.x, tab);
  __syncthreads();
  const float l1 = chain_fwd_thread<V, L1>(a, b, blk, threadIdx.x, tab);
  if (L1) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk + blk);
}

template <int V, bool L1, class SEQ>
__global__ __launch_bounds__(kThreads) void k_chain_fwd_static(ChainArgs a) { chain_fwd_static_body<V, L1, SEQ>(a); }

template <int V, bool L1>
__global__ __launch_bounds__(kThreads) void k_chain_bwd(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [accumulator cells: NB rows of kAccStride][save area]
  __shared__ float tab[kMaxChain * kTabStride];
  __shared__ float bsum[kMaxChainBins];
  int b, blk;
  wg_coords(a.nblk, b, blk);
  const int S = a.slot_off[kMaxChain], NB = a.bin_off[kMaxChain];
  for (int i = threadIdx.x; i < NB * kAccStride; i += kThreads) lds[i] = 0.0f;
  if ((int)threadIdx.x < a.K) chain_build_table(a, b, threadIdx.x, tab);
  __syncthreads();
  LdsAcc acc{lds};
  const float l1 = chain_bwd_thread<V, L1>(a, b, blk, threadIdx.x, tab, lds + NB * kAccStride, acc);
  if (L1 && a.loss_partials) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk + blk);   // value-and-gradient calls
  __syncthreads();
  for (int s = threadIdx.x; s < NB; s += kThreads) {            // cell rows -> per-workgroup sums, fixed order
    float sum = 0.0f;
    for (int q = 0; q < kThreads / 4; ++q) sum += lds[s * kAccStride + q];
    bsum[s] = sum;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += kThreads)
    a.partials[((size_t)b * a.nblk + blk) * S + s] = chain_slot_value(a, s, bsum);
}

// the backward for an operator list fixed at compile time: body in t2o_chain_kernels.h (shared with the hipRTC path,
// t2o_fused_sequence_prepare).  One instantiation per entry of the dispatch in fused_chain_launch_bwd.
template <bool L1, class SEQ, bool SV_LDS, int MINW, bool VAL = false>
__global__ __launch_bounds__(kThreads, MINW) void k_chain_bwd_static(ChainArgs a) { chain_bwd_static_body<L1, SEQ, SV_LDS, VAL>(a); }

// one workgroup per sample: per-block sums -> raw sums -> parameter gradients of every chain operator.
// 8 threads per slot walk the block rows (stride 8), then a fixed-order LDS combine.
__global__ __launch_bounds__(kThreads) void k_chain_finalize(ChainArgs a, float* gparams) {
  __shared__ float part[8][kMaxChainSlots];
  __shared__ float sums[kMaxChainSlots];
  const int b = blockIdx.x;
  const int S = a.slot_off[kMaxChain];
  for (int w = threadIdx.x; w < S * 8; w += kThreads) {
    const int s = w % S, lane8 = w / S;
    float acc = 0.0f;
    for (int k = lane8; k < a.nblk; k += 8) acc += a.partials[((size_t)b * a.nblk + k) * S + s];
    part[lane8][s] = acc;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += kThreads) {
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) v += part[j][s];
    sums[s] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < a.K) {
    const int k = threadIdx.x, op = a.ops[k];
    float* grow = gparams + ((size_t)a.src[k] * a.B + b) * a.gparam_stride;
    const int nz = a.gparam_stride < kMaxParam ? a.gparam_stride : kMaxParam;
    for (int i = 0; i < nz; ++i) grow[i] = 0.0f;
    finalize_param_grad(op, a.params + ((size_t)a.src[k] * a.B + b) * a.param_stride, sums + a.slot_off[k], grow);
  }
}

// ------------------------------------------------------------------ finalisation kernels
// One workgroup per sample: sum the per-block raw sums in a fixed order, then raw sums -> gparam.
// Thread t owns slot (t % w) of every (256 / w)-th block row starting at (t / w), w = 1 / 8 / 32 for operators with 1 / <= 8 /
// more sums per block: the one-parameter operators used 8 of the 256 threads before (w was always 32), which made their
// finalize 10.5 us at 16 x 512^2 (256 block rows per sample) against 4.7 us at 64 x 256^2 -- the shape dependence the
// round-2 review found in the materialised backward legs.
__global__ __launch_bounds__(kThreads) void k_finalize_params(OpArgs a, float* gparam, int gparam_stride,
                                                              int nblk_point, int nblk_sharp) {
  __shared__ float part[kThreads];
  __shared__ float sums[kRedSlots];
  const int b = blockIdx.x;
  const int op = (a.op == OP_DYNAMIC) ? a.op_id[b] : a.op;
  float* grow = gparam + (size_t)b * gparam_stride;
  const int np = op_num_params(op);
  if (op == OP_IDENTITY || op == OP_WHITE || np == 0) {            // uniform per workgroup
    if ((int)threadIdx.x < np) grow[threadIdx.x] = 0.0f;
    return;
  }
  const int n = nred_of(op);
  const int nb = (op == OP_SHARPNESS) ? nblk_sharp : nblk_point;
  const float* base = a.partials + (size_t)b * a.nblk_max * kRedSlots;
  const int w = n <= 1 ? 1 : (n <= 8 ? 8 : 32), nch = kThreads / w;           // (uniform per workgroup)
  const int slot = threadIdx.x % w, chunk = threadIdx.x / w;
  float acc = 0.0f;
  if (slot < n)
    for (int k = chunk; k < nb; k += nch) acc += base[(size_t)k * kRedSlots + slot];
  part[chunk * w + slot] = acc;
  __syncthreads();
  __shared__ float part2[8 * 32];
  if ((int)threadIdx.x < 8 * w) {                                              // 8 lanes per slot, every 8th chunk each: fixed order
    float s = 0.0f;
    for (int c = threadIdx.x / w; c < nch; c += 8) s += part[c * w + slot];
    part2[threadIdx.x] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += part2[j * w + threadIdx.x];
    sums[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) finalize_param_grad(op, a.param + (size_t)b * a.param_stride, sums, grow);
  if ((int)threadIdx.x >= np && (int)threadIdx.x < kMaxParam && (int)threadIdx.x < gparam_stride) grow[threadIdx.x] = 0.0f;
}

// The same for every operator of a materialised sequence in ONE launch: operator k keeps its
// per-block rows in its own region of the workspace.
struct MultiFinalize {
  const float* params;     // (K,B,24)
  float* gparams;          // (K,B,24)
  const float* partials;   // K regions of region_floats
  size_t region_floats;
  int ops[kMaxChain];
  int nblk[kMaxChain];
  int K, B, nblk_max;
};

__global__ __launch_bounds__(kThreads) void k_finalize_params_multi(MultiFinalize m) {
  __shared__ float part[kThreads];
  __shared__ float sums[kRedSlots];
  const int k = blockIdx.x / m.B, b = blockIdx.x % m.B;
  const int op = m.ops[k];
  float* grow = m.gparams + ((size_t)k * m.B + b) * kMaxParam;
  const int np = op_num_params(op);
  if (op == OP_IDENTITY || op == OP_WHITE || np == 0) {
    if ((int)threadIdx.x < kMaxParam) grow[threadIdx.x] = 0.0f;
    return;
  }
  const int n = nred_of(op), nb = m.nblk[k];
  const float* base = m.partials + (size_t)k * m.region_floats + (size_t)b * m.nblk_max * kRedSlots;
  const int w = n <= 1 ? 1 : (n <= 8 ? 8 : 32), nch = kThreads / w;           // (as k_finalize_params)
  const int slot = threadIdx.x % w, chunk = threadIdx.x / w;
  float acc = 0.0f;
  if (slot < n)
    for (int r = chunk; r < nb; r += nch) acc += base[(size_t)r * kRedSlots + slot];
  part[chunk * w + slot] = acc;
  __syncthreads();
  __shared__ float part2[8 * 32];
  if ((int)threadIdx.x < 8 * w) {                                              // (as k_finalize_params)
    float v = 0.0f;
    for (int c = threadIdx.x / w; c < nch; c += 8) v += part[c * w + slot];
    part2[threadIdx.x] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) v += part2[j * w + threadIdx.x];
    sums[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) finalize_param_grad(op, m.params + ((size_t)k * m.B + b) * kMaxParam, sums, grow);
  if ((int)threadIdx.x >= np && (int)threadIdx.x < kMaxParam) grow[threadIdx.x] = 0.0f;
}

// loss[0] = inv_n * sum of all loss partials (single workgroup, fixed order).  Flat over (sample, block
// row): every thread's loads are independent and issued back to back (the kernel is one memory round trip
// plus the launch, not one round trip per group of samples).
__global__ __launch_bounds__(kThreads) void k_finalize_loss(OpArgs a, float* loss, int nblk_point, int nblk_sharp) {
  const int nbmax = nblk_point > nblk_sharp ? nblk_point : nblk_sharp;
  const int total = a.B * nbmax;
  float acc = 0.0f;
#pragma unroll 8
  for (int idx = threadIdx.x; idx < total; idx += kThreads) {
    const int b = idx / nbmax, k = idx - b * nbmax;
    const int op = (a.op == OP_DYNAMIC) ? a.op_id[b] : a.op;
    const int nb = (op == OP_SHARPNESS) ? nblk_sharp : nblk_point;
    if (k < nb) acc += a.loss_partials[(size_t)b * a.nblk_max + k];
  }
  __shared__ float out1;
  block_reduce_store1(acc, &out1);
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = out1 * a.inv_n;
}

// ------------------------------------------------------------------ plain L1 over n floats
template <int V>
__global__ __launch_bounds__(kThreads) void k_l1_fwd(const float* pred, const float* target, float* partial,
                                                     size_t n, int iters) {
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    const size_t g = ((size_t)blockIdx.x * iters + it) * kThreads + threadIdx.x;
    if (g * V >= n) break;
    float p[V], t[V];
    load_vec<V>(pred + g * V, p);
    load_vec<V>(target + g * V, t);
#pragma unroll
    for (int i = 0; i < V; ++i) acc += fabsf(p[i] - t[i]);
  }
  block_reduce_store1(acc, partial + blockIdx.x);
}

__global__ __launch_bounds__(kThreads) void k_l1_finalize(const float* partial, int nblk, float inv_n, float* loss) {
  float acc = 0.0f;
  for (int k = threadIdx.x; k < nblk; k += kThreads) acc += partial[k];
  __shared__ float out1;
  block_reduce_store1(acc, &out1);
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = out1 * inv_n;
}

template <int V>
__global__ __launch_bounds__(kThreads) void k_l1_bwd(const float* pred, const float* target, const float* gloss,
                                                     float* gpred, size_t n, float inv_n, int iters) {
  const float gs = gloss[0] * inv_n;
  for (int it = 0; it < iters; ++it) {
    const size_t g = ((size_t)blockIdx.x * iters + it) * kThreads + threadIdx.x;
    if (g * V >= n) break;
    float p[V], t[V], o[V];
    load_vec<V>(pred + g * V, p);
    load_vec<V>(target + g * V, t);
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = sign_of(p[i] - t[i]) * gs;
    store_vec<V>(gpred + g * V, o);
  }
}

// ------------------------------------------------------------------ END select fused with the L1 loss
// train_seq2seqL1.py:78-85: the loss is taken on the image at each sample's first END token (else its last one):
// pred[b] = imgs[first[b]][b].  Forward and backward read only that image of every sample (no (B,T,3,H,W) stack, no
// gather, no zero-filled scatter target); the backward writes the gradient of ALL T step images in one launch
// (sign(pred - target) * gloss / n where selected, zero elsewhere).  Same element -> workgroup partition and summation
// order as k_l1_fwd on the gathered tensor: the loss is bit-identical to that path.
struct EndSelArgs {
  const float* img[8];
  float* gimg[8];
  int T;
};

__device__ __forceinline__ const float* end_sel_src(const EndSelArgs& s, int f) {
  const float* p = s.img[0];
#pragma unroll
  for (int t = 1; t < 8; ++t) p = (f == t) ? s.img[t] : p;
  return p;
}

template <int V>
__global__ __launch_bounds__(kThreads) void k_end_l1_fwd(EndSelArgs s, const long long* first, const float* target, float* partial,
                                                         size_t n, size_t row, int iters) {
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    const size_t g = ((size_t)blockIdx.x * iters + it) * kThreads + threadIdx.x;
    if (g * V >= n) break;
    const int f = (int)first[(g * V) / row];
    float p[V], t[V];
    load_vec<V>(end_sel_src(s, f) + g * V, p);
    load_vec<V>(target + g * V, t);
#pragma unroll
    for (int i = 0; i < V; ++i) acc += fabsf(p[i] - t[i]);
  }
  block_reduce_store1(acc, partial + blockIdx.x);
}

template <int V>
__global__ __launch_bounds__(kThreads) void k_end_l1_bwd(EndSelArgs s, const long long* first, const float* target, const float* gloss,
                                                         size_t n, size_t row, float inv_n, int iters) {
  const float gs = gloss[0] * inv_n;
  for (int it = 0; it < iters; ++it) {
    const size_t g = ((size_t)blockIdx.x * iters + it) * kThreads + threadIdx.x;
    if (g * V >= n) break;
    const int f = (int)first[(g * V) / row];
    float p[V], t[V], o[V], z[V];
    load_vec<V>(end_sel_src(s, f) + g * V, p);
    load_vec<V>(target + g * V, t);
#pragma unroll
    for (int i = 0; i < V; ++i) { o[i] = sign_of(p[i] - t[i]) * gs; z[i] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < s.T) store_vec<V>(s.gimg[k] + g * V, f == k ? o : z);
  }
}

// ------------------------------------------------------------------ SSIM forward (evaluation)
__global__ __launch_bounds__(kThreads) void k_ssim_fwd(SsimArgs s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int plane, tile;
  wg_coords(s.tiles, plane, tile);
  ssim_phase_load(s, plane, tile, threadIdx.x, lds);
  __syncthreads();
  ssim_phase_rows(s, threadIdx.x, lds);
  __syncthreads();
  const float v = ssim_phase_cols(s, tile, threadIdx.x, lds);
  block_reduce_store1(v, s.partials + (size_t)plane * s.tiles + tile);
}

// out[b] = mean over (C,H,W) of the SSIM map of sample b
__global__ __launch_bounds__(kThreads) void k_ssim_finalize(const float* partials, int per_sample, float inv, float* out) {
  float acc = 0.0f;
  for (int k = threadIdx.x; k < per_sample; k += kThreads) acc += partials[(size_t)blockIdx.x * per_sample + k];
  __shared__ float out1;
  block_reduce_store1(acc, &out1);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = out1 * inv;
}

// ------------------------------------------------------------------ SSIM backward (both image gradients, one launch)
__global__ __launch_bounds__(kThreads) void k_ssim_bwd(SsimBwdArgs s) {
  __shared__ __attribute__((aligned(16))) float lds[kSsimBD + 4 * kSsimBMid * kSsimBMidStride];      // 95.7 KB of the CU's 160
  int plane, tile;
  wg_coords(s.tiles, plane, tile);
  ssim_bwd_phase_load(s, plane, tile, threadIdx.x, lds);
  __syncthreads();
  ssim_bwd_phase_rows(s, threadIdx.x, lds);
  __syncthreads();
  ssim_bwd_phase_deriv(s, plane, tile, threadIdx.x, lds);
  __syncthreads();
  ssim_bwd_phase_drows(s, threadIdx.x, lds);
  __syncthreads();
  ssim_bwd_phase_out(s, plane, tile, threadIdx.x, lds);
}

// ------------------------------------------------------------------ planner: candidate sweep
// grid = nblk pixel chunks x ceil(C / kCandPerBlock) candidate groups
__global__ __launch_bounds__(kThreads) void k_candidates_l1(CandArgs a) {
  __shared__ float tab[kCandPerBlock * kTabStride];
  __shared__ float wsum[kCandPerBlock][kThreads / 64];
  const int blk = blockIdx.x % a.nblk, group = blockIdx.x / a.nblk;
  const int c0 = group * kCandPerBlock;
  if ((int)threadIdx.x < kCandPerBlock && c0 + (int)threadIdx.x < a.C)
    cand_build_table(a, c0 + threadIdx.x, tab + threadIdx.x * kTabStride);
  float x[3][kCandPix], tg[3][kCandPix];
  const int npx = cand_load(a, blk, threadIdx.x, x, tg);
  __syncthreads();
  for (int j = 0; j < kCandPerBlock; ++j) {
    if (c0 + j >= a.C) break;
    const float s = wave_sum(cand_eval(a, tab + j * kTabStride, x, tg, npx));
    if ((threadIdx.x & 63) == 0) wsum[j][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < kCandPerBlock && c0 + (int)threadIdx.x < a.C) {
    const int j = threadIdx.x;
    a.partials[(size_t)(c0 + j) * a.nblk + blk] = ((wsum[j][0] + wsum[j][1]) + wsum[j][2]) + wsum[j][3];
  }
}

// many (image, operator) jobs in one launch (a whole beam-search step: every beam x every 1-parameter operator):
// blockIdx.y = job; each job is the single-job kernel above on its own image / operator / candidate rows
constexpr int kMaxCandJobs = 64;
struct MultiCandArgs {
  const float* imgs;      // (n_img, 3, H, W)
  const float* target;    // (3, H, W)
  const float* params;    // (J, C, param_stride)
  float* partials;        // (J, C, nblk)
  int op[kMaxCandJobs];
  int img_index[kMaxCandJobs];
  int C, param_stride, H, W, nblk;
};

__global__ __launch_bounds__(kThreads) void k_candidates_multi_l1(MultiCandArgs m) {
  __shared__ float tab[kCandPerBlock * kTabStride];
  __shared__ float wsum[kCandPerBlock][kThreads / 64];
  const int job = blockIdx.y;
  CandArgs a;
  a.img = m.imgs + (size_t)m.img_index[job] * 3 * m.H * m.W;
  a.target = m.target;
  a.params = m.params + (size_t)job * m.C * m.param_stride;
  a.partials = m.partials + (size_t)job * m.C * m.nblk;
  a.op = m.op[job]; a.C = m.C; a.param_stride = m.param_stride; a.H = m.H; a.W = m.W; a.nblk = m.nblk;
  const int blk = blockIdx.x % a.nblk, group = blockIdx.x / a.nblk;
  const int c0 = group * kCandPerBlock;
  if ((int)threadIdx.x < kCandPerBlock && c0 + (int)threadIdx.x < a.C)
    cand_build_table(a, c0 + threadIdx.x, tab + threadIdx.x * kTabStride);
  float x[3][kCandPix], tg[3][kCandPix];
  const int npx = cand_load(a, blk, threadIdx.x, x, tg);
  __syncthreads();
  for (int j = 0; j < kCandPerBlock; ++j) {
    if (c0 + j >= a.C) break;
    const float s = wave_sum(cand_eval(a, tab + j * kTabStride, x, tg, npx));
    if ((threadIdx.x & 63) == 0) wsum[j][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < kCandPerBlock && c0 + (int)threadIdx.x < a.C) {
    const int j = threadIdx.x;
    a.partials[(size_t)(c0 + j) * a.nblk + blk] = ((wsum[j][0] + wsum[j][1]) + wsum[j][2]) + wsum[j][3];
  }
}

__global__ __launch_bounds__(64) void k_candidates_finalize(const float* partials, int nblk, float inv_n, float* loss) {
  float acc = 0.0f;
  for (int k = threadIdx.x; k < nblk; k += 64) acc += partials[(size_t)blockIdx.x * nblk + k];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[blockIdx.x] = acc * inv_n;
}

// ---------------------------------------------------------------- next-operator choice of the free-running decode
// (models/actor.py:222-236: probs = exp(logp) * (1 - e) + e; probs *= op_mask; probs /= sum + 1e-30; Categorical
// sample or arg-max; op_mask[b][choice] = 0).  ~17 framework launches per decoder step as one: a thread per sample.
// u: one uniform [0,1) number per sample (null: arg-max).  Inverse-CDF draw as actor.sample_categorical: the first
// index whose cumulative weight exceeds u * total; arg-max if rounding pushes it past the end.
__global__ __launch_bounds__(64) void k_choose_op(const float* logp, float* op_mask, const float* u, float explore,
                                                 long long* pred_op, int* exec_op, int B, int n) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float* lp = logp + (size_t)b * n;
  float* m = op_mask + (size_t)b * n;
  float p[32];
  float total = 0.0f;
  for (int k = 0; k < n; ++k) { p[k] = (expf(lp[k]) * (1.0f - explore) + explore) * m[k]; total += p[k]; }
  const float inv = 1.0f / (total + 1e-30f);
  int best = 0;
  float bestv = p[0] * inv;
  float cdf = 0.0f, cdf_total = 0.0f;
  for (int k = 0; k < n; ++k) { p[k] *= inv; cdf_total += p[k]; if (p[k] > bestv) { bestv = p[k]; best = k; } }
  int choice = best;
  if (u) {
    const float thr = u[b] * cdf_total;
    int idx = 0;
    for (int k = 0; k < n; ++k) { cdf += p[k]; idx += (cdf <= thr) ? 1 : 0; }
    choice = idx >= n ? best : idx;
  }
  pred_op[b] = choice;
  exec_op[b] = choice - 3;                 // executor index = operator-vocabulary id - 3 (actor.py:165)
  m[choice] = 0.0f;                        // an operator is used at most once (actor.py:235-236)
}

// ------------------------------------------------------------------ attention core
// One workgroup (4 waves) per sample; L <= 64 encoder rows of D <= 1024 columns.
// k_attn_fwd (any D % 64 == 0): wave w takes encoder rows w, w+4, ... (lanes stride the D columns, shuffle reduction); softmax
// over all L rows from LDS; mix: one thread per column.  One memory round trip per encoder row of a wave: 22.8 us for the
// actor's 2.2 MB (L = 17, D = 512).
// k_attn_fwd_wide (D % 256 == 0): a row is D / 4 column quads = D / 256 whole waves; the 4 waves form R = 1 / 2 / 4 row groups,
// thread t owns quad t % (D/4) of the rows of group t / (D/4).  16-byte loads, 8 rows' loads in flight per thread, a wave
// reduction per row, the waves of a row meet in LDS; the mix keeps the same ownership.  Three round trips per launch.
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(kThreads) void k_attn_fwd_wide(const float* q, const float* ctx, float* attn, float* mix,
                                                            int B, int L, int D) {
  __shared__ float ws[4][64];                                         // per wave: its part of every row's score
  __shared__ float sc[64];
  __shared__ float4 red[kThreads];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Q = D >> 2, wpr = Q >> 6;                                 // column quads per row; waves per row (1, 2, 3, 4)
  const int R = (kThreads / 64) / wpr;                                // row groups (4, 2, 1, 1); waves >= R * wpr idle in the row phases
  const int rg = wave / wpr, cq = (wave % wpr) * 64 + lane;
  const bool active = rg < R;
  const float* qb = q + (size_t)b * D;
  const float* cb = ctx + (size_t)b * L * D;
  const float4 qv = active ? ld4g(qb + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int l0 = rg; l0 < L; l0 += 8 * R) {                           // (wave-uniform bounds)
    float4 cv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int l = l0 + R * k;
      cv[k] = (active && l < L) ? ld4g(cb + (size_t)l * D + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int l = l0 + R * k;
      const float s = wave_sum((qv.x * cv[k].x + qv.y * cv[k].y) + (qv.z * cv[k].z + qv.w * cv[k].w));
      if (active && lane == 0 && l < L) ws[wave][l] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < L) {                                         // the waves of a row, in order
    const int l = threadIdx.x, g = l % R;
    float s = 0.0f;
    for (int w = 0; w < wpr; ++w) s += ws[g * wpr + w][l];
    sc[l] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, sc[l]);
  float den = 0.0f;
  for (int l = 0; l < L; ++l) den += expf(sc[l] - mx);
  __syncthreads();
  if ((int)threadIdx.x < L) {
    const float pr = expf(sc[threadIdx.x] - mx) / den;
    sc[threadIdx.x] = pr;
    attn[(size_t)b * L + threadIdx.x] = pr;
  }
  __syncthreads();
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (active)
    for (int l0 = rg; l0 < L; l0 += 8 * R) {
      float4 cv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int l = l0 + R * k;
        cv[k] = l < L ? ld4g(cb + (size_t)l * D + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int l = l0 + R * k;
        const float pl = l < L ? sc[l] : 0.0f;
        acc.x += pl * cv[k].x; acc.y += pl * cv[k].y; acc.z += pl * cv[k].z; acc.w += pl * cv[k].w;
      }
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < Q) {                                         // the row groups of a column quad, in order
    float4 a = red[threadIdx.x];                                      // (group 0's thread for quad t is thread t)
    for (int g = 1; g < R; ++g) { const float4 v = red[g * Q + threadIdx.x]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    *reinterpret_cast<float4*>(mix + (size_t)b * D + 4 * threadIdx.x) = a;
  }
}

__global__ __launch_bounds__(kThreads) void k_attn_fwd(const float* q, const float* ctx, float* attn, float* mix,
                                                       int B, int L, int D) {
  __shared__ float sc[64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qb = q + (size_t)b * D;
  const float* cb = ctx + (size_t)b * L * D;
  for (int l = wave; l < L; l += kThreads / 64) {
    float part = 0.0f;
    for (int e = lane; e < D; e += 64) part += qb[e] * cb[(size_t)l * D + e];
    const float s = wave_sum(part);
    if (lane == 0) sc[l] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, sc[l]);
  float den = 0.0f;
  for (int l = 0; l < L; ++l) den += expf(sc[l] - mx);
  __syncthreads();
  if ((int)threadIdx.x < L) {
    const float pr = expf(sc[threadIdx.x] - mx) / den;
    sc[threadIdx.x] = pr;
    attn[(size_t)b * L + threadIdx.x] = pr;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < D; e += kThreads) {
    float acc = 0.0f;
    for (int l = 0; l < L; ++l) acc += sc[l] * cb[(size_t)l * D + e];
    mix[(size_t)b * D + e] = acc;
  }
}

// backward in the same layout (D % 256 == 0): d loss / d prob_l as the forward's scores with gmix in place of q, the softmax
// backward from LDS, then every thread writes its column quad of gctx for its rows and adds up its part of gq
__global__ __launch_bounds__(kThreads) void k_attn_bwd_wide(const float* q, const float* ctx, const float* attn, const float* gmix,
                                                            const float* gattn, float* gq, float* gctx, int B, int L, int D) {
  __shared__ float ws[4][64];
  __shared__ float pr[64], ga[64], gs[64];
  __shared__ float4 red[kThreads];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Q = D >> 2, wpr = Q >> 6;
  const int R = (kThreads / 64) / wpr;
  const int rg = wave / wpr, cq = (wave % wpr) * 64 + lane;
  const bool active = rg < R;
  const float* cb = ctx + (size_t)b * L * D;
  const float4 gv = active ? ld4g(gmix + (size_t)b * D + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const float4 qv = active ? ld4g(q + (size_t)b * D + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if ((int)threadIdx.x < L) pr[threadIdx.x] = attn[(size_t)b * L + threadIdx.x];
  for (int l0 = rg; l0 < L; l0 += 8 * R) {
    float4 cv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int l = l0 + R * k;
      cv[k] = (active && l < L) ? ld4g(cb + (size_t)l * D + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int l = l0 + R * k;
      const float s = wave_sum((gv.x * cv[k].x + gv.y * cv[k].y) + (gv.z * cv[k].z + gv.w * cv[k].w));
      if (active && lane == 0 && l < L) ws[wave][l] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < L) {
    const int l = threadIdx.x, g = l % R;
    float s = 0.0f;
    for (int w = 0; w < wpr; ++w) s += ws[g * wpr + w][l];
    ga[l] = s + (gattn ? gattn[(size_t)b * L + l] : 0.0f);
  }
  __syncthreads();
  float dot = 0.0f;
  for (int l = 0; l < L; ++l) dot += pr[l] * ga[l];
  if ((int)threadIdx.x < L) gs[threadIdx.x] = pr[threadIdx.x] * (ga[threadIdx.x] - dot);   // softmax backward
  __syncthreads();
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (active)
    for (int l0 = rg; l0 < L; l0 += 8 * R) {
      float4 cv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int l = l0 + R * k;
        cv[k] = l < L ? ld4g(cb + (size_t)l * D + 4 * cq) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int l = l0 + R * k;
        if (l < L) {
          const float pl = pr[l], sl = gs[l];
          acc.x += sl * cv[k].x; acc.y += sl * cv[k].y; acc.z += sl * cv[k].z; acc.w += sl * cv[k].w;
          *reinterpret_cast<float4*>(gctx + ((size_t)b * L + l) * D + 4 * cq) =
              make_float4(pl * gv.x + sl * qv.x, pl * gv.y + sl * qv.y, pl * gv.z + sl * qv.z, pl * gv.w + sl * qv.w);
        }
      }
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < Q) {
    float4 a = red[threadIdx.x];
    for (int g = 1; g < R; ++g) { const float4 v = red[g * Q + threadIdx.x]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    *reinterpret_cast<float4*>(gq + (size_t)b * D + 4 * threadIdx.x) = a;
  }
}

__global__ __launch_bounds__(kThreads) void k_attn_bwd(const float* q, const float* ctx, const float* attn,
                                                       const float* gmix, const float* gattn, float* gq, float* gctx,
                                                       int B, int L, int D) {
  __shared__ float pr[64], ga[64], gs[64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qb = q + (size_t)b * D;
  const float* cb = ctx + (size_t)b * L * D;
  const float* gm = gmix + (size_t)b * D;
  for (int l = wave; l < L; l += kThreads / 64) {       // d loss / d prob_l
    float part = 0.0f;
    for (int e = lane; e < D; e += 64) part += gm[e] * cb[(size_t)l * D + e];
    const float s = wave_sum(part);
    if (lane == 0) ga[l] = s + (gattn ? gattn[(size_t)b * L + l] : 0.0f);
  }
  if ((int)threadIdx.x < L) pr[threadIdx.x] = attn[(size_t)b * L + threadIdx.x];
  __syncthreads();
  float dot = 0.0f;
  for (int l = 0; l < L; ++l) dot += pr[l] * ga[l];
  if ((int)threadIdx.x < L) gs[threadIdx.x] = pr[threadIdx.x] * (ga[threadIdx.x] - dot);   // softmax backward
  __syncthreads();
  for (int e = threadIdx.x; e < D; e += kThreads) {
    const float qe = qb[e], ge = gm[e];
    float acc = 0.0f;
    for (int l = 0; l < L; ++l) {
      acc += gs[l] * cb[(size_t)l * D + e];
      gctx[((size_t)b * L + l) * D + e] = pr[l] * ge + gs[l] * qe;
    }
    gq[(size_t)b * D + e] = acc;
  }
}

// ------------------------------------------------------------------ host-side launch logic
int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

Geometry launch_geometry(int B, int H, int W) {
  constexpr int forced = 0;                            // (pixel groups per thread: derived from the image size)
  return t2o::geometry(B, H, W, forced);
}

// Workspace = [raw parameter sums][loss partials].  Block rows per sample = the larger of the
// per-operator geometry and the fused-chain geometry; kMaxChainSlots floats per block row.
size_t ws_block_rows(const Geometry& g, int B, int H, int W) {
  int vec, iters, nblk;
  chain_geometry(B, H, W, 0, vec, iters, nblk, 1);   // the finer (vec 1) geometry bounds both
  return (size_t)(g.nblk_max > nblk ? g.nblk_max : nblk);
}
size_t ws_partials_floats(const Geometry& g, int B, int H, int W) { return (size_t)B * ws_block_rows(g, B, H, W) * kMaxChainSlots; }
size_t ws_loss_floats(const Geometry& g, int B, int H, int W) { return (size_t)B * ws_block_rows(g, B, H, W); }

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return T2O_ELAUNCH;
  }
  return T2O_OK;
}

bool op_supported(int op) { return op == OP_IDENTITY || (op >= 0 && op <= 7 && op != OP_INPAINT); }

#define T2O_POINT_OPS(KERNEL, V, M, L)                                                        \
  switch (op) {                                                                             \
    case OP_IDENTITY:   KERNEL<OP_IDENTITY, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break;   \
    case OP_BRIGHTNESS: KERNEL<OP_BRIGHTNESS, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break; \
    case OP_CONTRAST:   KERNEL<OP_CONTRAST, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break;   \
    case OP_SATURATION: KERNEL<OP_SATURATION, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break; \
    case OP_COLOR:      KERNEL<OP_COLOR, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break;      \
    case OP_TONE:       KERNEL<OP_TONE, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break;       \
    case OP_WHITE:      KERNEL<OP_WHITE, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break;      \
    case OP_DYNAMIC:    KERNEL<OP_DYNAMIC, V, M, L><<<grid, kThreads, 0, st>>>(a, nblk); break;    \
    default: break;                                                                         \
  }
#define T2O_POINT_CASES(KERNEL, V)                                    \
  if (masked) {                                                       \
    if (l1) { T2O_POINT_OPS(KERNEL, V, true, true) } else { T2O_POINT_OPS(KERNEL, V, true, false) }   \
  } else {                                                            \
    if (l1) { T2O_POINT_OPS(KERNEL, V, false, true) } else { T2O_POINT_OPS(KERNEL, V, false, false) } \
  }

void launch_point_fwd(const OpArgs& a, const Geometry& g, hipStream_t st) {
  const int op = a.op, nblk = g.nblk_point;
  const bool masked = a.mask_ch != 0, l1 = a.target != nullptr;
  const unsigned grid = (unsigned)a.B * nblk;
  if (g.vec == 4) { T2O_POINT_CASES(k_point_fwd, 4) } else { T2O_POINT_CASES(k_point_fwd, 1) }
}
void launch_point_bwd(const OpArgs& a, const Geometry& g, hipStream_t st) {
  const int op = a.op, nblk = g.nblk_point;
  const bool masked = a.mask_ch != 0, l1 = a.target != nullptr;
  const unsigned grid = (unsigned)a.B * nblk;
  if (g.vec == 4) { T2O_POINT_CASES(k_point_bwd, 4) } else { T2O_POINT_CASES(k_point_bwd, 1) }
}
// which stencil kernels run, and how many per-sample partial rows they write
bool sharp_uses_strips(const Geometry& g) {
  constexpr int mode = 1;                                       // LDS-free strips wherever W % 4 == 0 (tile kernels: ragged widths)
  return mode && g.vec_tile == 4;
}
bool sharp_bwd_uses_strips(const OpArgs&, const Geometry& g) { return sharp_uses_strips(g); }
int sharp_fwd_blocks(const OpArgs&, const Geometry& g) { return sharp_uses_strips(g) ? g.nblk_strip_fwd : g.nblk_sharp; }
int sharp_bwd_blocks(const OpArgs& a, const Geometry& g) { return sharp_bwd_uses_strips(a, g) ? g.nblk_strip_bwd : g.nblk_sharp; }

void launch_sharp_fwd(const OpArgs& a, const Geometry& g, hipStream_t st) {
  if (sharp_uses_strips(g)) {
    const int nseg = (a.W + 255) / 256, nblk = sharp_fwd_blocks(a, g);
    const unsigned grid = (unsigned)a.B * nblk;
    const bool dyn = a.op == OP_DYNAMIC;
    constexpr int R = kFwdStripRows;
    if (nseg > 1) { if (dyn) k_sharp_fwd_strip<true, true, R><<<grid, kThreads, 0, st>>>(a, nblk, nseg); else k_sharp_fwd_strip<false, true, R><<<grid, kThreads, 0, st>>>(a, nblk, nseg); }
    else          { if (dyn) k_sharp_fwd_strip<true, false, R><<<grid, kThreads, 0, st>>>(a, nblk, nseg); else k_sharp_fwd_strip<false, false, R><<<grid, kThreads, 0, st>>>(a, nblk, nseg); }
    return;
  }
  const unsigned grid = (unsigned)a.B * g.nblk_sharp;
  const size_t lds = sizeof(float) * sharp_fwd_lds_floats();
  const bool dyn = a.op == OP_DYNAMIC;
  if (g.vec_tile == 4) {
    if (dyn) k_sharp_fwd<true, 4><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
    else k_sharp_fwd<false, 4><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
  } else {
    if (dyn) k_sharp_fwd<true, 1><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
    else k_sharp_fwd<false, 1><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
  }
}
void launch_sharp_bwd(const OpArgs& a, const Geometry& g, hipStream_t st) {
  if (sharp_bwd_uses_strips(a, g)) {
    const int nseg = strip_bwd_segments(a.W), nblk = g.nblk_strip_bwd;
    const unsigned grid = (unsigned)a.B * nblk;
    const bool dyn = a.op == OP_DYNAMIC;
    if (a.loss_partials) {                      // value-and-gradient (run_bwd checked: fused L1, one operator, no mask)
      if (nseg > 1) k_sharp_bwd_strip<false, true, false, true><<<grid, kThreads, 0, st>>>(a, nblk, nseg);
      else k_sharp_bwd_strip<false, false, false, true><<<grid, kThreads, 0, st>>>(a, nblk, nseg);
      return;
    }
#define T2O_BWD_STRIP(D, Wd) \
    { if (a.mask_ch) k_sharp_bwd_strip<D, Wd, true><<<grid, kThreads, 0, st>>>(a, nblk, nseg); \
      else k_sharp_bwd_strip<D, Wd, false><<<grid, kThreads, 0, st>>>(a, nblk, nseg); }
    if (nseg > 1) { if (dyn) T2O_BWD_STRIP(true, true) else T2O_BWD_STRIP(false, true) }
    else          { if (dyn) T2O_BWD_STRIP(true, false) else T2O_BWD_STRIP(false, false) }
#undef T2O_BWD_STRIP
    return;
  }
  const unsigned grid = (unsigned)a.B * g.nblk_sharp;
  const size_t lds = sizeof(float) * sharp_bwd_lds_floats(a.mask_ch);
  const bool dyn = a.op == OP_DYNAMIC;
  if (g.vec_tile == 4) {
    if (dyn) k_sharp_bwd<true, 4><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
    else k_sharp_bwd<false, 4><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
  } else {
    if (dyn) k_sharp_bwd<true, 1><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
    else k_sharp_bwd<false, 1><<<grid, kThreads, lds, st>>>(a, g.nblk_sharp);
  }
}

int check_common(int op, const int* op_id, const float* img, const float* param, int param_stride,
                 const float* mask, int mask_ch, int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "B, H, W must be positive");
  if (!img) return fail(T2O_EINVAL, "img is null");
  if (op == OP_DYNAMIC) {
    if (!op_id) return fail(T2O_EINVAL, "op_id is null");
    if (!param || param_stride < kMaxParam) return fail(T2O_EINVAL, "apply: param rows must be padded to >= 24 floats");
  } else {
    if (!op_supported(op)) return fail(T2O_EUNSUPPORTED, "operator index not supported (4 = inpaint needs EdgeConnect)");
    if (op >= 0 && op != OP_WHITE && (!param || param_stride < op_num_params(op)))
      return fail(T2O_EINVAL, "param is null or param_stride < number of operator parameters");
  }
  if ((mask == nullptr) != (mask_ch == 0) || !(mask_ch == 0 || mask_ch == 1 || mask_ch == 3))
    return fail(T2O_EINVAL, "mask_ch must be 0 (mask NULL), 1 or 3");
  if ((size_t)B * (size_t)sharp_num_tiles(H, W) > 0x7fffffffull || (size_t)H * W / 1 > 0x7fffffffull)
    return fail(T2O_EINVAL, "image too large for the launch grid");
  return T2O_OK;
}

// forward of one operator (static or per-sample), optionally fused with the L1 loss
int run_fwd(int op, const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
            int mask_ch, const float* target, float* out, float* loss, void* ws, size_t ws_bytes, int B, int H, int W,
            void* stream) {
  if (int rc = check_common(op, op_id, img, param, param_stride, mask, mask_ch, B, H, W)) return rc;
  if (!out) return fail(T2O_EINVAL, "out is null");
  const Geometry g = launch_geometry(B, H, W);
  OpArgs a;
  memset(&a, 0, sizeof(a));
  a.img = img; a.param = param; a.mask = mask; a.op_id = op_id; a.target = target; a.out = out;
  a.op = op; a.param_stride = param_stride; a.mask_ch = mask_ch; a.B = B; a.H = H; a.W = W;
  a.iters = g.iters; a.nblk_max = g.nblk_max;
  a.inv_n = 1.0f / ((float)B * 3.0f * (float)H * (float)W);
  if (target) {
    if (!loss) return fail(T2O_EINVAL, "loss is null");
    if (!ws || ws_bytes < t2o_workspace_bytes(B, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
    a.loss_partials = (float*)ws + ws_partials_floats(g, B, H, W);
  }
  hipStream_t st = (hipStream_t)stream;
  if (op != OP_SHARPNESS) launch_point_fwd(a, g, st);
  if (op == OP_SHARPNESS || op == OP_DYNAMIC) launch_sharp_fwd(a, g, st);
  if (target) k_finalize_loss<<<1, kThreads, 0, st>>>(a, loss, g.nblk_point, sharp_fwd_blocks(a, g));
  return check_launch("operator forward");
}

int run_bwd(int op, const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
            int mask_ch, const float* gout, const float* target, const float* gloss, float* gimg, float* gparam,
            int gparam_stride, void* ws, size_t ws_bytes, int B, int H, int W, void* stream,
            float* partials_region = nullptr, int* nblk_out = nullptr, float* value_out = nullptr, float* value_loss = nullptr) {
  // value_loss (sharpness + L1 target on the strip kernels only): the backward also leaves the loss -- and the operator's
  // output image in value_out when that is given -- so a value-and-gradient call needs no forward launch for this operator
  if (int rc = check_common(op, op_id, img, param, param_stride, mask, mask_ch, B, H, W)) return rc;
  if (!target && !gout) return fail(T2O_EINVAL, "gout is null");
  if (target && !gloss) return fail(T2O_EINVAL, "gloss is null");
  if (!ws || ws_bytes < t2o_workspace_bytes(B, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  if (gparam && gparam_stride < (op == OP_DYNAMIC ? kMaxParam : op_num_params(op)))
    return fail(T2O_EINVAL, "gparam_stride too small");
  Geometry g = launch_geometry(B, H, W);
  if (op == OP_COLOR || op == OP_TONE) {
    // the 24 / 8 per-thread raw sums cost a wave + LDS reduction per workgroup: give each thread 4x the
    // pixels (48 vs 60 us for the color curve at bs=64 256x256); block rows only shrink, so the
    // workspace bound still holds
    constexpr int mult = 4;
    const size_t groups = (size_t)H * W / g.vec;
    int it = g.iters * (mult > 0 ? mult : 1);
    if (it > 8) it = 8;
    g.iters = it;
    g.nblk_point = (int)((groups + (size_t)kThreads * it - 1) / ((size_t)kThreads * it));
  }
  OpArgs a;
  memset(&a, 0, sizeof(a));
  a.img = img; a.param = param; a.mask = mask; a.op_id = op_id; a.gout = gout; a.target = target; a.gloss = gloss;
  a.gimg = gimg; a.partials = partials_region ? partials_region : (float*)ws;
  a.op = op; a.param_stride = param_stride; a.mask_ch = mask_ch; a.B = B; a.H = H; a.W = W;
  a.iters = g.iters; a.nblk_max = g.nblk_max;
  a.inv_n = 1.0f / ((float)B * 3.0f * (float)H * (float)W);
  hipStream_t st = (hipStream_t)stream;
  if (value_loss) {
    if (op != OP_SHARPNESS || !target || mask_ch || !sharp_bwd_uses_strips(a, g)) return fail(T2O_EUNSUPPORTED, "value-and-gradient: unmasked sharpness on the strip kernels only");
    a.out = value_out;
    a.loss_partials = (float*)ws + ws_partials_floats(g, B, H, W);
  }
  if (partials_region) {                        // the caller finalises every operator of the sequence at once
    if (op != OP_SHARPNESS) launch_point_bwd(a, g, st);
    else launch_sharp_bwd(a, g, st);
    if (nblk_out) *nblk_out = (op == OP_SHARPNESS) ? sharp_bwd_blocks(a, g) : g.nblk_point;
    return check_launch("operator backward");
  }
  if (op != OP_SHARPNESS) launch_point_bwd(a, g, st);
  if (op == OP_SHARPNESS || op == OP_DYNAMIC) launch_sharp_bwd(a, g, st);
  if (gparam && op != OP_IDENTITY)
    k_finalize_params<<<B, kThreads, 0, st>>>(a, gparam, gparam_stride, g.nblk_point, sharp_bwd_blocks(a, g));
  if (value_loss) k_finalize_loss<<<1, kThreads, 0, st>>>(a, value_loss, g.nblk_point, sharp_bwd_blocks(a, g));
  return check_launch("operator backward");
}

}  // namespace

// ==================================================================== C ABI
namespace {
__global__ __launch_bounds__(256) void k_zero_floats(float* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.0f;
}
}  // namespace

extern "C" {

int t2o_abi_version(void) { return 4; }   // 4: round 3 additions (accumulate forms, 1x1 / any-size convolutions, LSTM, choose_op, run-time specialisation)
// (t2o_source_digest: t2o_stamp.hip -- the only file compiled with the digest, so that an edit elsewhere recompiles one file)
const char* t2o_last_error(void) { return g_err; }

int t2o_op_num_params(int op) { return (op >= 0 && op <= 7) ? op_num_params(op) : -1; }

size_t t2o_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const Geometry g = launch_geometry(B, H, W);
  size_t f = ws_partials_floats(g, B, H, W) + ws_loss_floats(g, B, H, W);
  // plain L1 over the same number of floats uses the partial area too
  const size_t n = (size_t)B * 3 * H * W;
  const size_t l1blk = n / ((size_t)kThreads * 8) + 2;      // k_l1_fwd partials, worst case V = 1
  if (f < l1blk) f = l1blk;
  return f * sizeof(float);
}

int t2o_op_fwd(int op, const float* img, const float* param, int param_stride, const float* mask, int mask_ch,
               float* out, int B, int H, int W, void* stream) {
  if (op == OP_DYNAMIC) return fail(T2O_EUNSUPPORTED, "use t2o_apply_fwd for per-sample operators");
  return run_fwd(op, nullptr, img, param, param_stride, mask, mask_ch, nullptr, out, nullptr, nullptr, 0, B, H, W, stream);
}

int t2o_op_bwd(int op, const float* img, const float* param, int param_stride, const float* mask, int mask_ch,
               const float* gout, float* gimg, float* gparam, int gparam_stride, void* workspace,
               size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (op == OP_DYNAMIC) return fail(T2O_EUNSUPPORTED, "use t2o_apply_bwd for per-sample operators");
  return run_bwd(op, nullptr, img, param, param_stride, mask, mask_ch, gout, nullptr, nullptr, gimg, gparam,
                 gparam_stride, workspace, workspace_bytes, B, H, W, stream);
}

int t2o_apply_fwd(const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
                  int mask_ch, float* out, int B, int H, int W, void* stream) {
  return run_fwd(OP_DYNAMIC, op_id, img, param, param_stride, mask, mask_ch, nullptr, out, nullptr, nullptr, 0, B, H, W,
                 stream);
}

int t2o_apply_bwd(const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
                  int mask_ch, const float* gout, float* gimg, float* gparam, int gparam_stride, void* workspace,
                  size_t workspace_bytes, int B, int H, int W, void* stream) {
  return run_bwd(OP_DYNAMIC, op_id, img, param, param_stride, mask, mask_ch, gout, nullptr, nullptr, gimg, gparam,
                 gparam_stride, workspace, workspace_bytes, B, H, W, stream);
}

int t2o_op_fwd_l1(int op, const float* img, const float* param, int param_stride, const float* mask, int mask_ch,
                  const float* target, float* out, float* loss, void* workspace, size_t workspace_bytes, int B, int H,
                  int W, void* stream) {
  if (!target) return fail(T2O_EINVAL, "target is null");
  if (op == OP_DYNAMIC) return fail(T2O_EUNSUPPORTED, "fused L1 takes a static operator");
  return run_fwd(op, nullptr, img, param, param_stride, mask, mask_ch, target, out, loss, workspace, workspace_bytes, B,
                 H, W, stream);
}

int t2o_op_bwd_l1(int op, const float* img, const float* param, int param_stride, const float* mask, int mask_ch,
                  const float* target, const float* gloss, float* gimg, float* gparam, int gparam_stride,
                  void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (!target) return fail(T2O_EINVAL, "target is null");
  if (op == OP_DYNAMIC) return fail(T2O_EUNSUPPORTED, "fused L1 takes a static operator");
  return run_bwd(op, nullptr, img, param, param_stride, mask, mask_ch, nullptr, target, gloss, gimg, gparam,
                 gparam_stride, workspace, workspace_bytes, B, H, W, stream);
}

int t2o_l1_fwd(const float* pred, const float* target, float* loss, size_t n, void* workspace, size_t workspace_bytes,
               void* stream) {
  if (!pred || !target || !loss || n == 0) return fail(T2O_EINVAL, "l1_fwd: null pointer or n == 0");
  const int V = (n % 4 == 0) ? 4 : 1;
  const int iters = 8;
  const size_t groups = n / V;
  const size_t nblk = (groups + (size_t)kThreads * iters - 1) / ((size_t)kThreads * iters);
  if (!workspace || workspace_bytes < nblk * sizeof(float)) return fail(T2O_EWORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  if (V == 4) k_l1_fwd<4><<<(unsigned)nblk, kThreads, 0, st>>>(pred, target, partial, n, iters);
  else k_l1_fwd<1><<<(unsigned)nblk, kThreads, 0, st>>>(pred, target, partial, n, iters);
  k_l1_finalize<<<1, kThreads, 0, st>>>(partial, (int)nblk, 1.0f / (float)n, loss);
  return check_launch("l1 forward");
}

int t2o_l1_bwd(const float* pred, const float* target, const float* gloss, float* gpred, size_t n, void* stream) {
  if (!pred || !target || !gloss || !gpred || n == 0) return fail(T2O_EINVAL, "l1_bwd: null pointer or n == 0");
  const int V = (n % 4 == 0) ? 4 : 1;
  const int iters = 4;
  const size_t groups = n / V;
  const size_t nblk = (groups + (size_t)kThreads * iters - 1) / ((size_t)kThreads * iters);
  hipStream_t st = (hipStream_t)stream;
  if (V == 4) k_l1_bwd<4><<<(unsigned)nblk, kThreads, 0, st>>>(pred, target, gloss, gpred, n, 1.0f / (float)n, iters);
  else k_l1_bwd<1><<<(unsigned)nblk, kThreads, 0, st>>>(pred, target, gloss, gpred, n, 1.0f / (float)n, iters);
  return check_launch("l1 backward");
}

int t2o_end_select_l1_fwd(const float* const* imgs, int T, const long long* first, const float* target, float* loss, int B,
                          size_t row, void* workspace, size_t workspace_bytes, void* stream) {
  if (!imgs || !first || !target || !loss || T < 1 || T > 8 || B <= 0 || row == 0) return fail(T2O_EINVAL, "end_select_l1_fwd: null pointer or bad shape (1 <= T <= 8)");
  EndSelArgs s = {};
  s.T = T;
  for (int t = 0; t < T; ++t) { if (!imgs[t]) return fail(T2O_EINVAL, "end_select_l1_fwd: null image"); s.img[t] = imgs[t]; }
  const size_t n = (size_t)B * row;
  const int V = (row % 4 == 0) ? 4 : 1;
  const int iters = 8;
  const size_t groups = n / V;
  const size_t nblk = (groups + (size_t)kThreads * iters - 1) / ((size_t)kThreads * iters);
  if (!workspace || workspace_bytes < nblk * sizeof(float)) return fail(T2O_EWORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  if (V == 4) k_end_l1_fwd<4><<<(unsigned)nblk, kThreads, 0, st>>>(s, first, target, partial, n, row, iters);
  else k_end_l1_fwd<1><<<(unsigned)nblk, kThreads, 0, st>>>(s, first, target, partial, n, row, iters);
  k_l1_finalize<<<1, kThreads, 0, st>>>(partial, (int)nblk, 1.0f / (float)n, loss);
  return check_launch("END select + l1 forward");
}

int t2o_end_select_l1_bwd(const float* const* imgs, float* const* gimgs, int T, const long long* first, const float* target,
                          const float* gloss, int B, size_t row, void* stream) {
  if (!imgs || !gimgs || !first || !target || !gloss || T < 1 || T > 8 || B <= 0 || row == 0)
    return fail(T2O_EINVAL, "end_select_l1_bwd: null pointer or bad shape (1 <= T <= 8)");
  EndSelArgs s = {};
  s.T = T;
  for (int t = 0; t < T; ++t) {
    if (!imgs[t] || !gimgs[t]) return fail(T2O_EINVAL, "end_select_l1_bwd: null image");
    s.img[t] = imgs[t]; s.gimg[t] = gimgs[t];
  }
  const size_t n = (size_t)B * row;
  const int V = (row % 4 == 0) ? 4 : 1;
  const int iters = 4;
  const size_t groups = n / V;
  const size_t nblk = (groups + (size_t)kThreads * iters - 1) / ((size_t)kThreads * iters);
  hipStream_t st = (hipStream_t)stream;
  if (V == 4) k_end_l1_bwd<4><<<(unsigned)nblk, kThreads, 0, st>>>(s, first, target, gloss, n, row, 1.0f / (float)n, iters);
  else k_end_l1_bwd<1><<<(unsigned)nblk, kThreads, 0, st>>>(s, first, target, gloss, n, row, 1.0f / (float)n, iters);
  return check_launch("END select + l1 backward");
}

int t2o_sequence_fwd(const int* ops, int K, const float* img, const float* params, const float* target, float* acts,
                     float* loss, void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (!ops || K <= 0 || !params || !acts) return fail(T2O_EINVAL, "sequence_fwd: null pointer or K <= 0");
  const size_t img_floats = (size_t)B * 3 * H * W;
  const float* cur = img;
  for (int k = 0; k < K; ++k) {
    float* out = acts + (size_t)k * img_floats;
    const float* p = params + (size_t)k * B * kMaxParam;
    const bool last = (k == K - 1) && target;
    const int rc = run_fwd(ops[k], nullptr, cur, p, kMaxParam, nullptr, 0, last ? target : nullptr, out,
                           last ? loss : nullptr, workspace, workspace_bytes, B, H, W, stream);
    if (rc) return rc;
    cur = out;
  }
  return T2O_OK;
}

int t2o_sequence_bwd(const int* ops, int K, const float* img, const float* params, const float* target,
                     const float* acts, const float* gloss, float* gimg, float* gparams, float* gbuf, void* workspace,
                     size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (!ops || K <= 0 || !params || !acts || !target || !gloss || !gparams || !gbuf)
    return fail(T2O_EINVAL, "sequence_bwd: null pointer or K <= 0");
  if (!workspace || workspace_bytes < t2o_workspace_bytes(B, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  const size_t img_floats = (size_t)B * 3 * H * W;
  const Geometry g = launch_geometry(B, H, W);
  // one finalize launch for the whole sequence: the workspace holds kMaxChainSlots floats per block row,
  // i.e. kMaxChain regions of kRedSlots -- operator k keeps its rows in region k
  const bool multi = K <= kMaxChain;
  const size_t region = (size_t)B * ws_block_rows(g, B, H, W) * kRedSlots;
  MultiFinalize mf;
  memset(&mf, 0, sizeof(mf));
  mf.params = params; mf.gparams = gparams; mf.partials = (const float*)workspace; mf.region_floats = region;
  mf.K = K; mf.B = B; mf.nblk_max = g.nblk_max;
  const float* gcur = nullptr;
  for (int k = K - 1; k >= 0; --k) {
    const float* in = k == 0 ? img : acts + (size_t)(k - 1) * img_floats;
    const float* p = params + (size_t)k * B * kMaxParam;
    float* gp = gparams + (size_t)k * B * kMaxParam;
    float* gout_next = (k == 0) ? gimg : gbuf + (size_t)(k & 1) * img_floats;
    const bool last = k == K - 1;
    float* reg = (multi && ops[k] != OP_IDENTITY) ? (float*)workspace + (size_t)k * region : nullptr;
    int nb = 0;
    const int rc = run_bwd(ops[k], nullptr, in, p, kMaxParam, nullptr, 0, last ? nullptr : gcur, last ? target : nullptr,
                           last ? gloss : nullptr, gout_next, gp, kMaxParam, workspace, workspace_bytes, B, H, W, stream,
                           reg, &nb);
    if (rc) return rc;
    if (multi) { mf.ops[k] = ops[k]; mf.nblk[k] = nb; }
    gcur = gout_next;
  }
  if (multi) {
    k_finalize_params_multi<<<(unsigned)(K * B), kThreads, 0, (hipStream_t)stream>>>(mf);
    return check_launch("sequence backward finalize");
  }
  return T2O_OK;
}

int t2o_fused_sequence_buffers(const int* ops, int K) {
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  return ns < 0 ? -1 : ns - 1;
}

}  // extern "C" (templates need C++ linkage)

// operator lists with a compile-time instantiation of the backward: the benchmark / planner sequences
// (BASELINE.json configs[1] and configs[4] without their closing sharpness)
using SeqCfg2 = StaticChain<OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_COLOR, OP_TONE>;
using SeqCfg5 = StaticChain<OP_TONE, OP_COLOR, OP_TONE, OP_COLOR, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION>;

template <class SEQ>
static bool chain_is(const ChainArgs& a) {
  if (a.K != SEQ::K) return false;
  for (int k = 0; k < SEQ::K; ++k)
    if (a.ops[k] != SEQ::ops[k]) return false;
  return true;
}
template <class SEQ, bool SV_LDS, int MINW>
static void launch_static_bwd(ChainArgs& a, bool l1, hipStream_t st) {
  const unsigned grid = (unsigned)a.B * a.nblk;
  const size_t lds = sizeof(float) * ((size_t)a.bin_off[kMaxChain] * kAccStride + (SV_LDS ? chain_save_floats<1>(SEQ::K) : 0));
  if (l1 && a.loss_partials) k_chain_bwd_static<true, SEQ, SV_LDS, MINW, true><<<grid, kThreads, lds, st>>>(a);      // value-and-gradient
  else if (l1) k_chain_bwd_static<true, SEQ, SV_LDS, MINW><<<grid, kThreads, lds, st>>>(a);
  else k_chain_bwd_static<false, SEQ, SV_LDS, MINW><<<grid, kThreads, lds, st>>>(a);
}
template <class SEQ>
static void launch_static_fwd(ChainArgs& a, int vec, bool l1, hipStream_t st) {
  const unsigned grid = (unsigned)a.B * a.nblk;
  if (vec == 2) { if (l1) k_chain_fwd_static<2, true, SEQ><<<grid, kThreads, 0, st>>>(a); else k_chain_fwd_static<2, false, SEQ><<<grid, kThreads, 0, st>>>(a); }
  else          { if (l1) k_chain_fwd_static<1, true, SEQ><<<grid, kThreads, 0, st>>>(a); else k_chain_fwd_static<1, false, SEQ><<<grid, kThreads, 0, st>>>(a); }
}

template <class SEQ>
static void launch_static_bwd_variant(ChainArgs& a, int variant, bool l1, hipStream_t st) {
  switch (variant) {
    case 2: launch_static_bwd<SEQ, true, 1>(a, l1, st); break;       // operator inputs saved in LDS: no faster (106.9 vs 104.2 us)
    default: launch_static_bwd<SEQ, false, 1>(a, l1, st); break;
  }
}

extern "C" {

static int chain_static_variant() {
  static const int v = env_int("T2O_CHAIN_STATIC", 1);                   // 0: the run-time loop kernel for every list (A/B runs)
  return v;
}
static bool chain_has_aot(const ChainArgs& a) { return chain_is<SeqCfg2>(a) || chain_is<SeqCfg5>(a); }
static bool chain_has_static_bwd(const ChainArgs& a, int vec) {
  if (vec != 1 || !chain_static_variant()) return false;
  t2o::JitChain j;
  return chain_has_aot(a) || t2o::jit_lookup(a.ops, a.K, &j);
}

static int fused_chain_launch_fwd(ChainArgs& a, int vec, bool l1, hipStream_t st) {
  if (chain_static_variant()) {
    if (chain_is<SeqCfg2>(a)) { launch_static_fwd<SeqCfg2>(a, vec, l1, st); return 0; }
    if (chain_is<SeqCfg5>(a)) { launch_static_fwd<SeqCfg5>(a, vec, l1, st); return 0; }
    t2o::JitChain j;                       // an operator list specialised at run time (t2o_fused_sequence_prepare)
    if (t2o::jit_lookup(a.ops, a.K, &j)) return t2o::jit_launch(j.fwd[vec == 2 ? 1 : 0][l1 ? 1 : 0], a, (unsigned)a.B * a.nblk, 0, st);
  }
  const unsigned grid = (unsigned)a.B * a.nblk;
  if (vec == 2) { if (l1) k_chain_fwd<2, true><<<grid, kThreads, 0, st>>>(a); else k_chain_fwd<2, false><<<grid, kThreads, 0, st>>>(a); }
  else          { if (l1) k_chain_fwd<1, true><<<grid, kThreads, 0, st>>>(a); else k_chain_fwd<1, false><<<grid, kThreads, 0, st>>>(a); }
  return 0;
}

static int fused_chain_launch_bwd(ChainArgs& a, int vec, bool l1, hipStream_t st) {
  const int use_static = chain_static_variant();
  if (vec == 1 && use_static) {
    if (chain_is<SeqCfg2>(a)) { launch_static_bwd_variant<SeqCfg2>(a, use_static, l1, st); return 0; }
    if (chain_is<SeqCfg5>(a)) { launch_static_bwd_variant<SeqCfg5>(a, use_static, l1, st); return 0; }
    t2o::JitChain j;
    if (t2o::jit_lookup(a.ops, a.K, &j))
      return t2o::jit_launch(j.bwd[l1 ? 1 : 0], a, (unsigned)a.B * a.nblk, sizeof(float) * (size_t)a.bin_off[kMaxChain] * kAccStride, st);
  }
  const unsigned grid = (unsigned)a.B * a.nblk;
  const size_t lds = sizeof(float) * ((size_t)a.bin_off[kMaxChain] * kAccStride +
                                      (vec == 2 ? chain_save_floats<2>(a.K) : chain_save_floats<1>(a.K)));
  if (vec == 2) { if (l1) k_chain_bwd<2, true><<<grid, kThreads, lds, st>>>(a); else k_chain_bwd<2, false><<<grid, kThreads, lds, st>>>(a); }
  else          { if (l1) k_chain_bwd<1, true><<<grid, kThreads, lds, st>>>(a); else k_chain_bwd<1, false><<<grid, kThreads, lds, st>>>(a); }
  return 0;
}


int t2o_fused_sequence_prepare(const int* ops, int K) {
  if (!ops || K < 0) return fail(T2O_EINVAL, "fused_sequence_prepare: null pointer");
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  if (ns < 0) return fail(T2O_EUNSUPPORTED, "operator index not supported, or more than 64 segments");
  for (int s = 0; s < ns; ++s) {
    if (seg[s].sharp || seg[s].n <= 0) continue;
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    chain_fill(a, seg[s], 1, 4, 4, 1, 1);                // (only the operator list matters here)
    if (chain_has_aot(a)) continue;                      // compiled ahead of time
    const int rc = t2o::jit_prepare(a.ops, a.K);
    if (rc != T2O_OK) return rc;
  }
  return T2O_OK;
}

int t2o_fused_sequence_fwd(const int* ops, int K, const float* img, const float* params, const float* target,
                           float* out, float* loss, float* seg_bufs, void* workspace, size_t workspace_bytes, int B,
                           int H, int W, void* stream) {
  if (!ops || K < 0 || !img || !out || (K > 0 && !params)) return fail(T2O_EINVAL, "fused_sequence_fwd: null pointer");
  if (B <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "B, H, W must be positive");
  if (target && !loss) return fail(T2O_EINVAL, "loss is null");
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  if (ns < 0) return fail(T2O_EUNSUPPORTED, "operator index not supported, or more than 64 segments");
  if (ns > 1 && !seg_bufs) return fail(T2O_EINVAL, "seg_bufs is null (see t2o_fused_sequence_buffers)");
  if (target && (!workspace || workspace_bytes < t2o_workspace_bytes(B, H, W))) return fail(T2O_EWORKSPACE, "workspace too small");
  constexpr int forced = 0;
  int vec, iters, nblk;
  chain_geometry(B, H, W, forced, vec, iters, nblk, 0);
  const Geometry g = launch_geometry(B, H, W);
  const size_t img_floats = (size_t)B * 3 * H * W;
  hipStream_t st = (hipStream_t)stream;
  const float* cur = img;
  for (int s = 0; s < ns; ++s) {
    float* dst = (s == ns - 1) ? out : seg_bufs + (size_t)s * img_floats;
    const bool last = (s == ns - 1) && target;
    if (seg[s].sharp) {
      const int k = seg[s].first;
      const int rc = run_fwd(OP_SHARPNESS, nullptr, cur, params + (size_t)k * B * kMaxParam, kMaxParam, nullptr, 0,
                             last ? target : nullptr, dst, last ? loss : nullptr, workspace, workspace_bytes, B, H, W, stream);
      if (rc) return rc;
    } else {
      ChainArgs a;
      memset(&a, 0, sizeof(a));
      chain_fill(a, seg[s], B, H, W, iters, nblk);
      a.img = cur; a.params = params; a.out = dst; a.target = last ? target : nullptr;
      if (last) a.loss_partials = (float*)workspace + ws_partials_floats(g, B, H, W);
      fused_chain_launch_fwd(a, vec, last, st);
      if (last) k_l1_finalize<<<1, kThreads, 0, st>>>(a.loss_partials, B * nblk, a.inv_n, loss);
    }
    cur = dst;
  }
  return check_launch("fused sequence forward");
}

// value_loss != null: the LAST segment's backward also leaves the loss (and the final image in value_out when given): the
// caller ran the forward of every segment but the last (t2o_fused_sequence_l1_value_grad)
static int fused_sequence_bwd_impl(const int* ops, int K, const float* img, const float* params, const float* target,
                                   const float* gloss, const float* gout, float* gimg, float* gparams, const float* seg_bufs,
                                   float* gbuf, void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream,
                                   float* value_out, float* value_loss) {
  if (!ops || K < 0 || !img || (K > 0 && (!params || !gparams)))
    return fail(T2O_EINVAL, "fused_sequence_bwd: null pointer");
  if (target ? !gloss : !gout) return fail(T2O_EINVAL, "fused_sequence_bwd: give (target, gloss) or gout");
  if (B <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "B, H, W must be positive");
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  if (ns < 0) return fail(T2O_EUNSUPPORTED, "operator index not supported, or more than 64 segments");
  if (ns > 1 && (!seg_bufs || !gbuf)) return fail(T2O_EINVAL, "seg_bufs / gbuf is null");
  if (!workspace || workspace_bytes < t2o_workspace_bytes(B, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  constexpr int forced = 0;
  int vec, iters, nblk;
  // backward: one pixel per thread-iteration (measured 164 vs 182 us at bs=64 256x256: the LDS save
  // area halves, so more workgroups are resident); T2O_CHAIN_BWD_VEC=2 restores pixel pairs
  chain_geometry(B, H, W, forced, vec, iters, nblk, 1);
  const size_t img_floats = (size_t)B * 3 * H * W;
  hipStream_t st = (hipStream_t)stream;
  bool has_identity = false;                     // identity rows are written by no kernel
  for (int k = 0; k < K; ++k) has_identity = has_identity || ops[k] == OP_IDENTITY;
  if (has_identity) {   // (a kernel, not hipMemsetAsync: a memset node inside a captured hipGraph was seen to run out of order, t2o_conv.hip)
    const size_t n = (size_t)K * B * kMaxParam;
    k_zero_floats<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(gparams, n);
  }
  const float* gcur = nullptr;
  for (int s = ns - 1; s >= 0; --s) {
    const float* in = s == 0 ? img : seg_bufs + (size_t)(s - 1) * img_floats;
    float* gnext = s == 0 ? gimg : gbuf + (size_t)(s & 1) * img_floats;
    const bool last = (s == ns - 1) && target;     // fused L1: the last segment reads the target instead of a gradient
    if (s == ns - 1 && !target) gcur = gout;
    const bool value = last && value_loss != nullptr;
    if (seg[s].sharp) {
      const int k = seg[s].first;
      const int rc = run_bwd(OP_SHARPNESS, nullptr, in, params + (size_t)k * B * kMaxParam, kMaxParam, nullptr, 0,
                             last ? nullptr : gcur, last ? target : nullptr, last ? gloss : nullptr, gnext,
                             gparams + (size_t)k * B * kMaxParam, kMaxParam, workspace, workspace_bytes, B, H, W, stream,
                             nullptr, nullptr, value ? value_out : nullptr, value ? value_loss : nullptr);
      if (rc) return rc;
    } else {
      ChainArgs a;
      memset(&a, 0, sizeof(a));
      chain_fill(a, seg[s], B, H, W, iters, nblk);
      if (!forced && chain_has_static_bwd(a, vec)) {
        // the compile-time kernel flushes its parameter sums once per THREAD: twice the pixels per thread halves
        // that again (bs=64 256x256: 104.9 us at 4, 100.6 us at 8 pixels per thread); fewer, longer workgroups
        const int it2 = iters * 2 > 16 ? 16 : iters * 2;
        const size_t groups = (size_t)H * W;
        chain_fill(a, seg[s], B, H, W, it2, (int)((groups + (size_t)kThreads * it2 - 1) / ((size_t)kThreads * it2)));
      }
      a.img = in; a.params = params; a.gimg = gnext; a.partials = (float*)workspace;
      if (last) { a.target = target; a.gloss = gloss; } else { a.gout = gcur; }
      if (value) {
        const Geometry gg = launch_geometry(B, H, W);
        a.out = value_out;
        a.loss_partials = (float*)workspace + ws_partials_floats(gg, B, H, W);
      }
      fused_chain_launch_bwd(a, vec, last, st);
      if (a.K > 0) k_chain_finalize<<<B, kThreads, 0, st>>>(a, gparams);
      if (value) k_l1_finalize<<<1, kThreads, 0, st>>>(a.loss_partials, B * a.nblk, a.inv_n, value_loss);
    }
    gcur = gnext;
  }
  return check_launch("fused sequence backward");
}

int t2o_fused_sequence_bwd(const int* ops, int K, const float* img, const float* params, const float* target,
                           const float* gloss, const float* gout, float* gimg, float* gparams, const float* seg_bufs,
                           float* gbuf, void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  return fused_sequence_bwd_impl(ops, K, img, params, target, gloss, gout, gimg, gparams, seg_bufs, gbuf, workspace, workspace_bytes,
                                 B, H, W, stream, nullptr, nullptr);
}

int t2o_fused_sequence_l1_value_grad(const int* ops, int K, const float* img, const float* params, const float* target,
                                     const float* gloss, float* out, float* loss, float* gimg, float* gparams, float* seg_bufs,
                                     float* gbuf, void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (!ops || K <= 0 || !img || !params || !gparams || !target || !gloss || !loss)
    return fail(T2O_EINVAL, "fused_sequence_l1_value_grad: null pointer or empty sequence");
  if (B <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "B, H, W must be positive");
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  if (ns <= 0) return fail(T2O_EUNSUPPORTED, "operator index not supported, more than 64 segments, or nothing to apply");
  if (ns > 1 && (!seg_bufs || !gbuf)) return fail(T2O_EINVAL, "seg_bufs / gbuf is null (see t2o_fused_sequence_buffers)");
  if (!workspace || workspace_bytes < t2o_workspace_bytes(B, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  // the last segment's backward computes its forward anyway: a per-pixel chain recomputes it per pixel, the stencil's
  // strip kernel forms clamp(z + p Lap z) for the sign of (out - target).  Sharpness on the LDS-tile kernels (W % 4 != 0
  // ...) has no such form: forward + backward as two calls
  const Geometry g = launch_geometry(B, H, W);
  bool single = true;
  if (seg[ns - 1].sharp) {
    OpArgs probe;
    memset(&probe, 0, sizeof(probe));
    probe.op = OP_SHARPNESS; probe.B = B; probe.H = H; probe.W = W;
    single = sharp_bwd_uses_strips(probe, g);
  }
  if (!single) {
    if (!out && ns == 1) return fail(T2O_EINVAL, "fused_sequence_l1_value_grad: this image size needs `out` (sharpness on the tile kernels)");
    float* final_img = out ? out : gbuf + (size_t)(ns & 1) * (size_t)B * 3 * H * W;      // (a gradient buffer not yet in use)
    if (int rc = t2o_fused_sequence_fwd(ops, K, img, params, target, final_img, loss, seg_bufs, workspace, workspace_bytes, B, H, W, stream)) return rc;
    return t2o_fused_sequence_bwd(ops, K, img, params, target, gloss, nullptr, gimg, gparams, seg_bufs, gbuf, workspace, workspace_bytes, B, H, W, stream);
  }
  if (ns > 1) {                                  // forward of every segment but the last, materialising the segment boundaries
    int kcut = seg[ns - 1].first;                // operators in front of the last segment
    if (int rc = t2o_fused_sequence_fwd(ops, kcut, img, params, nullptr, seg_bufs + (size_t)(ns - 2) * (size_t)B * 3 * H * W, nullptr, seg_bufs,
                                        workspace, workspace_bytes, B, H, W, stream)) return rc;
  }
  return fused_sequence_bwd_impl(ops, K, img, params, target, gloss, nullptr, gimg, gparams, seg_bufs, gbuf, workspace, workspace_bytes,
                                 B, H, W, stream, out, loss);
}

size_t t2o_candidates_workspace_bytes(int C, int H, int W) {
  if (C <= 0 || H <= 0 || W <= 0) return 0;
  const size_t nblk = ((size_t)H * W + (size_t)kThreads * kCandPix - 1) / ((size_t)kThreads * kCandPix);
  return sizeof(float) * (size_t)C * nblk;
}

int t2o_op_candidates_l1(int op, const float* img, const float* target, const float* params, int C, int param_stride,
                         float* loss, void* workspace, size_t workspace_bytes, int H, int W, void* stream) {
  if (!img || !target || !params || !loss) return fail(T2O_EINVAL, "candidates_l1: null pointer");
  if (C <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "C, H, W must be positive");
  if (op == OP_SHARPNESS || op == OP_IDENTITY || !op_supported(op))
    return fail(T2O_EUNSUPPORTED, "candidate sweep supports the per-pixel operators (0,1,2,3,5,7)");
  if (param_stride < op_num_params(op)) return fail(T2O_EINVAL, "param_stride < number of operator parameters");
  if (!workspace || workspace_bytes < t2o_candidates_workspace_bytes(C, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  CandArgs a;
  memset(&a, 0, sizeof(a));
  a.img = img; a.target = target; a.params = params; a.partials = (float*)workspace;
  a.op = op; a.C = C; a.param_stride = param_stride; a.H = H; a.W = W;
  a.nblk = (int)(((size_t)H * W + (size_t)kThreads * kCandPix - 1) / ((size_t)kThreads * kCandPix));
  const unsigned groups = (unsigned)((C + kCandPerBlock - 1) / kCandPerBlock);
  hipStream_t st = (hipStream_t)stream;
  k_candidates_l1<<<(unsigned)a.nblk * groups, kThreads, 0, st>>>(a);
  k_candidates_finalize<<<(unsigned)C, 64, 0, st>>>(a.partials, a.nblk, 1.0f / (3.0f * (float)H * (float)W), loss);
  return check_launch("candidate sweep");
}

size_t t2o_candidates_multi_workspace_bytes(int J, int C, int H, int W) {
  if (J <= 0) return 0;
  return (size_t)J * t2o_candidates_workspace_bytes(C, H, W);
}

int t2o_op_candidates_multi_l1(const int* ops, const int* img_index, int J, const float* imgs, int n_img, const float* target,
                               const float* params, int C, int param_stride, float* loss, void* workspace,
                               size_t workspace_bytes, int H, int W, void* stream) {
  if (!ops || !img_index || !imgs || !target || !params || !loss) return fail(T2O_EINVAL, "candidates_multi_l1: null pointer");
  if (J <= 0 || J > kMaxCandJobs) return fail(T2O_EINVAL, "candidates_multi_l1: 1 <= J <= 64 jobs per launch");
  if (C <= 0 || H <= 0 || W <= 0 || n_img <= 0) return fail(T2O_EINVAL, "C, H, W, n_img must be positive");
  if (!workspace || workspace_bytes < t2o_candidates_multi_workspace_bytes(J, C, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  MultiCandArgs m;
  memset(&m, 0, sizeof(m));
  for (int j = 0; j < J; ++j) {
    const int op = ops[j];
    if (op == OP_SHARPNESS || op == OP_IDENTITY || !op_supported(op))
      return fail(T2O_EUNSUPPORTED, "candidate sweep supports the per-pixel operators (0,1,2,3,5,7)");
    if (param_stride < op_num_params(op)) return fail(T2O_EINVAL, "param_stride < number of operator parameters");
    if (img_index[j] < 0 || img_index[j] >= n_img) return fail(T2O_EINVAL, "img_index out of range");
    m.op[j] = op;
    m.img_index[j] = img_index[j];
  }
  m.imgs = imgs; m.target = target; m.params = params; m.partials = (float*)workspace;
  m.C = C; m.param_stride = param_stride; m.H = H; m.W = W;
  m.nblk = (int)(((size_t)H * W + (size_t)kThreads * kCandPix - 1) / ((size_t)kThreads * kCandPix));
  const unsigned groups = (unsigned)((C + kCandPerBlock - 1) / kCandPerBlock);
  hipStream_t st = (hipStream_t)stream;
  k_candidates_multi_l1<<<dim3((unsigned)m.nblk * groups, (unsigned)J), kThreads, 0, st>>>(m);
  k_candidates_finalize<<<(unsigned)(J * C), 64, 0, st>>>(m.partials, m.nblk, 1.0f / (3.0f * (float)H * (float)W), loss);
  return check_launch("candidate sweep (multi)");
}

size_t t2o_ssim_workspace_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const size_t tiles = (size_t)((W + kSsimTile - 1) / kSsimTile) * ((H + kSsimTile - 1) / kSsimTile);
  return sizeof(float) * (size_t)B * C * tiles;
}

int t2o_ssim_fwd(const float* img1, const float* img2, float* out, void* workspace, size_t workspace_bytes, int B, int C,
                 int H, int W, void* stream) {
  if (!img1 || !img2 || !out) return fail(T2O_EINVAL, "ssim_fwd: null pointer");
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "B, C, H, W must be positive");
  if (!workspace || workspace_bytes < t2o_ssim_workspace_bytes(B, C, H, W)) return fail(T2O_EWORKSPACE, "workspace too small");
  SsimArgs s;
  memset(&s, 0, sizeof(s));
  s.a = img1; s.b = img2; s.partials = (float*)workspace;
  ssim_window(s.g);
  s.B = B; s.C = C; s.H = H; s.W = W;
  s.tiles_x = (W + kSsimTile - 1) / kSsimTile;
  s.tiles = s.tiles_x * ((H + kSsimTile - 1) / kSsimTile);
  hipStream_t st = (hipStream_t)stream;
  k_ssim_fwd<<<(unsigned)(B * C * s.tiles), kThreads, sizeof(float) * ssim_lds_floats(), st>>>(s);
  k_ssim_finalize<<<B, kThreads, 0, st>>>(s.partials, C * s.tiles, 1.0f / ((float)C * (float)H * (float)W), out);
  return check_launch("ssim forward");
}

int t2o_ssim_bwd(const float* img1, const float* img2, const float* gout, float* g1, float* g2, int B, int C, int H, int W,
                 void* stream) {
  if (!img1 || !img2 || !gout || (!g1 && !g2)) return fail(T2O_EINVAL, "ssim_bwd: null pointer (one of g1 / g2 may be null, not both)");
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(T2O_EINVAL, "B, C, H, W must be positive");
  SsimBwdArgs s;
  memset(&s, 0, sizeof(s));
  s.a = img1; s.b = img2; s.gout = gout; s.ga = g1; s.gb = g2;
  ssim_window(s.g);
  s.B = B; s.C = C; s.H = H; s.W = W;
  s.tiles_x = (W + kSsimTile - 1) / kSsimTile;
  s.tiles = s.tiles_x * ((H + kSsimTile - 1) / kSsimTile);
  s.inv_n = 1.0f / ((float)C * (float)H * (float)W);
  const long long grid = (long long)B * C * s.tiles;
  if (grid >= ((long long)1 << 31)) return fail(T2O_EUNSUPPORTED, "ssim_bwd: too many tiles");
  k_ssim_bwd<<<(unsigned)grid, kThreads, 0, (hipStream_t)stream>>>(s);
  return check_launch("ssim backward");
}

int t2o_choose_op(const float* logp, float* op_mask, const float* u, float explore_prob, long long* pred_op, int* exec_op,
                  int B, int n_cls, void* stream) {
  if (!logp || !op_mask || !pred_op || !exec_op) return fail(T2O_EINVAL, "choose_op: null pointer");
  if (B <= 0 || n_cls <= 0 || n_cls > 32) return fail(T2O_EINVAL, "choose_op: 1..32 operator tokens");
  k_choose_op<<<(unsigned)((B + 63) / 64), 64, 0, (hipStream_t)stream>>>(logp, op_mask, u, explore_prob, pred_op, exec_op, B, n_cls);
  return check_launch("choose_op");
}

int t2o_attn_fwd(const float* q, const float* ctx, float* attn, float* mix, int B, int L, int D, void* stream) {
  if (!q || !ctx || !attn || !mix) return fail(T2O_EINVAL, "attn_fwd: null pointer");
  if (B <= 0 || L <= 0 || L > 64 || D <= 0 || D % 64 != 0 || D > 1024)
    return fail(T2O_EINVAL, "attn: need 1 <= L <= 64, D % 64 == 0, D <= 1024");
  const bool aligned = ((reinterpret_cast<size_t>(q) | reinterpret_cast<size_t>(ctx) | reinterpret_cast<size_t>(mix)) & 15) == 0;
  if (D % 256 == 0 && aligned) k_attn_fwd_wide<<<(unsigned)B, kThreads, 0, (hipStream_t)stream>>>(q, ctx, attn, mix, B, L, D);
  else k_attn_fwd<<<(unsigned)B, kThreads, 0, (hipStream_t)stream>>>(q, ctx, attn, mix, B, L, D);
  return check_launch("attention forward");
}

int t2o_attn_bwd(const float* q, const float* ctx, const float* attn, const float* gmix, const float* gattn, float* gq,
                 float* gctx, int B, int L, int D, void* stream) {
  if (!q || !ctx || !attn || !gmix || !gq || !gctx) return fail(T2O_EINVAL, "attn_bwd: null pointer");
  if (B <= 0 || L <= 0 || L > 64 || D <= 0 || D % 64 != 0 || D > 1024)
    return fail(T2O_EINVAL, "attn: need 1 <= L <= 64, D % 64 == 0, D <= 1024");
  const bool aligned = ((reinterpret_cast<size_t>(q) | reinterpret_cast<size_t>(ctx) | reinterpret_cast<size_t>(gmix) |
                         reinterpret_cast<size_t>(gq) | reinterpret_cast<size_t>(gctx)) & 15) == 0;
  if (D % 256 == 0 && aligned) k_attn_bwd_wide<<<(unsigned)B, kThreads, 0, (hipStream_t)stream>>>(q, ctx, attn, gmix, gattn, gq, gctx, B, L, D);
  else k_attn_bwd<<<(unsigned)B, kThreads, 0, (hipStream_t)stream>>>(q, ctx, attn, gmix, gattn, gq, gctx, B, L, D);
  return check_launch("attention backward");
}

}  // extern "C"
