// t2o_wino_wgrad.hip -- WEIGHT GRADIENT of the stride-1 3x3 convolutions in the Winograd F(2x2, 3x3) domain with both
// transforms ON CHIP (models/actor_resnet.py:24-44 BasicBlock; the 64- and 128-channel stages of the encoder, 64 x 64 and
// 32 x 32 maps at bs = 64, 256 x 256 input; the reference leaves the backward to autograd / the convolution library).
//
//   dw (Co,3,3,Ci) = G^T [ sum over the tiles t of  (A dY A^T)[t] (x) (B^T d B)[t] ] G            (t2o_winograd.hip: k_wino_dy, k_wino_dw)
//
//   The direct kernel (k_conv3x3_wgrad) executes 36 multiplies per (output pixel, channel pair) where this form needs 16; the
//   separate-pass pipeline (V and A dY A^T through HBM: 4 x the activation each) loses at these channel counts for the same
//   reason the forward does (t2o_wino_fused.hip).  Here a workgroup owns a 64 (co) x 64 (ci) tile of ALL 16 transformed planes:
//   each of its 4 waves a 32 x 32 quadrant, 16 accumulator blocks of v_mfma_f32_32x32x2_f32 = 256 registers (one wave per
//   SIMD), and walks a range of the tile index -- the contraction index of this product -- in steps of 8 tiles (one tile row
//   segment: 16 x 2 output pixels):
//     * every thread loads ONE (tile, channel pair) 4 x 4 input patch (16 x 8 bytes) and ONE 2 x 2 output-gradient block
//       (4 x 8 bytes) straight from global memory (scalar row bases + loop-invariant lane offsets: no vector address arithmetic
//       in the loop; rows outside the image come from a zero block, columns outside it are loaded clamped and multiplied by 0),
//       one step ahead of their use;
//     * forms B^T d B and A dY A^T in registers (v_pk_add_f32) and writes them to LDS as V[xi][tile][64 ci], Ad[xi][tile][64 co]
//       (2 x 32 KiB, double-buffered: 128 KiB);
//     * the MFMA operands are plain ds_read_b32 of those rows (a lane = one channel of one of the k-step's two tiles -- the
//       operand order of the instruction, as in k_gemm_tn), 4 MFMAs per plane and step, 64 per wave and step.
//   The vector instructions of a step's transforms sit in ONE gap between the wave's own MFMAs, its loads and stores one or two
//   per gap behind them (k_wino_wgrad's step comment has the measurements): a step takes ~4,990 cycles for the 4,096 of its MFMAs.  The workgroup's 16 x 64 x 64 partial sums go to `part[split]`; a fixed-order two-level
//   sum over the splits (k_wgw_reduce) and the existing G^T dU G kernel finish: deterministic, no atomics.
//
//   Algorithmic work per layer and encoder pass at bs = 64: 8.6 GFLOP (the direct kernel: 19.3).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kWgThreads = 256;
constexpr int kStepTiles = 8;                      // tiles (= contraction rows) per step
constexpr int kPlaneBytes = kStepTiles * 64 * 4;   // one plane of a step: 8 tiles x 64 channels
constexpr int kOpBuf = 16 * kPlaneBytes;           // 32 KiB per operand and buffer

struct WgwArgs {
  const float* x;        // (n_img, H, W, Ci)
  const float* dy;       // (n_img, H, W, Co)
  float* part;           // (splits, 16, Co, Ci)
  const float* zero;     // >= 17 * Ci * 4 + 256 bytes of zeros
  int n_img, H, W, Ci, Co;
  int steps_total;       // n_img * (H / 2) * (W / 16)
  int splits, combos, tiles_ci;
  unsigned long long* stamps;   // diagnostic builds only (tools/diag/wgw_clock.hip)
};

__device__ __forceinline__ void mfma_asm(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int kFirst, int kLast, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (kFirst < kLast) {
    f(std::integral_constant<int, kFirst>{});
    static_for<kFirst + 1, kLast>(f);
  }
}
typedef __attribute__((address_space(1))) const char* gptr;     // global memory, explicitly: scalar base + 32-bit lane offset loads
__device__ __forceinline__ v2f ld2(gptr base, unsigned off) { return *reinterpret_cast<__attribute__((address_space(1))) const v2f*>(base + off); }

__global__ __launch_bounds__(kWgThreads, 1) void k_wino_wgrad(WgwArgs a) {
  __shared__ __attribute__((aligned(16))) char Vs[2][kOpBuf];
  __shared__ __attribute__((aligned(16))) char As[2][kOpBuf];

  // workgroup -> (split of the step range, (co, ci) tile): the tiles of one split are neighbours inside an XCD (they read the
  // same x / dy rows)
  const int b = blockIdx.x;
  const int xcd = b % 8, k8 = b / 8;
  const int split = (k8 / a.combos) * 8 + xcd, combo = k8 % a.combos;
  if (split >= a.splits) return;
  const int co0 = (combo / a.tiles_ci) * 64, ci0 = (combo % a.tiles_ci) * 64;
  const int s_begin = (int)(((long long)a.steps_total * split) / a.splits);
  const int s_end = (int)(((long long)a.steps_total * (split + 1)) / a.splits);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;               // this wave's quadrant: co half, ci half
  const int ln = lane & 31, lh = lane >> 5;
  const int TH = a.H >> 1, SEG = a.W >> 4;

  // ---- transform role: tile tx of the step, channel pair cp (channels 2 cp, 2 cp + 1 of the workgroup's 64)
  const int tx = 2 * wave + (lane >> 5), cp = lane & 31;
  const unsigned xpix = (unsigned)a.Ci * 4u, ypix = (unsigned)a.Co * 4u;
  // byte offsets from the row base (which stands at column 16 sx - 1): column j of the patch is pixel 2 tx + j
  const unsigned xo1 = (unsigned)(2 * tx + 1) * xpix + (unsigned)cp * 8u;
  const unsigned xo0 = xo1 - xpix, xo3 = xo1 + 2u * xpix;
  const unsigned yo = (unsigned)(2 * tx) * ypix + (unsigned)cp * 8u;
  const bool is_first = tx == 0, is_last = tx == kStepTiles - 1;
  const unsigned wr_off = (unsigned)(tx * 256 + cp * 8);        // LDS: [xi][tile][64 channels]

  // ---- step walker (scalar): s -> (img, ty, sx)
  struct Pos { int img, ty, sx; };
  auto pos_of = [&](int s) {
    Pos p;
    p.sx = s % SEG;
    const int q = s / SEG;
    p.ty = q % TH;
    p.img = q / TH;
    return p;
  };
  auto advance = [&](Pos& p) {
    p.sx += 1;
    if (p.sx == SEG) { p.sx = 0; p.ty += 1; if (p.ty == TH) { p.ty = 0; p.img += 1; } }
  };

  v2f d[4][4], dyv[2][2];                                   // the loaded patch / block (one step ahead of its transform)
  float mL = 1.0f, mR = 1.0f;                               // ... and its edge-column masks
  const gptr xbase = (gptr)a.x + (size_t)ci0 * 4;
  const gptr ybase = (gptr)a.dy + (size_t)co0 * 4;
  const gptr zbase = (gptr)a.zero;
  // loop-invariant lane offsets: every load is scalar base + ONE 32-bit register.  They pass through an empty asm statement per
  // step: left to itself the compiler widens the invariant ones to 64 bits and adds the base with vector instructions (12 per step)
  unsigned xo1v = xo1, xo2v = xo1 + xpix, yo0v = yo, yo1v = yo + ypix;
  const size_t xrow = (size_t)a.W * xpix, yrow = (size_t)a.W * ypix;

  // the 20 loads of a step, in two parts: the step's addresses (scalar row bases, the two edge-column offsets and masks) ...
  struct LoadCtx { gptr xr[4]; gptr yr[2]; unsigned o0, o3; float mL, mR; };
  auto load_ctx = [&](const Pos& p) {
    LoadCtx c;
    const bool edgeL = p.sx == 0, edgeR = p.sx == SEG - 1;
    asm volatile("" : "+v"(xo1v), "+v"(xo2v), "+v"(yo0v), "+v"(yo1v));
    c.o0 = (edgeL && is_first) ? xo1v : xo0;                             // outside columns: a valid address, value masked to 0
    c.o3 = (edgeR && is_last) ? xo2v : xo3;
    c.mL = (edgeL && is_first) ? 0.0f : 1.0f;
    c.mR = (edgeR && is_last) ? 0.0f : 1.0f;
    // row 2 ty of the image at column 16 sx; the x bases stand one pixel LEFT of it (the lane offsets are all >= 0).  Pixel
    // indices fit 32 bits (wgw_supported).
    const unsigned pix = (unsigned)((p.img * a.H + 2 * p.ty) * a.W + 16 * p.sx);
    c.xr[1] = xbase + (size_t)pix * xpix - xpix;                         // patch row 1 = image row 2 ty
    c.xr[0] = p.ty == 0 ? zbase : c.xr[1] - xrow;
    c.xr[2] = c.xr[1] + xrow;
    c.xr[3] = p.ty == TH - 1 ? zbase : c.xr[2] + xrow;
    c.yr[0] = ybase + (size_t)pix * ypix;
    c.yr[1] = c.yr[0] + yrow;
    return c;
  };
  // ... and the loads one by one (k = 4 i + j: patch element (i, j); 16 .. 19: the 2 x 2 output-gradient block), so that the loop
  // can deal them out one per MFMA gap: issued in one batch, the 4 waves' 80 loads queued up in front of the CU's one
  // texture-address unit and the issuing plane took 1,370 cycles instead of 256 (tools/diag/wgw_clock.hip)
  auto load_one = [&](auto kc, const LoadCtx& c) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < 16) {
      constexpr int i = k >> 2, j = k & 3;
      d[i][j] = ld2(c.xr[i], j == 0 ? c.o0 : j == 1 ? xo1v : j == 2 ? xo2v : c.o3);
    } else {
      constexpr int r = (k - 16) >> 1, q = (k - 16) & 1;
      dyv[r][q] = ld2(c.yr[r], q == 0 ? yo0v : yo1v);
    }
  };

  // ---- transforms, statement by statement (dealt out between MFMAs in the loop)
  v2f tr[4][4], tv[4][4], ar[2][2], av[4][4];
  auto t_row = [&](auto kc) {                             // k = 4 j + r: B^T d, rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    constexpr int k = decltype(kc)::value, j = k >> 2, r = k & 3;
    constexpr int p = r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : 1, q = r == 0 ? 2 : r == 1 ? 2 : r == 2 ? 1 : 3;
    if constexpr (r == 1) tr[r][j] = d[p][j] + d[q][j];
    else tr[r][j] = d[p][j] - d[q][j];
  };
  // the same along the columns, k = 4 i + c.  The patch's outside columns (0 at the image's left edge, 3 at its right edge) were
  // loaded from a valid neighbour and count as zeros: their 0 / 1 mask rides in the column step as a fused multiply-add -- the
  // product with 0 or 1 is exact, so the result is that of masking first (8 vector instructions per step less)
  auto t_col = [&](auto kc, float cL, float cR) {
    constexpr int k = decltype(kc)::value, i = k >> 2, c = k & 3;
    if constexpr (c == 0) tv[i][0] = __builtin_elementwise_fma(tr[i][0], (v2f){cL, cL}, -tr[i][2]);
    else if constexpr (c == 1) tv[i][1] = tr[i][1] + tr[i][2];
    else if constexpr (c == 2) tv[i][2] = tr[i][2] - tr[i][1];
    else tv[i][3] = __builtin_elementwise_fma(-tr[i][3], (v2f){cR, cR}, tr[i][1]);
  };
  // A dY A^T of the 2 x 2 block WITHOUT its sign flips: rows (d0, d0 + d1, d0 - d1, d1), then per row (r0, r0 + r1, r0 - r1, r1).
  // k_wino_dy's planes 3, 7, 11 (-r1 of rows 0..2) and 12, 13, 14 (row 3 = -d1; plane 15 = -(-d1's r1) is positive) are the
  // negatives of these; a product with a negated operand is the negated product exactly, so k_wgw_reduce flips the sign of those
  // planes' sums instead (6 vector instructions per step less).
  auto a_row = [&](auto kc) {                             // k = 0, 1: column c of the block
    constexpr int c = decltype(kc)::value;
    ar[0][c] = dyv[0][c] + dyv[1][c];
    ar[1][c] = dyv[0][c] - dyv[1][c];
  };
  auto a_col = [&](auto kc) {                             // k = row i of A dY: (d0, d0 + d1, d0 - d1, d1)
    constexpr int i = decltype(kc)::value;
    const v2f r0 = i == 0 ? dyv[0][0] : i == 1 ? ar[0][0] : i == 2 ? ar[1][0] : dyv[1][0];
    const v2f r1 = i == 0 ? dyv[0][1] : i == 1 ? ar[0][1] : i == 2 ? ar[1][1] : dyv[1][1];
    av[i][0] = r0;
    av[i][1] = r0 + r1;
    av[i][2] = r0 - r1;
    av[i][3] = r1;
  };
  auto v_store = [&](auto kc, int buf) {
    constexpr int k = decltype(kc)::value;
    *reinterpret_cast<v2f*>(&Vs[buf][0] + k * kPlaneBytes + wr_off) = tv[k >> 2][k & 3];
  };
  auto a_store = [&](auto kc, int buf) {
    constexpr int k = decltype(kc)::value;
    *reinterpret_cast<v2f*>(&As[buf][0] + k * kPlaneBytes + wr_off) = av[k >> 2][k & 3];
  };
  // the transform of a step as ONE list of 38 statements: 0-15 the patch's row steps | 16-17 the block's row steps | 18-21 its
  // column steps (d and dyv are dead behind statement 21) | 22-37 the patch's column steps
#ifndef T2O_WGW_VPG
#define T2O_WGW_VPG 38
#endif
  constexpr int kVpg = T2O_WGW_VPG, kWork = 38;
  constexpr int kLoadGap = (22 + kVpg - 1) / kVpg, kStoreGap = (kWork + kVpg - 1) / kVpg;
  static_assert(kLoadGap + 20 <= 60 && kStoreGap + 16 <= 60, "the step's loads and stores end in front of its barrier");
  auto work = [&](auto wc) {
    constexpr int w = decltype(wc)::value;
    if constexpr (w < 16) t_row(std::integral_constant<int, w>{});
    else if constexpr (w < 18) a_row(std::integral_constant<int, w - 16>{});
    else if constexpr (w < 22) a_col(std::integral_constant<int, w - 18>{});
    else t_col(std::integral_constant<int, w - 22>{}, mL, mR);
  };

  // ---- MFMA operands: k-step e of plane xi covers tiles 2 e, 2 e + 1; lane (ln, lh) = channel ln of tile 2 e + lh
  const unsigned fa_off = (unsigned)(lh * 256 + (32 * wm + ln) * 4);
  const unsigned fb_off = (unsigned)(lh * 256 + (32 * wn + ln) * 4);
  float fa[2][4], fb[2][4];
  auto frag_read = [&](auto pc, auto slotc, int buf) {
    constexpr int p = decltype(pc)::value, slot = decltype(slotc)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      fa[slot][e] = *reinterpret_cast<const float*>(&As[buf][0] + p * kPlaneBytes + e * 512 + fa_off);
      fb[slot][e] = *reinterpret_cast<const float*>(&Vs[buf][0] + p * kPlaneBytes + e * 512 + fb_off);
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

  // ---- prologue: step s_begin transformed into buffer 0; the loads of step s_begin + 1 in flight.  Positions past the range's
  // end are clamped to its last step (loaded and transformed again, into a buffer nobody reads).
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  Pos pl = pos_of(s_begin);                                // position of the newest loads
  int sl = s_begin;                                        // ... and its step number
  auto next_ctx = [&]() {                                  // advance (clamped) and form the addresses
#ifndef T2O_WGW_SAMEROWS
    if (sl + 1 < s_end) { advance(pl); sl += 1; }
#endif
    return load_ctx(pl);
  };
  {
    const LoadCtx c0 = load_ctx(pl);
    static_for<0, 20>([&](auto kc) { load_one(kc, c0); });
    mL = c0.mL; mR = c0.mR;
  }
  static_for<0, kWork>([&](auto wc) { work(wc); });
  static_for<0, 16>([&](auto kc) { v_store(kc, 0); a_store(kc, 0); });
  {
    const LoadCtx c1 = next_ctx();
    static_for<0, 20>([&](auto kc) { load_one(kc, c1); });
    mL = c1.mL; mR = c1.mR;
  }
  __syncthreads();
  frag_read(I0{}, I0{}, 0);

  // ---- one step (bufc = step parity, a compile-time constant: the LDS addresses are immediates): the MFMAs of step s from LDS
  // buffer buf; the transform of step s + 1 (loaded during step s - 1) into buffer buf ^ 1; then the loads of step s + 2.
  // Gaps (one behind each of the 64 MFMAs), measured with tools/diag/wgw_clock.hip (cycles per step; the 64 MFMAs alone: 4,096):
  //   * the 38 transform statements ALL in gap 0: a gap that holds vector instructions costs ~20 cycles whatever their number
  //     plus 7-10 per instruction (two per gap over 19 gaps: 5,597; four: 5,366; eight: 5,238; nineteen: 5,163; all: 5,122);
  //   * the 20 loads ONE per gap from gap 1 on (in one batch the 4 waves' 80 loads queued up in front of the CU's one
  //     texture-address unit: 1,370 cycles for that plane instead of 256);
  //   * the 32 stores two per gap behind them (inside gap 0, right behind their producers: 5,544 against 4,985);
  //   * a second register set for the loads (consumed a whole step later) changed nothing (4,984): what the loads cost -- 230 of
  //     the step's ~890 cycles beside its MFMAs, by the same loop reading one row over and over -- is queueing in the memory
  //     system, not the latency of a single load.
#ifdef T2O_WGW_DIAG
  unsigned ph[16] = {};
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  auto step_body = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#ifdef T2O_WGW_DIAG
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    const LoadCtx cn = next_ctx();
    static_for<0, 16>([&](auto pc) {
      constexpr int p = decltype(pc)::value;               // plane
      if constexpr (p == 15) {
        // the step's barrier sits in front of the last plane: behind it the next step's first fragments are requested and
        // plane 15's MFMAs (operands in registers) cover their latency
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        frag_read(I0{}, I0{}, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<0, 4>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        mfma_asm(acc[p], fa[p & 1][e], fb[p & 1][e]);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int g = 4 * p + e;                       // gap number 0 .. 63
        // fragments of the next plane: behind this plane's first MFMA, into the other slot
        if constexpr (e == 0 && p + 1 < 15) frag_read(std::integral_constant<int, p + 1>{}, std::integral_constant<int, (p + 1) & 1>{}, buf);
        if constexpr (e == 0 && p == 14) frag_read(std::integral_constant<int, 15>{}, I1{}, buf);
#ifndef T2O_WGW_NO_XFORM
        static_for<kVpg * g, (kVpg * (g + 1) < kWork ? kVpg * (g + 1) : kWork)>([&](auto wc) { work(wc); });
        if constexpr (g >= kLoadGap && g < kLoadGap + 20) { load_one(std::integral_constant<int, g - kLoadGap>{}, cn); }
        if constexpr (g >= kStoreGap && g < kStoreGap + 16) {
          v_store(std::integral_constant<int, g - kStoreGap>{}, buf ^ 1);
          a_store(std::integral_constant<int, g - kStoreGap>{}, buf ^ 1);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
#ifdef T2O_WGW_DIAG
      { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[p] += (unsigned)(tn - tprev); tprev = tn; }
#endif
    });
    mL = cn.mL; mR = cn.mR;
  };
  // (pairs, then the odd step: a conditional second step INSIDE the loop made the compiler shuffle the 256 accumulators through
  // vector registers at the merge -- 284 v_accvgpr moves per step)
  int s = s_begin;
  for (; s + 1 < s_end; s += 2) {
    step_body(I0{});
    step_body(I1{});
  }
  if (s < s_end) step_body(I0{});
#ifdef T2O_WGW_DIAG
  if (a.stamps && lane == 0) {          // per wave: [loop cycles, steps, planes 0..15]
    unsigned long long* q = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 32;
    q[0] = __builtin_amdgcn_s_memtime() - t_loop; q[1] = (unsigned long long)(s_end - s_begin);
    for (int i = 0; i < 16; ++i) q[2 + i] = ph[i];
  }
#endif
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // (the last MFMA's 16 passes, before any accumulator is read)

  // ---- partial sums out: acc[xi][r] = dU[xi][co0 + 32 wm + (r & 3) + 8 (r >> 2) + 4 lh][ci0 + 32 wn + ln]
  float* const out = a.part + ((size_t)split * 16) * a.Co * a.Ci + (size_t)(co0 + 32 * wm + 4 * lh) * a.Ci + ci0 + 32 * wn + ln;
  const size_t plane = (size_t)a.Co * a.Ci;
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)p * plane + (size_t)((r & 3) + 8 * (r >> 2)) * a.Ci] = acc[p][r];
}

// part (S, E) -> sum (E), E = 16 * Co * Ci floats: a thread sums a group of S / kGroups consecutive splits of one float4 in
// order, the groups meet in LDS and are added in order: fixed order whatever the launch geometry.
constexpr int kRedGroups = 8, kRedQuads = 32;
// Planes 3, 7, 11, 12, 13, 14 were accumulated with A dY A^T's sign flips left out (k_wino_wgrad: a_col): their sums change sign here.
__global__ __launch_bounds__(256) void k_wgw_reduce(const float* __restrict__ part, float* __restrict__ sum, int S, size_t E4, size_t plane4) {
  __shared__ float4 sm[kRedGroups][kRedQuads];
  const int q = threadIdx.x % kRedQuads, g = threadIdx.x / kRedQuads;
  const size_t idx = (size_t)blockIdx.x * kRedQuads + q;
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (idx < E4) {
    const int per = (S + kRedGroups - 1) / kRedGroups;
    const int s0 = g * per, s1 = s0 + per < S ? s0 + per : S;
    for (int s = s0; s < s1; ++s) {
      const float4 t = reinterpret_cast<const float4*>(part)[(size_t)s * E4 + idx];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  }
  sm[g][q] = v;
  __syncthreads();
  if (g == 0 && idx < E4) {
    float4 r = sm[0][q];
#pragma unroll
    for (int k = 1; k < kRedGroups; ++k) { const float4 t = sm[k][q]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    const int xi = (int)(idx / plane4);
    if ((0x7888 >> xi) & 1) { r.x = -r.x; r.y = -r.y; r.z = -r.z; r.w = -r.w; }       // bits 3, 7, 11, 12, 13, 14
    reinterpret_cast<float4*>(sum)[idx] = r;
  }
}

bool wgw_supported(int n_img, int H, int W, int Ci, int Co) {
  return n_img > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0 && Ci >= 64 && Ci % 64 == 0 && Ci <= 512 && Co >= 64 && Co % 64 == 0 &&
         Co <= 512 && (long long)n_img * H * W < ((long long)1 << 31);            // (32-bit pixel indices in the kernel)
}

int wgw_splits(int n_img, int H, int W, int Ci, int Co) {
  // enough workgroups for every CU (256), in whole rounds of 8 splits (the XCD mapping), every split at least two steps
  const int combos = (Ci / 64) * (Co / 64);
  const long long steps = (long long)n_img * (H / 2) * (W / 16);
  int s = (256 + combos - 1) / combos;
  s = ((s + 7) / 8) * 8;
  while (s > 8 && (long long)s * 2 > steps) s -= 8;
  if ((long long)s * 2 > steps) s = steps >= 2 ? (int)(steps / 2) : 1;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" {

int t2o_wino_fused_wgrad_supported(int n_img, int H, int W, int Ci, int Co) { return wgw_supported(n_img, H, W, Ci, Co) ? 1 : 0; }

size_t t2o_wino_fused_wgrad_workspace_bytes(int n_img, int H, int W, int Ci, int Co) {
  if (!wgw_supported(n_img, H, W, Ci, Co)) return 0;
  const size_t E = (size_t)16 * Co * Ci;
  return ((size_t)wgw_splits(n_img, H, W, Ci, Co) + 1) * E * sizeof(float);
}


int t2o_wino_fused_wgrad_nhwc(const float* x, const float* dy, float* dw, const float* zeros, void* workspace, size_t workspace_bytes,
                              int n_img, int H, int W, int Ci, int Co, int accumulate, void* stream) {
  if (!x || !dy || !dw || !zeros || !workspace) return set_error(T2O_EINVAL, "wino_fused_wgrad: null pointer");
  if (((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(zeros) | reinterpret_cast<size_t>(workspace)) & 15) != 0)
    return set_error(T2O_EINVAL, "wino_fused_wgrad: x, dy, zeros and the workspace must be 16-byte aligned");
  if (!wgw_supported(n_img, H, W, Ci, Co)) return set_error(T2O_EUNSUPPORTED, "wino_fused_wgrad: H, W multiples of 16, Ci and Co multiples of 64 (<= 512), fewer than 2^31 pixels");
  if (workspace_bytes < t2o_wino_fused_wgrad_workspace_bytes(n_img, H, W, Ci, Co)) return set_error(T2O_EWORKSPACE, "wino_fused_wgrad: workspace too small");
  WgwArgs a = {};
  a.x = x; a.dy = dy; a.zero = zeros;
  a.part = (float*)workspace;
  a.n_img = n_img; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.steps_total = n_img * (H / 2) * (W / 16);
  a.splits = wgw_splits(n_img, H, W, Ci, Co);
  a.tiles_ci = Ci / 64;
  a.combos = (Ci / 64) * (Co / 64);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(((a.splits + 7) / 8) * 8 * a.combos);
  k_wino_wgrad<<<grid, kWgThreads, 0, st>>>(a);
  if (hipGetLastError() != hipSuccess) return set_error(T2O_ELAUNCH, "wino_fused_wgrad launch failed");
  const size_t E = (size_t)16 * Co * Ci, E4 = E / 4;
  float* sum = a.part + (size_t)a.splits * E;
  k_wgw_reduce<<<(unsigned)((E4 + kRedQuads - 1) / kRedQuads), 256, 0, st>>>(a.part, sum, a.splits, E4, (size_t)Co * Ci / 4);
  if (hipGetLastError() != hipSuccess) return set_error(T2O_ELAUNCH, "wino_fused_wgrad reduce launch failed");
  return t2o_wino_dw_transform(sum, dw, Co, Ci, 1, accumulate, stream);
}

}  // extern "C"
