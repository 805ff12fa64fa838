// t2o_conv.hip -- fp32 convolutions of the actor's image encoder (models/actor_resnet.py:27-44: the 3x3,
// stride-1, padding-1 convolutions of the BasicBlocks, 12 of the encoder's 21 convolutions and ~80 % of its
// FLOPs) as implicit GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per product,
// 64 FLOP/clk/SIMD = the chip's 157 TFLOP/s fp32 peak).  This is the one dense contraction of the hot path -- the
// only place MFMA belongs (everything else is per-pixel, HBM- or VALU-bound work).
//
// Layout: activations NHWC ((N,H,W,C) rows of C contiguous channels = torch.channels_last), weights
// (Co, 3, 3, Ci) (= channels_last storage of a (Co,Ci,3,3) weight).
//
// WEIGHT GRADIENT   dw[co][tap][ci] = sum_p dy[p][co] * x[p + shift(tap)][ci]      (p over all N*H*W pixels)
//   One GEMM per tap: M = Co, N = Ci, K = pixels.  Both operands are stored pixel-major with the channel
//   contiguous, which IS the MFMA operand order (lane l supplies A[m = l % 32][k = l / 32]: for one pixel k the 32
//   lanes read 32 consecutive channels), so tiles go global -> LDS as plain 16-byte row copies (LDS-DMA) and
//   fragments are conflict-free ds_read_b32 -- no transposes anywhere.
//   Workgroup = 256 threads = 2 x 2 waves = one kernel ROW (3 taps) of a TM x TN channel tile (128 x 64: 2 x 1 MFMA
//   blocks per wave and tap, 96 accumulator registers), K consumed in stages of 32 pixels through two LDS buffers.
//   K is split across workgroups (a stage-aligned pixel range each); every workgroup writes its tiles into its own
//   partial (split, Co, 9, Ci) array and a second kernel adds the partials in a fixed order: deterministic, no
//   float atomics.  The 3 kernel rows of one (pixel range, tile) get block indices congruent mod 8, i.e. land on the
//   same XCD and share its L2.
//   What the loop looks like, and why, is written at the kernel (measurements: tools/diag/*.hip, profiles/r02_*).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kConvThreads = 256;

struct WgradArgs {
  const float* x;      // (N,H,W,Ci)
  const float* dy;     // (N,H,W,Co)
  float* partial;      // (splits, Co, 9, Ci)
  int N, H, W, Ci, Co;
  int tiles_m, tiles_n;       // channel tiles
  int splits, stages_per_split, total_stages;
  const float* zero;            // >= 16 bytes of zeros: the LDS-DMA source of padding pixels and of the tail
  unsigned long long* stamps;   // diagnostic builds only (tools/diag/wgrad_clock.hip): per-workgroup clock stamps; null in the product
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// the zero region of a workspace (a multiple of 256 bytes).  A kernel, not hipMemsetAsync: inside a captured hipGraph
// the memset node was seen to run out of order with the kernels around it (replays read a stale region)
__global__ __launch_bounds__(256) void k_conv_zero(float4* p, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Instructions beside an fp32 MFMA stream (tools/diag/valu_beside_mfma.hip, two waves per SIMD, the other one
// streaming v_mfma_f32_32x32x2_f32): a VECTOR-ALU instruction of this wave takes 58 cycles when the partner's MFMAs
// are interleaved with LDS reads and 300-400 beside a bare stream (11 alone); a SCALAR instruction 15-18 (10 alone);
// LDS reads and LDS-DMA issue are not affected.  The loop below is therefore written with no vector-ALU
// instructions at all outside the MFMAs: addresses are scalar bases plus loop-invariant lane offsets (LDS-DMA in
// its saddr form, ds_read with immediate offsets), padding is decided per 1 KiB piece in scalar registers, and the
// two edge masks of an image row are applied under a scalar branch only in the k-pairs that hold a row end.

// One LDS-DMA wave-instruction (global_load_lds_dwordx4, saddr form): lane l's 16 bytes at sbase + voff[l] go to
// LDS byte address lds_dst + 16 * l (lds_dst wave-uniform, in M0).  Inline assembly: through the builtin the
// compiler cannot tell the DMA's LDS destination from the buffer the following ds_reads use and waits vmcnt(0) in
// front of them; completion is waited for explicitly (glds_wait) before the barrier that publishes the tile.
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
#if defined(T2O_DIAG_GLDS) && T2O_DIAG_GLDS == 0       // diagnostic: no DMA at all (results are wrong, timing only)
  return;
#elif defined(T2O_DIAG_GLDS) && T2O_DIAG_GLDS == 1     // diagnostic: M0 not restored
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  return;
#elif defined(T2O_DIAG_GLDS) && T2O_DIAG_GLDS == 2     // diagnostic: a plain load instead of the DMA (data dropped)
  float4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return;
#endif
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// ds_read_b32 with a compile-time byte offset: one loop-invariant address register per operand, no address arithmetic
template <int kOff>
__device__ __forceinline__ float lds_read(unsigned base) {
  static_assert(kOff >= 0 && kOff < 65536, "ds_read immediate offset");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(kOff));
  return v;
}
template <int kFirst, int kLast, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (kFirst < kLast) {
    f(std::integral_constant<int, kFirst>{});
    static_for<kFirst + 1, kLast>(f);
  }
}

constexpr int kStagePix = 32;        // pixels (K of the GEMM) per LDS stage
constexpr int kHalo = 4;             // x tile rows before the stage's first pixel (1 needed; 4 keeps 1 KiB pieces inside an image row)

// One workgroup = one kernel ROW (kh fixed, kw = 0, 1, 2) of a TM x TN channel tile over a range of pixels.  The three
// taps of a row read the same dy tile and the same x rows shifted by one pixel, so the x tile is staged once with a
// halo (row r of Bs = pixel p0 - 4 + r of the image row h + dh) and every dy fragment feeds three MFMAs.
// Requires W % 4 == 0: a 1 KiB DMA piece (2 or 4 pixels, aligned) then never crosses an image row, so a piece is
// either a run of real pixels or all padding (row h + dh outside the image, or beyond the last pixel) and reads the
// zero region instead -- a scalar select of the base address.  A pixel whose horizontal neighbour lies outside its
// image row (w == 0 for kw = 0, w == W-1 for kw = 2) has that fragment element zeroed in registers.
// kS = 2: the weight gradient of a STRIDE-2 convolution.  a.H, a.W are then the dy grid (x is 2H x 2W) and the tap
// (kh, kw) of dy pixel (alpha, beta) reads x at (2 alpha + kh - 1, 2 beta + kw - 1): the x tile is two planes, the
// even columns 2 beta (kw = 1) and the odd columns 2 beta + 1 (kw = 2, and kw = 0 one pixel to the left), each
// gathered by the LDS-DMA with a lane stride of two pixels.
template <int TM, int TN, int kMinWaves, int kS = 1>
__global__ __launch_bounds__(kConvThreads, kMinWaves) void k_conv3x3_wgrad(WgradArgs a) {
  constexpr int BM = TM / 64, BN = TN / 64;          // MFMA blocks per wave along m / n (wave tile = TM/2 x TN/2)
  constexpr int RWA = 256 / TM, RWB = 256 / TN;            // tile rows per 1 KiB DMA piece
  constexpr int NPA = kStagePix / RWA;                     // dy pieces per stage
  constexpr int PB0 = (kHalo - 1) / RWB;                   // first x piece holding a row that is read (rows kHalo-1 .. kHalo+kStagePix)
  constexpr int NA = NPA / 4, NB = ((kHalo + kStagePix) / RWB + 1 - PB0 + 3) / 4;      // pieces per wave
  constexpr int BROWS = (PB0 + 4 * NB) * RWB;              // x tile rows: every wave loads NB pieces (the last ones past the rows that are read)
  __shared__ __attribute__((aligned(16))) float As[2][kStagePix][TM];
  __shared__ __attribute__((aligned(16))) float Bs[2][kS][BROWS][TN];

  // block index -> (split, channel tile, kernel row).  The 3 kernel rows of one (split, tile) re-read the same dy
  // tile and neighbouring x rows: they get block indices congruent mod 8 (same XCD, shared L2) and adjacent in
  // dispatch order; consecutive (split, tile) units go to consecutive XCDs, so every XCD gets the same number of
  // workgroups.
  const int units = a.splits * a.tiles_m * a.tiles_n;
  const int b = blockIdx.x;
  const int unit = (b / 24) * 8 + b % 8;
  const int kh = (b % 24) / 8;
  if (unit >= units) return;
  const int tiles = a.tiles_m * a.tiles_n;
  // (integer division by a run-time value is done in vector registers even for wave-uniform operands: the results
  // go back to scalar registers here, once, so that everything derived from them stays scalar)
  const int split = __builtin_amdgcn_readfirstlane(unit / tiles), tile = unit - split * tiles;
  const int tile_m = __builtin_amdgcn_readfirstlane(tile / a.tiles_n), tile_n = tile - tile_m * a.tiles_n;
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  const int dh = kh - 1;

  unsigned long long t0 = 0, r0 = 0;
  if (a.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  const int P = a.N * a.H * a.W;
  const int s0 = split * a.stages_per_split;
  int s1 = s0 + a.stages_per_split;
  if (s1 > a.total_stages) s1 = a.total_stages;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lk = lane >> 5;

  f32x16 acc[3][BM][BN];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
      for (int jn = 0; jn < BN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][jn][r] = 0.0f;

  // ---- global -> LDS by LDS-DMA.  One wave-instruction writes 1 KiB of consecutive LDS = RW whole tile rows; lane l
  // supplies the 16 bytes at column 4 * (l % (T/4)) of row l / (T/4): a loop-invariant byte offset from the piece's
  // scalar base address.
  const unsigned lane_a = (unsigned)((lane / (TM / 4)) * a.Co + (lane % (TM / 4)) * 4) * 4u;
  const unsigned lane_b = (unsigned)((lane / (TN / 4)) * kS * a.Ci + (lane % (TN / 4)) * 4) * 4u;
  // (a.zero is read from the kernel arguments where it is used: as a local captured by the lambdas below, 'ok ? p :
  // zero' becomes a load through a selected ADDRESS of two captures, which keeps all captures in scratch memory)
  int pbase = s0 * kStagePix;                              // first pixel of the stage the next dma_stage() loads
  // 64-bit scalar bases of that stage's first dy / x piece of this wave
  const char* a_ptr = (const char*)(a.dy + m0) + ((long long)pbase + wave * RWA) * a.Co * 4;
  const char* b_ptr = (const char*)(a.x + n0) + ((long long)pbase - kHalo + (PB0 + wave) * RWB + (long long)dh * a.W) * a.Ci * 4;
  const long long step_a = (long long)4 * RWA * a.Co * 4, step_b = (long long)4 * RWB * a.Ci * 4;      // to this wave's next piece
  const long long stage_a = (long long)kStagePix * a.Co * 4, stage_b = (long long)kStagePix * a.Ci * 4;
  // A piece of x is padding iff its image row h + dh lies outside the image: with m = (pixel index) mod (H * W)
  // that is m < W for dh = -1 and m >= (H - 1) * W for dh = +1, i.e. valid iff lo <= m < lo + span.  m is tracked
  // per piece, advanced by one stage with an add and a conditional subtract.
  const int HW = a.H * a.W;
  const int m_lo = dh < 0 ? a.W : 0, m_span = (dh == 0 || (kS == 2 && dh > 0)) ? HW : HW - a.W;    // (stride 2: row 2 alpha + 1 always exists)
  const int adv_m = __builtin_amdgcn_readfirstlane(kStagePix % HW);
  int mrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
    mrow[i] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pbase - kHalo + (PB0 + wave + 4 * i) * RWB + HW) % (unsigned)HW));    // (+HW: >= 0)
  // stride 2: the column beta of each piece's first pixel (x pixel index of dy pixel q: 4 q - 2 beta + (kh-1) 2 W + plane)
  const int adv_b = __builtin_amdgcn_readfirstlane(kStagePix % a.W);
  int bcol[kS == 2 ? NB : 1];
  if constexpr (kS == 2) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      bcol[i] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pbase - kHalo + (PB0 + wave + 4 * i) * RWB + HW) % (unsigned)a.W));
  }
  const char* const x_tile = (const char*)(a.x + n0);
  const unsigned lds_a = lds_addr(&As[0][0][0]), lds_b = lds_addr(&Bs[0][0][0]);
  // The DMA of a stage is NA + NB pieces per wave, all scalar work: select the base address (the zero region for
  // padding, for the tail and when there is no next stage), issue, advance m.
  bool more = true;                                        // there is a stage to load
  auto dma_piece = [&](auto jc, int buf) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < NA) {
      const int piece = wave + 4 * j;
      const bool ok = more && pbase + piece * RWA < P;     // P % 4 == 0: a piece is all inside or all beyond the tail
      glds16(lane_a, ok ? a_ptr + j * step_a : (const char*)a.zero, lds_a + (unsigned)(buf * kStagePix * TM * 4 + piece * 1024));
    } else {
      constexpr int i = (j - NA) % NB, plane = (j - NA) / NB;
      const int piece = PB0 + wave + 4 * i;
      const int q0 = pbase - kHalo + piece * RWB;          // the pixel whose centre tap reads the piece's first row
      const bool ok = more && (unsigned)q0 < (unsigned)P && (unsigned)(mrow[i] - m_lo) < (unsigned)m_span;
      const char* src;
      if constexpr (kS == 1) src = b_ptr + i * step_b;
      else src = x_tile + ((long long)4 * q0 - 2 * bcol[i] + (long long)dh * 2 * a.W + plane) * a.Ci * 4;
      glds16(lane_b, ok ? src : (const char*)a.zero, lds_b + (unsigned)(((buf * kS + plane) * BROWS) * TN * 4 + piece * 1024));
      if constexpr (plane == kS - 1) {                     // (after the piece's last plane)
        mrow[i] += adv_m;
        mrow[i] -= mrow[i] >= HW ? HW : 0;
        if constexpr (kS == 2) {
          bcol[i] += adv_b;
          bcol[i] -= bcol[i] >= a.W ? a.W : 0;
        }
      }
    }
    if constexpr (j == NA + kS * NB - 1) {                 // the stage's last piece: on to the next stage
      pbase += kStagePix;
      a_ptr += stage_a;
      b_ptr += stage_b;
    }
  };
  auto dma_stage = [&](int buf) {                          // a whole stage at once (the prologue)
    static_for<0, NA + kS * NB>([&](auto jc) { dma_piece(jc, buf); });
  };

  // ---- fragments: lane l reads channel (l % 32) of pixel 2 * kk + l / 32 -- conflict-free ds_read_b32 from one
  // loop-invariant address per operand plus compile-time offsets
  constexpr int KP = kStagePix / 2;                       // k-pairs per stage
  const unsigned fa_base = lds_a + (unsigned)((lk * TM + wm * (TM / 2) + lr) * 4);
  const unsigned fb_base = lds_b + (unsigned)((lk * TN + wn * (TN / 2) + lr) * 4);
  float fa[2][BM], fb[2][3][BN];
  auto read_frags = [&](auto Qc, auto kkc, auto slotc) {
    constexpr int Q = decltype(Qc)::value, kk = decltype(kkc)::value, slot = decltype(slotc)::value;
    static_for<0, BM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      fa[slot][i] = lds_read<(Q * kStagePix + 2 * kk) * TM * 4 + i * 128>(fa_base);
    });
    static_for<0, 3>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      static_for<0, BN>([&](auto jc) {
        constexpr int jn = decltype(jc)::value;
        // stride 1: row of pixel + (kw - 1); stride 2: kw = 0 -> odd plane one pixel to the left, 1 -> even plane, 2 -> odd plane
        constexpr int plane = kS == 1 ? 0 : (t == 1 ? 0 : 1), row = kS == 1 ? kHalo - 1 + t : (t == 0 ? kHalo - 1 : kHalo);
        fb[slot][t][jn] = lds_read<((Q * kS + plane) * BROWS + 2 * kk + row) * TN * 4 + jn * 128>(fb_base);
      });
    });
  };
  // column of the image row that the k-pair being CONSUMED starts at (even: W % 4 == 0 and stages start at multiples of 32)
  int wk = __builtin_amdgcn_readfirstlane((int)((unsigned)(s0 * kStagePix) % (unsigned)a.W));

#ifdef T2O_CONV_DIAG
  unsigned kp_cycles[KP] = {}, dma_cycles = 0;
#endif
  // One continuous MFMA stream across stages.  Stage st (LDS buffer Q = parity of st - s0), KP k-pairs of
  // 3 * BM * BN MFMAs (64 cycles each):
  //   before k-pair 0  the LDS-DMA of stage st+1 is issued into the OTHER buffer (nobody reads it any more: its last
  //                    reads were before the previous stage's barrier) and flies under the whole stage
  //   k-pair  last-1   barrier (after this wave's DMA has landed): every wave's part of the next tile is there; the
  //                    matrix pipe still holds this wave's MFMAs to cover the skew
  //   k-pair  last     its fragment prefetch already reads k-pair 0 of stage st+1: no bubble at the stage boundary
  constexpr int NM = 3 * BM * BN;                         // MFMAs per k-pair
  constexpr int kSaluPerGap = (24 + NM - 1) / NM;         // ~24 scalar instructions per piece
  static_assert(NA + kS * NB <= KP - 3, "one DMA piece per k-pair, all issued well before the barrier");
  auto stage = [&](auto Qc, int st) {
    constexpr int Q = decltype(Qc)::value;
    const bool has_next = st + 1 < s1;
    more = has_next;
#ifdef T2O_CONV_DIAG
    unsigned long long tprev = __builtin_amdgcn_s_memtime();       // diagnostic build: cycles per k-pair position, summed over stages
#endif
    static_for<0, KP>([&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      constexpr int cur = kk & 1, nxt = cur ^ 1;
      lds_wait();                                         // the fragments of this k-pair (read one k-pair ago)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kk + 1 < KP) read_frags(Qc, std::integral_constant<int, kk + 1>{}, std::integral_constant<int, nxt>{});
      else read_frags(std::integral_constant<int, Q ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, nxt>{});
      // (unconditional: after the last stage the values are not used.  Under 'if (has_next)' the compiler merges the two
      // paths with register copies placed right behind the ds_reads -- it cannot know that an asm ds_read's result
      // arrives later -- and the copies pick up whatever was in the registers before)
      // image-row ends inside this k-pair (scalar conditions; one k-pair in W/2 has either)
      // (volatile asm inside the branches: as plain selects the compiler turns them into 6 unconditional v_cndmask per k-pair)
      if (wk == 0) {                                      // lanes of pixel 2kk sit at w == 0: kw = 0 reads pixel w-1, outside
#pragma unroll
        for (int jn = 0; jn < BN; ++jn) asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(fb[cur][0][jn]) : "s"(0x00000000ffffffffull));
      }
      wk += 2;
      if (wk == a.W) {                                    // lanes of pixel 2kk+1 sit at w == W-1: kw = 2 reads pixel w+1, outside
        wk = 0;                                           // (stride 2: column 2 beta + 1 always exists)
        if constexpr (kS == 1) {
#pragma unroll
          for (int jn = 0; jn < BN; ++jn) asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(fb[cur][2][jn]) : "s"(0xffffffff00000000ull));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the k-pair's MFMAs, and DMA piece kk of stage st+1 (into the OTHER buffer): ~20 scalar instructions that the
      // scheduler is told to deal out two per MFMA (in the shadow of the MFMA just issued) instead of in one block
      if constexpr (kk < NA + kS * NB) dma_piece(kkc, Q ^ 1);
      static_for<0, NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int t = m / (BM * BN), i = (m / BN) % BM, jn = m % BN;
        acc[t][i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][t][jn], acc[t][i][jn], 0, 0, 0);
      });
      if constexpr (kk < NA + kS * NB) {
#pragma unroll
        for (int g = 0; g < NM; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x004, kSaluPerGap, 0);      // scalar instructions of the piece
        }
      }
      if constexpr (kk == KP - 2) { __builtin_amdgcn_sched_barrier(0); glds_wait(); lds_wait(); __syncthreads(); }     // (pinned behind the k-pair's MFMAs)
      if constexpr (kk == KP - 1) lds_wait();             // (see k_conv3x3_fwd: prefetched fragments must have arrived where the compiler may copy them)
      __builtin_amdgcn_sched_barrier(0);
#ifdef T2O_CONV_DIAG
      { const unsigned long long tn = __builtin_amdgcn_s_memtime(); kp_cycles[kk] += (unsigned)(tn - tprev); tprev = tn; }
#endif
    });
  };

  if (s0 < s1) dma_stage(0);
  glds_wait();
  __syncthreads();
  if (s0 < s1) read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  lds_wait();
  __builtin_amdgcn_sched_barrier(0);
  for (int st = s0; st < s1; st += 2) {
    stage(std::integral_constant<int, 0>{}, st);
    if (st + 1 < s1) stage(std::integral_constant<int, 1>{}, st + 1);
  }
  lds_wait();

  if (a.stamps && threadIdx.x == 0) {
    a.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
#ifdef T2O_CONV_DIAG
    for (int kk = 0; kk < KP; ++kk) a.stamps[2 * gridDim.x + (size_t)blockIdx.x * (KP + 2) + kk] = kp_cycles[kk];
    a.stamps[2 * gridDim.x + (size_t)blockIdx.x * (KP + 2) + KP] = (unsigned long long)(s1 - s0);
    a.stamps[2 * gridDim.x + (size_t)blockIdx.x * (KP + 2) + KP + 1] = dma_cycles;
#endif
  }
  // C/D layout: column (n) = lane % 32, row (m) = (reg % 4) + 8 * (reg / 4) + 4 * (lane / 32)
  float* out = a.partial + (size_t)split * a.Co * 9 * a.Ci;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
      for (int jn = 0; jn < BN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int n = n0 + wn * (TN / 2) + jn * 32 + lr;
          out[((size_t)m * 9 + (kh * 3 + t)) * a.Ci + n] = acc[t][i][jn][r];
        }
}

// dw[i] = sum over splits of partial[s][i], in a fixed order.  A workgroup takes 32 consecutive float4 of the
// output; its 8 thread rows each add every 8th split (in split order), then the 8 partial sums are added in row
// order through LDS: deterministic, and 8 x as many loads in flight as one thread per output (with 168 splits of
// a 64-channel layer that serial loop took 50 us for 25 MB).
// acc != 0: the sum is ADDED to dw (a trainer's persistent, pre-zeroed gradient buffer; one writer per element).
__global__ __launch_bounds__(kConvThreads) void k_conv_wgrad_reduce(const float* partial, float* dw, size_t n4, int splits, size_t stride, int acc) {
  __shared__ float4 part[8][32];
  const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + col;
  float4 s = {0.0f, 0.0f, 0.0f, 0.0f};
  if (i < n4)
    // (8 independent loads in flight per thread, added in split order: the plain loop made 21 dependent round trips
    // for 168 splits -- 10 us per launch, 136 launches per train step)
    for (int k0 = row; k0 < splits; k0 += 64) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 8 * u;
        v[u] = k < splits ? ldg4(partial + (size_t)k * stride + 4 * i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  part[row][col] = s;
  __syncthreads();
  if (row == 0 && i < n4) {
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const float4 v = part[r][col];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (acc) {
      const float4 v = ldg4(dw + 4 * i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(dw + 4 * i) = s;
  }
}

// ================================================================================================================
// FORWARD / DATA GRADIENT   y[p][co] = sum_{kh,kw,ci} x[p + (kh-1) * W + (kw-1)][ci] * w[co][kh][kw][ci]   (zero padding)
//   One implicit GEMM: M = pixels, N = Co, K = 9 * Ci.  The data gradient is the same sum over dy with the weight
//   transposed and the taps mirrored (k_conv_flip_weight), so one kernel serves both.
//   Workgroup = 512 threads = 8 waves (4 along pixels x 2 along channels), one per CU: 256 consecutive pixels x 64
//   output channels; wave tile 64 x 32 (2 MFMA blocks, 32 accumulator registers).  The two waves of a SIMD belong
//   to the same workgroup, i.e. have the same age: the matrix pipe alternates between them (two 256-thread
//   workgroups per CU run one after the other instead -- the older wave wins every arbitration).
//   K is consumed in stages (kh, 32 input channels): the x tile is 272 pixel rows of 128 bytes (the 256 pixels with
//   a halo of 8 on both sides, for the image row h + kh - 1: the three kw taps read it shifted by one row), the w
//   tile 3 x 64 rows of 128 bytes; both double-buffered (118 KiB).
//   K runs along the 128-byte rows, and an MFMA wants a different row in every lane.  Rows are stored as 8 chunks
//   of 16 bytes with chunk c of row r at position c ^ ((r >> 1) & 7): lane l of a fragment reads chunk 2G + l / 32
//   of row l % 32 (+ tile offsets) with ONE conflict-free ds_read_b128 -- four k-steps of that lane (the reduction
//   order inside a group of 8 channels is 0..3 for lanes 0-31 and 4..7 for lanes 32-63, on both operands).  The
//   swizzle is applied on the global side of the LDS-DMA (which lane fetches which 16 bytes of a 128-byte line), so
//   every DMA piece is still 8 full lines.
//   Padding: a DMA piece (8 pixels, W % 8 == 0: inside one image row) whose input row lies outside the image reads
//   the zero region (scalar select, as in the weight gradient); a lane whose pixel sits in the first / last column
//   reads its kw = 0 / kw = 2 fragments from a zero row of the tile -- a loop-invariant address, no masking
//   instructions.  The loop has no vector-ALU instructions besides the MFMAs.
struct FwdArgs {
  const float* x;       // (N,H,W,Ci)
  const float* w;       // (Co,3,3,Ci)
  float* y;             // (N,H,W,Co)
  const float* zero;    // zero region (global), >= 8 * Ci * 4 bytes
  int N, H, W, Ci, Co;
  int tiles_p, tiles_n;
  float* stats;         // null, or (tiles_p, 2, Co): per pixel tile the sum and the sum of squares of y per channel
  const float* addend;  // null, or (N,H,W,Co): added to y in the epilogue (the data gradient of a block's first convolution
                        // plus the gradient its input receives through the identity shortcut: one pass instead of an add kernel)
  unsigned long long* stamps;   // diagnostic builds only
  // kBnb (data gradient in front of a batch norm + ReLU, y = relu(bn(bn_x))): the sums of the batch norm's backward, per pixel
  // tile and channel, of the gated gradient g = y' [bn_x * sc + sh > 0] and of g * xhat, straight from the accumulators
  const float* bn_x;            // (N,H,W,Co): the batch norm's input (the forward convolution's output)
  const float* bn_mean;         // (Co) batch mean, inverse standard deviation, gamma, beta
  const float* bn_invstd;
  const float* bn_w;
  const float* bn_b;
  float* bn_rows;               // (tiles_p, 2, Co)
};

constexpr int kFwdThreads = 512;
constexpr int kFwdCo = 64;            // output channels per workgroup
constexpr int kFwdHalo = 8;           // tile rows before the first pixel (1 needed; 8 = one DMA piece)
constexpr int kFwdWBuf = 3 * kFwdCo * 128;                      // [kw][co][32 ci]

// kBM: MFMA blocks per wave along the pixels; the workgroup takes 128 * kBM consecutive pixels (2: the normal tile;
// 1: layers whose 256-pixel tiles would leave CUs idle)
// kS = 2: the forward of a STRIDE-2 convolution.  a.H, a.W are then the OUTPUT grid (x is 2H x 2W) and the x tile is
// two planes, as in the stride-2 weight gradient: the even input columns 2 beta (kw = 1) and the odd ones 2 beta + 1
// (kw = 2, and kw = 0 one output pixel to the left).
// kAdd: a.addend is added in the epilogue.  Its 16 kBM values per lane are FETCHED at the top of the last loop iteration (one or
// two stages = 5-10 us before they are needed): with one workgroup per CU nothing else hides that round trip, and every
// workgroup of a round reaches its epilogue at the same time -- loading after the loop cost 20 us (128 channels) to 58 us (64
// channels: 67 MB) per launch over the same kernel without addend.
// kBnb: the epilogue also forms the backward sums of the batch norm this data gradient flows into (FwdArgs::bn_*): its input
// bn_x is fetched like the addend, the sums leave like the forward's statistics -- the batch norm's own sums pass (one read
// of the gradient and one of bn_x) is not launched.
template <int kBM, int kS = 1, bool kAdd = false, bool kBnb = false>
__global__ __launch_bounds__(kFwdThreads, 1) void k_conv3x3_fwd(FwdArgs a) {
  static_assert(!(kAdd && kBnb), "one prefetched epilogue operand");
  constexpr int kFwdPix = 128 * kBM;                              // output pixels per workgroup
  constexpr int kFwdXPieces = (kFwdPix + 2 * kFwdHalo) / 8;       // pieces of 8 rows (34 / 18)
  constexpr int kFwdPlane = kFwdXPieces * 1024;                   // one plane of the x tile
  constexpr int kFwdXBuf = (kS * kFwdXPieces + 1) * 1024;         // kS planes + one piece of zero rows
  constexpr int kFwdZeroRow = kS * kFwdXPieces * 8;
  __shared__ __attribute__((aligned(16))) char Xs[2][kFwdXBuf];
  __shared__ __attribute__((aligned(16))) char Ws[2][kFwdWBuf];

  // block -> (pixel tile, channel tile): the channel tiles of one pixel tile are neighbours inside one XCD (they
  // re-read the same x rows from its L2)
  const int b = blockIdx.x;
  const int xcd = b % 8, k8 = b / 8;
  const int pt = (k8 / a.tiles_n) * 8 + xcd, ct = k8 % a.tiles_n;
  if (pt >= a.tiles_p) return;
  const int P = a.N * a.H * a.W, HW = a.H * a.W;
  const int p0 = pt * kFwdPix, co0 = ct * kFwdCo;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ln = lane & 31, lh = lane >> 5;

  unsigned long long t0 = 0, r0 = 0;
  if (a.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

  f32x16 acc[kBM];
#pragma unroll
  for (int i = 0; i < kBM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  // ---- LDS-DMA pieces of this wave: x pieces wave, wave + 8, ... (< 34), w pieces (kw = 0..2, co rows 8*wave..)
  const unsigned lds_x = lds_addr(&Xs[0][0]), lds_w = lds_addr(&Ws[0][0]);
  constexpr int NXW = (kFwdXPieces + 7) / 8;              // x pieces per wave (the last one only for waves 0, 1)
  // lane l of a piece writes LDS bytes 16 l .. 16 l + 15 = (row l / 8, position l % 8) and fetches chunk
  // position ^ swizzle(row); rows of this wave's pieces are 8 * (wave + 8 i) + l / 8: the swizzle does not depend on i
  const int prow = lane >> 3, ppos = lane & 7;
  const int pswz = ((wave * 8 + prow) >> 1) & 7;
  const unsigned lane_x = (unsigned)(prow * kS * a.Ci * 4 + ((ppos ^ pswz) << 4));      // (stride 2: consecutive tile rows are two input pixels apart)
  const unsigned lane_w = (unsigned)(prow * 9 * a.Ci * 4 + ((ppos ^ pswz) << 4));    // (w tile rows 64 kw + 8 wave + l / 8)
  int mrow[NXW];                                          // (first pixel of the piece) mod (H * W): the image row it is in
#pragma unroll
  for (int i = 0; i < NXW; ++i)
    mrow[i] = __builtin_amdgcn_readfirstlane((int)((unsigned)(p0 - kFwdHalo + (wave + 8 * i) * 8 + HW) % (unsigned)HW));
  const long long xrow = (long long)a.Ci * 4;
  const char* const x0 = (const char*)a.x + ((long long)p0 - kFwdHalo + wave * 8) * xrow;       // this wave's first piece, kh = 1, ci = 0
  // stride 2: input pixel index of output pixel q = (t, beta): 4 q - 2 beta + (kh - 1) 2 W + plane
  int bcol[kS == 2 ? NXW : 1];
  if constexpr (kS == 2) {
#pragma unroll
    for (int i = 0; i < NXW; ++i)
      bcol[i] = __builtin_amdgcn_readfirstlane((int)((unsigned)(p0 - kFwdHalo + (wave + 8 * i) * 8 + HW) % (unsigned)a.W));
  }
  const char* const w0 = (const char*)a.w + (long long)(co0 + wave * 8) * 9 * xrow;            // its w rows, tap 0, ci = 0
  const int chunks = a.Ci / 32, stages = 3 * chunks;
  // stage st = kh * chunks + cc
  auto dma_piece = [&](auto jc, int buf, int kh, int cc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < kS * NXW) {
      constexpr int jx = j % NXW, plane = j / NXW;
      const int piece = wave + 8 * jx;
      if (piece < kFwdXPieces) {                           // wave-uniform
        const int q0 = p0 - kFwdHalo + piece * 8;          // output pixel of the piece's first row
        const int dh = kh - 1;
        const int lo = dh < 0 ? a.W : 0, span = (dh == 0 || (kS == 2 && dh > 0)) ? HW : HW - a.W;     // (stride 2: row 2 alpha + 1 always exists)
        const bool ok = (unsigned)q0 < (unsigned)P && (unsigned)(mrow[jx] - lo) < (unsigned)span;
        const char* src;
        if constexpr (kS == 1) src = x0 + ((long long)jx * 64 + (long long)dh * a.W) * xrow + cc * 128;
        else src = (const char*)a.x + ((long long)4 * q0 - 2 * bcol[jx] + (long long)dh * 2 * a.W + plane) * xrow + cc * 128;
        glds16(lane_x, ok ? src : (const char*)a.zero, lds_x + (unsigned)(buf * kFwdXBuf + plane * kFwdPlane + piece * 1024));
      }
    } else {
      constexpr int kw = j - kS * NXW;
      const char* src = w0 + (long long)(kh * 3 + kw) * xrow + cc * 128;
      glds16(lane_w, src, lds_w + (unsigned)(buf * kFwdWBuf + (kw * 8 + wave) * 1024));
    }
  };
  constexpr int NPW = kS * NXW + 3;                        // pieces per wave and stage
#ifndef T2O_FWD_DMA_GROUPS
#define T2O_FWD_DMA_GROUPS 2
#endif
  constexpr int kDmaGroups = T2O_FWD_DMA_GROUPS;           // the first groups of a stage carry the next stage's DMA

  // ---- fragment addresses (loop-invariant byte offsets into Xs / Ws): A = x rows (pixels), B = w rows (channels)
  unsigned xa[3][kBM][4], wb[4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int i = 0; i < kBM; ++i) {
      const int pl = wm * 32 * kBM + i * 32 + ln;          // pixel inside the tile
      const int wcol = (int)((unsigned)(p0 + pl) % (unsigned)a.W);
      // stride 1: row of pixel + (kw - 1); stride 2: kw = 0 -> odd plane one pixel to the left, 1 -> even plane, 2 -> odd plane
      const bool outside = (kw == 0 && wcol == 0) || (kS == 1 && kw == 2 && wcol == a.W - 1);
      const int inrow = kS == 1 ? kFwdHalo + pl + kw - 1 : (kw == 1 ? 0 : kFwdXPieces * 8) + kFwdHalo + pl - (kw == 0 ? 1 : 0);
      const int row = outside ? kFwdZeroRow : inrow;
#pragma unroll
      for (int g = 0; g < 4; ++g) xa[kw][i][g] = (unsigned)(row * 128 + (((2 * g + lh) ^ ((row >> 1) & 7)) << 4));
    }
  {
    const int row = wn * 32 + ln;
#pragma unroll
    for (int g = 0; g < 4; ++g) wb[g] = (unsigned)(row * 128 + (((2 * g + lh) ^ ((row >> 1) & 7)) << 4));
  }

  // the zero rows of both x buffers (never written again)
  if (wave == 0) {
    glds16((unsigned)(lane * 16), a.zero, lds_x + (unsigned)(kS * kFwdPlane));
    glds16((unsigned)(lane * 16), a.zero, lds_x + (unsigned)(kFwdXBuf + kS * kFwdPlane));
  }

  float4 fa[2][3][kBM], fb[2][3];
  auto read_frags = [&](auto Qc, auto gc, auto slotc) {
    constexpr int Q = decltype(Qc)::value, g = decltype(gc)::value, slot = decltype(slotc)::value;
    static_for<0, 3>([&](auto kwc) {
      constexpr int kw = decltype(kwc)::value;
      // (plain loads, not asm: these values live across the loop's back edge, where the compiler places register
      // copies -- it has to know that a ds_read's result arrives later.  Address = loop-invariant register + immediate.)
#pragma unroll
      for (int i = 0; i < kBM; ++i) fa[slot][kw][i] = *reinterpret_cast<const float4*>(&Xs[Q][0] + xa[kw][i][g]);
      fb[slot][kw] = *reinterpret_cast<const float4*>(&Ws[Q][0] + kw * kFwdCo * 128 + wb[g]);
    });
  };

  // One continuous MFMA stream across stages: 4 groups of 8 input channels = 24 MFMAs each.  Groups 0 and 1 carry
  // the DMA pieces of the next stage (scalar instructions dealt out between the MFMAs), after group 2 the barrier
  // publishes them, group 3 already prefetches the next stage's first fragments.
#ifdef T2O_CONV_DIAG
  unsigned g_cycles[4] = {};
  unsigned long long t_loop = 0;
#endif
  auto stage = [&](auto Qc, int st) {
    constexpr int Q = decltype(Qc)::value;
    const bool has_next = st + 1 < stages;
#ifdef T2O_CONV_DIAG
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    const int nst = has_next ? st + 1 : st;                // (the last stage reloads itself into the idle buffer)
    const int nkh = nst / chunks, ncc = nst - nkh * chunks;
    static_for<0, 4>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int cur = g & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + 1 < 4) read_frags(Qc, std::integral_constant<int, g + 1>{}, std::integral_constant<int, nxt>{});
      else read_frags(std::integral_constant<int, Q ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, nxt>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g < kDmaGroups) {
        static_for<0, (NPW + kDmaGroups - 1) / kDmaGroups>([&](auto jc) {
          constexpr int j = g * ((NPW + kDmaGroups - 1) / kDmaGroups) + decltype(jc)::value;
          if constexpr (j < NPW) dma_piece(std::integral_constant<int, j>{}, Q ^ 1, nkh, ncc);
        });
      }
      static_for<0, 3>([&](auto kwc) {
        constexpr int kw = decltype(kwc)::value;
        static_for<0, 4>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
#pragma unroll
          for (int i = 0; i < kBM; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][kw][i][s], fb[cur][kw][s], acc[i], 0, 0, 0);
        });
      });
      if constexpr (g < kDmaGroups) {                     // deal the pieces' scalar instructions out between the MFMAs
#pragma unroll
        for (int q = 0; q < 12 * kBM; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, (NPW * 16 / kDmaGroups + 12 * kBM - 1) / (12 * kBM), 0);
        }
      }
      if constexpr (g == 2) { __builtin_amdgcn_sched_barrier(0); glds_wait(); __syncthreads(); }     // (pinned behind the group's MFMAs)
      __builtin_amdgcn_sched_barrier(0);
#ifdef T2O_CONV_DIAG
      { const unsigned long long tn = __builtin_amdgcn_s_memtime(); g_cycles[g] += (unsigned)(tn - tprev); tprev = tn; }
#endif
    });
  };

  static_for<0, NPW>([&](auto jc) { dma_piece(jc, 0, 0, 0); });
  glds_wait();
  __syncthreads();
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
#ifdef T2O_CONV_DIAG
  t_loop = __builtin_amdgcn_s_memtime();
#endif
  float pre[(kAdd || kBnb) ? kBM : 1][16];
  for (int st = 0; st < stages; st += 2) {
    if constexpr (kAdd || kBnb) {
      if (st + 2 >= stages) {                              // (uniform) the last iteration: fetch the operand under its MFMAs
        const float* __restrict__ src = kAdd ? a.addend : a.bn_x;
#pragma unroll
        for (int i = 0; i < kBM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int p = p0 + wm * 32 * kBM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            pre[i][r] = p < P ? src[(size_t)p * a.Co + co0 + wn * 32 + ln] : 0.0f;
          }
      }
    }
    stage(std::integral_constant<int, 0>{}, st);
    if (st + 1 < stages) stage(std::integral_constant<int, 1>{}, st + 1);
  }
#ifdef T2O_CONV_DIAG
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
#endif

  // C/D layout: column (n = channel) = lane % 32, row (m = pixel) = (reg % 4) + 8 * (reg / 4) + 4 * (lane / 32)
#pragma unroll
  for (int i = 0; i < kBM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + wm * 32 * kBM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (p < P) {
        const size_t o = (size_t)p * a.Co + co0 + wn * 32 + ln;
        if constexpr (kAdd) a.y[o] = acc[i][r] + pre[i][r];
        else a.y[o] = acc[i][r];
      }
    }
  // Batch-norm statistics of the layer that follows, straight from the accumulators (a lane holds 16 * kBM pixels of
  // ONE channel): the statistics pass over y (one full read of the activation) is not needed.  Rows past the end
  // of the image batch accumulated zeros.  Fixed order: lane, its partner lane + 32, the four pixel waves.
  if (kBnb || a.stats) {               // (kernel argument: uniform)
    __shared__ float red[2][4][kFwdCo];
    float s1 = 0.0f, s2 = 0.0f;
    if constexpr (kBnb) {
      // the gate exactly as the batch norm's own backward evaluates it (t2o_norm.hip gated<false>): x * sc + sh > 0
      const int c = co0 + wn * 32 + ln;
      const float mean = a.bn_mean[c], invstd = a.bn_invstd[c];
      const float sc = a.bn_w[c] * invstd, sh = a.bn_b[c] - mean * sc;
#pragma unroll
      for (int i = 0; i < kBM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = pre[i][r];
          const float g = (x * sc + sh > 0.0f) ? acc[i][r] : 0.0f;        // (rows past the batch: acc = 0)
          s1 += g;
          s2 += g * ((x - mean) * invstd);
        }
    } else {
#pragma unroll
      for (int i = 0; i < kBM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1 += acc[i][r]; s2 += acc[i][r] * acc[i][r]; }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lh == 0) { red[0][wm][wn * 32 + ln] = s1; red[1][wm][wn * 32 + ln] = s2; }
    // (not __syncthreads(): that also waits for this wave's output stores to be acknowledged, ~2 us with nothing else on the CU)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < 2 * kFwdCo) {
      const int which = tid / kFwdCo, c = tid % kFwdCo;
      (kBnb ? a.bn_rows : a.stats)[((size_t)pt * 2 + which) * a.Co + co0 + c] = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
    }
  }
#ifdef T2O_CONV_DIAG
  if (a.stamps && tid == 0) {          // [start, prologue, loop, g0..g3, epilogue issue] per workgroup
    unsigned long long* q = a.stamps + (size_t)blockIdx.x * 8;
    q[0] = r0; q[1] = t_loop - t0; q[2] = t_end - t_loop;
    for (int g = 0; g < 3; ++g) q[3 + g] = g_cycles[g];
    q[6] = __builtin_amdgcn_s_memrealtime();             // (100 MHz, one counter for the chip)
    q[7] = __builtin_amdgcn_s_memtime() - t_end;
  }
#endif
}

// ================================================================================================================
// BATCHED GEMM   C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]      (both operands K-contiguous, fp32, one rounding order)
//   The 16 multiplications of a Winograd F(2x2,3x3) layer (t2o_winograd.hip): A = V[xi] (tiles x Ci), B = U[xi] (Co x Ci),
//   C = M[xi].  It is the forward convolution kernel above without taps, halos and padding: 512 threads = 8 waves
//   (4 along m x 2 along n), one workgroup per CU, tile 256 x (64 kBN), wave tile 64 x (32 kBN) = 2 x kBN MFMA blocks;
//   K in stages of 32 (rows of 128 bytes, LDS-DMA pieces of 8 rows, the c ^ ((r >> 1) & 7) chunk swizzle, one
//   conflict-free ds_read_b128 per fragment and 4 k-steps); A and B double-buffered (96 KiB at kBN = 2).  Per stage and
//   wave: 16 kBN MFMAs per group of 8 k, 4 groups; 4 + kBN DMA pieces dealt over groups 0 and 1; no vector-ALU instruction
//   in the loop besides the MFMAs.  Rows past M re-read row M - 1 (their results are not stored).
struct GemmNtArgs {
  const float* A;       // (batches, a_rows >= M, K): rows past M are never read
  const float* B;       // (batches, N, K)
  float* C;             // (batches, M, N)
  int M, N, K, batches, a_rows;
  int tiles_m, tiles_n;
};

template <int kBN>
__global__ __launch_bounds__(kFwdThreads, 1) void k_gemm_nt(GemmNtArgs a) {
  constexpr int kTM = 256, kTN = 64 * kBN;
  constexpr int kABuf = kTM * 128, kBBuf = kTN * 128;
  constexpr int kAPw = kTM / 64, kBPw = kTN / 64;          // DMA pieces (8 rows) per wave and stage: 4 of A, kBN of B
  __shared__ __attribute__((aligned(16))) char As[2][kABuf];
  __shared__ __attribute__((aligned(16))) char Bs[2][kBBuf];

  // block -> (row tile R = batch * tiles_m + m tile, n tile): the n tiles of one row tile are neighbours inside one XCD
  const int blk = blockIdx.x;
  const int xcd = blk % 8, k8 = blk / 8;
  const int R = (k8 / a.tiles_n) * 8 + xcd, nt = k8 % a.tiles_n;
  if (R >= a.batches * a.tiles_m) return;
  const int batch = R / a.tiles_m, mt = R - batch * a.tiles_m;
  const int m0 = mt * kTM, n0 = nt * kTN;
  const char* const Ab = (const char*)(a.A + (size_t)batch * a.a_rows * a.K);
  const char* const Bb = (const char*)(a.B + (size_t)batch * a.N * a.K);
  float* const Cb = a.C + (size_t)batch * a.M * a.N;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ln = lane & 31, lh = lane >> 5;

  f32x16 acc[2][kBN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kBN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- LDS-DMA: lane l of a piece writes LDS bytes 16 l .. = (row l / 8, position l % 8) and fetches chunk position ^ swizzle(row)
  const unsigned lds_a = lds_addr(&As[0][0]), lds_b = lds_addr(&Bs[0][0]);
  const int prow = lane >> 3, ppos = lane & 7;
  const int pswz = ((wave * 8 + prow) >> 1) & 7;           // (pieces wave + 8 i: the swizzle does not depend on i)
  const unsigned kbytes = (unsigned)a.K * 4u;
  unsigned voffA[kAPw], voffB[kBPw];                        // loop-invariant lane offsets from the operand's (batch, k chunk) base
#pragma unroll
  for (int i = 0; i < kAPw; ++i) {
    int row = m0 + (wave + 8 * i) * 8 + prow;
    row = row < a.M ? row : a.M - 1;
    voffA[i] = (unsigned)row * kbytes + (unsigned)((ppos ^ pswz) << 4);
  }
#pragma unroll
  for (int j = 0; j < kBPw; ++j) voffB[j] = (unsigned)(n0 + (wave + 8 * j) * 8 + prow) * kbytes + (unsigned)((ppos ^ pswz) << 4);
  const int stages = a.K / 32;
  auto dma_piece = [&](auto jc, int buf, int cc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < kAPw) glds16(voffA[j], Ab + cc * 128, lds_a + (unsigned)(buf * kABuf + (wave + 8 * j) * 1024));
    else glds16(voffB[j - kAPw], Bb + cc * 128, lds_b + (unsigned)(buf * kBBuf + (wave + 8 * (j - kAPw)) * 1024));
  };
  constexpr int NPW = kAPw + kBPw;
  constexpr int kDmaGroups = 2;

  // ---- fragment addresses (loop-invariant byte offsets)
  unsigned fa_off[2][4], fb_off[kBN][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + ln;
#pragma unroll
    for (int g = 0; g < 4; ++g) fa_off[i][g] = (unsigned)(row * 128 + (((2 * g + lh) ^ ((row >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int j = 0; j < kBN; ++j) {
    const int row = wn * 32 * kBN + j * 32 + ln;
#pragma unroll
    for (int g = 0; g < 4; ++g) fb_off[j][g] = (unsigned)(row * 128 + (((2 * g + lh) ^ ((row >> 1) & 7)) << 4));
  }

  float4 fa[2][2], fb[2][kBN];
  auto read_frags = [&](auto Qc, auto gc, auto slotc) {
    constexpr int Q = decltype(Qc)::value, g = decltype(gc)::value, slot = decltype(slotc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[slot][i] = *reinterpret_cast<const float4*>(&As[Q][0] + fa_off[i][g]);
#pragma unroll
    for (int j = 0; j < kBN; ++j) fb[slot][j] = *reinterpret_cast<const float4*>(&Bs[Q][0] + fb_off[j][g]);
  };

  auto stage = [&](auto Qc, int st) {
    constexpr int Q = decltype(Qc)::value;
    const int ncc = st + 1 < stages ? st + 1 : st;         // (the last stage reloads itself into the idle buffer)
    static_for<0, 4>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int cur = g & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + 1 < 4) read_frags(Qc, std::integral_constant<int, g + 1>{}, std::integral_constant<int, nxt>{});
      else read_frags(std::integral_constant<int, Q ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, nxt>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g < kDmaGroups) {
        static_for<0, (NPW + kDmaGroups - 1) / kDmaGroups>([&](auto jc) {
          constexpr int j = g * ((NPW + kDmaGroups - 1) / kDmaGroups) + decltype(jc)::value;
          if constexpr (j < NPW) dma_piece(std::integral_constant<int, j>{}, Q ^ 1, ncc);
        });
      }
      static_for<0, 4>([&](auto sc) {
        constexpr int sidx = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < kBN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][sidx], fb[cur][j][sidx], acc[i][j], 0, 0, 0);
      });
      if constexpr (g < kDmaGroups) {                     // deal the pieces' scalar instructions out between the MFMAs
#pragma unroll
        for (int q = 0; q < 8 * kBN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, (NPW * 16 / kDmaGroups + 8 * kBN - 1) / (8 * kBN), 0);
        }
      }
      if constexpr (g == 2) { __builtin_amdgcn_sched_barrier(0); glds_wait(); __syncthreads(); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  static_for<0, NPW>([&](auto jc) { dma_piece(jc, 0, 0); });
  glds_wait();
  __syncthreads();
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  for (int st = 0; st < stages; st += 2) {
    stage(std::integral_constant<int, 0>{}, st);
    if (st + 1 < stages) stage(std::integral_constant<int, 1>{}, st + 1);
  }

  // C/D layout: column (n) = lane % 32, row (m) = (reg % 4) + 8 * (reg / 4) + 4 * (lane / 32)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < a.M) {
        float* dst = Cb + (size_t)m * a.N + n0 + wn * 32 * kBN + ln;
#pragma unroll
        for (int j = 0; j < kBN; ++j) dst[32 * j] = acc[i][j][r];
      }
    }
}

// BATCHED GEMM over the rows   C[s][b][m][n] = sum_{t in split s} A[b][t][m] * B[b][t][n]     ("TN": K = the row index)
//   The weight gradient of a Winograd layer: A = A dY A^T (tiles x Co), B = V (tiles x Ci), C = dU[xi] (Co x Ci), the tile
//   range cut into `splits` pieces (fixed partition, partial results added in order by k_wino_dw: deterministic).
//   Tile 128 x 128, K stage = 64 rows of both operands (2 x 32 KiB, double-buffered: 128 KiB), 8 waves = 4 (m) x 2 (n),
//   wave tile 32 x 64.  Rows are m / n contiguous, which is the MFMA operand order already: lane (i, k) of a k-step reads
//   word (row 2 step + k, column base + i) -- plain ds_read_b32, conflict-free without a swizzle, immediate row offsets.
//   Both operands are padded with zero rows to a multiple of 64 * splits by their producers (no edge handling here).
struct GemmTnArgs {
  const float* A;       // (batches, Tpad, M)
  const float* B;       // (batches, Tpad, N)
  float* C;             // (splits, batches, M, N)
  int M, N, Tpad, batches, splits;
  int tiles_m, tiles_n;
  int ldA, ldB;         // rows per plane of the tensors A / B live in (>= Tpad: a row range of a larger (batches, R, .) arena)
};

__global__ __launch_bounds__(kFwdThreads, 1) void k_gemm_tn(GemmTnArgs a) {
  constexpr int kT = 128, kRows = 64, kBuf = kRows * kT * 4;      // 32 KiB per operand and stage
  __shared__ __attribute__((aligned(16))) char As[2][kBuf];
  __shared__ __attribute__((aligned(16))) char Bs[2][kBuf];

  const int blk = blockIdx.x;
  const int xcd = blk % 8, k8 = blk / 8;
  const int R = (k8 / a.tiles_n) * 8 + xcd, nt = k8 % a.tiles_n;
  if (R >= a.splits * a.batches * a.tiles_m) return;
  const int sb = R / a.tiles_m, mt = R - sb * a.tiles_m;          // sb = split * batches + batch
  const int split = sb / a.batches, batch = sb - split * a.batches;
  const int m0 = mt * kT, n0 = nt * kT;
  const int rows_per_split = a.Tpad / a.splits, t0 = split * rows_per_split;
  const char* const Ab = (const char*)(a.A + ((size_t)batch * a.ldA + t0) * a.M + m0);
  const char* const Bb = (const char*)(a.B + ((size_t)batch * a.ldB + t0) * a.N + n0);
  float* const Cb = a.C + (size_t)sb * a.M * a.N;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ln = lane & 31, lh = lane >> 5;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

  // ---- LDS-DMA: a piece = 2 rows x 512 bytes; lane l fetches row l / 32, bytes 16 (l % 32) .., lands at LDS 16 l (row-major tile)
  const unsigned lds_a = lds_addr(&As[0][0]), lds_b = lds_addr(&Bs[0][0]);
  const unsigned voffA = (unsigned)(lane >> 5) * (unsigned)a.M * 4u + (unsigned)((lane & 31) << 4);
  const unsigned voffB = (unsigned)(lane >> 5) * (unsigned)a.N * 4u + (unsigned)((lane & 31) << 4);
  const long long arow = (long long)a.M * 4, brow = (long long)a.N * 4;
  const int stages = rows_per_split / kRows;
  // pieces of this wave: A pieces wave, wave + 8, +16, +24 (rows 2 piece ..), the same of B
  auto dma_piece = [&](auto jc, int buf, int st) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < 4) {
      const int piece = wave + 8 * j;
      glds16(voffA, Ab + ((long long)st * kRows + 2 * piece) * arow, lds_a + (unsigned)(buf * kBuf + piece * 1024));
    } else {
      const int piece = wave + 8 * (j - 4);
      glds16(voffB, Bb + ((long long)st * kRows + 2 * piece) * brow, lds_b + (unsigned)(buf * kBuf + piece * 1024));
    }
  };
  constexpr int NPW = 8, kDmaGroups = 2;

  // ---- fragment addresses: word (row lh, column base + ln) of the stage tile; k-step s adds 2 s rows = 1024 s bytes
  const unsigned fa_off = (unsigned)(lh * 512 + (wm * 32 + ln) * 4);
  const unsigned fb_off0 = (unsigned)(lh * 512 + (wn * 64 + ln) * 4), fb_off1 = fb_off0 + 128;

  float fa[2][8], fb[2][2][8];
  auto read_frags = [&](auto Qc, auto gc, auto slotc) {
    constexpr int Q = decltype(Qc)::value, g = decltype(gc)::value, slot = decltype(slotc)::value;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      fa[slot][s] = *reinterpret_cast<const float*>(&As[Q][0] + fa_off + (g * 8 + s) * 1024);
      fb[slot][0][s] = *reinterpret_cast<const float*>(&Bs[Q][0] + fb_off0 + (g * 8 + s) * 1024);
      fb[slot][1][s] = *reinterpret_cast<const float*>(&Bs[Q][0] + fb_off1 + (g * 8 + s) * 1024);
    }
  };

  auto stage = [&](auto Qc, int st) {
    constexpr int Q = decltype(Qc)::value;
    const int nst = st + 1 < stages ? st + 1 : st;         // (the last stage reloads itself into the idle buffer)
    static_for<0, 4>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int cur = g & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + 1 < 4) read_frags(Qc, std::integral_constant<int, g + 1>{}, std::integral_constant<int, nxt>{});
      else read_frags(std::integral_constant<int, Q ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, nxt>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g < kDmaGroups) {
        static_for<0, NPW / kDmaGroups>([&](auto jc) {
          constexpr int j = g * (NPW / kDmaGroups) + decltype(jc)::value;
          dma_piece(std::integral_constant<int, j>{}, Q ^ 1, nst);
        });
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][0][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][1][s], acc[1], 0, 0, 0);
      }
      if constexpr (g < kDmaGroups) {                     // deal the pieces' scalar instructions out between the MFMAs
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
        }
      }
      if constexpr (g == 2) { __builtin_amdgcn_sched_barrier(0); glds_wait(); __syncthreads(); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  static_for<0, NPW>([&](auto jc) { dma_piece(jc, 0, 0); });
  glds_wait();
  __syncthreads();
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  for (int st = 0; st < stages; st += 2) {
    stage(std::integral_constant<int, 0>{}, st);
    if (st + 1 < stages) stage(std::integral_constant<int, 1>{}, st + 1);
  }

  // C/D layout: column (n) = lane % 32, row (m) = (reg % 4) + 8 * (reg / 4) + 4 * (lane / 32)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    float* dst = Cb + (size_t)m * a.N + n0 + wn * 64 + ln;
    dst[0] = acc[0][r];
    dst[32] = acc[1][r];
  }
}

// wt[ci][2-kh][2-kw][co] = w[co][kh][kw][ci] (flip: the stride-1 data gradient is the forward kernel on dy with these
// weights) or wt[ci][kh][kw][co] = w[co][kh][kw][ci] (no flip: the stride-2 data gradient indexes taps itself)
// (taps = gridDim.z: 9 for the 3x3 layers, 1 for the 1x1 shortcuts -- a plain transpose)
__global__ __launch_bounds__(kConvThreads) void k_conv_flip_weight(const float* w, float* wt, int Co, int Ci, int flip) {
  __shared__ float tile[32][33];
  const int taps = gridDim.z, tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) tile[r][tx] = w[((size_t)(co0 + r) * taps + tap) * Ci + ci0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) wt[((size_t)(ci0 + r) * taps + (flip ? taps - 1 - tap : tap)) * Co + co0 + tx] = tile[tx][r];
}

// ================================================================================================================
// DATA GRADIENT OF THE STRIDE-2 CONVOLUTIONS (the first convolution of every encoder stage, models/actor_resnet.py:32-36)
//   y[n][a][b][co] = sum x[n][2a+kh-1][2b+kw-1][ci] w[co][kh][kw][ci]   ==>   with i = 2 alpha + ph, j = 2 beta + pw:
//   dx[n][i][j][ci] = sum over the taps with kh = ph + 1 (mod 2), kw = pw + 1 (mod 2) of
//                     dy[n][alpha + (kh == 0)][beta + (kw == 0)][co] * w[co][kh][kw][ci]
//   i.e. every pixel (alpha, beta) of the dy grid produces the 2 x 2 block of dx pixels above it from its own row
//   and the next one, its own column and the next one: 1 + 2 + 2 + 4 = 9 taps for 4 outputs, no multiplications by
//   inserted zeros.  Same machinery as k_conv3x3_fwd (x := dy, reduction over co, weights transposed to
//   (Ci,3,3,Co)): a workgroup of 8 waves takes 128 consecutive dy pixels x 64 input channels and keeps FOUR
//   accumulators per wave (one per output parity).  Stages alternate between the dy rows alpha (6 taps: kh = 1, 2)
//   and alpha + 1 (3 taps: kh = 0) of a 32-channel chunk -- which is also the double buffering.
struct Dgrad2Args {
  const float* dy;      // (N,Ho,Wo,Co)
  const float* wt;      // (Ci,3,3,Co)
  float* dx;            // (N,2Ho,2Wo,Ci)
  const float* zero;
  int N, Ho, Wo, Ci, Co;
  int tiles_p, tiles_n;
};

// kWN: waves along the input channels.  2: 8 waves, 128 dy pixels x 64 channels per workgroup (the normal tile).  1: 4 waves,
// 128 x 32 -- for layers whose 64-channel tiles number fewer than half the CUs (stage 4 at bs = 64: 32 pixel tiles x 4 = 128
// workgroups took 135 us where the same FLOP on the other stages take 77-83); a wave's instruction stream is the same, a
// SIMD holds one wave instead of two.
template <int kWN>
__global__ __launch_bounds__(256 * kWN, 1) void k_conv3x3s2_dgrad(Dgrad2Args a) {
  constexpr int NW = 4 * kWN;                                    // waves
  constexpr int kPix = 128;
  constexpr int kXPieces = (kPix + 2 * kFwdHalo) / 8;            // 18
  constexpr int kXBuf = (kXPieces + 1) * 1024;                   // + zero rows
  constexpr int kZeroRow = kXPieces * 8;
  constexpr int kTapBytes = 32 * kWN * 128;                      // one tap of the w tile: 32 kWN input channels (rows) x 32 reduction channels
  __shared__ __attribute__((aligned(16))) char Xs[2][kXBuf];     // [0]: rows alpha (stage A), [1]: rows alpha + 1 (stage B)
  __shared__ __attribute__((aligned(16))) char WsA[6 * kTapBytes];
  __shared__ __attribute__((aligned(16))) char WsB[3 * kTapBytes];

  const int b = blockIdx.x;
  const int xcd = b % 8, k8 = b / 8;
  const int pt = (k8 / a.tiles_n) * 8 + xcd, ct = k8 % a.tiles_n;
  if (pt >= a.tiles_p) return;
  const int P = a.N * a.Ho * a.Wo, HW = a.Ho * a.Wo;
  const int p0 = pt * kPix, n0 = ct * 32 * kWN;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / kWN, wn = wave % kWN;
  const int ln = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];                                       // [ph][pw]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c >> 1][c & 1][r] = 0.0f;

  const unsigned lds_x = lds_addr(&Xs[0][0]), lds_wa = lds_addr(&WsA[0]), lds_wb = lds_addr(&WsB[0]);
  constexpr int NXW = (kXPieces + NW - 1) / NW;           // 3 / 5 (the last one only for waves 0, 1)
  const int prow = lane >> 3, ppos = lane & 7;
  const int pswz = ((wave * 8 + prow) >> 1) & 7;
  const unsigned lane_x = (unsigned)(prow * a.Co * 4 + ((ppos ^ pswz) << 4));
  const unsigned lane_w = (unsigned)(prow * 9 * a.Co * 4 + ((ppos ^ pswz) << 4));
  int mrow[NXW];
#pragma unroll
  for (int i = 0; i < NXW; ++i)
    mrow[i] = __builtin_amdgcn_readfirstlane((int)((unsigned)(p0 - kFwdHalo + (wave + NW * i) * 8 + HW) % (unsigned)HW));
  const long long xrow = (long long)a.Co * 4;
  const char* const x0 = (const char*)a.dy + ((long long)p0 - kFwdHalo + wave * 8) * xrow;
  const char* const w0 = (const char*)a.wt + (long long)(n0 + wave * 8) * 9 * xrow;
  const int chunks = a.Co / 32;
  // pieces of a stage's data: j < NXW: dy rows (dr = 0 for stage A, 1 for stage B); then its taps' weight rows
  auto dma_piece = [&](auto typec, auto jc, int cc) {
    constexpr int type = decltype(typec)::value, j = decltype(jc)::value;      // type 0 = A, 1 = B
    if constexpr (j < NXW) {
      const int piece = wave + NW * j;
      if (piece < kXPieces) {                              // wave-uniform
        const int q0 = p0 - kFwdHalo + piece * 8;
        const bool ok = (unsigned)q0 < (unsigned)P && (type == 0 || mrow[j] < HW - a.Wo);      // B: the row alpha + 1 exists
        const char* src = x0 + ((long long)j * NW * 8 + (long long)type * a.Wo) * xrow + cc * 128;
        glds16(lane_x, ok ? src : (const char*)a.zero, lds_x + (unsigned)(type * kXBuf + piece * 1024));
      }
    } else {
      constexpr int t = j - NXW;                           // A: taps (kh = 1 + t / 3, kw = t % 3); B: (kh = 0, kw = t)
      constexpr int tap = type == 0 ? 3 + t : t;           // kh * 3 + kw
      const char* src = w0 + (long long)tap * xrow + cc * 128;
      glds16(lane_w, src, (type == 0 ? lds_wa : lds_wb) + (unsigned)((t * NW + wave) * 1024));
    }
  };
  constexpr int NPA = NXW + 6, NPB = NXW + 3;             // pieces per wave of a stage A / B

  // fragment addresses: dy rows for the column shifts dc = 0, 1 (dc = 1 beyond the last column: zero row); w rows
  unsigned xa[2][4], wb[4];
  {
    const int pl = wm * 32 + ln;
    const int bcol = (int)((unsigned)(p0 + pl) % (unsigned)a.Wo);
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) {
      const int row = (dc == 1 && bcol == a.Wo - 1) ? kZeroRow : kFwdHalo + pl + dc;
#pragma unroll
      for (int g = 0; g < 4; ++g) xa[dc][g] = (unsigned)(row * 128 + (((2 * g + lh) ^ ((row >> 1) & 7)) << 4));
    }
    const int row = wn * 32 + ln;
#pragma unroll
    for (int g = 0; g < 4; ++g) wb[g] = (unsigned)(row * 128 + (((2 * g + lh) ^ ((row >> 1) & 7)) << 4));
  }
  if (wave == 0) {
    glds16((unsigned)(lane * 16), a.zero, lds_x + (unsigned)(kXPieces * 1024));
    glds16((unsigned)(lane * 16), a.zero, lds_x + (unsigned)(kXBuf + kXPieces * 1024));
  }

  float4 fa[2][2], fb[2][6];
  auto read_frags = [&](auto typec, auto gc, auto slotc) {
    constexpr int type = decltype(typec)::value, g = decltype(gc)::value, slot = decltype(slotc)::value;
    fa[slot][0] = *reinterpret_cast<const float4*>(&Xs[type][0] + xa[0][g]);
    fa[slot][1] = *reinterpret_cast<const float4*>(&Xs[type][0] + xa[1][g]);
    static_for<0, (type == 0 ? 6 : 3)>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      fb[slot][t] = *reinterpret_cast<const float4*>((type == 0 ? &WsA[0] : &WsB[0]) + t * kTapBytes + wb[g]);
    });
  };

  // one stage = 4 groups of 8 reduction channels; the next stage's data is loaded during groups 0 and 1
  auto stage = [&](auto typec, int cc, bool more) {
    constexpr int type = decltype(typec)::value;
    constexpr int NT = type == 0 ? 6 : 3;
    constexpr int NPN = type == 0 ? NPB : NPA;             // pieces of the NEXT stage (the other type)
    const int ncc = type == 0 ? cc : (more ? cc + 1 : cc);
    static_for<0, 4>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int cur = g & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + 1 < 4) read_frags(typec, std::integral_constant<int, g + 1>{}, std::integral_constant<int, nxt>{});
      else read_frags(std::integral_constant<int, type ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, nxt>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g < 2) {
        static_for<0, (NPN + 1) / 2>([&](auto jc) {
          constexpr int j = g * ((NPN + 1) / 2) + decltype(jc)::value;
          if constexpr (j < NPN) dma_piece(std::integral_constant<int, type ^ 1>{}, std::integral_constant<int, j>{}, ncc);
        });
      }
      static_for<0, 4>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        static_for<0, NT>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int kh = type == 0 ? 1 + t / 3 : 0, kw = t % 3;
          constexpr int ph = kh != 1, pw = kw != 1, dc = kw == 0;
          acc[ph][pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][dc][s], fb[cur][t][s], acc[ph][pw], 0, 0, 0);
        });
      });
      if constexpr (g < 2) {
#pragma unroll
        for (int q = 0; q < 4 * NT; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, (NPN * 16 / 2 + 4 * NT - 1) / (4 * NT), 0);
        }
      }
      if constexpr (g == 2) { __builtin_amdgcn_sched_barrier(0); glds_wait(); __syncthreads(); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  static_for<0, NPA>([&](auto jc) { dma_piece(std::integral_constant<int, 0>{}, jc, 0); });
  glds_wait();
  __syncthreads();
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  for (int cc = 0; cc < chunks; ++cc) {
    stage(std::integral_constant<int, 0>{}, cc, true);
    stage(std::integral_constant<int, 1>{}, cc, cc + 1 < chunks);
  }

  // dx pixel of dy pixel p = (t, beta), t = p / Wo = n * Ho + alpha:  (2 t + ph) * 2 Wo + 2 beta + pw
  //                                                                 = 2 p + 2 Wo (t + ph) + pw
  const int pbase = p0 + wm * 32 + 4 * lh;
  const int tbase = (int)((unsigned)pbase / (unsigned)a.Wo), bbase = pbase - tbase * a.Wo;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int off = (r & 3) + 8 * (r >> 2);
    const int p = pbase + off, bb = bbase + off;
    const int t = tbase + (bb >= a.Wo) + (bb >= 2 * a.Wo) + (bb >= 3 * a.Wo) + (bb >= 4 * a.Wo);      // off <= 27, Wo >= 8
    if (p < P) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ph = c >> 1, pw = c & 1;
        const size_t o = (size_t)2 * p + (size_t)2 * a.Wo * (t + ph) + pw;
        a.dx[o * a.Ci + n0 + wn * 32 + ln] = acc[ph][pw][r];
      }
    }
  }
}

// ================================================================================================================
// DATA GRADIENT OF THE STEM CONVOLUTION (3 -> Co channels, 3x3, stride 2: models/actor_resnet.py:87, needed whenever
// the encoder's input image carries a gradient -- every episode step after the first)
//   3 output channels are no matrix-core shape (the library's kernel for it runs at 9 TFLOP/s: 0.40 ms for 3.6 GFLOP);
//   it is a streaming kernel: read dy (N,Ho,Wo,Co) once, 27 FMAs per (pixel, co), write dx (N,2Ho,2Wo,3).
//   One thread per dy-grid pixel (alpha, beta) produces the 2 x 2 block of dx pixels above it (see
//   k_conv3x3s2_dgrad for the index algebra) from dy at (alpha + {0,1}, beta + {0,1}).  A workgroup stages a
//   9 x 33 pixel tile of dy in LDS, 32 channels at a time, with coalesced 16-byte loads, stored channel-quad-major
//   ([co/4][pixel][4]: the threads of a wave read consecutive 16-byte slots, no bank conflicts); weights are
//   wave-uniform scalar loads and enter the FMAs as scalar operands.
constexpr int kStemTH = 8, kStemTW = 32;                 // dy-grid tile of a workgroup (256 threads)
constexpr int kStemPix = (kStemTH + 1) * (kStemTW + 1);  // with the +1 row / column the taps kh = 0 / kw = 0 read

// kPlanar: dx is (N,3,2Ho,2Wo) -- the image's own NCHW layout -- and, with acc != 0, the result is ADDED to it (the image
// gradient already holds the operator's contribution: no add kernel, no layout conversion).
template <int kCo, bool kPlanar>
__global__ __launch_bounds__(256) void k_stem_dgrad(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                    int N, int Ho, int Wo, int acc_out) {
  constexpr int kPass = 32;                              // channels staged per pass: 38 KB of LDS, 4 workgroups per CU
  constexpr int CQ = kPass / 4;
  __shared__ float4 tile[CQ][kStemPix + 1];              // (+1: plane stride 298 * 16 B, co-prime with the 16 bank groups of a b128 access)
  const int tiles_w = (Wo + kStemTW - 1) / kStemTW, tiles_h = (Ho + kStemTH - 1) / kStemTH;
  const int tb = blockIdx.x;
  const int n = tb / (tiles_h * tiles_w), th = (tb / tiles_w) % tiles_h, tw = tb % tiles_w;
  const int a0 = th * kStemTH, b0 = tw * kStemTW;
  const int ta = threadIdx.x / kStemTW, tbeta = threadIdx.x % kStemTW;
  const int al = a0 + ta, be = b0 + tbeta;
  const int p00 = ta * (kStemTW + 1) + tbeta;            // (alpha, beta); +1: beta + 1; + kStemTW + 1: alpha + 1
  float acc[2][2][3] = {};
  for (int c0 = 0; c0 < kCo; c0 += kPass) {
    if (c0) __syncthreads();
    // stage dy[n][a0 .. a0+8][b0 .. b0+32][c0 .. c0+32): 8 consecutive threads read one pixel's 128 bytes
    constexpr int kLoads = (kStemPix * CQ + 255) / 256;    // 10 per thread: all in flight before the first LDS store
    float4 stage[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int f = threadIdx.x + 256 * k;
      const int px = f / CQ, cq = f % CQ;
      const int sa = a0 + px / (kStemTW + 1), sb = b0 + px % (kStemTW + 1);
      stage[k] = (f < kStemPix * CQ && sa < Ho && sb < Wo) ? ldg4(dy + (((size_t)n * Ho + sa) * Wo + sb) * kCo + c0 + 4 * cq)
                                                            : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int f = threadIdx.x + 256 * k;
      if (f < kStemPix * CQ) tile[f % CQ][f / CQ] = stage[k];
    }
    __syncthreads();
#pragma unroll 2
    for (int cq = 0; cq < CQ; ++cq) {
      const float4 d00 = tile[cq][p00], d01 = tile[cq][p00 + 1], d10 = tile[cq][p00 + kStemTW + 1], d11 = tile[cq][p00 + kStemTW + 2];
      const float v00[4] = {d00.x, d00.y, d00.z, d00.w}, v01[4] = {d01.x, d01.y, d01.z, d01.w};
      const float v10[4] = {d10.x, d10.y, d10.z, d10.w}, v11[4] = {d11.x, d11.y, d11.z, d11.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* wc = w + (size_t)(c0 + cq * 4 + k) * 27;   // w[co][kh][kw][c]: wave-uniform -> scalar loads
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int ph = kh != 1, pw = kw != 1;
            const float d = kh == 0 ? (kw == 0 ? v11[k] : v10[k]) : (kw == 0 ? v01[k] : v00[k]);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[ph][pw][c] = fmaf(d, wc[(kh * 3 + kw) * 3 + c], acc[ph][pw][c]);
          }
      }
    }
  }
  if (al < Ho && be < Wo) {
    if constexpr (kPlanar) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          float2* o = reinterpret_cast<float2*>(dx + (((size_t)n * 3 + c) * 2 * Ho + 2 * al + ph) * 2 * Wo + 2 * be);
          float2 v = make_float2(acc[ph][0][c], acc[ph][1][c]);
          if (acc_out) { const float2 u = *o; v.x += u.x; v.y += u.y; }
          *o = v;
        }
    } else {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        float* o = dx + ((((size_t)n * 2 * Ho + 2 * al + ph) * 2 * Wo) + 2 * be) * 3;      // 6 consecutive floats (pw = 0, 1)
#pragma unroll
        for (int q = 0; q < 6; ++q) o[q] = acc_out ? o[q] + acc[ph][q / 3][q % 3] : acc[ph][q / 3][q % 3];
      }
    }
  }
}

// ---- forward of the 3 -> 32 / 64 channel stem (stride 2, padding 1), with the batch-norm statistics of its output
// 27 multiply-adds per output: a streaming kernel like its data gradient above.  A workgroup stages the 17 x 65 x 3
// input window of an 8 x 32 output tile in LDS; thread = (pixel slot, channel quad): the 16 (8) threads of a pixel
// write its 256 (128) output bytes as consecutive float4s; the 4 x 27 weights of a thread's channels stay in
// registers across the workgroup's tiles.  stats (optional): one row (2, Co) per workgroup -- sum and sum of squares of
// its outputs per channel, pixel slots added in a fixed order -- for t2o_bn_relu_nhwc_fwd_partials.
constexpr int kSfTH = 8, kSfTW = 32;
constexpr int kSfRows = 2 * kSfTH + 1, kSfRowPix = 2 * kSfTW + 1, kSfRowFloats = kSfRowPix * 3;      // 17 rows of 65 pixels = 195 floats
constexpr int kSfTilesPerWg = 8;          // consecutive tiles of a workgroup: the next window is loaded during a tile's arithmetic

template <int kCo, bool kPlanar>
__global__ __launch_bounds__(256) void k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                  float* __restrict__ stats, int N, int Ho, int Wo, int tiles_total, int Hi, int Wi) {
  constexpr int CQ = kCo / 4, PP = 256 / CQ, kPasses = (kSfTH * kSfTW) / PP;
  __shared__ __attribute__((aligned(16))) float patch[kSfRows][kSfRowFloats + 1];       // (196 floats per row: 8-byte aligned pairs)
  __shared__ float red[2][PP][kCo];
  const int tid = threadIdx.x, cq = tid % CQ, ps = tid / CQ;
  const int tiles_w = (Wo + kSfTW - 1) / kSfTW, tiles_h = (Ho + kSfTH - 1) / kSfTH;
  // (Hi, Wi: the input size, 2 Ho or 2 Ho - 1 rows: odd sizes at full-resolution inference)
  float wr[4][27];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 27; ++t) wr[j][t] = w[(size_t)(4 * cq + j) * 27 + t];
  float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  constexpr int kLoads = (kSfRows * kSfRowFloats + 255) / 256;         // 13 per thread
  float stage[kLoads];
  // the input window of a tile into registers: input rows 2 a0 - 1 .. 2 a0 + 15, columns 2 b0 - 1 .. 2 b0 + 63 (a
  // window row is 195 consecutive floats of x); all loads in flight at once
  auto issue = [&](int tile) {
    const int n = tile / (tiles_h * tiles_w), th = (tile / tiles_w) % tiles_h, tw = tile % tiles_w;
    const int ih0 = 2 * th * kSfTH - 1, iw0 = 2 * tw * kSfTW - 1;
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int f = tid + 256 * k;
      const int rr = f / kSfRowFloats, cc = f - rr * kSfRowFloats;
      if constexpr (kPlanar) {                             // x is (N,3,Hi,Wi): a window row is 3 runs of 65 consecutive floats
        const int ci = cc / kSfRowPix, ih = ih0 + rr, iw = iw0 + (cc - ci * kSfRowPix);
        stage[k] = (f < kSfRows * kSfRowFloats && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) ? x[(((size_t)n * 3 + ci) * Hi + ih) * Wi + iw] : 0.0f;
      } else {
        const int ih = ih0 + rr, iw = iw0 + cc / 3;
        stage[k] = (f < kSfRows * kSfRowFloats && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) ? x[(((size_t)n * Hi + ih) * Wi + iw0) * 3 + cc] : 0.0f;
      }
    }
  };
  const int tile0 = blockIdx.x * kSfTilesPerWg;
  if (tile0 < tiles_total) issue(tile0);
  for (int tt = 0; tt < kSfTilesPerWg; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= tiles_total) break;                        // (workgroup-uniform)
    const int n = tile / (tiles_h * tiles_w), th = (tile / tiles_w) % tiles_h, tw = tile % tiles_w;
    const int a0 = th * kSfTH, b0 = tw * kSfTW;
    if (tt) __syncthreads();                               // every wave is done with the previous window
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int f = tid + 256 * k;
      const int rr = f / kSfRowFloats, cc = f - rr * kSfRowFloats;
      const int pc = kPlanar ? (cc % kSfRowPix) * 3 + cc / kSfRowPix : cc;      // (pixel, channel) interleaved in LDS either way
      if (f < kSfRows * kSfRowFloats) patch[rr][pc] = stage[k];
    }
    __syncthreads();
    if (tt + 1 < kSfTilesPerWg && tile + 1 < tiles_total) issue(tile + 1);     // the next window travels during this tile's arithmetic
#pragma unroll 2
    for (int pass = 0; pass < kPasses; ++pass) {
      const int pix = pass * PP + ps, r = pix / kSfTW, c = pix % kSfTW;
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* row = &patch[2 * r + kh][6 * c];      // (kw, ci) = 9 consecutive floats
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float xv = row[t];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, wr[j][kh * 9 + t], acc[j]);
        }
      }
      const int oh = a0 + r, ow = b0 + c;
      if (oh < Ho && ow < Wo) {
        *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oh) * Wo + ow) * kCo + 4 * cq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][ps][4 * cq + j] = s1[j]; red[1][ps][4 * cq + j] = s2[j]; }
    __syncthreads();
    if (tid < 2 * kCo) {
      const int which = tid / kCo, ch = tid % kCo;
      float t = 0.0f;
      for (int k = 0; k < PP; ++k) t += red[which][k][ch];
      stats[((size_t)blockIdx.x * 2 + which) * kCo + ch] = t;
    }
  }
}

// ---- weight gradient of the stem: dw[co][kh][kw][ci] = sum over output pixels of dy[p][co] * x[window of p][kh][kw][ci]
// The forward kernel with the roles of weights and accumulators exchanged: the same input window in LDS, the same
// thread = (pixel slot, channel quad); a thread keeps 4 x 27 accumulators across its workgroup's tiles and reads its
// pixel's 4 dy values with one 16-byte load (the 16 threads of a pixel: 256 consecutive bytes).  The pixel slots are
// combined through LDS in a fixed order, one partial (Co, 27) block per workgroup, k_conv_wgrad_reduce adds the
// blocks: deterministic, no atomics, no output clearing (the library's kernel for this layer adds atomically).
template <int kCo, bool kPlanar>
__global__ __launch_bounds__(256) void k_stem_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                    int N, int Ho, int Wo, int tiles_total) {
  constexpr int CQ = kCo / 4, PP = 256 / CQ, kPasses = (kSfTH * kSfTW) / PP;
  __shared__ __attribute__((aligned(16))) float patch[kSfRows][kSfRowFloats + 1];
  __shared__ float red[PP][kCo + 1];
  const int tid = threadIdx.x, cq = tid % CQ, ps = tid / CQ;
  const int tiles_w = (Wo + kSfTW - 1) / kSfTW, tiles_h = (Ho + kSfTH - 1) / kSfTH;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  float acc[4][27];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[j][t] = 0.0f;
  constexpr int kLoads = (kSfRows * kSfRowFloats + 255) / 256;
  float stage[kLoads];
  auto issue = [&](int tile) {
    const int n = tile / (tiles_h * tiles_w), th = (tile / tiles_w) % tiles_h, tw = tile % tiles_w;
    const int ih0 = 2 * th * kSfTH - 1, iw0 = 2 * tw * kSfTW - 1;
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int f = tid + 256 * k;
      const int rr = f / kSfRowFloats, cc = f - rr * kSfRowFloats;
      if constexpr (kPlanar) {                             // x is (N,3,Hi,Wi): a window row is 3 runs of 65 consecutive floats
        const int ci = cc / kSfRowPix, ih = ih0 + rr, iw = iw0 + (cc - ci * kSfRowPix);
        stage[k] = (f < kSfRows * kSfRowFloats && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) ? x[(((size_t)n * 3 + ci) * Hi + ih) * Wi + iw] : 0.0f;
      } else {
        const int ih = ih0 + rr, iw = iw0 + cc / 3;
        stage[k] = (f < kSfRows * kSfRowFloats && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) ? x[(((size_t)n * Hi + ih) * Wi + iw0) * 3 + cc] : 0.0f;
      }
    }
  };
  const int tile0 = blockIdx.x * kSfTilesPerWg;
  if (tile0 < tiles_total) issue(tile0);
  for (int tt = 0; tt < kSfTilesPerWg; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= tiles_total) break;                        // (workgroup-uniform)
    const int n = tile / (tiles_h * tiles_w), th = (tile / tiles_w) % tiles_h, tw = tile % tiles_w;
    const int a0 = th * kSfTH, b0 = tw * kSfTW;
    if (tt) __syncthreads();
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int f = tid + 256 * k;
      const int rr = f / kSfRowFloats, cc = f - rr * kSfRowFloats;
      const int pc = kPlanar ? (cc % kSfRowPix) * 3 + cc / kSfRowPix : cc;      // (pixel, channel) interleaved in LDS either way
      if (f < kSfRows * kSfRowFloats) patch[rr][pc] = stage[k];
    }
    __syncthreads();
    if (tt + 1 < kSfTilesPerWg && tile + 1 < tiles_total) issue(tile + 1);
#pragma unroll 2
    for (int pass = 0; pass < kPasses; ++pass) {
      const int pix = pass * PP + ps, r = pix / kSfTW, c = pix % kSfTW;
      const int oh = a0 + r, ow = b0 + c;
      float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // pixels outside the image add nothing
      if (oh < Ho && ow < Wo) d = ldg4(dy + (((size_t)n * Ho + oh) * Wo + ow) * kCo + 4 * cq);
      const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* row = &patch[2 * r + kh][6 * c];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float xv = row[t];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][kh * 9 + t] = fmaf(dv[j], xv, acc[j][kh * 9 + t]);
        }
      }
    }
  }
  // the PP pixel slots of every (channel, tap), one tap at a time through LDS, slots added in order
  float* out = partial + (size_t)blockIdx.x * kCo * 27;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[ps][4 * cq + j] = acc[j][t];
    __syncthreads();
    if (tid < kCo) {
      float sum = 0.0f;
      for (int k = 0; k < PP; ++k) sum += red[k][tid];
      out[tid * 27 + t] = sum;
    }
  }
}

// every layer's weight transform of an optimiser step in ONE launch (the encoder has 20: 5 us each when launched one by one)
constexpr int kMaxWtJobs = 32;
struct WtJobs {
  const float* w[kMaxWtJobs];
  float* wt[kMaxWtJobs];
  int Co[kMaxWtJobs], Ci[kMaxWtJobs], taps[kMaxWtJobs], flip[kMaxWtJobs];
  unsigned first[kMaxWtJobs + 1];       // first workgroup of job j; first[n] = the grid
  int n;
};

__global__ __launch_bounds__(kConvThreads) void k_conv_flip_weight_batch(WtJobs jb) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < jb.n && blockIdx.x >= jb.first[j + 1]) ++j;                 // (uniform)
  const unsigned local = blockIdx.x - jb.first[j];
  const int Co = jb.Co[j], Ci = jb.Ci[j], taps = jb.taps[j], flip = jb.flip[j];
  const int nci = Ci / 32, nco = Co / 32;
  const int tap = (int)(local / (unsigned)(nci * nco)), rem = (int)(local % (unsigned)(nci * nco));
  const int ci0 = (rem % nci) * 32, co0 = (rem / nci) * 32;
  const float* w = jb.w[j];
  float* wt = jb.wt[j];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) tile[r][tx] = w[((size_t)(co0 + r) * taps + tap) * Ci + ci0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) wt[((size_t)(ci0 + r) * taps + (flip ? taps - 1 - tap : tap)) * Co + co0 + tx] = tile[tx][r];
}

// 16-byte accesses (LDS-DMA pieces, float4 loads) straight from caller storage: every tensor base must be 16-byte aligned
bool misaligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(c) | reinterpret_cast<size_t>(d)) & 15) != 0;
}

bool fwd_supported(int N, int H, int W, int Ci, int Co) {
  return N > 0 && H > 0 && W >= 8 && W % 8 == 0 && Ci >= 32 && Ci % 32 == 0 && Co >= 64 && Co % 64 == 0 &&
         (size_t)N * H * W + (size_t)H * W + 1024 < ((size_t)1 << 31);
}

size_t fwd_zero_bytes(int Ci) { return ((size_t)8 * Ci * 4 + 1024 + 255) / 256 * 256; }

// pixels per workgroup tile of the forward kernel for a layer (256, or 128 where 256-pixel tiles would fill less than
// one round of workgroups, one per CU, and for the two-plane x tile of stride 2)
int fwd_tile_pixels(int P, int Co, int stride) {
  const int wg256 = ((P + 255) / 256) * (Co / kFwdCo);
  const int bm = stride == 2 ? 1 : (wg256 < 256 ? 1 : 2);
  return 128 * bm;
}

struct BnbOperands { const float *x, *mean, *invstd, *w, *b; float* rows; };

int launch_fwd(const float* x, const float* w, float* y, const float* zero, int N, int H, int W, int Ci, int Co, hipStream_t st,
               int stride = 1, float* stats = nullptr, const float* addend = nullptr, const BnbOperands* bnb = nullptr) {
  FwdArgs a = {};
  a.x = x; a.w = w; a.y = y; a.zero = zero;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  const int P = N * H * W;
  a.tiles_n = Co / kFwdCo;
  const int bm = fwd_tile_pixels(P, Co, stride) / 128;
  a.tiles_p = (P + 128 * bm - 1) / (128 * bm);
  a.stats = stats;
  a.addend = addend;
  a.stamps = nullptr;
  const unsigned grid = (unsigned)(((a.tiles_p + 7) / 8) * 8 * a.tiles_n);
  if (bnb) {
    if (addend || stats || stride != 1) return T2O_EINVAL;
    a.bn_x = bnb->x; a.bn_mean = bnb->mean; a.bn_invstd = bnb->invstd; a.bn_w = bnb->w; a.bn_b = bnb->b; a.bn_rows = bnb->rows;
    if (bm == 1) k_conv3x3_fwd<1, 1, false, true><<<grid, kFwdThreads, 0, st>>>(a);
    else k_conv3x3_fwd<2, 1, false, true><<<grid, kFwdThreads, 0, st>>>(a);
  } else if (stride == 2) {
    if (addend) return T2O_EINVAL;                       // (no caller: the stride-2 data gradient is its own kernel)
    k_conv3x3_fwd<1, 2><<<grid, kFwdThreads, 0, st>>>(a);
  } else if (bm == 1) {
    if (addend) k_conv3x3_fwd<1, 1, true><<<grid, kFwdThreads, 0, st>>>(a);
    else k_conv3x3_fwd<1><<<grid, kFwdThreads, 0, st>>>(a);
  } else {
    if (addend) k_conv3x3_fwd<2, 1, true><<<grid, kFwdThreads, 0, st>>>(a);
    else k_conv3x3_fwd<2><<<grid, kFwdThreads, 0, st>>>(a);
  }
  return hipGetLastError() == hipSuccess ? T2O_OK : T2O_ELAUNCH;
}

// A caller-owned block of device memory that holds zeros and is never written (t2o_conv_set_zero_region): when one
// is registered for the current device and is large enough, it is the padding source and the workspace's own zero
// region is not cleared (one small launch less per convolution call, ~40 per encoder pass).
constexpr int kMaxDevices = 64;
const void* g_zero_ptr[kMaxDevices] = {};
size_t g_zero_bytes[kMaxDevices] = {};

const float* zero_region(void* workspace, size_t bytes, hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices && g_zero_ptr[dev] && g_zero_bytes[dev] >= bytes)
    return (const float*)g_zero_ptr[dev];
  const size_t n16 = bytes / 16;
  k_conv_zero<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>((float4*)workspace, n16);
  return (const float*)workspace;
}

struct WgradPlan { int tm, tn, tiles_m, tiles_n, splits, stages_per_split, total_stages; size_t zero_bytes; };

bool wgrad_supported(int N, int H, int W, int Ci, int Co) {
  return N > 0 && H > 0 && W >= 4 && W % 4 == 0 && Ci >= 64 && Co >= 64 && Ci % 64 == 0 && Co % 64 == 0 &&
         (size_t)N * H * W + (size_t)H * W + 64 < ((size_t)1 << 31);               // pixel indices are ints
}

WgradPlan wgrad_plan(int N, int H, int W, int Ci, int Co, int stride = 1) {
  WgradPlan p;
  constexpr int tile_n = 64, tile_m = 128;      // measured best (128 x 128 tiles: twice the split-K partial bytes per workgroup)
  p.tm = (Co % 128 == 0 && tile_m == 128) ? 128 : 64;
  p.tn = (Ci % 128 == 0 && tile_n == 128 && stride == 1) ? 128 : 64;      // (stride 2: two x planes per tile, 64 wide)
  p.tiles_m = Co / p.tm;
  p.tiles_n = Ci / p.tn;
  const int P = N * H * W;
  p.total_stages = (P + kStagePix - 1) / kStagePix;
  // zeros behind the lane offsets of one DMA piece (up to 4 pixel rows of the wider tensor)
  p.zero_bytes = ((size_t)4 * stride * (Ci > Co ? Ci : Co) * 4 + 1024 + 255) / 256 * 256;
  // ONE round of workgroups: 2 are resident per CU, 64 slots per XCD; a second, partly filled round would cost a
  // whole round's time.  (split, tile) units are dealt to the 8 XCDs in turn, 3 workgroups (kernel rows) each:
  // 21 units per XCD = 63 slots.  At least 8 stages of K per workgroup.
  int splits = 168 / (p.tiles_m * p.tiles_n);      // 168 units x 3 kernel rows = 504 workgroups
  if (splits > p.total_stages / 8) splits = p.total_stages / 8;
  if (splits < 1) splits = 1;
  p.stages_per_split = (p.total_stages + splits - 1) / splits;
  p.splits = (p.total_stages + p.stages_per_split - 1) / p.stages_per_split;
  return p;
}

// (N, H, W): the dy grid (= the x grid for stride 1; x is 2H x 2W for stride 2)
int launch_wgrad(const float* x, const float* dy, float* dw, void* workspace, int N, int H, int W, int Ci, int Co, int stride,
                 hipStream_t st, int accumulate = 0) {
  const WgradPlan p = wgrad_plan(N, H, W, Ci, Co, stride);
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = (float*)((char*)workspace + p.zero_bytes);
  a.zero = zero_region(workspace, p.zero_bytes, st);
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n;
  a.splits = p.splits; a.stages_per_split = p.stages_per_split; a.total_stages = p.total_stages;
  a.stamps = nullptr;
  const int units = p.splits * p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(((units + 7) / 8) * 24);
#define T2O_WGRAD_LAUNCH(TM_, TN_, S_) k_conv3x3_wgrad<TM_, TN_, 2, S_><<<grid, kConvThreads, 0, st>>>(a)
  if (stride == 2) { if (p.tm == 128) T2O_WGRAD_LAUNCH(128, 64, 2); else T2O_WGRAD_LAUNCH(64, 64, 2); }
  else if (p.tm == 128 && p.tn == 128) T2O_WGRAD_LAUNCH(128, 128, 1);
  else if (p.tm == 128) T2O_WGRAD_LAUNCH(128, 64, 1);
  else if (p.tn == 128) T2O_WGRAD_LAUNCH(64, 128, 1);
  else T2O_WGRAD_LAUNCH(64, 64, 1);
#undef T2O_WGRAD_LAUNCH
  const size_t n = (size_t)Co * 9 * Ci, n4 = n / 4;
  k_conv_wgrad_reduce<<<(unsigned)((n4 + 31) / 32), kConvThreads, 0, st>>>(a.partial, dw, n4, p.splits, n, accumulate);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv3x3 weight-gradient launch failed");
}

int launch_dgrad2(const float* dy, const float* wt, float* dx, const float* zeros, int N, int Ho, int Wo, int Ci, int Co, hipStream_t st) {
  Dgrad2Args a;
  a.dy = dy; a.wt = wt; a.dx = dx; a.zero = zeros;
  a.N = N; a.Ho = Ho; a.Wo = Wo; a.Ci = Ci; a.Co = Co;
  const int P = N * Ho * Wo;
  a.tiles_p = (P + 127) / 128;
  const bool narrow = (long long)a.tiles_p * (Ci / 64) <= 128;      // fewer 64-channel tiles than half the CUs: 32-channel tiles, 4 waves
  a.tiles_n = Ci / (narrow ? 32 : 64);
  const unsigned grid = (unsigned)(((a.tiles_p + 7) / 8) * 8 * a.tiles_n);
  if (narrow) k_conv3x3s2_dgrad<1><<<grid, 256, 0, st>>>(a);
  else k_conv3x3s2_dgrad<2><<<grid, kFwdThreads, 0, st>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv3x3s2_dgrad launch failed");
}

}  // namespace

namespace t2o {
// fixed-order sum of split-K partial blocks (n floats each, n % 4 == 0) into dw (t2o_conv1x1.hip uses it too)
void launch_wgrad_reduce(const float* partial, float* dw, size_t n, int splits, int accumulate, hipStream_t st) {
  const size_t n4 = n / 4;
  k_conv_wgrad_reduce<<<(unsigned)((n4 + 31) / 32), kConvThreads, 0, st>>>(partial, dw, n4, splits, n, accumulate ? 1 : 0);
}
}  // namespace t2o

extern "C" {

size_t t2o_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
  if (!wgrad_supported(N, H, W, Ci, Co)) return 0;
  const WgradPlan p = wgrad_plan(N, H, W, Ci, Co);
  return p.zero_bytes + sizeof(float) * (size_t)p.splits * Co * 9 * Ci;      // [zero region][split-K partials]
}

int t2o_conv3x3_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, void* stream) {
  return t2o_conv3x3_wgrad_acc_nhwc(x, dy, dw, workspace, workspace_bytes, N, H, W, Ci, Co, 1, 0, stream);
}

int t2o_conv3x3_wgrad_acc_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                               int N, int Ho, int Wo, int Ci, int Co, int stride, int accumulate, void* stream) {
  if (!x || !dy || !dw || misaligned16(x, dy, dw)) return set_error(T2O_EINVAL, "conv3x3_wgrad: null or not 16-byte aligned pointer");
  if (stride != 1 && stride != 2) return set_error(T2O_EUNSUPPORTED, "conv3x3_wgrad: stride 1 or 2");
  const size_t need = stride == 1 ? t2o_conv3x3_wgrad_workspace_bytes(N, Ho, Wo, Ci, Co) : t2o_conv3x3s2_wgrad_workspace_bytes(N, Ho, Wo, Ci, Co);
  if (need == 0)
    return set_error(T2O_EUNSUPPORTED, "conv3x3_wgrad: channel counts must be multiples of 64 and the output-gradient width a multiple of 4");
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "conv3x3_wgrad: workspace too small");
  return launch_wgrad(x, dy, dw, workspace, N, Ho, Wo, Ci, Co, stride, (hipStream_t)stream, accumulate ? 1 : 0);
}

size_t t2o_conv3x3s2_wgrad_workspace_bytes(int N, int Ho, int Wo, int Ci, int Co) {
  if (!wgrad_supported(N, Ho, Wo, Ci, Co) || (size_t)N * Ho * Wo * 4 + 64 >= ((size_t)1 << 31)) return 0;
  const WgradPlan p = wgrad_plan(N, Ho, Wo, Ci, Co, 2);
  return p.zero_bytes + sizeof(float) * (size_t)p.splits * Co * 9 * Ci;
}

int t2o_conv3x3s2_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                             int N, int Ho, int Wo, int Ci, int Co, void* stream) {
  if (!x || !dy || !dw || misaligned16(x, dy, dw)) return set_error(T2O_EINVAL, "conv3x3s2_wgrad: null or not 16-byte aligned pointer");
  const size_t need = t2o_conv3x3s2_wgrad_workspace_bytes(N, Ho, Wo, Ci, Co);
  if (need == 0)
    return set_error(T2O_EUNSUPPORTED, "conv3x3s2_wgrad: channel counts must be multiples of 64 and the output-gradient width a multiple of 4");
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "conv3x3s2_wgrad: workspace too small");
  return launch_wgrad(x, dy, dw, workspace, N, Ho, Wo, Ci, Co, 2, (hipStream_t)stream);
}

int t2o_conv_weight_transform(const float* w, float* wt, int Co, int Ci, int taps, int flip, void* stream) {
  if (!w || !wt) return set_error(T2O_EINVAL, "conv_weight_transform: null pointer");
  if (Co < 32 || Ci < 32 || Co % 32 != 0 || Ci % 32 != 0 || taps < 1 || taps > 9)
    return set_error(T2O_EUNSUPPORTED, "conv_weight_transform: channel counts must be multiples of 32, 1..9 taps");
  k_conv_flip_weight<<<dim3((unsigned)(Ci / 32), (unsigned)(Co / 32), (unsigned)taps), kConvThreads, 0, (hipStream_t)stream>>>(w, wt, Co, Ci, flip ? 1 : 0);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv_weight_transform launch failed");
}

int t2o_conv_weight_transform_batch(const float* const* w, float* const* wt, const int* Co, const int* Ci, const int* taps,
                                    const int* flip, int n, void* stream) {
  if (!w || !wt || !Co || !Ci || !taps || !flip || n < 1 || n > kMaxWtJobs) return set_error(T2O_EINVAL, "conv_weight_transform_batch: null pointer or more than 32 jobs");
  WtJobs jb = {};
  jb.n = n;
  unsigned total = 0;
  for (int j = 0; j < n; ++j) {
    if (!w[j] || !wt[j]) return set_error(T2O_EINVAL, "conv_weight_transform_batch: null pointer");
    if (Co[j] < 32 || Ci[j] < 32 || Co[j] % 32 != 0 || Ci[j] % 32 != 0 || taps[j] < 1 || taps[j] > 9)
      return set_error(T2O_EUNSUPPORTED, "conv_weight_transform_batch: channel counts must be multiples of 32, 1..9 taps");
    jb.w[j] = w[j]; jb.wt[j] = wt[j]; jb.Co[j] = Co[j]; jb.Ci[j] = Ci[j]; jb.taps[j] = taps[j]; jb.flip[j] = flip[j] ? 1 : 0;
    jb.first[j] = total;
    total += (unsigned)(Ci[j] / 32) * (unsigned)(Co[j] / 32) * (unsigned)taps[j];
  }
  jb.first[n] = total;
  k_conv_flip_weight_batch<<<total, kConvThreads, 0, (hipStream_t)stream>>>(jb);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv_weight_transform_batch launch failed");
}

size_t t2o_conv3x3_fwd_workspace_bytes(int N, int H, int W, int Ci, int Co) {
  return fwd_supported(N, H, W, Ci, Co) ? fwd_zero_bytes(Ci) : 0;
}

int t2o_conv3x3_fwd_nhwc(const float* x, const float* w, float* y, void* workspace, size_t workspace_bytes,
                         int N, int H, int W, int Ci, int Co, void* stream) {
  if (!x || !w || !y || misaligned16(x, w, y)) return set_error(T2O_EINVAL, "conv3x3_fwd: null or not 16-byte aligned pointer");
  if (!fwd_supported(N, H, W, Ci, Co))
    return set_error(T2O_EUNSUPPORTED, "conv3x3_fwd: Ci must be a multiple of 32, Co of 64, the image width of 8");
  if (!workspace || workspace_bytes < fwd_zero_bytes(Ci)) return set_error(T2O_EWORKSPACE, "conv3x3_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_fwd(x, w, y, zero_region(workspace, fwd_zero_bytes(Ci), st), N, H, W, Ci, Co, st);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_fwd launch failed");
}

int t2o_conv3x3_fwd_stats_rows(int N, int Ho, int Wo, int Co, int stride) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Co < kFwdCo || Co % kFwdCo != 0 || (stride != 1 && stride != 2)) return 0;
  const int P = N * Ho * Wo, tp = fwd_tile_pixels(P, Co, stride);
  return (P + tp - 1) / tp;
}

int t2o_conv3x3_fwd_stats_nhwc(const float* x, const float* w, float* y, float* stats, void* workspace, size_t workspace_bytes,
                               int N, int Ho, int Wo, int Ci, int Co, int stride, void* stream) {
  if (!x || !w || !y || !stats || misaligned16(x, w, y)) return set_error(T2O_EINVAL, "conv3x3_fwd_stats: null or not 16-byte aligned pointer");
  if (stride != 1 && stride != 2) return set_error(T2O_EUNSUPPORTED, "conv3x3_fwd_stats: stride 1 or 2");
  const size_t need = stride == 1 ? t2o_conv3x3_fwd_workspace_bytes(N, Ho, Wo, Ci, Co) : t2o_conv3x3s2_fwd_workspace_bytes(N, Ho, Wo, Ci, Co);
  if (need == 0) return set_error(T2O_EUNSUPPORTED, "conv3x3_fwd_stats: Ci must be a multiple of 32, Co of 64, the output width of 8");
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "conv3x3_fwd_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_fwd(x, w, y, zero_region(workspace, need, st), N, Ho, Wo, Ci, Co, st, stride, stats);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_fwd_stats launch failed");
}

int t2o_stem_fwd_stats_rows(int N, int Ho, int Wo) {
  if (N <= 0 || Ho <= 0 || Wo <= 0) return 0;
  const long long tiles = (long long)N * ((Ho + kSfTH - 1) / kSfTH) * ((Wo + kSfTW - 1) / kSfTW);
  return (int)((tiles + kSfTilesPerWg - 1) / kSfTilesPerWg);
}

int t2o_stem_fwd_nhwc(const float* x, const float* w, float* y, float* stats, int N, int Ho, int Wo, int Co, void* stream) {
  return t2o_stem_fwd(x, w, y, stats, N, Ho, Wo, Co, 0, stream);
}

int t2o_stem_fwd(const float* x, const float* w, float* y, float* stats, int N, int Ho, int Wo, int Co, int planar, void* stream) {
  return t2o_stem_fwd_any(x, w, y, stats, N, 2 * Ho, 2 * Wo, Co, planar, stream);
}

int t2o_stem_fwd_any(const float* x, const float* w, float* y, float* stats, int N, int Hi, int Wi, int Co, int planar, void* stream) {
  const int Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
  if (!x || !w || !y) return set_error(T2O_EINVAL, "stem_fwd: null pointer");
  if (N <= 0 || Ho <= 0 || Wo <= 0 || (Co != 32 && Co != 64) || (size_t)N * Ho * Wo * 4 * 3 >= ((size_t)1 << 40))
    return set_error(T2O_EUNSUPPORTED, "stem_fwd: 3 input channels, 32 or 64 output channels");
  const long long tiles = (long long)N * ((Ho + kSfTH - 1) / kSfTH) * ((Wo + kSfTW - 1) / kSfTW);
  if (tiles >= ((long long)1 << 30)) return set_error(T2O_EUNSUPPORTED, "stem_fwd: too many tiles");
  const unsigned grid = (unsigned)t2o_stem_fwd_stats_rows(N, Ho, Wo);
  hipStream_t st = (hipStream_t)stream;
  if (Co == 64) { if (planar) k_stem_fwd<64, true><<<grid, 256, 0, st>>>(x, w, y, stats, N, Ho, Wo, (int)tiles, Hi, Wi); else k_stem_fwd<64, false><<<grid, 256, 0, st>>>(x, w, y, stats, N, Ho, Wo, (int)tiles, Hi, Wi); }
  else { if (planar) k_stem_fwd<32, true><<<grid, 256, 0, st>>>(x, w, y, stats, N, Ho, Wo, (int)tiles, Hi, Wi); else k_stem_fwd<32, false><<<grid, 256, 0, st>>>(x, w, y, stats, N, Ho, Wo, (int)tiles, Hi, Wi); }
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "stem_fwd launch failed");
}

size_t t2o_stem_wgrad_workspace_bytes(int N, int Ho, int Wo, int Co) {
  if (Co != 32 && Co != 64) return 0;
  return sizeof(float) * (size_t)t2o_stem_fwd_stats_rows(N, Ho, Wo) * Co * 27;
}

int t2o_stem_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int Ho, int Wo,
                        int Co, void* stream) {
  return t2o_stem_wgrad(x, dy, dw, workspace, workspace_bytes, N, Ho, Wo, Co, 0, 0, stream);
}

int t2o_stem_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int Ho, int Wo,
                   int Co, int planar, int accumulate, void* stream) {
  if (!x || !dy || !dw) return set_error(T2O_EINVAL, "stem_wgrad: null pointer");
  if (N <= 0 || Ho <= 0 || Wo <= 0 || (Co != 32 && Co != 64)) return set_error(T2O_EUNSUPPORTED, "stem_wgrad: 3 input channels, 32 or 64 output channels");
  const long long tiles = (long long)N * ((Ho + kSfTH - 1) / kSfTH) * ((Wo + kSfTW - 1) / kSfTW);
  if (tiles >= ((long long)1 << 30)) return set_error(T2O_EUNSUPPORTED, "stem_wgrad: too many tiles");
  const size_t need = t2o_stem_wgrad_workspace_bytes(N, Ho, Wo, Co);
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "stem_wgrad: workspace too small");
  const int rows = t2o_stem_fwd_stats_rows(N, Ho, Wo);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  if (Co == 64) { if (planar) k_stem_wgrad<64, true><<<(unsigned)rows, 256, 0, st>>>(x, dy, partial, N, Ho, Wo, (int)tiles); else k_stem_wgrad<64, false><<<(unsigned)rows, 256, 0, st>>>(x, dy, partial, N, Ho, Wo, (int)tiles); }
  else { if (planar) k_stem_wgrad<32, true><<<(unsigned)rows, 256, 0, st>>>(x, dy, partial, N, Ho, Wo, (int)tiles); else k_stem_wgrad<32, false><<<(unsigned)rows, 256, 0, st>>>(x, dy, partial, N, Ho, Wo, (int)tiles); }
  const size_t n = (size_t)Co * 27, n4 = n / 4;
  k_conv_wgrad_reduce<<<(unsigned)((n4 + 31) / 32), kConvThreads, 0, st>>>(partial, dw, n4, rows, n, accumulate ? 1 : 0);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "stem_wgrad launch failed");
}

int t2o_gemm_nt_batched(const float* A, const float* B, float* C, int batches, int M, int N, int K, int a_rows, void* stream) {
  if (!A || !B || !C || misaligned16(A, B, C)) return set_error(T2O_EINVAL, "gemm_nt_batched: null or not 16-byte aligned pointer");
  if (batches <= 0 || M <= 0 || a_rows < M || N < 64 || N % 64 != 0 || K < 32 || K % 32 != 0 || (size_t)M * K * 4 >= ((size_t)1 << 32) ||
      (size_t)N * K * 4 >= ((size_t)1 << 32))
    return set_error(T2O_EUNSUPPORTED, "gemm_nt_batched: N must be a multiple of 64, K of 32, an operand matrix below 4 GiB");
  GemmNtArgs a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.batches = batches; a.a_rows = a_rows;
  a.tiles_m = (M + 255) / 256;
  // 256 x 128 tiles where they still give every CU a workgroup, 256 x 64 otherwise (the deep stages of a 128 x 128 image: 16 planes
  // of 1024 x 256 are 128 of the wide tiles -- half the chip idle, round 6).  The same sums in the same order either way.
  const int bn = (N % 128 == 0 && (long long)batches * a.tiles_m * (N / 128) >= 256) ? 2 : 1;
  a.tiles_n = N / (64 * bn);
  const long long rows = (long long)batches * a.tiles_m;
  const long long grid = ((rows + 7) / 8) * 8 * a.tiles_n;
  if (grid >= ((long long)1 << 31)) return set_error(T2O_EUNSUPPORTED, "gemm_nt_batched: too many tiles");
  if (bn == 2) k_gemm_nt<2><<<(unsigned)grid, kFwdThreads, 0, (hipStream_t)stream>>>(a);
  else k_gemm_nt<1><<<(unsigned)grid, kFwdThreads, 0, (hipStream_t)stream>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "gemm_nt_batched launch failed");
}

int t2o_gemm_tn_splits(int batches, int Tpad, int M, int N) {
  // enough workgroups for every CU, stages of 64 rows: 1, 2 or 4 pieces of the row range
  if (batches <= 0 || Tpad <= 0 || Tpad % 256 != 0 || M % 128 != 0 || N % 128 != 0) return 0;
  const long long tiles = (long long)batches * (M / 128) * (N / 128);
  int s = 1;
  while (s < 4 && tiles * s < 256) s *= 2;
  return s;
}

int t2o_gemm_tn_batched_ld(const float* A, const float* B, float* C, int batches, int Tpad, int ldA, int ldB, int M, int N, int splits,
                           void* stream) {
  if (!A || !B || !C || misaligned16(A, B, C)) return set_error(T2O_EINVAL, "gemm_tn_batched: null or not 16-byte aligned pointer");
  if (batches <= 0 || M < 128 || M % 128 != 0 || N < 128 || N % 128 != 0 || Tpad <= 0 || (splits != 1 && splits != 2 && splits != 4) ||
      Tpad % (64 * splits) != 0 || ldA < Tpad || ldB < Tpad || (size_t)(ldA > ldB ? ldA : ldB) * (M > N ? M : N) * 4 >= ((size_t)1 << 40))
    return set_error(T2O_EUNSUPPORTED, "gemm_tn_batched: M, N multiples of 128, rows a multiple of 64 * splits (1, 2, 4), plane strides >= rows");
  GemmTnArgs a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.Tpad = Tpad; a.batches = batches; a.splits = splits;
  a.tiles_m = M / 128; a.tiles_n = N / 128; a.ldA = ldA; a.ldB = ldB;
  const long long rows = (long long)splits * batches * a.tiles_m;
  const long long grid = ((rows + 7) / 8) * 8 * a.tiles_n;
  if (grid >= ((long long)1 << 31)) return set_error(T2O_EUNSUPPORTED, "gemm_tn_batched: too many tiles");
  k_gemm_tn<<<(unsigned)grid, kFwdThreads, 0, (hipStream_t)stream>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "gemm_tn_batched launch failed");
}

int t2o_gemm_tn_batched(const float* A, const float* B, float* C, int batches, int Tpad, int M, int N, int splits, void* stream) {
  return t2o_gemm_tn_batched_ld(A, B, C, batches, Tpad, Tpad, Tpad, M, N, splits, stream);
}

size_t t2o_conv3x3s2_fwd_workspace_bytes(int N, int Ho, int Wo, int Ci, int Co) {
  return (fwd_supported(N, Ho, Wo, Ci, Co) && (size_t)N * Ho * Wo * 4 + 1024 < ((size_t)1 << 31)) ? fwd_zero_bytes(2 * Ci) : 0;
}

int t2o_conv3x3s2_fwd_nhwc(const float* x, const float* w, float* y, void* workspace, size_t workspace_bytes,
                           int N, int Ho, int Wo, int Ci, int Co, void* stream) {
  if (!x || !w || !y || misaligned16(x, w, y)) return set_error(T2O_EINVAL, "conv3x3s2_fwd: null or not 16-byte aligned pointer");
  const size_t need = t2o_conv3x3s2_fwd_workspace_bytes(N, Ho, Wo, Ci, Co);
  if (need == 0) return set_error(T2O_EUNSUPPORTED, "conv3x3s2_fwd: Ci must be a multiple of 32, Co of 64, the output width of 8");
  if (!workspace || workspace_bytes < need) return set_error(T2O_EWORKSPACE, "conv3x3s2_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_fwd(x, w, y, zero_region(workspace, need, st), N, Ho, Wo, Ci, Co, st, 2);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3s2_fwd launch failed");
}

size_t t2o_conv3x3_dgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
  // the data gradient is a forward convolution of dy (Co channels) to dx (Ci channels)
  return fwd_supported(N, H, W, Co, Ci) ? fwd_zero_bytes(Co) + sizeof(float) * (size_t)Co * 9 * Ci : 0;
}

int t2o_conv3x3_dgrad_nhwc(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, void* stream) {
  if (!dy || !w || !dx || misaligned16(dy, w, dx)) return set_error(T2O_EINVAL, "conv3x3_dgrad: null or not 16-byte aligned pointer");
  if (!fwd_supported(N, H, W, Co, Ci) || Co % 32 != 0 || Ci % 32 != 0)
    return set_error(T2O_EUNSUPPORTED, "conv3x3_dgrad: Co must be a multiple of 32, Ci of 64, the image width of 8");
  if (!workspace || workspace_bytes < t2o_conv3x3_dgrad_workspace_bytes(N, H, W, Ci, Co))
    return set_error(T2O_EWORKSPACE, "conv3x3_dgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t zb = fwd_zero_bytes(Co);
  const float* zeros = zero_region(workspace, zb, st);
  float* wt = (float*)((char*)workspace + zb);
  k_conv_flip_weight<<<dim3((unsigned)(Ci / 32), (unsigned)(Co / 32), 9), kConvThreads, 0, st>>>(w, wt, Co, Ci, 1);
  const int rc = launch_fwd(dy, wt, dx, zeros, N, H, W, Co, Ci, st);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_dgrad launch failed");
}

int t2o_conv3x3_dgrad_bnsums_rows(int N, int H, int W, int Ci, int Co) {
  if (!fwd_supported(N, H, W, Co, Ci) || Ci % 64 != 0) return 0;
  const int P = N * H * W, px = fwd_tile_pixels(P, Ci, 1);
  return (P + px - 1) / px;
}

int t2o_conv3x3_dgrad_pre_bnsums_nhwc(const float* dy, const float* wt, float* dx, const float* bn_x, const float* save_mean,
                                      const float* save_invstd, const float* weight, const float* bias, float* rows, void* workspace,
                                      size_t workspace_bytes, int N, int H, int W, int Ci, int Co, void* stream) {
  if (!dy || !wt || !dx || !bn_x || !save_mean || !save_invstd || !weight || !bias || !rows || misaligned16(dy, wt, dx))
    return set_error(T2O_EINVAL, "conv3x3_dgrad_pre_bnsums: null or not 16-byte aligned pointer");
  if (!fwd_supported(N, H, W, Co, Ci) || Co % 32 != 0 || Ci % 64 != 0)
    return set_error(T2O_EUNSUPPORTED, "conv3x3_dgrad_pre_bnsums: Co must be a multiple of 32, Ci of 64, the image width of 8");
  if (!workspace || workspace_bytes < fwd_zero_bytes(Co)) return set_error(T2O_EWORKSPACE, "conv3x3_dgrad_pre_bnsums: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnbOperands bnb = {bn_x, save_mean, save_invstd, weight, bias, rows};
  const int rc = launch_fwd(dy, wt, dx, zero_region(workspace, fwd_zero_bytes(Co), st), N, H, W, Co, Ci, st, 1, nullptr, nullptr, &bnb);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_dgrad_pre_bnsums launch failed");
}

int t2o_conv3x3_dgrad_pre_nhwc(const float* dy, const float* wt, const float* addend, float* dx, void* workspace, size_t workspace_bytes,
                               int N, int H, int W, int Ci, int Co, void* stream) {
  if (!dy || !wt || !dx || misaligned16(dy, wt, dx, addend)) return set_error(T2O_EINVAL, "conv3x3_dgrad_pre: null or not 16-byte aligned pointer");
  if (!fwd_supported(N, H, W, Co, Ci) || Co % 32 != 0 || Ci % 32 != 0)
    return set_error(T2O_EUNSUPPORTED, "conv3x3_dgrad_pre: Co must be a multiple of 32, Ci of 64, the image width of 8");
  if (!workspace || workspace_bytes < fwd_zero_bytes(Co)) return set_error(T2O_EWORKSPACE, "conv3x3_dgrad_pre: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_fwd(dy, wt, dx, zero_region(workspace, fwd_zero_bytes(Co), st), N, H, W, Co, Ci, st, 1, nullptr, addend);
  return rc == T2O_OK ? T2O_OK : set_error(rc, "conv3x3_dgrad_pre launch failed");
}

bool stem_dgrad_supported(int N, int Ho, int Wo, int Ci, int Co) {
  return N > 0 && Ho > 0 && Wo > 0 && Ci == 3 && (Co == 64 || Co == 32) && (size_t)N * Ho * Wo * 4 < ((size_t)1 << 31);
}

size_t t2o_conv3x3s2_dgrad_workspace_bytes(int N, int Ho, int Wo, int Ci, int Co) {
  if (stem_dgrad_supported(N, Ho, Wo, Ci, Co)) return 16;            // (none needed; a non-zero size says "supported")
  return fwd_supported(N, Ho, Wo, Co, Ci) ? fwd_zero_bytes(Co) + sizeof(float) * (size_t)Co * 9 * Ci : 0;
}

int t2o_conv3x3s2_dgrad_nhwc(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                             int N, int Ho, int Wo, int Ci, int Co, void* stream) {
  if (!dy || !w || !dx || misaligned16(dy, w, dx)) return set_error(T2O_EINVAL, "conv3x3s2_dgrad: null or not 16-byte aligned pointer");
  if (stem_dgrad_supported(N, Ho, Wo, Ci, Co)) {                     // the 3-channel stem: streaming kernel, no workspace
    const unsigned grid = (unsigned)(N * ((Ho + kStemTH - 1) / kStemTH) * ((Wo + kStemTW - 1) / kStemTW));
    if (Co == 64) k_stem_dgrad<64, false><<<grid, 256, 0, (hipStream_t)stream>>>(dy, w, dx, N, Ho, Wo, 0);
    else k_stem_dgrad<32, false><<<grid, 256, 0, (hipStream_t)stream>>>(dy, w, dx, N, Ho, Wo, 0);
    return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "conv3x3s2_dgrad (stem) launch failed");
  }
  if (!fwd_supported(N, Ho, Wo, Co, Ci) || (size_t)N * Ho * Wo * 4 + 1024 >= ((size_t)1 << 31))
    return set_error(T2O_EUNSUPPORTED, "conv3x3s2_dgrad: Co must be a multiple of 32, Ci of 64 and the output-gradient width of 8 -- or Ci = 3 with Co = 32 / 64");
  if (!workspace || workspace_bytes < t2o_conv3x3s2_dgrad_workspace_bytes(N, Ho, Wo, Ci, Co))
    return set_error(T2O_EWORKSPACE, "conv3x3s2_dgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t zb = fwd_zero_bytes(Co);
  const float* zeros = zero_region(workspace, zb, st);
  float* wt = (float*)((char*)workspace + zb);
  k_conv_flip_weight<<<dim3((unsigned)(Ci / 32), (unsigned)(Co / 32), 9), kConvThreads, 0, st>>>(w, wt, Co, Ci, 0);
  return launch_dgrad2(dy, wt, dx, zeros, N, Ho, Wo, Ci, Co, st);
}

int t2o_conv3x3s2_dgrad_pre_nhwc(const float* dy, const float* wt, float* dx, void* workspace, size_t workspace_bytes,
                                 int N, int Ho, int Wo, int Ci, int Co, void* stream) {
  if (!dy || !wt || !dx || misaligned16(dy, wt, dx)) return set_error(T2O_EINVAL, "conv3x3s2_dgrad_pre: null or not 16-byte aligned pointer");
  if (!fwd_supported(N, Ho, Wo, Co, Ci) || (size_t)N * Ho * Wo * 4 + 1024 >= ((size_t)1 << 31))
    return set_error(T2O_EUNSUPPORTED, "conv3x3s2_dgrad_pre: Co must be a multiple of 32, Ci of 64 and the output-gradient width of 8");
  if (!workspace || workspace_bytes < fwd_zero_bytes(Co)) return set_error(T2O_EWORKSPACE, "conv3x3s2_dgrad_pre: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  return launch_dgrad2(dy, wt, dx, zero_region(workspace, fwd_zero_bytes(Co), st), N, Ho, Wo, Ci, Co, st);
}

int t2o_stem_dgrad(const float* dy, const float* w, float* dx, int N, int Ho, int Wo, int Co, int planar, int accumulate, void* stream) {
  if (!dy || !w || !dx) return set_error(T2O_EINVAL, "stem_dgrad: null pointer");
  if (!stem_dgrad_supported(N, Ho, Wo, 3, Co)) return set_error(T2O_EUNSUPPORTED, "stem_dgrad: 3 input channels, 32 or 64 output channels");
  const unsigned grid = (unsigned)(N * ((Ho + kStemTH - 1) / kStemTH) * ((Wo + kStemTW - 1) / kStemTW));
  hipStream_t st = (hipStream_t)stream;
  const int acc = accumulate ? 1 : 0;
  if (Co == 64) { if (planar) k_stem_dgrad<64, true><<<grid, 256, 0, st>>>(dy, w, dx, N, Ho, Wo, acc); else k_stem_dgrad<64, false><<<grid, 256, 0, st>>>(dy, w, dx, N, Ho, Wo, acc); }
  else { if (planar) k_stem_dgrad<32, true><<<grid, 256, 0, st>>>(dy, w, dx, N, Ho, Wo, acc); else k_stem_dgrad<32, false><<<grid, 256, 0, st>>>(dy, w, dx, N, Ho, Wo, acc); }
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "stem_dgrad launch failed");
}

int t2o_conv_set_zero_region(int device, const void* zeros, size_t bytes) {
  if (device < 0 || device >= kMaxDevices) return set_error(T2O_EINVAL, "conv_set_zero_region: bad device index");
  g_zero_ptr[device] = zeros;
  g_zero_bytes[device] = zeros ? bytes : 0;
  return T2O_OK;
}

}  // extern "C"
