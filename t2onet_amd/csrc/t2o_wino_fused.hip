// t2o_wino_fused.hip -- Winograd F(2x2, 3x3) with V and M kept ON CHIP, for the stride-1 3x3 convolutions of the encoder's
// 64- and 128-channel stages (models/actor_resnet.py:24-44 BasicBlock; 64 x 64 and 32 x 32 maps at bs = 64, 256 x 256 input).
//
//   The separate-pass pipeline of t2o_winograd.hip (input transform -> 16 GEMMs -> output transform) moves V and M, 4 x the
//   activation each, through HBM: it wins at 256 / 512 channels (small maps) and loses badly at 64 / 128 (T2O_WINOGRAD_MIN_C
//   = 128 measured +2 ms per step).  Here ONE launch does all of it per block of 8 x 8 tiles (16 x 16 output pixels of one
//   image) x 64 output channels:
//     * the 18 x 18 input patch of the block arrives by LDS-DMA in chunks of 8 input channels (32 bytes per pixel);
//     * the workgroup transforms a chunk to V[xi][tile][8] in LDS (B^T d B: 32 add/sub per (tile, channel));
//     * each of the 4 waves owns 32 tiles x 32 output channels of ALL 16 xi planes: 16 accumulator blocks of
//       v_mfma_f32_32x32x2_f32 = 256 registers (one wave per SIMD, the 512-register budget); the U = G g G^T operand of a
//       chunk (16 planes x 64 channels x 8, 32 KiB contiguous per plane in a chunk-major layout (Ci/8, 16, Co, 8)) arrives
//       by LDS-DMA one chunk ahead (first version: straight from L2 into registers -- the compiler waited for all of a
//       chunk's loads at its end and copied 64 registers: 140 us);
//     * the output transform A^T M A happens in registers: a lane holds the same (tile, channel) element of every plane.
//   Per chunk: 64 MFMAs per wave; DMA of chunk c + 2, transform of chunk c + 1 and the MFMAs of chunk c overlap, one barrier
//   per chunk.  16 multiplies per output and channel pair instead of 36: 8.6 GFLOP for the layer that costs the direct kernel
//   19.3.  Epilogue options of the direct kernels: addend (identity-shortcut gradient), the following batch norm's
//   statistics (per block and channel: sum, sum of squares).
//
//   LDS layouts (all conflict-free for the accesses below; 4-byte banks, 64 of them):
//     Xs[buf][slot][8 ci]: pixel (r, c) of the 18 x 18 patch in slot r * 20 + (c & 1) * 9 + (c >> 1) (even columns first: the
//       transform's lanes step over tiles = column pairs), the two 16-byte halves of a slot swapped when (slot >> 3) & 1;
//     Vs[buf][xi][tile][8 ci]: the two halves swapped when (tile >> 3) & 1.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kWfThreads = 256;
constexpr int kXRow = 20;                        // slots per patch row (18 used)
constexpr int kXPieces = 12;                     // 1 KiB DMA pieces per chunk: 18 * 20 slots * 32 B = 11,520 B
constexpr int kXBuf = kXPieces * 1024;
constexpr int kXPiecesPerWave = kXPieces / 4;    // the 4 waves' shares: pieces wave, wave + 4, ... (dma_chunk, x_piece)
static_assert(kXPiecesPerWave * 4 == kXPieces, "every wave issues the same number of x pieces per chunk (the prologue's vmcnt relies on it)");
constexpr int kVBuf = 16 * 64 * 32;              // 16 planes x 64 tiles x 32 B
constexpr int kUBuf = 16 * 64 * 32;              // 16 planes x 64 output channels x 32 B

struct WfArgs {
  const float* x;        // (N,H,W,Ci)
  const float* uc;       // (Ci/8, 16, Co, 8) chunk-major U = G g G^T
  float* y;              // (N,H,W,Co)
  const float* zero;     // >= Ci * 4 + 32 bytes of zeros
  const float* addend;   // null or (N,H,W,Co)
  float* stats;          // null or (blocks, 2, Co)
  // kEpi == 2 (data gradient in front of y = relu(bn(bn_x))): the batch norm's backward sums per block and channel of the gated
  // gradient g = dx [bn_x * sc + sh > 0] and of g * xhat go to `stats` instead (as k_conv3x3_fwd's kBnb form)
  const float* bn_x;     // (N,H,W,Co)
  const float* bn_mean;  // (Co) batch mean, inverse standard deviation, gamma, beta
  const float* bn_invstd;
  const float* bn_w;
  const float* bn_b;
  int N, H, W, Ci, Co;
  int blocks, tiles_n;   // 16 x 16 pixel blocks (N * H/16 * W/16), 64-channel tiles
  unsigned long long* stamps;   // diagnostic builds only (tools/diag/wf_clock.hip)
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// LDS-DMA, per-lane 64-bit address form: lane l's 16 bytes at addr[l] go to LDS byte lds_dst + 16 l
// (M0 is declared clobbered, not saved and restored: nothing else in these kernels uses it, and every instruction between two
// MFMAs of a one-wave-per-SIMD kernel costs its ~10 issue cycles in full)
__device__ __forceinline__ void glds16v(const void* addr, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(addr), "s"(lds_dst) : "memory", "m0");
}
// ... scalar base + 32-bit lane offset form
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void vm_wait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The MFMA as a volatile asm statement: the intrinsic has no memory dependence, and the compiler moved the MFMAs of neighbouring
// plane pairs across sched_barriers to where their operands had only just been requested (lgkmcnt stalls of ~100 cycles each).
// Volatile asm statements keep their order.  (The accumulators are read only after the loop's last barrier and an explicit
// s_nop: the hazard recognizer does not see an MFMA here.)
__device__ __forceinline__ void mfma_asm(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// ... the first product of an accumulator block: C = 0 as an inline constant instead of 256 v_accvgpr_write per workgroup
__device__ __forceinline__ void mfma_asm_first(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) {           // a - b on both halves, one instruction
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <int kFirst, int kLast, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (kFirst < kLast) {
    f(std::integral_constant<int, kFirst>{});
    static_for<kFirst + 1, kLast>(f);
  }
}

// kEpi: 0 plain, 1 addend, 2 batch-norm backward sums.  The epilogue operand of 1 / 2 (64 values per lane) is requested in ONE
// batch behind the loop: loaded value by value in the output transform it cost 39 us per launch (141 vs 102 us: nothing else
// runs on the CU to hide a round trip).
template <int kEpi>
__global__ __launch_bounds__(kWfThreads, 1) void k_wino_fused(WfArgs a) {
  constexpr bool kAdd = kEpi == 1, kBnb = kEpi == 2;
  __shared__ __attribute__((aligned(16))) char Xs[2][kXBuf];
  __shared__ __attribute__((aligned(16))) char Vs[2][kVBuf];
  __shared__ __attribute__((aligned(16))) char Us[2][kUBuf];

  // workgroup -> (pixel block, channel tile): the channel tiles of one block are neighbours inside an XCD
#ifdef T2O_WF_DIAG
  const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int b = blockIdx.x;
  const int xcd = b % 8, k8 = b / 8;
  const int blk = (k8 / a.tiles_n) * 8 + xcd, ct = k8 % a.tiles_n;
  if (blk >= a.blocks) return;
  const int bw = a.W >> 4, bh = a.H >> 4;
  const int n = blk / (bh * bw), rem = blk - n * bh * bw, by = rem / bw, bx = rem - by * bw;
  const int co0 = ct * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int th = wave >> 1, ch = wave & 1;
  const int ln = lane & 31, lh = lane >> 5;
  const int chunks = a.Ci >> 3;

  // ---- DMA: this wave's pieces wave, wave + 4, wave + 8; per lane the global address of its 16 bytes (chunk 0)
  const char* xaddr[kXPiecesPerWave];
#pragma unroll
  for (int k = 0; k < kXPiecesPerWave; ++k) {
    const int piece = wave + 4 * k;
    const int slot = (piece * 64 + lane) >> 1, phys = lane & 1;
    const int half = phys ^ ((slot >> 3) & 1);
    const int r = slot / kXRow, q = slot - r * kXRow;
    const int c = q < 9 ? 2 * q : 2 * (q - 9) + 1;
    const int h = by * 16 - 1 + r, w = bx * 16 - 1 + c;
    const bool ok = r < 18 && q < 18 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
    xaddr[k] = ok ? (const char*)(a.x + (((size_t)n * a.H + h) * a.W + w) * a.Ci) + half * 16 : (const char*)a.zero + half * 16;
  }
  const unsigned lds_x = lds_addr(&Xs[0][0]);
  auto dma_chunk = [&](int buf, int advance) {           // the chunk the addresses stand at; advances them
#pragma unroll
    for (int k = 0; k < kXPiecesPerWave; ++k) {
      glds16v(xaddr[k], lds_x + (unsigned)(buf * kXBuf + (wave + 4 * k) * 1024));
      xaddr[k] += advance;
    }
  };

  // ---- transform: this thread's (tile, channel pair): tile = 16 wave + 8 tyl + tx, pair cp (channels 2 cp, 2 cp + 1)
  const int tx = lane & 7, tyl = (lane >> 3) & 1, cp = lane >> 4;
  const int tty = 2 * wave + tyl;
  unsigned xoff[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int slot = (2 * tty + i) * kXRow + (j & 1) * 9 + tx + (j >> 1);
      xoff[i][j] = (unsigned)(slot * 32 + (((cp >> 1) ^ ((slot >> 3) & 1)) << 4) + (cp & 1) * 8);
    }
  const unsigned voff_w = (unsigned)((16 * wave + 8 * tyl + tx) * 32 + (((cp >> 1) ^ tyl) << 4) + (cp & 1) * 8);
  // B^T d B of this thread's (tile, channel pair), statement by statement (the chunk loop deals the statements out between
  // MFMAs): 16 loads d[i][j]; rows tr = B^T d: (d0 - d2, d1 + d2, d2 - d1, d1 - d3) per column; the same along the columns
  // (t2o_winograd.hip k_wino_input); 16 stores, plane xi = 4 i + j
  v2f d[4][4], tr[4][4], tv[4][4];                        // (two channels per register pair: v_pk_add_f32, half the vector instructions)
  auto t_load = [&](auto kc, int xb) {
    constexpr int k = decltype(kc)::value;
    d[k >> 2][k & 3] = *reinterpret_cast<const v2f*>(&Xs[xb][0] + xoff[k >> 2][k & 3]);
  };
  auto t_row = [&](auto kc) {                             // k = 4 j + r
    constexpr int k = decltype(kc)::value, j = k >> 2, r = k & 3;
    constexpr int p = r == 0 ? 0 : r == 1 ? 1 : r == 2 ? 2 : 1, q = r == 0 ? 2 : r == 1 ? 2 : r == 2 ? 1 : 3;
    if constexpr (r == 1) tr[r][j] = d[p][j] + d[q][j];
    else tr[r][j] = d[p][j] - d[q][j];
    asm volatile("" : "+v"(tr[r][j]));                      // (pins the step between the MFMAs it was written between)
  };
  auto t_col = [&](auto kc) {                             // k = 4 i + c
    constexpr int k = decltype(kc)::value, i = k >> 2, c = k & 3;
    constexpr int p = c == 0 ? 0 : c == 1 ? 1 : c == 2 ? 2 : 1, q = c == 0 ? 2 : c == 1 ? 2 : c == 2 ? 1 : 3;
    if constexpr (c == 1) tv[i][c] = tr[i][p] + tr[i][q];
    else tv[i][c] = tr[i][p] - tr[i][q];
    asm volatile("" : "+v"(tv[i][c]));
  };
  auto t_store = [&](auto kc, int vb) {
    constexpr int k = decltype(kc)::value;
    *reinterpret_cast<v2f*>(&Vs[vb][0] + k * 2048 + voff_w) = tv[k >> 2][k & 3];
  };
  auto transform_all = [&](int xb, int vb) {              // (prologue: chunk 0)
    static_for<0, 16>([&](auto kc) { t_load(kc, xb); });
    static_for<0, 16>([&](auto kc) { t_row(kc); });
    static_for<0, 16>([&](auto kc) { t_col(kc); });
    static_for<0, 16>([&](auto kc) { t_store(kc, vb); });
  };

  // ---- MFMA operands: A = V rows (this wave's 32 tiles), B = U rows (its 32 output channels), both from LDS
  const int atile = 32 * th + ln;
  const unsigned aoff = (unsigned)(atile * 32 + ((lh ^ ((atile >> 3) & 1)) << 4));
  const unsigned boff = (unsigned)((32 * ch + ln) * 32 + (lh << 4));
  // U chunk c = 16 planes x 64 channels x 32 bytes = 32 pieces, contiguous per plane in the chunk-major layout: piece j of
  // this wave (plane xi = 2 j + (wave >> 1), half wave & 1) is 1 KiB at ((c * 16 + xi) * Co + co0 + 32 (wave & 1)) * 32 bytes
  const unsigned lds_u = lds_addr(&Us[0][0]);
  const unsigned ulane = (unsigned)(lane * 16);
  const size_t uplane = (size_t)a.Co * 32;                // bytes per (chunk, xi) plane
  // scalar bases of this wave's 8 pieces at chunk 0 (the chunk loop adds chunk * 16 planes: two scalar instructions per piece --
  // formed in the loop, the 64-bit products cost ~16 each and the DMA issue stopped hiding behind the MFMAs it sits between)
  const char* const ubase = (const char*)a.uc + ((size_t)co0 + 32 * (wave & 1)) * 32 + (size_t)(wave >> 1) * uplane;
  unsigned upiece[8];                                     // lane offsets of the 8 pieces from the chunk's base (a chunk is < 4 GiB)
#pragma unroll
  for (int j = 0; j < 8; ++j) upiece[j] = ulane + (unsigned)(2 * j * uplane);
  const size_t uchunk = 16 * uplane;
  const unsigned lds_u_wave = lds_u + (unsigned)((wave >> 1) * 2048 + (wave & 1) * 1024);
  auto dma_u_piece = [&](auto jc, size_t uoff, unsigned ldsb) {      // uoff = chunk * uchunk, ldsb = lds_u_wave + buf * kUBuf
    constexpr int j = decltype(jc)::value;
    glds16(upiece[j], ubase + uoff, ldsb + (unsigned)(j * 4096));
  };
  auto dma_u = [&](int cc, int buf) { static_for<0, 8>([&](auto jc) { dma_u_piece(jc, (size_t)cc * uchunk, lds_u_wave + (unsigned)(buf * kUBuf)); }); };

  f32x16 acc[16];                                         // (not cleared: chunk 0's first product of every block writes C = 0)

  float4 fa[2][2], fb[2][2];                              // fragments of a plane pair, double-buffered across pairs
#ifdef T2O_WF_DIAG
  unsigned ph[9] = {};
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  auto frag_read = [&](auto pc, auto slotc, int buf) {
    constexpr int p = decltype(pc)::value, slot = decltype(slotc)::value;
    fa[slot][0] = *reinterpret_cast<const float4*>(&Vs[buf][0] + p * 2048 + aoff);
    fa[slot][1] = *reinterpret_cast<const float4*>(&Vs[buf][0] + (p + 1) * 2048 + aoff);
    fb[slot][0] = *reinterpret_cast<const float4*>(&Us[buf][0] + p * 2048 + boff);
    fb[slot][1] = *reinterpret_cast<const float4*>(&Us[buf][0] + (p + 1) * 2048 + boff);
  };
  // MFMA m (0..7) of a plane pair: component m >> 1 of the fragments, plane p + (m & 1)
  auto mfma_one = [&](auto pc, auto slotc, auto mc, auto firstc) {
    constexpr int p = decltype(pc)::value, slot = decltype(slotc)::value, m = decltype(mc)::value, w = m & 1, e = m >> 1;
    const float av = e == 0 ? fa[slot][w].x : e == 1 ? fa[slot][w].y : e == 2 ? fa[slot][w].z : fa[slot][w].w;
    const float bv = e == 0 ? fb[slot][w].x : e == 1 ? fb[slot][w].y : e == 2 ? fb[slot][w].z : fb[slot][w].w;
    if constexpr (decltype(firstc)::value && e == 0) mfma_asm_first(acc[p + w], av, bv);     // (chunk 0: the block's first product)
    else mfma_asm(acc[p + w], av, bv);
  };
  auto x_piece = [&](auto kc, int buf, int adv) {
    constexpr int k = decltype(kc)::value;
    glds16v(xaddr[k], lds_x + (unsigned)(buf * kXBuf + (wave + 4 * k) * 1024));
    xaddr[k] += adv;
  };

  // One chunk, branch-free: chunk c's MFMAs from Vs / Us[kBuf]; DMA of x chunk c + 2 into Xs[kBuf] and of U chunk c + 1 into
  // Us[kBuf ^ 1]; transform of chunk c + 1 from Xs[kBuf ^ 1] into Vs[kBuf ^ 1].  Past the end the same work is done on the last
  // chunks again, into buffers nobody reads any more.  Measured (tools/diag/wf_clock.hip): a wave's 64 MFMAs take their 4,096
  // pipe cycles whatever else happens, and everything issued OUTSIDE them is added time (first version: 877 cycles of DMA issue
  // and loads + 551 of transform per chunk) -- so every other instruction sits BETWEEN two MFMAs of this wave, pinned there:
  //   behind the barrier (before pair 7): the NEXT transform's 16 loads; pair 1's first two gaps: this one's 16 row and 16 column
  //   steps (bunched: see below); pair 3: its 16 stores;
  //   pairs 0, 5, 6: the 11 DMA pieces, one per second gap; every pair's fragments are requested one pair ahead.
  auto chunk_body = [&](int kBuf, int c, auto firstc) {  // (kBuf as a run-time value: unrolled by two with constant buffers the
                                                        // register allocation spilled 86 values)
#ifdef T2O_WF_DIAG
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    const int adv = c + 3 < chunks ? 32 : 0;              // (the addresses stop at the last chunk)
    const int cu = c + 1 < chunks ? c + 1 : chunks - 1;
    const size_t uoff = (size_t)cu * uchunk;
    const unsigned ulds = lds_u_wave + (unsigned)((kBuf ^ 1) * kUBuf);
    // The chunk's barrier sits between plane pairs 6 and 7: behind it this wave requests the NEXT chunk's first fragments and
    // the next transform's 16 patch values, and pair 7's MFMAs (operands in registers since pair 5) cover their latency.  (With
    // the barrier at the chunk's end, pair 0 took 1,330 cycles: reads, LDS latency and only then its first MFMA.)
    static_for<0, 8>([&](auto kc) {
      constexpr int k = decltype(kc)::value;              // plane pair
      if constexpr (k == 7) {
        vm_wait0();
        __syncthreads();
        frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, kBuf ^ 1);     // pair 0 of chunk c + 1
        static_for<0, 16>([&](auto jc) { t_load(jc, kBuf); });                                        // patch of chunk c + 2
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<0, 8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        mfma_one(std::integral_constant<int, 2 * k>{}, std::integral_constant<int, k & 1>{}, mc, firstc);
        __builtin_amdgcn_sched_barrier(0);
        // the 11 DMA pieces one per second MFMA gap (back to back they cost ~50 cycles each beyond the MFMA they hide behind)
#ifndef T2O_WF_XPAIR
#define T2O_WF_XPAIR 1
#endif
        // the 3 x pieces of chunk c + 2 (into Xs[kBuf], free since the barrier of chunk c - 1) in gaps 2, 4, 6 of pair T2O_WF_XPAIR:
        // the barrier in front of pair 7 waits for them -- requested in pair 4 they had ~1,600 cycles to arrive
        if constexpr (k == T2O_WF_XPAIR && (m & 1) == 0 && m >= 2) {
          static_assert(kXPiecesPerWave == 3, "the x pieces of a chunk go out in gaps 2, 4, 6 of one plane pair");
          x_piece(std::integral_constant<int, (m - 2) / 2>{}, kBuf, adv);
        }
        if constexpr (k == 0) {
          dma_u_piece(mc, uoff, ulds);
#ifndef T2O_WF_VPG
#define T2O_WF_VPG 16
#endif
        } else if constexpr (k == 1 || k == 2) {
          // the 32 transform statements T2O_WF_VPG per gap from pair 1's first gap on: a gap that holds vector instructions costs
          // ~20 cycles whatever their number plus 7-10 per instruction (measured on k_wino_wgrad, tools/diag/wgw_clock.hip).  Cycles
          // per chunk at 64 / 128 channels (tools/diag/wf_variants.sh): two per gap 5,570 / 5,415; four 5,438 / 5,290; eight 5,381 /
          // 5,218; sixteen 5,363 / 5,198; all in one 5,361 / 5,202
          constexpr int g = 8 * (k - 1) + m, lo = T2O_WF_VPG * g, hi = T2O_WF_VPG * (g + 1) < 32 ? T2O_WF_VPG * (g + 1) : 32;
          static_for<lo, hi>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            if constexpr (w < 16) t_row(std::integral_constant<int, w>{});
            else t_col(std::integral_constant<int, w - 16>{});
          });
        } else if constexpr (k == 3) {
          t_store(std::integral_constant<int, 2 * m>{}, kBuf ^ 1);
          t_store(std::integral_constant<int, 2 * m + 1>{}, kBuf ^ 1);
        }
        // the fragments of pair k + 2 go into the slot pair k is leaving, behind its last MFMA (pair 6's successor, pair 0 of the
        // next chunk, is requested behind the barrier above; pair 7's is pair 1 of the next chunk)
        if constexpr (m == 7 && k + 2 < 8) frag_read(std::integral_constant<int, 2 * (k + 2)>{}, std::integral_constant<int, k & 1>{}, kBuf);
        if constexpr (m == 7 && k == 7) frag_read(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, kBuf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      });
#ifdef T2O_WF_DIAG
      { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += (unsigned)(tn - tprev); tprev = tn; }
#endif
    });
  };

  // ---- pipeline: DMA x(c + 2), U(c + 1) | transform(c + 1) | MFMA(c), one barrier per chunk
  // x chunk 0, U chunk 0 and x chunk 1 are requested together, but the first barrier waits only for the first two (a wave's DMA
  // pieces complete in order: all but the 3 newest): chunk 1 travels under the transform of chunk 0 AND has a round trip's head
  // start.  (Requested together and all waited for at once the prologue took 5,890 cycles instead of 5,200; requested behind the
  // first barrier -- the round-4 form -- it is waited for a second full round trip.)
  dma_chunk(0, chunks > 1 ? 32 : 0);
  dma_u(0, 0);
  dma_chunk(1, chunks > 2 ? 32 : 0);                      // (a one-chunk layer: the same chunk again, never used)
  // (all but the kXPiecesPerWave newest: the count is dma_chunk's own -- ADVICE r5)
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kXPiecesPerWave) : "memory");
  __builtin_amdgcn_s_barrier();
  transform_all(0, 0);
  vm_wait0();
  __syncthreads();
  frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);        // chunk 0: pairs 0 and 1, the patch of chunk 1
  frag_read(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, 0);
  static_for<0, 16>([&](auto jc) { t_load(jc, 1); });
#ifdef T2O_WF_DIAG
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime(), r_loop = __builtin_amdgcn_s_memrealtime();
#endif
  chunk_body(0, 0, std::true_type{});
  for (int c = 1; c < chunks; ++c) chunk_body(c & 1, c, std::false_type{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // (the last MFMA's 16 passes, before any accumulator is read)
#ifdef T2O_WF_DIAG
  const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
#endif
  // Output addressing: everything but the lane's own (tile column half, channel) is uniform -- a scalar byte offset per (r, i, j)
  // plus ONE 32-bit lane offset, i.e. scalar-base loads / stores (as 64-bit per-lane offsets the 64 stores and 64 epilogue loads
  // cost 134 + 134 vector instructions of address arithmetic per workgroup)
  typedef __attribute__((address_space(1))) char* gbytes;
  typedef __attribute__((address_space(1))) const char* gcbytes;
  const unsigned lane_off = (unsigned)(lh * 32 * a.Co + ln * 4);        // tile columns 4 lh .. (8 pixels), channel ln
  const size_t rowb = (size_t)a.W * a.Co * 4, colb = (size_t)a.Co * 4;
  auto sbyte = [&](int r) {                               // pixel (2 ty, 2 tx) of register r's tile for lh = 0, channel co0 + 32 ch
    return ((((size_t)n * a.H + by * 16 + 8 * th + 2 * (r >> 2)) * a.W + bx * 16 + 2 * (r & 3)) * a.Co + co0 + 32 * ch) * 4;
  };
  // the epilogue operand: all 64 loads of the lane at once (one round trip; fetched under the last chunk's MFMAs they would
  // not fit -- the loop leaves ~50 of the 256 non-accumulator registers free, 64 more spilled 130 values)
  float pre[kEpi ? 16 : 1][4];
  if constexpr (kEpi != 0) {
    const gcbytes src = (gcbytes)(kAdd ? a.addend : a.bn_x);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const gcbytes sb = src + sbyte(r);
      pre[r][0] = *reinterpret_cast<__attribute__((address_space(1))) const float*>(sb + lane_off);
      pre[r][1] = *reinterpret_cast<__attribute__((address_space(1))) const float*>(sb + colb + lane_off);
      pre[r][2] = *reinterpret_cast<__attribute__((address_space(1))) const float*>(sb + rowb + lane_off);
      pre[r][3] = *reinterpret_cast<__attribute__((address_space(1))) const float*>(sb + rowb + colb + lane_off);
    }
    __builtin_amdgcn_sched_barrier(0);                    // (all requested before the first is waited for)
  }

  // ---- output transform in registers: y tile = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]; lane = channel co0 + 32 ch + ln,
  // register r = tile row (r & 3) + 8 (r >> 2) + 4 lh of this wave's 32 tiles
  const int co = co0 + 32 * ch + ln;
  float s1 = 0.0f, s2 = 0.0f;
  float bmean = 0.0f, binv = 0.0f, bsc = 0.0f, bsh = 0.0f;
  if constexpr (kBnb) {                                   // the gate exactly as t2o_norm.hip gated<false>: x * sc + sh > 0
    bmean = a.bn_mean[co]; binv = a.bn_invstd[co];
    bsc = a.bn_w[co] * binv; bsh = a.bn_b[co] - bmean * bsc;
  }
  // (registers r, r + 1 side by side: the transform's 24 additions per register are v_pk_add_f32 on the pair -- the same
  // arithmetic element by element; stores and statistics stay in the order r, i)
#pragma unroll
  for (int r2 = 0; r2 < 8; ++r2) {
    v2f m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = (v2f){acc[4 * i + j][2 * r2], acc[4 * i + j][2 * r2 + 1]};
    v2f q[2][4], o0p[2], o1p[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[0][j] = (m[0][j] + m[1][j]) + m[2][j];
      // (the subtractions as explicit v_pk_add_f32 with a negated source: written as vector a - b, a + (-b) or fma(b, -1, a)
      // the compiler split them into scalar v_sub_f32 again)
      q[1][j] = pk_sub(pk_sub(m[1][j], m[2][j]), m[3][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      o0p[i] = (q[i][0] + q[i][1]) + q[i][2];
      o1p[i] = pk_sub(pk_sub(q[i][1], q[i][2]), q[i][3]);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * r2 + rr;
      const gbytes yb = (gbytes)a.y + sbyte(r);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float o0 = rr == 0 ? o0p[i].x : o0p[i].y;
        float o1 = rr == 0 ? o1p[i].x : o1p[i].y;
        if constexpr (kAdd) { o0 += pre[r][2 * i]; o1 += pre[r][2 * i + 1]; }
        *reinterpret_cast<__attribute__((address_space(1))) float*>(yb + (size_t)i * rowb + lane_off) = o0;
        *reinterpret_cast<__attribute__((address_space(1))) float*>(yb + (size_t)i * rowb + colb + lane_off) = o1;
        if constexpr (kBnb) {
          const float x0 = pre[r][2 * i], x1 = pre[r][2 * i + 1];
          const float g0 = (x0 * bsc + bsh > 0.0f) ? o0 : 0.0f, g1 = (x1 * bsc + bsh > 0.0f) ? o1 : 0.0f;
          s1 += g0 + g1;
          s2 += g0 * ((x0 - bmean) * binv) + g1 * ((x1 - bmean) * binv);
        } else {
          s1 += o0 + o1;
          s2 += o0 * o0 + o1 * o1;
        }
      }
    }
  }
  if (a.stats) {                                          // (uniform) fixed order: lane, its partner lane + 32, the two tile waves
    __shared__ float red[2][2][64];
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lh == 0) { red[0][th][32 * ch + ln] = s1; red[1][th][32 * ch + ln] = s2; }
    // (not __syncthreads(): it would also wait for this wave's 64 output stores -- ~2 us per workgroup, measured 9 us per launch)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      a.stats[((size_t)blk * 2 + which) * a.Co + co0 + c] = red[which][0][c] + red[which][1][c];
    }
  }
#ifdef T2O_WF_DIAG
  if (a.stamps && lane == 0) {         // per wave: [prologue, loop, epilogue issue, pairs 0..7, wait + barrier]
    unsigned long long* q = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 16;
    q[0] = t_loop - t_start; q[1] = t_end - t_loop; q[2] = __builtin_amdgcn_s_memtime() - t_end;
    for (int i = 0; i < 9; ++i) q[3 + i] = ph[i];
    q[12] = r_entry; q[13] = __builtin_amdgcn_s_memrealtime();     // (100 MHz, one counter for the chip)
    q[14] = r_loop; q[15] = r_end;
  }
#endif
}

// U (16, Cn, Ck) -> chunk-major (Ck/8, 16, Cn, 8)
__global__ __launch_bounds__(256) void k_wino_u_chunked(const float* __restrict__ U, float* __restrict__ Uc, int Cn, int Ck) {
  const size_t total = (size_t)16 * Cn * Ck / 4;          // float4s
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  // destination float4 index i = ((cc * 16 + xi) * Cn + n) * 2 + half
  const int half = (int)(i & 1);
  const size_t j = i >> 1;
  const int nn = (int)(j % Cn);
  const size_t k = j / Cn;
  const int xi = (int)(k & 15), cc = (int)(k >> 4);
  reinterpret_cast<float4*>(Uc)[i] = *reinterpret_cast<const float4*>(U + ((size_t)xi * Cn + nn) * Ck + cc * 8 + half * 4);
}

bool wf_supported(int N, int H, int W, int Ci, int Co) {
  // Ci <= 1024: the padding source `zeros` is read Ci * 4 + 32 bytes deep (the kernel advances it 32 bytes per chunk); the
  // documented requirement is a zero block of >= 4 KiB + 32 bytes (t2onet_hip.h)
  return N > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0 && Ci >= 8 && Ci % 8 == 0 && Ci <= 1024 && Co >= 64 && Co % 64 == 0 &&
         Co <= 1024 && (size_t)N * H * W * (size_t)(Ci > Co ? Ci : Co) < ((size_t)1 << 40);
}

}  // namespace

extern "C" {

int t2o_wino_fused_supported(int N, int H, int W, int Ci, int Co) { return wf_supported(N, H, W, Ci, Co) ? 1 : 0; }

int t2o_wino_fused_stats_rows(int N, int H, int W) { return (H % 16 == 0 && W % 16 == 0 && N > 0) ? N * (H / 16) * (W / 16) : 0; }

int t2o_wino_u_chunked(const float* U, float* Uc, int Cn, int Ck, void* stream) {
  if (!U || !Uc || Cn <= 0 || Ck <= 0 || Ck % 8 != 0) return set_error(T2O_EINVAL, "wino_u_chunked: null pointer or Ck not a multiple of 8");
  const size_t total = (size_t)16 * Cn * Ck / 4;
  k_wino_u_chunked<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(U, Uc, Cn, Ck);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_u_chunked launch failed");
}

int t2o_wino_fused_conv_nhwc(const float* x, const float* uc, const float* addend, float* y, float* stats, const float* zeros,
                             int N, int H, int W, int Ci, int Co, void* stream) {
  if (!x || !uc || !y || !zeros) return set_error(T2O_EINVAL, "wino_fused_conv: null pointer");
  if (((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(uc) | reinterpret_cast<size_t>(zeros)) & 15) != 0)
    return set_error(T2O_EINVAL, "wino_fused_conv: x, uc and zeros must be 16-byte aligned");
  if (!wf_supported(N, H, W, Ci, Co)) return set_error(T2O_EUNSUPPORTED, "wino_fused_conv: H, W multiples of 16, Ci of 8, Co of 64");
  WfArgs a = {};
  a.x = x; a.uc = uc; a.y = y; a.zero = zeros; a.addend = addend; a.stats = stats;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.blocks = N * (H / 16) * (W / 16);
  a.tiles_n = Co / 64;
  const unsigned grid = (unsigned)(((a.blocks + 7) / 8) * 8 * a.tiles_n);
  hipStream_t st = (hipStream_t)stream;
  if (addend) k_wino_fused<1><<<grid, kWfThreads, 0, st>>>(a);
  else k_wino_fused<0><<<grid, kWfThreads, 0, st>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_fused_conv launch failed");
}

int t2o_wino_fused_conv_bnsums_nhwc(const float* x, const float* uc, float* y, const float* bn_x, const float* save_mean,
                                    const float* save_invstd, const float* weight, const float* bias, float* rows, const float* zeros,
                                    int N, int H, int W, int Ci, int Co, void* stream) {
  if (!x || !uc || !y || !zeros || !bn_x || !save_mean || !save_invstd || !weight || !bias || !rows)
    return set_error(T2O_EINVAL, "wino_fused_conv_bnsums: null pointer");
  if (((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(uc) | reinterpret_cast<size_t>(zeros)) & 15) != 0)
    return set_error(T2O_EINVAL, "wino_fused_conv_bnsums: x, uc and zeros must be 16-byte aligned");
  if (!wf_supported(N, H, W, Ci, Co)) return set_error(T2O_EUNSUPPORTED, "wino_fused_conv_bnsums: H, W multiples of 16, Ci of 8, Co of 64");
  WfArgs a = {};
  a.x = x; a.uc = uc; a.y = y; a.zero = zeros; a.stats = rows;
  a.bn_x = bn_x; a.bn_mean = save_mean; a.bn_invstd = save_invstd; a.bn_w = weight; a.bn_b = bias;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.blocks = N * (H / 16) * (W / 16);
  a.tiles_n = Co / 64;
  const unsigned grid = (unsigned)(((a.blocks + 7) / 8) * 8 * a.tiles_n);
  k_wino_fused<2><<<grid, kWfThreads, 0, (hipStream_t)stream>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "wino_fused_conv_bnsums launch failed");
}

}  // extern "C"
