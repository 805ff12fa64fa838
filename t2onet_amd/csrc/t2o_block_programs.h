// Block programs: what ONE thread of ONE workgroup does in each kernel phase.
//
// The kernels in t2o_kernels.hip are thin wrappers: they map blockIdx/threadIdx
// to (sample, chunk/tile, tid), call these `__host__ __device__` phase functions
// with `__syncthreads()` between phases, and do the wave-shuffle reductions.
// tests/host_emul runs the same phase functions thread by thread on the CPU to
// check indexing, halos and bounds without a GPU (test harness only).
//
// Data layout: images are fp32 NCHW contiguous, RGB in
// [0,1]; a sample is 3 planes of H*W floats.  Parameters are rows of
// `param_stride` floats per sample.  Masks are (B,1,H,W) or (B,3,H,W).
#pragma once
#ifndef __HIPCC_RTC__      // (hipRTC: no system headers)
#include <stddef.h>
#include <stdint.h>
#endif

#include "t2o_pixel_math.h"

namespace t2o {

constexpr int kThreads = 256;      // 4 waves of 64
constexpr int kRedSlots = 24;      // widest per-sample parameter-gradient reduction (color curve)

// Stencil tile: 16 rows x 64 columns per workgroup, thread (ty,tx) owns 4 pixels of one row.
constexpr int kTileW = 64;
constexpr int kTileH = 16;
constexpr int kStripRows = 4;       // LDS-free stencil backward: rows per thread (x rows loaded / row = 2, dz rows computed / row = 1.5)
constexpr int kStripWideCols = 248;  // backward, W > 256: columns a wave outputs (62 quads; lanes 0 / 63 overlap the neighbours)
T2O_HD int strip_bwd_segments(int W) { return W <= 256 ? 1 : (W + kStripWideCols - 1) / kStripWideCols; }
constexpr int kFwdStripRows = 2;    // forward: 2 rows per thread measured best (1: 33.0, 2: 33.0, 4: 35.5 us; tile kernel 34.3)
constexpr int kRowStride = 72;     // floats per LDS row: [..halo][64 interior at +4][halo..], 16-B aligned rows
constexpr int kIntOff = 4;         // LDS column index of interior column 0

struct OpArgs {
  // inputs
  const float* img;        // (B,3,H,W)
  const float* param;      // (B,param_stride)
  const float* mask;       // null, (B,1,H,W) or (B,3,H,W)
  const int* op_id;        // (B) device ints, used when op == OP_DYNAMIC
  const float* gout;       // backward: gradient w.r.t. the operator output (null when l1 fused)
  const float* target;     // fused L1: (B,3,H,W) target image, else null
  const float* gloss;      // fused L1 backward: device scalar d(total)/d(loss)
  // outputs
  float* out;              // forward: (B,3,H,W)
  float* gimg;             // backward: (B,3,H,W) or null
  float* partials;         // backward: (B, nblk_max, kRedSlots) raw parameter-gradient sums
  float* loss_partials;    // fused L1 forward: (B, nblk_max)
  // geometry
  int op;                  // operator index or OP_DYNAMIC
  int param_stride;
  int mask_ch;             // 0, 1 or 3
  int B, H, W;
  int iters;               // pointwise: pixel groups per thread
  int nblk_max;            // stride (in blocks) of partials / loss_partials per sample
  float inv_n;             // fused L1: 1 / (B*3*H*W)
};

T2O_HD size_t plane_off(const OpArgs& a, int b, int c) { return ((size_t)b * 3 + c) * (size_t)a.H * a.W; }

T2O_HD float mask_at(const OpArgs& a, int b, int c, size_t px) {
  const int mc = a.mask_ch == 3 ? c : 0;
  return a.mask[((size_t)b * a.mask_ch + mc) * (size_t)a.H * a.W + px];
}

template <int V>
T2O_HD void load_vec(const float* p, float (&r)[V]) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    return;
  }
#endif
  T2O_UNROLL
  for (int i = 0; i < V; ++i) r[i] = p[i];
}

template <int V>
T2O_HD void store_vec(float* p, const float (&r)[V]) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    return;
  }
#endif
  T2O_UNROLL
  for (int i = 0; i < V; ++i) p[i] = r[i];
}

T2O_HD float sign_of(float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); }

// ===================================================================== curve lookup table
// Curves use a per-sample table (in LDS on the device): knots k[c][8], running sums
// P[c][j] = sum_{i<j} k_i/8 accumulated in the reference's order, sum[c], scale[c] and 1/sum[c].  For a
// pixel value x in segment i* = floor(8x):  total = P[i*] + clamp(x - i*/8, 0, 1/8) * k[i*]  -- bit-identical
// to the reference's 8-term loop (the terms before i* are exactly k_i/8, the ones after are exactly 0), at
// a quarter of the arithmetic.  Used by the fused chain kernels and by the single-operator backward.
constexpr int kTabStride = 64;           // floats per operator table
constexpr int kTabK = 0, kTabP = 24, kTabSum = 51, kTabScale = 54, kTabRsum = 57;

// p = the operator's parameter row (8 knots, or 3 x 8 for the color curve); one thread builds the table
T2O_HD void curve_table_build(const float* p, bool color, float* t) {
  for (int c = 0; c < 3; ++c) {
    const float* row = color ? p + c * kCurveSteps : p;
    float s = 0.0f, run = 0.0f;
    for (int i = 0; i < kCurveSteps; ++i) {
      t[kTabK + c * kCurveSteps + i] = row[i];
      t[kTabP + c * (kCurveSteps + 1) + i] = run;
      run = run + (1.0f / kCurveSteps) * row[i];
      s = s + row[i];
    }
    t[kTabP + c * (kCurveSteps + 1) + kCurveSteps] = run;
    s = s + 1e-10f;
    t[kTabSum + c] = s;
    t[kTabRsum + c] = 1.0f / s;
    t[kTabScale + c] = t[kTabRsum + c] * (float)kCurveSteps;
  }
}

// segment index and clamped offset inside it.  unit: the caller guarantees 0 <= x <= 1 (the clamped output of the
// previous operator of a fused chain): x - i/8 then lies in [0, 1/8] by construction and the two clamps are dropped
// -- same values, fewer vector instructions (the chain kernels are bound by their vector-instruction count).
T2O_HD void curve_locate(float x, int& i, float& frac, bool unit = false) {
  const float x8 = unit ? fminf(x * (float)kCurveSteps, (float)kCurveSteps - 0.5f)
                        : fminf(fmaxf(x * (float)kCurveSteps, 0.0f), (float)kCurveSteps - 0.5f);
  i = (int)x8;
  const float d = x - (float)i / kCurveSteps;
  frac = unit ? d : fminf(fmaxf(d, 0.0f), 1.0f / kCurveSteps);
}

// pre-clamp output of a curve operator for channel c
T2O_HD float curve_lut_fwd(const float* t, bool color, int c, float x, bool unit = false) {
  const int cc = color ? c : 0;
  int i; float frac;
  curve_locate(x, i, frac, unit);
  const float total = t[kTabP + cc * (kCurveSteps + 1) + i] + frac * t[kTabK + cc * kCurveSteps + i];
  return color ? total * t[kTabScale + cc]
               : div_by(total * (float)kCurveSteps, t[kTabSum + cc], t[kTabRsum + cc]);
}

// backward of one channel of a curve operator through the table: g = gradient w.r.t. the pre-clamp output
// (already zero where the clamp was active); adds the raw sums red_row[j] += g * t_j (red_row = this
// channel's 8 slots), returns the gradient w.r.t. x
T2O_HD float curve_lut_bwd_1(const float* t, bool color, int c, float x, float g, float* red_row, bool first = false,
                             bool unit = false) {
  const int cc = color ? c : 0;
  const float* kk = t + kTabK + cc * kCurveSteps;
  int i; float frac;
  curve_locate(x, i, frac, unit);
  const float d = x - (float)i / kCurveSteps;
  float slope = (unit || (d >= 0.0f && d <= 1.0f / kCurveSteps)) ? kk[i] : 0.0f;   // unit: d is inside by construction
  if (d == 0.0f && i > 0) slope += kk[i - 1];                       // on a knot both neighbours pass (inclusive clamp)
  curve_bins_accumulate(x, g, red_row, first);
  return g * t[kTabScale + cc] * slope;
}

// ===================================================================== pointwise operators
// Per-sample base pointers + 32-bit offsets inside the sample: keeps the address arithmetic in
// a handful of scalar registers (a sample is < 2^31 floats).
struct SampleView {
  const float* x;      // this sample's 3 input planes
  const float* g;      // gout, or the L1 target when fused
  const float* m;      // mask planes (or null)
  float* o;            // out / gimg planes (or null)
  const float* t;      // forward fused-L1 target planes (or null)
  unsigned hw;         // plane stride
  unsigned mstride;    // hw for a 3-channel mask, 0 for a 1-channel mask
};

// Forward: one thread, `iters` groups of V consecutive pixels of sample b.
// Returns this thread's partial sum of |out - target| (0 when not L1).
template <int V, bool MASKED, bool L1>
T2O_HD float pointwise_fwd_thread(const OpArgs& a, int op, int b, int blk, int tid) {
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const unsigned groups = hw / V;
  const size_t sb = (size_t)b * 3 * hw;
  SampleView s;
  s.x = a.img + sb;
  s.o = a.out + sb;
  s.t = L1 ? a.target + sb : nullptr;
  s.m = MASKED ? a.mask + (size_t)b * a.mask_ch * hw : nullptr;
  s.mstride = (MASKED && a.mask_ch == 3) ? hw : 0u;
  const float* prow = a.param ? a.param + (size_t)b * a.param_stride : nullptr;
  Curve cv;
  if (op == OP_COLOR || op == OP_TONE) curve_load(cv, prow, op == OP_COLOR);
  float p0[1] = {(op >= 0 && prow) ? prow[0] : 0.0f};
  float l1 = 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned g = ((unsigned)blk * a.iters + it) * kThreads + tid;
    if (g >= groups) break;
    const unsigned px = g * V;
    float x[3][V], o[3][V], mk[3][V];
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) {
      load_vec<V>(s.x + c * hw + px, x[c]);
      if (MASKED && (c == 0 || s.mstride)) load_vec<V>(s.m + c * s.mstride + px, mk[c]);
    }
    T2O_UNROLL
    for (int i = 0; i < V; ++i) {
      Rgb xi = {{x[0][i], x[1][i], x[2][i]}};
      if (op == OP_IDENTITY) {                       // executor.py:44-46: returned as is, no clamp
        o[0][i] = xi.c[0]; o[1][i] = xi.c[1]; o[2][i] = xi.c[2];
      } else {
        const Rgb r = pointwise_fwd(op, xi, p0, cv);
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) {
          float z = r.c[c];
          if (MASKED) z = blend(z, xi.c[c], s.mstride ? mk[c][i] : mk[0][i]);
          o[c][i] = clamp01(z);
        }
      }
    }
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) store_vec<V>(s.o + c * hw + px, o[c]);
    if (L1) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        float t[V];
        load_vec<V>(s.t + c * hw + px, t);
        T2O_UNROLL
        for (int i = 0; i < V; ++i) l1 += fabsf(o[c][i] - t[i]);
      }
    }
  }
  return l1;
}

// Backward: recompute the forward (for the clamp test), apply the closed-form
// derivative, write gimg, accumulate raw parameter-gradient sums into red[].
// `tab`: this sample's curve lookup table (curve_table_build), used when op is a curve operator.
template <int V, bool MASKED, bool L1>
T2O_HD void pointwise_bwd_thread(const OpArgs& a, int op, int b, int blk, int tid, float (&red)[kRedSlots], const float* tab) {
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const unsigned groups = hw / V;
  const size_t sb = (size_t)b * 3 * hw;
  SampleView s;
  s.x = a.img + sb;
  s.g = (L1 ? a.target : a.gout) + sb;
  s.o = a.gimg ? a.gimg + sb : nullptr;
  s.m = MASKED ? a.mask + (size_t)b * a.mask_ch * hw : nullptr;
  s.mstride = (MASKED && a.mask_ch == 3) ? hw : 0u;
  const float* prow = a.param ? a.param + (size_t)b * a.param_stride : nullptr;
  Curve cv;                                    // (curve operators go through `tab`: cv stays unused there)
  float p0[1] = {(op >= 0 && prow) ? prow[0] : 0.0f};
  const float gs = L1 ? a.gloss[0] * a.inv_n : 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned g = ((unsigned)blk * a.iters + it) * kThreads + tid;
    if (g >= groups) break;
    const unsigned px = g * V;
    float x[3][V], gg[3][V], gx[3][V], mk[3][V];
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) {
      load_vec<V>(s.x + c * hw + px, x[c]);
      load_vec<V>(s.g + c * hw + px, gg[c]);
      if (MASKED && (c == 0 || s.mstride)) load_vec<V>(s.m + c * s.mstride + px, mk[c]);
    }
    if (op == OP_TONE || op == OP_COLOR) {
      // channels are independent: run the thread's 3V pixel-channels strictly one after another
      // (dependency chain), so only one channel's 8 segment terms are live at a time
      const bool color = op == OP_COLOR;
      float dep = 0.0f;
      T2O_UNROLL
      for (int i = 0; i < V; ++i) {
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) {
          const float xv = T2O_CHAIN(x[c][i], dep);
          const float r = curve_lut_fwd(tab, color, c, xv);
          const float m = MASKED ? (s.mstride ? mk[c][i] : mk[0][i]) : 1.0f;
          const float z = MASKED ? blend(r, xv, m) : r;
          const float gz = L1 ? sign_of(clamp01(z) - gg[c][i]) * gs : gg[c][i];
          const float dz = (z >= 0.0f && z <= 1.0f) ? gz : 0.0f;
          const int kc = color ? c : 0;
          const float gi = curve_lut_bwd_1(tab, color, c, xv, MASKED ? dz * m : dz, red + kc * kCurveSteps);
          gx[c][i] = MASKED ? gi + dz * (1.0f - m) : gi;
          dep = gx[c][i];
          T2O_UNROLL
          for (int j = 0; j < kCurveSteps; ++j) T2O_KEEP(red[kc * kCurveSteps + j]);
        }
      }
    } else {
    T2O_UNROLL
    for (int i = 0; i < V; ++i) {
      Rgb xi = {{x[0][i], x[1][i], x[2][i]}};
      if (op == OP_IDENTITY) {
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) gx[c][i] = L1 ? sign_of(xi.c[c] - gg[c][i]) * gs : gg[c][i];
        continue;
      }
      Rgb go, gpass;
      if (!MASKED && !L1 && !clamp_can_act(op, xi)) {      // output provably inside [0,1]: dz = gout
        go.c[0] = gg[0][i]; go.c[1] = gg[1][i]; go.c[2] = gg[2][i];
        const Rgb gi = pointwise_bwd(op, xi, p0, cv, go, red);
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) gx[c][i] = gi.c[c];
        continue;
      }
      const Rgb r = pointwise_fwd(op, xi, p0, cv);
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        const float m = MASKED ? (s.mstride ? mk[c][i] : mk[0][i]) : 1.0f;
        const float z = MASKED ? blend(r.c[c], xi.c[c], m) : r.c[c];
        const float gz = L1 ? sign_of(clamp01(z) - gg[c][i]) * gs : gg[c][i];
        const float dz = (z >= 0.0f && z <= 1.0f) ? gz : 0.0f;     // clamp(0,1) backward, inclusive
        go.c[c] = MASKED ? dz * m : dz;
        gpass.c[c] = MASKED ? dz * (1.0f - m) : 0.0f;
      }
      const Rgb gi = pointwise_bwd(op, xi, p0, cv, go, red);
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) gx[c][i] = gi.c[c] + gpass.c[c];
    }
    }
    if (s.o) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) store_vec<V>(s.o + c * hw + px, gx[c]);
    }
  }
}

// ===================================================================== sharpness (3x3 stencil)
// LDS tile of one channel plane: rows of kRowStride floats; interior column j at kIntOff + j.
// `halo` = 1 (forward, gradient, mask tiles) or 2 (backward image tile).
T2O_HD int tile_rows(int halo) { return kTileH + 2 * halo; }
T2O_HD int tile_floats(int halo) { return tile_rows(halo) * kRowStride; }

// Cooperative load of a (kTileH+2*HALO) x (kTileW+2*HALO) window of `planes` planes starting at
// plane pointer src (plane stride hw) into lds; zero outside the image.  HALO is a template
// parameter so every divisor below is a compile-time constant (multiply-shift, not a divide
// sequence), and offsets inside a sample are 32-bit.
// V == 4 (W % 4 == 0): interior as aligned float4 + scalar halo columns; V == 1: scalars.
template <int V, int HALO>
T2O_HD void tile_load(const float* src, unsigned hw, int planes, int H, int W, int y0, int x0, float* lds, int tid) {
  constexpr int rows = kTileH + 2 * HALO;
  constexpr int plane_floats = rows * kRowStride;
  if (V == 4) {
    constexpr int per_plane = rows * (kTileW / 4);
    for (int i = tid; i < planes * per_plane; i += kThreads) {
      const int c = i / per_plane, rem = i % per_plane;
      const int r = rem / (kTileW / 4), q = rem % (kTileW / 4);
      const int gy = y0 - HALO + r, gx = x0 + 4 * q;
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (gy >= 0 && gy < H && gx < W) load_vec<4>(src + ((unsigned)c * hw + (unsigned)gy * (unsigned)W + (unsigned)gx), v);
      store_vec<4>(lds + c * plane_floats + r * kRowStride + kIntOff + 4 * q, v);
    }
    constexpr int hper = rows * 2 * HALO;
    for (int i = tid; i < planes * hper; i += kThreads) {
      const int c = i / hper, rem = i % hper;
      const int r = rem / (2 * HALO), k = rem % (2 * HALO);
      const int j = k < HALO ? k - HALO : kTileW + (k - HALO);     // column relative to the tile
      const int gy = y0 - HALO + r, gx = x0 + j;
      float v = 0.0f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = src[(unsigned)c * hw + (unsigned)gy * (unsigned)W + (unsigned)gx];
      lds[c * plane_floats + r * kRowStride + kIntOff + j] = v;
    }
  } else {
    constexpr int cols = kTileW + 2 * HALO;
    constexpr int per_plane = rows * cols;
    for (int i = tid; i < planes * per_plane; i += kThreads) {
      const int c = i / per_plane, rem = i % per_plane;
      const int r = rem / cols, j = rem % cols - HALO;
      const int gy = y0 - HALO + r, gx = x0 + j;
      float v = 0.0f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = src[(unsigned)c * hw + (unsigned)gy * (unsigned)W + (unsigned)gx];
      lds[c * plane_floats + r * kRowStride + kIntOff + j] = v;
    }
  }
}

// element (row r, column j) of plane c of a tile with the given halo; r, j relative to the tile origin
T2O_HD float tile_at(const float* lds, int halo, int c, int r, int j) {
  return lds[c * tile_floats(halo) + (r + halo) * kRowStride + kIntOff + j];
}

T2O_HD void tile_origin(const OpArgs& a, int tile, int& y0, int& x0) {
  const int tiles_x = (a.W + kTileW - 1) / kTileW;
  y0 = (tile / tiles_x) * kTileH;
  x0 = (tile % tiles_x) * kTileW;
}
T2O_HD int sharp_num_tiles(int H, int W) {
  return ((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH);
}

// ---- forward ----   LDS: [3 planes, halo 1]
T2O_HD int sharp_fwd_lds_floats() { return 3 * tile_floats(1); }

template <int V>
T2O_HD void sharp_fwd_phase_load(const OpArgs& a, int b, int tile, int tid, float* lds) {
  int y0, x0;
  tile_origin(a, tile, y0, x0);
  tile_load<V, 1>(a.img + plane_off(a, b, 0), (unsigned)a.H * (unsigned)a.W, 3, a.H, a.W, y0, x0, lds, tid);
}

// 4 horizontally adjacent window elements of plane c starting at tile column j (j % 4 == 0):
// one 16-byte LDS read (rows are 16-byte aligned, the interior starts at a 16-byte boundary)
T2O_HD void tile_quad(const float* lds, int halo, int c, int r, int j, float (&q)[4]) {
  load_vec<4>(lds + c * tile_floats(halo) + (r + halo) * kRowStride + kIntOff + j, q);
}

// Left / right neighbours of an aligned quad.  The lanes of a 16-lane DPP row hold the 16 quads of
// one tile row, so the neighbours are the adjacent lanes' end elements (one DPP move each); only
// the first / last quad of the row reads the halo column from LDS.  (Reading both from LDS on every
// lane is a stride-4 access: all 64 lanes on 8 banks.)  Host emulation reads LDS.
struct Sides { float L, R; };
T2O_HD Sides quad_sides(const float* lds, int halo, int c, int r, int j0, const float (&ce)[4]) {
  Sides s;
#if defined(__HIP_DEVICE_COMPILE__)
  const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ce[3]), 0x111, 0xF, 0xF, true));  // row_shr:1
  const float rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ce[0]), 0x101, 0xF, 0xF, true)); // row_shl:1
  // one LDS read per lane: the left halo column for the first quad, else the right one (a broadcast
  // for the 15 lanes that do not need it)
  const float e = tile_at(lds, halo, c, r, j0 == 0 ? -1 : kTileW);
  s.L = j0 == 0 ? e : l;
  s.R = j0 == kTileW - 4 ? e : rr;
#else
  s.L = tile_at(lds, halo, c, r, j0 - 1);
  s.R = tile_at(lds, halo, c, r, j0 + 4);
#endif
  return s;
}

template <int V>
T2O_HD float sharp_fwd_phase_compute(const OpArgs& a, int b, int tile, int tid, const float* lds) {
  int y0, x0;
  tile_origin(a, tile, y0, x0);
  const int ty = tid / (kTileW / 4), tx = tid % (kTileW / 4);
  const int gy = y0 + ty, gx0 = x0 + 4 * tx;
  const bool live = gy < a.H && gx0 < a.W;     // every lane runs the DPP exchanges; only stores are predicated
  const float p = a.param[(size_t)b * a.param_stride];
  float l1 = 0.0f;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    float up[4], ce[4], dn[4], o[4];
    tile_quad(lds, 1, c, ty - 1, 4 * tx, up);
    tile_quad(lds, 1, c, ty, 4 * tx, ce);
    tile_quad(lds, 1, c, ty + 1, 4 * tx, dn);
    const Sides sd = quad_sides(lds, 1, c, ty, 4 * tx, ce);
    const float L = sd.L, R = sd.R;
    T2O_UNROLL
    for (int i = 0; i < 4; ++i) {
      const float left = i == 0 ? L : ce[i > 0 ? i - 1 : 0], right = i == 3 ? R : ce[i < 3 ? i + 1 : 3];
      const float d = sharp_delta(ce[i], up[i], left, right, dn[i]);
      float z = ce[i] + p * d;
      if (a.mask_ch && live && gx0 + i < a.W) z = blend(z, ce[i], mask_at(a, b, c, (size_t)gy * a.W + gx0 + i));
      o[i] = clamp01(z);
    }
    if (!live) continue;
    float* dst = a.out + plane_off(a, b, c) + (size_t)gy * a.W + gx0;
    const float* tg = a.target ? a.target + plane_off(a, b, c) + (size_t)gy * a.W + gx0 : nullptr;
    if (V == 4) {
      store_vec<4>(dst, o);
      if (tg) {
        float t[4];
        load_vec<4>(tg, t);
        T2O_UNROLL
        for (int i = 0; i < 4; ++i) l1 += fabsf(o[i] - t[i]);
      }
    } else {
      for (int i = 0; i < 4 && gx0 + i < a.W; ++i) {
        dst[i] = o[i];
        if (tg) l1 += fabsf(o[i] - tg[i]);
      }
    }
  }
  return l1;
}

// ---- backward ----  LDS: [X: 3 planes halo 2][G: 3 planes halo 1][M: mask_ch planes halo 1]
T2O_HD int sharp_bwd_x_off() { return 0; }
T2O_HD int sharp_bwd_g_off() { return 3 * tile_floats(2); }
T2O_HD int sharp_bwd_m_off() { return 3 * tile_floats(2) + 3 * tile_floats(1); }
T2O_HD int sharp_bwd_lds_floats(int mask_ch) { return sharp_bwd_m_off() + mask_ch * tile_floats(1); }

template <int V>
T2O_HD void sharp_bwd_phase_load(const OpArgs& a, int b, int tile, int tid, float* lds) {
  int y0, x0;
  tile_origin(a, tile, y0, x0);
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  tile_load<V, 2>(a.img + plane_off(a, b, 0), hw, 3, a.H, a.W, y0, x0, lds + sharp_bwd_x_off(), tid);
  tile_load<V, 1>((a.target ? a.target : a.gout) + plane_off(a, b, 0), hw, 3, a.H, a.W, y0, x0, lds + sharp_bwd_g_off(), tid);
  if (a.mask_ch)
    tile_load<V, 1>(a.mask + (size_t)b * a.mask_ch * hw, hw, a.mask_ch, a.H, a.W, y0, x0, lds + sharp_bwd_m_off(), tid);
}

// Phase 2: at every window position (interior + 1-px halo) replace G by
// dz = [0 <= z <= 1] * gradient of the clamped output, zero outside the image; interior positions
// also add do * Laplacian(x) (do = dz * m) to red0, the raw sum of d loss / d p.
// One quad (4 columns) of one row of all 3 planes:
T2O_HD void sharp_dz_quad(const OpArgs& a, float p, float gs, int y0, int x0, int c, int r, int j0, int n, float* lds,
                          float& red0) {
  const float* X = lds + sharp_bwd_x_off();
  float* G = lds + sharp_bwd_g_off();
  const float* M = lds + sharp_bwd_m_off();
  const int gy = y0 + r;
  const bool interior = r >= 0 && r < kTileH && j0 >= 0 && j0 < kTileW;
  float up[4], ce[4], dn[4], gq[4], mq[4], dz[4];
  if (n == 4) {
    tile_quad(X, 2, c, r - 1, j0, up);
    tile_quad(X, 2, c, r, j0, ce);
    tile_quad(X, 2, c, r + 1, j0, dn);
    tile_quad(G, 1, c, r, j0, gq);
    if (a.mask_ch) tile_quad(M, 1, a.mask_ch == 3 ? c : 0, r, j0, mq);
  } else {
    up[0] = tile_at(X, 2, c, r - 1, j0); ce[0] = tile_at(X, 2, c, r, j0); dn[0] = tile_at(X, 2, c, r + 1, j0);
    gq[0] = tile_at(G, 1, c, r, j0);
    if (a.mask_ch) mq[0] = tile_at(M, 1, a.mask_ch == 3 ? c : 0, r, j0);
  }
  float L, R;
  if (n == 4) { const Sides sd = quad_sides(X, 2, c, r, j0, ce); L = sd.L; R = sd.R; }
  else { L = tile_at(X, 2, c, r, j0 - 1); R = tile_at(X, 2, c, r, j0 + n); }
  // an aligned quad (n == 4, W % 4 == 0) is entirely inside or outside the image
  const bool row_in = gy >= 0 && gy < a.H;
  const bool quad_in = row_in && x0 + j0 >= 0 && x0 + j0 < a.W;
  T2O_UNROLL
  for (int i = 0; i < 4; ++i) {
    if (i < n) {
      const bool in = n == 4 ? quad_in : (row_in && x0 + j0 + i >= 0 && x0 + j0 + i < a.W);
      const float left = i == 0 ? L : ce[i > 0 ? i - 1 : 0], right = i == n - 1 ? R : ce[i < 3 ? i + 1 : 3];
      const float d = sharp_delta(ce[i], up[i], left, right, dn[i]);
      const float m = a.mask_ch ? mq[i] : 1.0f;
      float z = ce[i] + p * d;
      if (a.mask_ch) z = blend(z, ce[i], m);
      const float gz = a.target ? sign_of(clamp01(z) - gq[i]) * gs : gq[i];
      dz[i] = (in && z >= 0.0f && z <= 1.0f) ? gz : 0.0f;
      if (interior) red0 += dz[i] * m * d;
    }
  }
  float* dst = G + c * tile_floats(1) + (r + 1) * kRowStride + kIntOff + j0;
  if (n == 4) store_vec<4>(dst, dz); else dst[0] = dz[0];
}

// Work items are (plane, row, quad): 3 x 18 x 16 = 864 aligned quads + 108 halo elements over 256
// threads = 3.8 rounds at 95 % occupancy (whole pixels per item left the second round 87 % idle).
template <int V>
T2O_HD void sharp_bwd_phase_dz(const OpArgs& a, int b, int tile, int tid, float* lds, float& red0) {
  int y0, x0;
  tile_origin(a, tile, y0, x0);
  const float p = a.param[(size_t)b * a.param_stride];
  const float gs = a.target ? a.gloss[0] * a.inv_n : 0.0f;
  const int rows = kTileH + 2;
  if (V == 4) {
    const int per = rows * (kTileW / 4);
    for (int i = tid; i < 3 * per; i += kThreads) {                       // aligned quads of the 64 interior columns
      const int c = i / per, rem = i % per;
      sharp_dz_quad(a, p, gs, y0, x0, c, rem / (kTileW / 4) - 1, 4 * (rem % (kTileW / 4)), 4, lds, red0);
    }
    for (int i = tid; i < 3 * rows * 2; i += kThreads) {                  // the two halo columns
      const int c = i / (rows * 2), rem = i % (rows * 2);
      sharp_dz_quad(a, p, gs, y0, x0, c, rem / 2 - 1, (rem % 2) ? kTileW : -1, 1, lds, red0);
    }
  } else {
    const int cols = kTileW + 2, per = rows * cols;
    for (int i = tid; i < 3 * per; i += kThreads) {
      const int c = i / per, rem = i % per;
      sharp_dz_quad(a, p, gs, y0, x0, c, rem / cols - 1, rem % cols - 1, 1, lds, red0);
    }
  }
}

// Phase 3: gimg = dz (1 - m) + do + p * Laplacian(do), do = dz * m
template <int V>
T2O_HD void sharp_bwd_phase_out(const OpArgs& a, int b, int tile, int tid, const float* lds) {
  int y0, x0;
  tile_origin(a, tile, y0, x0);
  const int ty = tid / (kTileW / 4), tx = tid % (kTileW / 4);
  const int gy = y0 + ty, gx0 = x0 + 4 * tx;
  const bool live = gy < a.H && gx0 < a.W && a.gimg;
  const float p = a.param[(size_t)b * a.param_stride];
  const float* G = lds + sharp_bwd_g_off();
  const float* M = lds + sharp_bwd_m_off();
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    const int mc = a.mask_ch == 3 ? c : 0;
    float up[4], ce[4], dn[4], o[4], pass[4];
    tile_quad(G, 1, c, ty - 1, 4 * tx, up);
    tile_quad(G, 1, c, ty, 4 * tx, ce);
    tile_quad(G, 1, c, ty + 1, 4 * tx, dn);
    const Sides sg = quad_sides(G, 1, c, ty, 4 * tx, ce);
    float L = sg.L, R = sg.R;
    T2O_UNROLL
    for (int i = 0; i < 4; ++i) pass[i] = 0.0f;
    if (a.mask_ch) {
      float mu[4], mcq[4], md[4];
      tile_quad(M, 1, mc, ty - 1, 4 * tx, mu);
      tile_quad(M, 1, mc, ty, 4 * tx, mcq);
      tile_quad(M, 1, mc, ty + 1, 4 * tx, md);
      const Sides sm = quad_sides(M, 1, mc, ty, 4 * tx, mcq);
      L *= sm.L;
      R *= sm.R;
      T2O_UNROLL
      for (int i = 0; i < 4; ++i) {
        pass[i] = ce[i] * (1.0f - mcq[i]);
        up[i] *= mu[i]; ce[i] *= mcq[i]; dn[i] *= md[i];
      }
    }
    T2O_UNROLL
    for (int i = 0; i < 4; ++i) {
      const float left = i == 0 ? L : ce[i > 0 ? i - 1 : 0], right = i == 3 ? R : ce[i < 3 ? i + 1 : 3];
      o[i] = pass[i] + (ce[i] + p * sharp_delta(ce[i], up[i], left, right, dn[i]));
    }
    if (!live) continue;
    float* dst = a.gimg + plane_off(a, b, c) + (size_t)gy * a.W + gx0;
    if (V == 4) store_vec<4>(dst, o);
    else for (int i = 0; i < 4 && gx0 + i < a.W; ++i) dst[i] = o[i];
  }
}

// ===================================================================== fused pointwise chain
// A known operator list (executor benchmark, planner) applied in ONE pass: the image is read
// once, the K pointwise operators run in registers, the result is written once.  Sharpness
// (a stencil) ends a chain segment; the host splits sequences at it (t2o_kernels.hip).
//
// Curves go through the per-sample lookup table above (one table row per chain operator, in LDS).
constexpr int kMaxChain = 8;
constexpr int kAccStride = 65;           // floats per slot row of the LDS accumulators (64 quads + pad)

struct ChainArgs {
  const float* img;        // (B,3,H,W) segment input
  const float* params;     // (Ktotal,B,24) parameters of the whole sequence; operator k uses row src[k]
  const float* gout;       // backward: gradient w.r.t. the segment output (null when L1 fused)
  const float* target;     // fused L1 target or null
  const float* gloss;      // fused L1 backward: device scalar
  float* out;              // forward: segment output
  float* gimg;             // backward: gradient w.r.t. the segment input (or null)
  float* partials;         // backward: (B, nblk, S) raw parameter-gradient sums
  float* loss_partials;    // fused L1 forward: (B, nblk)
  int ops[kMaxChain];
  int src[kMaxChain];            // index of operator k in the sequence's params / gparams
  int slot_off[kMaxChain + 1];   // first raw-sum slot of operator k in `partials`; S = slot_off[kMaxChain]
  int bin_off[kMaxChain + 1];    // first LDS accumulator cell row of operator k (curves: 16 per curve row)
  int K, B, H, W, iters, nblk;
  int param_stride, gparam_stride;
  float inv_n;
};

T2O_HD bool is_curve(int op) { return op == OP_COLOR || op == OP_TONE; }

// One thread per operator builds that operator's table row.
T2O_HD void chain_build_table(const ChainArgs& a, int b, int k, float* tab) {
  float* t = tab + k * kTabStride;
  const float* p = a.params + ((size_t)a.src[k] * a.B + b) * a.param_stride;
  const int op = a.ops[k];
  if (!is_curve(op)) { t[0] = p[0]; return; }
  curve_table_build(p, op == OP_COLOR, t);
}

// forward of chain operator `op` on one pixel (pre-clamp); t = this operator's table row
T2O_HD Rgb chain_op_fwd(int op, const Rgb& x, const float* t, bool unit = false) {
  Rgb o;
  switch (op) {
    case OP_BRIGHTNESS: return brightness_fwd(x, t[0]);
    case OP_CONTRAST:   return contrast_fwd(x, t[0]);
    case OP_SATURATION: return saturation_fwd(x, t[0]);
    case OP_COLOR:
    case OP_TONE:
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) o.c[c] = curve_lut_fwd(t, op == OP_COLOR, c, x.c[c], unit);
      return o;
    case OP_WHITE: o.c[0] = o.c[1] = o.c[2] = 1.0f; return o;
    default: return x;
  }
}

// backward of a curve operator on one pixel (COLOR: per-channel curves, else one shared curve):
// dz = gradient w.r.t. the PRE-clamp output (already zeroed where the clamp was active);
// returns the gradient w.r.t. the input, adds the raw sums red[8*row + j] += dz * t_j (first = true:
// this is the thread's first pixel, the sums are assigned instead and need no zeroing).
// (A segment-histogram variant -- two LDS float atomics per channel instead of 8 multiply-adds --
// measured 3x SLOWER on MI355X: ds_add_f32 with per-lane addresses runs at roughly one lane per
// 3 cycles per CU.  Kept out.)
template <bool COLOR>
T2O_HD Rgb chain_curve_bwd(const Rgb& x, const float* t, const Rgb& dz, float* red, bool first, bool unit = false) {
  Rgb gx;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    constexpr int kOne = COLOR ? 1 : 0;
    const int cc = c * kOne;                                         // static after unrolling
    gx.c[c] = curve_lut_bwd_1(t, COLOR, c, x.c[c], dz.c[c], red + cc * kCurveSteps, first && (COLOR || c == 0), unit);
  }
  return gx;
}

// raw-sum slot s of `partials` from this workgroup's summed accumulator cells
T2O_HD float chain_slot_value(const ChainArgs& a, int s, const float* bsum) { return bsum[s]; }

// backward of a one-parameter chain operator (brightness / contrast / saturation); dz as above
T2O_HD Rgb chain_scalar_bwd(int op, const Rgb& x, const float* t, const Rgb& dz, float* red) {
  Curve unused;
  float p0[1] = {t[0]};
  return pointwise_bwd(op, x, p0, unused, dz, red);
}

template <int V, bool L1>
T2O_HD float chain_fwd_thread(const ChainArgs& a, int b, int blk, int tid, const float* tab) {
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const unsigned groups = hw / V;
  const size_t sb = (size_t)b * 3 * hw;
  const float* xin = a.img + sb;
  float* o = a.out + sb;
  float l1 = 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned g = ((unsigned)blk * a.iters + it) * kThreads + tid;
    if (g >= groups) break;
    const unsigned px = g * V;
    float x[3][V];
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) load_vec<V>(xin + c * hw + px, x[c]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = 0; k < a.K; ++k) {                 // a real loop: the forward keeps no per-operator state
      T2O_RELOAD_FENCE();
      {
        const int op = a.ops[k];
        T2O_UNROLL
        for (int i = 0; i < V; ++i) {
          Rgb xi = {{x[0][i], x[1][i], x[2][i]}};
          const Rgb r = chain_op_fwd(op, xi, tab + k * kTabStride);
          T2O_UNROLL
          for (int c = 0; c < 3; ++c) x[c][i] = clamp01(r.c[c]);
        }
      }
    }
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) store_vec<V>(o + c * hw + px, x[c]);
    if (L1) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        float t[V];
        load_vec<V>(a.target + sb + c * hw + px, t);
        T2O_UNROLL
        for (int i = 0; i < V; ++i) l1 += fabsf(x[c][i] - t[i]);
      }
    }
  }
  return l1;
}

// ACC::add(slot, value) accumulates a raw parameter sum (device: quad reduce + LDS; host: array).
// `sv` = per-thread save area in LDS for the input of every operator: element
// ((k*3 + c)*V + i) of thread tid lives at sv[(((k*3 + c)*V + i) * kThreads) + tid] (bank-conflict
// free), so both operator loops are real loops with a run-time k: compact code, few registers.
template <int V>
T2O_HD int chain_save_floats(int K) { return K * 3 * V * kThreads; }

// Returns (L1 forms) the thread's sum of |chain(img) - target| over its live pixels: the "value" of a value-and-gradient
// call -- the backward recomputes the forward anyway, so the loss (and, when a.out is set, the final image) costs no
// separate forward launch (t2o_fused_sequence_l1_value_grad).
template <int V, bool L1, class ACC>
T2O_HD float chain_bwd_thread(const ChainArgs& a, int b, int blk, int tid, const float* tab, float* sv, ACC& acc) {
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const unsigned groups = hw / V;
  const size_t sb = (size_t)b * 3 * hw;
  const float* xin = a.img + sb;
  const float* gin = (L1 ? a.target : a.gout) + sb;
  const float gs = L1 ? a.gloss[0] * a.inv_n : 0.0f;
  float l1 = 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned g = ((unsigned)blk * a.iters + it) * kThreads + tid;
    const bool live = g < groups;                 // dead threads still take part in the quad reductions
    const unsigned px = live ? g * V : 0;
    float x[3][V], gg[3][V];
    unsigned pass[V];                              // bit 3k+c: the clamp after operator k passed channel c
    T2O_UNROLL
    for (int i = 0; i < V; ++i) pass[i] = 0u;
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) {
      load_vec<V>(xin + c * hw + px, x[c]);
      load_vec<V>(gin + c * hw + px, gg[c]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = 0; k < a.K; ++k) {
      T2O_RELOAD_FENCE();
      const int op = a.ops[k];
      T2O_UNROLL
      for (int i = 0; i < V; ++i) {
        Rgb xi = {{x[0][i], x[1][i], x[2][i]}};
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) sv[((k * 3 + c) * V + i) * kThreads + tid] = xi.c[c];
        const Rgb r = chain_op_fwd(op, xi, tab + k * kTabStride);
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) {
          pass[i] |= (r.c[c] >= 0.0f && r.c[c] <= 1.0f) ? (1u << (3 * k + c)) : 0u;   // clamp(0,1) backward is inclusive
          x[c][i] = clamp01(r.c[c]);
        }
      }
    }
    if (a.out && live) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) store_vec<V>(a.out + sb + c * hw + px, x[c]);
    }
    if (L1) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        T2O_UNROLL
        for (int i = 0; i < V; ++i) {
          const float d = x[c][i] - gg[c][i];
          if (live) l1 += fabsf(d);
          gg[c][i] = sign_of(d) * gs;
        }
      }
    }
    if (!live) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        T2O_UNROLL
        for (int i = 0; i < V; ++i) gg[c][i] = 0.0f;
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = a.K - 1; k >= 0; --k) {
      T2O_RELOAD_FENCE();
      const int op = a.ops[k];
      const float* t = tab + k * kTabStride;
      const int bin0 = a.bin_off[k];
      // one specialised body per operator class: static register indices, static slot counts
#define T2O_CHAIN_BWD_BODY(NRED, CALL)                                                   \
      {                                                                                  \
        float red[NRED];                                                                 \
        if (NRED == 1) red[0] = 0.0f;      /* curve bodies assign on their first pixel */ \
        T2O_UNROLL                                                                       \
        for (int i = 0; i < V; ++i) {                                                    \
          Rgb xi, gi;                                                                    \
          T2O_UNROLL                                                                     \
          for (int c = 0; c < 3; ++c) {                                                  \
            xi.c[c] = sv[((k * 3 + c) * V + i) * kThreads + tid];                        \
            gi.c[c] = ((pass[i] >> (3 * k + c)) & 1u) ? gg[c][i] : 0.0f;                 \
          }                                                                              \
          const Rgb gx = CALL;                                                           \
          T2O_UNROLL                                                                     \
          for (int c = 0; c < 3; ++c) gg[c][i] = gx.c[c];                                \
        }                                                                                \
        acc.template add_n<NRED>(bin0, red);                                             \
      }
      if (op == OP_COLOR) T2O_CHAIN_BWD_BODY(24, chain_curve_bwd<true>(xi, t, gi, red, i == 0))
      else if (op == OP_TONE) T2O_CHAIN_BWD_BODY(8, chain_curve_bwd<false>(xi, t, gi, red, i == 0))
      else if (op == OP_WHITE) {
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) {
          T2O_UNROLL
          for (int i = 0; i < V; ++i) gg[c][i] = 0.0f;
        }
      } else T2O_CHAIN_BWD_BODY(1, chain_scalar_bwd(op, xi, t, gi, red))
#undef T2O_CHAIN_BWD_BODY
    }
    if (a.gimg && live) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) store_vec<V>(a.gimg + sb + c * hw + px, gg[c]);
    }
  }
  return l1;
}

// ---------------------------------------------------------------------------------------------
// The same backward for an operator list known at COMPILE time (template arguments): the two sweeps are
// unrolled over the operators, so there is no `for k` / `switch (op)` control flow and none of its scalar
// traffic, every operator's raw parameter sums live in their own registers for ALL the thread's pixels and
// are flushed (quad reduction + owner-lane LDS add) once per thread instead of once per pixel, and the saved
// operator inputs are plain registers (no LDS save area).  Same arithmetic per pixel as chain_bwd_thread.
// Instantiated for the sequences callers actually fix ahead of time (t2o_kernels.hip: kStaticChains).
template <int... OPS>
struct StaticChain {
  static constexpr int K = sizeof...(OPS);
  static constexpr int ops[K > 0 ? K : 1] = {OPS...};
};
T2O_HD constexpr int chain_nred(int op) { return op == OP_COLOR ? 24 : op == OP_TONE ? 8 : op == OP_WHITE ? 0 : 1; }

// forward for a compile-time operator list: the operator loop unrolled (no `for k`, no `switch`), V pixels per
// thread-iteration as the run-time-loop program
template <int V, bool L1, class SEQ>
T2O_HD float chain_fwd_thread_static(const ChainArgs& a, int b, int blk, int tid, const float* tab) {
  constexpr int K = SEQ::K;
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const unsigned groups = hw / V;
  const size_t sb = (size_t)b * 3 * hw;
  const float* xin = a.img + sb;
  float* o = a.out + sb;
  float l1 = 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned g = ((unsigned)blk * a.iters + it) * kThreads + tid;
    if (g >= groups) break;
    const unsigned px = g * V;
    float x[3][V];
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) load_vec<V>(xin + c * hw + px, x[c]);
    T2O_UNROLL
    for (int k = 0; k < K; ++k) {
      T2O_UNROLL
      for (int i = 0; i < V; ++i) {
        Rgb xi = {{x[0][i], x[1][i], x[2][i]}};
        const Rgb r = chain_op_fwd(SEQ::ops[k], xi, tab + k * kTabStride, k > 0);   // k > 0: xi = clamp01(...)
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) x[c][i] = clamp01(r.c[c]);
      }
    }
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) store_vec<V>(o + c * hw + px, x[c]);
    if (L1) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        float t[V];
        load_vec<V>(a.target + sb + c * hw + px, t);
        T2O_UNROLL
        for (int i = 0; i < V; ++i) l1 += fabsf(x[c][i] - t[i]);
      }
    }
  }
  return l1;
}

template <bool L1, class SEQ, bool SV_LDS, class ACC, bool VAL = L1>
T2O_HD float chain_bwd_thread_static(const ChainArgs& a, int b, int blk, int tid, const float* tab, float* svl, ACC& acc) {
  constexpr int K = SEQ::K;
  float l1 = 0.0f;                                   // (L1 forms: sum |chain(img) - target| of the live pixels, see chain_bwd_thread)
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const size_t sb = (size_t)b * 3 * hw;
  const float* xin = a.img + sb;
  const float* gin = (L1 ? a.target : a.gout) + sb;
  const float gs = L1 ? a.gloss[0] * a.inv_n : 0.0f;
  float red[K][24];                                  // statically indexed everywhere: registers, unused entries vanish
  T2O_UNROLL
  for (int k = 0; k < K; ++k) red[k][0] = 0.0f;      // curve rows are ASSIGNED on the first pixel instead
  // software prefetch: the loads of pixel it+1 are issued before the ~900 vector instructions of pixel it, so the
  // 3 waves per SIMD this kernel's registers allow no longer sit in s_waitcnt at the top of every iteration
  // (SQ_WAIT_ANY was 33 % of the wave cycles without it)
  float xn[3], gn[3];
  {
    const unsigned g0 = ((unsigned)blk * a.iters) * kThreads + tid;
    const unsigned p0 = g0 < hw ? g0 : 0;
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) { xn[c] = xin[c * hw + p0]; gn[c] = gin[c * hw + p0]; }
  }
  for (int it = 0; it < a.iters; ++it) {
    const unsigned g = ((unsigned)blk * a.iters + it) * kThreads + tid;
    const bool live = g < hw;                        // dead threads carry zeros into the reductions
    const unsigned px = live ? g : 0;
    float x[3], gg[3], sv[SV_LDS ? 1 : K][3];        // SV_LDS: operator inputs saved in the LDS area instead (fewer registers)
    bool pass[K][3];                                 // clamp passed? -- lane masks in scalar registers, no vector register
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) { x[c] = xn[c]; gg[c] = gn[c]; }
    if (it + 1 < a.iters) {
      const unsigned g1 = g + kThreads;
      const unsigned p1 = g1 < hw ? g1 : 0;
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) { xn[c] = xin[c * hw + p1]; gn[c] = gin[c * hw + p1]; }
    }
    T2O_UNROLL
    for (int k = 0; k < K; ++k) {
      Rgb xi = {{x[0], x[1], x[2]}};
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        if (SV_LDS) svl[(k * 3 + c) * kThreads + tid] = xi.c[c]; else sv[k][c] = xi.c[c];
      }
      const Rgb r = chain_op_fwd(SEQ::ops[k], xi, tab + k * kTabStride, k > 0);   // k > 0: xi = clamp01(...)
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        x[c] = clamp01(r.c[c]);
        pass[k][c] = x[c] == r.c[c];                   // == (0 <= r && r <= 1), NaN included: one compare on the clamped value
      }
    }
    if constexpr (L1 && VAL) {
      if (a.out && live) {
        T2O_UNROLL
        for (int c = 0; c < 3; ++c) a.out[sb + c * hw + px] = x[c];
      }
    }
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) {
      if (L1) {
        const float d = x[c] - gg[c];
        if constexpr (VAL) { if (live) l1 += fabsf(d); }
        gg[c] = sign_of(d) * gs;
      }
      if (!live) gg[c] = 0.0f;
    }
    T2O_UNROLL
    for (int kk = 0; kk < K; ++kk) {
      const int k = K - 1 - kk;
      const int op = SEQ::ops[k];
      const float* t = tab + k * kTabStride;
      Rgb xi, gi;
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) {
        xi.c[c] = SV_LDS ? svl[(k * 3 + c) * kThreads + tid] : sv[k][c];
        gi.c[c] = pass[k][c] ? gg[c] : 0.0f;
      }
      Rgb gx;
      if (op == OP_COLOR) gx = chain_curve_bwd<true>(xi, t, gi, red[k], it == 0, k > 0);
      else if (op == OP_TONE) gx = chain_curve_bwd<false>(xi, t, gi, red[k], it == 0, k > 0);
      else if (op == OP_WHITE) gx.c[0] = gx.c[1] = gx.c[2] = 0.0f;
      else gx = chain_scalar_bwd(op, xi, t, gi, red[k]);
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) gg[c] = gx.c[c];
    }
    if (a.gimg && live) {
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) a.gimg[sb + c * hw + px] = gg[c];
    }
  }
  T2O_UNROLL
  for (int k = 0; k < K; ++k) {                      // one flush per thread
    constexpr int kZero = 0;
    (void)kZero;
    const int op = SEQ::ops[k];
    if (op == OP_COLOR) { float (&r)[24] = red[k]; acc.template add_n<24>(a.bin_off[k], r); }
    else if (op == OP_TONE) { float (&r)[8] = reinterpret_cast<float (&)[8]>(red[k]); acc.template add_n<8>(a.bin_off[k], r); }
    else if (op != OP_WHITE) { float (&r)[1] = reinterpret_cast<float (&)[1]>(red[k]); acc.template add_n<1>(a.bin_off[k], r); }
  }
  return l1;
}

// ===================================================================== SSIM (evaluation metric)
// utils/ssim/__init__.py:20-40: 11x11 Gaussian window (sigma 1.5, zero padding 5) over x, y,
// x^2, y^2, xy per channel; C1 = 0.01^2, C2 = 0.03^2; mean of the SSIM map.  The 2-D window is the
// outer product of a normalised 1-D Gaussian, so it is applied separably: rows, then columns.
constexpr int kSsimWin = 11, kSsimPad = 5;
constexpr int kSsimTile = 32;                          // 32 x 32 outputs per workgroup
constexpr int kSsimIn = kSsimTile + 2 * kSsimPad;      // 42 x 42 input window
constexpr int kSsimInStride = kSsimIn + 1;             // LDS row strides (odd: no bank conflicts on column walks)
constexpr int kSsimHStride = kSsimTile + 1;

struct SsimArgs {
  const float* a;        // (B,C,H,W)
  const float* b;
  float* partials;       // (B*C, tiles)
  float g[kSsimWin];     // normalised 1-D Gaussian
  int B, C, H, W, tiles_x, tiles;
};

T2O_HD int ssim_lds_floats() { return 2 * kSsimIn * kSsimInStride + 5 * kSsimIn * kSsimHStride; }

#ifndef __HIPCC_RTC__      // (host-side set-up)
inline void ssim_window(float* g) {                     // utils/ssim/__init__.py:7-10
  float s = 0.0f;
  for (int i = 0; i < kSsimWin; ++i) {
    g[i] = expf(-(float)((i - kSsimWin / 2) * (i - kSsimWin / 2)) / (2.0f * 1.5f * 1.5f));
    s += g[i];
  }
  for (int i = 0; i < kSsimWin; ++i) g[i] /= s;
}
#endif

// phase 1: load the two 42x42 input windows (zero outside the image)
T2O_HD void ssim_phase_load(const SsimArgs& s, int plane, int tile, int tid, float* lds) {
  const int y0 = (tile / s.tiles_x) * kSsimTile - kSsimPad, x0 = (tile % s.tiles_x) * kSsimTile - kSsimPad;
  const size_t base = (size_t)plane * s.H * s.W;
  for (int i = tid; i < kSsimIn * kSsimIn; i += kThreads) {
    const int r = i / kSsimIn, c = i % kSsimIn, gy = y0 + r, gx = x0 + c;
    const bool in = gy >= 0 && gy < s.H && gx >= 0 && gx < s.W;
    lds[r * kSsimInStride + c] = in ? s.a[base + (size_t)gy * s.W + gx] : 0.0f;
    lds[kSsimIn * kSsimInStride + r * kSsimInStride + c] = in ? s.b[base + (size_t)gy * s.W + gx] : 0.0f;
  }
}

// phase 2: horizontal pass -> 5 maps of 42 rows x 32 columns
T2O_HD void ssim_phase_rows(const SsimArgs& s, int tid, float* lds) {
  const float* A = lds;
  const float* Bm = lds + kSsimIn * kSsimInStride;
  float* Hm = lds + 2 * kSsimIn * kSsimInStride;
  for (int i = tid; i < kSsimIn * kSsimTile; i += kThreads) {
    const int r = i / kSsimTile, c = i % kSsimTile;
    float m1 = 0.0f, m2 = 0.0f, e11 = 0.0f, e22 = 0.0f, e12 = 0.0f;
    for (int k = 0; k < kSsimWin; ++k) {
      const float x = A[r * kSsimInStride + c + k], y = Bm[r * kSsimInStride + c + k], w = s.g[k];
      m1 += w * x; m2 += w * y; e11 += w * (x * x); e22 += w * (y * y); e12 += w * (x * y);
    }
    float* o = Hm + r * kSsimHStride + c;
    o[0] = m1; o[kSsimIn * kSsimHStride] = m2; o[2 * kSsimIn * kSsimHStride] = e11;
    o[3 * kSsimIn * kSsimHStride] = e22; o[4 * kSsimIn * kSsimHStride] = e12;
  }
}

// phase 3: vertical pass + SSIM map; returns this thread's sum over its in-image outputs
T2O_HD float ssim_phase_cols(const SsimArgs& s, int tile, int tid, const float* lds) {
  const float* Hm = lds + 2 * kSsimIn * kSsimInStride;
  const int y0 = (tile / s.tiles_x) * kSsimTile, x0 = (tile % s.tiles_x) * kSsimTile;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float sum = 0.0f;
  for (int i = tid; i < kSsimTile * kSsimTile; i += kThreads) {
    const int r = i / kSsimTile, c = i % kSsimTile;
    if (y0 + r >= s.H || x0 + c >= s.W) continue;
    float v[5];
    for (int m = 0; m < 5; ++m) {
      float acc = 0.0f;
      for (int k = 0; k < kSsimWin; ++k) acc += s.g[k] * Hm[m * kSsimIn * kSsimHStride + (r + k) * kSsimHStride + c];
      v[m] = acc;
    }
    const float mu1 = v[0], mu2 = v[1];
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = v[2] - mu1_sq, s2 = v[3] - mu2_sq, s12 = v[4] - mu12;
    sum += ((2.0f * mu12 + C1) * (2.0f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
  }
  return sum;
}

// --------------------------------------------------------------------- SSIM backward
// The reference has no SSIM backward of its own (utils/ssim/__init__.py:20-40 under autograd); this is the closed form.
// With W the (symmetric, zero-padded) Gaussian window, mu1 = W*x, mu2 = W*y, e11 = W*(x x), e22 = W*(y y), e12 = W*(x y),
//   A1 = 2 mu1 mu2 + C1,  A2 = 2 (e12 - mu1 mu2) + C2,  B1 = mu1^2 + mu2^2 + C1,  B2 = e11 - mu1^2 + e22 - mu2^2 + C2,
//   S = A1 A2 / (B1 B2);  out[b] = mean_{c,h,w} S;  gS = gout[b] / (C H W) at every in-image position q, and
//   d_e   = dS/de11 = dS/de22 = -S / B2
//   d_e12 = 2 A1 / (B1 B2)
//   d_mu1 = 2 mu2 (A2 - A1) / (B1 B2) - 2 mu1 S (1 / B1 - 1 / B2)       (d_mu2: mu1 <-> mu2)
//   dL/dx = W*(gS d_mu1) + 2 x W*(gS d_e) + y W*(gS d_e12),   dL/dy = W*(gS d_mu2) + 2 y W*(gS d_e) + x W*(gS d_e12).
// One workgroup = one 32 x 32 tile of one plane: it needs the four derivative maps on the 42 x 42 positions around the tile,
// hence the statistics there, hence a 52 x 52 input window; everything stays in LDS (the four maps never reach HBM).
constexpr int kSsimBIn = kSsimTile + 4 * kSsimPad;      // 52 x 52 input window
constexpr int kSsimBMid = kSsimTile + 2 * kSsimPad;     // 42 x 42 positions whose SSIM value reaches the tile
constexpr int kSsimBInStride = kSsimBIn + 1, kSsimBMidStride = kSsimBMid + 1, kSsimBOutStride = kSsimTile + 1;
constexpr int kSsimBX = 0;                                               // x window, then y window
constexpr int kSsimBH = 2 * kSsimBIn * kSsimBInStride;                   // 5 row-filtered maps (52 rows x 42); later 4 x (42 rows x 32)
constexpr int kSsimBD = kSsimBH + 5 * kSsimBIn * kSsimBMidStride;        // 4 derivative maps (42 x 42)
T2O_HD int ssim_bwd_lds_floats() { return kSsimBD + 4 * kSsimBMid * kSsimBMidStride; }

struct SsimBwdArgs {
  const float* a;        // (B,C,H,W)
  const float* b;
  const float* gout;     // (B): gradient w.r.t. out[b] of t2o_ssim_fwd
  float* ga;             // (B,C,H,W) or null
  float* gb;             // (B,C,H,W) or null
  float g[kSsimWin];
  int B, C, H, W, tiles_x, tiles;
  float inv_n;           // 1 / (C H W)
};

// phase 1: the two 52 x 52 input windows (zero outside the image)
T2O_HD void ssim_bwd_phase_load(const SsimBwdArgs& s, int plane, int tile, int tid, float* lds) {
  const int y0 = (tile / s.tiles_x) * kSsimTile - 2 * kSsimPad, x0 = (tile % s.tiles_x) * kSsimTile - 2 * kSsimPad;
  const size_t base = (size_t)plane * s.H * s.W;
  for (int i = tid; i < kSsimBIn * kSsimBIn; i += kThreads) {
    const int r = i / kSsimBIn, c = i % kSsimBIn, gy = y0 + r, gx = x0 + c;
    const bool in = gy >= 0 && gy < s.H && gx >= 0 && gx < s.W;
    lds[kSsimBX + r * kSsimBInStride + c] = in ? s.a[base + (size_t)gy * s.W + gx] : 0.0f;
    lds[kSsimBX + kSsimBIn * kSsimBInStride + r * kSsimBInStride + c] = in ? s.b[base + (size_t)gy * s.W + gx] : 0.0f;
  }
}

// phase 2: horizontal pass -> 5 maps of 52 rows x 42 columns
T2O_HD void ssim_bwd_phase_rows(const SsimBwdArgs& s, int tid, float* lds) {
  const float* A = lds + kSsimBX;
  const float* Bm = A + kSsimBIn * kSsimBInStride;
  float* Hm = lds + kSsimBH;
  for (int i = tid; i < kSsimBIn * kSsimBMid; i += kThreads) {
    const int r = i / kSsimBMid, c = i % kSsimBMid;
    float m1 = 0.0f, m2 = 0.0f, e11 = 0.0f, e22 = 0.0f, e12 = 0.0f;
    for (int k = 0; k < kSsimWin; ++k) {
      const float x = A[r * kSsimBInStride + c + k], y = Bm[r * kSsimBInStride + c + k], w = s.g[k];
      m1 += w * x; m2 += w * y; e11 += w * (x * x); e22 += w * (y * y); e12 += w * (x * y);
    }
    float* o = Hm + r * kSsimBMidStride + c;
    constexpr int kMap = kSsimBIn * kSsimBMidStride;
    o[0] = m1; o[kMap] = m2; o[2 * kMap] = e11; o[3 * kMap] = e22; o[4 * kMap] = e12;
  }
}

// phase 3: vertical pass -> the statistics at the 42 x 42 positions -> the four derivative maps times gS (zero outside the image)
T2O_HD void ssim_bwd_phase_deriv(const SsimBwdArgs& s, int plane, int tile, int tid, float* lds) {
  const float* Hm = lds + kSsimBH;
  float* D = lds + kSsimBD;
  constexpr int kMap = kSsimBIn * kSsimBMidStride, kDMap = kSsimBMid * kSsimBMidStride;
  const int y0 = (tile / s.tiles_x) * kSsimTile - kSsimPad, x0 = (tile % s.tiles_x) * kSsimTile - kSsimPad;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const float gs = s.gout[plane / s.C] * s.inv_n;
  for (int i = tid; i < kSsimBMid * kSsimBMid; i += kThreads) {
    const int r = i / kSsimBMid, c = i % kSsimBMid;
    float d1 = 0.0f, d2 = 0.0f, de = 0.0f, d12 = 0.0f;
    if (y0 + r >= 0 && y0 + r < s.H && x0 + c >= 0 && x0 + c < s.W) {
      float v[5];
      for (int m = 0; m < 5; ++m) {
        float acc = 0.0f;
        for (int k = 0; k < kSsimWin; ++k) acc += s.g[k] * Hm[m * kMap + (r + k) * kSsimBMidStride + c];
        v[m] = acc;
      }
      const float mu1 = v[0], mu2 = v[1];
      const float A1 = 2.0f * mu1 * mu2 + C1, A2 = 2.0f * (v[4] - mu1 * mu2) + C2;
      const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = (v[2] - mu1 * mu1) + (v[3] - mu2 * mu2) + C2;
      const float rB1 = 1.0f / B1, rB2 = 1.0f / B2;
      const float S = A1 * A2 * rB1 * rB2;
      const float common = 2.0f * (A2 - A1) * rB1 * rB2, diff = 2.0f * S * (rB1 - rB2);
      d1 = gs * (mu2 * common - mu1 * diff);
      d2 = gs * (mu1 * common - mu2 * diff);
      de = -gs * S * rB2;
      d12 = gs * 2.0f * A1 * rB1 * rB2;
    }
    float* o = D + r * kSsimBMidStride + c;
    o[0] = d1; o[kDMap] = d2; o[2 * kDMap] = de; o[3 * kDMap] = d12;
  }
}

// phase 4: horizontal pass over the derivative maps -> 4 maps of 42 rows x 32 columns (over the row-filtered statistics, no longer needed)
T2O_HD void ssim_bwd_phase_drows(const SsimBwdArgs& s, int tid, float* lds) {
  const float* D = lds + kSsimBD;
  float* Hd = lds + kSsimBH;
  constexpr int kDMap = kSsimBMid * kSsimBMidStride, kHMap = kSsimBMid * kSsimBOutStride;
  for (int i = tid; i < kSsimBMid * kSsimTile; i += kThreads) {
    const int r = i / kSsimTile, c = i % kSsimTile;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < kSsimWin; ++k) {
      const float w = s.g[k];
      for (int m = 0; m < 4; ++m) acc[m] += w * D[m * kDMap + r * kSsimBMidStride + c + k];
    }
    for (int m = 0; m < 4; ++m) Hd[m * kHMap + r * kSsimBOutStride + c] = acc[m];
  }
}

// phase 5: vertical pass + the chain rule through x x, y y, x y -> the two image gradients of the tile
T2O_HD void ssim_bwd_phase_out(const SsimBwdArgs& s, int plane, int tile, int tid, const float* lds) {
  const float* Hd = lds + kSsimBH;
  const float* A = lds + kSsimBX;
  const float* Bm = A + kSsimBIn * kSsimBInStride;
  constexpr int kHMap = kSsimBMid * kSsimBOutStride;
  const int y0 = (tile / s.tiles_x) * kSsimTile, x0 = (tile % s.tiles_x) * kSsimTile;
  const size_t base = (size_t)plane * s.H * s.W;
  for (int i = tid; i < kSsimTile * kSsimTile; i += kThreads) {
    const int r = i / kSsimTile, c = i % kSsimTile;
    if (y0 + r >= s.H || x0 + c >= s.W) continue;
    float G[4];
    for (int m = 0; m < 4; ++m) {
      float acc = 0.0f;
      for (int k = 0; k < kSsimWin; ++k) acc += s.g[k] * Hd[m * kHMap + (r + k) * kSsimBOutStride + c];
      G[m] = acc;
    }
    const float x = A[(r + 2 * kSsimPad) * kSsimBInStride + c + 2 * kSsimPad], y = Bm[(r + 2 * kSsimPad) * kSsimBInStride + c + 2 * kSsimPad];
    const size_t o = base + (size_t)(y0 + r) * s.W + x0 + c;
    if (s.ga) s.ga[o] = G[0] + 2.0f * x * G[2] + y * G[3];
    if (s.gb) s.gb[o] = G[1] + 2.0f * y * G[2] + x * G[3];
  }
}

// ===================================================================== planner: candidate sweep
// Operation planning (utils/beam_search.py:65-91) fits one operator's parameter by minimising
// dist(execute(img, op, param), target) with one executor call + one `.item()` per evaluation.
// Here C candidate parameter rows are evaluated against ONE image pair in a single launch: a
// workgroup keeps its pixels in registers and walks a group of candidates (curve tables in LDS).
constexpr int kCandPerBlock = 8;      // candidates per workgroup
constexpr int kCandPix = 8;           // pixels per thread

struct CandArgs {
  const float* img;       // (3,H,W)
  const float* target;    // (3,H,W)
  const float* params;    // (C, param_stride)
  float* partials;        // (C, nblk) per-workgroup sums of |out - target|
  int op, C, param_stride, H, W, nblk;
};

T2O_HD void cand_build_table(const CandArgs& a, int cand, float* t) {
  const float* p = a.params + (size_t)cand * a.param_stride;
  if (!is_curve(a.op)) { t[0] = p[0]; return; }
  for (int c = 0; c < 3; ++c) {
    const float* row = a.op == OP_COLOR ? p + c * kCurveSteps : p;
    float s = 0.0f, run = 0.0f;
    for (int i = 0; i < kCurveSteps; ++i) {
      t[kTabK + c * kCurveSteps + i] = row[i];
      t[kTabP + c * (kCurveSteps + 1) + i] = run;
      run = run + (1.0f / kCurveSteps) * row[i];
      s = s + row[i];
    }
    t[kTabP + c * (kCurveSteps + 1) + kCurveSteps] = run;
    s = s + 1e-10f;
    t[kTabSum + c] = s;
    t[kTabRsum + c] = 1.0f / s;
    t[kTabScale + c] = t[kTabRsum + c] * (float)kCurveSteps;
  }
}

// sum over this thread's pixels of |clamp(op(x; candidate)) - target| for candidate slot j of the group
T2O_HD float cand_eval(const CandArgs& a, const float* t, const float (&x)[3][kCandPix], const float (&tg)[3][kCandPix],
                       int npx) {
  float s = 0.0f;
  for (int i = 0; i < kCandPix; ++i) {
    if (i < npx) {
      Rgb xi = {{x[0][i], x[1][i], x[2][i]}};
      const Rgb r = chain_op_fwd(a.op, xi, t);
      T2O_UNROLL
      for (int c = 0; c < 3; ++c) s += fabsf(clamp01(r.c[c]) - tg[c][i]);
    }
  }
  return s;
}

// loads this thread's pixels (strided by 256 inside the block's chunk); returns how many are valid
T2O_HD int cand_load(const CandArgs& a, int blk, int tid, float (&x)[3][kCandPix], float (&tg)[3][kCandPix]) {
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  int n = 0;
  T2O_UNROLL
  for (int i = 0; i < kCandPix; ++i) {
    const unsigned px = ((unsigned)blk * kCandPix + i) * kThreads + tid;
    const bool ok = px < hw;
    T2O_UNROLL
    for (int c = 0; c < 3; ++c) {
      x[c][i] = ok ? a.img[c * hw + px] : 0.0f;
      tg[c][i] = ok ? a.target[c * hw + px] : 0.0f;
    }
    n += ok ? 1 : 0;
  }
  return n;     // valid pixels are the first n (px grows with i)
}

// ===================================================================== sequence planning (host)
// A sequence is cut into segments: maximal runs of pointwise operators (<= kMaxChain) that one
// fused kernel pair handles, and single sharpness operators (stencil kernels).  Identity (-1)
// entries are dropped.  Only segment boundaries are materialised in HBM.
struct Segment {
  int first, count;     // operators [first, first+count) of the ORIGINAL list (count == 1 for sharpness)
  bool sharp;
  int ops[kMaxChain];   // chain: the operator ids; identity entries removed
  int src[kMaxChain];   // chain: index in the original list of each kept operator
  int n;                // chain: number of kept operators (may be 0: pure copy)
};

// returns the number of segments (>= 1), or -1 for an unsupported operator id
inline int plan_segments(const int* ops, int K, Segment* seg, int max_seg) {
  int ns = 0;
  Segment cur;
  cur.first = 0; cur.count = 0; cur.sharp = false; cur.n = 0;
  for (int k = 0; k < K; ++k) {
    const int op = ops[k];
    if (!(op == OP_IDENTITY || (op >= 0 && op <= 7 && op != OP_INPAINT))) return -1;
    if (op == OP_SHARPNESS || (op != OP_IDENTITY && cur.n == kMaxChain)) {
      if (cur.n > 0) {
        if (ns >= max_seg) return -1;
        seg[ns++] = cur;
      }
      cur.first = k; cur.count = 0; cur.sharp = false; cur.n = 0;
    }
    if (op == OP_SHARPNESS) {
      Segment s;
      s.first = k; s.count = 1; s.sharp = true; s.n = 0;
      if (ns >= max_seg) return -1;
      seg[ns++] = s;
      cur.first = k + 1;
      continue;
    }
    if (op != OP_IDENTITY) { cur.ops[cur.n] = op; cur.src[cur.n] = k; ++cur.n; }
    ++cur.count;
  }
  if (cur.n > 0 || ns == 0) {
    if (ns >= max_seg) return -1;
    seg[ns++] = cur;
  }
  return ns;
}

inline void chain_geometry(int B, int H, int W, int forced_iters, int& vec, int& iters, int& nblk, int forced_vec = 0) {
  const size_t hw = (size_t)H * W;
  vec = (hw % 2 == 0 && forced_vec != 1) ? 2 : 1;
  const size_t groups = hw / vec;
  size_t it = forced_iters > 0 ? (size_t)forced_iters : (groups * (size_t)B) / ((size_t)kThreads * 4096);
  if (it < 1) it = 1;
  if (it > (forced_iters > 0 ? 64 : 8)) it = forced_iters > 0 ? 64 : 8;
  iters = (int)it;
  nblk = (int)((groups + (size_t)kThreads * it - 1) / ((size_t)kThreads * it));
}

inline void chain_fill(ChainArgs& a, const Segment& s, int B, int H, int W, int iters, int nblk) {
  a.K = s.n; a.B = B; a.H = H; a.W = W; a.iters = iters; a.nblk = nblk;
  a.inv_n = 1.0f / ((float)B * 3.0f * (float)H * (float)W);
  a.param_stride = kMaxParam; a.gparam_stride = kMaxParam;
  int off = 0, bin = 0;
  for (int k = 0; k < kMaxChain; ++k) {
    a.ops[k] = k < s.n ? s.ops[k] : OP_IDENTITY;
    a.src[k] = k < s.n ? s.src[k] : 0;
    a.slot_off[k] = off;
    a.bin_off[k] = off;
    if (k < s.n) {
      off += (s.ops[k] == OP_COLOR ? 24 : s.ops[k] == OP_TONE ? 8 : s.ops[k] == OP_WHITE ? 0 : 1);
      bin = off;                                  // accumulator cells == raw-sum slots
    }
  }
  a.slot_off[kMaxChain] = off;
  a.bin_off[kMaxChain] = bin;
}
constexpr int kMaxChainSlots = kMaxChain * kMaxParam;   // 192 raw-sum slots per block row
constexpr int kMaxChainBins = kMaxChainSlots;

// ===================================================================== launch geometry (host)
struct Geometry {
  int vec;         // 4 when H*W % 4 == 0 (pointwise kernels use 16-byte accesses)
  int vec_tile;    // 4 when W % 4 == 0 (stencil kernels)
  int iters;       // pixel groups per thread (pointwise)
  int nblk_point;  // workgroups per sample (pointwise)
  int nblk_sharp;  // tiles per sample (stencil, LDS-tile kernels)
  int nblk_strip_fwd, nblk_strip_bwd;   // workgroups per sample of the LDS-free stencil kernels
  int nblk_max;    // stride of the per-sample partial-sum rows
};

inline Geometry geometry(int B, int H, int W, int forced_iters = 0) {
  Geometry g;
  const size_t hw = (size_t)H * W;
  g.vec = (hw % 4 == 0) ? 4 : 1;
  g.vec_tile = (W % 4 == 0) ? 4 : 1;
  const size_t groups = hw / g.vec;
  // aim for >= 16 workgroups per CU (4096 on 256 CUs) before giving a thread more work
  size_t iters = forced_iters > 0 ? (size_t)forced_iters : (groups * (size_t)B) / ((size_t)kThreads * 4096);
  if (iters < 1) iters = 1;
  if (iters > 8) iters = 8;
  g.iters = (int)iters;
  g.nblk_point = (int)((groups + (size_t)kThreads * g.iters - 1) / ((size_t)kThreads * g.iters));
  g.nblk_sharp = sharp_num_tiles(H, W);
  g.nblk_max = g.nblk_point > g.nblk_sharp ? g.nblk_point : g.nblk_sharp;
  // LDS-free stencil kernels (W % 4 == 0): workgroups of 4 waves x kFwdStripRows / kStripRows rows x 256 columns
  g.nblk_strip_fwd = ((H + 4 * kFwdStripRows - 1) / (4 * kFwdStripRows)) * ((W + 255) / 256);
  g.nblk_strip_bwd = ((H + 4 * kStripRows - 1) / (4 * kStripRows)) * strip_bwd_segments(W);
  if (g.vec_tile == 4) {
    if (g.nblk_strip_fwd > g.nblk_max) g.nblk_max = g.nblk_strip_fwd;
    if (g.nblk_strip_bwd > g.nblk_max) g.nblk_max = g.nblk_strip_bwd;
  }
  return g;
}

// ===================================================================== finalisation
// partials (B, nblk_max, kRedSlots) -> gparam (B, gparam_stride); one call per sample.
T2O_HD void finalize_sample(int op, const float* param_row, const float* sums, float* gparam_row) {
  finalize_param_grad(op, param_row, sums, gparam_row);
}

}  // namespace t2o
