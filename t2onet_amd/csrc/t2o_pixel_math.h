// Per-pixel arithmetic of the T2ONet edit operators, forward and backward.
//
// Every function here is `__host__ __device__`: the HIP kernels in
// t2o_kernels.hip are the product path; tests/host_emul compiles the SAME
// functions with g++ so the arithmetic can be checked against the oracle on a
// machine without a GPU.  (That host build is a test harness, never a fallback:
// the Python package refuses to run without the HIP library.)
//
// Forward functions follow the reference's operation ORDER step for step, one
// fp32 rounding per step (build with -ffp-contract=off), so results track the
// reference's eager fp32 path to the last bit or two:
//   brightness  models/operators.py:277-283  (+ HSV spec: oracle/hsv_spec.py)
//   contrast    models/operators.py:240-245, utils/operator_utils.py:5-11
//   saturation  models/operators.py:473-479
//   color curve models/operators.py:607-616
//   tone curve  models/operators.py:571-585
//   sharpness   models/operators.py:351-358
//   white       models/operators.py:509-511
//   epilogue    models/operators.py:129-130 (mask blend + clamp)
// Backward functions are the closed-form derivatives of those formulas with
// PyTorch's conventions (clamp passes the gradient on [lo,hi] inclusive;
// max/min over channels route to one index).
#pragma once
#ifndef __HIPCC_RTC__      // (hipRTC: no system headers; the device math functions and size_t are built in)
#include <math.h>
#endif

#if defined(__HIPCC__)
#define T2O_HD __host__ __device__ __forceinline__
#define T2O_UNROLL _Pragma("unroll")
#else
#define T2O_HD static inline
#define T2O_UNROLL
#endif

// Register-pressure controls for the gfx950 build (no-ops on the host).  hipcc freely reorders
// and defers a thread's independent pixel-channels for ILP; for the curve operators that keeps
// the 8 segment terms of all 12 pixel-channels live (256 VGPRs, 1 wave per SIMD).  These three
// empty-asm helpers pin the order instead.
#if defined(__HIP_DEVICE_COMPILE__)
// v through an empty asm: the optimiser cannot see it is the same value, so cheap terms are
// RECOMPUTED where they are needed again rather than kept live.
__device__ __forceinline__ float t2o_opaque(float v) { asm volatile("" : "+v"(v)); return v; }
// v, made to depend on `dep`: work on v cannot start before dep has been computed.
__device__ __forceinline__ float t2o_chain(float v, float dep) { asm volatile("" : "+v"(v) : "v"(dep)); return v; }
// a / b by hardware reciprocal (1 ulp): for the closed-form DERIVATIVES only, whose tolerance is
// 1e-5; every forward formula keeps IEEE division for parity with the reference's rounding.
__device__ __forceinline__ float t2o_fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#define T2O_FDIV(a, b) t2o_fdiv(a, b)
#define T2O_OPAQUE(v) t2o_opaque(v)
#define T2O_CHAIN(v, dep) t2o_chain(v, dep)
// v must have been computed by this point (volatile asms keep their order): stops accumulations
// from being deferred with their inputs held in registers.
#define T2O_KEEP(v) asm volatile("" ::"v"(v))
// Compiler-level memory barrier: values loaded from LDS tables before it are not kept in
// registers across it (stops loop-invariant table reads being hoisted into 50+ VGPRs).
#define T2O_RELOAD_FENCE() asm volatile("" ::: "memory")
#else
#define T2O_FDIV(a, b) ((a) / (b))
#define T2O_OPAQUE(v) (v)
#define T2O_CHAIN(v, dep) (v)
#define T2O_KEEP(v) ((void)0)
#define T2O_RELOAD_FENCE() ((void)0)
#endif

namespace t2o {

enum : int {
  OP_IDENTITY = -1,
  OP_BRIGHTNESS = 0,
  OP_CONTRAST = 1,
  OP_SATURATION = 2,
  OP_COLOR = 3,
  OP_INPAINT = 4,   // needs the EdgeConnect network: not a per-pixel operator, unsupported
  OP_TONE = 5,
  OP_SHARPNESS = 6,
  OP_WHITE = 7,
  OP_DYNAMIC = -2   // kernels only: read the operator of each sample from op_id[b]
};

constexpr int kCurveSteps = 8;    // options/seq2seqGAN_base_options.py:87
constexpr int kMaxParam = 24;     // 3 x kCurveSteps, also the padded width in actor.py:166
constexpr float kHsvEps = 1e-6f;  // oracle/hsv_spec.py HSV_EPS
constexpr float kTwoPi = 6.283185307179586f;
constexpr float kTwoPiF = 6.283185307179586f;
constexpr float kPi = 3.141592653589793f;

T2O_HD int op_num_params(int op) {
  return op == OP_COLOR ? 24 : op == OP_TONE ? 8 : (op >= 0 && op <= 7) ? 1 : 0;
}

T2O_HD float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// a / d from the correctly rounded reciprocal r = RN(1/d) (Markstein): q = RN(a r),
// q' = RN(q + RN(a - d q) r) with the residual exact in an fma.  3 instructions instead of the
// ~10 of the IEEE sequence.  For d = 6 and d = float(2 pi) it was checked EXHAUSTIVELY against IEEE
// division over every float in [2^-30, 8) (both signs for 6): 0 mismatches, so the forward stays
// bit-identical to the reference's `h / 6.0` and `H / (2 pi)`.  For a general d it is correctly
// rounded unless d's mantissa is all ones (Markstein 1990).
T2O_HD float div_by(float a, float d, float r) {
  const float q = a * r;
  return fmaf(fmaf(-q, d, a), r, q);
}
// a / b for a general divisor: hardware reciprocal (1 ulp) + one Newton step, then the Markstein
// quotient above -- 6 instructions against ~10 (with three slow ones) for the IEEE sequence.  In 2e8
// random trials over the ranges used here (quotients in [-6, 6], divisors >= 1e-6, reciprocal
// perturbed by +-1 ulp) it returned the correctly rounded quotient every time; no scaling / fix-up
// for denormals or infinities (not needed: inputs are image values and their differences).
T2O_HD float recip_refined(float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float y0 = __builtin_amdgcn_rcpf(b);
#else
  const float y0 = 1.0f / b;
#endif
  return fmaf(fmaf(-b, y0, 1.0f), y0, y0);
}
T2O_HD float div_fast(float a, float b) { return div_by(a, b, recip_refined(b)); }
constexpr float kSixth = 1.0f / 6.0f;
constexpr float kInvTwoPi = 1.0f / kTwoPiF;
// torch.remainder(x, 6) for x in [0, 12)
T2O_HD float rem6(float x) { return x >= 6.0f ? x - 6.0f : x; }

struct Rgb {
  float c[3];
};

// r.c[idx] += v without dynamic register indexing (which would spill to scratch)
T2O_HD void add_at(Rgb& r, int idx, float v) {
  r.c[0] += (idx == 0) ? v : 0.0f;
  r.c[1] += (idx == 1) ? v : 0.0f;
  r.c[2] += (idx == 2) ? v : 0.0f;
}

// ---------------------------------------------------------------- HSV (oracle/hsv_spec.py)
struct Hsv {
  float h, s, v;
};

T2O_HD Hsv rgb_to_hsv(float r, float g, float b) {
  const float maxc = fmaxf(r, fmaxf(g, b));
  const float minc = fminf(r, fminf(g, b));
  const float delta = maxc - minc;
  Hsv o;
  o.v = maxc;
  o.s = div_fast(delta, maxc + kHsvEps);
  const float ds = (delta == 0.0f) ? 1.0f : delta;
  const float rc = maxc - r, gc = maxc - g, bc = maxc - b;
  // all three candidates, then two selects: no divergent branches.  The FIRST channel that attains
  // the maximum decides (torch.max index convention): r == maxc, else g == maxc, else b.
  // (T2O_OPAQUE pins each candidate where it is: otherwise hipcc sinks them back into a branch tree)
  const float hn0 = T2O_OPAQUE(bc - gc), hn1 = T2O_OPAQUE((rc - bc) + 2.0f * ds), hn2 = T2O_OPAQUE((gc - rc) + 4.0f * ds);
  const float hn = (r == maxc) ? hn0 : ((g == maxc) ? hn1 : hn2);
  float h = div_fast(hn, ds);
  h = div_by(h, 6.0f, kSixth);
  h = h - truncf(h);            // fmod(h, 1)
  if (h < 0.0f) h += 1.0f;      // torch.remainder sign fix-up
  o.h = kTwoPi * h;
  return o;
}

T2O_HD Rgb hsv_to_rgb(float H, float s, float v) {
  const float h = div_by(H, kTwoPi, kInvTwoPi);
  const float h6 = h * 6.0f;
  const float hi = rem6(floorf(h6));
  const float f = rem6(h6) - hi;
  // out_c = v (1 - w_c s) with w_c = 0 (the "v" role), 1 ("p"), f ("q") or 1 - f ("t"): the same products
  // as the reference's p, q, t (1*s and 0*s are exact), chosen per channel by its phase
  // j = sector - 2c in -4..5 -> w = {1, 1, 1-f, 0 | 0, f, 1, 1, 1-f, 0}[j + 4].  That table is the sum of two
  // clamped ramps, min((j-1) + f, (4-j) + (1-f)) and (-2-j) + (1-f), each clamped to [0,1]: where a ramp is
  // strictly inside (0,1) its integer part is 0, so it returns f or 1-f exactly; elsewhere it saturates
  // with a margin (f in [0,1)), and at most one of the two is non-zero.  No compares, no selects.
  const float fq = T2O_OPAQUE(f), omf = T2O_OPAQUE(1.0f - f);
  Rgb o;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    // with j = hi - 2c folded into the constants (small integers: exact either way)
    float w = clamp01(fminf((hi - (float)(2 * c + 1)) + fq, ((float)(2 * c + 4) - hi) + omf));
    if (c > 0) w = w + clamp01(((float)(2 * c - 2) - hi) + omf);
    o.c[c] = v * (1.0f - w * s);
  }
  return o;
}

// ---------------------------------------------------------------- curve parameters
// One sample's curve, loaded once per block (wave-uniform -> scalar registers).
struct Curve {
  float k[3][kCurveSteps];  // tone: the three rows are the same curve
  float sum[3];             // sum_i k_i + 1e-10
  float rsum[3];            // RN(1 / sum)
  float scale[3];           // d out / d total = 8 / sum
};

T2O_HD void curve_load(Curve& cv, const float* p, bool color) {
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    const float* row = color ? p + c * kCurveSteps : p;
    float s = 0.0f;
    T2O_UNROLL
  for (int i = 0; i < kCurveSteps; ++i) {
      cv.k[c][i] = row[i];
      s = s + row[i];
    }
    s = s + 1e-10f;
    cv.sum[c] = s;
    cv.rsum[c] = 1.0f / s;
    cv.scale[c] = cv.rsum[c] * (float)kCurveSteps;   // torch: n / tensor == reciprocal(tensor) * n
  }
}

T2O_HD float curve_total(const float k[kCurveSteps], float x) {
  float total = 0.0f;
  T2O_UNROLL
  for (int i = 0; i < kCurveSteps; ++i) {
    const float t = fminf(fmaxf(x - (float)i / kCurveSteps, 0.0f), 1.0f / kCurveSteps);
    total = total + t * k[i];
  }
  return total;
}

// ---------------------------------------------------------------- forward, per pixel
// `process()` of the pointwise operators.  p = this sample's parameter row.
T2O_HD Rgb brightness_fwd(const Rgb& x, float p) {
  const Hsv a = rgb_to_hsv(x.c[0], x.c[1], x.c[2]);
  const float v2 = clamp01(a.v * (1.0f + p));
  return hsv_to_rgb(a.h, a.s, v2);
}

T2O_HD Rgb saturation_fwd(const Rgb& x, float p) {
  const Hsv a = rgb_to_hsv(x.c[0], x.c[1], x.c[2]);
  const float s2 = clamp01(a.s * (1.0f + p));
  return hsv_to_rgb(a.h, s2, a.v);
}

T2O_HD float luminance(const Rgb& x) { return (0.27f * x.c[0] + 0.67f * x.c[1]) + 0.06f * x.c[2]; }

T2O_HD Rgb contrast_fwd(const Rgb& x, float p) {
  const float L = fminf(fmaxf(luminance(x), 0.0f), 1.0f);
  const float cl = (-cosf(kPi * L)) * 0.5f + 0.5f;
  const float Le = L + 1e-6f;
  const float rLe = recip_refined(Le);     // one refined reciprocal, three Markstein quotients
  const float om = 1.0f - p;
  Rgb o;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    const float ci = div_by(x.c[c], Le, rLe) * cl;
    o.c[c] = om * x.c[c] + p * ci;
  }
  return o;
}

// one channel of the tone (shared curve) / color (per-channel curve) operator
T2O_HD float curve_fwd_1(const Curve& cv, bool color, int c, float x) {
  return color ? curve_total(cv.k[c], x) * cv.scale[c]
               : div_by(curve_total(cv.k[0], x) * (float)kCurveSteps, cv.sum[0], cv.rsum[0]);
}

T2O_HD Rgb tone_fwd(const Rgb& x, const Curve& cv) {
  Rgb o;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c)
    o.c[c] = div_by(curve_total(cv.k[0], x.c[c]) * (float)kCurveSteps, cv.sum[0], cv.rsum[0]);
  return o;
}

T2O_HD Rgb color_fwd(const Rgb& x, const Curve& cv) {
  Rgb o;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) o.c[c] = curve_total(cv.k[c], x.c[c]) * cv.scale[c];
  return o;
}

// 3x3 Laplacian-style kernel [[0,-1,0],[-1,4,-1],[0,-1,0]], row-major accumulation
T2O_HD float sharp_delta(float c, float up, float left, float right, float down) {
  return ((((-up) - left) + 4.0f * c) - right) - down;
}

// Operator.execute epilogue: blend with the mask, clamp.  z is the pre-clamp value.
T2O_HD float blend(float o, float x, float m) { return o * m + x * (1.0f - m); }

// Pointwise dispatcher (everything except sharpness / identity).
T2O_HD Rgb pointwise_fwd(int op, const Rgb& x, const float* p, const Curve& cv) {
  switch (op) {
    case OP_BRIGHTNESS: return brightness_fwd(x, p[0]);
    case OP_CONTRAST:   return contrast_fwd(x, p[0]);
    case OP_SATURATION: return saturation_fwd(x, p[0]);
    case OP_COLOR:      return color_fwd(x, cv);
    case OP_TONE:       return tone_fwd(x, cv);
    case OP_WHITE: { Rgb o; o.c[0] = o.c[1] = o.c[2] = 1.0f; return o; }
    default:            return x;
  }
}

// ---------------------------------------------------------------- backward, per pixel
// g  = gradient w.r.t. the operator's process() output at this pixel
// gx = gradient w.r.t. the input pixel THROUGH process() (the caller adds the mask path)
// red[] accumulates this sample's parameter-gradient raw sums:
//   1-parameter ops: red[0] += d loss / d p
//   tone:            red[i]      += g * t_i(x)            (i < 8, all channels)
//   color:           red[8c + i] += g_c * t_i(x_c)
// (curve raw sums are turned into gradients by curve_param_grad below.)
T2O_HD void argmaxmin(const Rgb& x, int& amax, int& amin, float& vmax, float& vmin) {
  vmax = x.c[0]; amax = 0;
  if (x.c[1] > vmax) { vmax = x.c[1]; amax = 1; }
  if (x.c[2] > vmax) { vmax = x.c[2]; amax = 2; }
  vmin = x.c[0]; amin = 0;
  if (x.c[1] < vmin) { vmin = x.c[1]; amin = 1; }
  if (x.c[2] < vmin) { vmin = x.c[2]; amin = 2; }
}

// Exact autograd of the HSV round trip, used where channels tie (r == g, g == b, r == b): there
// brightness / saturation are not differentiable and PyTorch's answer is the derivative of the
// BRANCH the forward took (first-index max/min, the hue sector floor() picked).  The closed forms
// below equal it everywhere else, at half the arithmetic.  sat = false: V = clamp(v (1+P)), S = s;
// sat = true: V = v, S = clamp(s (1+P)).  out_c = V (1 - w_c S), w_c in {0, 1, f, 1-f} by sector.
T2O_HD Rgb hsv_literal_bwd(bool sat, const Rgb& x, float P, const Rgb& g, float* red) {
  const float r = x.c[0], gr = x.c[1], b = x.c[2];
  float M = r; int a = 0;
  if (gr > M) { M = gr; a = 1; }
  if (b > M) { M = b; a = 2; }
  float m = r; int im = 0;
  if (gr < m) { m = gr; im = 1; }
  if (b < m) { m = b; im = 2; }
  const float delta = M - m;
  const float ve = M + kHsvEps;
  const float s = div_fast(delta, ve);
  const float ds = (delta == 0.0f) ? 1.0f : delta;
  const float rc = M - r, gc = M - gr, bc = M - b;
  const float hn = a == 0 ? (bc - gc) : a == 1 ? ((rc - bc) + 2.0f * ds) : ((gc - rc) + 4.0f * ds);
  float h = div_fast(hn, ds);
  h = div_by(h, 6.0f, kSixth);
  h = h - truncf(h);
  if (h < 0.0f) h += 1.0f;
  const float H = kTwoPi * h;
  const float h6 = div_by(H, kTwoPi, kInvTwoPi) * 6.0f;
  const float hi = rem6(floorf(h6));
  const float f = rem6(h6) - hi;
  const int k = (int)hi;
  // forward clamp of the scaled quantity
  const float tq = (sat ? s : M) * (1.0f + P);
  const bool in = tq >= 0.0f && tq <= 1.0f;
  const float V = sat ? M : clamp01(tq);
  const float S = sat ? clamp01(tq) : s;
  // role of each output channel in sector k:  0 = v, 1 = p, 2 = q (w = f), 3 = t (w = 1 - f)
  //            sector:   0  1  2  3  4  5
  const int roleR = k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 1 : k == 3 ? 1 : k == 4 ? 3 : 0;
  const int roleG = k == 0 ? 3 : k == 1 ? 0 : k == 2 ? 0 : k == 3 ? 2 : k == 4 ? 1 : 1;
  const int roleB = k == 0 ? 1 : k == 1 ? 1 : k == 2 ? 3 : k == 3 ? 0 : k == 4 ? 0 : 2;
  const int role[3] = {roleR, roleG, roleB};
  float GV = 0.0f, GS = 0.0f, Gf = 0.0f;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) {
    const float w = role[c] == 0 ? 0.0f : role[c] == 1 ? 1.0f : role[c] == 2 ? f : 1.0f - f;
    const float sg = role[c] == 2 ? 1.0f : role[c] == 3 ? -1.0f : 0.0f;
    GV += g.c[c] * (1.0f - w * S);
    GS += g.c[c] * w;
    Gf += g.c[c] * sg;
  }
  GS *= -V;
  Gf *= -V * S;
  float Gv, Gs;
  if (sat) { Gv = GV; Gs = in ? GS * (1.0f + P) : 0.0f; red[0] += in ? GS * s : 0.0f; }
  else     { Gv = in ? GV * (1.0f + P) : 0.0f; Gs = GS; red[0] += in ? GV * M : 0.0f; }
  const float rve = T2O_FDIV(1.0f, ve), rds = T2O_FDIV(1.0f, ds);   // derivative only: hardware reciprocals
  float Gd = Gs * rve;                        // d / d delta through s
  Gv += -Gs * delta * (rve * rve);
  const float Ghn = Gf * rds;
  if (delta != 0.0f) Gd += -Gf * hn * (rds * rds) + (a == 1 ? 2.0f * Ghn : a == 2 ? 4.0f * Ghn : 0.0f);
  Rgb gx;
  gx.c[0] = gx.c[1] = gx.c[2] = 0.0f;
  // hn = (g - b), (b - r) + 2 ds, (r - g) + 4 ds for argmax = r, g, b
  add_at(gx, a == 0 ? 1 : a == 1 ? 2 : 0, Ghn);
  add_at(gx, a == 0 ? 2 : a == 1 ? 0 : 1, -Ghn);
  add_at(gx, a, Gd + Gv);
  add_at(gx, im, -Gd);
  return gx;
}

T2O_HD bool has_channel_tie(const Rgb& x) { return x.c[0] == x.c[1] || x.c[1] == x.c[2] || x.c[0] == x.c[2]; }

// out_c = v' (x_c + eps) / (v + eps),  v' = clamp(v (1 + p), 0, 1)
T2O_HD Rgb brightness_bwd(const Rgb& x, float p, const Rgb& g, float* red) {
  if (has_channel_tie(x)) return hsv_literal_bwd(false, x, p, g, red);
  int amax, amin; float v, mn;
  argmaxmin(x, amax, amin, v, mn);
  const float ve = v + kHsvEps;
  const float t = v * (1.0f + p);
  // three clamp regimes (t < 0: v' = 0;  t > 1: v' = 1;  else v' = t) blended with 0/1 masks: no branches
  const float m_hi = t > 1.0f ? 1.0f : 0.0f;
  const float m_in = (t >= 0.0f && t <= 1.0f) ? 1.0f : 0.0f;
  const float rve = T2O_FDIV(1.0f, ve > 0.5f * kHsvEps ? ve : 1.0f);   // inputs below 0 only occur in the t < 0 regime
  const float a = m_in * (t * rve) + m_hi * rve;
  const float da_dv = m_in * ((1.0f + p) * kHsvEps * rve * rve) - m_hi * (rve * rve);
  const float da_dp = m_in * (v * rve);
  const float S = g.c[0] * (x.c[0] + kHsvEps) + g.c[1] * (x.c[1] + kHsvEps) + g.c[2] * (x.c[2] + kHsvEps);
  Rgb gx;
  red[0] += S * da_dp;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) gx.c[c] = a * g.c[c];
  add_at(gx, amax, S * da_dv);
  return gx;
}

// out_c = v (1 - s' (v - x_c) / delta),  s' = clamp(s (1 + p), 0, 1),  s = delta / (v + eps)
T2O_HD Rgb saturation_bwd(const Rgb& x, float p, const Rgb& g, float* red) {
  if (has_channel_tie(x)) return hsv_literal_bwd(true, x, p, g, red);
  int amax, amin; float v, mn;
  argmaxmin(x, amax, amin, v, mn);
  const float ve = v + kHsvEps;
  const float delta = v - mn;                  // > 0: no two channels are equal here
  const float rve = T2O_FDIV(1.0f, ve > 0.5f * kHsvEps ? ve : 1.0f);
  const float rd = T2O_FDIV(1.0f, delta);
  const float s = div_fast(delta, ve);        // the clamp decision uses the forward's exact s
  const float t = s * (1.0f + p);
  const float u0 = v - x.c[0], u1 = v - x.c[1], u2 = v - x.c[2];
  const float G = g.c[0] + g.c[1] + g.c[2];
  const float GU = g.c[0] * u0 + g.c[1] * u1 + g.c[2] * u2;
  // regimes: t < 0 (s' = 0, every channel becomes v), t > 1 (s' = 1), else s' = s (1 + p); 0/1 masks, no branches
  const float m_lo = t < 0.0f ? 1.0f : 0.0f;
  const float m_hi = t > 1.0f ? 1.0f : 0.0f;
  const float m_in = (t >= 0.0f && t <= 1.0f) ? 1.0f : 0.0f;
  const float vd = v * rd;
  const float b = v * (1.0f + p) * rve;
  const float db = (1.0f + p) * kHsvEps * rve * rve;
  const float coef = m_hi * vd + m_in * b;
  Rgb gx;
  T2O_UNROLL
  for (int c = 0; c < 3; ++c) gx.c[c] = coef * g.c[c];
  add_at(gx, amax, m_lo * G + m_hi * (G * (1.0f - vd) + GU * (vd - 1.0f) * rd) + m_in * (G * (1.0f - b) - GU * db));
  add_at(gx, amin, m_hi * (-GU * vd * rd));
  red[0] += m_in * (-GU * (v * rve));
  return gx;
}

// out_c = x_c ((1 - p) + p q(L)),  q = cl(L) / (L + 1e-6)
// sin(pi L), cos(pi L) for L in [0,1] to 2e-7 absolute: with r = L - 1/2, sin(pi L) = cos(pi r) and
// cos(pi L) = -sin(pi r), |pi r| <= pi/2, Taylor polynomials in r^2 (truncation < 6e-8).  13 multiply-adds
// instead of libm's general sincosf (argument reduction for any magnitude, ~50 instructions).
T2O_HD void sincospi_unit(float L, float& sn, float& cs) {
  const float r = L - 0.5f, r2 = r * r;
  float ps = -0.007370430945714348f;
  ps = fmaf(ps, r2, 0.08214588661112819f);
  ps = fmaf(ps, r2, -0.5992645293207919f);
  ps = fmaf(ps, r2, 2.550164039877345f);
  ps = fmaf(ps, r2, -5.167712780049969f);
  ps = fmaf(ps, r2, 3.141592653589793f);
  float pc = 0.001929574309403922f;
  pc = fmaf(pc, r2, -0.02580689139001405f);
  pc = fmaf(pc, r2, 0.23533063035889312f);
  pc = fmaf(pc, r2, -1.3352627688545893f);
  pc = fmaf(pc, r2, 4.058712126416768f);
  pc = fmaf(pc, r2, -4.934802200544679f);
  pc = fmaf(pc, r2, 1.0f);
  sn = pc;
  cs = -(ps * r);
}

T2O_HD Rgb contrast_bwd(const Rgb& x, float p, const Rgb& g, float* red) {
  const float lum = luminance(x);
  const float L = fminf(fmaxf(lum, 0.0f), 1.0f);
  // torch.min(torch.max(lum, 0), 1): elementwise max/min split the gradient 1/2 - 1/2 at a tie
  const float inside = (lum > 0.0f && lum < 1.0f) ? 1.0f : ((lum == 0.0f || lum == 1.0f) ? 0.5f : 0.0f);
  float sn, cs;
  sincospi_unit(L, sn, cs);                    // derivative only (1e-5 bar); the forward keeps libm's cosf
  const float cl = (-cs) * 0.5f + 0.5f;
  const float Le = L + 1e-6f;
  const float rLe = T2O_FDIV(1.0f, Le);
  const float q = cl * rLe;
  const float dq = (0.5f * kPi * sn * Le - cl) * rLe * rLe;
  const float S = g.c[0] * x.c[0] + g.c[1] * x.c[1] + g.c[2] * x.c[2];
  const float k0 = (1.0f - p) + p * q;
  const float k1 = inside * p * S * dq;
  Rgb gx;
  gx.c[0] = g.c[0] * k0 + k1 * 0.27f;
  gx.c[1] = g.c[1] * k0 + k1 * 0.67f;
  gx.c[2] = g.c[2] * k0 + k1 * 0.06f;
  red[0] += S * (q - 1.0f);
  return gx;
}

// red[j] += g * clamp(x - j/8, 0, 1/8) for the 8 curve segments: the per-pixel part of a curve's
// parameter gradient.  On gfx950 two segments per instruction: a packed fp32 add with the [0,1]
// output clamp gives w_j = clamp(8x - j, 0, 1) (= 8 * the term above exactly: 8x is a power-of-two
// scaling and 8x - j is exact wherever it is positive), then one packed multiply-add with g/8.
// first = true: red[j] = ... instead of += (the caller's first contribution: saves zeroing the sums).
T2O_HD void curve_bins_accumulate(float x, float g, float* red, bool first = false) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float t = x * (float)kCurveSteps, g8 = g * (1.0f / kCurveSteps);
  const f32x2 tt = {t, t}, gg = {g8, g8};
  T2O_UNROLL
  for (int j = 0; j < kCurveSteps; j += 2) {
    const f32x2 nj = {-(float)j, -(float)(j + 1)};     // wave-uniform constants: a scalar register pair
    f32x2 w;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(w) : "v"(tt), "s"(nj));
    f32x2 acc;
    if (first) {
      acc = gg * w;
    } else {
      acc = f32x2{red[j], red[j + 1]};
      acc = __builtin_elementwise_fma(gg, w, acc);
    }
    red[j] = acc.x;
    red[j + 1] = acc.y;
  }
#else
  for (int j = 0; j < kCurveSteps; ++j) {
    const float t = g * fminf(fmaxf(x - (float)j / kCurveSteps, 0.0f), 1.0f / kCurveSteps);
    red[j] = first ? t : red[j] + t;
  }
#endif
}

// Can the final clamp(0,1) of operator `op` be active for input pixel x?  For brightness and
// saturation the HSV round trip returns v' * [0,1] factors (resp. v * [0,1] factors): with the
// input inside [0,1] the output provably is too, in fp32 as well (every factor is a rounded
// quotient/product of numbers <= 1), so the backward pass needs no forward recompute there.
T2O_HD bool clamp_can_act(int op, const Rgb& x) {
  const float mn = fminf(x.c[0], fminf(x.c[1], x.c[2]));
  const float mx = fmaxf(x.c[0], fmaxf(x.c[1], x.c[2]));
  if (op == OP_BRIGHTNESS) return !(mn >= 0.0f);
  if (op == OP_SATURATION) return !(mn >= 0.0f && mx <= 1.0f);
  return true;
}

// (the curve operators' backward goes through the lookup table: curve_lut_bwd_1 in t2o_block_programs.h)
T2O_HD Rgb pointwise_bwd(int op, const Rgb& x, const float* p, const Curve& cv, const Rgb& g, float* red) {
  (void)cv;
  switch (op) {
    case OP_BRIGHTNESS: return brightness_bwd(x, p[0], g, red);
    case OP_CONTRAST:   return contrast_bwd(x, p[0], g, red);
    case OP_SATURATION: return saturation_bwd(x, p[0], g, red);
    case OP_WHITE: { Rgb z; z.c[0] = z.c[1] = z.c[2] = 0.0f; return z; }
    default:            return g;
  }
}

// Raw per-sample sums -> parameter gradients.  out = total * 8 / sum, total = sum_j k_j t_j:
//   d out / d k_i = scale t_i - (scale / sum) total   =>   gk_i = scale A_i - (scale / sum) sum_j k_j A_j
T2O_HD void curve_param_grad(const float* k, const float* A, float* gk) {
  float s = 0.0f, dot = 0.0f;
  T2O_UNROLL
  for (int i = 0; i < kCurveSteps; ++i) { s = s + k[i]; dot += k[i] * A[i]; }
  s = s + 1e-10f;
  const float scale = (float)kCurveSteps / s;
  T2O_UNROLL
  for (int i = 0; i < kCurveSteps; ++i) gk[i] = scale * A[i] - (scale / s) * dot;
}

T2O_HD void finalize_param_grad(int op, const float* param, const float* red, float* gparam) {
  if (op == OP_TONE) {
    curve_param_grad(param, red, gparam);
  } else if (op == OP_COLOR) {
    T2O_UNROLL
  for (int c = 0; c < 3; ++c)
      curve_param_grad(param + c * kCurveSteps, red + c * kCurveSteps, gparam + c * kCurveSteps);
  } else if (op == OP_WHITE || op == OP_IDENTITY) {
    const int n = op_num_params(op);
    for (int i = 0; i < n; ++i) gparam[i] = 0.0f;
  } else {
    gparam[0] = red[0];
  }
}

}  // namespace t2o
