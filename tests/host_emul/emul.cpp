// Host emulation of the HIP kernels' block programs -- TEST HARNESS ONLY.
//
// Compiles t2onet_amd/csrc/t2o_pixel_math.h + t2o_block_programs.h with g++ and runs the
// SAME per-thread phase functions the gfx950 kernels run, workgroup by workgroup and thread by
// thread, on host memory.  It lets the CPU test suite check the operators' arithmetic, tile
// indexing, halos, bounds and reductions against the oracle without a GPU.  It is not shipped,
// not importable from the package, and never used as a fallback.
#include <math.h>
#include <string.h>

#include <vector>

#include "../../t2onet_amd/csrc/t2o_block_programs.h"

using namespace t2o;

namespace {

OpArgs make_args(int op, const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
                 int mask_ch, int B, int H, int W, const Geometry& g) {
  OpArgs a;
  memset(&a, 0, sizeof(a));
  a.img = img; a.param = param; a.mask = mask; a.op_id = op_id;
  a.op = op; a.param_stride = param_stride; a.mask_ch = mask_ch; a.B = B; a.H = H; a.W = W;
  a.iters = g.iters; a.nblk_max = g.nblk_max;
  a.inv_n = 1.0f / ((float)B * 3.0f * (float)H * (float)W);
  return a;
}

template <int V, bool M, bool L>
float point_fwd_block(const OpArgs& a, int op, int b, int blk) {
  float s = 0.0f;
  for (int tid = 0; tid < kThreads; ++tid) s += pointwise_fwd_thread<V, M, L>(a, op, b, blk, tid);
  return s;
}
template <int V, bool M, bool L>
void point_bwd_block(const OpArgs& a, int op, int b, int blk, float* sums) {
  float tab[kTabStride] = {0.0f};               // the kernel builds this in LDS once per workgroup
  if (op == OP_COLOR || op == OP_TONE) curve_table_build(a.param + (size_t)b * a.param_stride, op == OP_COLOR, tab);
  for (int tid = 0; tid < kThreads; ++tid) {
    float red[kRedSlots];
    for (int i = 0; i < kRedSlots; ++i) red[i] = 0.0f;
    pointwise_bwd_thread<V, M, L>(a, op, b, blk, tid, red, tab);
    for (int i = 0; i < kRedSlots; ++i) sums[i] += red[i];
  }
}

#define DISPATCH3(FN, V, M, L, ...)                                                   \
  ((V) == 4 ? ((M) ? ((L) ? FN<4, true, true>(__VA_ARGS__) : FN<4, true, false>(__VA_ARGS__))      \
                   : ((L) ? FN<4, false, true>(__VA_ARGS__) : FN<4, false, false>(__VA_ARGS__)))   \
            : ((M) ? ((L) ? FN<1, true, true>(__VA_ARGS__) : FN<1, true, false>(__VA_ARGS__))      \
                   : ((L) ? FN<1, false, true>(__VA_ARGS__) : FN<1, false, false>(__VA_ARGS__))))

struct HostAcc {
  float* sums;
  template <int N>
  void add_n(int slot0, float (&v)[N]) { for (int j = 0; j < N; ++j) sums[slot0 + j] += v[j]; }
};

// the operator lists with a compile-time backward (mirrors t2o_kernels.hip: SeqCfg2 / SeqCfg5)
using EmuSeq2 = StaticChain<OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_COLOR, OP_TONE>;
using EmuSeq5 = StaticChain<OP_TONE, OP_COLOR, OP_TONE, OP_COLOR, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION>;
template <class SEQ>
static bool emu_chain_is(const ChainArgs& a) {
  if (a.K != SEQ::K) return false;
  for (int k = 0; k < SEQ::K; ++k)
    if (a.ops[k] != SEQ::ops[k]) return false;
  return true;
}


}  // namespace

extern "C" {

// forward of one operator (op_id == NULL) or per-sample operators (op == OP_DYNAMIC);
// target/loss optional (fused L1).  Returns 0.
int emul_fwd(int op, const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
             int mask_ch, const float* target, float* out, float* loss, int B, int H, int W, int forced_iters) {
  const Geometry g = geometry(B, H, W, forced_iters);
  OpArgs a = make_args(op, op_id, img, param, param_stride, mask, mask_ch, B, H, W, g);
  a.out = out; a.target = target;
  double total = 0.0;
  for (int b = 0; b < B; ++b) {
    const int ob = op == OP_DYNAMIC ? op_id[b] : op;
    if (ob != OP_SHARPNESS) {
      for (int blk = 0; blk < g.nblk_point; ++blk)
        total += DISPATCH3(point_fwd_block, g.vec, mask_ch != 0, target != nullptr, a, ob, b, blk);
    } else {
      std::vector<float> lds(sharp_fwd_lds_floats());
      for (int tile = 0; tile < g.nblk_sharp; ++tile) {
        for (int tid = 0; tid < kThreads; ++tid) {
          if (g.vec_tile == 4) sharp_fwd_phase_load<4>(a, b, tile, tid, lds.data());
          else sharp_fwd_phase_load<1>(a, b, tile, tid, lds.data());
        }
        for (int tid = 0; tid < kThreads; ++tid)
          total += g.vec_tile == 4 ? sharp_fwd_phase_compute<4>(a, b, tile, tid, lds.data())
                                   : sharp_fwd_phase_compute<1>(a, b, tile, tid, lds.data());
      }
    }
  }
  if (target && loss) loss[0] = (float)(total * a.inv_n);
  return 0;
}

int emul_bwd(int op, const int* op_id, const float* img, const float* param, int param_stride, const float* mask,
             int mask_ch, const float* gout, const float* target, const float* gloss, float* gimg, float* gparam,
             int gparam_stride, int B, int H, int W, int forced_iters) {
  const Geometry g = geometry(B, H, W, forced_iters);
  OpArgs a = make_args(op, op_id, img, param, param_stride, mask, mask_ch, B, H, W, g);
  a.gout = gout; a.target = target; a.gloss = gloss; a.gimg = gimg;
  for (int b = 0; b < B; ++b) {
    const int ob = op == OP_DYNAMIC ? op_id[b] : op;
    float sums[kRedSlots];
    for (int i = 0; i < kRedSlots; ++i) sums[i] = 0.0f;
    if (ob != OP_SHARPNESS) {
      for (int blk = 0; blk < g.nblk_point; ++blk)
        DISPATCH3(point_bwd_block, g.vec, mask_ch != 0, target != nullptr, a, ob, b, blk, sums);
    } else {
      std::vector<float> lds(sharp_bwd_lds_floats(mask_ch));
      for (int tile = 0; tile < g.nblk_sharp; ++tile) {
        for (int tid = 0; tid < kThreads; ++tid) {
          if (g.vec_tile == 4) sharp_bwd_phase_load<4>(a, b, tile, tid, lds.data());
          else sharp_bwd_phase_load<1>(a, b, tile, tid, lds.data());
        }
        for (int tid = 0; tid < kThreads; ++tid) {
          if (g.vec_tile == 4) sharp_bwd_phase_dz<4>(a, b, tile, tid, lds.data(), sums[0]);
          else sharp_bwd_phase_dz<1>(a, b, tile, tid, lds.data(), sums[0]);
        }
        for (int tid = 0; tid < kThreads; ++tid) {
          if (g.vec_tile == 4) sharp_bwd_phase_out<4>(a, b, tile, tid, lds.data());
          else sharp_bwd_phase_out<1>(a, b, tile, tid, lds.data());
        }
      }
    }
    if (gparam && ob != OP_IDENTITY)
      finalize_param_grad(ob, param + (size_t)b * param_stride, sums, gparam + (size_t)b * gparam_stride);
  }
  return 0;
}



// fused sequence, forward: chain segments + sharpness segments, boundaries in seg_bufs
int emul_fused_fwd(const int* ops, int K, const float* img, const float* params, const float* target, float* out,
                   float* loss, float* seg_bufs, int B, int H, int W, int forced_iters) {
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  if (ns < 0) return 2;
  int vec, iters, nblk;
  chain_geometry(B, H, W, forced_iters, vec, iters, nblk);
  const size_t img_floats = (size_t)B * 3 * H * W;
  const float* cur = img;
  for (int s = 0; s < ns; ++s) {
    float* dst = (s == ns - 1) ? out : seg_bufs + (size_t)s * img_floats;
    const bool last = (s == ns - 1) && target;
    if (seg[s].sharp) {
      const int k = seg[s].first;
      emul_fwd(OP_SHARPNESS, nullptr, cur, params + (size_t)k * B * kMaxParam, kMaxParam, nullptr, 0,
               last ? target : nullptr, dst, loss, B, H, W, forced_iters);
    } else {
      ChainArgs a;
      memset(&a, 0, sizeof(a));
      chain_fill(a, seg[s], B, H, W, iters, nblk);
      a.img = cur; a.params = params; a.out = dst; a.target = last ? target : nullptr;
      double total = 0.0;
      for (int b = 0; b < B; ++b) {
        std::vector<float> tab(kMaxChain * kTabStride, 0.0f);
        for (int k = 0; k < a.K; ++k) chain_build_table(a, b, k, tab.data());
        for (int blk = 0; blk < nblk; ++blk)
          for (int tid = 0; tid < kThreads; ++tid) {
            if (vec == 2) total += last ? chain_fwd_thread<2, true>(a, b, blk, tid, tab.data()) : chain_fwd_thread<2, false>(a, b, blk, tid, tab.data());
            else total += last ? chain_fwd_thread<1, true>(a, b, blk, tid, tab.data()) : chain_fwd_thread<1, false>(a, b, blk, tid, tab.data());
          }
      }
      if (last) loss[0] = (float)(total * a.inv_n);
    }
    cur = dst;
  }
  return 0;
}

// use_static != 0: segments whose operator list has a compile-time instantiation run chain_bwd_thread_static (one
// pixel per thread-iteration, parameter sums kept across the thread's pixels), as the device dispatch does
int emul_fused_bwd(const int* ops, int K, const float* img, const float* params, const float* target,
                   const float* gloss, float* gimg, float* gparams, const float* seg_bufs, float* gbuf, int B, int H,
                   int W, int forced_iters, int use_static, float* value_out, float* value_loss) {
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  if (ns < 0) return 2;
  int vec, iters, nblk;
  chain_geometry(B, H, W, forced_iters, vec, iters, nblk, use_static ? 1 : 0);
  const size_t img_floats = (size_t)B * 3 * H * W;
  memset(gparams, 0, sizeof(float) * (size_t)K * B * kMaxParam);
  const float* gcur = nullptr;
  for (int s = ns - 1; s >= 0; --s) {
    const float* in = s == 0 ? img : seg_bufs + (size_t)(s - 1) * img_floats;
    float* gnext = s == 0 ? gimg : gbuf + (size_t)(s & 1) * img_floats;
    const bool last = s == ns - 1;
    if (seg[s].sharp) {
      const int k = seg[s].first;
      emul_bwd(OP_SHARPNESS, nullptr, in, params + (size_t)k * B * kMaxParam, kMaxParam, nullptr, 0,
               last ? nullptr : gcur, last ? target : nullptr, gloss, gnext, gparams + (size_t)k * B * kMaxParam,
               kMaxParam, B, H, W, forced_iters);
    } else {
      ChainArgs a;
      memset(&a, 0, sizeof(a));
      chain_fill(a, seg[s], B, H, W, iters, nblk);
      a.img = in; a.params = params; a.gimg = gnext;
      if (last) { a.target = target; a.gloss = gloss; a.out = value_out; } else { a.gout = gcur; }
      double l1_total = 0.0;                      // value outputs of the last segment's backward (t2o_fused_sequence_l1_value_grad)
      for (int b = 0; b < B; ++b) {
        std::vector<float> tab(kMaxChain * kTabStride, 0.0f);
        for (int k = 0; k < a.K; ++k) chain_build_table(a, b, k, tab.data());
        float sums[kMaxChainSlots], bins[kMaxChainBins];
        for (int i = 0; i < kMaxChainSlots; ++i) sums[i] = 0.0f;
        for (int i = 0; i < kMaxChainBins; ++i) bins[i] = 0.0f;
        HostAcc acc{bins};
        std::vector<float> sv(chain_save_floats<2>(kMaxChain));
        const bool st2 = use_static && vec == 1 && emu_chain_is<EmuSeq2>(a), st5 = use_static && vec == 1 && emu_chain_is<EmuSeq5>(a);
        for (int blk = 0; blk < nblk; ++blk)
          for (int tid = 0; tid < kThreads; ++tid) {
            if (st2) { if (last) l1_total += chain_bwd_thread_static<true, EmuSeq2, false>(a, b, blk, tid, tab.data(), sv.data(), acc); else chain_bwd_thread_static<false, EmuSeq2, false>(a, b, blk, tid, tab.data(), sv.data(), acc); }
            else if (st5) { if (last) l1_total += chain_bwd_thread_static<true, EmuSeq5, false>(a, b, blk, tid, tab.data(), sv.data(), acc); else chain_bwd_thread_static<false, EmuSeq5, false>(a, b, blk, tid, tab.data(), sv.data(), acc); }
            else if (vec == 2) { if (last) l1_total += chain_bwd_thread<2, true>(a, b, blk, tid, tab.data(), sv.data(), acc); else chain_bwd_thread<2, false>(a, b, blk, tid, tab.data(), sv.data(), acc); }
            else { if (last) l1_total += chain_bwd_thread<1, true>(a, b, blk, tid, tab.data(), sv.data(), acc); else chain_bwd_thread<1, false>(a, b, blk, tid, tab.data(), sv.data(), acc); }
          }
        for (int sl = 0; sl < a.slot_off[kMaxChain]; ++sl) sums[sl] = chain_slot_value(a, sl, bins);
        for (int k = 0; k < a.K; ++k) {
          float* grow = gparams + ((size_t)a.src[k] * B + b) * kMaxParam;
          finalize_param_grad(a.ops[k], params + ((size_t)a.src[k] * B + b) * kMaxParam, sums + a.slot_off[k], grow);
        }
      }
      if (last && value_loss) value_loss[0] = (float)(l1_total * a.inv_n);
    }
    gcur = gnext;
  }
  return 0;
}

int emul_ssim(const float* a, const float* b, float* out, int B, int C, int H, int W) {
  SsimArgs s;
  memset(&s, 0, sizeof(s));
  s.a = a; s.b = b;
  ssim_window(s.g);
  s.B = B; s.C = C; s.H = H; s.W = W;
  s.tiles_x = (W + kSsimTile - 1) / kSsimTile;
  s.tiles = s.tiles_x * ((H + kSsimTile - 1) / kSsimTile);
  std::vector<float> lds(ssim_lds_floats());
  for (int bb = 0; bb < B; ++bb) {
    double total = 0.0;
    for (int c = 0; c < C; ++c)
      for (int tile = 0; tile < s.tiles; ++tile) {
        const int plane = bb * C + c;
        for (int tid = 0; tid < kThreads; ++tid) ssim_phase_load(s, plane, tile, tid, lds.data());
        for (int tid = 0; tid < kThreads; ++tid) ssim_phase_rows(s, tid, lds.data());
        for (int tid = 0; tid < kThreads; ++tid) total += ssim_phase_cols(s, tile, tid, lds.data());
      }
    out[bb] = (float)(total / ((double)C * H * W));
  }
  return 0;
}

int emul_ssim_bwd(const float* a, const float* b, const float* gout, float* ga, float* gb, int B, int C, int H, int W) {
  SsimBwdArgs s;
  memset(&s, 0, sizeof(s));
  s.a = a; s.b = b; s.gout = gout; s.ga = ga; s.gb = gb;
  ssim_window(s.g);
  s.B = B; s.C = C; s.H = H; s.W = W;
  s.tiles_x = (W + kSsimTile - 1) / kSsimTile;
  s.tiles = s.tiles_x * ((H + kSsimTile - 1) / kSsimTile);
  s.inv_n = 1.0f / ((float)C * (float)H * (float)W);
  std::vector<float> lds(ssim_bwd_lds_floats());
  for (int plane = 0; plane < B * C; ++plane)
    for (int tile = 0; tile < s.tiles; ++tile) {
      for (size_t i = 0; i < lds.size(); ++i) lds[i] = NAN;       // (a phase reading what no phase wrote shows up)
      for (int tid = 0; tid < kThreads; ++tid) ssim_bwd_phase_load(s, plane, tile, tid, lds.data());
      for (int tid = 0; tid < kThreads; ++tid) ssim_bwd_phase_rows(s, tid, lds.data());
      for (int tid = 0; tid < kThreads; ++tid) ssim_bwd_phase_deriv(s, plane, tile, tid, lds.data());
      for (int tid = 0; tid < kThreads; ++tid) ssim_bwd_phase_drows(s, tid, lds.data());
      for (int tid = 0; tid < kThreads; ++tid) ssim_bwd_phase_out(s, plane, tile, tid, lds.data());
    }
  return 0;
}

int emul_fused_buffers(const int* ops, int K) {
  Segment seg[64];
  const int ns = plan_segments(ops, K, seg, 64);
  return ns < 0 ? -1 : ns - 1;
}

}  // extern "C"
