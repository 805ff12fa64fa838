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
  for (int tid = 0; tid < kThreads; ++tid) {
    float red[kRedSlots];
    for (int i = 0; i < kRedSlots; ++i) red[i] = 0.0f;
    pointwise_bwd_thread<V, M, L>(a, op, b, blk, tid, red);
    for (int i = 0; i < kRedSlots; ++i) sums[i] += red[i];
  }
}

#define DISPATCH3(FN, V, M, L, ...)                                                   \
  ((V) == 4 ? ((M) ? ((L) ? FN<4, true, true>(__VA_ARGS__) : FN<4, true, false>(__VA_ARGS__))      \
                   : ((L) ? FN<4, false, true>(__VA_ARGS__) : FN<4, false, false>(__VA_ARGS__)))   \
            : ((M) ? ((L) ? FN<1, true, true>(__VA_ARGS__) : FN<1, true, false>(__VA_ARGS__))      \
                   : ((L) ? FN<1, false, true>(__VA_ARGS__) : FN<1, false, false>(__VA_ARGS__))))

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
        for (int tid = 0; tid < kThreads; ++tid) sharp_bwd_phase_dz(a, b, tile, tid, lds.data());
        for (int tid = 0; tid < kThreads; ++tid) {
          if (g.vec_tile == 4) sharp_bwd_phase_out<4>(a, b, tile, tid, lds.data(), sums[0]);
          else sharp_bwd_phase_out<1>(a, b, tile, tid, lds.data(), sums[0]);
        }
      }
    }
    if (gparam && ob != OP_IDENTITY)
      finalize_param_grad(ob, param + (size_t)b * param_stride, sums, gparam + (size_t)b * gparam_stride);
  }
  return 0;
}

}  // extern "C"
