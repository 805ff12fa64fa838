// t2o_optim.hip -- Adam over ONE flat fp32 buffer (experiments/t2onet/train_seq2seqL1.py:169: torch.optim.Adam with
// the default betas / eps, no weight decay, no amsgrad; the reference steps 199 separate tensors).  The trainer keeps
// every parameter, gradient and moment as a view of four flat buffers (the gradient one is also the single
// all-reduce payload), so the whole optimiser step is one streaming pass: 16 bytes read + 12 written per parameter.
#include <hip/hip_runtime.h>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

// torch.optim.Adam's single-tensor arithmetic, per element:
//   m = m + (g - m) * (1 - b1);  v = v * b2 + g * g * (1 - b2);  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_c, float b2, float om1, float om2,
                                      float rsq_bc2, float eps) {
  m = m + (g - m) * om1;
  v = v * b2 + (g * g) * om2;
  const float denom = sqrtf(v) * rsq_bc2 + eps;
  p = p - lr_c * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(float* p, const float* g, float* m, float* v, size_t n, float lr_c, float b2,
                                              float om1, float om2, float rsq_bc2, float eps) {
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    adam1(pp.x, gg.x, mm.x, vv.x, lr_c, b2, om1, om2, rsq_bc2, eps);
    adam1(pp.y, gg.y, mm.y, vv.y, lr_c, b2, om1, om2, rsq_bc2, eps);
    adam1(pp.z, gg.z, mm.z, vv.z, lr_c, b2, om1, om2, rsq_bc2, eps);
    adam1(pp.w, gg.w, mm.w, vv.w, lr_c, b2, om1, om2, rsq_bc2, eps);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    adam1(p[i], g[i], m[i], v[i], lr_c, b2, om1, om2, rsq_bc2, eps);
  }
}

}  // namespace

extern "C" {

int t2o_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, int step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return set_error(T2O_EINVAL, "adam_step: null pointer");
  if (n == 0 || step < 1) return set_error(T2O_EINVAL, "adam_step: n and step must be positive");
  if (((size_t)param | (size_t)grad | (size_t)exp_avg | (size_t)exp_avg_sq) & 15) return set_error(T2O_EINVAL, "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_c = (float)((double)lr / bc1), rsq_bc2 = (float)(1.0 / sqrt(bc2));
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  k_adam<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, lr_c, beta2, 1.0f - beta1,
                                                            1.0f - beta2, rsq_bc2, eps);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "adam_step launch failed");
}

}  // extern "C"
