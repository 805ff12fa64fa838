// t2o_optim.hip -- Adam over ONE flat fp32 buffer (experiments/t2onet/train_seq2seqL1.py:169: torch.optim.Adam with
// the default betas / eps, no weight decay, no amsgrad; the reference steps 199 separate tensors).  The trainer keeps
// every parameter, gradient and moment as a view of four flat buffers (the gradient one is also the single
// all-reduce payload), so the whole optimiser step is one streaming pass: 16 bytes read + 12 written per parameter.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

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

// ---------------------------------------------------------------------------------------------------------
// hipGraph hygiene.  On this stack (ROCm 7.2, gfx950) a MEMSET node of a captured graph was seen to run out of
// order with the kernel nodes around it on replay (t2o_conv.hip: a workspace cleared late; and, inside the library's
// atomic weight-gradient kernels, an output cleared late -> garbage gradients on some boxes, depending on which
// solver the library's find step picked).  t2o_graph_memsets_to_kernels rewrites a captured, not yet instantiated
// graph: every memset node becomes a kernel node (k_graph_fill) with the same dependencies and dependents.
__global__ __launch_bounds__(256) void k_graph_fill(unsigned char* dst, unsigned value, unsigned elem, size_t width, size_t height, size_t pitch) {
  const size_t n = width * height;                    // elements
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned char* p = dst + (i / width) * pitch + (i % width) * elem;
    if (elem == 4) *reinterpret_cast<unsigned*>(p) = value;
    else if (elem == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)value;
    else *p = (unsigned char)value;
  }
}

extern "C" {

int t2o_graph_memsets_to_kernels(void* graph_, int* replaced) {
  hipGraph_t graph = (hipGraph_t)graph_;
  if (replaced) *replaced = 0;
  if (!graph) return set_error(T2O_EINVAL, "graph_memsets_to_kernels: null graph");
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphGetNodes failed");
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphGetNodes failed");
  int count = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphNodeGetType failed");
    if (type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    if (hipGraphMemsetNodeGetParams(nodes[i], &mp) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphMemsetNodeGetParams failed");
    size_t nd = 0, nt = 0;
    hipGraphNodeGetDependencies(nodes[i], nullptr, &nd);
    hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nt);
    std::vector<hipGraphNode_t> deps(nd), outs(nt);
    if (nd && hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphNodeGetDependencies failed");
    if (nt && hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &nt) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphNodeGetDependentNodes failed");
    unsigned char* dst = (unsigned char*)mp.dst;
    unsigned value = mp.value, elem = mp.elementSize ? mp.elementSize : 1;
    size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    void* args[6] = {&dst, &value, &elem, &width, &height, &pitch};
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof(kp));
    size_t blocks = (width * height + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    kp.func = (void*)k_graph_fill;
    kp.gridDim = dim3((unsigned)blocks); kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
    hipGraphNode_t fill;
    if (hipGraphAddKernelNode(&fill, graph, deps.data(), nd, &kp) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphAddKernelNode failed");
    for (size_t k = 0; k < nt; ++k)
      if (hipGraphAddDependencies(graph, &fill, &outs[k], 1) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphAddDependencies failed");
    if (hipGraphDestroyNode(nodes[i]) != hipSuccess) return set_error(T2O_ELAUNCH, "hipGraphDestroyNode failed");
    ++count;
  }
  if (replaced) *replaced = count;
  return T2O_OK;
}

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
