// Diagnostic (not part of the product): what does ISSUING an LDS-DMA piece cost the issuing wave?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/glds_issue tools/diag/glds_issue.hip
// Every wave issues NP pieces (1 KiB each) back to back, stamps s_memtime around the issue and again after vmcnt(0).
// Variants: 0 = M0 saved / set / restored around every piece (the form t2o_conv.hip used), 1 = M0 set per piece, not
// restored, 2 = the compiler builtin, 3 = M0 set once, pieces told apart by the instruction's immediate offset
// (which moves the global address too: the lane's pointer is pre-decremented), 4 = plain global_load_dwordx4 into
// registers.  Optionally waves 4..7 of every (512-thread) workgroup stream fp32 MFMAs instead (mfma = 1): the partner-wave
// situation of the convolution loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NP = 9;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

template <int V>
__global__ __launch_bounds__(512, 1) void k(const float* src, unsigned long long* out, int iters, int mfma, size_t stride_floats, float* sink) {
  __shared__ __attribute__((aligned(16))) float buf[8 * NP * 256];     // 72 KiB: 2 workgroups per CU like the conv kernel
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long issue = 0, land = 0;
  f32x16 acc[4] = {};
  float4 regs[NP];
  if (mfma && wave >= 4) {                                           // shares a SIMD with DMA wave (wave - 4)                                            // MFMA streamer
    float a = threadIdx.x * 0.001f, b = 1.0f;
    for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
      for (int r = 0; r < 12; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r & 3], 0, 0, 0);
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.0f) sink[0] = 1.0f;
    return;
  }
  const float* p = src + ((size_t)blockIdx.x * 8 + wave) * stride_floats + lane * 4;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (V == 0) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        unsigned keep; const float* s = p + i * 256;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(s), "s"(lds_addr(&buf[(wave * NP + i) * 256])) : "memory");
      }
    } else if constexpr (V == 1) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const float* s = p + i * 256;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(s), "s"(lds_addr(&buf[(wave * NP + i) * 256])) : "memory");
      }
    } else if constexpr (V == 2) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 256),
                                         (__attribute__((address_space(3))) void*)&buf[(wave * NP + i) * 256], 16, 0, 0);
    } else if constexpr (V == 3) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_addr(&buf[wave * NP * 256])) : "memory");
#define PIECE(I) { const float* s = p + (I) * 256 - (I) * 256; asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(s), "n"((I) * 1024) : "memory"); }
      PIECE(0) PIECE(1) PIECE(2) PIECE(3)
#undef PIECE
      // offsets beyond 12 bits are not encodable: the remaining pieces move M0 (no restore)
#pragma unroll
      for (int i = 4; i < NP; ++i) {
        const float* s = p + i * 256;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(s), "s"(lds_addr(&buf[(wave * NP + i) * 256])) : "memory");
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) regs[i] = *reinterpret_cast<const float4*>(p + i * 256);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if constexpr (V == 4) {
#pragma unroll
      for (int i = 0; i < NP; ++i) *reinterpret_cast<float4*>(&buf[(wave * NP + i) * 256 + lane * 4]) = regs[i];
    }
    issue += t1 - t0; land += t2 - t0;
    p += NP * 256;
    __builtin_amdgcn_s_sleep(8);
  }
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = issue; out[(blockIdx.x * 8 + wave) * 2 + 1] = land; }
  if (buf[threadIdx.x] == 123.456f) sink[1] = 1.0f;
}

int main() {
  const int grid = 256, iters = 64;
  const size_t stride = (size_t)iters * NP * 256;                      // floats per wave
  const size_t n = (size_t)grid * 8 * stride;
  float* src; hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4);
  unsigned long long* out; hipMalloc(&out, grid * 8 * 16);
  float* sink; hipMalloc(&sink, 64);
  const char* names[5] = {"asm, M0 saved+restored per piece", "asm, M0 set per piece", "builtin", "M0 once + imm offset (4), rest M0 per piece", "global_load_dwordx4 to registers"};
  const int order[10] = {2, 1, 0, 3, 4, 1, 2, 0, 2, 1};
  for (int mfma = 0; mfma < 2; ++mfma)
    for (int oi = 0; oi < 10; ++oi) {
      const int v = order[oi];
      hipMemset(out, 0, grid * 8 * 16);
      for (int rep = 0; rep < 3; ++rep) {
        switch (v) {
          case 0: k<0><<<grid, 512>>>(src, out, iters, mfma, stride, sink); break;
          case 1: k<1><<<grid, 512>>>(src, out, iters, mfma, stride, sink); break;
          case 2: k<2><<<grid, 512>>>(src, out, iters, mfma, stride, sink); break;
          case 3: k<3><<<grid, 512>>>(src, out, iters, mfma, stride, sink); break;
          default: k<4><<<grid, 512>>>(src, out, iters, mfma, stride, sink); break;
        }
        hipDeviceSynchronize();
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(grid * 8 * 2);
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> is, la;
      for (int w = 0; w < grid * 8; ++w) if (h[2 * w + 1]) { is.push_back((double)h[2 * w] / iters); la.push_back((double)h[2 * w + 1] / iters); }
      std::sort(is.begin(), is.end()); std::sort(la.begin(), la.end());
      printf("mfma partner %d  %-46s issue of %d pieces: median %6.0f cyc (%4.0f per piece)   issue -> all landed: %6.0f cyc   (%zu waves)\n",
             mfma, names[v], NP, is[is.size() / 2], is[is.size() / 2] / NP, la[la.size() / 2], is.size());
    }
  return 0;
}
