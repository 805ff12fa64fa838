// Diagnostic (not part of the product): what does one LDS-DMA piece cost INSIDE a wave's own MFMA stream?
// 512-thread workgroups, one per CU, all 8 waves (two per SIMD) run the same loop: 12 independent
// v_mfma_f32_32x32x2_f32 per iteration, and -- depending on the variant -- one 1 KiB piece in the middle:
//   0 nothing   1 global_load_lds_dwordx4 (saddr form, M0 saved/restored)   2 the same without touching M0 (set once)
//   3 plain global_load_dwordx4 into registers (never waited for until the end of the iteration block)
//   4 like 1, then s_waitcnt vmcnt(0) every 16th iteration (the conv loop's barrier)
// plus optionally 8 ds_read_b32 per iteration (lds = 1), like the conv loop's fragment reads.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/glds_in_mfma tools/diag/glds_in_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

template <int V, int LDS>
__global__ __launch_bounds__(512, 1) void k(const float* src, unsigned long long* out, int iters, size_t stride_floats, float* sink) {
  __shared__ __attribute__((aligned(16))) float buf[8 * 16 * 256];     // 128 KiB: 16 pieces per wave, reused
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[12] = {};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  const char* base = (const char*)(src + ((size_t)blockIdx.x * 8 + wave) * stride_floats);
  const unsigned voff = lane * 16;
  const unsigned rbase = lds_addr(&buf[wave * 16 * 256 + lane]);
  if (V == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_addr(&buf[wave * 16 * 256])) : "memory");
  float4 dump = {0, 0, 0, 0};
  float r[8] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      a += r[0] * 1e-30f; b += r[5] * 1e-30f;
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[q]) : "v"(rbase), "n"(q * 512));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 6; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned dst = lds_addr(&buf[(wave * 16 + (it & 15)) * 256]);
    if (V == 1 || V == 4) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    } else if (V == 2) {
      asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base) : "memory");
    } else if (V == 3) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dump) : "v"(voff), "s"(base) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 6; m < 12; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (V == 4 && (it & 15) == 14) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    base += 1024;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = dump.x + r[3];
  for (int m = 0; m < 12; ++m) s += acc[m][0];
  if (s == 123.0f) sink[0] = buf[threadIdx.x];
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  const int grid = 256, iters = 2048;
  const size_t stride = (size_t)iters * 256;
  float* src; hipMalloc(&src, (size_t)grid * 8 * stride * 4); hipMemset(src, 0, (size_t)grid * 8 * stride * 4);
  unsigned long long* out; hipMalloc(&out, grid * 8 * 8);
  float* sink; hipMalloc(&sink, 64);
  const char* names[5] = {"MFMAs only", "+ LDS-DMA piece (M0 saved/set/restored)", "+ LDS-DMA piece (M0 set once)", "+ global_load_dwordx4 to registers", "+ LDS-DMA piece, vmcnt(0)+barrier every 16"};
  for (int lds = 0; lds < 2; ++lds)
    for (int v = 0; v < 5; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
#define L(V_) case V_: if (lds) k<V_, 1><<<grid, 512>>>(src, out, iters, stride, sink); else k<V_, 0><<<grid, 512>>>(src, out, iters, stride, sink); break;
        switch (v) { L(0) L(1) L(2) L(3) L(4) }
#undef L
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h(grid * 8);
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> t; for (auto c : h) t.push_back((double)c / iters);
      std::sort(t.begin(), t.end());
      printf("fragment reads %d  %-46s cycles per iteration (12 MFMAs = 768 pipe cycles; x2 waves per SIMD = 1536): median %6.0f  min %6.0f  max %6.0f\n",
             lds, names[v], t[t.size() / 2], t.front(), t.back());
    }
  return 0;
}
