// Device-side pieces shared by the ahead-of-time kernels (t2o_kernels.hip) and the kernels t2o_fused_sequence_prepare
// compiles at run time with hipRTC for an operator list that has no ahead-of-time instantiation: wave / workgroup
// reductions, the XCD-aware workgroup mapping, the LDS accumulator of the chain backward, and the BODIES of the
// compile-time-operator-list chain kernels (k_chain_fwd_static / k_chain_bwd_static are one-line wrappers around them).
// Device code only (DPP, readfirstlane, LDS): not part of the host emulation harness.
#pragma once
#include "t2o_block_programs.h"

namespace t2o {

// Sum over the 64 lanes, returned in every lane.  Pure VALU: an inclusive scan inside each 16-lane
// row by DPP row shifts (1, 2, 4, 8), row_bcast:15 / row_bcast:31 to fold the four rows, then a
// broadcast of lane 63 -- 6 DPP adds instead of 6 LDS-crossbar shuffles (ds_bpermute).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_shift_add<0x111, 0xF>(v);   // row_shr:1
  v = dpp_shift_add<0x112, 0xF>(v);   // row_shr:2
  v = dpp_shift_add<0x114, 0xF>(v);   // row_shr:4
  v = dpp_shift_add<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of every row holds the row sum
  v = dpp_shift_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_shift_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2).  Give every XCD a
// CONTIGUOUS range of logical work items so neighbouring tiles (shared halo rows) and
// consecutive chunks of one image meet in the same L2.  Bijective for any total.
__device__ __forceinline__ unsigned xcd_remap(unsigned lin, unsigned total) {
  const unsigned q = total / 8, r = total % 8, xcd = lin % 8, slot = lin / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
// (sample, block-in-sample) of this workgroup, forced into scalar registers: the integer
// division is lowered through the vector ALU, and without readfirstlane every per-sample
// parameter load would become a per-lane vector load holding 24+ VGPRs.
__device__ __forceinline__ void wg_coords(int per_sample, int& b, int& blk) {
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  b = __builtin_amdgcn_readfirstlane((int)(w / (unsigned)per_sample));
  blk = __builtin_amdgcn_readfirstlane((int)(w % (unsigned)per_sample));
}

__device__ __forceinline__ int nred_of(int op) { return op == OP_COLOR ? 24 : op == OP_TONE ? 8 : 1; }

// red[0..n) of every thread -> partials row of this block (fixed order => reproducible)
__device__ __forceinline__ void block_reduce_store(const float (&red)[kRedSlots], int n, float* dst) {
  __shared__ float sred[kThreads / 64][kRedSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kRedSlots; ++i) {
    if (i < n) {
      const float s = wave_sum(red[i]);
      if (lane == 0) sred[wave][i] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n)
    dst[threadIdx.x] = ((sred[0][threadIdx.x] + sred[1][threadIdx.x]) + sred[2][threadIdx.x]) + sred[3][threadIdx.x];
}

__device__ __forceinline__ void block_reduce_store1(float v, float* dst) {
  __shared__ float s1[kThreads / 64];
  const float s = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dst[0] = ((s1[0] + s1[1]) + s1[2]) + s1[3];
}


// raw parameter sums: quad (4-lane) DPP reduction, then one owner lane adds into its private LDS cell
struct LdsAcc {
  float* acc;
  // N raw sums of this thread: 4-lane (quad) DPP reduction, then the quad's first lane adds into
  // the quad's private LDS cells with plain read-add-write (sole owner: no atomics, fixed order).
  // Measured alternatives on MI355X (5-operator chain, 126 us with this scheme): one cell per THREAD
  // with ds_add_f32 -- 845 us (LDS float atomics cost ~190 cycles per wave instruction even with
  // conflict-free addresses); one cell per thread with read-add-write -- 221 us (4x the LDS cells to
  // zero and reduce per workgroup, 3 workgroups per CU instead of 5).
  template <int N>
  __device__ __forceinline__ void add_n(int slot0, float (&v)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      v[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[j]), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
      v[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[j]), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
      asm volatile("" : "+v"(v[j]));   // finish the sum HERE (one v_add_f32_dpp), not as a dpp move + an add inside the owner branch
    }
    if ((threadIdx.x & 3) == 0) {
      float* cell = acc + slot0 * kAccStride + (threadIdx.x >> 2);
      float old[N];
#pragma unroll
      for (int j = 0; j < N; ++j) old[j] = cell[j * kAccStride];
#pragma unroll
      for (int j = 0; j < N; ++j) cell[j * kAccStride] = old[j] + v[j];
    }
  }
};


template <int V, bool L1, class SEQ>
__device__ __forceinline__ void chain_fwd_static_body(const ChainArgs& a) {
  __shared__ float tab[kMaxChain * kTabStride];
  int b, blk;
  wg_coords(a.nblk, b, blk);
  if ((int)threadIdx.x < SEQ::K) chain_build_table(a, b, threadIdx.x, tab);
  __syncthreads();
  const float l1 = chain_fwd_thread_static<V, L1, SEQ>(a, b, blk, threadIdx.x, tab);
  if (L1) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk + blk);
}


// VAL: the launch also emits the loss partials (and the image when a.out is set) of a value-and-gradient call.  Kernels
// compiled at run time (t2o_jit.hip) take the default: their fused-L1 form is value-capable; the ahead-of-time kernels keep
// a plain L1 instantiation beside it (3 waves per SIMD at 164-166 VGPRs; the value outputs cost 4-12 more).
template <bool L1, class SEQ, bool SV_LDS, bool VAL = L1>
__device__ __forceinline__ void chain_bwd_static_body(const ChainArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // accumulator cells: NB rows of kAccStride [+ save area]
  __shared__ float tab[kMaxChain * kTabStride];
  __shared__ float bsum[kMaxChainBins];
  int b, blk;
  wg_coords(a.nblk, b, blk);
  const int S = a.slot_off[kMaxChain], NB = a.bin_off[kMaxChain];
  for (int i = threadIdx.x; i < NB * kAccStride; i += kThreads) lds[i] = 0.0f;
  if ((int)threadIdx.x < SEQ::K) chain_build_table(a, b, threadIdx.x, tab);
  __syncthreads();
  LdsAcc acc{lds};
  const float l1 = chain_bwd_thread_static<L1, SEQ, SV_LDS, LdsAcc, VAL>(a, b, blk, threadIdx.x, tab, lds + NB * kAccStride, acc);
  if constexpr (L1 && VAL) { if (a.loss_partials) block_reduce_store1(l1, a.loss_partials + (size_t)b * a.nblk + blk); }   // value-and-gradient calls
  __syncthreads();
  for (int s = threadIdx.x; s < NB; s += kThreads) {
    float sum = 0.0f;
    for (int q = 0; q < kThreads / 4; ++q) sum += lds[s * kAccStride + q];
    bsum[s] = sum;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += kThreads)
    a.partials[((size_t)b * a.nblk + blk) * S + s] = chain_slot_value(a, s, bsum);
}


}  // namespace t2o
