// Registry of run-time compiled chain kernels (t2o_jit.hip), consulted by the chain dispatch in t2o_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "t2o_block_programs.h"

namespace t2o {

struct JitChain {
  hipFunction_t fwd[2][2];     // [pixels per thread-iteration - 1][fused L1]
  hipFunction_t bwd[2];        // [fused L1]
};

bool jit_lookup(const int* ops, int K, JitChain* out);    // true: *out = the kernels of this exact operator list
int jit_prepare(const int* ops, int K);                    // compile (hipRTC) / load from the disk cache if absent
int jit_launch(hipFunction_t f, const ChainArgs& a, unsigned grid, size_t lds_bytes, hipStream_t st);
int jit_count();

}  // namespace t2o
