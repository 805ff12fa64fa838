// t2o_gemm.hip -- one general fp32 matrix-core GEMM for the dense products of the request encoder and the decoder tape that
// no specialised kernel of this library takes (models/lang_encoder.py:91-102: the LSTM's input projection over all time steps
// and its weight / input gradients; models/action_decoder.py:52-63: the weight gradients of vis_linear, both LSTM cells,
// attention.linear_out and out_linear over all decoder steps of a train step; models/actor_resnet.py:107 fc):
//
//     C (M,N) = beta * C + op(A) op(B),   op(A) (M,K), op(B) (K,N),   beta in {0, 1}
//
// with either operand stored contraction-major ([K][M] / [K][N]: "dy^T x" weight gradients sum over the ROWS of both operands)
// or contraction-contiguous ([M][K] / [N][K]: nn.Linear's x W^T), ANY M, N, K (11 output classes, 300 embedding columns, 812
// decoder inputs, request lengths) and leading dimensions (column slices of a larger matrix).  Rounding is fixed by the
// shape alone: one workgroup owns a 64 x 64 tile of C and walks K front to back in steps of 16, a lane's sum is the matrix
// instruction's (v_mfma_f32_32x32x2_f32: exact fp32 multiply-adds, k-pairs in order) -- no split-K, no atomics, the same bits on
// every box and run.  That is what it is for: the library GEMMs it replaces pick a kernel (and a reduction order) per machine, and
// the episode step's gradient norms moved with it (tests/test_gpu_actor.py).
//
//   256 threads = 4 waves (2 x 2), each a 32 x 32 block of the tile.  LDS: As[k][m], Bs[k][n] (16 x 64 floats each, two
//   buffers); row k's columns are stored at c ^ ((k & 1) << 5), so the MFMA operand read -- lanes 0-31 row 2s, lanes 32-63 row
//   2s + 1, 32 consecutive columns each -- touches every bank once.  Global loads of chunk i + 1 travel in registers under the
//   MFMAs of chunk i; one barrier per chunk.  A contraction-major operand is read as rows of 64 consecutive floats (coalesced
//   16-byte loads), a contraction-contiguous one as 16-float row pieces (one 64-byte segment per tile row).
#include <hip/hip_runtime.h>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmAnyArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int beta;              // 0: C = A B, 1: C += A B
  int tiles_m, tiles_n;
  int vec_a, vec_b;      // 16-byte loads allowed (base and leading dimension aligned)
};

// four consecutive floats at p, the first `valid` of them inside the matrix (the rest read as 0)
__device__ __forceinline__ float4 load4(const float* p, int valid, int vec) {
  if (valid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

// kAK / kBK: operand stored contraction-major ([K][M] resp. [K][N]); else contraction-contiguous ([M][K] resp. [N][K])
template <bool kAK, bool kBK>
__global__ __launch_bounds__(256) void k_gemm_any(GemmAnyArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][16][64];
  __shared__ __attribute__((aligned(16))) float Bs[2][16][64];
  const int bid = blockIdx.x;
  const int tm = bid / a.tiles_n, tn = bid - tm * a.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, ln = lane & 31, lh = lane >> 5;

  // this thread's piece of a chunk: contraction-major -> (row k = tid / 16, columns 4 (tid % 16) ..); contraction-contiguous ->
  // (tile row tid % 64, k = 4 (tid / 64) ..)
  const int ka = kAK ? (tid >> 4) : ((tid >> 6) << 2), ca = kAK ? ((tid & 15) << 2) : (tid & 63);
  const int kb = kBK ? (tid >> 4) : ((tid >> 6) << 2), cb = kBK ? ((tid & 15) << 2) : (tid & 63);
  float4 ra, rb;
  auto gload = [&](int k0) {
    if constexpr (kAK) {
      const int k = k0 + ka, valid = k < a.K ? a.M - (m0 + ca) : 0;
      ra = load4(a.A + (size_t)k * a.lda + m0 + ca, valid, a.vec_a);
    } else {
      const int m = m0 + ca, valid = m < a.M ? a.K - (k0 + ka) : 0;
      ra = load4(a.A + (size_t)m * a.lda + k0 + ka, valid, a.vec_a);
    }
    if constexpr (kBK) {
      const int k = k0 + kb, valid = k < a.K ? a.N - (n0 + cb) : 0;
      rb = load4(a.B + (size_t)k * a.ldb + n0 + cb, valid, a.vec_b);
    } else {
      const int n = n0 + cb, valid = n < a.N ? a.K - (k0 + kb) : 0;
      rb = load4(a.B + (size_t)n * a.ldb + k0 + kb, valid, a.vec_b);
    }
  };
  auto sstore = [&](int buf) {
    if constexpr (kAK) {
      *reinterpret_cast<float4*>(&As[buf][ka][ca ^ ((ka & 1) << 5)]) = ra;
    } else {
      As[buf][ka][ca] = ra.x; As[buf][ka + 1][ca ^ 32] = ra.y; As[buf][ka + 2][ca] = ra.z; As[buf][ka + 3][ca ^ 32] = ra.w;
    }
    if constexpr (kBK) {
      *reinterpret_cast<float4*>(&Bs[buf][kb][cb ^ ((kb & 1) << 5)]) = rb;
    } else {
      Bs[buf][kb][cb] = rb.x; Bs[buf][kb + 1][cb ^ 32] = rb.y; Bs[buf][kb + 2][cb] = rb.z; Bs[buf][kb + 3][cb ^ 32] = rb.w;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int chunks = (a.K + 15) >> 4;
  gload(0);
  sstore(0);
  __syncthreads();
  const int acol = (wm * 32 + ln) ^ (lh << 5), bcol = (wn * 32 + ln) ^ (lh << 5);     // (row 2s + lh: odd rows are swizzled)
  for (int c = 0; c < chunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < chunks) gload((c + 1) << 4);
#pragma unroll
    for (int s = 0; s < 8; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + lh][acol], Bs[buf][2 * s + lh][bcol], acc, 0, 0, 0);
    if (c + 1 < chunks) sstore(buf ^ 1);
    __syncthreads();
  }
  // C/D layout: column = lane % 32, row = (r % 4) + 8 (r / 4) + 4 (lane / 32)
  const int n = n0 + wn * 32 + ln;
  if (n < a.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < a.M) {
        float* dst = a.C + (size_t)m * a.ldc + n;
        *dst = a.beta ? *dst + acc[r] : acc[r];
      }
    }
  }
}

// out[n] = beta * out[n] + sum over the rows r (in order) of X[r][n]: 64 columns x 4 row classes per workgroup -- thread (g, n)
// adds rows g, g + 4, ... front to back, class sums are added 0, 1, 2, 3
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ X, float* __restrict__ out, int R, int N, int ldx, int beta) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float s = 0.0f;
  if (n < N) {
    int r = g;
    for (; r + 12 < R; r += 16) {
      const float v0 = X[(size_t)r * ldx + n], v1 = X[(size_t)(r + 4) * ldx + n], v2 = X[(size_t)(r + 8) * ldx + n], v3 = X[(size_t)(r + 12) * ldx + n];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; r < R; r += 4) s += X[(size_t)r * ldx + n];
  }
  part[g][c] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    const float t = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
    out[n] = beta ? out[n] + t : t;
  }
}

}  // namespace

extern "C" {

int t2o_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kmajor, int b_kmajor,
             int accumulate, void* stream) {
  if (!A || !B || !C) return set_error(T2O_EINVAL, "gemm: null pointer");
  if (M <= 0 || N <= 0 || K <= 0) return set_error(T2O_EINVAL, "gemm: M, N, K must be positive");
  if (lda < (a_kmajor ? M : K) || ldb < (b_kmajor ? N : K) || ldc < N) return set_error(T2O_EINVAL, "gemm: leading dimension smaller than the row");
  const long long tiles_m = (M + 63) / 64, tiles_n = (N + 63) / 64;
  if (tiles_m * tiles_n > 0x7fffffffLL) return set_error(T2O_EUNSUPPORTED, "gemm: more than 2^31 tiles");
  GemmAnyArgs a;
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.beta = accumulate ? 1 : 0;
  a.tiles_m = (int)tiles_m; a.tiles_n = (int)tiles_n;
  a.vec_a = ((reinterpret_cast<size_t>(A) & 15) == 0 && lda % 4 == 0) ? 1 : 0;
  a.vec_b = ((reinterpret_cast<size_t>(B) & 15) == 0 && ldb % 4 == 0) ? 1 : 0;
  const unsigned grid = (unsigned)(tiles_m * tiles_n);
  hipStream_t st = (hipStream_t)stream;
  if (a_kmajor && b_kmajor) k_gemm_any<true, true><<<grid, 256, 0, st>>>(a);
  else if (a_kmajor) k_gemm_any<true, false><<<grid, 256, 0, st>>>(a);
  else if (b_kmajor) k_gemm_any<false, true><<<grid, 256, 0, st>>>(a);
  else k_gemm_any<false, false><<<grid, 256, 0, st>>>(a);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "gemm launch failed");
}

int t2o_colsum(const float* X, float* out, int R, int N, int ldx, int accumulate, void* stream) {
  if (!X || !out) return set_error(T2O_EINVAL, "colsum: null pointer");
  if (R <= 0 || N <= 0 || ldx < N) return set_error(T2O_EINVAL, "colsum: R, N positive, ldx >= N");
  k_colsum<<<(unsigned)((N + 63) / 64), 256, 0, (hipStream_t)stream>>>(X, out, R, N, ldx, accumulate ? 1 : 0);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "colsum launch failed");
}

}  // extern "C"
