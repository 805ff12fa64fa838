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
// shape alone: one workgroup owns a 64 x 64 tile of C, its 1, 2 or 4 contraction groups (by K) walk their chunks of 32 front to back and are
// added in group order; a lane's sum is the matrix instruction's (v_mfma_f32_32x32x2_f32: exact fp32 multiply-adds, k-pairs in order) -- no split-K, no atomics, the same bits on
// every box and run.  That is what it is for: the library GEMMs it replaces pick a kernel (and a reduction order) per machine, and
// the episode step's gradient norms moved with it (tests/test_gpu_actor.py).
//
//   A contraction group = 256 threads = 4 waves (2 x 2), each a 32 x 32 block of the tile (1, 2 or 4 groups per workgroup, see kKG).  LDS: As[k][m], Bs[k][n] (32 x 64 floats each, two
//   buffers); row k's columns are stored at c ^ swz(k) (bit 5 = k's parity), so the MFMA operand read -- lanes 0-31 row 2s, lanes 32-63 row
//   2s + 1, 32 consecutive columns each -- touches every bank once.  Global loads of chunks i + 1 and i + 2 travel in registers under the
//   MFMAs of chunk i; one barrier per chunk.  A contraction-major operand is read as rows of 64 consecutive floats (coalesced
//   16-byte loads), a contraction-contiguous one as 32-float row pieces (one 128-byte line per tile row).
#include <hip/hip_runtime.h>

#include <type_traits>

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
// kKG: contraction groups.  These products are small (85-416 tiles for 256 CUs) and long in K: one 4-wave workgroup per CU
// walking K alone is a chain of memory round trips (first version: 106-123 us for K = 2048 where the matrix work is 6 us).  So a
// workgroup is kKG groups of 4 waves; group g takes the chunks g, g + kKG, ... of the contraction (its own LDS buffers, its own
// accumulators), all groups step together, and at the end the groups' tiles are added in LDS in group order 0, 1, ... -- a fixed
// order again, chosen by the shape alone (kKG is a function of K).
constexpr int kChunk = 32;             // contraction steps per LDS stage and group

template <bool kAK, bool kBK, int kKG>
__global__ __launch_bounds__(256 * kKG) void k_gemm_any(GemmAnyArgs a) {
  __shared__ __attribute__((aligned(16))) float As[kKG][2][kChunk][64];
  __shared__ __attribute__((aligned(16))) float Bs[kKG][2][kChunk][64];
  const int bid = blockIdx.x;
  const int tm = bid / a.tiles_n, tn = bid - tm * a.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const int kg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);          // contraction group of this wave
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, ln = lane & 31, lh = lane >> 5;

  // this thread's two pieces of a chunk: contraction-major -> (rows k = tid / 16 + 16 j, columns 4 (tid % 16) ..);
  // contraction-contiguous -> (tile rows tid / 8 + 32 j, k = 4 (tid % 8) ..)
  // (contraction-contiguous: 8 neighbouring lanes read the 128 bytes of ONE row piece -- with a lane per row every 16-byte request
  // was a line of its own and the texture addresser the bottleneck: 62-82 us for K = 2048, round 6 -- rows tid / 8 + 32 j)
  const int ka = kAK ? (tid >> 4) : ((tid & 7) << 2), ca = kAK ? ((tid & 15) << 2) : (tid >> 3);
  const int kb = kBK ? (tid >> 4) : ((tid & 7) << 2), cb = kBK ? ((tid & 15) << 2) : (tid >> 3);
  // a tile whose 64 rows / columns all exist and whose operands allow 16-byte loads reads its full chunks without a single
  // predicate (uniform per workgroup): the loads of two chunks then stay in flight across the MFMAs of a third
  const bool full_a = a.vec_a && m0 + 64 <= a.M, full_b = a.vec_b && n0 + 64 <= a.N;
  const float* const pa = kAK ? a.A + (size_t)ka * a.lda + m0 + ca : a.A + (size_t)(m0 + ca) * a.lda + ka;
  const float* const pb = kBK ? a.B + (size_t)kb * a.ldb + n0 + cb : a.B + (size_t)(n0 + cb) * a.ldb + kb;
  const size_t ja = kAK ? (size_t)16 * a.lda : (size_t)32 * a.lda, jb = kBK ? (size_t)16 * a.ldb : (size_t)32 * a.ldb;      // piece j = 1
  float4 ra[2][2], rb[2][2];
  auto gload = [&](auto sc, int k0) {
    constexpr int S = decltype(sc)::value;
    const bool whole = k0 + kChunk <= a.K;
    const size_t oa = kAK ? (size_t)k0 * a.lda : (size_t)k0, ob = kBK ? (size_t)k0 * a.ldb : (size_t)k0;
    if (whole && full_a) {
      ra[S][0] = *reinterpret_cast<const float4*>(pa + oa);
      ra[S][1] = *reinterpret_cast<const float4*>(pa + oa + ja);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (kAK) {
          const int k = k0 + ka + 16 * j, valid = k < a.K ? a.M - (m0 + ca) : 0;
          ra[S][j] = load4(pa + oa + j * ja, valid, a.vec_a);
        } else {
          const int kk = k0 + ka, valid = m0 + ca + 32 * j < a.M ? a.K - kk : 0;
          ra[S][j] = load4(pa + oa + j * ja, valid, a.vec_a);
        }
      }
    }
    if (whole && full_b) {
      rb[S][0] = *reinterpret_cast<const float4*>(pb + ob);
      rb[S][1] = *reinterpret_cast<const float4*>(pb + ob + jb);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (kBK) {
          const int k = k0 + kb + 16 * j, valid = k < a.K ? a.N - (n0 + cb) : 0;
          rb[S][j] = load4(pb + ob + j * jb, valid, a.vec_b);
        } else {
          const int kk = k0 + kb, valid = n0 + cb + 32 * j < a.N ? a.K - kk : 0;
          rb[S][j] = load4(pb + ob + j * jb, valid, a.vec_b);
        }
      }
    }
  };
  // LDS column swizzle of row k: bit 5 from k's parity (the MFMA operand read: lanes 0-31 row 2s, lanes 32-63 row 2s + 1 -- the
  // two halves of the banks), bits 3-4 from (k / 4) % 4 (the contraction-contiguous store: lanes with different k land in
  // different column blocks, 2-way conflicts instead of 8-way); multiples of 8, so a 16-byte store stays in one piece
  auto swz = [](int k) { return ((k & 1) << 5) | (((k >> 2) & 3) << 3); };
  auto sstore = [&](auto sc, int buf) {
    constexpr int S = decltype(sc)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (kAK) {
        const int k = ka + 16 * j;
        *reinterpret_cast<float4*>(&As[kg][buf][k][ca ^ swz(k)]) = ra[S][j];
      } else {                                             // (ka is a multiple of 4: rows ka .. ka + 3 share bits 3-4 of the swizzle)
        const int c = (ca + 32 * j) ^ swz(ka);
        As[kg][buf][ka][c] = ra[S][j].x; As[kg][buf][ka + 1][c ^ 32] = ra[S][j].y;
        As[kg][buf][ka + 2][c] = ra[S][j].z; As[kg][buf][ka + 3][c ^ 32] = ra[S][j].w;
      }
      if constexpr (kBK) {
        const int k = kb + 16 * j;
        *reinterpret_cast<float4*>(&Bs[kg][buf][k][cb ^ swz(k)]) = rb[S][j];
      } else {
        const int c = (cb + 32 * j) ^ swz(kb);
        Bs[kg][buf][kb][c] = rb[S][j].x; Bs[kg][buf][kb + 1][c ^ 32] = rb[S][j].y;
        Bs[kg][buf][kb + 2][c] = rb[S][j].z; Bs[kg][buf][kb + 3][c ^ 32] = rb[S][j].w;
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // operand columns of row k = 2s + lh: (k / 4) % 4 = (s / 2) % 4 -- four loop-invariant variants
  int acol[4], bcol[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    acol[v] = (wm * 32 + ln) ^ (lh << 5) ^ (v << 3);
    bcol[v] = (wn * 32 + ln) ^ (lh << 5) ^ (v << 3);
  }
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < kChunk / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kg][buf][2 * s + lh][acol[(s >> 1) & 3]], Bs[kg][buf][2 * s + lh][bcol[(s >> 1) & 3]], acc, 0, 0, 0);
  };
  // group kg's chunks are kg, kg + kKG, ...: `rounds` steps for every group (a group past the end of K multiplies zeros: its
  // loads are predicated off and the stored pieces are zero)
  const int chunks = (a.K + kChunk - 1) / kChunk, rounds = (chunks + kKG - 1) / kKG;
  auto k_of = [&](int round) { return (round * kKG + kg) * kChunk; };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  // the same without a predicate or a branch: full tiles, whole chunks (what the steady-state loop below runs on -- with loads
  // that are issued on every path the compiler can wait for exactly the older register set, vmcnt(4); behind conditional loads
  // it waits for everything, and only one chunk was ever in flight)
  auto gload_fast = [&](auto sc, int k0) {
    constexpr int S = decltype(sc)::value;
    const size_t oa = kAK ? (size_t)k0 * a.lda : (size_t)k0, ob = kBK ? (size_t)k0 * a.ldb : (size_t)k0;
    ra[S][0] = *reinterpret_cast<const float4*>(pa + oa);
    ra[S][1] = *reinterpret_cast<const float4*>(pa + oa + ja);
    rb[S][0] = *reinterpret_cast<const float4*>(pb + ob);
    rb[S][1] = *reinterpret_cast<const float4*>(pb + ob + jb);
  };
  gload(S0, k_of(0));
  sstore(S0, 0);
  __syncthreads();
  if (rounds > 1) gload(S1, k_of(1));
  int c = 0;
  if (full_a && full_b) {
    const int rounds_fast = a.K / (kKG * kChunk);           // rounds whose chunks are whole for every group
    for (; c + 3 < rounds_fast; c += 2) {
      gload_fast(S0, k_of(c + 2));
      compute(0);
      sstore(S1, 1);
      __syncthreads();
      gload_fast(S1, k_of(c + 3));
      compute(1);
      sstore(S0, 0);
      __syncthreads();
    }
  }
  for (; c < rounds; c += 2) {
    // LDS buffer 0 holds round c, register set 1 round c + 1
    if (c + 2 < rounds) gload(S0, k_of(c + 2));
    compute(0);
    if (c + 1 < rounds) sstore(S1, 1);
    __syncthreads();
    if (c + 1 >= rounds) break;
    // LDS buffer 1 holds round c + 1, register set 0 round c + 2
    if (c + 3 < rounds) gload(S1, k_of(c + 3));
    compute(1);
    if (c + 2 < rounds) sstore(S0, 0);
    __syncthreads();
  }
  // the groups' tiles, added in group order in LDS (a wave's 32 x 32 block as 16 rows of 64 lanes: its own cells only)
  if constexpr (kKG > 1) {
    float* red = &As[0][0][0][0] + (size_t)wave * 1024;      // 4 waves x 4 KiB of the (now idle) first A buffers
    for (int g = 1; g < kKG; ++g) {
      __syncthreads();
      if (kg == g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane];
      }
    }
    if (kg != 0) return;
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
  // contraction groups per workgroup: a function of K alone (the reduction order must not depend on anything else)
  const int kg = K >= 1024 ? 4 : K >= 192 ? 2 : 1;
#define T2O_GEMM_LAUNCH(AK, BK)                                                              \
  do {                                                                                       \
    if (kg == 4) k_gemm_any<AK, BK, 4><<<grid, 1024, 0, st>>>(a);                            \
    else if (kg == 2) k_gemm_any<AK, BK, 2><<<grid, 512, 0, st>>>(a);                        \
    else k_gemm_any<AK, BK, 1><<<grid, 256, 0, st>>>(a);                                     \
  } while (0)
  if (a_kmajor && b_kmajor) T2O_GEMM_LAUNCH(true, true);
  else if (a_kmajor) T2O_GEMM_LAUNCH(true, false);
  else if (b_kmajor) T2O_GEMM_LAUNCH(false, true);
  else T2O_GEMM_LAUNCH(false, false);
#undef T2O_GEMM_LAUNCH
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "gemm launch failed");
}

int t2o_colsum(const float* X, float* out, int R, int N, int ldx, int accumulate, void* stream) {
  if (!X || !out) return set_error(T2O_EINVAL, "colsum: null pointer");
  if (R <= 0 || N <= 0 || ldx < N) return set_error(T2O_EINVAL, "colsum: R, N positive, ldx >= N");
  k_colsum<<<(unsigned)((N + 63) / 64), 256, 0, (hipStream_t)stream>>>(X, out, R, N, ldx, accumulate ? 1 : 0);
  return hipGetLastError() == hipSuccess ? T2O_OK : set_error(T2O_ELAUNCH, "colsum launch failed");
}

}  // extern "C"
