// t2o_comm.hip -- the train step's ONE collective through the C ABI (t2o_allreduce(float* flat, size_t n,
// ncclComm_t, hipStream_t)): the sum all-reduce of the flat fp32 gradient buffer (22.2 M floats, 88.7 MB) over RCCL / xGMI,
// stream-ordered on the caller's stream, in place.  The reference has no data parallelism (train_seq2seqL1.py:74-88 runs one
// process); the boundary is the build-side contract of SURVEY 8(b) / 8(e).
//
// RCCL is NOT a link-time dependency of this library: its entry points are resolved at first use with dlopen / dlsym, first
// from an RCCL the process already holds (RTLD_NOLOAD: the framework's bundled copy when torch.distributed is in use -- two
// RCCL instances in one process would each run their own proxy threads), then from the loader path, then from /opt/rocm/lib.
// A host without RCCL gets T2O_EUNSUPPORTED with the loader's message from every entry point here, nothing else changes.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "t2onet_hip.h"

namespace t2o { int set_error(int code, const char* msg); }
using t2o::set_error;

namespace {

// the slice of rccl.h this file needs (ABI of NCCL 2.x: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[128]; };
using comm_t = void*;
constexpr int kFloat32 = 7, kSum = 0;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*CommCount)(comm_t, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char why[256] = {0};
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)                       // an RCCL this process already holds first
      if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!r.handle)
      for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) {
      const char* e = dlerror();
      snprintf(r.why, sizeof(r.why), "RCCL not available: %s", e ? e : "librccl.so not found");
      return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.AllReduce) {
      snprintf(r.why, sizeof(r.why), "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclCommCount / ncclAllReduce");
      r.handle = nullptr;
    }
  });
  return r;
}

int rccl_fail(const char* what, int rc) {
  char msg[320];
  Rccl& r = rccl();
  snprintf(msg, sizeof(msg), "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
  return set_error(T2O_ELAUNCH, msg);
}

__global__ __launch_bounds__(256) void k_scale(float* p, size_t n, float s) {
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<float4*>(p)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    reinterpret_cast<float4*>(p)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] *= s;
}

}  // namespace

extern "C" {

int t2o_comm_available(void) { return rccl().handle ? 1 : 0; }

int t2o_comm_unique_id(void* id128) {
  Rccl& r = rccl();
  if (!r.handle) return set_error(T2O_EUNSUPPORTED, r.why);
  if (!id128) return set_error(T2O_EINVAL, "comm_unique_id: null pointer");
  UniqueId id;
  const int rc = r.GetUniqueId(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return T2O_OK;
}

int t2o_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
  Rccl& r = rccl();
  if (!r.handle) return set_error(T2O_EUNSUPPORTED, r.why);
  if (!comm || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return set_error(T2O_EINVAL, "comm_init_rank: null pointer or rank outside [0, nranks)");
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  comm_t c = nullptr;
  const int rc = r.CommInitRank(&c, nranks, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  *comm = c;
  return T2O_OK;
}

int t2o_comm_destroy(void* comm) {
  Rccl& r = rccl();
  if (!r.handle) return set_error(T2O_EUNSUPPORTED, r.why);
  if (!comm) return T2O_OK;
  const int rc = r.CommDestroy((comm_t)comm);
  return rc == 0 ? T2O_OK : rccl_fail("ncclCommDestroy", rc);
}

int t2o_allreduce(float* flat, size_t n, void* comm, void* stream) {
  Rccl& r = rccl();
  if (!r.handle) return set_error(T2O_EUNSUPPORTED, r.why);
  if (!flat || !comm) return set_error(T2O_EINVAL, "allreduce: null buffer or communicator");
  if (n == 0) return T2O_OK;
  const int rc = r.AllReduce(flat, flat, n, kFloat32, kSum, (comm_t)comm, (hipStream_t)stream);
  return rc == 0 ? T2O_OK : rccl_fail("ncclAllReduce", rc);
}

int t2o_allreduce_mean(float* flat, size_t n, void* comm, void* stream) {
  Rccl& r = rccl();
  if (!r.handle) return set_error(T2O_EUNSUPPORTED, r.why);
  if (!flat || !comm) return set_error(T2O_EINVAL, "allreduce_mean: null buffer or communicator");
  if (((uintptr_t)flat & 15) != 0) return set_error(T2O_EINVAL, "allreduce_mean: buffer not 16-byte aligned");
  int nranks = 0;
  int rc = r.CommCount((comm_t)comm, &nranks);
  if (rc != 0 || nranks <= 0) return rccl_fail("ncclCommCount", rc);
  if (n == 0) return T2O_OK;
  rc = r.AllReduce(flat, flat, n, kFloat32, kSum, (comm_t)comm, (hipStream_t)stream);
  if (rc != 0) return rccl_fail("ncclAllReduce", rc);
  if (nranks > 1) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    k_scale<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(flat, n, 1.0f / (float)nranks);
    if (hipGetLastError() != hipSuccess) return set_error(T2O_ELAUNCH, "allreduce_mean: scale launch failed");
  }
  return T2O_OK;
}

}  // extern "C"
