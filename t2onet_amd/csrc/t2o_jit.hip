// t2o_jit.hip -- run-time specialisation of the fused operator-chain kernels for ANY operator list.
//
// The fast chain kernels (k_chain_fwd_static / k_chain_bwd_static, t2o_chain_kernels.h) take the operator list as a
// template argument: both sweeps unrolled, no `for k` / `switch (op)`, raw parameter sums in registers for all of a
// thread's pixels (backward 91 vs 124 us for the run-time-loop kernel at bs=64 256x256).  Ahead of time only the two
// benchmark lists are instantiated; every other order -- whatever Executor.execute permits (executors/executor.py:33-55),
// what the planner enumerates (utils/beam_search.py:218-231) -- gets ITS instantiation here, compiled with hipRTC from
// the very headers the ahead-of-time build uses (embedded in the library at build time: t2o_jit_embedded.h), with the
// same flags (-O3 -ffp-contract=off: same arithmetic, bit-identical images), loaded with hipModuleLoadData and kept in
// a process-wide registry; the code object is also kept on disk (t2o_jit_set_cache_dir) so that a list is compiled once
// per source digest.  libhiprtc is opened with dlopen: the library has no link-time dependency on it, and a machine
// without it simply keeps the run-time-loop kernels (t2o_fused_sequence_prepare then reports T2O_EUNSUPPORTED).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <mutex>
#include <string>
#include <vector>

#include "t2onet_hip.h"
#include "t2o_block_programs.h"
#include "t2o_jit.h"
#include "t2o_jit_embedded.h"


namespace t2o {
int set_error(int code, const char* msg);

namespace {

struct Rtc {
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  bool ok = false;
};

const Rtc& rtc() {
  static Rtc r = [] {
    Rtc x;
    void* h = nullptr;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return x;
#define T2O_SYM(field, sym) x.field = reinterpret_cast<decltype(x.field)>(dlsym(h, #sym))
    T2O_SYM(create, hiprtcCreateProgram); T2O_SYM(compile, hiprtcCompileProgram); T2O_SYM(log_size, hiprtcGetProgramLogSize);
    T2O_SYM(log, hiprtcGetProgramLog); T2O_SYM(code_size, hiprtcGetCodeSize); T2O_SYM(code, hiprtcGetCode);
    T2O_SYM(destroy, hiprtcDestroyProgram);
#undef T2O_SYM
    x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
    return x;
  }();
  return r;
}

struct Entry {
  int device;                           // hipModuleLoadData binds a module to the device current at load time
  int K;
  int ops[kMaxChain];
  JitChain fn;
};
#ifndef T2O_ARCH
#define T2O_ARCH "gfx950"               // (t2onet_amd/build.py passes the offload architecture of the ahead-of-time build)
#endif

int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  return d;
}
std::mutex g_mu;
std::vector<Entry> g_entries;           // (entries are never removed: pointers into it are not handed out, copies are)
std::string g_cache_dir;

unsigned long long fnv64(const std::string& s) {
  unsigned long long h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

std::string chain_source(const int* ops, int K) {
  std::string seq;
  for (int k = 0; k < K; ++k) seq += (k ? ", " : "") + std::to_string(ops[k]);
  std::string s = "#include \"t2o_chain_kernels.h\"\nusing SEQ = t2o::StaticChain<" + seq + ">;\n";
  const char* fmt_fwd = "extern \"C\" __global__ __launch_bounds__(256) void %s(t2o::ChainArgs a) { t2o::chain_fwd_static_body<%d, %s, SEQ>(a); }\n";
  const char* fmt_bwd = "extern \"C\" __global__ __launch_bounds__(256) void %s(t2o::ChainArgs a) { t2o::chain_bwd_static_body<%s, SEQ, false>(a); }\n";
  char line[512];
  for (int v = 1; v <= 2; ++v)
    for (int l1 = 0; l1 < 2; ++l1) {
      const std::string name = std::string("t2o_jit_fwd_v") + std::to_string(v) + (l1 ? "_l1" : "");
      snprintf(line, sizeof(line), fmt_fwd, name.c_str(), v, l1 ? "true" : "false");
      s += line;
    }
  for (int l1 = 0; l1 < 2; ++l1) {
    snprintf(line, sizeof(line), fmt_bwd, l1 ? "t2o_jit_bwd_l1" : "t2o_jit_bwd", l1 ? "true" : "false");
    s += line;
  }
  return s;
}

bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n > 0 && fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

void write_file_atomic(const std::string& path, const std::vector<char>& data) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
  ok = (fclose(f) == 0) && ok;          // (a full disk shows up here)
  if (ok) rename(tmp.c_str(), path.c_str()); else remove(tmp.c_str());
}

int compile(const std::string& src, std::vector<char>& code, std::string& err) {
  const Rtc& r = rtc();
  if (!r.ok) { err = "libhiprtc.so not found: no run-time specialisation on this machine"; return T2O_EUNSUPPORTED; }
  hiprtcProgram prog = nullptr;
  if (r.create(&prog, src.c_str(), "t2o_jit_chain.hip", kJitHeaderCount, kJitHeaderSources, kJitHeaderNames) != HIPRTC_SUCCESS) {
    err = "hiprtcCreateProgram failed";
    return T2O_ELAUNCH;
  }
  const char* opts[] = {"--offload-arch=" T2O_ARCH, "-O3", "-ffp-contract=off", "-std=c++17"};
  const hiprtcResult rc = r.compile(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    r.log_size(prog, &n);
    std::vector<char> log(n + 1, 0);
    if (n) r.log(prog, log.data());
    err = std::string("hipRTC compile failed: ") + std::string(log.data()).substr(0, 180);
    r.destroy(&prog);
    return T2O_ELAUNCH;
  }
  size_t n = 0;
  r.code_size(prog, &n);
  code.resize(n);
  r.code(prog, code.data());
  r.destroy(&prog);
  return T2O_OK;
}

}  // namespace

static bool lookup_locked(int device, const int* ops, int K, JitChain* out) {
  for (const Entry& e : g_entries)
    if (e.device == device && e.K == K && memcmp(e.ops, ops, sizeof(int) * K) == 0) { *out = e.fn; return true; }
  return false;
}

bool jit_lookup(const int* ops, int K, JitChain* out) {
  const int device = current_device();
  std::lock_guard<std::mutex> lk(g_mu);
  return lookup_locked(device, ops, K, out);
}

// code object -> module -> the six kernels; false (module unloaded again) when anything is missing
static bool load_entry(const std::vector<char>& code, Entry& e, hipModule_t* mod_out) {
  hipModule_t mod = nullptr;
  if (code.empty() || hipModuleLoadData(&mod, code.data()) != hipSuccess) return false;
  const char* names[6] = {"t2o_jit_fwd_v1", "t2o_jit_fwd_v1_l1", "t2o_jit_fwd_v2", "t2o_jit_fwd_v2_l1", "t2o_jit_bwd", "t2o_jit_bwd_l1"};
  hipFunction_t* slots[6] = {&e.fn.fwd[0][0], &e.fn.fwd[0][1], &e.fn.fwd[1][0], &e.fn.fwd[1][1], &e.fn.bwd[0], &e.fn.bwd[1]};
  for (int i = 0; i < 6; ++i)
    if (hipModuleGetFunction(slots[i], mod, names[i]) != hipSuccess) { (void)hipModuleUnload(mod); return false; }
  *mod_out = mod;
  return true;
}

int jit_prepare(const int* ops, int K) {
  if (K <= 0 || K > kMaxChain) return set_error(T2O_EINVAL, "jit_prepare: 1..8 operators");
  JitChain have;
  if (jit_lookup(ops, K, &have)) return T2O_OK;
  const std::string src = chain_source(ops, K);
  char key[64];
  snprintf(key, sizeof(key), "%016llx", fnv64(std::string(t2o_source_digest()) + "|" T2O_ARCH "|O3|nocontract|" + src));
  std::string cache_dir;
  { std::lock_guard<std::mutex> lk(g_mu); cache_dir = g_cache_dir; }
  const std::string path = cache_dir.empty() ? std::string() : cache_dir + "/chain_" + key + ".hsaco";
  Entry e;
  e.device = current_device();
  e.K = K;
  memcpy(e.ops, ops, sizeof(int) * K);
  hipModule_t mod = nullptr;
  std::vector<char> code;
  bool loaded = !path.empty() && read_file(path, code) && load_entry(code, e, &mod);
  if (!loaded) {
    // nothing cached, or a cached file this runtime cannot use (truncated, another compiler): evict it and compile
    if (!path.empty()) remove(path.c_str());
    std::string err;
    const int rc = compile(src, code, err);
    if (rc != T2O_OK) return set_error(rc, err.c_str());
    if (!load_entry(code, e, &mod)) return set_error(T2O_ELAUNCH, "jit_prepare: the compiled module does not load (hipModuleLoadData / hipModuleGetFunction)");
    if (!path.empty()) write_file_atomic(path, code);
  }
  std::lock_guard<std::mutex> lk(g_mu);
  JitChain other;
  if (lookup_locked(e.device, ops, K, &other)) { (void)hipModuleUnload(mod); return T2O_OK; }   // another thread was faster
  g_entries.push_back(e);
  return T2O_OK;
}

int jit_launch(hipFunction_t f, const ChainArgs& a, unsigned grid, size_t lds_bytes, hipStream_t st) {
  ChainArgs copy = a;
  void* params[] = {&copy};
  return hipModuleLaunchKernel(f, grid, 1, 1, kThreads, 1, 1, (unsigned)lds_bytes, st, params, nullptr) == hipSuccess ? T2O_OK : T2O_ELAUNCH;
}

int jit_count() {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_entries.size();
}

}  // namespace t2o

extern "C" {

int t2o_jit_set_cache_dir(const char* dir) {
  std::lock_guard<std::mutex> lk(t2o::g_mu);
  t2o::g_cache_dir = dir ? dir : "";
  return T2O_OK;
}

int t2o_jit_specialisations(void) { return t2o::jit_count(); }

}  // extern "C"
