// hiprtc build of the fused block kernel + on-disk code-object cache, module loading, and the register-budget rule
// (no kernel runs from scratch memory).  gfx950 only.
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <limits.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

#include <sys/file.h>

#include "fz_runtime.hpp"

extern char** environ;

namespace fz {

int device_count()
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
   }
   return n;
}

void require_device()
{
   if (device_count() <= 0)
      fail(FZ_E_NO_DEVICE, "no HIP device visible: libflowz_hip evaluates flow-graphs on an MI355X only "
                           "(there is no CPU fallback in the product path)");
}

Kernel::~Kernel()
{
   if (loaded.empty()) return;
   int cur = 0;
   (void)hipGetDevice(&cur);
   for (const Loaded& l : loaded) {
      (void)hipSetDevice(l.device);
      (void)hipDeviceSynchronize();          // launches are asynchronous: never unload code that may still run
      (void)hipModuleUnload((hipModule_t)l.module);
   }
   (void)hipSetDevice(cur);
}

void* Kernel::function_on_current_device(const std::string& symbol)
{
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   for (const Loaded& l : loaded)
      if (l.device == dev) return l.function;
   hipModule_t mod;
   FZ_HIP(hipModuleLoadData(&mod, code.data()));
   hipFunction_t fn;
   FZ_HIP(hipModuleGetFunction(&fn, mod, symbol.c_str()));
   loaded.push_back(Loaded{dev, mod, fn});
   return fn;
}

// ---- kernel cache -----------------------------------------------------------------------------------
static std::vector<const char*> build_options(const Graph& g, const Variant& v)
{
   // -ffp-contract=off: one rounding per graph node (no v_fma/v_fmac); IEEE division.
   // The SLP vectoriser is off: with one stream per lane it pairs unrelated scalar
   // mul/add into v_pk_* at the price of v_mov shuffles, a net VALU loss on gfx950.
   std::vector<const char*> o = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                 "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"};
   // The parts of a wave split carry ONE or two packed pairs of segments: so little instruction-level parallelism that the default
   // scheduler (which orders for occupancy) leaves dependent v_pk_mul / v_pk_add back to back -- two s_nop per step in the ISA on
   // top of the wait.  The max-ILP strategy interleaves the atoms: 411 instead of 477 instructions per round of 32 steps, no
   // s_nop; measured +9 % at 16 384 streams and +1-4 % at 32 768 with rounds of 32 steps (profiles/r03/sweep_sched_strategy.txt).
   // Rounds of 16 steps next to an I/O wave LOSE 15-20 % with it, and the single-wave kernels (three pairs: enough ILP) 0-4 %.
   if (ws_parts(v.flags) >= 2 && v.U == 32) {
      o.push_back("-mllvm");
      o.push_back("-amdgpu-sched-strategy=max-ilp");
   }
   // The pair long-run stream-major body (two streams per lane, ONE wave per SIMD): the graph of a step is one serial chain of packed
   // operations and nothing else shares the SIMD, so the default order (a step after the other: every v_pk_add behind the v_pk_mul
   // it waits for, 857 s_nop in the cascade's code) would run at the latency of the chain.  The iterative ILP scheduler overlaps
   // the stages of consecutive steps (286 s_nop, two to three chains in flight).
   // (deep graphs only: on shallow ones it hoists every LDS read of the unrolled steps and runs out of registers)
   if ((v.flags & FZ_VF_STREAM_MAJOR) && (v.flags & FZ_VF_SM_LONG) && v.P == 2 && g.n_ops > 27) {
      o.push_back("-mllvm");
      o.push_back("-amdgpu-sched-strategy=iterative-ilp");
   }
   // developer hook (kernel experiments: -DFZ_DBG_NOLOAD ... and compiler flags); part of the cache key like every option
   static const std::vector<std::string> extra = [] {
      std::vector<std::string> e;
      if (const char* env = std::getenv("FLOWZ_HIP_EXTRA_OPTS")) {
         std::istringstream is(env);
         for (std::string t; is >> t;) e.push_back(t);
      }
      return e;
   }();
   for (const std::string& e : extra) o.push_back(e.c_str());
   return o;
}


// Where code objects are cached: FLOWZ_HIP_CACHE, else <package>/_kcache next to the library when that is
// writable (build() pre-fills it), else a PER-USER directory under /tmp (mode 0700, owner checked: another
// local user must not be able to plant a code object there).
// <package>/_kcache next to the library (build() pre-fills it); "" when the library's path is unknown
static std::string package_cache_dir()
{
   Dl_info info;
   if (!dladdr((const void*)&package_cache_dir, &info) || !info.dli_fname) return "";
   std::string p = info.dli_fname;                   // .../zignal_amd/lib/libflowz_hip.so
   size_t s = p.rfind('/');
   if (s != std::string::npos) p = p.substr(0, s);
   s = p.rfind('/');
   if (s != std::string::npos) p = p.substr(0, s);
   return p + "/_kcache";
}

std::string cache_dir()
{
   if (const char* env = std::getenv("FLOWZ_HIP_CACHE")) return env;
   const std::string pkg = package_cache_dir();
   if (!pkg.empty()) {
      ::mkdir(pkg.c_str(), 0755);
      if (::access(pkg.c_str(), W_OK | X_OK) == 0) return pkg;
   }
   const std::string d = "/tmp/flowz_hip_kcache-" + std::to_string((long)getuid());
   ::mkdir(d.c_str(), 0700);
   struct stat st;
   if (::lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return "";   // no cache
   return d;
}

// cache file = code object (an ELF: llvm-objdump / readelf still read the file) + trailer {magic, payload bytes, fnv1a of the payload}
struct CacheHeader {
   char magic[8];
   uint64_t size;
   uint64_t hash;
};
static const char kCacheMagic[8] = {'F', 'Z', 'K', 'C', '0', '0', '0', '3'};   // (0002: before the compiler was pinned -- objects of either hiprtc under one name)

static uint64_t fnv1a_bytes(const char* d, size_t n)
{
   uint64_t h = 1469598103934665603ull;
   for (size_t i = 0; i < n; ++i) {
      h ^= (unsigned char)d[i];
      h *= 1099511628211ull;
   }
   return h;
}

static bool cache_load(const std::string& path, std::vector<char>& code, bool may_delete = true)
{
   std::ifstream f(path, std::ios::binary);
   if (!f) return false;
   std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
   CacheHeader h;
   bool ok = raw.size() > sizeof h;
   if (ok) {
      std::memcpy(&h, raw.data() + raw.size() - sizeof h, sizeof h);
      ok = std::memcmp(h.magic, kCacheMagic, 8) == 0 && h.size == raw.size() - sizeof h && h.size > 64 &&
           h.hash == fnv1a_bytes(raw.data(), (size_t)h.size) && std::memcmp(raw.data(), "\x7f" "ELF", 4) == 0;
   }
   if (!ok) {
      if (may_delete) ::unlink(path.c_str());        // truncated / foreign / stale: never try it again
      return false;
   }
   raw.resize((size_t)h.size);
   code.swap(raw);
   return true;
}

static void cache_store(const std::string& dir, const std::string& path, const std::vector<char>& code)
{
   if (dir.empty()) return;
   ::mkdir(dir.c_str(), 0755);
   const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
   CacheHeader h;
   std::memcpy(h.magic, kCacheMagic, 8);
   h.size = code.size();
   h.hash = fnv1a_bytes(code.data(), code.size());
   bool ok = false;
   {
      std::ofstream f(tmp, std::ios::binary);
      if (f) {
         f.write(code.data(), (std::streamsize)code.size());
         f.write(reinterpret_cast<const char*>(&h), sizeof h);
         f.close();
         ok = f.good();                              // a short write (ENOSPC ...) must not be installed
      }
   }
   if (!ok || ::rename(tmp.c_str(), path.c_str()) != 0) ::unlink(tmp.c_str());
}

// ---- which hiprtc compiles the kernels ----------------------------------------------------------------------------------
// The library links libhiprtc.so.7 of the ROCm installation it was built against.  A host process that has ANOTHER copy with that
// soname loaded already -- a PyTorch wheel bundles the ROCm release it was built with, hiprtc and comgr (the compiler) included --
// binds us to that copy instead, and the code would then depend on who imported what first: the wheel's older compiler needs 22 more
// registers for the four-streams-per-lane headline kernel, which therefore "has scratch" and the library steps down to two
// (0.73 instead of 0.77 of peak).  Round 4 closes that: a library that finds itself bound to a foreign hiprtc hands every build
// to fz_rtc_worker (fz_rtc_worker.cpp, installed next to the library): a fresh process whose only hiprtc is the installation's.
// Same compiler, same options, same text: the code objects are byte-identical to what a process without torch builds, and they
// are cached under the installation's name.  Only when the worker cannot be run (not installed, not executable, bound to
// something else itself) does the host process's compiler build the kernel -- under a cache name of its own, never standing in
// for the installation's, and with ONE warning on stderr (FLOWZ_HIP_QUIET=1 silences it).
// (Round 3 tried the same with dlmopen -- the installation's hiprtc in a link-map namespace of its own inside the host process;
//  one of five full test runs ended in a segmentation fault nobody could explain.  A process boundary has no such failure mode.)
#ifndef FZ_ROCM_LIB_DIR
#define FZ_ROCM_LIB_DIR "/opt/rocm/lib"
#endif
struct Rtc {
   std::string identity;                               // part of every cache key
   std::string path;                                   // the hiprtc that builds the kernels
   std::string worker;                                 // "" : in-process; else the fz_rtc_worker executable
};

static std::string real_path(const std::string& p)
{
   char buf[PATH_MAX];
   return ::realpath(p.c_str(), buf) ? std::string(buf) : p;
}

// the identity of the installation's compiler: "libhiprtc.so.7.2.70200"
static const std::string& preferred_identity()
{
   static const std::string id = [] {
      const std::string ours = real_path(std::string(FZ_ROCM_LIB_DIR) + "/libhiprtc.so.7");
      const size_t s = ours.rfind('/');
      return s == std::string::npos ? ours : ours.substr(s + 1);
   }();
   return id;
}

static std::string library_dir()
{
   Dl_info info;
   if (!dladdr((const void*)&library_dir, &info) || !info.dli_fname) return "";
   std::string p = info.dli_fname;                   // .../zignal_amd/lib/libflowz_hip.so
   const size_t s = p.rfind('/');
   return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

// run the worker: argv = {worker, request, output}; environment without LD_LIBRARY_PATH / LD_PRELOAD; its stdout goes to out_path
static int run_worker(const std::string& worker, const std::string& request, const std::string& output, const std::string& stdout_path)
{
   std::vector<std::string> envs;
   for (char** e = environ; e && *e; ++e)
      if (std::strncmp(*e, "LD_LIBRARY_PATH=", 16) != 0 && std::strncmp(*e, "LD_PRELOAD=", 11) != 0) envs.push_back(*e);
   std::vector<char*> envp;
   for (std::string& e : envs) envp.push_back(&e[0]);
   envp.push_back(nullptr);
   std::string a0 = worker, a1 = request, a2 = output;
   char* argv[] = {&a0[0], &a1[0], output.empty() ? nullptr : &a2[0], nullptr};
   posix_spawn_file_actions_t fa;
   posix_spawn_file_actions_init(&fa);
   posix_spawn_file_actions_addopen(&fa, 1, stdout_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
   pid_t pid = 0;
   const int rc = posix_spawn(&pid, worker.c_str(), &fa, nullptr, argv, envp.data());
   posix_spawn_file_actions_destroy(&fa);
   if (rc != 0) return -1;
   int status = 0;
   while (waitpid(pid, &status, 0) < 0)
      if (errno != EINTR) return -1;
   return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

static std::string slurp(const std::string& path)
{
   std::ifstream f(path, std::ios::binary);
   return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// a private scratch directory for the worker's request / output files
static std::string worker_tmp_dir()
{
   const char* t = std::getenv("TMPDIR");
   std::string tmpl = std::string(t && *t ? t : "/tmp") + "/fz_rtc_XXXXXX";
   return ::mkdtemp(&tmpl[0]) ? tmpl : std::string();
}

static const Rtc& rtc()
{
   static const Rtc r = [] {
      Rtc t;
      Dl_info info;
      const std::string bound = dladdr((const void*)&hiprtcCompileProgram, &info) && info.dli_fname ? real_path(info.dli_fname) : std::string("?");
      const std::string ours = real_path(std::string(FZ_ROCM_LIB_DIR) + "/libhiprtc.so.7");
      const bool debug = std::getenv("FLOWZ_HIP_DEBUG") != nullptr;
      t.path = bound;
      bool foreign = bound != ours && ::access(ours.c_str(), R_OK) == 0;
      std::string why;
      if (foreign) {
         // the worker must exist, run, and be bound to the installation's hiprtc itself
         const std::string w = library_dir() + "/fz_rtc_worker";
         if (::access(w.c_str(), X_OK) != 0) (void)::chmod(w.c_str(), 0755);   // (a snapshot that dropped the mode bits)
         if (::access(w.c_str(), X_OK) != 0) why = w + " is missing or not executable";
         else {
            const std::string d = worker_tmp_dir();
            if (d.empty()) why = "no temporary directory";
            else {
               const int rc = run_worker(w, "--identify", "", d + "/stdout");
               const std::string said = slurp(d + "/stdout");
               ::unlink((d + "/stdout").c_str());
               ::rmdir(d.c_str());
               if (rc != 0 || said.rfind("hiprtc ", 0) != 0) why = "the worker did not start (exit " + std::to_string(rc) + ")";
               else if (real_path(said.substr(7, said.find('\n') - 7)) != ours) why = "the worker is bound to " + said.substr(7, said.find('\n') - 7);
               else {
                  t.worker = w;
                  t.path = ours;
                  foreign = false;
               }
            }
         }
      }
      // identity: the installation's hiprtc by its versioned file name (computable without loading it: see preferred_identity),
      // any other by path and size
      struct stat st;
      t.identity = foreign ? "foreign:" + bound + ":" + std::to_string(::stat(bound.c_str(), &st) == 0 ? (long long)st.st_size : -1LL) : preferred_identity();
      if (foreign && !std::getenv("FLOWZ_HIP_QUIET"))
         std::fprintf(stderr, "[flowz_hip] warning: kernels that are not in the cache will be built by %s, the hiprtc the host process loaded first, not by the "
                              "ROCm installation's (%s): %s.  Such kernels may need more registers (a spilling variant steps down to a slower one); "
                              "objects pre-built by the installation's compiler are still preferred.\n", bound.c_str(), ours.c_str(), why.c_str());
      if (debug)
         std::fprintf(stderr, "[flowz_hip] kernels are built by %s%s\n", t.path.c_str(), t.worker.empty() ? "" : " in a process of its own (fz_rtc_worker: the host process is bound to another hiprtc)");
      return t;
   }();
   return r;
}

static std::vector<char> jit_compile_in_process(const std::string& skel, const std::string& cfg, const std::string& body, const std::vector<const char*>& opts)
{
   const char* headers[2] = {cfg.c_str(), body.c_str()};
   const char* names[2] = {"fz_graph_config.h", "fz_graph_body.h"};
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, skel.c_str(), "fz_block_kernel.hip", 2, headers, names) != HIPRTC_SUCCESS)
      fail(FZ_E_COMPILE, "hiprtcCreateProgram failed");
   hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), const_cast<const char**>(opts.data()));
   if (r != HIPRTC_SUCCESS) {
      size_t n = 0;
      hiprtcGetProgramLogSize(prog, &n);
      std::string log(n, ' ');
      if (n) hiprtcGetProgramLog(prog, &log[0]);
      hiprtcDestroyProgram(&prog);
      fail(FZ_E_COMPILE, std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log);
   }
   size_t n = 0;
   hiprtcGetCodeSize(prog, &n);
   std::vector<char> code(n);
   hiprtcGetCode(prog, code.data());
   hiprtcDestroyProgram(&prog);
   return code;
}

static std::vector<char> jit_compile_in_worker(const std::string& worker, const std::string& skel, const std::string& cfg, const std::string& body, const std::vector<const char*>& opts)
{
   const std::string d = worker_tmp_dir();
   if (d.empty()) fail(FZ_E_COMPILE, "fz_rtc_worker: no temporary directory for the request");
   const std::string req = d + "/request", out = d + "/code", so = d + "/stdout";
   {
      std::ofstream f(req, std::ios::binary);
      auto section = [&](const char* kind, const char* name, const std::string& data) {
         f << kind << ' ' << name << ' ' << data.size() << '\n';
         f.write(data.data(), (std::streamsize)data.size());
         f << '\n';
      };
      f << "FZRTC1 " << (3 + opts.size()) << '\n';
      section("source", "fz_block_kernel.hip", skel);
      section("header", "fz_graph_config.h", cfg);
      section("header", "fz_graph_body.h", body);
      for (const char* o : opts) section("option", "-", o);
   }
   const int rc = run_worker(worker, req, out, so);
   std::vector<char> code;
   std::string log;
   if (rc == 0) {
      const std::string bytes = slurp(out);
      code.assign(bytes.begin(), bytes.end());
   } else if (rc == 3) {
      log = slurp(out + ".log");
   }
   for (const char* n : {"/request", "/code", "/code.log", "/stdout"}) ::unlink((d + n).c_str());
   ::rmdir(d.c_str());
   if (rc == 3) fail(FZ_E_COMPILE, log);
   if (rc != 0 || code.size() < 64) fail(FZ_E_COMPILE, "fz_rtc_worker failed (exit " + std::to_string(rc) + ")");
   return code;
}

// (manifest builds compile in parallel: every thread hands its kernels to a compiler process of its own)
static thread_local bool tl_force_worker = false;

static std::vector<char> jit_compile(const Graph& g, const Variant& v)
{
   const std::string cfg = gen_config(g, v), body = gen_body(g, v);
   const std::vector<const char*> opts = build_options(g, v);
   const Rtc& R = rtc();
   std::string worker = R.worker;
   if (worker.empty() && tl_force_worker && R.identity == preferred_identity()) {
      const std::string w = library_dir() + "/fz_rtc_worker";
      if (::access(w.c_str(), X_OK) == 0) worker = w;
   }
   const std::string& skel = skeleton_source(v.flags);
   if (!worker.empty()) return jit_compile_in_worker(worker, skel, cfg, body, opts);
   static std::mutex in_process;                          // (hiprtc in one process: one build at a time)
   std::lock_guard<std::mutex> lock(in_process);
   return jit_compile_in_process(skel, cfg, body, opts);
}

// One field of the kernel's metadata map (code object v3+: an ELF note holding msgpack; one kernel per code object here).
// The key is a msgpack string, the value the msgpack unsigned integer right behind it.
static uint32_t note_uint(const std::vector<char>& code, const char* key)
{
   const size_t kl = std::strlen(key);
   const unsigned char* b = reinterpret_cast<const unsigned char*>(code.data());
   for (size_t i = 1; i + kl + 1 <= code.size(); ++i) {
      if (std::memcmp(b + i, key, kl) != 0) continue;
      const bool fixstr = b[i - 1] == (0xa0u | kl), str8 = i >= 2 && b[i - 2] == 0xd9 && b[i - 1] == kl;
      if (!fixstr && !str8) continue;                         // (the text inside a longer key or a value)
      const unsigned char* v = b + i + kl;
      const size_t left = code.size() - (i + kl);
      if (v[0] <= 0x7f) return v[0];
      if (v[0] == 0xcc && left >= 2) return v[1];
      if (v[0] == 0xcd && left >= 3) return (uint32_t)v[1] << 8 | v[2];
      if (v[0] == 0xce && left >= 5) return (uint32_t)v[1] << 24 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 8 | v[4];
      return 0xFFFFFFFFu;
   }
   return 0;
}

static KernelResources read_resources(const std::vector<char>& code)
{
   KernelResources r;
   r.vgprs = note_uint(code, ".vgpr_count");
   r.agprs = note_uint(code, ".agpr_count");
   r.sgprs = note_uint(code, ".sgpr_count");
   r.scratch_bytes = note_uint(code, ".private_segment_fixed_size");
   r.lds_bytes = note_uint(code, ".group_segment_fixed_size");
   r.vgpr_spills = note_uint(code, ".vgpr_spill_count");
   r.sgpr_spills = note_uint(code, ".sgpr_spill_count");
   return r;
}

// A frame kernel that spills keeps part of its prefetch buffers / delay lines in scratch memory.  There the unroll is only
// the prefetch depth, so it is an UPPER bound: halved until nothing spills.  A graph that spills even at unroll 1 runs as it
// is.  Stream-major kernels are left alone: their unroll is also the length of a stream's run in memory, and the 4-wire sum
// measured 0.98 ms with 32-sample chunks and 64 spilled registers against 1.33 ms with 16-sample chunks and none.
Variant settle_variant(fz_program* p, Variant v)
{
   static const bool off = std::getenv("FLOWZ_HIP_KEEP_SPILLS") != nullptr;   // (developer switch: measure the spilling kernel itself)
   if (off) return v;
   for (;;) {
      const auto k = get_kernel(p, v, nullptr);
      if (k->res.scratch_bytes == 0) return v;
      if (v.flags & FZ_VF_STREAM_MAJOR) {
         // the long-run body with 64-sample phases shares a SIMD between two waves (256 registers each): where that spills (the
         // ROCm 7.0 compiler: 60 bytes for the 6-biquad cascade) the 128-sample phases of a lone wave (512 registers) run instead
         if ((v.flags & FZ_VF_SM_LONG) && v.U == 64 && v.P == 1) {
            v.U = 128;
            continue;
         }
         // (the pair long-run body keeps 256 staging registers next to the graph's: a graph that does not fit runs the one-stream body)
         if ((v.flags & FZ_VF_SM_LONG) && v.P == 2) {
            v.P = 1;
            v.U = 128;
            // (with the stage packing the one-stream body would have had by itself: resolve_variant's rule for deep graphs)
            if (p->g.split.ok && p->g.split.atoms() <= 13 && p->g.n_ops > 27) v.flags |= FZ_VF_STAGE_PACK;
            continue;
         }
         return v;
      }
      if (ws_parts(v.flags) && v.block * ws_waves(v.flags) > 256 && v.block > 64) {
         v.block /= 2;                                   // more than four waves per workgroup cap the registers of a lane at 256: fewer tuples per workgroup first
         continue;
      }
      if (v.U <= (ws_parts(v.flags) ? 8u : 1u)) return v;
      v.U /= 2;
   }
}

// file name of a variant's code object: a hash of (generated source, build options, hiprtc version)
static std::string cache_file_of(const fz_program* p, const Variant& v, const std::string& compiler)
{
   std::string key_src = full_source(p->g, v);
   for (const char* o : build_options(p->g, v)) key_src += o;
   key_src += compiler;                                 // who built it (Rtc::identity / preferred_identity)
   char name[64];
   std::snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)fnv1a(key_src));
   return name;
}

// identity of a variant's CODE (16 hex digits): the name of its code object in the kernel cache under the installation's compiler.  Two
// kernels share it only when generated source, build options and compiler agree -- what counter tables are keyed by (profiles/)
// A kernel this process already holds answers with the id of the object it actually LOADED: a process bound to another hiprtc (a PyTorch
// wheel's) that had to build the kernel itself runs other instructions than the pre-built object of the same variant, and counters
// measured on one must not be attached to the other.
std::string kernel_code_id(fz_program* p, const Variant& v)
{
   {
      std::lock_guard<std::mutex> lock(p->mu);
      auto it = p->kernels.find(v);
      if (it != p->kernels.end() && it->second && it->second->built.load() && !it->second->code_id.empty()) return it->second->code_id;
   }
   return cache_file_of(p, v, preferred_identity()).substr(1, 16);
}

static bool cache_in_use(const std::string& dir) { return !std::getenv("FLOWZ_HIP_NO_CACHE") && !dir.empty(); }

// lookups: the installation's compiler first (pre-built objects), then whoever compiles in this process
static std::vector<std::string> compilers_to_look_for()
{
   std::vector<std::string> w{preferred_identity()};
   if (rtc().identity != w[0]) w.push_back(rtc().identity);
   return w;
}

bool kernel_at_hand(fz_program* p, const Variant& v)
{
   {
      std::lock_guard<std::mutex> lock(p->mu);
      auto it = p->kernels.find(v);
      if (it != p->kernels.end() && it->second && it->second->built.load()) return true;
   }
   const std::string dir = cache_dir(), pkg = package_cache_dir();
   if (!cache_in_use(dir)) return false;
   const bool ro_pkg = !pkg.empty() && pkg != dir && !std::getenv("FLOWZ_HIP_CACHE");
   for (const std::string& who : compilers_to_look_for()) {
      const std::string name = cache_file_of(p, v, who);
      if (::access((dir + name).c_str(), R_OK) == 0 || (ro_pkg && ::access((pkg + name).c_str(), R_OK) == 0)) return true;
   }
   return false;
}

thread_local bool tl_no_jit = false;

// ---- kernel manifests ---------------------------------------------------------------------------------------------------------
// FLOWZ_HIP_MANIFEST=<file>: every kernel a process resolves for the first time is appended as (recipe of its program, variant) -- a few
// hundred bytes.  fz_manifest_build replays such a file WITHOUT a GPU: compiles the programs again and builds, in parallel compiler
// processes, whatever the kernel cache lacks.  The records name expressions and variants, not generated text: a replay after the kernel
// skeleton or the code generator changed builds the NEW kernels of the same launches (round 5: the GPU test suite launches ~1800 kernels;
// a box that has to JIT them all needs 10 minutes for what takes 80 s from a warm cache).
static void manifest_record(const fz_program* p, const Variant& v)
{
   static const char* const path = std::getenv("FLOWZ_HIP_MANIFEST");
   if (!path || !*path || p->recipe.empty()) return;
   char head[96];
   std::snprintf(head, sizeof head, "FZM1 %u %u %u %u %zu\n", v.P, v.U, v.block, v.flags, p->recipe.size());
   const std::string rec = head + p->recipe;
   {
      static std::mutex mu;
      static std::set<uint64_t> seen;                      // (a test suite compiles the same graphs hundreds of times)
      std::lock_guard<std::mutex> lock(mu);
      if (!seen.insert(fnv1a(rec)).second) return;
   }
   const int fd = ::open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
   if (fd < 0) return;
   (void)::flock(fd, LOCK_EX);                             // (several processes may share the file: records never interleave)
   size_t off = 0;
   while (off < rec.size()) {
      const ssize_t n = ::write(fd, rec.data() + off, rec.size() - off);
      if (n <= 0) break;
      off += (size_t)n;
   }
   (void)::flock(fd, LOCK_UN);
   ::close(fd);
}

int manifest_build(const std::string& path, unsigned n_workers, uint32_t counts[4])
{
   const std::string text = slurp(path);
   if (text.empty()) fail(FZ_E_INVALID, "kernel manifest: cannot read " + path);
   // records -> unique (recipe, variant) pairs
   std::map<std::string, std::set<Variant>> want;
   size_t pos = 0;
   uint32_t bad_records = 0;
   while (pos < text.size()) {
      const size_t eol = text.find('\n', pos);
      if (eol == std::string::npos) break;
      Variant v;
      size_t n = 0;
      if (std::sscanf(text.c_str() + pos, "FZM1 %u %u %u %u %zu", &v.P, &v.U, &v.block, &v.flags, &n) != 5 || n > text.size() - (eol + 1))
         fail(FZ_E_INVALID, "kernel manifest: damaged record at byte " + std::to_string(pos));
      pos = eol + 1 + n;
      // (the file is data from elsewhere: a variant no launch could have resolved -- it would divide by P or size a workgroup by `block`
      //  further down -- is counted as failed, not built)
      if ((v.P != 1 && v.P != 2 && v.P != 4) || v.U == 0 || v.U > 128 || v.block == 0 || v.block % 64 != 0 || v.block > 1024) {
         ++bad_records;
         continue;
      }
      want[text.substr(eol + 1, n)].insert(v);
   }
   struct Item { fz_program* p; Variant v; };
   std::vector<std::unique_ptr<fz_program>> programs;
   std::vector<Item> items;
   counts[0] = counts[3] = bad_records;                    // records, at hand, built, failed
   counts[1] = counts[2] = 0;
   for (const auto& kv : want) {
      const std::string& recipe = kv.first;
      const size_t eol = recipe.find('\n');
      unsigned typed = 0;
      if (eol == std::string::npos || std::sscanf(recipe.c_str(), "typed %u", &typed) != 1) fail(FZ_E_INVALID, "kernel manifest: damaged recipe");
      std::vector<uint32_t> dt;
      {
         std::istringstream is(recipe.substr(7, eol - 7));
         for (unsigned d; is >> d;) dt.push_back(d);
      }
      fz_expr* e = parse_expr(recipe.substr(eol + 1));
      fz_program* p = nullptr;
      const int rc = !e ? FZ_E_INVALID : typed ? fz_compile_typed(e, dt.empty() ? nullptr : dt.data(), (uint32_t)dt.size(), &p) : fz_compile(e, &p);
      fz_expr_release(e);
      counts[0] += (uint32_t)kv.second.size();
      if (rc != FZ_OK || !p) {                              // (a graph this build of the library no longer accepts)
         counts[3] += (uint32_t)kv.second.size();
         continue;
      }
      programs.emplace_back(p);
      for (const Variant& v : kv.second) items.push_back(Item{p, v});
   }
   std::atomic<size_t> next{0};
   std::atomic<uint32_t> at_hand{0}, built{0}, failed{0};
   auto work = [&] {
      tl_force_worker = true;
      for (size_t i; (i = next.fetch_add(1)) < items.size();) {
         try {
            if (kernel_at_hand(items[i].p, items[i].v)) {
               ++at_hand;
               continue;
            }
            (void)get_kernel(items[i].p, items[i].v, nullptr);
            ++built;
         } catch (const Error&) {
            ++failed;                                       // (a variant the graph no longer allows, a kernel that no longer compiles)
         } catch (const std::exception&) {
            ++failed;                                       // (anything else a damaged record provokes: never std::terminate from a worker thread)
         }
      }
      tl_force_worker = false;
   };
   std::vector<std::thread> ths;
   for (unsigned t = 1; t < std::max(1u, n_workers); ++t) ths.emplace_back(work);
   work();
   for (std::thread& t : ths) t.join();
   counts[1] = at_hand;
   counts[2] = built;
   counts[3] += failed;
   return FZ_OK;
}

// The program mutex is held only to find (or create) the variant's slot; cache lookup, the hiprtc build (seconds) and module
// loading happen under the SLOT's own mutex, so other launches of the program -- other variants, other threads -- go on.
std::shared_ptr<Kernel> get_kernel(fz_program* p, const Variant& v, void** fn_out)
{
   std::shared_ptr<Kernel> k;
   {
      std::lock_guard<std::mutex> lock(p->mu);
      auto& slot = p->kernels[v];
      if (!slot) slot = std::make_shared<Kernel>();
      k = slot;
   }
   std::lock_guard<std::mutex> build_lock(k->mu);
   if (!k->built.load()) {
      const std::string dir = cache_dir(), pkg = package_cache_dir(), path = dir + cache_file_of(p, v, rtc().identity);   // (path: where a build of THIS process goes)
      const bool use_cache = cache_in_use(dir);
      // (a package cache this user cannot write to -- installed by root, pre-filled by build() -- is still read)
      const bool ro_pkg = use_cache && !pkg.empty() && pkg != dir && !std::getenv("FLOWZ_HIP_CACHE");
      bool found = false;
      if (use_cache)
         for (const std::string& who : compilers_to_look_for()) {
            const std::string name = cache_file_of(p, v, who);
            if (cache_load(dir + name, k->code)) {
               k->cache_path = dir + name;
               found = true;
            } else if (ro_pkg && cache_load(pkg + name, k->code, false)) {
               found = true;                             // (no cache_path: not ours to delete)
            }
            if (found) {
               k->code_id = name.substr(1, 16);
               break;
            }
         }
      if (!found) {
         // (the plan measurement a first big launch makes by itself never waits for a build: NoJitScope)
         if (tl_no_jit) fail(FZ_E_UNSUPPORTED, "kernel not at hand (it would have to be built)");
         k->code = jit_compile(p->g, v);
         k->code_id = cache_file_of(p, v, rtc().identity).substr(1, 16);
         if (use_cache) {
            cache_store(dir, path, k->code);
            k->cache_path = path;
         }
      }
      k->res = read_resources(k->code);
      k->built.store(true);
      manifest_record(p, v);
   }
   if (fn_out) {
      require_device();
      try {
         *fn_out = k->function_on_current_device(kernel_symbol(p->g, v));
      } catch (const Error& er) {
         // a cached code object the DRIVER refuses to load (built for another code-object version, damaged in a way the trailer
         // does not see): delete it, build afresh, store that, try once more.  Anything else -- out of memory, no device, a
         // missing symbol -- is not the file's fault and is passed on.
         const bool image = er.msg.find("hipModuleLoadData") != std::string::npos &&
                            (er.msg.find("invalid") != std::string::npos || er.msg.find("binary") != std::string::npos ||
                             er.msg.find("image") != std::string::npos || er.msg.find("shared object") != std::string::npos);
         if (k->cache_path.empty() || !image) throw;
         ::unlink(k->cache_path.c_str());
         k->code = jit_compile(p->g, v);
         k->res = read_resources(k->code);
         k->cache_path = cache_dir() + cache_file_of(p, v, rtc().identity);   // (under the name of the compiler that built it)
         k->code_id = cache_file_of(p, v, rtc().identity).substr(1, 16);
         cache_store(cache_dir(), k->cache_path, k->code);
         *fn_out = k->function_on_current_device(kernel_symbol(p->g, v));
      }
   }
   return k;
}

}  // namespace fz
