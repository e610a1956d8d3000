// Runtime of libflowz_hip: hiprtc build + on-disk code-object cache, module loading, the launch
// of the fused block kernel, device-resident closure state (fz_bank) and the AOT utility kernels
// (synthetic input fill, copy-bandwidth probe).  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "fz_internal.hpp"

namespace fz {

#define FZ_HIP(call)                                                                            \
   do {                                                                                         \
      hipError_t e_ = (call);                                                                   \
      if (e_ != hipSuccess)                                                                     \
         fail(FZ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                     \
   } while (0)

int device_count()
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
   }
   return n;
}

static void require_device()
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
static std::vector<const char*> build_options(const Variant& v)
{
   // -ffp-contract=off: one rounding per graph node (no v_fma/v_fmac); IEEE division.
   // The SLP vectoriser is off by default: with one stream per lane it pairs unrelated scalar
   // mul/add into v_pk_* at the price of v_mov shuffles, a net VALU loss on gfx950.
   std::vector<const char*> o = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                 "-fhip-fp32-correctly-rounded-divide-sqrt"};
   if (!(v.flags & FZ_VF_SLP)) o.push_back("-fno-slp-vectorize");
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

static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull)
{
   for (unsigned char ch : s) {
      h ^= ch;
      h *= 1099511628211ull;
   }
   return h;
}

// Where code objects are cached: FLOWZ_HIP_CACHE, else <package>/_kcache next to the library when that is
// writable (build() pre-fills it), else a PER-USER directory under /tmp (mode 0700, owner checked: another
// local user must not be able to plant a code object there).
static std::string cache_dir()
{
   if (const char* env = std::getenv("FLOWZ_HIP_CACHE")) return env;
   Dl_info info;
   if (dladdr((const void*)&cache_dir, &info) && info.dli_fname) {
      std::string p = info.dli_fname;                // .../zignal_amd/lib/libflowz_hip.so
      size_t s = p.rfind('/');
      if (s != std::string::npos) p = p.substr(0, s);
      s = p.rfind('/');
      if (s != std::string::npos) p = p.substr(0, s);
      const std::string d = p + "/_kcache";
      ::mkdir(d.c_str(), 0755);
      if (::access(d.c_str(), W_OK | X_OK) == 0) return d;
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
static const char kCacheMagic[8] = {'F', 'Z', 'K', 'C', '0', '0', '0', '2'};

static uint64_t fnv1a_bytes(const char* d, size_t n)
{
   uint64_t h = 1469598103934665603ull;
   for (size_t i = 0; i < n; ++i) {
      h ^= (unsigned char)d[i];
      h *= 1099511628211ull;
   }
   return h;
}

static bool cache_load(const std::string& path, std::vector<char>& code)
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
      ::unlink(path.c_str());                        // truncated / foreign / stale: never try it again
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

static std::vector<char> jit_compile(const Graph& g, const Variant& v)
{
   const std::string cfg = gen_config(g, v), body = gen_body(g, v);
   const char* headers[2] = {cfg.c_str(), body.c_str()};
   const char* names[2] = {"fz_graph_config.h", "fz_graph_body.h"};
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, skeleton_source(), "fz_block_kernel.hip", 2, headers, names) != HIPRTC_SUCCESS)
      fail(FZ_E_COMPILE, "hiprtcCreateProgram failed");
   std::vector<const char*> opts = build_options(v);
   hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
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
      if (v.flags & FZ_VF_STREAM_MAJOR) return v;
      if (ws_parts(v.flags) && v.block * ws_waves(v.flags) > 256 && v.block > 64) {
         v.block /= 2;                                   // more than four waves per workgroup cap the registers of a lane at 256: fewer tuples per workgroup first
         continue;
      }
      if (v.U <= (ws_parts(v.flags) ? 8u : 1u)) return v;
      v.U /= 2;
   }
}

std::shared_ptr<Kernel> get_kernel(fz_program* p, const Variant& v, void** fn_out)
{
   std::lock_guard<std::mutex> lock(p->mu);
   auto& slot = p->kernels[v];
   if (!slot) {
      auto k = std::make_shared<Kernel>();
      std::string key_src = full_source(p->g, v);
      for (const char* o : build_options(v)) key_src += o;
      int rtc_major = 0, rtc_minor = 0;
      hiprtcVersion(&rtc_major, &rtc_minor);
      key_src += "hiprtc" + std::to_string(rtc_major) + "." + std::to_string(rtc_minor);
      char name[64];
      std::snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)fnv1a(key_src));
      const std::string dir = cache_dir(), path = dir + name;
      const bool use_cache = !std::getenv("FLOWZ_HIP_NO_CACHE") && !dir.empty();
      k->cache_path = use_cache ? path : std::string();
      if (!(use_cache && cache_load(path, k->code))) {
         k->code = jit_compile(p->g, v);
         if (use_cache) cache_store(dir, path, k->code);
      }
      k->res = read_resources(k->code);
      slot = k;
   }
   if (fn_out) {
      require_device();
      try {
         *fn_out = slot->function_on_current_device(kernel_name(p->g, v));
      } catch (const Error&) {
         // a cached code object the driver refuses: delete it, build afresh, try once more
         if (slot->cache_path.empty()) throw;
         ::unlink(slot->cache_path.c_str());
         slot->cache_path.clear();
         slot->code = jit_compile(p->g, v);
         *fn_out = slot->function_on_current_device(kernel_name(p->g, v));
      }
   }
   return slot;
}

// ---- variant selection ---------------------------------------------------------------------------------
constexpr uint64_t kMaxLdsBytes = 160 * 1024;

Variant resolve_variant(const Graph& g, const fz_variant* uv, uint64_t n_streams, uint32_t n_samples)
{
   Variant v;
   const uint32_t reqP = uv ? uv->streams_per_lane : 0, reqU = uv ? uv->unroll : 0, reqB = uv ? uv->block_threads : 0;
   v.flags = uv ? uv->flags : 0;
   if (reqP != 0 && reqP != 1 && reqP != 2 && reqP != 4) fail(FZ_E_INVALID, "streams_per_lane must be 0, 1, 2 or 4");
   if (g.typed && (v.flags & FZ_VF_OUT_F64))
      fail(FZ_E_INVALID, "FZ_VF_OUT_F64 does not apply to fz_compile_typed programs: their frames carry every wire in its own type");
   if (reqU > 32 && !((v.flags & FZ_VF_SM_LONG) && (reqU == 64 || reqU == 128))) fail(FZ_E_INVALID, "unroll must be <= 32");
   if (reqB != 0 && (reqB % 64 != 0 || reqB > 1024)) fail(FZ_E_INVALID, "block_threads must be a multiple of 64, <= 1024");
   if (const uint32_t W = ws_parts(v.flags)) {
      // W compute waves per 64 streams, each evaluating one part of the serial graph (fz_split.cpp: find_wave_roles), and with
      // FZ_VF_IO_WAVE one more wave for the frame I/O
      const uint32_t waves = ws_waves(v.flags);
      if (!g.wave_roles(W))
         fail(FZ_E_UNSUPPORTED, W == 1 ? "FZ_VF_IO_WAVE: the graph is not stage-packable (1 in, 1 out, register delay lines)"
                                       : "wave split: the graph is not that many groups of isomorphic segments in series (1 in, 1 out, register delay lines)");
      if (reqP > 1) fail(FZ_E_INVALID, "wave split needs streams_per_lane == 1");
      if (reqB && (reqB % 64 || reqB * waves > 1024)) fail(FZ_E_INVALID, "wave split: block_threads counts the streams of a workgroup: a multiple of 64, at most 1024 / waves per tuple");
      if (reqU && reqU != 8 && reqU != 16 && reqU != 32) fail(FZ_E_INVALID, "wave split: unroll must be 8, 16 or 32");
      if (v.flags & (FZ_VF_STREAM_MAJOR | FZ_VF_OUT_F64 | FZ_VF_PREFETCH3))
         fail(FZ_E_UNSUPPORTED, "wave split: time-major / tiled float32 frames, double buffering only");
      v.P = 1;
      v.U = reqU ? reqU : (W == 1 ? 16 : 32);           // (one barrier per round: waves in lockstep do better with longer rounds -- two parts +1.5 %, three +8 %;
                                                        //  the lone compute wave next to an I/O wave keeps 16: its rings fill the LDS at 32)
      // the waves of a workgroup go to consecutive SIMDs of a CU: pairs come two to a workgroup (one wave on each of the
      // four SIMDs), triples and quadruples one
      // (with I/O waves: one compute wave on each SIMD and the I/O waves next to them -- 4 tuples for one part, 2 for two)
      v.block = reqB ? reqB : (ws_io(v.flags) ? (W == 1 ? 256 : W == 2 ? 128 : 64) : (W == 2 ? 128 : 64));
      {  // the rings of a workgroup must fit the CU's LDS: tuples x hand-offs x ring x 1 KiB
         const uint32_t K0 = (*g.wave_roles(W))[0].split.K, ring = (K0 - 1 > 4 ? v.U : v.U / 2), nring = W - 1 + 2 * ws_io(v.flags);
         while ((uint64_t)(v.block / 64) * nring * ring * 1024 > kMaxLdsBytes && !reqB && v.block > 64) v.block /= 2;
         if ((uint64_t)(v.block / 64) * nring * ring * 1024 > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "wave split: the hand-off rings do not fit the LDS with this unroll and block size");
      }
      v.flags &= ~(uint32_t)(FZ_VF_STAGE_PACK | FZ_VF_NO_STAGE_PACK | FZ_VF_SLP);   // (each part is stage-packed by itself)
      return v;
   }
   if (reqP) {
      if (n_streams % reqP) fail(FZ_E_INVALID, "n_streams must be a multiple of streams_per_lane");
      v.P = reqP;
   } else {
      // fill the chip first (256 CUs x 4 SIMDs, several waves each), then pack two streams per lane
      // (v_pk_* issue at the scalar rate on gfx950: twice the lane-ops per cycle)
      // -- unless the frames are already wide (>= 3 wires: 12+ bytes per lane with one stream)
      // (measured crossover on the 6-biquad cascade: 2^18 streams, profiles/r01/sweep_stream_counts.txt)
      v.P = (n_streams >= (1u << 18) && n_streams % 2 == 0 && g.n_in <= 2 && g.n_out <= 2) ? 2 : 1;
   }
   // deep graphs: keep the register-resident delay lines + prefetch buffers inside the 512-entry
   // VGPR/AGPR file (measured: a 24-stage cascade needs ~300 VGPRs at 2 streams per lane)
   uint32_t reg_state = 0;
   for (const Line& l : g.lines)
      if (!l.in_lds) reg_state += l.depth;
   if (!reqP && v.P == 2 && reg_state > 36) v.P = 1;
   // prefetch depth in time steps: 16 rows in flight per lane; 32 once the chip is oversubscribed
   // with packed lanes (fewer, fatter waves: 2 per SIMD)
   // (per-stream coefficients sit in VGPRs too: with 31 of them the deep prefetch costs 10 %)
   const uint32_t reg_values = (reg_state + g.n_param) * v.P;
   v.U = reqU ? reqU : ((v.P == 2 && n_streams >= (1u << 19) && g.n_in == 1 && g.n_out == 1 && reg_values <= 40) ? 32 : 16);
   // wide frames, one stream per lane, chip oversubscribed: 32 rows in flight per lane (measured on three boards, 4-wire
   // frames at 1 M streams: 13.4-14.1 ms against 14.4-14.6 ms with 16; profiles/r02/tune_logs.txt)
   if (!reqU && v.P == 1 && g.n_in >= 3 && n_streams >= (1u << 19) && !(v.flags & FZ_VF_STAGE_PACK)) v.U = 32;
   // (few streams, one stream per lane, no stage packing: 16 against 32 rows is board-dependent -- the fan-out 4-biquad sum at
   //  65 536 streams measured 0.65 / 0.74 of peak on one board and 0.79 / 0.69 on the next; fz_program_tune tries both)
   if (!reqU && reg_state * v.P > 60) v.U = 8;
   if (!g.far_lines.empty()) {
      // far (HBM ring) reads are prefetched one chunk ahead: a read must be two chunks old, so the chunk
      // is at most half the youngest ring read (16 steps from kFarMinDelay = 32 on, 4 for a 9-sample read)
      const uint32_t cap = std::min(16u, std::max(1u, g.far_min_read ? g.far_min_read / 2 : 16u));
      if (reqU > cap) fail(FZ_E_INVALID, "graphs with delays beyond LDS need unroll <= " + std::to_string(cap));
      if (v.flags & FZ_VF_PREFETCH3) fail(FZ_E_INVALID, "FZ_VF_PREFETCH3 is not available with delays beyond LDS");
      v.U = std::min(v.U, cap);
   }
   // wave split: fewer streams than 128 per CU -- W waves per 64 streams, each one part of the serial graph (measured,
   // 6-biquad cascade, ms per 4096 samples: 32 768 streams 0.23 with two parts against 0.31-0.33 with the single wave;
   // 16 384 streams 0.17 with three parts, 0.22 with two, 0.30 single; at 49 152 streams the 384 workgroups of pairs no
   // longer spread evenly over 256 CUs and the single-wave kernel wins again; profiles/r02/sweep_wave_split.txt)
   if (!reqP && !reqB && n_samples >= 256 && (reqU == 0 || reqU == 8 || reqU == 16 || reqU == 32) &&
       !(v.flags & (FZ_VF_STAGE_PACK | FZ_VF_NO_STAGE_PACK | FZ_VF_OUT_F64 | FZ_VF_PREFETCH3 | FZ_VF_STREAM_MAJOR | FZ_VF_SLP))) {
      // the most parts whose waves still find a SIMD each: W waves per 64 streams on 1024 SIMDs, whole workgroups per CU
      // (pairs: two to a workgroup, 128 streams per CU; triples / quadruples: one workgroup of 64 streams per CU)
      uint32_t W = 0;
      if (n_streams <= 16384) W = g.wave_roles(4) ? 4 : g.wave_roles(3) ? 3 : 0;
      if (!W && n_streams <= 32768 && g.wave_roles(2)) W = 2;
      // The splits come with an I/O wave (FZ_VF_IO_WAVE): +2-4 % on every board measured (three parts at 16 384 streams:
      // 0.397 / 0.406 / 0.397 / 0.428 of peak against 0.387 / 0.394 / 0.384 / 0.410; two parts at 32 768: 0.596 / 0.593 against
      // 0.580 / 0.574).  The lone compute wave with an I/O wave at 65 536 streams is NOT a default: +3 % on one board -- 0.374 ms
      // against 0.387 ms per 4096 samples, 97 % of what a plain copy gets there -- and -4 % on the next; fz_program_tune tries
      // it (profiles/r02/sweep_io_wave.txt)
      if (W) {
         fz_variant q{1, reqU, 0, v.flags | (W - 1) << 10 | (W < 4 ? (uint32_t)FZ_VF_IO_WAVE : 0u)};
         return resolve_variant(g, &q, n_streams, n_samples);
      }
   }
   // stage packing: one stream per lane, pairs of isomorphic graph segments in one v_pk_* (fz_split.cpp)
   if (v.flags & FZ_VF_STAGE_PACK) {
      if (!g.split.ok) fail(FZ_E_UNSUPPORTED, "FZ_VF_STAGE_PACK: the graph is not a series of isomorphic segments");
      if (v.P != 1) fail(FZ_E_INVALID, "FZ_VF_STAGE_PACK needs streams_per_lane == 1");
   } else if (!reqP && v.P == 1 && g.split.ok && !(v.flags & FZ_VF_NO_STAGE_PACK) &&
              n_samples >= 32u * (g.split.K - 1)) {
      // automatic below 2^18 streams, unless the block is so short that the K-1 masked steps at
      // either end would dominate
      v.flags |= FZ_VF_STAGE_PACK;
   }
   v.flags &= ~(uint32_t)FZ_VF_NO_STAGE_PACK;
   v.block = reqB ? reqB : 256;
   if (v.flags & FZ_VF_STREAM_MAJOR) {
      // stream-major frames (fz_run_block_stream_major): one stream per lane, chunks of whole float4 pieces
      if (reqP > 2) fail(FZ_E_INVALID, "stream-major frames take one or two streams per lane");
      if (reqU % 4) fail(FZ_E_INVALID, "stream-major frames need unroll % 4 == 0");
      if (!g.far_lines.empty()) fail(FZ_E_UNSUPPORTED, "stream-major frames: delay lines beyond 256 samples are not supported");
      if (v.flags & (FZ_VF_OUT_F64 | FZ_VF_PREFETCH3)) fail(FZ_E_UNSUPPORTED, "stream-major frames: float32 frames, double buffering only");
      // one stream per lane: two (packed FP32) are possible but measured slower everywhere -- twice the
      // patch traffic per wave and 360 VGPRs (profiles/r01/stream_major_kernel.txt)
      v.P = reqP ? reqP : 1u;
      // stage packing (one stream per lane) carries over: the skew only shifts which output chunk a step completes.
      // It is what lifts deep serial graphs off the VALU floor here, whatever the stream count.
      if (v.P != 1 || !g.split.ok) v.flags &= ~(uint32_t)FZ_VF_STAGE_PACK;
      // (automatic only for graphs deep enough to be VALU-bound with one stream per lane: packing takes the in-runs of
      //  the long-run body off the 512-byte grid, which costs ~10 % of the read rate -- measured: a 2-stage cascade runs
      //  5.3-5.9 TB/s unpacked against 4.6-5.5 packed, a 6-stage one 4.6 against 5.3)
      else if (!(uv && (uv->flags & FZ_VF_NO_STAGE_PACK)) && n_samples >= 32u * (g.split.K - 1) && g.n_ops > 27) v.flags |= FZ_VF_STAGE_PACK;
      const uint32_t nw = std::max<uint32_t>(std::max(g.n_in, g.n_out), 1);
      // long-run body (512-byte runs per stream): 1-in/1-out graphs with register-resident state, blocks of at least two phases
      const bool long_ok = g.n_in == 1 && g.n_out == 1 && v.P == 1 && g.n_lds_slots == 0 && (!g.split.ok || g.split.K <= 8);
      const bool want_short = uv && (uv->flags & FZ_VF_SM_SHORT);
      v.flags &= ~(uint32_t)FZ_VF_SM_SHORT;
      if (v.flags & FZ_VF_SM_LONG) {
         if (!long_ok) fail(FZ_E_UNSUPPORTED, "FZ_VF_SM_LONG: needs a 1-in/1-out graph, one stream per lane, no delay lines beyond 8 samples");
         if (reqU && reqU != 64 && reqU != 128) fail(FZ_E_INVALID, "FZ_VF_SM_LONG: unroll must be 64 or 128");
      } else if (long_ok && !want_short && !reqU && n_samples >= 256) {
         v.flags |= FZ_VF_SM_LONG;
      }
      if (v.flags & FZ_VF_SM_LONG) {
         v.U = reqU ? reqU : 128;
         // stage packing rides along when the block is long enough for the masked ends not to matter
         if ((v.flags & FZ_VF_STAGE_PACK) && !g.split.ok) v.flags &= ~(uint32_t)FZ_VF_STAGE_PACK;
         auto lds_long = [&](const Variant& w) { return (uint64_t)(w.block / 64) * 64 * (w.U + 12) * 4; };
         while (lds_long(v) > kMaxLdsBytes && !reqB && v.block > 64) v.block /= 2;
         if (lds_long(v) > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "FZ_VF_SM_LONG: the LDS patches do not fit this block size");
         return v;
      }
      auto lds = [&](const Variant& w) { return (uint64_t)w.block * w.P * (w.U * nw + 4) * 4 + (uint64_t)g.n_lds_slots * w.block * 4 * w.P; };
      if (!reqU) {
         // the longer the run of one stream inside a chunk the better it streams (measured: 128 B per
         // stream and wire 2x faster than 64 B): the deepest chunk whose patches fit the CU's LDS
         v.U = 32;
         while (v.U > 4 && lds(v) > kMaxLdsBytes) v.U /= 2;
      }
      if ((v.flags & FZ_VF_STAGE_PACK) && v.U <= g.split.K - 1) {
         if (uv && (uv->flags & FZ_VF_STAGE_PACK)) fail(FZ_E_INVALID, "stage-packed stream-major frames need unroll > number of segments - 1");
         v.flags &= ~(uint32_t)FZ_VF_STAGE_PACK;
      }
      while (lds(v) > kMaxLdsBytes && !reqB && v.block > 64) v.block /= 2;
      if (lds(v) > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "stream-major frames: the LDS patches do not fit (too many wires per frame)");
      // two streams per lane with patches so large that a single wave fills the CU's LDS: measured 20 x slower than one
      // stream per lane (4-wire frames, 32-sample chunks: 10.3 ms against 0.95 ms) -- refuse instead of crawling
      if (v.P == 2 && (uint64_t)64 * v.P * (v.U * nw + 4) * 4 > kMaxLdsBytes / 4)   // the PATCH of one wave: fewer than one wave per SIMD fit
         fail(FZ_E_UNSUPPORTED, "stream-major frames: two streams per lane leave one wave per CU with this many wires per frame and this "
                                "unroll; use one stream per lane or a shorter unroll");
      return v;
   }
   if (g.n_lds_slots) {
      // LDS rings: slots * block * 4P bytes must fit the CU's 160 KiB of LDS (one workgroup may take it all)
      auto bytes = [&](const Variant& w) { return (uint64_t)g.n_lds_slots * w.block * 4u * w.P; };
      while (bytes(v) > kMaxLdsBytes && !reqB && v.block > 64) v.block /= 2;
      while (bytes(v) > kMaxLdsBytes && !reqP && v.P > 1) v.P /= 2;
      if (bytes(v) > kMaxLdsBytes)
         fail(FZ_E_UNSUPPORTED, "delay lines too long for the LDS ring buffers of this build (" +
                                   std::to_string(g.n_lds_slots) + " slots)");
   }
   return v;
}

// ---- persisted plans ---------------------------------------------------------------------------------------------
// fz_program_tune's winner is remembered across processes: <kernel cache>/plans.txt, one line per
// (graph structure, n_streams, tile_streams, board) -- the board by its UUID, because the winner differs from board to
// board.  A launch without a variant consults it once per shape.  FLOWZ_HIP_NO_PLAN_CACHE=1 turns it off.
uint64_t graph_structure_hash(const Graph& g)
{
   std::ostringstream o;
   o << g.n_in << ' ' << g.n_out << ' ' << g.n_param << ' ' << (g.typed ? 1 : 0) << '|';
   for (const Node& n : g.nodes) o << n.kind << ',' << n.a << ',' << n.b << ',' << n.c << ',' << (n.f64 ? 1 : 0) << ';';
   o << '|';
   for (uint32_t v : g.outputs) o << v << ',';
   o << '|';
   for (const Line& l : g.lines) o << l.src << ':' << l.depth << ':' << (l.f64 ? 1 : 0) << ',';
   return fnv1a(o.str());
}

static std::string board_id()
{
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess) return "";
   hipUUID uuid;
   if (hipDeviceGetUuid(&uuid, dev) == hipSuccess) {
      char buf[40];
      for (int i = 0; i < 16; ++i) std::snprintf(buf + 2 * i, 3, "%02x", (unsigned)(unsigned char)uuid.bytes[i]);
      return buf;
   }
   (void)hipGetLastError();
   return "dev" + std::to_string(dev);
}

static bool plan_cache_on() { return !std::getenv("FLOWZ_HIP_NO_PLAN_CACHE") && !std::getenv("FLOWZ_HIP_NO_CACHE"); }

static void plan_store(const fz_program* p, uint64_t n_streams, uint32_t tile, const fz_variant& v, float ms)
{
   const std::string dir = cache_dir(), id = board_id();
   if (!plan_cache_on() || dir.empty() || id.empty()) return;
   ::mkdir(dir.c_str(), 0755);
   char line[256];
   const int n = std::snprintf(line, sizeof line, "%016llx %llu %u %s %u %u %u %u %.5f\n", (unsigned long long)p->graph_hash,
                               (unsigned long long)n_streams, tile, id.c_str(), v.streams_per_lane, v.unroll, v.block_threads, v.flags, ms);
   if (n <= 0 || n >= (int)sizeof line) return;
   if (FILE* f = std::fopen((dir + "/plans.txt").c_str(), "a")) {     // one short append per tune: later lines win
      std::fwrite(line, 1, (size_t)n, f);
      std::fclose(f);
   }
}

static bool plan_load(const fz_program* p, uint64_t n_streams, uint32_t tile, fz_variant* out)
{
   const std::string dir = cache_dir(), id = board_id();
   if (!plan_cache_on() || dir.empty() || id.empty()) return false;
   std::ifstream f(dir + "/plans.txt");
   if (!f) return false;
   bool found = false;
   std::string ln;
   while (std::getline(f, ln)) {
      unsigned long long h = 0, ns = 0;
      unsigned t = 0, P = 0, U = 0, B = 0, fl = 0;
      char idbuf[64] = {0};
      float ms = 0.f;
      if (std::sscanf(ln.c_str(), "%llx %llu %u %63s %u %u %u %u %f", &h, &ns, &t, idbuf, &P, &U, &B, &fl, &ms) != 9) continue;
      if (h != p->graph_hash || ns != n_streams || t != tile || id != idbuf) continue;
      if ((P != 0 && P != 1 && P != 2 && P != 4) || U > 128 || B > 1024 || (B % 64)) continue;   // (a damaged line)
      *out = fz_variant{P, U, B, fl};
      found = true;
   }
   return found;
}

fz_variant planned_variant(fz_program* p, uint64_t n_streams, uint32_t tile_streams)
{
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   if (tile_streams >= n_streams) tile_streams = 0;
   const auto key = std::make_tuple(n_streams, tile_streams, dev);
   {
      std::lock_guard<std::mutex> lock(p->mu);
      auto it = p->plans.find(key);
      if (it != p->plans.end()) return it->second;
      if (!p->plan_looked_up.insert(key).second) return fz_variant{0, 0, 0, 0};
   }
   fz_variant v{0, 0, 0, 0};
   if (plan_load(p, n_streams, tile_streams, &v) && (v.streams_per_lane || v.unroll || v.block_threads || v.flags)) {
      std::lock_guard<std::mutex> lock(p->mu);
      p->plans[key] = v;
      return v;
   }
   return fz_variant{0, 0, 0, 0};
}

// ---- launch ------------------------------------------------------------------------------------------------
struct ArgsHeader {
   const float* in;
   float* out;
   float* state;
   const float* params;
   const float* mod;
   unsigned long long n_streams;
   unsigned int n_samples;
   unsigned int n_groups;
   unsigned int tile_streams;
   unsigned int tile_blocks;
   unsigned int rows_total;
   unsigned int row0;
   unsigned int mod_stride;
   unsigned int pad_;
};
static_assert(sizeof(ArgsHeader) % 8 == 0 && sizeof(ArgsHeader) == 5 * 8 + 8 + 8 * 4, "ArgsHeader must match the head of the kernel's fz_args without padding");

int launch(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
           uint32_t n_samples, const fz_variant* uv, void* stream, uint32_t tile_streams, uint32_t rows_total, uint32_t row0)
{
   if (rows_total == 0) rows_total = n_samples;             // the block is the whole buffer
   if ((uint64_t)row0 + n_samples > rows_total) fail(FZ_E_INVALID, "row0 + n_samples exceeds rows_total");
   const bool stream_major = uv && (uv->flags & FZ_VF_STREAM_MAJOR);
   if (stream_major) {
      if (tile_streams) fail(FZ_E_INVALID, "stream-major frames are not tiled");
      if ((uint64_t)rows_total * std::max(p->g.n_in, p->g.n_out) >= (1ull << 24))
         fail(FZ_E_UNSUPPORTED, "stream-major frames: more than 2^24 floats per stream buffer (a wave's 64 rows are addressed through one 4 GiB descriptor): use a window");
      if (((uint64_t)rows_total * p->g.n_in) % 4 || ((uint64_t)row0 * p->g.n_in) % 4 || ((uint64_t)rows_total * p->g.n_out) % 4 ||
          ((uint64_t)row0 * p->g.n_out) % 4)
         fail(FZ_E_INVALID, "stream-major frames: rows_total and row0 times the wires per frame must be multiples of 4 floats");
   }
   const Graph& g = p->g;
   if (n_streams == 0 || n_samples == 0) return FZ_OK;      // an empty block: nothing to evaluate, state unchanged
   if (n_samples == 0xFFFFFFFFu) fail(FZ_E_INVALID, "n_samples must be below 2^32 - 1");
   if (!out) fail(FZ_E_INVALID, "out is null");
   if (g.n_in && !in) fail(FZ_E_INVALID, "in is null but the graph has input wires");
   if (g.n_state && !state) fail(FZ_E_INVALID, "state is null but the graph has delay lines");
   if (g.n_param && !params) fail(FZ_E_INVALID, "params is null but the graph has per-stream coefficients");
   auto mis = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) != 0; };
   if (mis(in) || mis(out) || mis(state) || mis(params)) fail(FZ_E_INVALID, "device pointers must be 16-byte aligned");
   const uint64_t wmax = std::max<uint64_t>(std::max(g.n_in, g.n_out), 1);
   if (tile_streams == 0 || tile_streams >= n_streams) tile_streams = 0;    // one tile == plain time-major
   const uint64_t row_streams = tile_streams ? tile_streams : n_streams;
   const uint64_t out_w = (uint64_t)std::max<uint32_t>(g.n_out, 1) * ((uv && (uv->flags & FZ_VF_OUT_F64)) ? 2 : 1);
   if (!stream_major && row_streams * std::max(wmax, out_w) >= (1ull << 30)) fail(FZ_E_UNSUPPORTED, "row longer than 4 GiB: shard or tile the streams");
   if (n_streams >= (1ull << 32)) fail(FZ_E_UNSUPPORTED, "more than 2^32 streams per launch: shard the streams");
   if (tile_streams && n_streams % tile_streams) fail(FZ_E_INVALID, "n_streams must be a multiple of tile_streams");
   require_device();
   fz_variant planned;
   if (!uv) {                                               // a measured plan for this shape on this device?
      int dev = 0;
      FZ_HIP(hipGetDevice(&dev));
      const auto key = std::make_tuple(n_streams, tile_streams, dev);
      bool known = false;
      (void)planned_variant(p, n_streams, tile_streams);      // first launch of this shape: a plan persisted by an earlier process?
      {
         std::lock_guard<std::mutex> lock(p->mu);
         auto it = p->plans.find(key);
         known = it != p->plans.end() || p->tuned_default.count(key) != 0;
         if (it != p->plans.end()) {
            planned = it->second;
            uv = &planned;
         }
      }
      // FLOWZ_HIP_AUTOTUNE=1: the first big block of a shape measures the plan by itself (on the caller's
      // buffers; the state is saved and restored around the measurement, `out` is recomputed below)
      static const bool autotune = [] { const char* e = std::getenv("FLOWZ_HIP_AUTOTUNE"); return e && *e && *e != '0'; }();
      bool can_tune = autotune && !known && rows_total == n_samples && row0 == 0 && n_streams * (uint64_t)n_samples >= (1ull << 26);
      if (can_tune) {
         // not while the stream is being captured into a hipGraph (the measurement allocates and synchronises), and not
         // in place: the candidates run on the caller's buffers, an aliased `in` would be overwritten before the real launch
         hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
         if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) (void)hipGetLastError();
         const char* ib = reinterpret_cast<const char*>(in);
         const char* ob = reinterpret_cast<const char*>(out);
         const size_t ibytes = (size_t)n_streams * n_samples * g.n_in * 4, obytes = (size_t)n_streams * n_samples * out_w * 4;
         const bool overlap = in && ib < ob + obytes && ob < ib + ibytes;
         can_tune = cap == hipStreamCaptureStatusNone && !overlap;
      }
      if (can_tune) {
         {
            std::lock_guard<std::mutex> lock(p->mu);
            p->tuned_default.insert(key);                   // (also stops the recursion through tune -> launch)
         }
         const size_t sb = (size_t)g.n_state * n_streams * 4;
         // the state is saved before and restored after the measurement ON EVERY EXIT PATH
         struct Saved {
            float* copy = nullptr;
            float* state;
            size_t bytes;
            hipStream_t st;
            ~Saved()
            {
               if (!copy) return;
               (void)hipMemcpyAsync(state, copy, bytes, hipMemcpyDeviceToDevice, st);
               (void)hipStreamSynchronize(st);
               (void)hipFree(copy);
            }
         } saved{nullptr, state, sb, (hipStream_t)stream};
         bool have_copy = true;
         if (sb) {
            if (hipMalloc((void**)&saved.copy, sb) != hipSuccess) {      // no room for the snapshot (multi-GiB state): do not tune
               (void)hipGetLastError();
               saved.copy = nullptr;
               have_copy = false;
            } else {
               FZ_HIP(hipMemcpyAsync(saved.copy, state, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream));
            }
         }
         if (have_copy) {
            fz_variant chosen{0, 0, 0, 0};
            const int rc = tune(p, in, out, state, params, n_streams, n_samples, tile_streams, stream, &chosen, nullptr);
            if (rc != FZ_OK) return rc;
            if (chosen.streams_per_lane || chosen.unroll || chosen.block_threads || chosen.flags) {
               planned = chosen;
               uv = &planned;
            }
         }
      }
   }
   Variant v = resolve_variant(g, uv, n_streams, n_samples);
   // time-major frames of many streams: the rows are megabytes apart, every row in flight is another page, and 16 rows per
   // lane do better than 32 (1 M streams: 6.46 ms against 6.88 ms; stream-tiled frames keep 32: tools/slab_probe.py)
   if (!tile_streams && !(v.flags & FZ_VF_STREAM_MAJOR) && !(uv && uv->unroll) && v.P == 2 && v.U == 32 && n_streams >= (1u << 19)) v.U = 16;
   if (tile_streams) {
      // a workgroup must not straddle tiles: shrink the lane packing / block until it divides
      const bool fixedP = uv && uv->streams_per_lane, fixedB = uv && uv->block_threads;
      while (tile_streams % (v.P * v.block) && !fixedP && v.P > 1) v.P /= 2;
      while (tile_streams % (v.P * v.block) && !fixedB && v.block > 64) v.block /= 2;
      if (tile_streams % (v.P * v.block))
         fail(FZ_E_INVALID, "tile_streams must be a multiple of streams_per_lane * block_threads");
   }
   {  // a chunk of U rows is addressed through ONE buffer descriptor: it must stay below 4 GiB
      const uint64_t row_bytes = stream_major ? 0 : row_streams * std::max(wmax, out_w) * 4;
      while (row_bytes * v.U >= (1ull << 32) && v.U > 1) {
         if (uv && uv->unroll) fail(FZ_E_INVALID, "unroll x row bytes must stay below 4 GiB: lower the unroll or tile the streams");
         v.U /= 2;
      }
   }
   v = settle_variant(p, v);
   void* fn = nullptr;
   auto k = get_kernel(p, v, &fn);

   // kernarg image of `struct fz_args` (8-byte aligned: pad the coefficient tail)
   // (built on the stack for ordinary graphs: no allocation on the launch path)
   const size_t off64 = (sizeof(ArgsHeader) + sizeof(float) * std::max<size_t>(g.consts.size(), 1) + 7) & ~size_t(7);
   const size_t kbytes = off64 + sizeof(double) * std::max<size_t>(g.consts64.size(), 1);
   alignas(8) char small[1024];
   std::vector<char> big;
   char* const kbuf = kbytes <= sizeof small ? small : (big.resize(kbytes), big.data());
   const float* mod_dev = nullptr;
   uint32_t mod_stride = 0;
   if (g.n_mod) {
      std::lock_guard<std::mutex> lock(p->mu);
      mod_dev = p->mod_dev;
      mod_stride = p->mod_stride;
      if (!mod_dev) fail(FZ_E_INVALID, "the graph has sample-rate modulators: call fz_program_set_modulation first");
      if (mod_stride < rows_total) fail(FZ_E_INVALID, "fz_program_set_modulation: stride is shorter than the rows of this launch");
   }
   ArgsHeader h{in, out, state, params, mod_dev, (unsigned long long)n_streams, n_samples, (unsigned int)(n_streams / v.P),
                (unsigned int)row_streams, tile_streams ? (unsigned int)(tile_streams / (v.P * v.block)) : 0u, rows_total, row0, mod_stride, 0u};
   std::memcpy(kbuf, &h, sizeof h);
   {
      std::lock_guard<std::mutex> lock(p->mu);
      if (!g.consts.empty()) std::memcpy(kbuf + sizeof h, g.consts.data(), sizeof(float) * g.consts.size());
      if (!g.consts64.empty()) std::memcpy(kbuf + off64, g.consts64.data(), sizeof(double) * g.consts64.size());
   }
   size_t size = kbytes;
   void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, kbuf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
   const unsigned grid = (unsigned)((h.n_groups + v.block - 1) / v.block);
   // (wave split: v.block counts the 64 streams of a workgroup; two waves evaluate them)
   const unsigned threads = ws_parts(v.flags) ? v.block * ws_waves(v.flags) : v.block;
   FZ_HIP(hipModuleLaunchKernel((hipFunction_t)fn, grid, 1, 1, threads, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
   static const bool debug = std::getenv("FLOWZ_HIP_DEBUG") != nullptr;
   if (debug) {
      FZ_HIP(hipStreamSynchronize((hipStream_t)stream));
      std::fprintf(stderr, "[flowz_hip] launched grid=%u block=%u P=%u U=%u flags=%u n_streams=%llu n_samples=%u kernarg=%zu B vgprs=%u scratch=%u B/lane\n",
                   grid, v.block, v.P, v.U, v.flags, (unsigned long long)n_streams, n_samples, size, k->res.vgprs + k->res.agprs, k->res.scratch_bytes);
   }
   return FZ_OK;
}

// the variants fz_program_tune measures for a shape (the first one is the library default)
std::vector<fz_variant> tune_candidates(const Graph& g, uint64_t n_streams, uint32_t n_samples)
{
   const Variant d = resolve_variant(g, nullptr, n_streams, n_samples);
   std::vector<fz_variant> cands{fz_variant{0, 0, 0, 0}};
   if (const uint32_t W = ws_parts(d.flags)) {            // few streams: wave splits, with and without an I/O wave, against the single stage-packed wave
      const uint32_t wbits = (W - 1) << 10;
      cands.push_back(fz_variant{1, 0, 0, wbits | (ws_io(d.flags) ? 0u : (uint32_t)FZ_VF_IO_WAVE)});   // the same split without / with the I/O wave
      cands.push_back(fz_variant{1, 16, 0, wbits | (d.flags & FZ_VF_IO_WAVE)});
      if (W > 2 && g.wave_roles(2)) cands.push_back(fz_variant{1, 16, 0, FZ_VF_WAVE_SPLIT});
      cands.push_back(fz_variant{1, 16, 0, FZ_VF_STAGE_PACK});
   } else if (d.flags & FZ_VF_STAGE_PACK) {
      if (n_streams <= 65536 && g.wave_roles(1)) cands.push_back(fz_variant{1, 16, 0, FZ_VF_IO_WAVE});   // one compute + one I/O wave per 64 streams
      cands.push_back(fz_variant{1, 24, 0, FZ_VF_STAGE_PACK});
      cands.push_back(fz_variant{1, 32, 0, FZ_VF_STAGE_PACK});
   } else if (d.P == 2) {            // many streams, narrow frames: lane packing x prefetch depth x workgroups per CU
      cands.push_back(fz_variant{2, 16, 0, 0});
      cands.push_back(fz_variant{4, 8, 0, 0});
      cands.push_back(fz_variant{2, 16, 256, FZ_VF_MAX_WG(2)});
      cands.push_back(fz_variant{2, 32, 256, FZ_VF_MAX_WG(2)});
      cands.push_back(fz_variant{4, 8, 256, FZ_VF_MAX_WG(1)});
      cands.push_back(fz_variant{4, 12, 256, FZ_VF_MAX_WG(1)});
      cands.push_back(fz_variant{4, 16, 256, FZ_VF_MAX_WG(1)});
      cands.push_back(fz_variant{4, 8, 128, FZ_VF_MAX_WG(2)});
      cands.push_back(fz_variant{4, 4, 256, FZ_VF_MAX_WG(2)});
   } else {
      cands.push_back(fz_variant{1, 32, 0, 0});
   }
   if (!(d.flags & FZ_VF_STAGE_PACK) && d.P != 2 && n_streams >= (1u << 17)) {
      cands.push_back(fz_variant{1, 16, 256, FZ_VF_MAX_WG(1)});
      cands.push_back(fz_variant{1, 8, 256, FZ_VF_MAX_WG(1)});
      cands.push_back(fz_variant{1, 16, 256, FZ_VF_MAX_WG(2)});
      if (n_streams % 2 == 0 && g.n_in <= 2 && g.n_out <= 2) cands.push_back(fz_variant{2, 16, 0, 0});
   }
   return cands;
}

// ---- plan selection ------------------------------------------------------------------------------------------
// The variants differ by a few percent, and which one wins depends on the board (measured: the same
// variant is +5 % on one MI355X of the pool and -3 % on the next), so -- like FFTW_MEASURE -- time the
// candidates on the caller's own buffers once and remember the winner for this shape.
int tune(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
         uint32_t n_samples, uint32_t tile_streams, void* stream, fz_variant* chosen, float* chosen_ms)
{
   const Graph& g = p->g;
   if (!n_streams || !n_samples) fail(FZ_E_INVALID, "fz_program_tune: empty block");
   require_device();
   if (tile_streams == 0 || tile_streams >= n_streams) tile_streams = 0;
   std::vector<fz_variant> cands = tune_candidates(g, n_streams, n_samples);
   hipEvent_t e0, e1;
   FZ_HIP(hipEventCreate(&e0));
   FZ_HIP(hipEventCreate(&e1));
   float best_ms = 0.f, default_ms = 0.f;
   int best = -1;
   std::string first_error;
   // the default is measured twice: the first pass only brings the clocks and the memory system up to
   // speed (whoever runs first would otherwise look slower than it is)
   for (size_t cc = 0; cc <= cands.size(); ++cc) {
      const bool warmup = cc == 0;
      const size_t c = warmup ? 0 : cc - 1;
      try {
         launch(p, in, out, state, params, n_streams, n_samples, &cands[c], stream, tile_streams);   // build, load, first touch
         // one launch to size the measurement (>= ~25 ms of kernel time: sub-millisecond kernels need
         // dozens of launches before their timing settles), then the measurement proper
         float ms = 0.f;
         int reps = 1;
         for (int pass = 0; pass < 2; ++pass) {
            FZ_HIP(hipEventRecord(e0, (hipStream_t)stream));
            for (int r = 0; r < reps; ++r) launch(p, in, out, state, params, n_streams, n_samples, &cands[c], stream, tile_streams);
            FZ_HIP(hipEventRecord(e1, (hipStream_t)stream));
            FZ_HIP(hipEventSynchronize(e1));
            FZ_HIP(hipEventElapsedTime(&ms, e0, e1));
            ms /= (float)reps;
            if (pass == 0) reps = std::max(3, std::min(100, (int)(25.f / std::max(ms, 1e-3f))));
         }
         if (std::getenv("FLOWZ_HIP_DEBUG") || std::getenv("FLOWZ_HIP_TUNE_LOG"))
            std::fprintf(stderr, "[flowz_hip] tune %s n_streams=%llu tile=%u: P=%u U=%u block=%u flags=%u: %.4f ms%s\n",
                         kernel_name(g, resolve_variant(g, &cands[c], n_streams, n_samples)).c_str(), (unsigned long long)n_streams,
                         tile_streams, cands[c].streams_per_lane, cands[c].unroll, cands[c].block_threads, cands[c].flags, ms,
                         warmup ? " (warm-up pass)" : "");
         if (!warmup && c == 0) default_ms = ms;
         if (!warmup && (best < 0 || ms < best_ms)) {
            best = (int)c;
            best_ms = ms;
         }
      } catch (const Error& er) {                          // a candidate this graph / shape does not allow
         if (er.code == FZ_E_HIP || er.code == FZ_E_NO_DEVICE) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            throw;
         }
         if (first_error.empty()) first_error = er.msg;
      }
   }
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   if (best < 0) fail(FZ_E_INVALID, "fz_program_tune: no variant could run: " + first_error);
   // repeated measurements of one variant scatter by 1-2 %: a candidate replaces the library default only when it wins by more
   if (best > 0 && default_ms > 0.f && best_ms > 0.985f * default_ms) {
      best = 0;
      best_ms = default_ms;
   }
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   {
      std::lock_guard<std::mutex> lock(p->mu);
      if (best == 0) p->plans.erase(std::make_tuple(n_streams, tile_streams, dev));
      else p->plans[std::make_tuple(n_streams, tile_streams, dev)] = cands[(size_t)best];
      p->plan_looked_up.insert(std::make_tuple(n_streams, tile_streams, dev));
   }
   plan_store(p, n_streams, tile_streams, cands[(size_t)best], best_ms);
   if (chosen) *chosen = cands[(size_t)best];
   if (chosen_ms) *chosen_ms = best_ms;
   return FZ_OK;
}

// ---- AOT utility kernels ---------------------------------------------------------------------------------------
typedef float fzr_f4 __attribute__((ext_vector_type(4)));

// reactive_equations/reactive_filter_coeff.cpp:38-58, one stream per thread
__global__ void __launch_bounds__(256) fz_rbj_lowpass_kernel(const float* __restrict__ freq, const float* __restrict__ q, float sr,
                                                             unsigned long long n, float* raw6, float* df1)
{
   const unsigned long long s = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
   if (s >= n) return;
   const float two_pi = (float)(8. * 0.78539816339744830962);     // const float two_pi = 8. * std::atan(1.)
   const float w0 = two_pi * freq[s] / sr;
   const float cosw0 = (float)cos((double)w0);                      // std::cos(float) to within 1 ULP
   const float sinw0 = (float)sin((double)w0);
   const float alpha = (float)(sinw0 / (2. * q[s]));
   const float b0 = (float)((1. - cosw0) / 2.);
   const float b1 = (float)(1. - cosw0);
   const float b2 = (float)((1. - cosw0) / 2.);
   const float a0 = (float)(1. + alpha);
   const float a1 = (float)(-2. * cosw0);
   const float a2 = (float)(1. - alpha);
   if (raw6) {
      raw6[0 * n + s] = a0; raw6[1 * n + s] = a1; raw6[2 * n + s] = a2;
      raw6[3 * n + s] = b0; raw6[4 * n + s] = b1; raw6[5 * n + s] = b2;
   }
   if (df1) {
      df1[0 * n + s] = b0 / a0; df1[1 * n + s] = b1 / a0; df1[2 * n + s] = b2 / a0;
      df1[3 * n + s] = -a1 / a0; df1[4 * n + s] = -a2 / a0;
   }
}

__device__ __forceinline__ unsigned fmix32(unsigned h)
{
   h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
   return h;
}

// dst[t][s][w] for one row t per blockIdx.y; a thread produces 4 consecutive floats of the row
__global__ void __launch_bounds__(256) fz_synth_fill_kernel(float* dst, unsigned long long row_floats, unsigned n_wires,
                                                            unsigned seed, unsigned long long stream0,
                                                            unsigned long long t0, unsigned n_rows,
                                                            unsigned long long tile_floats)
{
   // row_floats = n_streams * n_wires of the logical time-major row; tile_floats = floats of one
   // tile's row segment (== row_floats when untiled).  Logical element i of row t is stored at
   // (i / tile_floats) * n_rows * tile_floats + t * tile_floats + i % tile_floats.
   const unsigned long long i0 = ((unsigned long long)blockIdx.x * 256u + threadIdx.x) * 4ull;
   if (i0 >= row_floats) return;
   const unsigned long long tl = i0 / tile_floats, within = i0 - tl * tile_floats;
   for (unsigned t = blockIdx.y; t < n_rows; t += gridDim.y) {
      const unsigned tt = (unsigned)((t0 + t) * 0x85EBCA6Bull);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
         const unsigned long long sid = stream0 * n_wires + i0 + j;     // (stream0+s)*n_wires + w
         const unsigned h = fmix32(fmix32(seed ^ (unsigned)(sid * 0x9E3779B9ull) ^ tt));
         v[j] = (float)(int)(h >> 8) * 0x1p-23f - 1.0f;
      }
      float* row = dst + (size_t)tl * n_rows * tile_floats + (size_t)t * tile_floats;
      if (i0 + 4 <= row_floats && (tile_floats & 3ull) == 0) {   // segments stay 16-byte aligned
         fzr_f4 q = {v[0], v[1], v[2], v[3]};
         __builtin_nontemporal_store(q, reinterpret_cast<fzr_f4*>(row + within));
      } else {
         for (int j = 0; j < 4 && i0 + j < row_floats; ++j) {
            const unsigned long long i = i0 + j, tj = i / tile_floats;
            dst[(size_t)tj * n_rows * tile_floats + (size_t)t * tile_floats + (i - tj * tile_floats)] = v[j];
         }
      }
   }
}

// one-shot float4 copy, four independent nt loads in flight per lane before the stores: the fastest
// plain copy of profiles/r01/hbm_copy_patterns_microbench.txt (5.8-6.0 TB/s; a 2048-block grid-stride
// loop and hipMemcpyDtoD stay at 4.8-4.9)
__global__ void __launch_bounds__(256) fz_copy_kernel(const fzr_f4* __restrict__ src, fzr_f4* __restrict__ dst,
                                                      unsigned long long n4)
{
   const unsigned long long base = (unsigned long long)blockIdx.x * 1024u + threadIdx.x;
   fzr_f4 v[4];
#pragma unroll
   for (int k = 0; k < 4; ++k)
      if (base + 256u * k < n4) v[k] = __builtin_nontemporal_load(src + base + 256u * k);
#pragma unroll
   for (int k = 0; k < 4; ++k)
      if (base + 256u * k < n4) __builtin_nontemporal_store(v[k], dst + base + 256u * k);
}

// Stream-major <-> frame layout adapter.  Callers of the reference hold one contiguous sample buffer
// per closure ([stream][t][wire], the loop of test/benchmark.cpp:137-147); the block kernel wants
// frames with the stream index fastest ([t][stream][wire], optionally tiled).  One workgroup moves a
// 64-stream x CT-column patch (CT = whole frames, <= 64 floats) through LDS so that both the reads and
// the writes are contiguous runs: rows of the stream-major side, (stream, wire) runs of the frame side.
template <bool TO_STREAM_MAJOR>
__global__ void __launch_bounds__(256) fz_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           unsigned long long n_streams, unsigned n_samples, unsigned W,
                                                           unsigned tile_streams, unsigned nt /* frames per patch */,
                                                           unsigned gx, unsigned gy)
{
   __shared__ float patch[64][65];
   // workgroups that run at the same time cover a 16 x 16 block of patches, so that each side sees
   // 4 KiB runs (16 patches x 256 B) instead of isolated 256 B pieces
   constexpr unsigned SX = 16, SY = 16;
   const unsigned sbx = (gx + SX - 1) / SX;
   const unsigned long long b = blockIdx.x;
   const unsigned long long sup = b / (SX * SY);
   const unsigned within = (unsigned)(b % (SX * SY));
   const unsigned px = (unsigned)(sup % sbx) * SX + within % SX, py = (unsigned)(sup / sbx) * SY + within / SX;
   if (px >= gx || py >= gy) return;
   const unsigned long long s0 = (unsigned long long)px * 64u;
   const unsigned t0 = py * nt;
   const unsigned tid = threadIdx.x;
   const unsigned long long TW = (unsigned long long)n_samples * W;
   const unsigned ns_here = (unsigned)(n_streams - s0 < 64u ? n_streams - s0 : 64u);
   const unsigned nt_here = n_samples - t0 < nt ? n_samples - t0 : nt;
   // frame side: element (t, s, w) at fbase + (t0 + t) * row_streams * W + s * W + w
   const unsigned long long tile = tile_streams ? s0 / tile_streams : 0u;
   const unsigned long long row_streams = tile_streams ? tile_streams : n_streams;
   const unsigned long long s_in_tile = tile_streams ? s0 % tile_streams : s0;
   const unsigned long long fbase = tile * (unsigned long long)n_samples * row_streams * W + s_in_tile * W;
   const unsigned run = 64u * W;                                     // floats of one frame row of the patch
   // stream-major side: element (s, c) at (s0 + s) * TW + t0 * W + c, c < nt * W
   // full patches of 16-byte-aligned layouts move as float4 (all 4 loads of a thread in flight at once);
   // edge patches and odd wire counts take the scalar path
   const bool vec = nt * W == 64u && ns_here == 64u && nt_here == nt && (TW & 3u) == 0 && ((row_streams * W) & 3u) == 0;
   if (vec) {
      const unsigned q = tid & 15u, r0 = tid >> 4;                  // float4 column, first row
      if (!TO_STREAM_MAJOR) {
         fzr_f4 v[4];
#pragma unroll
         for (int k = 0; k < 4; ++k)
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const fzr_f4*>(src + (s0 + r0 + 16u * k) * TW + (unsigned long long)t0 * W) + q);
#pragma unroll
         for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) patch[r0 + 16u * k][q * 4u + j] = v[k][j];
         __syncthreads();
         // frame rows: nt rows of `run` floats; float4 index over the whole patch output
         for (unsigned e = tid; e < nt * run / 4u; e += 256u) {
            const unsigned t = e / (run / 4u), r = (e - t * (run / 4u)) * 4u;
            fzr_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
               const unsigned rr = r + j, sl = rr / W, w = rr - sl * W;
               o[j] = patch[sl][t * W + w];
            }
            __builtin_nontemporal_store(o, reinterpret_cast<fzr_f4*>(dst + fbase + (unsigned long long)(t0 + t) * row_streams * W + r));
         }
      } else {
         for (unsigned e = tid; e < nt * run / 4u; e += 256u) {
            const unsigned t = e / (run / 4u), r = (e - t * (run / 4u)) * 4u;
            const fzr_f4 o = __builtin_nontemporal_load(reinterpret_cast<const fzr_f4*>(src + fbase + (unsigned long long)(t0 + t) * row_streams * W + r));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
               const unsigned rr = r + j, sl = rr / W, w = rr - sl * W;
               patch[sl][t * W + w] = o[j];
            }
         }
         __syncthreads();
#pragma unroll
         for (int k = 0; k < 4; ++k) {
            fzr_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = patch[r0 + 16u * k][q * 4u + j];
            __builtin_nontemporal_store(o, reinterpret_cast<fzr_f4*>(dst + (s0 + r0 + 16u * k) * TW + (unsigned long long)t0 * W) + q);
         }
      }
      return;
   }
   if (!TO_STREAM_MAJOR) {
      for (unsigned e = tid; e < 64u * 64u; e += 256u) {
         const unsigned sl = e >> 6, c = e & 63u;
         if (sl < ns_here && c < nt_here * W) patch[sl][c] = __builtin_nontemporal_load(src + (s0 + sl) * TW + (unsigned long long)t0 * W + c);
      }
      __syncthreads();
      for (unsigned e = tid; e < nt * run; e += 256u) {
         const unsigned t = e / run, r = e - t * run, sl = r / W, w = r - sl * W;
         if (t < nt_here && sl < ns_here)
            __builtin_nontemporal_store(patch[sl][t * W + w], dst + fbase + (unsigned long long)(t0 + t) * row_streams * W + r);
      }
   } else {
      for (unsigned e = tid; e < nt * run; e += 256u) {
         const unsigned t = e / run, r = e - t * run, sl = r / W, w = r - sl * W;
         if (t < nt_here && sl < ns_here)
            patch[sl][t * W + w] = __builtin_nontemporal_load(src + fbase + (unsigned long long)(t0 + t) * row_streams * W + r);
      }
      __syncthreads();
      for (unsigned e = tid; e < 64u * 64u; e += 256u) {
         const unsigned sl = e >> 6, c = e & 63u;
         if (sl < ns_here && c < nt_here * W) __builtin_nontemporal_store(patch[sl][c], dst + (s0 + sl) * TW + (unsigned long long)t0 * W + c);
      }
   }
}

}  // namespace fz

using namespace fz;

#define FZ_GUARD(...)                                                           \
   try { __VA_ARGS__ }                                                                 \
   catch (const fz::Error& er) { fz::set_error(er.msg); return er.code; }       \
   catch (const std::exception& ex) { fz::set_error(ex.what()); return FZ_E_INVALID; }

struct fz_bank {
   fz_program* prog = nullptr;
   uint64_t n_streams = 0;
   int device = 0;              // the bank's buffers live on this device
   float* state = nullptr;
   float* params = nullptr;
   float* stage_in = nullptr;
   float* stage_out = nullptr;
   size_t stage_in_cap = 0, stage_out_cap = 0;
   // streams / events of the pipelined host path (created on first use)
   hipStream_t s_h2d = nullptr, s_run = nullptr, s_d2h = nullptr;
   hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
};

extern "C" {

int fz_device_count(void) { return fz::device_count(); }

int fz_synth_fill(float* dst, uint64_t n_streams, uint32_t n_samples, uint32_t n_wires, uint32_t seed,
                  uint64_t stream0, uint64_t t0, uint32_t tile_streams, void* hip_stream)
{
   FZ_GUARD(
      if (!dst || !n_streams || !n_samples || !n_wires) fail(FZ_E_INVALID, "fz_synth_fill: bad arguments");
      require_device();
      const unsigned long long row = n_streams * n_wires;
      if (tile_streams && n_streams % tile_streams) fail(FZ_E_INVALID, "n_streams must be a multiple of tile_streams");
      const unsigned long long tile_floats = (tile_streams && tile_streams < n_streams) ? (unsigned long long)tile_streams * n_wires : row;
      dim3 grid((unsigned)((row + 1023) / 1024), std::min<uint32_t>(n_samples, 64u));
      hipLaunchKernelGGL(fz_synth_fill_kernel, grid, dim3(256), 0, (hipStream_t)hip_stream, dst, row, n_wires, seed,
                         (unsigned long long)stream0, (unsigned long long)t0, n_samples, tile_floats);
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

int fz_rbj_lowpass(const float* freq, const float* q, float sample_rate, uint64_t n_streams, float* raw6, float* df1,
                   void* hip_stream)
{
   FZ_GUARD(
      if (!freq || !q || !n_streams || (!raw6 && !df1)) fail(FZ_E_INVALID, "fz_rbj_lowpass: bad arguments");
      require_device();
      hipLaunchKernelGGL(fz_rbj_lowpass_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                         freq, q, sample_rate, (unsigned long long)n_streams, raw6, df1);
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

int fz_copy_probe(const float* src, float* dst, uint64_t n_floats, void* hip_stream)
{
   FZ_GUARD(
      if (!src || !dst || (n_floats & 3)) fail(FZ_E_INVALID, "fz_copy_probe: need non-null pointers and n_floats % 4 == 0");
      require_device();
      const unsigned long long n4 = n_floats / 4;
      if (n4 > 1024ull * 0x7FFFFFFFull) fail(FZ_E_INVALID, "fz_copy_probe: buffer too large");
      hipLaunchKernelGGL(fz_copy_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)hip_stream,
                         (const fzr_f4*)src, (fzr_f4*)dst, (unsigned long long)(n_floats / 4));
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

int fz_transpose_frames(const float* src, float* dst, uint64_t n_streams, uint32_t n_samples, uint32_t n_wires,
                        uint32_t tile_streams, int to_stream_major, void* hip_stream)
{
   FZ_GUARD(
      if (!src || !dst || !n_streams || !n_samples || !n_wires) fail(FZ_E_INVALID, "fz_transpose_frames: bad arguments");
      if (n_wires > 64) fail(FZ_E_UNSUPPORTED, "fz_transpose_frames: more than 64 wires per frame");
      if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) fail(FZ_E_INVALID, "device pointers must be 16-byte aligned");
      if (tile_streams >= n_streams) tile_streams = 0;
      if (tile_streams && (tile_streams % 64 || n_streams % tile_streams))
         fail(FZ_E_INVALID, "tile_streams must be a multiple of 64 and divide n_streams");
      require_device();
      const unsigned nt = 64u / n_wires;
      const uint64_t gx = (n_streams + 63) / 64, gy = ((uint64_t)n_samples + nt - 1) / nt;
      const uint64_t blocks = ((gx + 15) / 16) * ((gy + 15) / 16) * 256;
      if (gx > 0xFFFFFFFFull || blocks > 0x7FFFFFFFull) fail(FZ_E_UNSUPPORTED, "fz_transpose_frames: too many patches for one launch: split the block");
      if (to_stream_major)
         hipLaunchKernelGGL(fz_transpose_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src, dst,
                            (unsigned long long)n_streams, n_samples, n_wires, tile_streams, nt, (unsigned)gx, (unsigned)gy);
      else
         hipLaunchKernelGGL(fz_transpose_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src, dst,
                            (unsigned long long)n_streams, n_samples, n_wires, tile_streams, nt, (unsigned)gx, (unsigned)gy);
      FZ_HIP(hipGetLastError());
      return FZ_OK;)
}

// ---- fz_bank -------------------------------------------------------------------------------------------------
static void check_bank_device(const fz_bank* b)
{
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   if (dev != b->device)
      fail(FZ_E_INVALID, "the bank's buffers live on device " + std::to_string(b->device) + ", the current device is " + std::to_string(dev));
}

int fz_bank_create(fz_program* p, uint64_t n_streams, fz_bank** out)
{
   FZ_GUARD(
      if (!p || !out || !n_streams) fail(FZ_E_INVALID, "fz_bank_create: bad arguments");
      require_device();
      std::unique_ptr<fz_bank, void (*)(fz_bank*)> guard(new fz_bank(), fz_bank_destroy);
      fz_bank* b = guard.get();
      b->prog = p;
      b->n_streams = n_streams;
      FZ_HIP(hipGetDevice(&b->device));
      const size_t sb = std::max<size_t>((size_t)p->g.n_state * n_streams * 4, 16);
      FZ_HIP(hipMalloc((void**)&b->state, sb));
      FZ_HIP(hipMemset(b->state, 0, sb));                 // zero-initialised float state, flowz.hpp:1245
      if (p->g.n_param) {
         const size_t pb = (size_t)p->g.n_param * n_streams * 4;
         FZ_HIP(hipMalloc((void**)&b->params, pb));
         FZ_HIP(hipMemset(b->params, 0, pb));
      }
      *out = guard.release();
      return FZ_OK;)
}

int fz_bank_clone(const fz_bank* src, fz_bank** out)
{
   FZ_GUARD(
      if (!src || !out) fail(FZ_E_INVALID, "fz_bank_clone: bad arguments");
      check_bank_device(src);
      fz_bank* b = nullptr;
      int rc = fz_bank_create(src->prog, src->n_streams, &b);
      if (rc != FZ_OK) return rc;
      std::unique_ptr<fz_bank, void (*)(fz_bank*)> guard(b, fz_bank_destroy);
      const size_t sb = (size_t)src->prog->g.n_state * src->n_streams * 4;
      if (sb) FZ_HIP(hipMemcpy(b->state, src->state, sb, hipMemcpyDeviceToDevice));
      if (src->params)
         FZ_HIP(hipMemcpy(b->params, src->params, (size_t)src->prog->g.n_param * src->n_streams * 4, hipMemcpyDeviceToDevice));
      *out = guard.release();
      return FZ_OK;)
}

void fz_bank_destroy(fz_bank* b)
{
   if (!b) return;
   (void)hipFree(b->state);
   (void)hipFree(b->params);
   (void)hipFree(b->stage_in);
   (void)hipFree(b->stage_out);
   for (hipStream_t st : {b->s_h2d, b->s_run, b->s_d2h})
      if (st) (void)hipStreamDestroy(st);
   for (int i = 0; i < 2; ++i)
      for (hipEvent_t e : {b->ev_in[i], b->ev_run[i], b->ev_out[i]})
         if (e) (void)hipEventDestroy(e);
   delete b;
}

int fz_bank_reset(fz_bank* b)
{
   FZ_GUARD(
      if (!b) fail(FZ_E_INVALID, "null bank");
      check_bank_device(b);
      const size_t sb = (size_t)b->prog->g.n_state * b->n_streams * 4;
      if (sb) FZ_HIP(hipMemset(b->state, 0, sb));
      return FZ_OK;)
}

int fz_bank_set_params_host(fz_bank* b, const float* params)
{
   FZ_GUARD(
      if (!b || !params) fail(FZ_E_INVALID, "fz_bank_set_params_host: bad arguments");
      if (!b->params) fail(FZ_E_INVALID, "graph has no per-stream coefficients");
      check_bank_device(b);
      FZ_HIP(hipMemcpy(b->params, params, (size_t)b->prog->g.n_param * b->n_streams * 4, hipMemcpyHostToDevice));
      return FZ_OK;)
}

float* fz_bank_state_device(fz_bank* b) { return b ? b->state : nullptr; }

int fz_bank_process(fz_bank* b, const float* in_dev, float* out_dev, uint32_t n_samples, const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!b) fail(FZ_E_INVALID, "null bank");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      return fz::launch(b->prog, in_dev, out_dev, g.n_state ? b->state : nullptr, b->params, b->n_streams, n_samples, v, hip_stream, 0);)
}

int fz_bank_process_tiled(fz_bank* b, const float* in_dev, float* out_dev, uint32_t n_samples, uint32_t tile_streams,
                          const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!b) fail(FZ_E_INVALID, "null bank");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      return fz::launch(b->prog, in_dev, out_dev, g.n_state ? b->state : nullptr, b->params, b->n_streams, n_samples, v,
                        hip_stream, tile_streams);)
}

int fz_bank_process_stream_major(fz_bank* b, const float* in_dev, float* out_dev, uint32_t rows_total, uint32_t row0, uint32_t n_samples,
                                 const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!b || !rows_total) fail(FZ_E_INVALID, "fz_bank_process_stream_major: bad arguments");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      fz_variant sm = v ? *v : fz_variant{0, 0, 0, 0};
      sm.flags |= FZ_VF_STREAM_MAJOR;
      return fz::launch(b->prog, in_dev, out_dev, g.n_state ? b->state : nullptr, b->params, b->n_streams, n_samples, &sm, hip_stream, 0,
                        rows_total, row0);)
}

int fz_bank_process_blocks(fz_bank* b, const float* in_dev, float* out_dev, uint32_t rows_total, uint32_t block_len,
                           const float* params_blocks, uint32_t tile_streams, const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!b || !rows_total || !block_len) fail(FZ_E_INVALID, "fz_bank_process_blocks: bad arguments");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      if (params_blocks && !g.n_param) fail(FZ_E_INVALID, "the graph has no per-stream coefficients");
      const size_t pstride = (size_t)g.n_param * b->n_streams;      // floats of one block's coefficient set
      uint32_t k = 0;
      for (uint32_t row0 = 0; row0 < rows_total; row0 += block_len, ++k) {
         const uint32_t n = std::min(block_len, rows_total - row0);
         const float* pr = params_blocks ? params_blocks + (size_t)k * pstride : b->params;
         int rc = fz::launch(b->prog, in_dev, out_dev, g.n_state ? b->state : nullptr, pr, b->n_streams, n, v, hip_stream,
                             tile_streams, rows_total, row0);
         if (rc != FZ_OK) return rc;
      }
      return FZ_OK;)
}

int fz_bank_tune(fz_bank* b, const float* in_dev, float* out_dev, uint32_t n_samples, uint32_t tile_streams, void* hip_stream,
                 fz_variant* chosen, float* chosen_ms)
{
   FZ_GUARD(
      if (!b) fail(FZ_E_INVALID, "null bank");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      return fz::tune(b->prog, in_dev, out_dev, g.n_state ? b->state : nullptr, b->params, b->n_streams, n_samples, tile_streams,
                      hip_stream, chosen, chosen_ms);)
}

static void ensure_stage(fz_bank* b, size_t ib, size_t ob)
{
   if (ib > b->stage_in_cap) {
      (void)hipFree(b->stage_in);
      b->stage_in = nullptr;
      b->stage_in_cap = 0;
      FZ_HIP(hipMalloc((void**)&b->stage_in, ib));
      b->stage_in_cap = ib;
   }
   if (ob > b->stage_out_cap) {
      (void)hipFree(b->stage_out);
      b->stage_out = nullptr;
      b->stage_out_cap = 0;
      FZ_HIP(hipMalloc((void**)&b->stage_out, ob));
      b->stage_out_cap = ob;
   }
}

static void drain_pipeline(fz_bank* b)
{
   for (hipStream_t st : {b->s_h2d, b->s_run, b->s_d2h})
      if (st) (void)hipStreamSynchronize(st);
}

// Host frames in, host frames out.  Short blocks (the per-sample call protocol) take one synchronous
// H2D / kernel / D2H round trip.  Long blocks are cut along TIME into chunks that flow through a
// three-stage pipeline on three HIP streams -- H2D of chunk k+1, the kernel of chunk k and D2H of chunk
// k-1 overlap (time-major frames: a time chunk is contiguous; the recurrence only orders the kernels,
// which run back to back on one stream).  With pinned host memory (hipHostMalloc / hipHostRegister /
// torch pin_memory) both PCIe directions run concurrently; pageable memory still works, HIP then
// stages the copies itself.
static int bank_process_host(fz_bank* b, const float* in_host, void* out_host, uint32_t n_samples, bool f64)
{
   FZ_GUARD(
      if (!b || !out_host || !n_samples) fail(FZ_E_INVALID, "fz_bank_process_host: bad arguments");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      const size_t irow = (size_t)b->n_streams * g.n_in * 4, orow = (size_t)b->n_streams * g.n_out * (f64 ? 8 : 4);
      if (irow && !in_host) fail(FZ_E_INVALID, "in_host is null but the graph has input wires");
      const fz_variant v64{0, 0, 0, FZ_VF_OUT_F64};
      const fz_variant* uv = f64 ? &v64 : nullptr;
      constexpr size_t kChunkBytes = 32u << 20;            // per direction and pipeline slot
      const size_t row = std::max(irow, orow);
      uint32_t chunk_t = (uint32_t)std::max<size_t>(1, kChunkBytes / std::max<size_t>(row, 1));
      if ((size_t)n_samples * row <= 2 * kChunkBytes || chunk_t >= n_samples) {                    // one round trip
         ensure_stage(b, irow * n_samples, orow * n_samples);
         if (irow) FZ_HIP(hipMemcpy(b->stage_in, in_host, irow * n_samples, hipMemcpyHostToDevice));
         int rc = fz::launch(b->prog, g.n_in ? b->stage_in : nullptr, b->stage_out, g.n_state ? b->state : nullptr, b->params,
                             b->n_streams, n_samples, uv, nullptr, 0);
         if (rc != FZ_OK) return rc;
         FZ_HIP(hipMemcpy(out_host, b->stage_out, orow * n_samples, hipMemcpyDeviceToHost));
         return FZ_OK;
      }
      if (chunk_t >= 64) chunk_t &= ~31u;                   // whole prefetch chunks
      // the second pipeline slot must start 16-byte aligned whatever n_streams and chunk_t are
      const size_t islot = (irow * chunk_t + 255) & ~size_t(255), oslot = (orow * chunk_t + 255) & ~size_t(255);
      ensure_stage(b, 2 * islot, 2 * oslot);
      if (!b->s_h2d) {
         FZ_HIP(hipStreamCreateWithFlags(&b->s_h2d, hipStreamNonBlocking));
         FZ_HIP(hipStreamCreateWithFlags(&b->s_run, hipStreamNonBlocking));
         FZ_HIP(hipStreamCreateWithFlags(&b->s_d2h, hipStreamNonBlocking));
         for (int i = 0; i < 2; ++i) {
            FZ_HIP(hipEventCreateWithFlags(&b->ev_in[i], hipEventDisableTiming));
            FZ_HIP(hipEventCreateWithFlags(&b->ev_run[i], hipEventDisableTiming));
            FZ_HIP(hipEventCreateWithFlags(&b->ev_out[i], hipEventDisableTiming));
         }
      }
      FZ_HIP(hipDeviceSynchronize());                       // the bank's state may still be in use on other streams
      const char* hin = reinterpret_cast<const char*>(in_host);
      char* hout = reinterpret_cast<char*>(out_host);
      uint32_t k = 0;
      for (uint32_t t0 = 0; t0 < n_samples; t0 += chunk_t, ++k) {
         const uint32_t nt = std::min(chunk_t, n_samples - t0);
         const int slot = (int)(k & 1u);
         float* din = reinterpret_cast<float*>(reinterpret_cast<char*>(b->stage_in) + (size_t)slot * islot);
         float* dout = reinterpret_cast<float*>(reinterpret_cast<char*>(b->stage_out) + (size_t)slot * oslot);
         if (irow) {
            if (k >= 2) FZ_HIP(hipStreamWaitEvent(b->s_h2d, b->ev_run[slot], 0));        // kernel k-2 has consumed this slot
            FZ_HIP(hipMemcpyAsync(din, hin + (size_t)t0 * irow, irow * nt, hipMemcpyHostToDevice, b->s_h2d));
            FZ_HIP(hipEventRecord(b->ev_in[slot], b->s_h2d));
            FZ_HIP(hipStreamWaitEvent(b->s_run, b->ev_in[slot], 0));
         }
         if (k >= 2) FZ_HIP(hipStreamWaitEvent(b->s_run, b->ev_out[slot], 0));            // D2H k-2 has drained this slot
         int rc = FZ_OK;
         try {
            rc = fz::launch(b->prog, g.n_in ? din : nullptr, dout, g.n_state ? b->state : nullptr, b->params, b->n_streams, nt, uv,
                            b->s_run, 0);
         } catch (...) {                                    // chunks already in flight still write into the caller's memory
            drain_pipeline(b);
            throw;
         }
         if (rc != FZ_OK) {
            drain_pipeline(b);
            return rc;
         }
         FZ_HIP(hipEventRecord(b->ev_run[slot], b->s_run));
         FZ_HIP(hipStreamWaitEvent(b->s_d2h, b->ev_run[slot], 0));
         FZ_HIP(hipMemcpyAsync(hout + (size_t)t0 * orow, dout, orow * nt, hipMemcpyDeviceToHost, b->s_d2h));
         FZ_HIP(hipEventRecord(b->ev_out[slot], b->s_d2h));
      }
      FZ_HIP(hipStreamSynchronize(b->s_d2h));
      FZ_HIP(hipStreamSynchronize(b->s_run));
      return FZ_OK;)
}

int fz_bank_process_host(fz_bank* b, const float* in_host, float* out_host, uint32_t n_samples)
{
   return bank_process_host(b, in_host, out_host, n_samples, false);
}

// Host buffers in the reference's own calling convention: one contiguous sample buffer per stream
// ([n_streams][n_samples][wires]).  Time chunks travel as 2-D copies (one row per stream) into compact
// device patches [n_streams][chunk][wires], run through the stream-major kernel and travel back; the
// same three-stream pipeline as the frame path.
int fz_bank_process_host_stream_major(fz_bank* b, const float* in_host, float* out_host, uint32_t n_samples)
{
   FZ_GUARD(
      if (!b || !out_host || !n_samples) fail(FZ_E_INVALID, "fz_bank_process_host_stream_major: bad arguments");
      check_bank_device(b);
      const Graph& g = b->prog->g;
      if (g.n_in && !in_host) fail(FZ_E_INVALID, "in_host is null but the graph has input wires");
      const uint32_t nw = std::max<uint32_t>(std::max(g.n_in, g.n_out), 1);
      constexpr size_t kChunkBytes = 32u << 20;
      uint32_t chunk_t = (uint32_t)std::max<size_t>(32, kChunkBytes / ((size_t)b->n_streams * nw * 4) / 32 * 32);
      chunk_t = std::min(chunk_t, (n_samples + 31u) / 32u * 32u);
      const size_t ipitch = (size_t)chunk_t * g.n_in * 4, opitch = (size_t)chunk_t * g.n_out * 4;     // device rows
      const size_t hip = (size_t)n_samples * g.n_in * 4, hop = (size_t)n_samples * g.n_out * 4;       // host rows
      ensure_stage(b, 2 * ipitch * b->n_streams, 2 * opitch * b->n_streams);
      if (!b->s_h2d) {
         FZ_HIP(hipStreamCreateWithFlags(&b->s_h2d, hipStreamNonBlocking));
         FZ_HIP(hipStreamCreateWithFlags(&b->s_run, hipStreamNonBlocking));
         FZ_HIP(hipStreamCreateWithFlags(&b->s_d2h, hipStreamNonBlocking));
         for (int i = 0; i < 2; ++i) {
            FZ_HIP(hipEventCreateWithFlags(&b->ev_in[i], hipEventDisableTiming));
            FZ_HIP(hipEventCreateWithFlags(&b->ev_run[i], hipEventDisableTiming));
            FZ_HIP(hipEventCreateWithFlags(&b->ev_out[i], hipEventDisableTiming));
         }
      }
      FZ_HIP(hipDeviceSynchronize());
      const fz_variant sm{0, 0, 0, FZ_VF_STREAM_MAJOR};
      const char* hin = reinterpret_cast<const char*>(in_host);
      char* hout = reinterpret_cast<char*>(out_host);
      uint32_t k = 0;
      for (uint32_t t0 = 0; t0 < n_samples; t0 += chunk_t, ++k) {
         const uint32_t nt = std::min(chunk_t, n_samples - t0);
         const int slot = (int)(k & 1u);
         float* din = reinterpret_cast<float*>(reinterpret_cast<char*>(b->stage_in) + (size_t)slot * ipitch * b->n_streams);
         float* dout = reinterpret_cast<float*>(reinterpret_cast<char*>(b->stage_out) + (size_t)slot * opitch * b->n_streams);
         if (g.n_in) {
            if (k >= 2) FZ_HIP(hipStreamWaitEvent(b->s_h2d, b->ev_run[slot], 0));
            FZ_HIP(hipMemcpy2DAsync(din, ipitch, hin + (size_t)t0 * g.n_in * 4, hip, (size_t)nt * g.n_in * 4, b->n_streams,
                                    hipMemcpyHostToDevice, b->s_h2d));
            FZ_HIP(hipEventRecord(b->ev_in[slot], b->s_h2d));
            FZ_HIP(hipStreamWaitEvent(b->s_run, b->ev_in[slot], 0));
         }
         if (k >= 2) FZ_HIP(hipStreamWaitEvent(b->s_run, b->ev_out[slot], 0));
         int rc = FZ_OK;
         try {
            rc = fz::launch(b->prog, g.n_in ? din : nullptr, dout, g.n_state ? b->state : nullptr, b->params, b->n_streams, nt, &sm,
                            b->s_run, 0, chunk_t, 0);
         } catch (...) {
            drain_pipeline(b);
            throw;
         }
         if (rc != FZ_OK) {
            drain_pipeline(b);
            return rc;
         }
         FZ_HIP(hipEventRecord(b->ev_run[slot], b->s_run));
         FZ_HIP(hipStreamWaitEvent(b->s_d2h, b->ev_run[slot], 0));
         FZ_HIP(hipMemcpy2DAsync(hout + (size_t)t0 * g.n_out * 4, hop, dout, opitch, (size_t)nt * g.n_out * 4, b->n_streams,
                                 hipMemcpyDeviceToHost, b->s_d2h));
         FZ_HIP(hipEventRecord(b->ev_out[slot], b->s_d2h));
      }
      FZ_HIP(hipStreamSynchronize(b->s_d2h));
      FZ_HIP(hipStreamSynchronize(b->s_run));
      return FZ_OK;)
}

int fz_bank_process_host_f64(fz_bank* b, const float* in_host, double* out_host, uint32_t n_samples)
{
   return bank_process_host(b, in_host, out_host, n_samples, true);
}

}  // extern "C"
