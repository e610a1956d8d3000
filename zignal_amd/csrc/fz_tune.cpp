// Measured plans: what fz_program_tune measures (tune_candidates), the measurement, and the winners' persistence per board.
// The static choice they compete with: fz_plan.cpp.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "fz_runtime.hpp"

namespace fz {

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

static bool plan_cache_on()
{
   const char* off = std::getenv("FLOWZ_HIP_NO_PLAN_CACHE");             // (set to anything but "" / "0": neither read nor written)
   return !(off && *off && std::strcmp(off, "0") != 0) && !std::getenv("FLOWZ_HIP_NO_CACHE");
}

// One line per tune: "<format tag> <graph hash> <n_streams> <tile> <board> <P> <U> <block> <flags> <ms> <n_samples>".  The tag names
// the layout of the line AND of the flag bits (they were re-assigned between rounds): lines with another tag are not ours to read.
static const char kPlanTag[] = "fzplan3";

static void plan_store(const fz_program* p, uint64_t n_streams, uint32_t n_samples, uint32_t tile, const fz_variant& v, float ms)
{
   const std::string dir = cache_dir(), id = board_id();
   if (!plan_cache_on() || dir.empty() || id.empty()) return;
   ::mkdir(dir.c_str(), 0755);
   char line[256];
   const int n = std::snprintf(line, sizeof line, "%s %016llx %llu %u %s %u %u %u %u %.5f %u\n", kPlanTag, (unsigned long long)p->graph_hash,
                               (unsigned long long)n_streams, tile, id.c_str(), v.streams_per_lane, v.unroll, v.block_threads, v.flags, ms, n_samples);
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
      unsigned t = 0, P = 0, U = 0, B = 0, fl = 0, T = 0;
      char idbuf[64] = {0}, tag[16] = {0};
      float ms = 0.f;
      if (std::sscanf(ln.c_str(), "%15s %llx %llu %u %63s %u %u %u %u %f %u", tag, &h, &ns, &t, idbuf, &P, &U, &B, &fl, &ms, &T) != 11) continue;
      if (std::strcmp(tag, kPlanTag) != 0 || h != p->graph_hash || ns != n_streams || t != tile || id != idbuf) continue;
      if ((P != 0 && P != 1 && P != 2 && P != 4) || U > 128 || B > 1024 || (B % 64)) continue;   // (a damaged line)
      // only what tune_candidates can emit: a stale or damaged line must not turn a default launch into another LAYOUT or output type
      // (FZ_VF_STREAM_MAJOR / FZ_VF_OUT_F64 would write past a float32 time-major `out`)
      constexpr unsigned kPlanFlags = FZ_VF_STAGE_PACK | FZ_VF_WAVE_SPLIT | FZ_VF_WAVE_SPLIT3 | FZ_VF_IO_WAVE | FZ_VF_IO_WAVE2 | FZ_VF_LOCKSTEP | FZ_VF_GRID_SYNC | FZ_VF_PREFETCH3 | FZ_VF_MAX_WG(7);
      if (fl & ~kPlanFlags) continue;
      *out = fz_variant{P, U, B, fl};
      found = true;
   }
   return found;
}

// a plan that no longer resolves (a stale or damaged line that passed the range checks): forget it for this process
void drop_plan(fz_program* p, uint64_t n_streams, uint32_t tile_streams)
{
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess) return;
   if (tile_streams >= n_streams) tile_streams = 0;
   std::lock_guard<std::mutex> lock(p->mu);
   p->plans.erase(std::make_tuple(n_streams, tile_streams, dev));
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

// the variants fz_program_tune measures for a shape (the first one is the library default).  Round 6 kept only candidates that a bench line
// or a sweep of rounds 3-5 saw ahead of the default on some board (profiles/NOTES.md "Tune candidates"): 26 of them left.
std::vector<fz_variant> tune_candidates(const Graph& g, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams)
{
   if (tile_streams >= n_streams) tile_streams = 0;
   const Variant d = resolve_variant(g, nullptr, n_streams, n_samples, tile_streams);
   std::vector<fz_variant> cands{fz_variant{0, 0, 0, 0}};
   const uint32_t LG = FZ_VF_LOCKSTEP | FZ_VF_GRID_SYNC;
   if (const uint32_t W = ws_parts(d.flags)) {
      if (W == 1) {                                       // one wave per SIMD (config 2), two I/O waves by default: the lone stage-packed wave, one I/O wave, a packed pair per wave
         cands.push_back(fz_variant{1, 16, 0, FZ_VF_STAGE_PACK});
         cands.push_back(fz_variant{1, 16, 0, FZ_VF_IO_WAVE});
         const uint32_t Wp = g.split.K / 2;
         if (n_streams % 256 == 0 && Wp >= 2 && Wp <= 4 && g.wave_roles(Wp)) cands.push_back(fz_variant{1, 16, 256, FZ_VF_WAVES(Wp)});
         cands.push_back(fz_variant{1, 24, 0, FZ_VF_STAGE_PACK});
      } else {                                            // few streams: the same split without / with the I/O wave, and the single stage-packed wave
         cands.push_back(fz_variant{1, 0, 0, ((W - 1) << 10) | (ws_io(d.flags) ? 0u : (uint32_t)FZ_VF_IO_WAVE)});
         cands.push_back(fz_variant{1, 16, 0, FZ_VF_STAGE_PACK});
      }
   } else if (d.flags & FZ_VF_STAGE_PACK) {               // one stream per lane, stage-packed (65 536 < streams < 2^18, short blocks): longer chunks
      cands.push_back(fz_variant{1, 24, 0, FZ_VF_STAGE_PACK});
   } else if ((d.flags & FZ_VF_LOCKSTEP) && g.n_lds_slots) {   // LDS rings in step: against the free-running four-wave workgroups with 32-row chunks
      cands.push_back(fz_variant{1, 32, 256, 0});
   } else if ((d.flags & FZ_VF_LOCKSTEP) && tile_streams) {   // tiles in step: against two / any number of free-running workgroups per CU
      cands.push_back(fz_variant{2, 16, 256, FZ_VF_MAX_WG(2)});
      cands.push_back(fz_variant{2, 16, 256, 0});
   } else if (d.flags & FZ_VF_LOCKSTEP) {                 // plain rows, many streams: the walk in lockstep against its neighbours in the geometry table
      const uint32_t G = d.flags & FZ_VF_GRID_SYNC;
      cands.push_back(fz_variant{std::min(d.P, 2u), 16, 256, 0});                                  // four-wave workgroups running free
      if (!(d.flags & FZ_VF_STAGE_PACK)) {
         for (uint32_t P : {d.P * 2, d.P / 2}) {                                                    // the next packing up and down, each at its own geometry
            if (P < 1 || P > 4 || n_streams % P) continue;
            const TmGeometry o = time_major_geometry(n_streams, P, g.n_ops > 30, false, P);
            if (o.P != P || o.main_streams != n_streams) continue;
            cands.push_back(fz_variant{P, o.U, o.lanes, FZ_VF_LOCKSTEP | G | (o.U == 1 ? (uint32_t)FZ_VF_PREFETCH3 : 0u)});
         }
         if (g.split.ok && n_samples >= 16u * (g.split.atoms() - 1)) cands.push_back(fz_variant{1, 4, 1024, FZ_VF_LOCKSTEP | G | FZ_VF_STAGE_PACK});   // (register-heavy graphs)
      }
   } else if (d.P == 2) {                                 // stream tiles, many streams: workgroups per CU, and the lockstep geometries on tiles
      cands.push_back(fz_variant{2, 16, 256, (d.flags & FZ_VF_MAX_WG(7)) ? 0u : FZ_VF_MAX_WG(2)});
      if (tile_streams && tile_streams % 2048 == 0) cands.push_back(fz_variant{2, 2, 1024, LG});
      if (tile_streams && tile_streams % 4096 == 0 && n_streams % 4 == 0) cands.push_back(fz_variant{4, 1, 1024, LG | FZ_VF_PREFETCH3});
      if (tile_streams && tile_streams % 1024 == 0) cands.push_back(fz_variant{1, 4, 1024, LG});   // (register-heavy graphs: +3.6 % for the oscillator chain)
      cands.push_back(fz_variant{4, 8, 256, FZ_VF_MAX_WG(1)});
   } else {                                               // one stream per lane, free-running (wide frames on tiles, rings, short blocks)
      cands.push_back(fz_variant{1, d.U == 32 ? 16u : 32u, 0, 0});
      if (!g.n_lds_slots && n_streams >= (1u << 18) && (!tile_streams || tile_streams % 1024 == 0)) cands.push_back(fz_variant{1, 4, 1024, LG});
      if (!g.n_lds_slots && n_streams >= (1u << 17) && n_streams % 2 == 0 && g.n_in <= 2 && g.n_out <= 2) cands.push_back(fz_variant{2, 16, 0, 0});
   }
   return cands;
}

// ---- plan selection ------------------------------------------------------------------------------------------
// The variants differ by a few percent, and which one wins depends on the board (measured: the same
// variant is +5 % on one MI355X of the pool and -3 % on the next), so -- like FFTW_MEASURE -- time the
// candidates on the caller's own buffers once and remember the winner for this shape.
// implicit: the measurement a first big launch makes by itself -- only candidates whose code object is at hand (in memory or in
// the on-disk cache) take part: a launch never waits for hiprtc builds of kernels nobody asked for (fz_program_tune builds them all)
int tune(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
         uint32_t n_samples, uint32_t tile_streams, void* stream, fz_variant* chosen, float* chosen_ms, bool implicit)
{
   const Graph& g = p->g;
   if (!n_streams || !n_samples) fail(FZ_E_INVALID, "fz_program_tune: empty block");
   require_device();
   if (tile_streams == 0 || tile_streams >= n_streams) tile_streams = 0;
   std::vector<fz_variant> cands = tune_candidates(g, n_streams, n_samples, tile_streams);
   hipEvent_t e0, e1;
   FZ_HIP(hipEventCreate(&e0));
   FZ_HIP(hipEventCreate(&e1));
   float best_ms = 0.f, default_ms = 0.f;
   int best = -1;
   std::string first_error;
   // The boards are power-managed: a kernel that sits at the package power cap (the 6-biquad cascade at 1 M streams does,
   // profiles/r03/power_and_clocks.txt) runs its first hundred milliseconds at a higher clock than it sustains, so whoever is
   // measured first looks faster than it is.  Hence: a warm-up of the default (>= 100 ms), then TWO passes over the candidates,
   // forwards and backwards -- every candidate is measured at the same mean position -- and the two times of a candidate averaged.
   std::vector<int> reps_of(cands.size(), 0);                       // 0: not measured (not at hand / not allowed / scratch)
   std::vector<float> ms_sum(cands.size(), 0.f);
   auto timed = [&](size_t c, int reps) {
      float ms = 0.f;
      FZ_HIP(hipEventRecord(e0, (hipStream_t)stream));
      for (int r = 0; r < reps; ++r) launch(p, in, out, state, params, n_streams, n_samples, &cands[c], stream, tile_streams);
      FZ_HIP(hipEventRecord(e1, (hipStream_t)stream));
      FZ_HIP(hipEventSynchronize(e1));
      FZ_HIP(hipEventElapsedTime(&ms, e0, e1));
      return ms / (float)reps;
   };
   const bool log = std::getenv("FLOWZ_HIP_DEBUG") || std::getenv("FLOWZ_HIP_TUNE_LOG");
   for (int pass = 0; pass < 2; ++pass) {
      for (size_t k = 0; k < cands.size(); ++k) {
         const size_t c = pass == 0 ? k : cands.size() - 1 - k;
         if (pass == 1 && reps_of[c] == 0) continue;
         try {
            // (implicit: a candidate takes part only when every kernel its resolution touches -- the variant itself and whatever a
            //  spilling one settles down to -- is in memory or in the on-disk cache; nothing is built for the measurement)
            NoJitScope no_jit(implicit && c != 0);
            if (pass == 0) {
               // (a candidate that would run from scratch memory even with its unroll lowered -- a 1024-lane lockstep workgroup of a
               //  register-heavy graph -- is not measured: no kernel of this library runs from scratch, see DESIGN "Register budget")
               if (c != 0 && get_kernel(p, finalize_variant(p, &cands[c], n_streams, n_samples, tile_streams), nullptr)->res.scratch_bytes != 0) continue;
               launch(p, in, out, state, params, n_streams, n_samples, &cands[c], stream, tile_streams);   // build, load, first touch
               // one launch to size the measurement (>= ~25 ms of kernel time: sub-millisecond kernels need dozens of launches
               // before their timing settles)
               const float ms1 = timed(c, 1);
               reps_of[c] = std::max(3, std::min(100, (int)(25.f / std::max(ms1, 1e-3f))));
               if (c == 0) {                                          // the warm-up: clocks and memory system up to speed, power settled
                  const int wreps = std::max(reps_of[c], std::min(400, (int)(100.f / std::max(ms1, 1e-3f))));
                  const float wms = timed(c, wreps);
                  if (log) std::fprintf(stderr, "[flowz_hip] tune warm-up: %d launches of the default, %.4f ms each\n", wreps, wms);
               }
            }
            const float ms = timed(c, reps_of[c]);
            ms_sum[c] += ms;
            if (log)
               std::fprintf(stderr, "[flowz_hip] tune %s n_streams=%llu tile=%u: P=%u U=%u block=%u flags=%u: %.4f ms (pass %d)\n",
                            kernel_name(g, finalize_variant(p, &cands[c], n_streams, n_samples, tile_streams)).c_str(), (unsigned long long)n_streams,
                            tile_streams, cands[c].streams_per_lane, cands[c].unroll, cands[c].block_threads, cands[c].flags, ms, pass + 1);
         } catch (const Error& er) {                          // a candidate this graph / shape does not allow
            // a HIP error (no memory for a candidate's arrival counters ...) takes the measurement down only when it is the
            // DEFAULT that failed: the caller's launch would fail the same way.  Any other candidate is skipped.
            if ((er.code == FZ_E_HIP && c == 0) || er.code == FZ_E_NO_DEVICE) {
               (void)hipEventDestroy(e0);
               (void)hipEventDestroy(e1);
               throw;
            }
            if (er.code == FZ_E_HIP) (void)hipGetLastError();
            reps_of[c] = 0;
            if (first_error.empty()) first_error = er.msg;
         }
      }
   }
   for (size_t c = 0; c < cands.size(); ++c) {
      if (reps_of[c] == 0) continue;
      const float ms = 0.5f * ms_sum[c];
      if (c == 0) default_ms = ms;
      if (best < 0 || ms < best_ms) {
         best = (int)c;
         best_ms = ms;
      }
   }
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   if (best < 0) fail(FZ_E_INVALID, "fz_program_tune: no variant could run: " + first_error);
   // repeated measurements of one variant scatter by 1-2 % (more on a board at its power cap): a candidate replaces the INCUMBENT --
   // the plan this program already runs this shape with on this device, else the library default -- only when it wins by more
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   int incumbent = 0;
   {
      std::lock_guard<std::mutex> lock(p->mu);
      auto it = p->plans.find(std::make_tuple(n_streams, tile_streams, dev));
      if (it != p->plans.end())
         for (size_t c = 1; c < cands.size(); ++c)
            if (reps_of[c] != 0 && cands[c].streams_per_lane == it->second.streams_per_lane && cands[c].unroll == it->second.unroll &&
                cands[c].block_threads == it->second.block_threads && cands[c].flags == it->second.flags)
               incumbent = (int)c;
   }
   // (the measurement a first launch makes by itself asks for 3 %: it runs once, at whatever moment the caller's first block comes, and
   //  round 5's bench lines caught it replacing the default by a kernel that an explicit fz_program_tune, minutes later on the same
   //  buffers, found 5 % SLOWER -- the fan-out sum on one board; what it is there for, the I/O waves of config 2 and the like, wins by more)
   const float margin = implicit ? 0.97f : 0.985f;
   const float incumbent_ms = reps_of[(size_t)incumbent] != 0 ? 0.5f * ms_sum[(size_t)incumbent] : 0.f;
   if (best != incumbent && incumbent_ms > 0.f && best_ms > margin * incumbent_ms) {
      best = incumbent;
      best_ms = incumbent_ms;
   }
   if (best > 0 && default_ms > 0.f && best_ms > margin * default_ms) {   // (and the default is preferred to anything it is level with)
      best = 0;
      best_ms = default_ms;
   }
   {
      std::lock_guard<std::mutex> lock(p->mu);
      if (best == 0) p->plans.erase(std::make_tuple(n_streams, tile_streams, dev));
      else p->plans[std::make_tuple(n_streams, tile_streams, dev)] = cands[(size_t)best];
      p->plan_looked_up.insert(std::make_tuple(n_streams, tile_streams, dev));
   }
   plan_store(p, n_streams, n_samples, tile_streams, cands[(size_t)best], best_ms);
   if (chosen) *chosen = cands[(size_t)best];
   if (chosen_ms) *chosen_ms = best_ms;
   return FZ_OK;
}

}  // namespace fz
