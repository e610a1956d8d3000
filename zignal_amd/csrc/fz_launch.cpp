// The launch of the fused block kernel: argument checks, plan lookup / first-launch measurement, kernarg image.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "fz_runtime.hpp"

fz_program::~fz_program()
{
   for (auto& kv : sync_dev)
      if (kv.second.first) (void)hipFree(kv.second.first);   // (hipFree waits for whatever still runs on the device)
   for (void* q : sync_retired) (void)hipFree(q);
   for (auto& kv : side) {
      if (kv.second.stream) (void)hipStreamDestroy((hipStream_t)kv.second.stream);
      if (kv.second.fork) (void)hipEventDestroy((hipEvent_t)kv.second.fork);
      if (kv.second.join) (void)hipEventDestroy((hipEvent_t)kv.second.join);
   }
}

namespace fz {

// ---- launch ------------------------------------------------------------------------------------------------
// compute units of the current device (hipDeviceAttributeMultiprocessorCount, asked once per device; 256 -- the MI355X -- on a
// box without a GPU, where only names and resources are asked for)
unsigned chip_cus()
{
   static std::mutex mu;
   static std::map<int, unsigned> known;
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess) {
      (void)hipGetLastError();
      return kChipCUs;
   }
   std::lock_guard<std::mutex> lock(mu);
   unsigned& c = known[dev];
   if (!c) {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) {
         (void)hipGetLastError();
         cus = (int)kChipCUs;
      }
      c = (unsigned)cus;
   }
   return c;
}

// workgroups of this kernel the chip holds at a time: occupancy x CUs, in whole eights (the XCDs); asked once per kernel and device
static unsigned resident_workgroups(void* fn, int threads)
{
   static std::mutex mu;
   static std::map<std::pair<void*, int>, unsigned> known;
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   std::lock_guard<std::mutex> lock(mu);
   unsigned& r = known[{fn, dev}];
   if (!r) {
      int per_cu = 0;
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (hipFunction_t)fn, threads, 0) != hipSuccess || per_cu < 1) {
         (void)hipGetLastError();
         per_cu = 1;
      }
      r = std::max(8u, chip_cus() * (unsigned)per_cu / 8u * 8u);
   }
   return r;
}

// FZ_VF_GRID_SYNC: `bytes` of arrival counters for one launch -- a slice of a small buffer the program owns per device, 16 slices
// handed out in turn (launches on different streams may overlap and must not share counters; a slice comes round again after 15
// other launches).  A buffer that turns out too small is REPLACED, never freed before the program is: a hipGraph captured earlier
// may still hold its slices.
static unsigned int* sync_counters(fz_program* p, size_t bytes, void* stream)
{
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   std::lock_guard<std::mutex> lock(p->mu);
   auto& slot = p->sync_dev[dev];
   if (slot.second < bytes) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) (void)hipGetLastError();
      if (cap != hipStreamCaptureStatusNone)
         fail(FZ_E_INVALID, "FZ_VF_GRID_SYNC: the arrival counters cannot be allocated while the stream is being captured: launch this "
                            "shape once before the capture");
      if (slot.first) p->sync_retired.push_back(slot.first);        // (freed with the program)
      slot = {nullptr, 0};
      const size_t per = std::max<size_t>((bytes + 4095) / 4096 * 4096, 16384);
      FZ_HIP(hipMalloc(&slot.first, per * 16));
      slot.second = per;
   }
   return reinterpret_cast<unsigned int*>(static_cast<char*>(slot.first) + (size_t)(p->sync_next++ % 16u) * slot.second);
}

struct ArgsHeader {
   const float* in;
   float* out;
   float* state;
   const float* params;
   const float* mod;
   unsigned int* sync;
   unsigned long long n_streams;
   unsigned int n_samples;
   unsigned int n_groups;
   unsigned int tile_streams;
   unsigned int tile_blocks;
   unsigned int rows_total;
   unsigned int row0;
   unsigned int mod_stride;
   unsigned int n_blocks;
   unsigned int group0;
   unsigned int reserved0;
};
// the program's side stream on the current device and the two events that fork it from / join it to a caller's stream (created on
// first use, destroyed with the program).  An event wait takes the event's LATEST record: with two host threads launching remainder
// shapes at once, thread A's hipStreamWaitEvent(side, fork) could pick up thread B's record and A's remainder would no longer be
// ordered behind A's own earlier work.  The whole fork ... join sequence of a launch therefore runs under the side stream's mutex
// (enqueue calls only: microseconds; the kernels are resolved before it is taken).
static SideStream& side_stream(fz_program* p)
{
   int dev = 0;
   FZ_HIP(hipGetDevice(&dev));
   std::lock_guard<std::mutex> lock(p->mu);
   SideStream& s = p->side[dev];                            // (std::map: the reference stays valid)
   if (!s.stream) {
      hipStream_t st;
      hipEvent_t a, b;
      FZ_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      FZ_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
      FZ_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      s.stream = st;
      s.fork = a;
      s.join = b;
      s.mu.reset(new std::mutex());
   }
   return s;
}

static_assert(sizeof(ArgsHeader) % 8 == 0 && sizeof(ArgsHeader) == 6 * 8 + 8 + 10 * 4, "ArgsHeader must match the head of the kernel's fz_args without padding");

int launch(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
           uint32_t n_samples, const fz_variant* uv, void* stream, uint32_t tile_streams, uint32_t rows_total, uint32_t row0,
           uint32_t mod_row0)
{
   if (rows_total == 0) rows_total = n_samples;             // the block is the whole buffer
   if ((uint64_t)row0 + n_samples > rows_total) fail(FZ_E_INVALID, "row0 + n_samples exceeds rows_total");
   const bool stream_major = uv && (uv->flags & FZ_VF_STREAM_MAJOR);
   if (stream_major) {
      if (tile_streams) fail(FZ_E_INVALID, "stream-major frames are not tiled");
      if ((uint64_t)rows_total * std::max(p->g.n_in, p->g.n_out) >= ((uv->flags & FZ_VF_SM_LONG) && uv->streams_per_lane == 2 ? (1ull << 23) : (1ull << 24)))
         fail(FZ_E_UNSUPPORTED, "stream-major frames: more than 2^24 floats per stream buffer (2^23 with two streams per lane; the rows of a wave are addressed through one 4 GiB descriptor): use a window");
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
   // (state and coefficient rows go through one-row buffer descriptors: 32-bit byte offsets and sizes, n_streams * 4 < 2^32)
   if (n_streams >= (1ull << 30)) fail(FZ_E_UNSUPPORTED, "2^30 streams or more per launch: shard the streams");
   if (tile_streams && n_streams % tile_streams) fail(FZ_E_INVALID, "n_streams must be a multiple of tile_streams");
   require_device();
   fz_variant planned;
   bool from_plan = false;
   if (!uv) {                                               // a measured plan for this shape on this device?
      int dev = 0;
      FZ_HIP(hipGetDevice(&dev));
      const auto key = std::make_tuple(n_streams, tile_streams, dev);
      bool known = false;
      (void)planned_variant(p, n_streams, tile_streams);      // first launch of this shape: a plan persisted by an earlier process?
      // A launch without a variant runs the plan fz_program_tune measured for the shape (this process or an earlier one: plans.txt), else
      // the library's static choice.  Round 6: the measurement is never made behind the caller's back any more -- FLOWZ_HIP_AUTOTUNE=1
      // opts in to what round 3-5 did by default: the first BIG block of a shape (>= 2^26 stream-samples) measures the candidates whose
      // code objects are at hand on the caller's buffers (the state is saved and restored around it, `out` is recomputed below; about
      // ten launches each, nothing is JIT-compiled for it).
      const char* const at_env = std::getenv("FLOWZ_HIP_AUTOTUNE");      // (read at every launch: a process may turn it on for some of its work)
      const bool autotune = at_env && *at_env == '1';
      bool may_tune = autotune && rows_total == n_samples && row0 == 0 && n_streams * (uint64_t)n_samples >= (1ull << 26);
      if (may_tune) {
         // not while the stream is being captured into a hipGraph (the measurement allocates and synchronises), and not
         // in place: the candidates run on the caller's buffers, an aliased `in` would be overwritten before the real launch
         hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
         if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) (void)hipGetLastError();
         const char* ib = reinterpret_cast<const char*>(in);
         const char* ob = reinterpret_cast<const char*>(out);
         const size_t ibytes = (size_t)n_streams * n_samples * g.n_in * 4, obytes = (size_t)n_streams * n_samples * out_w * 4;
         const bool overlap = in && ib < ob + obytes && ob < ib + ibytes;
         may_tune = cap == hipStreamCaptureStatusNone && !overlap;
      }
      bool can_tune = false;
      {
         // (another thread's first big launch of this shape may be measuring the plan right now, on ITS buffers: wait for the
         //  result instead of racing it -- the measuring thread's own launches carry explicit variants and never come here)
         // ONE critical section decides who measures: the thread whose insert into `measuring` succeeds; everybody else of the
         // shape waits above until that thread is done
         std::unique_lock<std::mutex> lock(p->mu);
         p->measured.wait(lock, [&] { return p->measuring.count(key) == 0; });
         auto it = p->plans.find(key);
         known = it != p->plans.end() || p->tuned_default.count(key) != 0;
         // (a plan is measured on blocks of thousands of samples: the wave-split kernels pay several masked rounds per launch and
         //  are not what a short block -- the per-sample call protocol -- should run, whatever was tuned for the shape)
         if (it != p->plans.end() && !(ws_parts(it->second.flags) && n_samples < 256)) {
            planned = it->second;
            uv = &planned;
            from_plan = true;
         }
         if (may_tune && !known) {
            p->tuned_default.insert(key);                   // (also stops the recursion through tune -> launch)
            can_tune = p->measuring.insert(key).second;     // other launches of this shape wait until the plan is known
         }
      }
      if (can_tune) {
         struct Done {                                      // ... on every exit path
            fz_program* p;
            decltype(key) k;
            ~Done()
            {
               {
                  std::lock_guard<std::mutex> lock(p->mu);
                  p->measuring.erase(k);
               }
               p->measured.notify_all();
            }
         } done{p, key};
         const size_t sb = (size_t)g.n_state * n_streams * 4;
         // The candidates run on the caller's buffers: the state is saved before and restored after the measurement, and the
         // restore is CHECKED -- a state that could not be put back is an error of this launch, never a silent one.  A failure
         // inside the measurement itself (a candidate's HIP error) is not the caller's problem: the default launch below goes ahead.
         float* copy = nullptr;
         bool have_copy = true;
         if (sb) {
            if (hipMalloc((void**)&copy, sb) != hipSuccess) {           // no room for the snapshot (multi-GiB state): do not tune
               (void)hipGetLastError();
               copy = nullptr;
               have_copy = false;
            } else if (hipMemcpyAsync(copy, state, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
               (void)hipGetLastError();
               (void)hipFree(copy);
               copy = nullptr;
               have_copy = false;
            }
         }
         if (have_copy) {
            fz_variant chosen{0, 0, 0, 0};
            int rc = FZ_E_INVALID;
            std::string why;
            try {
               rc = tune(p, in, out, state, params, n_streams, n_samples, tile_streams, stream, &chosen, nullptr, true);
            } catch (const Error& er) {
               rc = er.code;
               why = er.msg;
               if (er.code == FZ_E_HIP) (void)hipGetLastError();
            }
            if (copy) {
               hipError_t e1 = hipMemcpyAsync(state, copy, sb, hipMemcpyDeviceToDevice, (hipStream_t)stream);
               hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
               (void)hipFree(copy);
               if (e1 != hipSuccess || e2 != hipSuccess)
                  fail(FZ_E_HIP, std::string("the closure state could not be restored after the plan measurement of this shape (") +
                                    hipGetErrorString(e1 != hipSuccess ? e1 : e2) + "): `state` is advanced by the measurement's blocks -- reset it (the measurement was asked for with FLOWZ_HIP_AUTOTUNE=1)");
            }
            if (rc == FZ_OK && (chosen.streams_per_lane || chosen.unroll || chosen.block_threads || chosen.flags)) {
               planned = chosen;
               uv = &planned;
               from_plan = true;
            } else if (rc != FZ_OK && std::getenv("FLOWZ_HIP_DEBUG")) {
               std::fprintf(stderr, "[flowz_hip] plan measurement of this shape failed (%s): the library default runs\n", why.c_str());
            }
         }
      }
   }
   Variant v;
   try {
      v = finalize_variant(p, uv, n_streams, n_samples, tile_streams);
   } catch (const Error&) {
      // a remembered plan that does not resolve any more (persisted by another build of the library, a damaged line that passed the
      // range checks): forget it and run the library's own choice instead of failing every default launch of this shape
      if (!from_plan) throw;
      drop_plan(p, n_streams, tile_streams);
      uv = nullptr;
      v = finalize_variant(p, nullptr, n_streams, n_samples, tile_streams);
   }
   if (stream_major && (v.flags & FZ_VF_SM_LONG) && v.P == 2 && (uint64_t)rows_total >= (1ull << 23)) {
      // the library's own choice of the pair body (128 rows per descriptor) on buffers too long for it: one stream per lane
      fz_variant one = *uv;
      one.streams_per_lane = 1;
      v = finalize_variant(p, &one, n_streams, n_samples, tile_streams);
   }
   // sample-rate modulators: one array for all streams
   const float* mod_dev = nullptr;
   uint32_t mod_stride = 0;
   if (g.n_mod) {
      std::lock_guard<std::mutex> lock(p->mu);
      mod_dev = p->mod_dev;
      mod_stride = p->mod_stride;
      if (!mod_dev) fail(FZ_E_INVALID, "the graph has sample-rate modulators: call fz_program_set_modulation first");
      // (the host-frames pipelines hand time chunks of a long block to the kernel as buffers of their own: row 0 of such a
      //  buffer is sample mod_row0 of the block, and the modulator rows must follow)
      if ((uint64_t)mod_row0 + row0 + n_samples > mod_stride) fail(FZ_E_INVALID, "fz_program_set_modulation: stride is shorter than the rows of this launch");
      mod_dev += mod_row0;
   }
   // kernarg image of `struct fz_args` (8-byte aligned: pad the coefficient tail)
   // (built on the stack for ordinary graphs: no allocation on the launch path)
   const size_t off64 = (sizeof(ArgsHeader) + sizeof(float) * std::max<size_t>(g.consts.size(), 1) + 7) & ~size_t(7);
   const size_t kbytes = off64 + sizeof(double) * std::max<size_t>(g.consts64.size(), 1);
   alignas(8) char small[1024];
   std::vector<char> big;
   char* const kbuf = kbytes <= sizeof small ? small : (big.resize(kbytes), big.data());
   {
      std::lock_guard<std::mutex> lock(p->mu);
      if (!g.consts.empty()) std::memcpy(kbuf + sizeof(ArgsHeader), g.consts.data(), sizeof(float) * g.consts.size());
      if (!g.consts64.empty()) std::memcpy(kbuf + off64, g.consts64.data(), sizeof(double) * g.consts64.size());
   }
   static const bool debug = std::getenv("FLOWZ_HIP_DEBUG") != nullptr;

   // One kernel variant over the streams [first, first + count) of the block (the pointers and the row pitch stay the block's:
   // the kernel adds the first stream group itself).
   auto run_part = [&](const Variant& w, uint64_t first, uint64_t count, void* stream) {
      void* fn = nullptr;
      auto k = get_kernel(p, w, &fn);
      const unsigned group0 = (unsigned)(first / w.P);
      const unsigned groups = (unsigned)((count + ((w.flags & FZ_VF_RAGGED) ? w.P - 1 : 0)) / w.P);   // lanes of work (FZ_VF_RAGGED: the last one is partial)
      const unsigned n_blocks = (groups + w.block - 1) / w.block;
      // (wave split: w.block counts the 64 streams of a workgroup; two waves evaluate them)
      const unsigned threads = ws_parts(w.flags) ? w.block * ws_waves(w.flags) : w.block;
      // FZ_VF_GRID_SYNC needs every workgroup that synchronises RUNNING: with more blocks than the chip holds workgroups of this
      // kernel (occupancy x CUs) the streams are cut into LAPS -- contiguous ranges of at most one workgroup per resident slot,
      // the same number of blocks in every lap (whole eights: the XCDs) -- and every lap is a launch of its own (the one-lap kernel has no
      // loop to pay registers for: four streams per lane fit where round 3's persistent kernel stepped down; that kernel left in round 6).
      unsigned laps = 1, per_lap = n_blocks, grid = n_blocks;
      size_t sync_bytes = 0;
      if (w.flags & FZ_VF_GRID_SYNC) {
         const unsigned resident = resident_workgroups(fn, (int)threads);
         if (n_blocks > resident) {
            laps = (n_blocks + resident - 1) / resident;
            per_lap = std::min(resident, ((n_blocks + laps - 1) / laps + 7u) / 8u * 8u);
            grid = per_lap;
         }
         sync_bytes = (size_t)8 * 128;                       // one arrival counter per XCD, zeroed in stream order before every lap's launch
      }
      ArgsHeader h{in, out, state, params, mod_dev, nullptr, (unsigned long long)n_streams, n_samples, group0 + groups,
                   (unsigned int)row_streams, tile_streams ? (unsigned int)(tile_streams / (w.P * w.block)) : 0u, rows_total, row0, mod_stride, n_blocks, group0, 0u};
      for (unsigned lap = 0; lap < laps; ++lap) {
         if (laps > 1) {
            h.group0 = group0 + lap * per_lap * w.block;
            grid = std::min(per_lap, n_blocks - lap * per_lap);
         }
         if (sync_bytes) {
            h.sync = sync_counters(p, sync_bytes, stream);
            FZ_HIP(hipMemsetAsync(h.sync, 0, sync_bytes, (hipStream_t)stream));
         }
         std::memcpy(kbuf, &h, sizeof h);
         size_t size = kbytes;
         void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, kbuf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
         FZ_HIP(hipModuleLaunchKernel((hipFunction_t)fn, grid, 1, 1, threads, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
      }
      if (debug) {
         FZ_HIP(hipStreamSynchronize((hipStream_t)stream));
         std::fprintf(stderr, "[flowz_hip] launched streams [%llu, %llu) of %llu: %u lap(s) grid=%u block=%u P=%u U=%u flags=%u n_samples=%u kernarg=%zu B vgprs=%u scratch=%u B/lane\n",
                      (unsigned long long)first, (unsigned long long)(first + count), (unsigned long long)n_streams, laps, grid, w.block, w.P, w.U, w.flags, n_samples, kbytes,
                      k->res.vgprs + k->res.agprs, k->res.scratch_bytes);
      }
   };
   // the library's own lockstep choice may cover whole laps only and leave the last few streams to a launch of their own
   // (time_major_geometry): those run what a block of that few streams runs by itself
   const uint64_t main_streams = stream_major ? n_streams : lockstep_streams(g, uv, v, n_streams, tile_streams);
   if (main_streams == n_streams) {
      run_part(v, 0, n_streams, stream);
      return FZ_OK;
   }
   // The remainder runs NEXT TO the laps, not behind them: on a side stream of the program, forked from and joined to the caller's
   // stream by events (capturable like any fork / join).  Its few waves walk all the rows of the block on their own -- every row
   // another page, ~1.4 ms per 4096 rows for ONE stream behind a million (measured: 7.07 ms for 1 048 577 streams against 5.62 ms for
   // 1 048 576 when it ran behind the lap) -- and the lap's workgroups leave room for them (92 of a SIMD's 128 registers per lane).
   const uint64_t rem = n_streams - main_streams;
   const Variant r = remainder_variant(p, uv, n_streams, n_samples, rem);
   (void)get_kernel(p, r, nullptr);                          // (resolved -- built, if need be -- before the side stream's mutex is taken)
   (void)get_kernel(p, v, nullptr);
   SideStream& side = side_stream(p);
   std::lock_guard<std::mutex> fork_join(*side.mu);
   FZ_HIP(hipEventRecord((hipEvent_t)side.fork, (hipStream_t)stream));
   FZ_HIP(hipStreamWaitEvent((hipStream_t)side.stream, (hipEvent_t)side.fork, 0));
   run_part(r, main_streams, rem, side.stream);
   FZ_HIP(hipEventRecord((hipEvent_t)side.join, (hipStream_t)side.stream));
   run_part(v, 0, main_streams, stream);
   FZ_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)side.join, 0));
   return FZ_OK;
}

}  // namespace fz
