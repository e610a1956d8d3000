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
}

namespace fz {

// ---- launch ------------------------------------------------------------------------------------------------
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
};
static_assert(sizeof(ArgsHeader) % 8 == 0 && sizeof(ArgsHeader) == 6 * 8 + 8 + 8 * 4, "ArgsHeader must match the head of the kernel's fz_args without padding");

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
   if (n_streams >= (1ull << 32)) fail(FZ_E_UNSUPPORTED, "more than 2^32 streams per launch: shard the streams");
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
      {
         std::lock_guard<std::mutex> lock(p->mu);
         auto it = p->plans.find(key);
         known = it != p->plans.end() || p->tuned_default.count(key) != 0;
         // (a plan is measured on blocks of thousands of samples: the wave-split kernels pay several masked rounds per launch and
         //  are not what a short block -- the per-sample call protocol -- should run, whatever was tuned for the shape)
         if (it != p->plans.end() && !(ws_parts(it->second.flags) && n_samples < 256)) {
            planned = it->second;
            uv = &planned;
            from_plan = true;
         }
      }
      // The first BIG block of a shape measures the plan by itself (round 3: on by default; FLOWZ_HIP_AUTOTUNE=0 turns it off):
      // which variant streams fastest differs from board to board by more than the variants differ on one board (the same
      // kernel: +5 % here, -13 % there), so the library's static choice is only the first candidate.  The measurement runs on the
      // caller's buffers (the state is saved and restored around it, `out` is recomputed below), takes the candidates whose
      // code objects are at hand (build() pre-builds them for the BASELINE graphs; nothing is JIT-compiled for it) and
      // costs about ten launches each; blocks below 2^26 stream-samples (a few hundred microseconds) never trigger it.
      static const bool autotune = [] { const char* e = std::getenv("FLOWZ_HIP_AUTOTUNE"); return !(e && *e == '0'); }();
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
            const int rc = tune(p, in, out, state, params, n_streams, n_samples, tile_streams, stream, &chosen, nullptr, true);
            if (rc != FZ_OK) return rc;
            if (chosen.streams_per_lane || chosen.unroll || chosen.block_threads || chosen.flags) {
               planned = chosen;
               uv = &planned;
               from_plan = true;
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
      // (the host-frames pipelines hand time chunks of a long block to the kernel as buffers of their own: row 0 of such a
      //  buffer is sample mod_row0 of the block, and the modulator rows must follow)
      if ((uint64_t)mod_row0 + row0 + n_samples > mod_stride) fail(FZ_E_INVALID, "fz_program_set_modulation: stride is shorter than the rows of this launch");
      mod_dev += mod_row0;
   }
   const unsigned n_blocks = (unsigned)(((unsigned int)(n_streams / v.P) + v.block - 1) / v.block);
   unsigned grid = n_blocks;
   // (wave split: v.block counts the 64 streams of a workgroup; two waves evaluate them)
   const unsigned threads = ws_parts(v.flags) ? v.block * ws_waves(v.flags) : v.block;
   unsigned int* sync_dev = nullptr;
   if (v.flags & FZ_VF_GRID_SYNC) {
      // persistent launch: the grid is what the chip holds of THIS kernel at a time (occupancy x CUs, whole laps over the 8 XCDs);
      // per-(lap, XCD) arrival counters, zeroed in stream order before the launch
      int dev = 0;
      FZ_HIP(hipGetDevice(&dev));
      unsigned resident = 0;
      {  // (asked once per kernel and device: the launch path stays free of driver queries)
         static std::mutex mu;
         static std::map<std::pair<void*, int>, unsigned> known;
         std::lock_guard<std::mutex> lock(mu);
         unsigned& r = known[{fn, dev}];
         if (!r) {
            int per_cu = 0, cus = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (hipFunction_t)fn, (int)threads, 0) != hipSuccess || per_cu < 1) {
               (void)hipGetLastError();
               per_cu = 1;
            }
            FZ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            r = std::max(8u, (unsigned)(cus * per_cu) / 8u * 8u);
         }
         resident = r;
      }
      if ((v.flags & FZ_VF_PERSIST) && n_blocks > resident) grid = resident;   // (one lap: the waits of workgroups that are not running yet are bounded)
      const size_t bytes = (size_t)((n_blocks + grid - 1) / grid) * 8 * 128;
      {
         std::lock_guard<std::mutex> lock(p->mu);
         auto& slot = p->sync_dev[dev];
         if (slot.second < bytes) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) (void)hipGetLastError();
            if (cap != hipStreamCaptureStatusNone)
               fail(FZ_E_INVALID, "FZ_VF_GRID_SYNC: the arrival counters cannot be allocated while the stream is being captured: launch this "
                                  "shape once before the capture");
            if (slot.first) FZ_HIP(hipFree(slot.first));   // (synchronises: nothing in flight uses the old counters)
            slot = {nullptr, 0};
            const size_t per = std::max<size_t>((bytes + 4095) / 4096 * 4096, 16384);
            FZ_HIP(hipMalloc(&slot.first, per * 16));
            slot.second = per;
         }
         sync_dev = reinterpret_cast<unsigned int*>(static_cast<char*>(slot.first) + (size_t)(p->sync_next++ % 16u) * slot.second);
      }
      FZ_HIP(hipMemsetAsync(sync_dev, 0, bytes, (hipStream_t)stream));
   }
   ArgsHeader h{in, out, state, params, mod_dev, sync_dev, (unsigned long long)n_streams, n_samples, (unsigned int)(n_streams / v.P),
                (unsigned int)row_streams, tile_streams ? (unsigned int)(tile_streams / (v.P * v.block)) : 0u, rows_total, row0, mod_stride, n_blocks};
   std::memcpy(kbuf, &h, sizeof h);
   {
      std::lock_guard<std::mutex> lock(p->mu);
      if (!g.consts.empty()) std::memcpy(kbuf + sizeof h, g.consts.data(), sizeof(float) * g.consts.size());
      if (!g.consts64.empty()) std::memcpy(kbuf + off64, g.consts64.data(), sizeof(double) * g.consts64.size());
   }
   size_t size = kbytes;
   void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, kbuf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
   FZ_HIP(hipModuleLaunchKernel((hipFunction_t)fn, grid, 1, 1, threads, 1, 1, 0, (hipStream_t)stream, nullptr, extra));
   static const bool debug = std::getenv("FLOWZ_HIP_DEBUG") != nullptr;
   if (debug) {
      FZ_HIP(hipStreamSynchronize((hipStream_t)stream));
      std::fprintf(stderr, "[flowz_hip] launched grid=%u block=%u P=%u U=%u flags=%u n_streams=%llu n_samples=%u kernarg=%zu B vgprs=%u scratch=%u B/lane\n",
                   grid, v.block, v.P, v.U, v.flags, (unsigned long long)n_streams, n_samples, size, k->res.vgprs + k->res.agprs, k->res.scratch_bytes);
   }
   return FZ_OK;
}

}  // namespace fz
