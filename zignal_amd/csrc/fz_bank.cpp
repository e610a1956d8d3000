// fz_bank: the closure's state_ member (flowz.hpp:1190-1191) for n_streams streams, resident in HBM, and the
// host-frames paths (the reference's per-sample call protocol; long blocks through a three-stream H2D / kernel / D2H pipeline).
#include <algorithm>
#include <cstring>
#include <memory>

#include "fz_runtime.hpp"

using namespace fz;

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
                            b->s_run, 0, 0, 0, t0);              // (sample-rate modulators: this chunk starts at sample t0)
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
                            b->s_run, 0, chunk_t, 0, t0);
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
