// extern "C" entry points of libflowz_hip that are pure host work: compile(), IR inspection,
// kernel build (hiprtc needs no GPU), and the fz_run_block wrapper.
#include <cmath>
#include <cstring>
#include <memory>

#include "fz_internal.hpp"

using namespace fz;

#define FZ_GUARD(...)                                                           \
   try { __VA_ARGS__ }                                                                 \
   catch (const fz::Error& er) { fz::set_error(er.msg); return er.code; }       \
   catch (const std::exception& ex) { fz::set_error(ex.what()); return FZ_E_INVALID; }

// a program the SHIPPED reference would evaluate differently: the compile succeeds -- this library routes per the
// reference's own arity table -- and says so: fz_info.differs_from_reference, and this note in fz_last_error()
static void note_divergence(const Graph& g)
{
   if (g.ref_divergent)
      set_error("note: a feedback in this graph keeps " + std::to_string(g.ref_divergent) + " external input(s) for its promise part next to a future part that "
                "reads external inputs too: the reference's shipped binary_feedback hands the future part the wrong wires there (flowz.hpp:1045-1050, "
                "std::min(0, ...)); this library routes per the arity table (flowz.hpp:162-246), so its results differ from the reference's closure");
}

extern "C" {

int fz_compile(const fz_expr* e, fz_program** out)
{
   FZ_GUARD(
      if (!e || !out) fail(FZ_E_INVALID, "fz_compile: null argument");
      auto* p = new fz_program();
      try {
         p->g = lower(e);
         p->recipe = "typed 0\n" + serialize_expr(e);
         p->graph_hash = graph_structure_hash(p->g);
         p->g.sym_tag = (uint32_t)p->graph_hash;
      } catch (...) {
         delete p;
         throw;
      }
      *out = p;
      note_divergence(p->g);
      return FZ_OK;)
}

int fz_program_wave_part(const fz_program* p, uint32_t n_parts, uint32_t k, fz_program** out)
{
   FZ_GUARD(
      if (!p || !out) fail(FZ_E_INVALID, "fz_program_wave_part: null argument");
      const std::vector<Graph>* roles = p->g.wave_roles(n_parts);
      if (!roles) fail(FZ_E_UNSUPPORTED, "fz_program_wave_part: the graph does not split into that many parts");
      if (k >= n_parts) fail(FZ_E_INVALID, "fz_program_wave_part: part index out of range");
      auto* q = new fz_program();
      q->g = (*roles)[k];
      q->g.wave_splits.assign(5, {});
      q->graph_hash = graph_structure_hash(q->g);
      q->g.sym_tag = (uint32_t)q->graph_hash;
      *out = q;
      return FZ_OK;)
}

int fz_compile_typed(const fz_expr* e, const uint32_t* in_dtypes, uint32_t n_in_wires, fz_program** out)
{
   FZ_GUARD(
      if (!e || !out) fail(FZ_E_INVALID, "fz_compile_typed: null argument");
      if (in_dtypes && n_in_wires != (uint32_t)e->in_arity)
         fail(FZ_E_INVALID, "fz_compile_typed: in_dtypes must have one entry per input wire (input_arity)");
      LowerOptions opt;
      opt.typed = true;
      for (uint32_t i = 0; in_dtypes && i < n_in_wires; ++i) {
         if (in_dtypes[i] > FZ_DT_CF64) fail(FZ_E_INVALID, "fz_compile_typed: unknown fz_dtype");
         opt.in_dtype.push_back((uint8_t)in_dtypes[i]);
      }
      std::unique_ptr<fz_program> p(new fz_program());
      p->g = lower(e, opt);
      p->recipe = "typed 1";
      for (uint8_t d : opt.in_dtype) p->recipe += " " + std::to_string((unsigned)d);
      p->recipe += "\n" + serialize_expr(e);
      p->graph_hash = graph_structure_hash(p->g);
      p->g.sym_tag = (uint32_t)p->graph_hash;
      note_divergence(p->g);
      *out = p.release();
      return FZ_OK;)
}

int fz_program_input_dtypes(const fz_program* p, uint32_t* dtypes, uint32_t cap)
{
   if (!p) { set_error("null program"); return FZ_E_INVALID; }
   for (size_t k = 0; k < p->g.in_dtype.size() && k < cap && dtypes; ++k) dtypes[k] = p->g.in_dtype[k];
   return (int)p->g.in_dtype.size();
}

int fz_program_line_dtypes(const fz_program* p, uint32_t* dtypes, uint32_t cap)
{
   if (!p) { set_error("null program"); return FZ_E_INVALID; }
   for (size_t k = 0; k < p->g.lines.size() && k < cap && dtypes; ++k)
      dtypes[k] = p->g.lines[k].part ? (p->g.lines[k].f64 ? 3u : 1u) + p->g.lines[k].part : (p->g.lines[k].f64 ? 1u : 0u);
   return (int)p->g.lines.size();
}

void fz_program_destroy(fz_program* p) { delete p; }

int fz_program_info(const fz_program* p, fz_info* info)
{
   FZ_GUARD(
      if (!p || !info) fail(FZ_E_INVALID, "fz_program_info: null argument");
      const Graph& g = p->g;
      info->n_in = g.n_in;
      info->n_out = g.n_out;
      info->n_nodes = (uint32_t)g.nodes.size();
      info->n_ops = g.n_ops;
      info->n_lines = (uint32_t)g.lines.size();
      info->n_state = g.n_state;
      info->n_const = (uint32_t)g.consts.size();
      info->n_param = g.n_param;
      info->max_delay = g.max_delay;
      info->n_lds_slots = g.n_lds_slots;
      info->stage_packable = g.split.ok ? 1u : 0u;
      info->n_const64 = (uint32_t)g.consts64.size();
      info->n_out_wires = g.n_out_wires;
      info->n_in_wires = (uint32_t)g.in_dtype.size();
      info->typed = g.typed ? 1u : 0u;
      info->n_mod = g.n_mod;
      info->differs_from_reference = g.ref_divergent;
      return FZ_OK;)
}

int fz_program_ir(const fz_program* p, fz_ir_node* nodes, uint32_t cap)
{
   if (!p) { set_error("null program"); return FZ_E_INVALID; }
   const Graph& g = p->g;
   for (size_t k = 0; k < g.nodes.size() && k < cap; ++k) {
      nodes[k].kind = g.nodes[k].kind;
      nodes[k].a = g.nodes[k].a;
      nodes[k].b = g.nodes[k].b;
      nodes[k].c = g.nodes[k].c;
      const bool c = g.nodes[k].kind == FZ_IR_CONST;
      nodes[k].dtype = g.nodes[k].f64 ? 1u : 0u;
      nodes[k].value = (c && !g.nodes[k].f64) ? g.consts[g.nodes[k].a] : 0.f;
      nodes[k].value64 = (c && g.nodes[k].f64) ? g.consts64[g.nodes[k].a] : 0.0;
   }
   return (int)g.nodes.size();
}

int fz_program_outputs(const fz_program* p, uint32_t* ids, uint32_t cap)
{
   if (!p) { set_error("null program"); return FZ_E_INVALID; }
   for (size_t k = 0; k < p->g.outputs.size() && k < cap; ++k) ids[k] = p->g.outputs[k];
   return (int)p->g.outputs.size();
}

int fz_program_output_dtypes(const fz_program* p, uint32_t* dtypes, uint32_t cap)
{
   if (!p) { set_error("null program"); return FZ_E_INVALID; }
   for (size_t k = 0; k < p->g.outputs.size() && k < cap; ++k) {
      const uint8_t part = p->g.out_part[k];
      const bool f64 = p->g.nodes[p->g.outputs[k]].f64;
      dtypes[k] = (part == 1 || part == 2) && f64 ? 9u + part         // a complex<double> part narrowed to the float frame
                  : part ? 1u + part : (f64 ? 1u : 0u);
   }
   return (int)p->g.outputs.size();
}

int fz_program_lines(const fz_program* p, uint32_t* src, uint32_t* depth, uint32_t cap)
{
   if (!p) { set_error("null program"); return FZ_E_INVALID; }
   for (size_t k = 0; k < p->g.lines.size() && k < cap; ++k) {
      if (src) src[k] = p->g.lines[k].src;
      if (depth) depth[k] = p->g.lines[k].depth;
   }
   return (int)p->g.lines.size();
}

int fz_program_get_const(const fz_program* p, uint32_t slot, float* value)
{
   FZ_GUARD(
      if (!p || !value) fail(FZ_E_INVALID, "null argument");
      if (slot >= p->g.consts.size()) fail(FZ_E_INVALID, "coefficient slot out of range");
      *value = p->g.consts[slot];
      return FZ_OK;)
}

int fz_program_set_const(fz_program* p, uint32_t slot, float value)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null argument");
      if (slot >= p->g.consts.size()) fail(FZ_E_INVALID, "coefficient slot out of range");
      std::lock_guard<std::mutex> lock(p->mu);
      p->g.consts[slot] = value;
      return FZ_OK;)
}

int fz_program_set_uniform(fz_program* p, uint32_t k, float value)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null argument");
      auto it = p->g.uniform_slot.find(k);
      if (it == p->g.uniform_slot.end()) fail(FZ_E_INVALID, "graph has no uniform coefficient with this index");
      std::lock_guard<std::mutex> lock(p->mu);
      p->g.consts[it->second] = value;
      return FZ_OK;)
}

int fz_program_set_modulation(fz_program* p, const float* mod_dev, uint32_t stride)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null argument");
      if (!p->g.n_mod) fail(FZ_E_INVALID, "the graph has no sample-rate modulators");
      if (mod_dev && !stride) fail(FZ_E_INVALID, "fz_program_set_modulation: stride must be > 0");
      std::lock_guard<std::mutex> lock(p->mu);
      p->mod_dev = mod_dev;
      p->mod_stride = stride;
      return FZ_OK;)
}

int fz_program_build(fz_program* p, const fz_variant* v)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null program");
      // "auto" fields resolve as for a large stream count
      (void)get_kernel(p, resolve_variant(p->g, v, 1ull << 20), nullptr);
      return FZ_OK;)
}

int fz_program_build_for(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams)
{
   FZ_GUARD(
      if (!p || !n_streams || !n_samples) fail(FZ_E_INVALID, "fz_program_build_for: bad arguments");
      const Variant rv = finalize_variant(p, v, n_streams, n_samples, tile_streams);
      (void)get_kernel(p, rv, nullptr);
      // (a plain time-major block whose laps leave a few streams over launches a second kernel next to them)
      if (!(tile_streams && tile_streams < n_streams) && !(v && (v->flags & FZ_VF_STREAM_MAJOR))) {
         const uint64_t main_streams = lockstep_streams(p->g, v, rv, n_streams, tile_streams);
         if (main_streams < n_streams) (void)get_kernel(p, remainder_variant(p, v, n_streams, n_samples, n_streams - main_streams), nullptr);
      }
      return FZ_OK;)
}

int fz_program_kernel_resources(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                                int as_launched, fz_kernel_resources* out)
{
   FZ_GUARD(
      if (!p || !out || !n_streams || !n_samples) fail(FZ_E_INVALID, "fz_program_kernel_resources: bad arguments");
      const Variant rv = finalize_variant(p, v, n_streams, n_samples, tile_streams, as_launched != 0);
      const auto k = get_kernel(p, rv, nullptr);
      *out = fz_kernel_resources{k->res.vgprs, k->res.agprs, k->res.sgprs, k->res.scratch_bytes, k->res.lds_bytes, k->res.vgpr_spills,
                                 k->res.sgpr_spills, rv.U};
      return FZ_OK;)
}

long fz_program_kernel_name(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                            char* buf, size_t cap)
{
   try {
      if (!p) fail(FZ_E_INVALID, "null program");
      const std::string s = kernel_name(p->g, finalize_variant(p, v, n_streams ? n_streams : (1ull << 20), n_samples ? n_samples : (1u << 20), tile_streams));
      if (buf && cap) {
         const size_t n = std::min(cap - 1, s.size());
         std::memcpy(buf, s.data(), n);
         buf[n] = 0;
      }
      return (long)s.size();
   } catch (const fz::Error& er) {
      set_error(er.msg);
      return er.code;
   }
}

long fz_program_kernel_symbol(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                            char* buf, size_t cap)
{
   try {
      if (!p) fail(FZ_E_INVALID, "null program");
      const std::string s = kernel_symbol(p->g, finalize_variant(p, v, n_streams ? n_streams : (1ull << 20), n_samples ? n_samples : (1u << 20), tile_streams));
      if (buf && cap) {
         const size_t n = std::min(cap - 1, s.size());
         std::memcpy(buf, s.data(), n);
         buf[n] = 0;
      }
      return (long)s.size();
   } catch (const fz::Error& er) {
      set_error(er.msg);
      return er.code;
   }
}

long fz_program_kernel_code_id(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                               char* buf, size_t cap)
{
   try {
      if (!p) fail(FZ_E_INVALID, "null program");
      const std::string s = kernel_code_id(p, finalize_variant(p, v, n_streams ? n_streams : (1ull << 20), n_samples ? n_samples : (1u << 20), tile_streams));
      if (buf && cap) {
         const size_t n = std::min(cap - 1, s.size());
         std::memcpy(buf, s.data(), n);
         buf[n] = 0;
      }
      return (long)s.size();
   } catch (const fz::Error& er) {
      set_error(er.msg);
      return er.code;
   }
}

long fz_expr_recipe(const fz_expr* e, char* buf, size_t cap)
{
   try {
      if (!e) fail(FZ_E_INVALID, "null expression");
      const std::string s = serialize_expr(e);
      if (buf && cap) {
         const size_t n = std::min(cap - 1, s.size());
         std::memcpy(buf, s.data(), n);
         buf[n] = 0;
      }
      return (long)s.size();
   } catch (const fz::Error& er) {
      set_error(er.msg);
      return er.code;
   }
}

fz_expr* fz_expr_from_recipe(const char* text) { return text ? parse_expr(text) : nullptr; }

int fz_manifest_build(const char* path, uint32_t n_workers, uint32_t* counts)
{
   FZ_GUARD(
      if (!path || !counts) fail(FZ_E_INVALID, "fz_manifest_build: null argument");
      return manifest_build(path, n_workers, counts);)
}

long fz_program_source(fz_program* p, const fz_variant* v, char* buf, size_t cap)
{
   try {
      if (!p) fail(FZ_E_INVALID, "null program");
      const std::string s = full_source(p->g, resolve_variant(p->g, v, 1ull << 20));
      if (buf && cap) {
         const size_t n = std::min(cap - 1, s.size());
         std::memcpy(buf, s.data(), n);
         buf[n] = 0;
      }
      return (long)s.size();
   } catch (const fz::Error& er) {
      set_error(er.msg);
      return er.code;
   }
}

uint32_t fz_recommended_tile_streams(const fz_program* p)
{
   if (!p) return 0;
   // row segments of ~32 KiB stream best from HBM3E on MI355X (profiles/r01): with unequal frame
   // widths aim the geometric mean of the input and output segment at 32 KiB
   const double wi = p->g.n_in ? p->g.n_in : 1, wo = p->g.n_out ? p->g.n_out : 1;
   double want = 8192.0 / std::sqrt(wi * wo);
   uint32_t t = 1024;
   while (t * 1.5 < want && t < 65536) t *= 2;
   return t;
}

int fz_run_block(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                 uint32_t n_samples, const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null program");
      return launch(p, in, out, state, params, n_streams, n_samples, v, hip_stream, 0);)
}

int fz_run_block_tiled(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                       uint32_t n_samples, uint32_t tile_streams, const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null program");
      return launch(p, in, out, state, params, n_streams, n_samples, v, hip_stream, tile_streams);)
}

int fz_run_block_window(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                        uint32_t rows_total, uint32_t row0, uint32_t n_samples, uint32_t tile_streams, const fz_variant* v,
                        void* hip_stream)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null program");
      if (!rows_total) fail(FZ_E_INVALID, "rows_total must be > 0");
      return launch(p, in, out, state, params, n_streams, n_samples, v, hip_stream, tile_streams, rows_total, row0);)
}

int fz_run_block_stream_major(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                              uint32_t rows_total, uint32_t row0, uint32_t n_samples, const fz_variant* v, void* hip_stream)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null program");
      if (!rows_total) fail(FZ_E_INVALID, "rows_total must be > 0");
      fz_variant sm = v ? *v : fz_variant{0, 0, 0, 0};
      sm.flags |= FZ_VF_STREAM_MAJOR;
      return launch(p, in, out, state, params, n_streams, n_samples, &sm, hip_stream, 0, rows_total, row0);)
}

int fz_program_tune(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                    uint32_t n_samples, uint32_t tile_streams, void* hip_stream, fz_variant* chosen, float* chosen_ms)
{
   FZ_GUARD(
      if (!p) fail(FZ_E_INVALID, "null program");
      return tune(p, in, out, state, params, n_streams, n_samples, tile_streams, hip_stream, chosen, chosen_ms);)
}

int fz_program_plan(fz_program* p, uint64_t n_streams, uint32_t tile_streams, fz_variant* out)
{
   FZ_GUARD(
      if (!p || !out || !n_streams) fail(FZ_E_INVALID, "fz_program_plan: bad arguments");
      if (device_count() <= 0) fail(FZ_E_NO_DEVICE, "fz_program_plan: plans are per board; no HIP device visible");
      *out = planned_variant(p, n_streams, tile_streams);
      return FZ_OK;)
}

int fz_program_tune_candidates(fz_program* p, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams, fz_variant* out, uint32_t cap)
{
   FZ_GUARD(
      if (!p || !n_streams || !n_samples) fail(FZ_E_INVALID, "fz_program_tune_candidates: bad arguments");
      const std::vector<fz_variant> c = tune_candidates(p->g, n_streams, n_samples, tile_streams);
      for (size_t i = 0; i < c.size() && i < cap && out; ++i) out[i] = c[i];
      return (int)c.size();)
}

}  // extern "C"
