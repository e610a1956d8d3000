// Expression trees and the static-analysis transforms of the Flowz EDSL.
//
// Re-designed, not translated: the reference computes these at C++ compile time with
// Boost.Proto transforms over expression *types* (flowz.hpp:162-246 input/output_arity,
// :443-506 max_input_delays); here they are plain functions over a reference-counted run-time
// tree, the arities evaluated once when a node is constructed.
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "fz_internal.hpp"

namespace fz {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
[[noreturn]] void fail(int code, const std::string& msg) { throw Error{code, msg}; }

static fz_expr* mk(EK k, fz_expr* a = nullptr, fz_expr* b = nullptr)
{
   auto* e = new fz_expr();
   e->kind = k;
   e->a = a;
   e->b = b;
   if (a) fz_expr_retain(a);
   if (b) fz_expr_retain(b);
   return e;
}

// per-wire deepest delayed read, flowz.hpp:443-502.  The tuple length follows the reference's
// generator (make_arity :286-301): a leaf _i contributes i entries.
static std::vector<uint32_t> zipmax(const std::vector<uint32_t>& l, const std::vector<uint32_t>& r)
{
   std::vector<uint32_t> out(std::max(l.size(), r.size()), 0u);   // max_delay_of_wires :364-379
   for (size_t k = 0; k < out.size(); ++k) {
      uint32_t x = k < l.size() ? l[k] : 0u, y = k < r.size() ? r[k] : 0u;
      out[k] = std::max(x, y);
   }
   return out;
}

static std::vector<uint32_t> drop(std::vector<uint32_t> v, size_t n)
{
   if (n >= v.size()) return {};           // tuple_drop of a shorter tuple is () (tuple_tools.hpp:147-151)
   v.erase(v.begin(), v.begin() + (long)n);
   return v;
}

static std::vector<uint32_t> cat(std::vector<uint32_t> a, const std::vector<uint32_t>& b)
{
   a.insert(a.end(), b.begin(), b.end());
   return a;
}

std::vector<uint32_t> max_input_delays(const fz_expr* e)
{
   switch (e->kind) {
      case EK::Delayed: {
         std::vector<uint32_t> v(e->i, 0u);
         v[e->i - 1] = e->n;
         return v;
      }
      case EK::Placeholder: return std::vector<uint32_t>(e->i, 0u);
      case EK::Literal:
      case EK::Uniform:
      case EK::Modulator:
      case EK::Param: return {};
      case EK::Feedback: return drop(max_input_delays(e->a), (size_t)e->a->out_arity);            // :459-465
      case EK::Parallel: return cat(max_input_delays(e->a), max_input_delays(e->b));              // :479-482
      case EK::Sequence:                                                                          // :483-492
         return cat(max_input_delays(e->a), drop(max_input_delays(e->b), (size_t)e->a->out_arity));
      case EK::Neg: return max_input_delays(e->a);
      case EK::Arith:
      case EK::Channel: return zipmax(max_input_delays(e->a), max_input_delays(e->b));            // :493-496
   }
   return {};
}

}  // namespace fz

using namespace fz;

#define FZ_GUARD_PTR(...)                                        \
   try { __VA_ARGS__ }                                                  \
   catch (const fz::Error& er) { fz::set_error(er.msg); return nullptr; } \
   catch (const std::exception& ex) { fz::set_error(ex.what()); return nullptr; }

extern "C" {

const char* fz_last_error(void) { return g_last_error.c_str(); }
const char* fz_version(void) { return "flowz_hip 0.1 (gfx950)"; }

void fz_expr_retain(fz_expr* e)
{
   if (e) e->refs.fetch_add(1, std::memory_order_relaxed);
}

void fz_expr_release(fz_expr* e)
{
   if (!e) return;
   if (e->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
      fz_expr_release(e->a);
      fz_expr_release(e->b);
      delete e;
   }
}

fz_expr* fz_placeholder(uint32_t i)
{
   FZ_GUARD_PTR(
      if (i < 1 || i > 4096) fail(FZ_E_INVALID, "placeholder index must be >= 1");
      auto* e = mk(EK::Placeholder);
      e->i = i;
      e->in_arity = (int)i;              // placeholder_arity, flowz.hpp:167-170
      e->out_arity = 1;
      return e;)
}

fz_expr* fz_delayed(uint32_t i, uint32_t n)
{
   FZ_GUARD_PTR(
      if (i < 1 || i > 4096) fail(FZ_E_INVALID, "placeholder index must be >= 1");
      if (n < 1) fail(FZ_E_INVALID, "delay must be >= 1 (use the plain placeholder for delay 0)");
      if (n > (1u << 20)) fail(FZ_E_UNSUPPORTED, "delay too long");
      auto* e = mk(EK::Delayed);
      e->i = i;
      e->n = n;
      e->in_arity = (int)i;              // flowz.hpp:163-166
      e->out_arity = 1;
      return e;)
}

fz_expr* fz_literal(float value)
{
   auto* e = mk(EK::Literal);
   e->value = value;
   e->in_arity = 0;                      // flowz.hpp:171-174
   return e;
}

fz_expr* fz_literal_f64(double value)
{
   auto* e = mk(EK::Literal);
   e->value = (float)value;
   e->value64 = value;
   e->f64 = true;
   e->in_arity = 0;
   return e;
}

fz_expr* fz_literal_c32(float re, float im)
{
   auto* e = mk(EK::Literal);
   e->value = re;
   e->value_im = im;
   e->cplx = true;
   e->in_arity = 0;
   return e;
}

fz_expr* fz_literal_c64(double re, double im)
{
   auto* e = mk(EK::Literal);
   e->value = (float)re;
   e->value_im = (float)im;
   e->value64 = re;
   e->value64_im = im;
   e->cplx = true;
   e->f64 = true;
   e->in_arity = 0;
   return e;
}

fz_expr* fz_uniform(uint32_t k, float initial)
{
   FZ_GUARD_PTR(
      if (k >= (1u << 20)) fail(FZ_E_INVALID, "uniform coefficient index too large");
      auto* e = mk(EK::Uniform);
      e->i = k;
      e->value = initial;
      e->in_arity = 0;
      return e;)
}

fz_expr* fz_modulator(uint32_t k)
{
   FZ_GUARD_PTR(
      if (k >= 256) fail(FZ_E_INVALID, "modulator index too large");
      auto* e = mk(EK::Modulator);
      e->i = k;
      e->in_arity = 0;
      return e;)
}

fz_expr* fz_stream_param(uint32_t k)
{
   FZ_GUARD_PTR(
      if (k >= 4096) fail(FZ_E_INVALID, "per-stream parameter index too large");
      auto* e = mk(EK::Param);
      e->i = k;
      e->in_arity = 0;
      return e;)
}

fz_expr* fz_arith(fz_op op, fz_expr* a, fz_expr* b)
{
   FZ_GUARD_PTR(
      if (!a) fail(FZ_E_INVALID, "null operand");
      if (op == FZ_OP_NEG) {
         if (a->out_arity != 1) fail(FZ_E_GRAPH, "arithmetic operand must have exactly one output wire");
         auto* e = mk(EK::Neg, a);
         e->op = op;
         e->in_arity = a->in_arity;
         return e;
      }
      if (!b) fail(FZ_E_INVALID, "null operand");
      if (op < FZ_OP_ADD || op > FZ_OP_DIV) fail(FZ_E_INVALID, "unknown arithmetic operator");
      if (a->out_arity != 1 || b->out_arity != 1)
         fail(FZ_E_GRAPH, "arithmetic operand must have exactly one output wire");
      auto* e = mk(EK::Arith, a, b);
      e->op = op;
      e->in_arity = std::max(a->in_arity, b->in_arity);     // nary fold with max, flowz.hpp:209-212
      e->out_arity = 1;                                     // otherwise<1>, :244
      return e;)
}

fz_expr* fz_channel(fz_expr* a, fz_expr* b)
{
   FZ_GUARD_PTR(
      if (!a || !b) fail(FZ_E_INVALID, "null operand");
      auto* e = mk(EK::Channel, a, b);
      e->in_arity = std::max(a->in_arity, b->in_arity);     // :209-212
      e->out_arity = a->out_arity + b->out_arity;           // :218-221
      return e;)
}

fz_expr* fz_parallel(fz_expr* a, fz_expr* b)
{
   FZ_GUARD_PTR(
      if (!a || !b) fail(FZ_E_INVALID, "null operand");
      auto* e = mk(EK::Parallel, a, b);
      e->in_arity = a->in_arity + b->in_arity;              // :195-198
      e->out_arity = a->out_arity + b->out_arity;           // :230-233
      return e;)
}

fz_expr* fz_sequence(fz_expr* a, fz_expr* b)
{
   FZ_GUARD_PTR(
      if (!a || !b) fail(FZ_E_INVALID, "null operand");
      auto* e = mk(EK::Sequence, a, b);
      e->in_arity = a->in_arity + std::max(0, b->in_arity - a->out_arity);     // :199-208
      e->out_arity = b->out_arity + std::max(0, a->out_arity - b->in_arity);   // :234-243
      return e;)
}

fz_expr* fz_feedback(fz_expr* a)
{
   FZ_GUARD_PTR(
      if (!a) fail(FZ_E_INVALID, "null operand");
      auto* e = mk(EK::Feedback, a);
      e->in_arity = std::max(0, a->in_arity - a->out_arity);                   // :175-181
      e->out_arity = a->out_arity;                                             // :222-225
      return e;)
}

int fz_input_arity(const fz_expr* e)
{
   if (!e) { set_error("null expression"); return FZ_E_INVALID; }
   return e->in_arity;
}

int fz_output_arity(const fz_expr* e)
{
   if (!e) { set_error("null expression"); return FZ_E_INVALID; }
   return e->out_arity;
}

int fz_max_input_delays(const fz_expr* e, uint32_t* out, uint32_t cap)
{
   if (!e) { set_error("null expression"); return FZ_E_INVALID; }
   auto v = max_input_delays(e);
   for (size_t k = 0; k < v.size() && k < cap; ++k) out[k] = v[k];
   return (int)v.size();
}

}  // extern "C"
