// Expression trees and the static-analysis transforms of the Flowz EDSL.
//
// Re-designed, not translated: the reference computes these at C++ compile time with
// Boost.Proto transforms over expression *types* (flowz.hpp:162-246 input/output_arity,
// :443-506 max_input_delays); here they are plain functions over a reference-counted run-time
// tree, the arities evaluated once when a node is constructed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

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

// per-wire YOUNGEST read, flowz.hpp:503 (min_input_delays: the same recursion with the combiner of :389-421): 0 = the wire is read
// without a delay somewhere, n = only through _i[_n] and older, -1 = not read at all
static std::vector<int> zipmin(const std::vector<int>& l, const std::vector<int>& r)
{
   std::vector<int> out(std::max(l.size(), r.size()), -1);
   for (size_t k = 0; k < out.size(); ++k) {
      const int x = k < l.size() ? l[k] : -1, y = k < r.size() ? r[k] : -1;
      out[k] = x < 0 ? y : y < 0 ? x : std::min(x, y);
   }
   return out;
}

static std::vector<int> min_input_delays(const fz_expr* e)
{
   auto dropn = [](std::vector<int> v, size_t n) {
      if (n >= v.size()) return std::vector<int>();
      v.erase(v.begin(), v.begin() + (long)n);
      return v;
   };
   auto catn = [](std::vector<int> a, const std::vector<int>& b) {
      a.insert(a.end(), b.begin(), b.end());
      return a;
   };
   switch (e->kind) {
      case EK::Delayed:
      case EK::Placeholder: {
         std::vector<int> v(e->i, -1);
         v[e->i - 1] = e->kind == EK::Delayed ? (int)e->n : 0;
         return v;
      }
      case EK::Literal:
      case EK::Uniform:
      case EK::Modulator:
      case EK::Param: return {};
      case EK::Feedback: return dropn(min_input_delays(e->a), (size_t)e->a->out_arity);
      case EK::Parallel: return catn(min_input_delays(e->a), min_input_delays(e->b));
      case EK::Sequence: return catn(min_input_delays(e->a), dropn(min_input_delays(e->b), (size_t)e->a->out_arity));
      case EK::Neg: return min_input_delays(e->a);
      case EK::Arith:
      case EK::Channel: return zipmin(min_input_delays(e->a), min_input_delays(e->b));
   }
   return {};
}

// Where the SHIPPED reference evaluates a feedback differently from its own arity table.  compile() rewrites `~x` into
// binary_feedback(promise, future) (flowz.hpp:862-884: the chain  front |= s1 |= ... |= sn  is cut before the first box that needs none
// of the wires in front of it undelayed: promise = front |= s1 .. sk, future = the rest) and binary_feedback hands the future part the
// external inputs from position std::min(0, in(promise) - out(future)) = 0 on (:1045-1050, "TODO" there) instead of from behind the
// promise's own.  Harmless unless the promise part consumes external inputs AND the future part reads some: then the reference's
// closure reads the wrong wires, and this library -- which routes per the arity table, :162-246 -- computes other values.
// Returns the number of external inputs the promise part takes in that case, else 0.
uint32_t feedback_promise_inputs(const fz_expr* fb)
{
   if (fb->kind != EK::Feedback) return 0;
   std::vector<const fz_expr*> chain;
   std::vector<const fz_expr*> todo{fb->a};
   while (!todo.empty()) {                                  // boxes of the body's `|=` chain, in evaluation order
      const fz_expr* x = todo.back();
      todo.pop_back();
      if (x->kind == EK::Sequence) {
         todo.push_back(x->b);
         todo.push_back(x->a);
      } else {
         chain.push_back(x);
      }
   }
   // arity and youngest reads of the remainders r_k = s(k+1) |= ... |= sn, built from the back (the arity table of :162-246)
   const size_t n = chain.size();
   std::vector<int> rin(n + 1, 0), rout(n + 1, 0);
   std::vector<std::vector<int>> rmin(n + 1);
   for (size_t k = n; k-- > 0;) {
      const fz_expr* s = chain[k];
      if (k + 1 == n) {
         rin[k] = s->in_arity;
         rout[k] = s->out_arity;
         rmin[k] = min_input_delays(s);
      } else {
         rin[k] = s->in_arity + std::max(0, rin[k + 1] - s->out_arity);
         rout[k] = rout[k + 1] + std::max(0, s->out_arity - rin[k + 1]);
         std::vector<int> tail = rmin[k + 1];
         if ((size_t)s->out_arity >= tail.size()) tail.clear();
         else tail.erase(tail.begin(), tail.begin() + s->out_arity);
         rmin[k] = min_input_delays(s);
         rmin[k].insert(rmin[k].end(), tail.begin(), tail.end());
      }
   }
   int lin = fb->a->out_arity, lout = fb->a->out_arity;     // the front panel: out(x) wires passed through
   for (size_t k = 0; k < n; ++k) {
      bool direct = false;
      for (int w = 0; w < lout && w < (int)rmin[k].size(); ++w) direct = direct || rmin[k][(size_t)w] == 0;
      if (!direct) {
         const int own = lin - rout[k];
         return own > 0 && rin[k] > lout ? (uint32_t)own : 0u;
      }
      const fz_expr* s = chain[k];
      lin = lin + std::max(0, s->in_arity - lout);
      lout = s->out_arity + std::max(0, lout - s->in_arity);
   }
   return 0;                                                // (no cut: a delay-free loop, rejected by the lowering)
}

// ---- recipes: an expression as text (kernel manifests, fz_kernel_cache.cpp) --------------------------------------------------
// One line per node of the DAG in dependency order, shared sub-expressions once: "<kind letter> <fields>"; operands are line numbers.
// Floating-point values travel as bit patterns.
static uint32_t bits32(float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }
static uint64_t bits64(double v) { uint64_t u; std::memcpy(&u, &v, 8); return u; }
static float from32(uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; }
static double from64(uint64_t u) { double v; std::memcpy(&v, &u, 8); return v; }

std::string serialize_expr(const fz_expr* root)
{
   std::map<const fz_expr*, size_t> id;
   std::string out;
   // (iterative post-order: cascades of hundreds of stages are deep left spines)
   std::vector<std::pair<const fz_expr*, int>> st{{root, 0}};
   while (!st.empty()) {
      auto& top = st.back();
      const fz_expr* e = top.first;
      if (id.count(e)) { st.pop_back(); continue; }
      if (top.second == 0) { top.second = 1; if (e->a && !id.count(e->a)) { st.push_back({e->a, 0}); continue; } }
      if (top.second == 1) { top.second = 2; if (e->b && !id.count(e->b)) { st.push_back({e->b, 0}); continue; } }
      char buf[160];
      const unsigned long a = e->a ? (unsigned long)id[e->a] : 0ul, b = e->b ? (unsigned long)id[e->b] : 0ul;
      switch (e->kind) {
         case EK::Placeholder: std::snprintf(buf, sizeof buf, "P %u\n", e->i); break;
         case EK::Delayed: std::snprintf(buf, sizeof buf, "D %u %u\n", e->i, e->n); break;
         case EK::Literal:
            std::snprintf(buf, sizeof buf, "L %d %d %08x %08x %016llx %016llx\n", (int)e->f64, (int)e->cplx, bits32(e->value), bits32(e->value_im),
                          (unsigned long long)bits64(e->value64), (unsigned long long)bits64(e->value64_im));
            break;
         case EK::Uniform: std::snprintf(buf, sizeof buf, "U %u %08x\n", e->i, bits32(e->value)); break;
         case EK::Param: std::snprintf(buf, sizeof buf, "Q %u\n", e->i); break;
         case EK::Modulator: std::snprintf(buf, sizeof buf, "M %u\n", e->i); break;
         case EK::Arith: std::snprintf(buf, sizeof buf, "A %d %lu %lu\n", (int)e->op, a, b); break;
         case EK::Neg: std::snprintf(buf, sizeof buf, "N %lu\n", a); break;
         case EK::Channel: std::snprintf(buf, sizeof buf, "C %lu %lu\n", a, b); break;
         case EK::Parallel: std::snprintf(buf, sizeof buf, "B %lu %lu\n", a, b); break;
         case EK::Sequence: std::snprintf(buf, sizeof buf, "S %lu %lu\n", a, b); break;
         case EK::Feedback: std::snprintf(buf, sizeof buf, "F %lu\n", a); break;
      }
      out += buf;
      const size_t n = id.size();
      id[e] = n;
      st.pop_back();
   }
   return out;
}

// the expression of a recipe (one reference, the caller's); nullptr + fz_last_error for text that is not one
fz_expr* parse_expr(const std::string& text)
{
   std::vector<fz_expr*> nodes;
   auto cleanup = [&] { for (fz_expr* e : nodes) fz_expr_release(e); };
   auto bad = [&](const char* why) -> fz_expr* { cleanup(); set_error(std::string("recipe: ") + why); return nullptr; };
   size_t pos = 0;
   while (pos < text.size()) {
      size_t eol = text.find('\n', pos);
      if (eol == std::string::npos) eol = text.size();
      const std::string ln = text.substr(pos, eol - pos);
      pos = eol + 1;
      if (ln.empty()) continue;
      unsigned i = 0, n = 0, f64 = 0, cx = 0, v = 0, vi = 0;
      unsigned long a = 0, b = 0;
      unsigned long long d = 0, di = 0;
      int op = 0;
      fz_expr* e = nullptr;
      auto ref = [&](unsigned long k) -> fz_expr* { return k < nodes.size() ? nodes[k] : nullptr; };
      const char* s = ln.c_str() + 1;
      switch (ln[0]) {
         case 'P': if (std::sscanf(s, "%u", &i) == 1) e = fz_placeholder(i); break;
         case 'D': if (std::sscanf(s, "%u %u", &i, &n) == 2) e = fz_delayed(i, n); break;
         case 'L':
            if (std::sscanf(s, "%u %u %x %x %llx %llx", &f64, &cx, &v, &vi, &d, &di) == 6)
               e = cx ? (f64 ? fz_literal_c64(from64(d), from64(di)) : fz_literal_c32(from32(v), from32(vi))) : (f64 ? fz_literal_f64(from64(d)) : fz_literal(from32(v)));
            break;
         case 'U': if (std::sscanf(s, "%u %x", &i, &v) == 2) e = fz_uniform(i, from32(v)); break;
         case 'Q': if (std::sscanf(s, "%u", &i) == 1) e = fz_stream_param(i); break;
         case 'M': if (std::sscanf(s, "%u", &i) == 1) e = fz_modulator(i); break;
         case 'A': if (std::sscanf(s, "%d %lu %lu", &op, &a, &b) == 3 && ref(a) && ref(b)) e = fz_arith((fz_op)op, ref(a), ref(b)); break;
         case 'N': if (std::sscanf(s, "%lu", &a) == 1 && ref(a)) e = fz_arith(FZ_OP_NEG, ref(a), nullptr); break;
         case 'C': if (std::sscanf(s, "%lu %lu", &a, &b) == 2 && ref(a) && ref(b)) e = fz_channel(ref(a), ref(b)); break;
         case 'B': if (std::sscanf(s, "%lu %lu", &a, &b) == 2 && ref(a) && ref(b)) e = fz_parallel(ref(a), ref(b)); break;
         case 'S': if (std::sscanf(s, "%lu %lu", &a, &b) == 2 && ref(a) && ref(b)) e = fz_sequence(ref(a), ref(b)); break;
         case 'F': if (std::sscanf(s, "%lu", &a) == 1 && ref(a)) e = fz_feedback(ref(a)); break;
         default: return bad("unknown node kind");
      }
      if (!e) return bad("malformed line");
      nodes.push_back(e);
   }
   if (nodes.empty()) return bad("empty");
   fz_expr* root = nodes.back();
   fz_expr_retain(root);
   cleanup();
   return root;
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
      // the logical operators of C++ on arithmetic operands, spelled with comparisons (values identical; nothing to short-circuit):
      //   !a = (a == 0)     a && b = (a != 0) * (b != 0)     a || b = ((a != 0) + (b != 0)) != 0
      if (op == FZ_OP_NOT || op == FZ_OP_AND || op == FZ_OP_OR) {
         if (op != FZ_OP_NOT && !b) fail(FZ_E_INVALID, "null operand");
         struct Hold { fz_expr* e; ~Hold() { if (e) fz_expr_release(e); } };
         auto check = [](fz_expr* e) { if (!e) throw Error{FZ_E_GRAPH, fz_last_error()}; return e; };
         Hold zero{check(fz_literal(0.f))};
         if (op == FZ_OP_NOT) return check(fz_arith(FZ_OP_EQ, a, zero.e));
         Hold ta{check(fz_arith(FZ_OP_NE, a, zero.e))}, tb{check(fz_arith(FZ_OP_NE, b, zero.e))};
         if (op == FZ_OP_AND) return check(fz_arith(FZ_OP_MUL, ta.e, tb.e));
         Hold sum{check(fz_arith(FZ_OP_ADD, ta.e, tb.e))};
         return check(fz_arith(FZ_OP_NE, sum.e, zero.e));
      }
      if (!b) fail(FZ_E_INVALID, "null operand");
      if (!((op >= FZ_OP_ADD && op <= FZ_OP_DIV) || (op >= FZ_OP_LT && op <= FZ_OP_NE))) fail(FZ_E_INVALID, "unknown arithmetic operator");
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
