// compile(): expression tree -> flat per-sample DAG + delay-line (state) layout.
//
// What the reference does at C++ compile time with the front panel (flowz.hpp:261-277), the
// canonicaliser (make_canonical / split_future_subexpr, :794-935: every `~x` becomes
// binary_feedback(promise, future) so that an evaluation order exists) and build_state
// (:685-725: nested tuples of std::array<float,D> per sequence/feedback node) is done here at
// run time on a wire graph:
//   1. wiring      -- every combinator routes wire ids per the arity table (sequence :974-999,
//                     parallel :1087-1099, channel :765-768, feedback: the first out(a) inputs
//                     of `a` are its own outputs, :1018-1027/:1043-1069);
//   2. ordering    -- a delayed read `_i[_n]` is a *source* (it only reads state), so feedback
//                     cycles are broken exactly where the reference splits promise/future; a
//                     cycle with no delayed read is a delay-free loop and is rejected;
//   3. sharing     -- value-identical nodes are merged (same op, same operands) and there is ONE
//                     delay line per delayed wire, as deep as its deepest reader (max_input_delays
//                     :502); a 6-stage DF1 cascade therefore carries 14 floats of state where the
//                     reference's nested tuples carry 24 (TODO.md:59) -- values are unaffected;
//   4. no algebra  -- nothing is re-associated, folded or simplified: one float32 rounding per
//                     node of the user's tree (proto::_default, :769-772).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

#include "fz_internal.hpp"

namespace fz {
namespace {

constexpr uint32_t K_FWD = 100;   // forward reference of a fed-back wire (resolved before output)

struct Raw {
   uint32_t kind;
   int a = -1, b = -1;
   float value = 0.f;
   uint32_t n = 0;
   bool f64 = false;        // literal: C++ double; arithmetic: decided after the forward references resolve
   double value64 = 0.0;
};

struct Elab {
   bool typed = false;      // fz_compile_typed: delay lines keep the pushed type (ResultType, flowz.hpp:585-644)
   std::vector<Raw> raw;
   // std::complex<float> wires (test/tests.cpp:206-207) are lowered to PAIRS of float nodes while the
   // wire graph is built: a complex wire is known by its real part (always a fresh node), imag_of maps
   // it to the imaginary part.  The operators expand the way <complex> defines them for complex<float>:
   //   z*s, s*z = (re*s, im*s)    z/s = (re/s, im/s)    z+s, s+z = (re+s, im)    z-s = (re-s, im)
   //   s-z = ((-re)+s, -im)   [complex r = -z; r += s]    z+-w componentwise    -z = (-re, -im)
   //   z*w = (ac-bd, ad+bc): the arithmetic of __mulsc3 (_Complex float) for finite values; its
   //         inf/nan recovery branch is not reproduced.   s/z, z/w: __divsc3, see wide_div below.
   std::map<int, int> imag_of;
   std::string mix_error;   // a complex / scalar type mismatch seen while feedback wire types were still assumptions
   int fixpoint_depth = 0;
   uint32_t promise_inputs = 0;   // a feedback whose promise part takes external inputs next to a future part that reads some

   int add(uint32_t kind, int a = -1, int b = -1, float value = 0.f, uint32_t n = 0)
   {
      raw.push_back(Raw{kind, a, b, value, n, false, 0.0});
      return (int)raw.size() - 1;
   }

   static std::vector<int> slice(const std::vector<int>& v, size_t lo, size_t hi)
   {
      hi = std::min(hi, v.size());
      if (lo >= hi) return {};
      return std::vector<int>(v.begin() + (long)lo, v.begin() + (long)hi);
   }

   int one(const fz_expr* e, const std::vector<int>& ins)
   {
      auto w = run(e, ins);
      if (w.size() != 1) fail(FZ_E_GRAPH, "arithmetic operand must have exactly one output wire");
      return w[0];
   }

   std::vector<int> run(const fz_expr* e, const std::vector<int>& ins)
   {
      switch (e->kind) {
         case EK::Placeholder:                                          // place_the_holder :941-948
            if (e->i > ins.size()) fail(FZ_E_GRAPH, "placeholder _" + std::to_string(e->i) + " has no wire to bind to");
            return {ins[e->i - 1]};
         case EK::Delayed:                                              // place_delay :950-958
            if (e->i > ins.size()) fail(FZ_E_GRAPH, "placeholder _" + std::to_string(e->i) + " has no wire to bind to");
            if (imag_of.count(ins[e->i - 1])) {
               if (!typed)
                  fail(FZ_E_GRAPH, "a std::complex wire cannot be read through a delay line: compile() stores float state (flowz.hpp:1245); "
                                   "fz_compile_typed keeps the wire's type");
               // one float line per part
               const int src = ins[e->i - 1];
               const int re = add(FZ_IR_DELAY, src, -1, 0.f, e->n), im = add(FZ_IR_DELAY, imag_of[src], -1, 0.f, e->n);
               raw[(size_t)re].f64 = raw[(size_t)src].f64;      // std::complex<double>: two double lines
               raw[(size_t)im].f64 = raw[(size_t)imag_of[src]].f64;
               imag_of[re] = im;
               return {re};
            } else {
               const int id = add(FZ_IR_DELAY, ins[e->i - 1], -1, 0.f, e->n);
               if (typed) raw[(size_t)id].f64 = raw[(size_t)ins[e->i - 1]].f64;   // the line stores what is pushed (tests.cpp:219)
               return {id};
            }
         case EK::Literal: {
            if (e->cplx) {
               int re = add(FZ_IR_CONST, -1, -1, e->value), im = add(FZ_IR_CONST, -1, -1, e->value_im);
               if (e->f64) {                                  // std::complex<double>: both parts are double terminals
                  raw[(size_t)re].f64 = raw[(size_t)im].f64 = true;
                  raw[(size_t)re].value64 = e->value64;
                  raw[(size_t)im].value64 = e->value64_im;
               }
               imag_of[re] = im;
               return {re};
            }
            int id = add(FZ_IR_CONST, -1, -1, e->value);
            raw[(size_t)id].f64 = e->f64;
            raw[(size_t)id].value64 = e->value64;
            return {id};
         }
         case EK::Uniform: return {add(FZ_IR_CONST, -1, -1, e->value, e->i + 1)};   // n = id + 1: own slot
         case EK::Param: return {add(FZ_IR_PARAM, -1, -1, 0.f, e->i)};
         case EK::Modulator: return {add(FZ_IR_MOD, -1, -1, 0.f, e->i)};
         case EK::Arith: {                                              // _default<eval_it> :769-772
            int a = one(e->a, ins), b = one(e->b, ins);
            if (e->op >= FZ_OP_LT && e->op <= FZ_OP_NE) {               // a comparison: 1.0f / 0.0f, a float whatever was compared
               if (imag_of.count(a) || imag_of.count(b)) fail(FZ_E_GRAPH, "comparison operators do not apply to std::complex wires");
               return {add(FZ_IR_LT + (uint32_t)(e->op - FZ_OP_LT), a, b)};
            }
            uint32_t k = e->op == FZ_OP_ADD ? FZ_IR_ADD : e->op == FZ_OP_SUB ? FZ_IR_SUB
                       : e->op == FZ_OP_MUL ? FZ_IR_MUL : FZ_IR_DIV;
            if (imag_of.count(a) || imag_of.count(b)) return {complex_arith(k, a, b)};
            return {arith(k, a, b)};
         }
         case EK::Neg: {
            int a = one(e->a, ins);
            int re = arith(FZ_IR_NEG, a);
            if (imag_of.count(a)) imag_of[re] = arith(FZ_IR_NEG, imag_of[a]);
            return {re};
         }
         case EK::Channel: {                                            // :765-768 same inputs to both
            auto l = run(e->a, ins), r = run(e->b, ins);
            l.insert(l.end(), r.begin(), r.end());
            return l;
         }
         case EK::Parallel: {                                           // :1087-1099 inputs split at in(a)
            size_t na = (size_t)e->a->in_arity, nb = (size_t)e->b->in_arity;
            if (na + nb > ins.size()) fail(FZ_E_GRAPH, "parallel box needs more input wires than it is given");
            auto l = run(e->a, slice(ins, 0, na)), r = run(e->b, slice(ins, na, na + nb));
            l.insert(l.end(), r.begin(), r.end());
            return l;
         }
         case EK::Sequence: {                                           // :974-999
            size_t na = (size_t)e->a->in_arity, nb = (size_t)e->b->in_arity;
            if (na > ins.size()) fail(FZ_E_GRAPH, "sequence needs more input wires than it is given");
            auto ao = run(e->a, slice(ins, 0, na));
            std::vector<int> bi = ao;                                   // a_out ++ rest of the inputs
            for (size_t k = na; k < ins.size(); ++k) bi.push_back(ins[k]);
            auto bo = run(e->b, bi);
            for (size_t k = nb; k < ao.size(); ++k) bo.push_back(ao[k]);   // wires around the next box
            return bo;
         }
         case EK::Feedback: {                                           // :1031-1074, arity-table routing
            size_t k = (size_t)e->a->out_arity;
            promise_inputs = std::max(promise_inputs, feedback_promise_inputs(e));
            if (!typed) {
               std::vector<int> in2;
               for (size_t j = 0; j < k; ++j) in2.push_back(add(K_FWD));
               std::vector<int> fwd = in2;
               in2.insert(in2.end(), ins.begin(), ins.end());
               auto ao = run(e->a, in2);
               if (ao.size() != k) fail(FZ_E_GRAPH, "feedback body arity mismatch");
               for (int w : ao)
                  if (imag_of.count(w))
                     fail(FZ_E_GRAPH, "a std::complex wire cannot be fed back: compile() stores float state (flowz.hpp:1245); "
                                      "fz_compile_typed keeps the wire's type");
               for (size_t j = 0; j < k; ++j) raw[(size_t)fwd[j]].a = ao[j];
               return ao;
            }
            // Typed: the type of a fed-back wire is the least one that is consistent around the loop -- what ResultType's
            // absorber computes (:602-620: the recursion variable is absorbed by whatever it meets; float is the bottom of
            // the usual arithmetic conversions).  Start from float, elaborate, raise, repeat (at most twice per wire).
            std::vector<uint8_t> assume(k, 0);
            for (int round = 0;; ++round) {
               if (round > 4) fail(FZ_E_GRAPH, "internal: feedback wire types do not settle");
               const size_t mark = raw.size();
               const std::map<int, int> imag_mark = imag_of;
               const std::string error_mark = mix_error;
               struct Depth { int& d; Depth(int& x) : d(x) { ++d; } ~Depth() { --d; } } depth(fixpoint_depth);
               std::vector<int> in2, fwd_re(k), fwd_im(k, -1);
               for (size_t j = 0; j < k; ++j) {
                  fwd_re[j] = add(K_FWD);
                  if (assume[j] == 1 || assume[j] == 3) raw[(size_t)fwd_re[j]].f64 = true;
                  if (assume[j] >= 2) {
                     fwd_im[j] = add(K_FWD);
                     raw[(size_t)fwd_im[j]].f64 = assume[j] == 3;
                     imag_of[fwd_re[j]] = fwd_im[j];
                  }
                  in2.push_back(fwd_re[j]);
               }
               in2.insert(in2.end(), ins.begin(), ins.end());
               auto ao = run(e->a, in2);
               if (ao.size() != k) fail(FZ_E_GRAPH, "feedback body arity mismatch");
               bool same = true;
               for (size_t j = 0; j < k; ++j) {
                  const uint8_t act = (imag_of.count(ao[j]) ? 2 : 0) + (raw[(size_t)ao[j]].f64 ? 1 : 0);   // 0 float, 1 double, 2 complex<float>, 3 complex<double>
                  if (act == assume[j]) continue;
                  same = false;
                  // float is absorbed by everything, double by complex<double>; complex<float> meets neither double nor complex<double>
                  const uint8_t lo = std::min(act, assume[j]), hi = std::max(act, assume[j]);
                  if (lo != 0 && !(lo == 1 && hi == 3))
                     fail(FZ_E_GRAPH, "a fed-back wire has two types that C++ does not convert into each other (double / std::complex<float> / std::complex<double>)");
                  assume[j] = hi;
               }
               if (same) {
                  for (size_t j = 0; j < k; ++j) {
                     raw[(size_t)fwd_re[j]].a = ao[j];
                     if (fwd_im[j] >= 0) raw[(size_t)fwd_im[j]].a = imag_of[ao[j]];
                  }
                  return ao;
               }
               raw.resize(mark);                                  // forget this attempt
               imag_of = imag_mark;
               mix_error = error_mark;
            }
         }
      }
      fail(FZ_E_GRAPH, "unknown expression node");
   }

   // arithmetic node; its C++ type (double if an operand is) is known right away: forward references
   // and delayed reads are float
   int arith(uint32_t kind, int a, int b = -1)
   {
      int id = add(kind, a, b);
      raw[(size_t)id].f64 = raw[(size_t)a].f64 || (b >= 0 && raw[(size_t)b].f64);
      return id;
   }

   int complex_arith(uint32_t k, int a, int b)
   {
      const bool ca = imag_of.count(a) != 0, cb = imag_of.count(b) != 0;
      // operator(complex<T>, T) / (complex<T>, complex<T>) only: the part type of the complex operand(s) must be the
      // type of the other operand
      const bool za = raw[(size_t)a].f64, zb = raw[(size_t)b].f64;
      if (za != zb) {
         // inside a typed feedback body the operand may be a recursion variable whose type is still being raised: the
         // verdict waits until the types around the loop have settled (lower() checks mix_error at the end)
         const char* msg = "std::complex<float> does not mix with double operands nor std::complex<double> with float ones (C++ has no such operator)";
         if (!typed || !fixpoint_depth) fail(FZ_E_GRAPH, msg);
         mix_error = msg;
      }
      const bool dbl = za || zb;
      int re = -1, im = -1;
      if (ca && cb) {
         const int ar = a, ai = imag_of[a], br = b, bi = imag_of[b];
         switch (k) {
            case FZ_IR_ADD: case FZ_IR_SUB: re = arith(k, ar, br); im = arith(k, ai, bi); break;
            case FZ_IR_MUL: {
               const int ac = arith(FZ_IR_MUL, ar, br), bd = arith(FZ_IR_MUL, ai, bi);
               const int ad = arith(FZ_IR_MUL, ar, bi), bc = arith(FZ_IR_MUL, ai, br);
               re = arith(FZ_IR_SUB, ac, bd);
               im = arith(FZ_IR_ADD, ad, bc);
               break;
            }
            default: return dbl ? smith_div(ar, ai, br, bi) : wide_div(ar, ai, br, bi);
         }
      } else if (ca) {                                      // complex (op) scalar
         const int ar = a, ai = imag_of[a];
         switch (k) {
            case FZ_IR_ADD: case FZ_IR_SUB: re = arith(k, ar, b); im = ai; break;
            default: re = arith(k, ar, b); im = arith(k, ai, b); break;       // MUL, DIV
         }
      } else {                                              // scalar (op) complex
         const int br = b, bi = imag_of[b];
         switch (k) {
            case FZ_IR_ADD: re = arith(FZ_IR_ADD, br, a); im = bi; break;
            case FZ_IR_SUB: re = arith(FZ_IR_ADD, arith(FZ_IR_NEG, br), a); im = arith(FZ_IR_NEG, bi); break;
            case FZ_IR_MUL: re = arith(FZ_IR_MUL, br, a); im = arith(FZ_IR_MUL, bi, a); break;
            default: {                                      // s / w: complex<float> r = s; r /= w  (<complex>), b = +0.f
               const int zero = add(FZ_IR_CONST, -1, -1, 0.f);
               if (dbl) {
                  raw[(size_t)zero].f64 = true;
                  raw[(size_t)zero].value64 = 0.0;
                  return smith_div(a, zero, br, bi);
               }
               return wide_div(a, zero, br, bi);
            }
         }
      }
      imag_of[re] = im;
      return re;
   }

   // (a + ib) / (c + id) as g++ computes it for std::complex<float> / float _Complex: libgcc's __divsc3, which handles
   // float with double precision (libgcc2.c, "float is handled with double precision when double precision hardware is
   // available": the simple formula, no Smith scaling):
   //    aa = a, bb = b, cc = c, dd = d (double);  denom = cc*cc + dd*dd;
   //    x = (float)((aa*cc + bb*dd) / denom);     y = (float)((bb*cc - aa*dd) / denom)
   // Its recovery branch for NaN results (zero or infinite operands) is not reproduced: finite values, nonzero divisor.
   int wide_div(int a, int b, int c, int d)
   {
      auto widen = [&](int v) {
         const int id = add(FZ_IR_WIDEN, v);
         raw[(size_t)id].f64 = true;
         return id;
      };
      const int aa = widen(a), bb = widen(b), cc = widen(c), dd = widen(d);
      const int denom = arith(FZ_IR_ADD, arith(FZ_IR_MUL, cc, cc), arith(FZ_IR_MUL, dd, dd));
      const int xn = arith(FZ_IR_ADD, arith(FZ_IR_MUL, aa, cc), arith(FZ_IR_MUL, bb, dd));
      const int yn = arith(FZ_IR_SUB, arith(FZ_IR_MUL, bb, cc), arith(FZ_IR_MUL, aa, dd));
      const int re = add(FZ_IR_NARROW, arith(FZ_IR_DIV, xn, denom)), im = add(FZ_IR_NARROW, arith(FZ_IR_DIV, yn, denom));
      imag_of[re] = im;
      return re;
   }

   // (a + ib) / (c + id) for std::complex<double>: libgcc's __divdc3 = Smith's method,
   //    |c| < |d| :  ratio = c/d, denom = c*ratio + d, x = (a*ratio + b)/denom, y = (b*ratio - a)/denom
   //    else      :  ratio = d/c, denom = d*ratio + c, x = (b*ratio + a)/denom, y = (b - a*ratio)/denom
   // (its scaling branches for extreme magnitudes and its NaN recovery are not restated; 200 000 random quotients of
   // std::complex<double> compiled by g++ here agree bit for bit).  Both sides are evaluated, the result is selected.
   int smith_div(int a, int b, int c, int d)
   {
      auto sel = [&](int m, int x, int y) {
         const int id = add(FZ_IR_SELECT, m, x, 0.f, (uint32_t)y);
         raw[(size_t)id].f64 = true;
         return id;
      };
      const int m = arith(FZ_IR_ABSLT, c, d);
      const int r1 = arith(FZ_IR_DIV, c, d), den1 = arith(FZ_IR_ADD, arith(FZ_IR_MUL, c, r1), d);
      const int x1 = arith(FZ_IR_DIV, arith(FZ_IR_ADD, arith(FZ_IR_MUL, a, r1), b), den1);
      const int y1 = arith(FZ_IR_DIV, arith(FZ_IR_SUB, arith(FZ_IR_MUL, b, r1), a), den1);
      const int r2 = arith(FZ_IR_DIV, d, c), den2 = arith(FZ_IR_ADD, arith(FZ_IR_MUL, d, r2), c);
      const int x2 = arith(FZ_IR_DIV, arith(FZ_IR_ADD, arith(FZ_IR_MUL, b, r2), a), den2);
      const int y2 = arith(FZ_IR_DIV, arith(FZ_IR_SUB, b, arith(FZ_IR_MUL, a, r2)), den2);
      const int re = sel(m, x1, x2), im = sel(m, y1, y2);
      imag_of[re] = im;
      return re;
   }

   int resolve(int id) const
   {
      size_t guard = 0;
      while (raw[(size_t)id].kind == K_FWD) {
         id = raw[(size_t)id].a;
         if (id < 0 || ++guard > raw.size())
            fail(FZ_E_GRAPH, "delay-free feedback loop (a fed-back wire is wired straight to itself)");
      }
      return id;
   }
};

uint32_t bits_of(float f)
{
   uint32_t u;
   std::memcpy(&u, &f, 4);
   return u;
}

uint64_t bits_of64(double f)
{
   uint64_t u;
   std::memcpy(&u, &f, 8);
   return u;
}

}  // namespace

Graph lower(const fz_expr* e, const LowerOptions& opt)
{
   if (!e) fail(FZ_E_INVALID, "null expression");
   Elab el;
   el.typed = opt.typed;
   const uint32_t n_in_wires = (uint32_t)e->in_arity;
   // front panel: one wire per external input; typed programs: a double wire takes two frame slots (low, high word),
   // a std::complex<float> wire two (re, im)
   std::vector<int> ins;
   std::vector<uint8_t> in_dtype(n_in_wires, 0);
   uint32_t n_in = 0;                                        // frame slots
   for (uint32_t i = 0; i < n_in_wires; ++i) {
      const uint8_t dt = (opt.typed && i < opt.in_dtype.size()) ? opt.in_dtype[i] : 0;
      if (dt > 3) fail(FZ_E_INVALID, "unknown input wire type");
      in_dtype[i] = dt;
      const int id = el.add(FZ_IR_INPUT, -1, -1, 0.f, n_in);
      if (dt == 1 || dt == 3) el.raw[(size_t)id].f64 = true;
      if (dt == 2) el.imag_of[id] = el.add(FZ_IR_INPUT, -1, -1, 0.f, n_in + 1);
      if (dt == 3) {                                           // complex<double>: (re double, im double) = 4 slots
         const int im = el.add(FZ_IR_INPUT, -1, -1, 0.f, n_in + 2);
         el.raw[(size_t)im].f64 = true;
         el.imag_of[id] = im;
      }
      n_in += dt == 3 ? 4 : dt ? 2 : 1;
      ins.push_back(id);
   }
   std::vector<int> input_nodes;                             // every INPUT node (kept alive even when unused)
   for (size_t i = 0; i < el.raw.size(); ++i) input_nodes.push_back((int)i);
   std::vector<int> out_wires = el.run(e, ins);
   if (!el.mix_error.empty()) fail(FZ_E_GRAPH, el.mix_error);
   if ((int)out_wires.size() != e->out_arity) fail(FZ_E_GRAPH, "output arity mismatch between arity table and routing");
   if (out_wires.empty()) fail(FZ_E_GRAPH, "graph has no output wire");
   // output frame slots: a complex wire takes two (re, im); typed programs: a double wire two (low, high word)
   std::vector<int> outs;
   std::vector<uint8_t> out_part;
   for (int w : out_wires) {
      auto it = el.imag_of.find(w);
      outs.push_back(w);
      out_part.push_back(it == el.imag_of.end() ? 0 : 1);
      if (it != el.imag_of.end()) {
         outs.push_back(it->second);
         out_part.push_back(2);
      }
   }

   auto& raw = el.raw;
   const size_t N = raw.size();
   // resolve forward references in every operand
   for (size_t i = 0; i < N; ++i) {
      if (raw[i].kind == K_FWD) continue;
      if (raw[i].a >= 0) raw[i].a = el.resolve(raw[i].a);
      if (raw[i].b >= 0) raw[i].b = el.resolve(raw[i].b);
      if (raw[i].kind == FZ_IR_SELECT) raw[i].n = (uint32_t)el.resolve((int)raw[i].n);
   }
   for (auto& o : outs) o = el.resolve(o);

   // reachability: same-sample operands, plus the source of every reachable delayed read
   std::vector<char> live(N, 0);
   {
      std::vector<int> work(outs.begin(), outs.end());
      for (int v : input_nodes) work.push_back(v);
      while (!work.empty()) {
         int v = work.back();
         work.pop_back();
         if (live[(size_t)v]) continue;
         live[(size_t)v] = 1;
         if (raw[(size_t)v].a >= 0) work.push_back(raw[(size_t)v].a);
         if (raw[(size_t)v].b >= 0) work.push_back(raw[(size_t)v].b);
         if (raw[(size_t)v].kind == FZ_IR_SELECT) work.push_back((int)raw[(size_t)v].n);
      }
   }

   // topological order of the same-sample dependencies (delayed reads are sources)
   std::vector<int> order;
   {
      std::vector<char> color(N, 0);   // 0 white, 1 on stack, 2 done
      struct Frame { int v; int stage; };
      auto visit = [&](int root) {
         if (color[(size_t)root]) return;
         std::vector<Frame> st;
         st.push_back({root, 0});
         color[(size_t)root] = 1;
         while (!st.empty()) {
            Frame& f = st.back();
            const Raw& r = raw[(size_t)f.v];
            int dep = -1;
            if (r.kind != FZ_IR_DELAY) {
               if (f.stage == 0) { dep = r.a; f.stage = 1; }
               else if (f.stage == 1) { dep = r.b; f.stage = 2; }
               else if (f.stage == 2 && r.kind == FZ_IR_SELECT) { dep = (int)r.n; f.stage = 4; }
               else f.stage = 3;
            } else f.stage = 3;
            if (f.stage == 4 && dep < 0) f.stage = 3;
            if (f.stage == 3) {
               color[(size_t)f.v] = 2;
               order.push_back(f.v);
               st.pop_back();
               continue;
            }
            if (dep < 0) continue;
            if (color[(size_t)dep] == 1)
               fail(FZ_E_GRAPH, "delay-free feedback loop: every cycle needs at least one delayed read _i[_n]");
            if (color[(size_t)dep] == 0) {
               color[(size_t)dep] = 1;
               st.push_back({dep, 0});
            }
         }
      };
      for (int v : input_nodes) visit(v);
      for (int o : outs) visit(o);
      // sources of delayed reads that are not otherwise needed this sample
      for (size_t i = 0; i < N; ++i)
         if (live[i] && raw[i].kind == FZ_IR_DELAY) visit(raw[i].a);
      for (size_t i = 0; i < N; ++i)
         if (live[i]) visit((int)i);
   }

   // arithmetic types (C++ usual arithmetic conversions): double if any operand is double; wires
   // read from delay lines, inputs and per-stream coefficients are float (state is float, flowz.hpp:1245)
   for (int v : order) {
      Raw& r = raw[(size_t)v];
      switch (r.kind) {
         case FZ_IR_ADD: case FZ_IR_SUB: case FZ_IR_MUL: case FZ_IR_DIV:
            r.f64 = raw[(size_t)r.a].f64 || raw[(size_t)r.b].f64;
            break;
         case FZ_IR_NEG: r.f64 = raw[(size_t)r.a].f64; break;
         case FZ_IR_ABSLT: r.f64 = raw[(size_t)r.a].f64 || raw[(size_t)r.b].f64; break;
         case FZ_IR_SELECT: r.f64 = raw[(size_t)r.b].f64 || raw[(size_t)r.n].f64; break;
         case FZ_IR_CONST: break;
         case FZ_IR_WIDEN: r.f64 = true; break;
         case FZ_IR_NARROW: r.f64 = false; break;
         case FZ_IR_DELAY:
         case FZ_IR_INPUT:
            if (!opt.typed) r.f64 = false;               // typed: set when the node was made (the line / wire type)
            break;
         default: r.f64 = false; break;
      }
   }

   // uniform coefficient slots: one per distinct bit pattern
   Graph g;
   std::map<uint64_t, uint32_t> const_slot64;
   auto slot_of64 = [&](double v) {
      auto it = const_slot64.find(bits_of64(v));
      if (it != const_slot64.end()) return it->second;
      uint32_t s = (uint32_t)g.consts64.size();
      g.consts64.push_back(v);
      const_slot64[bits_of64(v)] = s;
      return s;
   };
   std::map<uint32_t, uint32_t> const_slot;
   auto slot_of = [&](float v, uint32_t uid1) {
      if (uid1) {                                   // run-time uniform: one private slot per id
         auto it = g.uniform_slot.find(uid1 - 1);
         if (it != g.uniform_slot.end()) return it->second;
         uint32_t s = (uint32_t)g.consts.size();
         g.consts.push_back(v);
         g.uniform_slot[uid1 - 1] = s;
         return s;
      }
      auto it = const_slot.find(bits_of(v));
      if (it != const_slot.end()) return it->second;
      uint32_t s = (uint32_t)g.consts.size();
      g.consts.push_back(v);
      const_slot[bits_of(v)] = s;
      return s;
   };

   // common-subexpression merging to a fixpoint (delayed reads may reference later nodes)
   std::vector<int> rep(N);
   for (size_t i = 0; i < N; ++i) rep[i] = (int)i;
   for (bool changed = true; changed;) {
      changed = false;
      std::map<std::tuple<uint32_t, int, int, int, uint64_t>, int> seen;
      for (int v : order) {
         const Raw& r = raw[(size_t)v];
         std::tuple<uint32_t, int, int, int, uint64_t> key;
         switch (r.kind) {
            case FZ_IR_INPUT: key = {r.kind, -1, -1, -1, r.n}; break;
            case FZ_IR_CONST:
               if (r.f64) key = {r.kind, -2, -1, -1, bits_of64(r.value64)};
               else key = {r.kind, r.n ? (int)r.n : -1, -1, -1, r.n ? 0u : bits_of(r.value)};
               break;
            case FZ_IR_PARAM: case FZ_IR_MOD: key = {r.kind, -1, -1, -1, r.n}; break;
            case FZ_IR_DELAY: key = {r.kind, rep[(size_t)r.a], -1, -1, r.n}; break;
            case FZ_IR_NEG: case FZ_IR_WIDEN: case FZ_IR_NARROW: key = {r.kind, rep[(size_t)r.a], -1, -1, 0}; break;
            case FZ_IR_SELECT: key = {r.kind, rep[(size_t)r.a], rep[(size_t)r.b], rep[(size_t)r.n], 0}; break;
            default: key = {r.kind, rep[(size_t)r.a], rep[(size_t)r.b], -1, 0}; break;
         }
         auto it = seen.find(key);
         int nv = v;
         if (it == seen.end()) seen[key] = v; else nv = it->second;
         if (rep[(size_t)v] != nv) { rep[(size_t)v] = nv; changed = true; }
      }
   }

   // renumber representatives in evaluation order
   std::vector<int> newid(N, -1);
   for (int v : order) {
      if (rep[(size_t)v] != v) continue;
      newid[(size_t)v] = (int)g.nodes.size();
      g.nodes.push_back(Node{});
   }
   auto nid = [&](int rawid) { return (uint32_t)newid[(size_t)rep[(size_t)rawid]]; };
   for (int v : order) {
      if (rep[(size_t)v] != v) continue;
      const Raw& r = raw[(size_t)v];
      Node& n = g.nodes[(size_t)newid[(size_t)v]];
      n.kind = r.kind;
      n.f64 = r.f64;
      switch (r.kind) {
         case FZ_IR_INPUT: n.a = r.n; break;
         case FZ_IR_CONST:
            if (r.f64) { n.a = slot_of64(r.value64); n.value64 = r.value64; n.value = r.value; }
            else { n.a = slot_of(r.value, r.n); n.value = r.value; }
            break;
         case FZ_IR_PARAM: n.a = r.n; g.n_param = std::max(g.n_param, r.n + 1); break;
         case FZ_IR_MOD: n.a = r.n; g.n_mod = std::max(g.n_mod, r.n + 1); break;
         case FZ_IR_DELAY: n.a = nid(r.a); n.b = r.n; break;
         case FZ_IR_NEG: case FZ_IR_WIDEN: case FZ_IR_NARROW: n.a = nid(r.a); ++g.n_ops; break;
         case FZ_IR_SELECT: n.a = nid(r.a); n.b = nid(r.b); n.c = nid((int)r.n); ++g.n_ops; break;
         default: n.a = nid(r.a); n.b = nid(r.b); ++g.n_ops; break;
      }
   }
   g.n_in = n_in;
   g.typed = opt.typed;
   g.ref_divergent = el.promise_inputs;
   g.in_dtype = in_dtype;
   g.n_out_wires = (uint32_t)out_wires.size();
   for (size_t k = 0; k < outs.size(); ++k) {
      const uint32_t id = nid(outs[k]);
      if (opt.typed && g.nodes[id].f64) {   // a double (part) leaves un-narrowed: two slots (low word, high word)
         const uint8_t base = out_part[k] == 0 ? 3 : out_part[k] == 1 ? 5 : 7;   // real wire / re / im of a complex<double>
         g.outputs.push_back(id);
         g.out_part.push_back(base);
         g.outputs.push_back(id);
         g.out_part.push_back((uint8_t)(base + 1));
      } else {
         g.outputs.push_back(id);
         g.out_part.push_back(out_part[k]);
      }
   }
   g.n_out = (uint32_t)g.outputs.size();

   // delay lines: one per delayed wire, depth = deepest reader
   std::map<uint32_t, uint32_t> depth;
   for (const Node& n : g.nodes)
      if (n.kind == FZ_IR_DELAY) depth[n.a] = std::max(depth[n.a], n.b);
   g.line_of_node.assign(g.nodes.size(), -1);
   uint32_t row = 0, lds = 0;
   // which delayed nodes are parts of complex wires (typed programs; informational)
   std::map<uint32_t, uint8_t> part_of;
   if (opt.typed)
      for (auto& kv : el.imag_of) {
         if (newid[(size_t)rep[(size_t)kv.first]] >= 0 && !part_of.count(nid(kv.first))) part_of[nid(kv.first)] = 1;
         if (newid[(size_t)rep[(size_t)kv.second]] >= 0 && !part_of.count(nid(kv.second))) part_of[nid(kv.second)] = 2;
      }
   // double lines first (two float rows per slot: their rows start at even row numbers, so a row of n_streams doubles
   // is 8-byte aligned whatever n_streams is), then the float lines; inside each group in node order
   for (int pass = 0; pass < 2; ++pass)
   for (auto& kv : depth) {
      const bool f64 = opt.typed && g.nodes[kv.first].f64;
      if (f64 != (pass == 0)) continue;
      Line l{};
      l.src = kv.first;
      l.depth = kv.second;
      l.row0 = row;
      l.f64 = f64;
      l.part = part_of.count(kv.first) ? part_of[kv.first] : 0;
      l.far = l.depth > kLdsMaxDepth;
      l.in_lds = l.depth > kRegMaxDepth && !l.far;
      if (f64 && l.far)
         fail(FZ_E_UNSUPPORTED, "a double delay line deeper than " + std::to_string(kLdsMaxDepth) + " samples (the rings in HBM hold floats)");
      if (l.in_lds) {
         // ring slots: the next power of two (the ring index is one scalar AND).  Exact sizes -- more streams resident per CU, the
         // index a wave-uniform modulo -- were measured in round 4 and do not pay: 0.59-0.62 of peak against 0.60-0.71 for the two
         // combs of 40 and 23 samples at 1 M streams (profiles/r04/sweep_next_rows.txt)
         uint32_t sz = 1;
         while (sz < l.depth) sz <<= 1;
         l.lds_slot0 = lds;
         l.lds_size = sz;
         lds += f64 ? 2 * sz : sz;                 // a double ring: the low words in [slot0, slot0 + sz), the high words behind them
      }
      row += l.depth * (f64 ? 2u : 1u);
      g.max_delay = std::max(g.max_delay, l.depth);
      g.line_of_node[l.src] = (int)g.lines.size();
      g.lines.push_back(l);
   }
   // far lines: classify their readers, add one phase row per line
   for (size_t li = 0; li < g.lines.size(); ++li) {
      Line& l = g.lines[li];
      if (!l.far) continue;
      g.far_lines.push_back((uint32_t)li);
      l.phase_row = row++;
      for (const Node& n : g.nodes) {
         if (n.kind != FZ_IR_DELAY || n.a != l.src) continue;
         if (n.b <= kRegMaxDepth) l.shadow = std::max(l.shadow, n.b);
         else {
            bool seen = false;
            for (const FarRead& fr : g.far_reads) seen = seen || (fr.line == li && fr.n == n.b);
            if (!seen) g.far_reads.push_back(FarRead{(uint32_t)li, n.b});
            // the ring read of step t is prefetched a chunk ahead: the younger the read, the shorter the chunk
            g.far_min_read = g.far_min_read ? std::min(g.far_min_read, n.b) : n.b;
         }
      }
   }
   g.n_state = row;
   g.n_lds_slots = lds;
   g.split = find_stage_split(g);
   g.wave_splits.assign(5, {});
   for (uint32_t W = 2; W <= 4; ++W) g.wave_splits[W] = find_wave_roles(g, W);
   if (g.split.ok && g.n_in == 1 && g.n_out == 1 && !g.typed && !g.n_lds_slots && g.far_lines.empty()) {
      Graph whole = g;                                   // one part: the compute wave next to an I/O wave (FZ_VF_IO_WAVE)
      whole.wave_splits.clear();
      if (whole.split.atoms() > 9) whole.split = find_stage_split(whole, false, 0, 9);   // (a hand-off spans at most 8 samples of lag)
      if (whole.split.ok) g.wave_splits[1] = {whole};
   }
   return g;
}

}  // namespace fz
