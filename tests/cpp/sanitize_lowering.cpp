// Host-only robustness harness (built with -fsanitize=address,undefined by tests/test_sanitize.py):
// random flow-graphs in prefix notation on stdin, one per line, go through the expression
// constructors, lower(), the stage-split analysis and the code generator of every variant family.
// No HIP involved: the sources under test are plain C++.
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "fz_internal.hpp"

using namespace fz;

static fz_expr* parse(std::istringstream& in)
{
   std::string op;
   in >> op;
   auto un = [&](auto f) { fz_expr* a = parse(in); fz_expr* r = a ? f(a) : nullptr; fz_expr_release(a); return r; };
   auto bin = [&](auto f) {
      fz_expr* a = parse(in);
      fz_expr* b = parse(in);
      fz_expr* r = (a && b) ? f(a, b) : nullptr;
      fz_expr_release(a);
      fz_expr_release(b);
      return r;
   };
   if (op == "in") { unsigned i; in >> i; return fz_placeholder(i); }
   if (op == "del") { unsigned i, n; in >> i >> n; return fz_delayed(i, n); }
   if (op == "lit") { float v; in >> v; return fz_literal(v); }
   if (op == "lit64") { double v; in >> v; return fz_literal_f64(v); }
   if (op == "litc") { float a, b; in >> a >> b; return fz_literal_c32(a, b); }
   if (op == "param") { unsigned k; in >> k; return fz_stream_param(k); }
   if (op == "uniform") { unsigned k; float v; in >> k >> v; return fz_uniform(k, v); }
   if (op == "neg") return un([](fz_expr* a) { return fz_arith(FZ_OP_NEG, a, nullptr); });
   if (op == "fb") return un([](fz_expr* a) { return fz_feedback(a); });
   if (op == "add") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_ADD, a, b); });
   if (op == "sub") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_SUB, a, b); });
   if (op == "mul") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_MUL, a, b); });
   if (op == "div") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_DIV, a, b); });
   if (op == "not") return un([](fz_expr* a) { return fz_arith(FZ_OP_NOT, a, nullptr); });
   if (op == "lt") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_LT, a, b); });
   if (op == "le") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_LE, a, b); });
   if (op == "gt") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_GT, a, b); });
   if (op == "ge") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_GE, a, b); });
   if (op == "eq") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_EQ, a, b); });
   if (op == "ne") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_NE, a, b); });
   if (op == "and") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_AND, a, b); });
   if (op == "or") return bin([](fz_expr* a, fz_expr* b) { return fz_arith(FZ_OP_OR, a, b); });
   if (op == "chan") return bin(fz_channel);
   if (op == "par") return bin(fz_parallel);
   if (op == "seq") return bin(fz_sequence);
   return nullptr;
}

int main()
{
   std::string line;
   unsigned n = 0, lowered = 0, rejected = 0, packable = 0, split = 0;
   size_t bytes = 0;
   while (std::getline(std::cin, line)) {
      if (line.empty()) continue;
      std::istringstream in(line);
      fz_expr* e = parse(in);
      ++n;
      if (!e) { ++rejected; continue; }
      try {
         Graph g = lower(e);
         ++lowered;
         packable += g.split.ok;
         for (uint32_t P : {1u, 2u, 4u})
            for (uint32_t U : {1u, 8u, 16u}) {
               Variant v;
               v.P = P; v.U = U; v.block = 256; v.flags = 0;
               bytes += gen_config(g, v).size() + gen_body(g, v).size();
               v.flags = FZ_VF_OUT_F64 | FZ_VF_MAX_WG(1);
               bytes += gen_config(g, v).size() + gen_body(g, v).size();
            }
         if (g.split.ok) {
            Variant v;
            v.P = 1; v.U = 16; v.block = 256; v.flags = FZ_VF_STAGE_PACK;
            bytes += full_source(g, v).size();
         }
         for (uint32_t W = 1; W <= 4; ++W)                  // wave splits / the I/O wave: role extraction ran in lower(), now their bodies
            if (g.wave_roles(W)) {
               Variant v;
               v.P = 1; v.U = 16; v.block = 64; v.flags = (W > 1 ? (W - 1) << 10 : 0u) | (W == 1 || (n & 1) ? (uint32_t)FZ_VF_IO_WAVE : 0u);
               bytes += full_source(g, v).size();
               ++split;
            }
         (void)max_input_delays(e);
      } catch (const Error&) {
         ++rejected;
      }
      fz_expr_release(e);
   }
   std::printf("graphs %u lowered %u rejected %u stage-packable %u wave-split bodies %u generated %zu bytes\n", n, lowered, rejected, packable, split, bytes);
   return lowered > 0 ? 0 : 1;
}
