// Stage packing analysis: find a cut wire that splits a 1-in/1-out graph into two ISOMORPHIC
// halves A (inputs -> cut) and B (cut -> output).
//
// Why: with one stream per lane (few streams: one wave per SIMD is all there is) the kernel is
// bound by the dependent scalar FP32 chain.  If the graph is a serial composition B after A of two
// structurally identical halves -- e.g. stages 1-3 and 4-6 of a biquad cascade -- then half A at
// time t and half B at time t-1 are independent, and because they are isomorphic every pair of
// corresponding nodes is ONE v_pk_mul_f32 / v_pk_add_f32 on (A-value, B-value) with a packed
// coefficient pair.  Same arithmetic, same order, same roundings per node; only the schedule is
// skewed by one sample, so the instruction count and the dependent chain both halve.
// This is a re-timing of the reference's per-sample evaluation order (sequence :960-1001 evaluates
// left then right within a call); values are unaffected because B only ever consumes A's output.
#include <algorithm>
#include <functional>
#include <map>
#include <set>

#include "fz_internal.hpp"

namespace fz {

namespace {

bool is_arith(uint32_t k) { return k == FZ_IR_ADD || k == FZ_IR_SUB || k == FZ_IR_MUL || k == FZ_IR_DIV || k == FZ_IR_NEG; }

struct Matcher {
   const Graph& g;
   uint32_t in, cut;
   const std::vector<char>& inA;
   std::map<std::pair<uint32_t, uint32_t>, int> pair_id;   // (a, b) -> index in pairs
   std::vector<std::pair<uint32_t, uint32_t>> pairs;
   std::map<uint32_t, uint32_t> a2b, b2a;                  // bijection on arithmetic nodes

   bool match(uint32_t a, uint32_t b)
   {
      auto key = std::make_pair(a, b);
      if (pair_id.count(key)) return true;
      const Node& na = g.nodes[a];
      const Node& nb = g.nodes[b];
      if (na.kind == FZ_IR_INPUT) {
         if (b != cut) return false;                        // A's input wire <-> B's input wire (the cut)
         pair_id[key] = -1;
         register_pair(key);
         return true;
      }
      if (nb.kind == FZ_IR_INPUT) return false;
      if (na.kind != nb.kind) return false;
      if (is_arith(na.kind)) {
         if (!inA[a] || inA[b]) return false;               // a in half A, b in half B
         auto ia = a2b.find(a);
         auto ib = b2a.find(b);
         if (ia != a2b.end() || ib != b2a.end()) return false;   // (a,b) not paired before, so a clash
         a2b[a] = b;
         b2a[b] = a;
      }
      pair_id[key] = -1;                                    // provisional: breaks feedback cycles
      bool ok = true;
      switch (na.kind) {
         case FZ_IR_CONST: break;                           // values may differ: packed coefficient pair
         case FZ_IR_PARAM: break;
         case FZ_IR_DELAY: ok = na.b == nb.b && match(na.a, nb.a); break;
         case FZ_IR_NEG: ok = match(na.a, nb.a); break;
         default: ok = match(na.a, nb.a) && match(na.b, nb.b); break;
      }
      if (!ok) return false;
      register_pair(key);
      return true;
   }

   void register_pair(const std::pair<uint32_t, uint32_t>& key)
   {
      pair_id[key] = (int)pairs.size();
      pairs.push_back(key);
   }
};

}  // namespace

StageSplit find_stage_split(const Graph& g)
{
   StageSplit none;
   if (g.n_in != 1 || g.n_out != 1 || g.n_lds_slots != 0 || g.n_ops < 2 || (g.n_ops & 1)) return none;
   const uint32_t N = (uint32_t)g.nodes.size();
   uint32_t in = N;
   for (uint32_t i = 0; i < N; ++i)
      if (g.nodes[i].kind == FZ_IR_INPUT) in = i;
   const uint32_t out = g.outputs[0];
   if (in == N || !is_arith(g.nodes[out].kind)) return none;

   for (uint32_t cut = 0; cut < N; ++cut) {
      if (cut == out || !is_arith(g.nodes[cut].kind)) continue;
      // half A = everything the cut wire depends on, through operands and delay lines
      std::vector<char> inA(N, 0);
      std::vector<uint32_t> work{cut};
      uint32_t opsA = 0;
      while (!work.empty()) {
         uint32_t v = work.back();
         work.pop_back();
         if (inA[v]) continue;
         inA[v] = 1;
         const Node& n = g.nodes[v];
         if (is_arith(n.kind)) {
            ++opsA;
            work.push_back(n.a);
            if (n.kind != FZ_IR_NEG) work.push_back(n.b);
         } else if (n.kind == FZ_IR_DELAY) work.push_back(n.a);
      }
      if (opsA * 2 != g.n_ops || inA[out]) continue;
      // half B may touch half A only through the cut wire (now or delayed) and shared leaves
      bool clean = true;
      auto b_operand_ok = [&](uint32_t o) {
         const Node& n = g.nodes[o];
         if (!inA[o]) return n.kind != FZ_IR_INPUT;
         if (o == cut) return true;
         if (n.kind == FZ_IR_CONST || n.kind == FZ_IR_PARAM) return true;
         return n.kind == FZ_IR_DELAY && n.a == cut;
      };
      for (uint32_t v = 0; v < N && clean; ++v) {
         if (inA[v]) continue;
         const Node& n = g.nodes[v];
         if (is_arith(n.kind)) clean = b_operand_ok(n.a) && (n.kind == FZ_IR_NEG || b_operand_ok(n.b));
         else if (n.kind == FZ_IR_DELAY) clean = (!inA[n.a] || n.a == cut);
         else if (n.kind == FZ_IR_INPUT) clean = false;
      }
      if (!clean) continue;
      Matcher m{g, in, cut, inA, {}, {}, {}, {}};
      if (!m.match(cut, out)) continue;
      if (m.a2b.size() != opsA) continue;                    // every operation of A has its partner
      // every delay line must be covered by a packed line (a source pair that is delayed somewhere)
      StageSplit s;
      s.ok = true;
      s.in_node = in;
      s.cut_node = cut;
      s.out_node = out;
      // evaluation order: by the A-side node's topological position; leaves and delays first
      std::vector<std::pair<uint32_t, uint32_t>> ordered = m.pairs;
      std::stable_sort(ordered.begin(), ordered.end(), [&](const auto& x, const auto& y) {
         const bool ax = is_arith(g.nodes[x.first].kind), ay = is_arith(g.nodes[y.first].kind);
         if (ax != ay) return !ax;
         return ax ? x.first < y.first : false;
      });
      s.pairs = ordered;
      std::map<std::pair<uint32_t, uint32_t>, uint32_t> line_depth;
      for (auto& p : s.pairs)
         if (g.nodes[p.first].kind == FZ_IR_DELAY) {
            auto src = std::make_pair(g.nodes[p.first].a, g.nodes[p.second].a);
            line_depth[src] = std::max(line_depth[src], g.nodes[p.first].b);
         }
      std::set<uint32_t> covered;
      for (auto& kv : line_depth) {
         const int la = g.line_of_node[kv.first.first], lb = g.line_of_node[kv.first.second];
         if (la < 0 || lb < 0) { s.ok = false; break; }
         PackedLine pl;
         pl.src_a = kv.first.first;
         pl.src_b = kv.first.second;
         pl.depth = std::max(g.lines[(size_t)la].depth, g.lines[(size_t)lb].depth);
         if (pl.depth > kRegMaxDepth) { s.ok = false; break; }
         s.lines.push_back(pl);
         covered.insert(pl.src_a);
         covered.insert(pl.src_b);
      }
      if (!s.ok) continue;
      for (const Line& l : g.lines)
         if (!covered.count(l.src)) s.ok = false;
      if (!s.ok) continue;
      return s;
   }
   return none;
}

}  // namespace fz
