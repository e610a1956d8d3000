// Stage packing analysis: cut a 1-in/1-out graph into K ISOMORPHIC segments in series,
//     in = c_0 -> [S_0] -> c_1 -> [S_1] -> ... -> [S_{K-1}] -> c_K = out          (K even, 2..8)
//
// Why: with one stream per lane (few streams: one wave per SIMD is all there is) the kernel is
// bound by the dependent FP32 chain of the whole graph.  Re-timed so that segment j runs at time
// t-j, all K segments are independent inside one step; and because they are isomorphic, segments
// i and i+K/2 share ONE v_pk_mul_f32 / v_pk_add_f32 per node (with a packed coefficient pair).
// c_0 may also be an internal wire: everything it depends on is then a scalar PREFIX that runs
// un-packed together with segment 0 (an oscillator in front of a cascade, an odd first stage).
// Likewise c_K need not be the output: what the output makes of it (an output gain, a smoothing
// one-pole, the odd last stage) is a scalar SUFFIX that runs un-packed with the last segment.
// A 6-stage biquad cascade becomes 3 independent packed instruction streams with a dependent
// chain of 5 instead of one scalar chain of 30: half the instructions, six times the ILP.
// Same arithmetic, same association order, same roundings per node: only the schedule is skewed
// (segment j+1 consumes the value segment j produced one step earlier), which is a re-timing of
// the reference's left-then-right evaluation inside one call (sequence, flowz.hpp:960-1001).
#include <algorithm>
#include <cmath>
#include <map>
#include <set>

#include "fz_internal.hpp"

namespace fz {

namespace {

bool is_arith(uint32_t k) { return k == FZ_IR_ADD || k == FZ_IR_SUB || k == FZ_IR_MUL || k == FZ_IR_DIV || k == FZ_IR_NEG; }

using Tuple = std::vector<uint32_t>;

struct Matcher {
   const Graph& g;
   const std::vector<uint32_t>& cuts;      // c_0 .. c_K
   const std::vector<int>& seg_of;         // arithmetic node -> segment index
   uint32_t K;
   std::map<Tuple, int> id;                // tuple -> index (or -1 while being matched)
   std::vector<Tuple> tuples;
   std::vector<std::set<uint32_t>> used;   // per segment: arithmetic nodes already matched

   bool match(const Tuple& t)
   {
      if (id.count(t)) return true;
      const Node& n0 = g.nodes[t[0]];
      if (t[0] == cuts[0]) {                               // the chain's input wire (graph input or prefix output)
         for (uint32_t j = 1; j < K; ++j)
            if (t[j] != cuts[j]) return false;           // a segment's input wire is the previous cut
         add(t);
         return true;
      }
      if (n0.kind == FZ_IR_INPUT) return false;
      for (uint32_t j = 1; j < K; ++j) {
         const Node& nj = g.nodes[t[j]];
         if (nj.kind == FZ_IR_INPUT || nj.kind != n0.kind) return false;
      }
      if (is_arith(n0.kind)) {
         for (uint32_t j = 0; j < K; ++j) {
            if (seg_of[t[j]] != (int)j) return false;     // node j lives in segment j
            if (!used[j].insert(t[j]).second) return false;   // bijection
         }
      }
      id[t] = -1;                                          // provisional: breaks feedback cycles
      auto operand = [&](bool second) {
         Tuple o(K);
         for (uint32_t j = 0; j < K; ++j) o[j] = second ? g.nodes[t[j]].b : g.nodes[t[j]].a;
         return o;
      };
      bool ok = true;
      switch (n0.kind) {
         case FZ_IR_CONST:
         case FZ_IR_PARAM: break;                          // values may differ: packed coefficient pair
         case FZ_IR_DELAY:
            for (uint32_t j = 1; j < K; ++j) ok = ok && g.nodes[t[j]].b == n0.b;
            ok = ok && match(operand(false));
            break;
         case FZ_IR_NEG: ok = match(operand(false)); break;
         default: ok = match(operand(false)) && match(operand(true)); break;
      }
      if (!ok) return false;
      add(t);
      return true;
   }

   void add(const Tuple& t)
   {
      id[t] = (int)tuples.size();
      tuples.push_back(t);
   }
};

}  // namespace

// plain: the chain runs from the graph input to the graph output (no scalar prefix / suffix); divisor: only segment counts
// that are multiples of it (both for the wave split, which cuts the chain itself into parts)
StageSplit find_stage_split(const Graph& g, bool plain, uint32_t divisor, uint32_t max_atoms, bool force_atoms)
{
   StageSplit none;
   if (g.n_in != 1 || g.n_out != 1 || g.n_lds_slots != 0 || !g.far_lines.empty() || g.n_ops < 2) return none;
   for (const Node& n : g.nodes)
      // packed halves are float32 pairs; a modulator has ONE value per time step; comparisons / selections have no packed instruction to share
      if (n.f64 || n.kind == FZ_IR_MOD || n.kind >= FZ_IR_ABSLT) return none;
   const uint32_t N = (uint32_t)g.nodes.size();
   uint32_t in = N;
   for (uint32_t i = 0; i < N; ++i)
      if (g.nodes[i].kind == FZ_IR_INPUT) in = i;
   const uint32_t gout = g.outputs[0];
   if (in == N || !is_arith(g.nodes[gout].kind)) return none;

   // backward closure (operands and delay lines) and its operation count, for every arithmetic node
   std::vector<std::vector<char>> closure(N);
   std::vector<uint32_t> cnt(N, 0);
   for (uint32_t c = 0; c < N; ++c) {
      if (!is_arith(g.nodes[c].kind)) continue;
      std::vector<char>& in_c = closure[c];
      in_c.assign(N, 0);
      std::vector<uint32_t> work{c};
      while (!work.empty()) {
         uint32_t v = work.back();
         work.pop_back();
         if (in_c[v]) continue;
         in_c[v] = 1;
         const Node& n = g.nodes[v];
         if (is_arith(n.kind)) {
            ++cnt[c];
            work.push_back(n.a);
            if (n.kind != FZ_IR_NEG) work.push_back(n.b);
         } else if (n.kind == FZ_IR_DELAY) work.push_back(n.a);
      }
   }
   if (cnt[gout] != g.n_ops) return none;

   // chain end candidates: the output itself (no suffix), then every arithmetic wire e such that the
   // rest of the graph (closure(output) minus closure(e)) is a scalar SUFFIX: it sees the chain only
   // through e (now or delayed), constants and itself.  Later wires first (the shortest suffix).
   std::vector<uint32_t> ends{gout};
   for (uint32_t e = N; e-- > 0 && !plain;) {
      if (!is_arith(g.nodes[e].kind) || e == gout || !closure[gout][e] || (g.n_ops - cnt[e]) * 2 > g.n_ops) continue;
      bool ok = true;
      for (uint32_t v = 0; v < N && ok; ++v) {
         if (!closure[gout][v] || closure[e][v]) continue;           // v is a suffix node
         const Node& n = g.nodes[v];
         auto fine = [&](uint32_t o) {                                // operand of a suffix node
            if (!closure[e][o]) return true;                          // another suffix node
            const uint32_t ok_kind = g.nodes[o].kind;
            return o == e || ok_kind == FZ_IR_CONST || ok_kind == FZ_IR_PARAM ||
                   (ok_kind == FZ_IR_DELAY && g.nodes[o].a == e);         // a delayed read of e shared with the chain
         };
         if (n.kind == FZ_IR_INPUT) ok = false;
         else if (n.kind == FZ_IR_DELAY) ok = !closure[e][n.a] || n.a == e;
         else if (is_arith(n.kind)) ok = fine(n.a) && (n.kind == FZ_IR_NEG || fine(n.b));
      }
      if (ok) ends.push_back(e);
   }
   // most segments first (the shortest dependent chains), then the shortest suffix, then the shortest prefix
   for (uint32_t K = 8; K >= 2; K -= 2)
   for (uint32_t out : ends) {
   if (divisor && K % divisor) continue;
   const uint32_t chain_ops = cnt[out];
   // chain input candidates: the graph input itself (no prefix), then every arithmetic wire p whose
   // closure is a scalar PREFIX evaluated at the time of segment 0 (e.g. an oscillator in front of a cascade)
   std::vector<uint32_t> starts{in};
   for (uint32_t c = 0; c < N && !plain; ++c)
      if (is_arith(g.nodes[c].kind) && c != out && closure[out][c] && cnt[c] * 2 <= chain_ops) starts.push_back(c);
   for (uint32_t p0 : starts) {
      const uint32_t base = p0 == in ? 0u : cnt[p0];
      if (chain_ops <= base || (chain_ops - base) % K) continue;
      const uint32_t unit = (chain_ops - base) / K;
      // cut wires: nested closures with j * unit operations
      std::vector<uint32_t> cuts(K + 1, N);
      cuts[0] = p0;
      cuts[K] = out;
      bool found = true;
      for (uint32_t j = K - 1; j >= 1 && found; --j) {
         found = false;
         // every node of a feedback loop has the same closure; the wire that leaves the segment is
         // the topologically last one, so scan from the back
         for (uint32_t c = N; c-- > 0 && !found;)
            if (is_arith(g.nodes[c].kind) && cnt[c] == base + j * unit && closure[cuts[j + 1]][c] && c != cuts[j + 1] && (p0 == in || closure[c][p0])) {
               cuts[j] = c;
               found = true;
            }
      }
      if (!found) continue;
      // segment of every arithmetic node
      std::vector<int> seg_of(N, -1);                      // -2: prefix
      for (uint32_t v = 0; v < N; ++v) {
         if (!is_arith(g.nodes[v].kind) || !closure[out][v]) continue;   // (suffix nodes stay -1)
         if (p0 != in && closure[p0][v]) { seg_of[v] = -2; continue; }
         for (uint32_t j = 0; j < K; ++j)
            if (closure[cuts[j + 1]][v]) { seg_of[v] = (int)j; break; }
      }
      // a segment may touch earlier ones only through its input cut wire (now or delayed) and leaves
      bool clean = true;
      auto operand_ok = [&](uint32_t o, int j) {
         const Node& n = g.nodes[o];
         if (n.kind == FZ_IR_CONST || n.kind == FZ_IR_PARAM) return true;
         if (o == cuts[(size_t)j]) return true;            // the segment's input wire
         if (n.kind == FZ_IR_INPUT) return false;
         if (n.kind == FZ_IR_DELAY) return n.a == cuts[(size_t)j] || (is_arith(g.nodes[n.a].kind) && seg_of[n.a] == j);
         return seg_of[o] == j;
      };
      for (uint32_t v = 0; v < N && clean; ++v) {
         if (!is_arith(g.nodes[v].kind) || seg_of[v] < 0) continue;
         const Node& n = g.nodes[v];
         clean = operand_ok(n.a, seg_of[v]) && (n.kind == FZ_IR_NEG || operand_ok(n.b, seg_of[v]));
      }
      if (!clean) continue;

      Matcher m{g, cuts, seg_of, K, {}, {}, std::vector<std::set<uint32_t>>(K)};
      Tuple root(K);
      for (uint32_t j = 0; j < K; ++j) root[j] = cuts[j + 1];
      if (!m.match(root)) continue;
      if (m.used[0].size() != unit) continue;              // every operation of S_0 has its partners

      StageSplit s;
      s.ok = true;
      s.K = K;
      s.cuts = cuts;
      // evaluation order: leaves and delayed reads first, then by the S_0 node's topological position
      s.tuples = m.tuples;
      std::stable_sort(s.tuples.begin(), s.tuples.end(), [&](const Tuple& x, const Tuple& y) {
         const bool ax = is_arith(g.nodes[x[0]].kind), ay = is_arith(g.nodes[y[0]].kind);
         if (ax != ay) return !ax;
         return ax ? x[0] < y[0] : false;
      });
      // ---- sub-atoms: internal single-wire cuts of the segment (checked on segment 0; the others are isomorphic to it) ----
      // A wire v of the segment is a cut when the rest of the segment sees what v depends on only through v itself, now or
      // delayed, and does not look at the segment's input wire any more.  Nested cuts compose (each later atom reads the
      // earlier ones only through the cut in front of it).  Wanted: atoms of 4-5 operations (a biquad: feed-forward sum |
      // recursion), at most max_atoms skewed units in all.
      {
         auto in0 = [&](uint32_t u) { return is_arith(g.nodes[u].kind) && seg_of[u] == 0; };
         auto ops0 = [&](uint32_t v) {
            uint32_t n = 0;
            for (uint32_t u = 0; u < N; ++u) n += in0(u) && closure[v][u];
            return n;
         };
         auto valid_cut = [&](uint32_t v) {
            for (uint32_t n = 0; n < N; ++n) {
               if (!in0(n) || closure[v][n]) continue;        // n: a node of the segment behind the cut
               const Node& nd = g.nodes[n];
               auto fine = [&](uint32_t o) {
                  const Node& on = g.nodes[o];
                  if (on.kind == FZ_IR_CONST || on.kind == FZ_IR_PARAM) return true;
                  if (o == cuts[0] || on.kind == FZ_IR_INPUT) return false;            // the segment's input: only the first atom reads it
                  if (on.kind == FZ_IR_DELAY) {
                     if (on.a == cuts[0]) return false;
                     if (in0(on.a) && closure[v][on.a]) return on.a == v;              // a delayed read of the cut wire itself
                     return true;
                  }
                  if (in0(o) && closure[v][o]) return o == v;
                  return true;
               };
               if (!fine(nd.a) || (nd.kind != FZ_IR_NEG && !fine(nd.b))) return false;
            }
            return true;
         };
         uint32_t m = unit >= 8 ? (unit >= 18 ? 3u : 2u) : 1u;
         // three or more packed pairs already give a wave enough independent work; there the atoms would only lengthen the masked
         // ends of every block (one slow step per skewed unit at either end: ~0.3 us each, 2 % of a 4096-sample block of config 2)
         if (K >= 6 && !force_atoms) m = 1;
         while (m > 1 && K * m > max_atoms) --m;
         std::vector<std::pair<uint32_t, uint32_t>> cand;    // (operations of the segment up to and including v, v)
         if (m > 1)
            for (uint32_t v = 0; v < N; ++v)
               if (in0(v) && v != cuts[1] && valid_cut(v)) cand.push_back({ops0(v), v});
         std::sort(cand.begin(), cand.end());
         std::vector<uint32_t> chosen;
         for (; m > 1 && chosen.empty(); --m) {
            uint32_t prev = N;
            for (uint32_t a = 1; a < m; ++a) {
               const double want = (double)unit * a / m;
               uint32_t best = N;
               double bd = 1e9;
               for (auto& c : cand) {
                  if (c.first == 0 || c.first >= unit) continue;
                  if (prev != N && (!closure[c.second][prev] || c.second == prev)) continue;   // nested behind the previous cut
                  const double d = std::abs((double)c.first - want);
                  if (d < bd) { bd = d; best = c.second; }
               }
               if (best == N || bd > (double)unit / (2.0 * m)) { chosen.clear(); break; }
               chosen.push_back(best);
               prev = best;
            }
            if (!chosen.empty()) { ++m; break; }              // (undo the loop's decrement: this m worked)
         }
         s.m = chosen.empty() ? 1u : (uint32_t)chosen.size() + 1u;
         // the atom of every tuple: how many of the chosen cuts lie strictly in front of its segment-0 node
         s.sub.assign(s.tuples.size(), 0);
         for (size_t k = 0; k < s.tuples.size(); ++k) {
            const uint32_t u = s.tuples[k][0];
            if (!is_arith(g.nodes[u].kind) || u == cuts[0]) continue;
            uint32_t a = 0;
            for (uint32_t c : chosen) a += !closure[c][u];
            s.sub[k] = a;
         }
         s.icuts.clear();
         for (uint32_t c : chosen)
            for (auto& t : s.tuples)
               if (t[0] == c) s.icuts.push_back(t);
         if (s.icuts.size() != chosen.size()) { s.m = 1; s.icuts.clear(); std::fill(s.sub.begin(), s.sub.end(), 0u); }
      }
      // delay lines: one copy per (source tuple, atom that reads it) -- an atom sees a wire's past in its own time frame
      std::map<Tuple, size_t> tix;
      for (size_t k = 0; k < s.tuples.size(); ++k) tix[s.tuples[k]] = k;
      std::map<std::pair<Tuple, uint32_t>, uint32_t> line_depth;   // (tuple of line sources, frame) -> depth
      for (size_t k = 0; k < s.tuples.size(); ++k) {
         const auto& t = s.tuples[k];
         if (!is_arith(g.nodes[t[0]].kind) || t[0] == cuts[0]) continue;
         for (int side = 0; side < (g.nodes[t[0]].kind == FZ_IR_NEG ? 1 : 2); ++side) {
            Tuple o(K);
            for (uint32_t j = 0; j < K; ++j) o[j] = side ? g.nodes[t[j]].b : g.nodes[t[j]].a;
            if (g.nodes[o[0]].kind != FZ_IR_DELAY) continue;
            Tuple src(K);
            for (uint32_t j = 0; j < K; ++j) src[j] = g.nodes[o[j]].a;
            auto key = std::make_pair(src, s.sub[k]);
            line_depth[key] = std::max(line_depth[key], g.nodes[o[0]].b);
         }
      }
      std::set<uint32_t> covered;
      std::map<Tuple, uint32_t> last_frame;                // per source tuple: its copy in the latest atom's frame
      for (auto& kv : line_depth) last_frame[kv.first.first] = std::max(last_frame[kv.first.first], kv.first.second);
      for (auto& kv : line_depth) {
         PackedLine pl;
         pl.srcs = kv.first.first;
         pl.frame = kv.first.second;
         pl.depth = kv.second;
         for (uint32_t j = 0; j < K && s.ok; ++j) {
            const int l = g.line_of_node[pl.srcs[j]];
            if (l < 0) { s.ok = false; break; }
            // one copy per wire keeps the line's FULL depth (a scalar prefix / suffix may read further back than the chain does,
            // and the state buffer wants every row): the copy of the latest atom, which is where those parts look
            if (pl.frame == last_frame[pl.srcs]) pl.depth = std::max(pl.depth, g.lines[(size_t)l].depth);
            covered.insert(pl.srcs[j]);
         }
         if (pl.depth > kRegMaxDepth) s.ok = false;
         if (!s.ok) break;
         s.lines.push_back(pl);
      }
      if (!s.ok) continue;
      // prefix: every node the chain input depends on, in evaluation order; its private delay lines
      if (p0 != in)
         for (uint32_t v = 0; v < N; ++v)
            if (closure[p0][v]) s.prefix.push_back(v);
      for (size_t l = 0; l < g.lines.size(); ++l) {
         const uint32_t src = g.lines[l].src;
         if (covered.count(src)) continue;
         const bool in_prefix = p0 != in && (src == in || closure[p0][src]);
         const bool in_suffix = out != gout && (src == out || (closure[gout][src] && !closure[out][src]));
         if (in_prefix) s.prefix_lines.push_back((uint32_t)l);
         else if (in_suffix) s.suffix_lines.push_back((uint32_t)l);
         else { s.ok = false; break; }
      }
      if (!s.ok) continue;
      // suffix: what the output makes of the chain's end wire, in evaluation order
      if (out != gout)
         for (uint32_t v = 0; v < N; ++v)
            if (closure[gout][v] && !closure[out][v]) s.suffix.push_back(v);
      return s;
   }
   }
   return none;
}

// ---- wave split ---------------------------------------------------------------------------------------------------
// Fewer streams than the chip has lanes: SIMDs idle while every wave carries the whole serial graph.  A graph that is a
// series of K isomorphic segments (no scalar prefix or suffix) is cut into W parts of K / W segments at the wires
// cuts[K/W], cuts[2K/W], ...:
//    in -> [ part 0 ] -> m_1 -> [ part 1 ] -> m_2 -> ... -> [ part W-1 ] -> out
// W waves of a workgroup evaluate the parts for the same 64 streams; the cut wires travel through LDS, every part one
// chunk of samples behind the one before (fz_block_kernel.hip.inc, FZ_VF_WAVE_SPLIT).  Each part is a graph of its own here
// -- node ids renumbered, constant slots and state rows the parent's -- so that the ordinary generators (and the stage
// packing of each part: a 3-biquad part is one packed pair and a scalar stage) apply to it unchanged.  Delay lines of a
// cut wire are kept by both neighbours (the earlier part reads them as its own output's past, the later one as its
// input's past): the same values in both.
static std::vector<char> closure_of(const Graph& g, uint32_t root)
{
   std::vector<char> in(g.nodes.size(), 0);
   std::vector<uint32_t> work{root};
   while (!work.empty()) {
      const uint32_t v = work.back();
      work.pop_back();
      if (in[v]) continue;
      in[v] = 1;
      const Node& n = g.nodes[v];
      if (is_arith(n.kind)) {
         work.push_back(n.a);
         if (n.kind != FZ_IR_NEG) work.push_back(n.b);
      } else if (n.kind == FZ_IR_DELAY) work.push_back(n.a);
   }
   return in;
}

// part between the wires cin (its input; N: the graph input) and cout (its output): the nodes cout depends on that cin
// does not, constants, and the delayed reads of cin
static bool extract_part(const Graph& g, const std::vector<char>* before, const std::vector<char>& upto, uint32_t cin, uint32_t cout, Graph& r)
{
   const uint32_t N = (uint32_t)g.nodes.size();
   const bool first = cin == N;
   std::vector<int> nid(N, -1);
   r = Graph();
   r.n_in = r.n_out = r.n_out_wires = 1;
   r.consts = g.consts;                                   // same slots: the kernarg image is the parent's
   r.uniform_slot = g.uniform_slot;
   r.in_dtype = {0};
   auto add = [&](uint32_t v, const Node& n) {
      nid[v] = (int)r.nodes.size();
      r.nodes.push_back(n);
   };
   if (!first) {
      Node in{};
      in.kind = FZ_IR_INPUT;
      add(cin, in);                                       // the cut wire is this part's input
   }
   auto mine = [&](uint32_t v) {
      const Node& n = g.nodes[v];
      if (!first && v == cin) return false;               // (added above)
      if (!upto[v]) return false;
      if (n.kind == FZ_IR_CONST || n.kind == FZ_IR_PARAM) return true;
      if (n.kind == FZ_IR_DELAY && !first && n.a == cin) return true;
      return first || !(*before)[v];
   };
   for (uint32_t v = 0; v < N; ++v) {
      if (!mine(v)) continue;
      Node n = g.nodes[v];
      auto op = [&](uint32_t o) -> int { return o < N ? nid[o] : -1; };
      switch (n.kind) {
         case FZ_IR_INPUT: if (!first) return false; break;
         case FZ_IR_CONST: break;
         case FZ_IR_PARAM: r.n_param = g.n_param; break;   // per-stream coefficient: the parent's row
         case FZ_IR_DELAY: break;                          // source patched below (it may come later in the order)
         case FZ_IR_NEG:
            if (op(n.a) < 0) return false;
            n.a = (uint32_t)op(n.a);
            ++r.n_ops;
            break;
         case FZ_IR_ADD: case FZ_IR_SUB: case FZ_IR_MUL: case FZ_IR_DIV:
            if (op(n.a) < 0 || op(n.b) < 0) return false;  // an operand from an earlier part that is not the cut wire
            n.a = (uint32_t)op(n.a);
            n.b = (uint32_t)op(n.b);
            ++r.n_ops;
            break;
         default: return false;                            // per-stream coefficients, modulators, typed nodes: not split
      }
      add(v, n);
   }
   for (uint32_t v = 0; v < N; ++v) {                       // delayed reads: the source is a node of this part (its input included)
      if (nid[v] < 0 || g.nodes[v].kind != FZ_IR_DELAY || (!first && v == cin)) continue;
      const uint32_t src = g.nodes[v].a;
      if (src >= N || nid[src] < 0) return false;
      r.nodes[(size_t)nid[v]].a = (uint32_t)nid[src];
   }
   if (nid[cout] < 0) return false;
   r.outputs = {(uint32_t)nid[cout]};
   r.out_part = {0};
   // lines: the parent's, for the sources this part reads through a delay; rows and depths are the parent's
   r.line_of_node.assign(r.nodes.size(), -1);
   for (const Line& L : g.lines) {
      if (nid[L.src] < 0) continue;
      bool read_here = false;
      for (uint32_t v = 0; v < N && !read_here; ++v)
         read_here = nid[v] >= 0 && !(!first && v == cin) && g.nodes[v].kind == FZ_IR_DELAY && g.nodes[v].a == L.src;
      if (!read_here) continue;
      if (L.in_lds || L.far || L.f64) return false;
      Line l = L;
      l.src = (uint32_t)nid[L.src];
      r.line_of_node[l.src] = (int)r.lines.size();
      r.lines.push_back(l);
      r.max_delay = std::max(r.max_delay, l.depth);
   }
   r.n_state = g.n_state;
   r.split = find_stage_split(r, false, 0, 9);            // (a hand-off spans at most 8 samples of lag: at most 9 skewed units per part)
   return r.split.ok;
}

std::vector<Graph> find_wave_roles(const Graph& g, uint32_t W)
{
   if (W < 2 || !g.split.ok || g.typed || g.n_mod || g.n_lds_slots || !g.far_lines.empty() || g.n_in != 1 || g.n_out != 1) return {};
   // the chain itself, input to output, in a number of segments the parts divide (the graph's own stage split may prefer
   // more segments behind a scalar prefix: a 12-biquad cascade is 8 segments of 13 operations after 4, but also 6 x 2 biquads);
   // failing that the graph's own split, scalar prefix and suffix included: they go with the first and the last part
   // (an oscillator in front of a cascade: part 0 is the oscillator and the first stages)
   StageSplit sp = find_stage_split(g, true, W);
   if (!sp.ok || sp.K < W || sp.K % W) sp = g.split;
   if (!sp.ok || sp.K < W || sp.K % W) return {};
   const uint32_t N = (uint32_t)g.nodes.size(), m = sp.K / W;
   std::vector<std::vector<char>> clo;                     // closure of the wire that ends part k
   std::vector<uint32_t> ends(W);                          // the wire behind part k; the last part ends at the graph output (suffix included)
   for (uint32_t k = 0; k < W; ++k) ends[k] = k + 1 < W ? sp.cuts[(k + 1) * m] : g.outputs[0];
   for (uint32_t k = 0; k < W; ++k) clo.push_back(closure_of(g, ends[k]));
   std::vector<Graph> roles(W);
   uint32_t ops = 0;
   for (uint32_t k = 0; k < W; ++k) {
      if (!extract_part(g, k ? &clo[k - 1] : nullptr, clo[k], k ? ends[k - 1] : N, ends[k], roles[k])) return {};
      ops += roles[k].n_ops;
   }
   if (ops != g.n_ops) return {};                          // every operation sits in exactly one part
   return roles;
}

}  // namespace fz
