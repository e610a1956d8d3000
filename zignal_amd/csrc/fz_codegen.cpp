// Lowered DAG -> the two generated headers the kernel skeleton includes.
//
// "fz_graph_config.h": sizes and variant knobs as macros.
// "fz_graph_body.h"  : struct fz_graph -- register delay lines / per-stream coefficients as
//                      members, and step(): the whole graph for one sample as straight-line code,
//                      one statement per DAG node in evaluation order (explicit temporaries, so the
//                      compiler cannot re-associate; contraction is disabled on the command line).
#include <algorithm>
#include <cstdio>
#include <map>
#include <set>
#include <sstream>

#include "fz_internal.hpp"

namespace fz {

// fz_block_kernel.hip.inc and the three kernel bodies, embedded at build time (embed.py)
extern const char* const kSkeletonHead;
extern const char* const kSkeletonBody_stream_major;
extern const char* const kSkeletonBody_wave_split;
extern const char* const kSkeletonBody_frames;

// the hand-written text of a variant's kernel: the common head + the ONE body its flags select
const std::string& skeleton_source(uint32_t flags)
{
   static const std::string sm = std::string(kSkeletonHead) + kSkeletonBody_stream_major, ws = std::string(kSkeletonHead) + kSkeletonBody_wave_split,
                            fr = std::string(kSkeletonHead) + kSkeletonBody_frames;
   return (flags & FZ_VF_STREAM_MAJOR) ? sm : ws_parts(flags) ? ws : fr;
}

// one symbol per variant, so that profilers (rocprofv3 --stats) keep the variants apart:
// fz_block_kernel_p<streams/lane>u<unroll>b<block>[s<segments>]f<flags>
std::string kernel_name(const Graph& g, const Variant& v)
{
   std::string n = "fz_block_kernel_p" + std::to_string(v.P) + "u" + std::to_string(v.U) + "b" + std::to_string(v.block);
   if ((v.flags & FZ_VF_STAGE_PACK) && g.split.ok) n += "s" + std::to_string(g.split.K) + (g.split.m > 1 ? "a" + std::to_string(g.split.m) : "");
   if (ws_parts(v.flags)) n += "w" + std::to_string(ws_parts(v.flags)) + (ws_io(v.flags) ? (ws_io_waves(v.flags) == 2 ? "io2" : "io") : "");
   // (the flags the caller can set, then one letter per INTERNAL bit the launch path added: R rows clipped per descriptor (a count that is
   //  not a multiple of the streams per lane), M merging stores (rows off the 64-byte grid), L the lane's four streams as
   //  two pairs 128 apart, S the lane's streams 64 apart)
   constexpr uint32_t internal = FZ_VF_RAGGED | FZ_VF_ST_MERGE | FZ_VF_LANE_PAIRS | FZ_VF_LANE_SINGLES;
   return n + "f" + std::to_string(v.flags & ~internal) + ((v.flags & FZ_VF_RAGGED) ? "R" : "") + ((v.flags & FZ_VF_ST_MERGE) ? "M" : "") +
          ((v.flags & FZ_VF_LANE_PAIRS) ? "L" : "") + ((v.flags & FZ_VF_LANE_SINGLES) ? "S" : "");
}

// The symbol a profiler sees: the variant AND the graph -- two graphs that run the same variant (the 6-biquad cascade and a single
// biquad both take four streams per lane in 1024-lane workgroups) must not share a row of `rocprofv3 --stats` (round 4: the average
// over "fz_block_kernel_p4u1b1024f8912928" mixed a dozen workloads of the bench line).  The tag is the graph's STRUCTURE (no
// coefficient values: graphs that differ in literals only share code objects and symbols).
std::string kernel_symbol(const Graph& g, const Variant& v)
{
   char tag[16];
   std::snprintf(tag, sizeof tag, "_g%08x", g.sym_tag);
   return kernel_name(g, v) + tag;
}

// LDS rings of the frame kernels, VECTORISED IN TIME (round 5).  Round 4 kept a ring as `ring[slot][lane]`: one ds_read and one
// ds_write of a lane's 4 P bytes per line and step, and a read younger than the chunk (the 23-sample comb under 32-row chunks) was
// issued where it was needed -- one LDS round trip per step with one or two waves per SIMD to hide it behind.  Now a lane's slots
// are CONTIGUOUS (`ring[line][lane][slot][stream of the lane]`, rows padded by 16 bytes: the lanes of every 16-byte access group fall on
// distinct banks), the chunk runs in SUB-CHUNKS of G steps (G = the largest power of two <= the youngest ring read, <= the chunk): all
// ring reads of a sub-chunk refer to slots written before it began and are fetched together as 16-byte vectors (4 / P time steps each),
// and its pushes are kept in registers and written as 16-byte vectors when it ends.  A read `d` samples back starts (-d) mod (4 / P)
// slots off the 16-byte grid -- a constant: the aligned vectors around it are read and the step's value is a register of them.
RingPlan ring_plan(const Graph& g, const Variant& v)
{
   RingPlan rp;
   rp.slots = g.n_lds_slots;
   if (!g.n_lds_slots || (v.flags & (FZ_VF_STREAM_MAJOR | FZ_VF_STAGE_PACK)) || ws_parts(v.flags) || v.P > 4 || v.U < 2) return rp;
   const uint32_t TW = 4 / v.P;
   uint32_t min_read = ~0u;
   std::set<std::pair<uint32_t, uint32_t>> reads;
   for (const Node& nd : g.nodes) {
      if (nd.kind != FZ_IR_DELAY) continue;
      const Line& L = g.lines[(size_t)g.line_of_node[nd.a]];
      if (!L.in_lds) continue;
      if (L.f64) return rp;                                  // (double rings: two word planes, read in place)
      min_read = std::min(min_read, nd.b);
      reads.insert({(uint32_t)g.line_of_node[nd.a], nd.b});
   }
   uint32_t n_lines = 0;
   for (const Line& L : g.lines)
      if (L.in_lds) {
         if (L.f64 || (L.lds_size & (L.lds_size - 1)) || L.lds_size % TW) return rp;
         ++n_lines;
      }
   if (min_read == ~0u) min_read = v.U;                      // (lines nobody reads from LDS: pushes only)
   uint32_t G = 1;
   while (G * 2 <= min_read && G * 2 <= v.U) G *= 2;
   while (G > 1 && v.U % G) G /= 2;
   auto regs = [&](uint32_t gg) {
      uint32_t r = n_lines * gg * v.P;
      for (const auto& rd : reads) r += ((((TW - rd.second % TW) % TW) + gg + TW - 1) / TW) * 4;
      return r;
   };
   while (G > TW && regs(G) > 200) G /= 2;
   if (G < TW || G < 2 || regs(G) > 200) return rp;
   rp.vec = true;
   rp.G = G;
   rp.TW = TW;
   rp.pad = 4 / v.P;
   uint32_t lane = 0;
   for (const Line& L : g.lines)
      if (L.in_lds) lane += (L.lds_size + rp.pad) * v.P;
   rp.lane_floats = lane;
   rp.slots = (lane + v.P - 1) / v.P;
   return rp;
}

std::string gen_config(const Graph& g, const Variant& v)
{
   std::ostringstream o;
   o << "// generated by libflowz_hip -- graph configuration\n";
   o << "#define FZ_NIN " << g.n_in << "\n";
   o << "#define FZ_NOUT " << g.n_out << "\n";
   o << "#define FZ_NCONST " << g.consts.size() << "\n";
   o << "#define FZ_NCONST64 " << g.consts64.size() << "\n";
   o << "#define FZ_NPARAM " << g.n_param << "\n";
   o << "#define FZ_NMOD " << g.n_mod << "\n";
   o << "#define FZ_NSTATE " << g.n_state << "\n";
   o << "#define FZ_P " << v.P << "\n";
   o << "#define FZ_U " << v.U << "\n";
   o << "#define FZ_BLOCK " << v.block << "\n";
   o << "#define FZ_FLAGS " << v.flags << "u\n";
   const RingPlan rp = ring_plan(g, v);
   o << "#define FZ_LDS_SLOTS " << rp.slots << "   // V slots per lane of the LDS rings\n";
   o << "#define FZ_RING_G " << (rp.vec ? rp.G : 0u) << "   // LDS rings vectorised in time: reads fetched / pushes flushed every so many steps (0: in place)\n";
   // the long-run stream-major body with 64-sample phases: patches of 19 KB per wave, so two workgroups fit a CU -- if the
   // kernel stays within 256 registers (it needs 258 left alone): ask for two waves per SIMD
   o << "#define FZ_MINWAVES " << (((v.flags & FZ_VF_SM_LONG) && v.U == 64 && v.P == 1) ? 2 : 0) << "\n";
   {  // occupancy cap (flags bits 20..22 = max workgroups per CU): pad the workgroup's LDS so that no
      // more than that many fit into the CU's 160 KiB
      const uint32_t cap = (v.flags >> 20) & 7u;
      uint32_t pad = 0;
      if (cap) {
         const uint32_t ring = rp.slots * v.block * 4u * v.P, want = 160u * 1024u / (cap + 1) + 1024u;
         pad = want > ring ? (std::min(want, 160u * 1024u) - ring) / 4u : 0u;
      }
      o << "#define FZ_OCC_PAD " << pad << "\n";
   }
   o << "#define FZ_KERNEL " << kernel_symbol(g, v) << "\n";
   o << "#define FZ_NFR " << g.far_reads.size() << "   // far (HBM ring) delayed reads per sample\n";
   o << "#define FZ_NFW " << g.far_lines.size() << "   // far delay lines = ring rows written per sample\n";
   // FZ_SKEW = lag of the last segment behind the first one (number of segments - 1), 0 = off
   o << "#define FZ_SKEW " << ((v.flags & FZ_VF_STAGE_PACK) && g.split.ok ? g.split.atoms() - 1 : 0) << "\n";
   if (const uint32_t W = ws_parts(v.flags)) {
      const std::vector<Graph>* roles = g.wave_roles(W);
      if (!roles) fail(FZ_E_UNSUPPORTED, "wave split: the graph is not that many isomorphic parts in series");
      o << "#define FZ_WS_W " << W << "   // compute waves per stream tuple = parts of the graph\n";
      o << "#define FZ_WS_IO " << ws_io(v.flags) << "   // one more wave per tuple for the frame I/O\n";
      o << "#define FZ_WS_IOW " << ws_io_waves(v.flags) << "   // ... or two: a loader and a storer\n";
      for (uint32_t k = 0; k < 4; ++k)
         o << "#define FZ_WS_K" << k << " " << (k < W && roles ? (*roles)[k].split.atoms() : 0u) << "   // skewed units (segments x atoms) of part " << k << "\n";
   }
   return o.str();
}

static std::string gen_body_skew(const Graph& g, const StageSplit& sp);

std::string gen_body(const Graph& g, const Variant& v)
{
   if (const uint32_t W = ws_parts(v.flags)) {
      // the parts of the graph, each a stage-packed body of its own (struct fz_r0::fz_graph, fz_r1::fz_graph, ...)
      const std::vector<Graph>* roles = g.wave_roles(W);
      if (!roles) fail(FZ_E_UNSUPPORTED, "wave split: the graph is not that many isomorphic parts in series");
      std::string s;
      for (uint32_t r = 0; r < 4; ++r) {
         if (r < W) s += "namespace fz_r" + std::to_string(r) + " {\n" + gen_body_skew((*roles)[r], (*roles)[r].split) + "}\n#undef FZ_NSEG\n";
         else s += "namespace fz_r" + std::to_string(r) + " { typedef fz_r0::fz_graph fz_graph; }\n";   // (unused: keeps the dispatch uniform)
      }
      return s;
   }
   if (v.flags & FZ_VF_STAGE_PACK) return gen_body_skew(g, g.split);
   std::ostringstream o;
   auto val = [&](uint32_t id) { return "v" + std::to_string(id); };
   auto reg = [&](size_t line, uint32_t age) { return "r" + std::to_string(line) + "_" + std::to_string(age); };
   // slot of ring position `pos` (an unsigned expression that may have wrapped below zero by at most a ring's size: callers write
   // `n - delay`, `0u - 1u - j`): a mask for power-of-two rings, else a modulo of the position shifted up by a multiple of the size
   // that the wrap cannot reach (2^32 is not a multiple of the size, so the wrapped value itself must not be reduced)
   auto ring_idx = [&](const Line& l, const std::string& pos) {
      if ((l.lds_size & (l.lds_size - 1)) == 0) return "((" + pos + ") & " + std::to_string(l.lds_size - 1) + "u)";
      return "fz_ring_mod<" + std::to_string(l.lds_size) + "u>(" + pos + ")";
   };
   const RingPlan rp = ring_plan(g, v);
   // (vectorised rings: float offset of line l's row of this lane = (floats of the lines before it) * FZ_BLOCK + tid * (floats per row))
   std::vector<uint32_t> ring_before(g.lines.size(), 0), ring_row(g.lines.size(), 0);
   if (rp.vec) {
      uint32_t acc = 0;
      for (size_t l = 0; l < g.lines.size(); ++l)
         if (g.lines[l].in_lds) {
            ring_before[l] = acc;
            ring_row[l] = (g.lines[l].lds_size + rp.pad) * v.P;
            acc += ring_row[l];
         }
   }
   auto line_index = [&](const Line& l) { return (size_t)(&l - g.lines.data()); };
   auto ring_base = [&](const Line& l) {
      const size_t li = line_index(l);
      return "(" + std::to_string(ring_before[li]) + "u * FZ_BLOCK + tid * " + std::to_string(ring_row[li]) + "u)";
   };
   auto ring_at = [&](const Line& l, const std::string& pos) {
      if (rp.vec) return "(*reinterpret_cast<V*>(reinterpret_cast<float*>(ring) + " + ring_base(l) + " + " + ring_idx(l, pos) + " * " + std::to_string(v.P) + "u))";
      return "ring[(" + std::to_string(l.lds_slot0) + "u + " + ring_idx(l, pos) + ") * FZ_BLOCK + tid]";
   };
   auto ring_vec_at = [&](const Line& l, const std::string& pos) {      // 16 bytes = 4 / P consecutive slots of the lane, `pos` on that grid
      return "(*reinterpret_cast<fz_f4*>(reinterpret_cast<float*>(ring) + " + ring_base(l) + " + " + ring_idx(l, pos) + " * " + std::to_string(v.P) + "u))";
   };

   // a double ring keeps low and high words in two float rings of lds_size slots each
   auto ring_hi = [&](const Line& l, const std::string& pos) {
      return "ring[(" + std::to_string(l.lds_slot0 + l.lds_size) + "u + " + ring_idx(l, pos) + ") * FZ_BLOCK + tid]";
   };
   o << "// generated by libflowz_hip -- graph body: " << g.nodes.size() << " nodes, " << g.n_ops
     << " float32 ops/sample, " << g.lines.size() << " delay lines, " << g.n_state << " state floats\n";
   // (coefficient VALUES are not part of the text: they travel in the kernarg, and graphs that differ only in
   // literal values share one code object and one kernel-cache entry)
   o << "__device__ __forceinline__ V fz_div(V a, V b) { return a / b; }\n";
   o << "__device__ __forceinline__ VD fz_div(VD a, VD b) { return a / b; }\n";
   bool has_cmp = false;
   for (const Node& nd : g.nodes) has_cmp = has_cmp || (nd.kind >= FZ_IR_LT && nd.kind <= FZ_IR_NE);
   if (has_cmp) {
      // the comparison operators of C++ on the streams of a lane: 1.0f / 0.0f (IEEE: every comparison with a NaN is false, != true); operands in
      // their common type.  Part of the GRAPH's text: kernels of graphs without comparisons keep their code.
      o << "template <int K, typename T> __device__ __forceinline__ bool fz_cmp1(T a, T b) { return K == " << FZ_IR_LT << " ? a < b : K == " << FZ_IR_LE
        << " ? a <= b : K == " << FZ_IR_GT << " ? a > b : K == " << FZ_IR_GE << " ? a >= b : K == " << FZ_IR_EQ << " ? a == b : a != b; }\n";
      o << "#if FZ_P == 1\n";
      o << "template <int K> __device__ __forceinline__ V fz_cmp(V a, V b) { return fz_cmp1<K>(a, b) ? 1.f : 0.f; }\n";
      o << "template <int K> __device__ __forceinline__ V fz_cmp(VD a, VD b) { return fz_cmp1<K>(a, b) ? 1.f : 0.f; }\n";
      o << "#else\n";
      for (const char* T : {"V", "VD"})
         o << "template <int K> __device__ __forceinline__ V fz_cmp(" << T << " a, " << T << " b)\n{\n   V r;\n#pragma unroll\n   for (int j = 0; j < FZ_P; ++j) r[j] = fz_cmp1<K>(a[j], b[j]) ? 1.f : 0.f;\n   return r;\n}\n";
      o << "#endif\n";
   }
   // operand `id` as seen by a node of type f64/f32 (C++ usual arithmetic conversions: float -> double is exact)
   auto opnd = [&](uint32_t id, bool want64) {
      return (want64 && !g.nodes[id].f64) ? "fz_cvt_d(" + val(id) + ")" : val(id);
   };
   // value of node `id` as float32 (delay lines and output frames are float: flowz.hpp:1245, rotate_push_back :130-137)
   auto as_f32 = [&](uint32_t id) { return g.nodes[id].f64 ? "fz_cvt_f(" + val(id) + ")" : val(id); };
   o << "struct fz_graph {\n";
   for (size_t l = 0; l < g.lines.size(); ++l) {
      const Line& L = g.lines[l];
      if (L.in_lds) {
         o << "   // line " << l << ": node " << L.src << ", depth " << L.depth << " -> LDS ring slots ["
           << L.lds_slot0 << ", " << L.lds_slot0 + L.lds_size << ")\n";
         continue;
      }
      if (L.far) {
         o << "   // line " << l << ": node " << L.src << ", depth " << L.depth << " -> ring in HBM: state rows ["
           << L.row0 << ", " << L.row0 + L.depth << "), phase in row " << L.phase_row << "\n";
         continue;
      }
      o << (L.f64 ? "   VD" : "   V");
      for (uint32_t a = 1; a <= L.depth; ++a) o << (a > 1 ? ", " : " ") << reg(l, a);
      o << ";   // line " << l << ": node " << L.src << " delayed by 1.." << L.depth << (L.f64 ? " (double state)" : "") << "\n";
   }
   for (uint32_t k = 0; k < g.n_param; ++k) o << "   V p" << k << ";\n";
   // LDS ring reads of a chunk, fetched TOGETHER at the start of the chunk (round 4).  A read `n - d` inside step() cannot be moved
   // above the ring write of the step before it (the compiler cannot tell the slots apart), so every step waited one LDS round trip per
   // line with nothing to hide it behind (one or two waves per SIMD: the rings fill the LDS) -- 240 clocks per step for the two combs of
   // the bench line's lds_ring graph.  A read whose delay is at least the chunk length refers to a slot written BEFORE the chunk
   // started, for every step of the chunk: all of a chunk's reads can be issued up front, into registers (`lr<k>[u]`), one round
   // trip per chunk.  Float lines, frame kernels (the stream-major bodies call step() without a chunk position and read in place);
   // at most 96 registers.
   struct RingRead { uint32_t line, d, m, nv; };              // m: slots between the 16-byte grid and the read's first slot; nv: vectors per sub-chunk
   std::vector<RingRead> ring_reads;
   std::map<std::pair<uint32_t, uint32_t>, size_t> ring_read_of;
   if (rp.vec) {
      for (const Node& nd : g.nodes) {
         if (nd.kind != FZ_IR_DELAY) continue;
         const int l = g.line_of_node[nd.a];
         if (!g.lines[(size_t)l].in_lds) continue;
         const uint32_t m = (rp.TW - nd.b % rp.TW) % rp.TW;
         if (ring_read_of.emplace(std::make_pair((uint32_t)l, nd.b), ring_reads.size()).second)
            ring_reads.push_back(RingRead{(uint32_t)l, nd.b, m, (m + rp.G + rp.TW - 1) / rp.TW});
      }
      for (size_t k = 0; k < ring_reads.size(); ++k)
         o << "   fz_f4 lr" << k << "[" << ring_reads[k].nv << "];   // line " << ring_reads[k].line << " read " << ring_reads[k].d
           << " samples back: the sub-chunk's slots as 16-byte vectors, the first value " << ring_reads[k].m << " slots in\n";
      for (size_t l = 0; l < g.lines.size(); ++l)
         if (g.lines[l].in_lds) o << "   fz_f4 wb" << l << "[" << rp.G / rp.TW << "];   // the sub-chunk's pushes of line " << l << "\n";
   } else if (!(v.flags & FZ_VF_STREAM_MAJOR) && v.U >= 2) {
      for (const Node& nd : g.nodes) {
         if (nd.kind != FZ_IR_DELAY) continue;
         const int l = g.line_of_node[nd.a];
         const Line& L = g.lines[(size_t)l];
         if (!L.in_lds || L.f64 || L.far || nd.b < v.U) continue;
         if (ring_read_of.emplace(std::make_pair((uint32_t)l, nd.b), ring_reads.size()).second) ring_reads.push_back(RingRead{(uint32_t)l, nd.b, 0, 0});
      }
      if (ring_reads.size() * v.U * v.P > 96) {
         ring_reads.clear();
         ring_read_of.clear();
      }
      for (size_t k = 0; k < ring_reads.size(); ++k)
         o << "   V lr" << k << "[FZ_U];   // line " << ring_reads[k].line << " read " << ring_reads[k].d << " samples back, the steps of the chunk at hand\n";
   }
   o << "   const float* mod = nullptr;   // sample-rate modulators [n_mod][mod_stride], set by the kernel (row 0 of the block)\n";
   o << "   unsigned mod_stride = 0;\n";
   // far lines: ring geometry for the skeleton, shadow registers for their short reads
   auto fidx = [&](size_t line) {
      for (size_t k = 0; k < g.far_lines.size(); ++k)
         if (g.far_lines[k] == line) return k;
      fail(FZ_E_GRAPH, "internal: not a far line");
   };
   auto shadow = [&](size_t line, uint32_t age) { return "fs" + std::to_string(line) + "_" + std::to_string(age); };
   if (!g.far_lines.empty()) {
      auto table = [&](const char* name, size_t n, auto get) {
         o << "   static constexpr unsigned " << name << "[" << std::max<size_t>(n, 1) << "] = {";
         for (size_t k = 0; k < std::max<size_t>(n, 1); ++k) o << (k ? ", " : "") << (k < n ? get(k) : 0u);
         o << "};\n";
      };
      table("fw_row0", g.far_lines.size(), [&](size_t k) { return g.lines[g.far_lines[k]].row0; });
      table("fw_depth", g.far_lines.size(), [&](size_t k) { return g.lines[g.far_lines[k]].depth; });
      table("fw_phase_row", g.far_lines.size(), [&](size_t k) { return g.lines[g.far_lines[k]].phase_row; });
      table("fr_line", g.far_reads.size(), [&](size_t k) { return (uint32_t)fidx(g.far_reads[k].line); });
      table("fr_n", g.far_reads.size(), [&](size_t k) { return g.far_reads[k].n; });
      for (uint32_t li : g.far_lines)
         if (g.lines[li].shadow) {
            o << "   V";
            for (uint32_t a = 1; a <= g.lines[li].shadow; ++a) o << (a > 1 ? ", " : " ") << shadow(li, a);
            o << ";   // newest values of far line " << li << " (node " << g.lines[li].src << ")\n";
         }
   }

   // ---- per-stream coefficients
   o << "   __device__ __forceinline__ void load_params(const float* pp, size_t ns, unsigned soff)\n   {\n";
   o << "      (void)pp; (void)ns; (void)soff;\n";
   for (uint32_t k = 0; k < g.n_param; ++k)
      o << "      p" << k << " = fz_ld_row(pp + (size_t)" << k << " * ns, soff, ns);\n";
   o << "   }\n";

   // ---- state in: row (row0 + j) holds the wire's value at t-1-j
   o << "   __device__ __forceinline__ void load_state(const float* st, size_t ns, unsigned soff, V* ring, unsigned tid, const unsigned* ph)\n   {\n";
   o << "      (void)st; (void)ns; (void)soff; (void)ring; (void)tid; (void)ph;\n";
   for (size_t l = 0; l < g.lines.size(); ++l) {
      const Line& L = g.lines[l];
      if (L.far) {
         // the newest values sit just behind the ring phase: age j+1 at slot (ph - 1 - j) mod D
         for (uint32_t j = 0; j < L.shadow; ++j)
            o << "      " << shadow(l, j + 1) << " = fz_ld_row(st + (size_t)(" << L.row0 << "u + (ph[" << fidx(l) << "] + " << (L.depth - 1 - j)
              << "u) % " << L.depth << "u) * ns, soff, ns);\n";
         continue;
      }
      if (L.f64 && L.in_lds) {
         o << "      for (unsigned j = 0; j < " << L.depth << "u; ++j) {\n";
         o << "         const VD d_ = fz_ld_row64(st + (size_t)(" << L.row0 << "u + 2u * j) * ns, soff);\n";
         o << "         " << ring_at(L, "0u - 1u - j") << " = fz_lo_d(d_);\n";
         o << "         " << ring_hi(L, "0u - 1u - j") << " = fz_hi_d(d_);\n";
         o << "      }\n";
      } else if (L.f64) {
         for (uint32_t j = 0; j < L.depth; ++j)
            o << "      " << reg(l, j + 1) << " = fz_ld_row64(st + (size_t)" << (L.row0 + 2 * j) << " * ns, soff);\n";
      } else if (!L.in_lds) {
         for (uint32_t j = 0; j < L.depth; ++j)
            o << "      " << reg(l, j + 1) << " = fz_ld_row(st + (size_t)" << (L.row0 + j) << " * ns, soff, ns);\n";
      } else {
         o << "      for (unsigned j = 0; j < " << L.depth << "u; ++j)\n";
         o << "         " << ring_at(L, "0u - 1u - j") << " = fz_ld_row(st + (size_t)(" << L.row0 << "u + j) * ns, soff, ns);\n";
      }
   }
   o << "   }\n";

   // ---- state out after n_done samples
   o << "   __device__ __forceinline__ void store_state(float* st, size_t ns, unsigned soff, V* ring, unsigned tid, unsigned n_done)\n   {\n";
   o << "      (void)st; (void)ns; (void)soff; (void)ring; (void)tid; (void)n_done;\n";
   for (size_t l = 0; l < g.lines.size(); ++l) {
      const Line& L = g.lines[l];
      if (L.far) continue;                 // the ring rows are written sample by sample
      if (L.f64 && L.in_lds) {
         o << "      for (unsigned j = 0; j < " << L.depth << "u; ++j)\n";
         o << "         fz_st_row64(st + (size_t)(" << L.row0 << "u + 2u * j) * ns, soff, fz_join_d(" << ring_at(L, "n_done - 1u - j") << ", "
           << ring_hi(L, "n_done - 1u - j") << "));\n";
      } else if (L.f64) {
         for (uint32_t j = 0; j < L.depth; ++j)
            o << "      fz_st_row64(st + (size_t)" << (L.row0 + 2 * j) << " * ns, soff, " << reg(l, j + 1) << ");\n";
      } else if (!L.in_lds) {
         for (uint32_t j = 0; j < L.depth; ++j)
            o << "      fz_st_row(st + (size_t)" << (L.row0 + j) << " * ns, soff, ns, " << reg(l, j + 1) << ");\n";
      } else {
         o << "      for (unsigned j = 0; j < " << L.depth << "u; ++j)\n";
         o << "         fz_st_row(st + (size_t)(" << L.row0 << "u + j) * ns, soff, ns, " << ring_at(L, "n_done - 1u - j") << ");\n";
      }
   }
   o << "   }\n";

   // ---- one sample
   o << "   // hr: values of the far delayed reads of this sample (prefetched from the HBM rings); hw: values to append to the rings\n";
   o << "   // mv / mvs: this sample's modulator values, modulator k at mv[k * mvs]\n";
   o << "   // the LDS ring reads of the chunk that starts at sample n0 (see lr<k> above)\n";
   o << "   __device__ __forceinline__ void ring_prefetch(V* ring, unsigned tid, unsigned n0)\n   {\n";
   o << "      (void)ring; (void)tid; (void)n0;\n";
   if (rp.vec) {
      // n0: first sample of the sub-chunk (a multiple of FZ_RING_G); vector j of read k starts at sample n0 - d - m + j * (4 / P)
      for (size_t k = 0; k < ring_reads.size(); ++k)
         for (uint32_t j = 0; j < ring_reads[k].nv; ++j)
            o << "      lr" << k << "[" << j << "] = " << ring_vec_at(g.lines[ring_reads[k].line], "n0 + " + std::to_string(j * rp.TW) + "u - " + std::to_string(ring_reads[k].d + ring_reads[k].m) + "u") << ";\n";
   } else if (!ring_reads.empty()) {
      o << "      _Pragma(\"unroll\") for (int u = 0; u < FZ_U; ++u)\n      {\n";
      for (size_t k = 0; k < ring_reads.size(); ++k)
         o << "         lr" << k << "[u] = " << ring_at(g.lines[ring_reads[k].line], "n0 + (unsigned)u - " + std::to_string(ring_reads[k].d) + "u") << ";\n";
      o << "      }\n";
   }
   o << "   }\n";
   o << "   // the pushes of the sub-chunk that started at sample n0, kept in registers by its steps: 16 bytes per store\n";
   o << "   __device__ __forceinline__ void ring_flush(V* ring, unsigned tid, unsigned n0)\n   {\n";
   o << "      (void)ring; (void)tid; (void)n0;\n";
   if (rp.vec)
      for (size_t l = 0; l < g.lines.size(); ++l)
         if (g.lines[l].in_lds)
            for (uint32_t j = 0; j < rp.G / rp.TW; ++j)
               o << "      " << ring_vec_at(g.lines[l], "n0 + " + std::to_string(j * rp.TW) + "u") << " = wb" << l << "[" << j << "];\n";
   o << "   }\n";
   o << "   // u: the step's position in the chunk whose ring reads were prefetched (ring_prefetch), or -1: read the rings in place\n";
   o << "   __device__ __forceinline__ void step(const V* x, VO* y, const float* c, const double* cd, V* ring, unsigned tid, unsigned n, const V* hr, V* hw, const float* mv, unsigned mvs, int u = -1)\n   {\n";
   o << "      (void)x; (void)c; (void)cd; (void)ring; (void)tid; (void)n; (void)hr; (void)hw; (void)mv; (void)mvs; (void)u;\n";
   for (size_t id = 0; id < g.nodes.size(); ++id) {
      const Node& nd = g.nodes[id];
      const bool d = nd.f64;
      o << "      const " << (d ? "VD " : "V ") << val((uint32_t)id) << " = ";
      switch (nd.kind) {
         case FZ_IR_INPUT:
            if (d) o << "fz_join_d(x[" << nd.a << "], x[" << nd.a + 1 << "])";     // a double input wire: (low word, high word) slots
            else o << "x[" << nd.a << "]";
            break;
         case FZ_IR_CONST:
            if (d) o << "(VD)(cd[" << nd.a << "])";
            else o << "(V)(c[" << nd.a << "])";
            break;
         case FZ_IR_PARAM: o << "p" << nd.a; break;
         case FZ_IR_MOD: o << "(V)(fz_uniform_f(mv[" << nd.a << "u * mvs]))"; break;     // this sample's value of modulator a: wave-uniform (a scalar load, or a value the kernel prefetched with the chunk's rows)
         case FZ_IR_DELAY: {
            const int l = g.line_of_node[nd.a];
            const Line& L = g.lines[(size_t)l];
            if (L.far) {
               if (nd.b <= L.shadow) o << shadow((size_t)l, nd.b);
               else {
                  size_t r = 0;
                  while (r < g.far_reads.size() && !(g.far_reads[r].line == (uint32_t)l && g.far_reads[r].n == nd.b)) ++r;
                  o << "hr[" << r << "]";
               }
            } else if (!L.in_lds) o << reg((size_t)l, nd.b);
            else if (L.f64) o << "fz_join_d(" << ring_at(L, "n - " + std::to_string(nd.b) + "u") << ", " << ring_hi(L, "n - " + std::to_string(nd.b) + "u") << ")";
            else {
               const auto it = ring_read_of.find(std::make_pair((uint32_t)l, nd.b));
               if (it != ring_read_of.end() && rp.vec)
                  o << "(u >= 0 ? fz_ring_pick(lr" << it->second << ", " << ring_reads[it->second].m << " + (u % " << rp.G << ")) : " << ring_at(L, "n - " + std::to_string(nd.b) + "u") << ")";
               else if (it != ring_read_of.end()) o << "(u >= 0 ? lr" << it->second << "[u] : " << ring_at(L, "n - " + std::to_string(nd.b) + "u") << ")";
               else o << ring_at(L, "n - " + std::to_string(nd.b) + "u");
            }
            break;
         }
         case FZ_IR_ADD: o << opnd(nd.a, d) << " + " << opnd(nd.b, d); break;
         case FZ_IR_SUB: o << opnd(nd.a, d) << " - " << opnd(nd.b, d); break;
         case FZ_IR_MUL: o << opnd(nd.a, d) << " * " << opnd(nd.b, d); break;
         case FZ_IR_DIV: o << "fz_div(" << opnd(nd.a, d) << ", " << opnd(nd.b, d) << ")"; break;
         case FZ_IR_NEG: o << "-" << val(nd.a); break;
         case FZ_IR_LT: case FZ_IR_LE: case FZ_IR_GT: case FZ_IR_GE: case FZ_IR_EQ: case FZ_IR_NE: {
            const bool dd = g.nodes[nd.a].f64 || g.nodes[nd.b].f64;   // compared in double when one operand is; the node itself is a float
            o << "fz_cmp<" << nd.kind << ">(" << opnd(nd.a, dd) << ", " << opnd(nd.b, dd) << ")";
            break;
         }
         case FZ_IR_ABSLT: o << "fz_abs_lt(" << opnd(nd.a, d) << ", " << opnd(nd.b, d) << ")"; break;
         case FZ_IR_SELECT: o << "fz_select(" << val(nd.a) << ", " << opnd(nd.b, d) << ", " << opnd(nd.c, d) << ")"; break;
         case FZ_IR_WIDEN: o << "fz_cvt_d(" << val(nd.a) << ")"; break;
         case FZ_IR_NARROW: o << "fz_cvt_f(" << val(nd.a) << ")"; break;
         default: fail(FZ_E_GRAPH, "internal: unknown IR node kind");
      }
      o << ";\n";
   }
   for (size_t j = 0; j < g.outputs.size(); ++j) {
      const uint32_t id = g.outputs[j];
      if (g.out_part[j] >= 3 && (g.out_part[j] & 1)) o << "      y[" << j << "] = fz_lo_d(" << val(id) << ");\n";   // typed: a double (part), low word
      else if (g.out_part[j] >= 4) o << "      y[" << j << "] = fz_hi_d(" << val(id) << ");\n";
      else if (v.flags & FZ_VF_OUT_F64) o << "      y[" << j << "] = " << (g.nodes[id].f64 ? val(id) : "fz_cvt_d(" + val(id) + ")") << ";\n";
      else o << "      y[" << j << "] = " << as_f32(id) << ";\n";
   }
   // pushes last: every read above saw the values of previous samples (flowz.hpp:994, :1067)
   for (size_t l = 0; l < g.lines.size(); ++l) {
      const Line& L = g.lines[l];
      if (L.far) {
         o << "      hw[" << fidx(l) << "] = " << as_f32(L.src) << ";\n";
         for (uint32_t a = L.shadow; a >= 2; --a) o << "      " << shadow(l, a) << " = " << shadow(l, a - 1) << ";\n";
         if (L.shadow) o << "      " << shadow(l, 1) << " = " << as_f32(L.src) << ";\n";
         continue;
      }
      if (!L.in_lds) {
         for (uint32_t a = L.depth; a >= 2; --a) o << "      " << reg(l, a) << " = " << reg(l, a - 1) << ";\n";
         o << "      " << reg(l, 1) << " = " << (L.f64 ? val(L.src) : as_f32(L.src)) << ";\n";
      } else if (L.f64) {
         o << "      " << ring_at(L, "n") << " = fz_lo_d(" << val(L.src) << ");\n";
         o << "      " << ring_hi(L, "n") << " = fz_hi_d(" << val(L.src) << ");\n";
      } else if (rp.vec) {
         o << "      if (u >= 0) fz_ring_put(wb" << l << ", u % " << rp.G << ", " << as_f32(L.src) << ");\n";
         o << "      else " << ring_at(L, "n") << " = " << as_f32(L.src) << ";\n";
      } else {
         o << "      " << ring_at(L, "n") << " = " << as_f32(L.src) << ";\n";
      }
   }
   o << "   }\n";
   o << "};\n";
   return o.str();
}

// Stage-packed body (FZ_VF_STAGE_PACK, one stream per lane): K isomorphic segments, segment j at
// time t-j; stream i of packed float2 operations carries (segment i, segment i + K/2), so that the
// packed output of stream i-1 IS the packed input of stream i in the next step (no shuffles).
static std::string gen_body_skew(const Graph& g, const StageSplit& sp)
{
   if (!sp.ok) fail(FZ_E_UNSUPPORTED, "graph is not a series of isomorphic segments: stage packing does not apply");
   const uint32_t K = sp.K, NS = K / 2, M = sp.m, NA = K * M;   // segments, packed streams, atoms per segment, atoms in all
   std::ostringstream o;
   std::map<std::vector<uint32_t>, size_t> tid;
   for (size_t k = 0; k < sp.tuples.size(); ++k) tid[sp.tuples[k]] = k;
   auto operand_tuple = [&](const std::vector<uint32_t>& t, bool second) {
      std::vector<uint32_t> r(K);
      for (uint32_t j = 0; j < K; ++j) r[j] = second ? g.nodes[t[j]].b : g.nodes[t[j]].a;
      return r;
   };
   auto sub_of = [&](const std::vector<uint32_t>& t) -> uint32_t {
      auto it = tid.find(t);
      if (it == tid.end()) fail(FZ_E_GRAPH, "internal: unmatched operand in stage packing");
      return sp.sub.empty() ? 0u : sp.sub[it->second];
   };
   std::map<std::pair<std::vector<uint32_t>, uint32_t>, size_t> lid;   // (source tuple, frame) -> packed line
   for (size_t l = 0; l < sp.lines.size(); ++l) lid[{sp.lines[l].srcs, sp.lines[l].frame}] = l;
   auto q = [&](size_t l, uint32_t i, uint32_t age) {
      return "q" + std::to_string(l) + "_" + std::to_string(i) + "_" + std::to_string(age);
   };
   auto cy = [&](uint32_t a, uint32_t i) { return "cy" + std::to_string(a) + "_" + std::to_string(i); };   // carry INTO atom a
   // the packed value of tuple t as atom `rs` of stream i sees it in this step: its own atom's fresh value, the carry that
   // brought the internal cut wire over from the atom before (one step ago), or a delayed read in the atom's own time frame
   auto w = [&](const std::vector<uint32_t>& t, uint32_t i, uint32_t rs = ~0u) -> std::string {
      auto it = tid.find(t);
      if (it == tid.end()) fail(FZ_E_GRAPH, "internal: unmatched operand in stage packing");
      const Node& n0 = g.nodes[t[0]];
      if (rs != ~0u && n0.kind == FZ_IR_DELAY) {
         auto li = lid.find({operand_tuple(t, false), rs});
         if (li == lid.end()) fail(FZ_E_GRAPH, "internal: delayed read without a line in the reader's time frame");
         return q(li->second, i, n0.b);
      }
      const bool leaf = n0.kind == FZ_IR_CONST || n0.kind == FZ_IR_PARAM || t[0] == sp.cuts[0];
      if (rs != ~0u && !leaf && sub_of(t) != rs) {
         if (rs == 0 || rs - 1 >= sp.icuts.size() || sp.icuts[rs - 1] != t)
            fail(FZ_E_GRAPH, "internal: an atom reads an earlier atom past its cut wire");
         return cy(rs, i);
      }
      return "w" + std::to_string(it->second) + "_" + std::to_string(i);
   };
   auto row_of = [&](uint32_t s, uint32_t j) -> long {
      const int l = g.line_of_node[s];
      if (l < 0 || j >= g.lines[(size_t)l].depth) return -1;
      return (long)g.lines[(size_t)l].row0 + j;
   };
   std::vector<uint32_t> root(K);
   for (uint32_t j = 0; j < K; ++j) root[j] = sp.cuts[j + 1];

   o << "// generated by libflowz_hip -- STAGE-PACKED graph body: " << K << " isomorphic segments of " << g.n_ops / K
     << " float32 ops, cut at nodes";
   for (uint32_t j = 1; j < K; ++j) o << " " << sp.cuts[j];
   o << "; segment j runs at time t-j, segments (i, i+" << NS << ") share packed stream i";
   if (M > 1) o << "; every segment is " << M << " atoms in series (atom a of segment j at time t-(j*" << M << "+a))";
   o << "\n#define FZ_NSEG " << NA << "   // skewed units in series (mask bits)\n";
   o << "struct fz_graph {\n";
   for (size_t l = 0; l < sp.lines.size(); ++l)
      for (uint32_t i = 0; i < NS; ++i) {
         o << "   fz_f2";
         for (uint32_t a = 1; a <= sp.lines[l].depth; ++a) o << (a > 1 ? ", " : " ") << q(l, i, a);
         o << ";   // (node " << sp.lines[l].srcs[i] << ", node " << sp.lines[l].srcs[i + NS] << ") delayed by 1.."
           << sp.lines[l].depth << "\n";
      }
   std::vector<size_t> ptuples;
   for (size_t k = 0; k < sp.tuples.size(); ++k)
      if (g.nodes[sp.tuples[k][0]].kind == FZ_IR_PARAM) ptuples.push_back(k);
   for (size_t k : ptuples)
      for (uint32_t i = 0; i < NS; ++i) o << "   fz_f2 pp" << k << "_" << i << ";\n";
   // scalar prefix (runs with segment 0): private delay lines and per-stream coefficients
   auto ps = [&](uint32_t l, uint32_t age) { return "ps" + std::to_string(l) + "_" + std::to_string(age); };
   std::vector<char> in_prefix(g.nodes.size(), 0);
   for (uint32_t v : sp.prefix) in_prefix[v] = 1;
   std::vector<char> is_prefix_line(g.lines.size(), 0);
   for (uint32_t l : sp.prefix_lines) {
      is_prefix_line[l] = 1;
      o << "   float";
      for (uint32_t a = 1; a <= g.lines[l].depth; ++a) o << (a > 1 ? ", " : " ") << ps(l, a);
      o << ";   // prefix line: node " << g.lines[l].src << "\n";
   }
   // scalar suffix (runs with the last segment): its private delay lines
   auto ss = [&](uint32_t l, uint32_t age) { return "ss" + std::to_string(l) + "_" + std::to_string(age); };
   std::vector<char> is_suffix_line(g.lines.size(), 0);
   for (uint32_t l : sp.suffix_lines) {
      is_suffix_line[l] = 1;
      o << "   float";
      for (uint32_t a = 1; a <= g.lines[l].depth; ++a) o << (a > 1 ? ", " : " ") << ss(l, a);
      o << ";   // suffix line: node " << g.lines[l].src << "\n";
   }
   std::vector<uint32_t> prefix_params;                   // per-stream coefficients of the scalar parts
   for (uint32_t v : sp.prefix)
      if (g.nodes[v].kind == FZ_IR_PARAM) prefix_params.push_back(g.nodes[v].a);
   for (uint32_t v : sp.suffix) {
      auto add_param = [&](uint32_t o) {
         if (g.nodes[o].kind == FZ_IR_PARAM && std::find(prefix_params.begin(), prefix_params.end(), g.nodes[o].a) == prefix_params.end())
            prefix_params.push_back(g.nodes[o].a);
      };
      add_param(v);
      const Node& nd = g.nodes[v];
      if (nd.kind >= FZ_IR_ADD && nd.kind <= FZ_IR_NEG) {
         add_param(nd.a);
         if (nd.kind != FZ_IR_NEG) add_param(nd.b);
      }
   }
   for (uint32_t k : prefix_params) o << "   float pf" << k << ";\n";
   o << "   const float* mod = nullptr;\n   unsigned mod_stride = 0;\n";
   o << "   fz_f2";
   for (uint32_t i = 0; i < NS; ++i) o << (i ? ", " : " ") << "cq" << i;
   o << ";   // cq_i: packed output of stream i = packed input of stream i+1 in the next step\n";
   for (uint32_t a = 1; a < M; ++a) {
      o << "   fz_f2";
      for (uint32_t i = 0; i < NS; ++i) o << (i ? ", " : " ") << cy(a, i);
      o << ";   // the internal cut wire in front of atom " << a << ", one step old\n";
   }

   o << "   __device__ __forceinline__ void load_params(const float* pp, size_t ns, unsigned soff)\n   {\n";
   o << "      (void)pp; (void)ns; (void)soff;\n";
   for (uint32_t i = 0; i < NS; ++i) o << "      cq" << i << " = (fz_f2){0.f, 0.f};\n";
   for (uint32_t a = 1; a < M; ++a)
      for (uint32_t i = 0; i < NS; ++i) o << "      " << cy(a, i) << " = (fz_f2){0.f, 0.f};\n";
   for (uint32_t k : prefix_params) o << "      pf" << k << " = fz_ld_f(pp + (size_t)" << k << " * ns, soff, ns);\n";
   for (size_t k : ptuples)
      for (uint32_t i = 0; i < NS; ++i)
         o << "      pp" << k << "_" << i << " = (fz_f2){fz_ld_f(pp + (size_t)" << g.nodes[sp.tuples[k][i]].a << " * ns, soff, ns), fz_ld_f(pp + (size_t)"
           << g.nodes[sp.tuples[k][i + NS]].a << " * ns, soff, ns)};\n";
   o << "   }\n";

   o << "   __device__ __forceinline__ void load_state(const float* st, size_t ns, unsigned soff, V* ring, unsigned tid, const unsigned* ph)\n   {\n";
   o << "      (void)st; (void)ns; (void)soff; (void)ring; (void)tid; (void)ph;\n";
   auto ld = [&](long r) { return r < 0 ? std::string("0.f") : "fz_ld_f(st + (size_t)" + std::to_string(r) + " * ns, soff, ns)"; };
   for (size_t l = 0; l < sp.lines.size(); ++l)
      for (uint32_t i = 0; i < NS; ++i)
         for (uint32_t j = 0; j < sp.lines[l].depth; ++j)
            o << "      " << q(l, i, j + 1) << " = (fz_f2){" << ld(row_of(sp.lines[l].srcs[i], j)) << ", "
              << ld(row_of(sp.lines[l].srcs[i + NS], j)) << "};\n";
   for (uint32_t l : sp.prefix_lines)
      for (uint32_t j = 0; j < g.lines[l].depth; ++j)
         o << "      " << ps(l, j + 1) << " = fz_ld_f(st + (size_t)" << (g.lines[l].row0 + j) << " * ns, soff, ns);\n";
   for (uint32_t l : sp.suffix_lines)
      for (uint32_t j = 0; j < g.lines[l].depth; ++j)
         o << "      " << ss(l, j + 1) << " = fz_ld_f(st + (size_t)" << (g.lines[l].row0 + j) << " * ns, soff, ns);\n";
   o << "   }\n";

   o << "   __device__ __forceinline__ void store_state(float* st, size_t ns, unsigned soff, V* ring, unsigned tid, unsigned n_done)\n   {\n";
   o << "      (void)st; (void)ns; (void)soff; (void)ring; (void)tid; (void)n_done;\n";
   for (size_t li = 0; li < g.lines.size(); ++li) {
      const Line& L = g.lines[li];
      for (uint32_t j = 0; j < L.depth; ++j) {
         if (is_prefix_line[li] || is_suffix_line[li]) {
            o << "      fz_st_f(st + (size_t)" << (L.row0 + j) << " * ns, soff, ns, " << (is_prefix_line[li] ? ps((uint32_t)li, j + 1) : ss((uint32_t)li, j + 1)) << ");\n";
            continue;
         }
         // every copy of a shared line holds the same values once all segments have caught up
         std::string src;
         for (size_t l = 0; l < sp.lines.size() && src.empty(); ++l)
            for (uint32_t p = 0; p < K && src.empty(); ++p)
               if (sp.lines[l].srcs[p] == L.src && j < sp.lines[l].depth) src = q(l, p % NS, j + 1) + (p >= NS ? ".y" : ".x");
         if (src.empty()) fail(FZ_E_GRAPH, "internal: delay line not covered by stage packing");
         o << "      fz_st_f(st + (size_t)" << (L.row0 + j) << " * ns, soff, ns, " << src << ");\n";
      }
   }
   o << "   }\n";

   o << "   // one step: segment j consumes/produces time t-j.  MASKED (prologue/epilogue steps only): bit j of\n";
   o << "   // `mask` says whether segment j runs; an idle segment keeps its delay lines and its carry.\n";
   o << "   template <bool MASKED>\n";
   o << "   __device__ __forceinline__ void step2(const V* x, VO* y, const float* c, unsigned mask)\n   {\n";
   o << "      (void)c; (void)mask;\n";
   // scalar prefix at the time of segment 0
   // a scalar part's delayed read of a wire the chain keeps packed lines of: the copy in the time frame of the atom the scalar
   // part runs with (prefix: atom 0 of segment 0; suffix: the last atom of the last segment)
   auto packed_comp_of = [&](uint32_t src, uint32_t age, uint32_t frame) -> std::string {
      for (size_t l = 0; l < sp.lines.size(); ++l)
         for (uint32_t pz = 0; pz < K; ++pz)
            if (sp.lines[l].frame == frame && sp.lines[l].srcs[pz] == src && age <= sp.lines[l].depth) return q(l, pz % NS, age) + (pz >= NS ? ".y" : ".x");
      fail(FZ_E_GRAPH, "internal: a scalar prefix / suffix reads a delay line that is not materialised in its time frame");
   };
   for (uint32_t v : sp.prefix) {
      const Node& nd = g.nodes[v];
      o << "      const float u" << v << " = ";
      switch (nd.kind) {
         case FZ_IR_INPUT: o << "x[0]"; break;
         case FZ_IR_CONST: o << "c[" << nd.a << "]"; break;
         case FZ_IR_PARAM: o << "pf" << nd.a; break;
         case FZ_IR_DELAY: {
            const int l = g.line_of_node[nd.a];
            if (l >= 0 && is_prefix_line[(size_t)l]) o << ps((uint32_t)l, nd.b);
            else o << packed_comp_of(nd.a, nd.b, 0);
            break;
         }
         case FZ_IR_ADD: o << "u" << nd.a << " + u" << nd.b; break;
         case FZ_IR_SUB: o << "u" << nd.a << " - u" << nd.b; break;
         case FZ_IR_MUL: o << "u" << nd.a << " * u" << nd.b; break;
         case FZ_IR_DIV: o << "u" << nd.a << " / u" << nd.b; break;
         case FZ_IR_NEG: o << "-u" << nd.a; break;
         default: fail(FZ_E_GRAPH, "internal: unknown IR node kind");
      }
      o << ";\n";
   }
   const std::string chain_in = g.nodes[sp.cuts[0]].kind == FZ_IR_INPUT ? std::string("x[0]") : "u" + std::to_string(sp.cuts[0]);
   for (size_t k = 0; k < sp.tuples.size(); ++k) {
      const auto& t = sp.tuples[k];
      const Node& n0 = g.nodes[t[0]];
      if (n0.kind == FZ_IR_DELAY) continue;                                    // (read where it is used, in the reader's time frame)
      for (uint32_t i = 0; i < NS; ++i) {
         o << "      const fz_f2 w" << k << "_" << i << " = ";
         const Node& na = g.nodes[t[i]];
         const Node& nb = g.nodes[t[i + NS]];
         if (t[0] == sp.cuts[0]) {                                        // the chain's input wire
            if (i == 0) o << "(fz_f2){" << chain_in << ", cq" << NS - 1 << ".x};\n";   // segment 0 <- input / prefix, segment NS <- segment NS-1
            else o << "cq" << i - 1 << ";\n";                             // segments (i, i+NS) <- segments (i-1, i-1+NS)
            continue;
         }
         const uint32_t rs = sp.sub.empty() ? 0u : sp.sub[k];                 // the atom this operation belongs to
         switch (n0.kind) {
            case FZ_IR_INPUT:
               break;
            case FZ_IR_CONST: o << "(fz_f2){c[" << na.a << "], c[" << nb.a << "]}"; break;
            case FZ_IR_PARAM: o << "pp" << k << "_" << i; break;
            case FZ_IR_ADD: o << w(operand_tuple(t, false), i, rs) << " + " << w(operand_tuple(t, true), i, rs); break;
            case FZ_IR_SUB: o << w(operand_tuple(t, false), i, rs) << " - " << w(operand_tuple(t, true), i, rs); break;
            case FZ_IR_MUL: o << w(operand_tuple(t, false), i, rs) << " * " << w(operand_tuple(t, true), i, rs); break;
            case FZ_IR_DIV: o << w(operand_tuple(t, false), i, rs) << " / " << w(operand_tuple(t, true), i, rs); break;
            case FZ_IR_NEG: o << "-" << w(operand_tuple(t, false), i, rs); break;
            default: fail(FZ_E_GRAPH, "internal: unknown IR node kind");
         }
         o << ";\n";
      }
   }
   if (sp.suffix.empty()) {
      o << "      y[0] = " << w(root, NS - 1) << ".y;\n";
   } else {
      // scalar suffix at the time of the last segment: what the output makes of the chain's end wire
      const uint32_t e = sp.cuts[K];
      auto sval = [&](uint32_t v) -> std::string {
         if (v == e) return w(root, NS - 1) + ".y";
         const Node& nd = g.nodes[v];
         if (nd.kind == FZ_IR_CONST) return "c[" + std::to_string(nd.a) + "]";
         if (nd.kind == FZ_IR_PARAM) return "pf" + std::to_string(nd.a);
         if (nd.kind == FZ_IR_DELAY && nd.a == e && std::find(sp.suffix.begin(), sp.suffix.end(), v) == sp.suffix.end())
            return packed_comp_of(e, nd.b, M - 1);                          // a delayed read of e shared with the chain
         return "z" + std::to_string(v);
      };
      for (uint32_t v : sp.suffix) {
         const Node& nd = g.nodes[v];
         if (nd.kind == FZ_IR_CONST || nd.kind == FZ_IR_PARAM) continue;
         o << "      const float z" << v << " = ";
         switch (nd.kind) {
            case FZ_IR_DELAY: {
               const int l = g.line_of_node[nd.a];
               if (l >= 0 && is_suffix_line[(size_t)l]) o << ss((uint32_t)l, nd.b);
               else o << packed_comp_of(nd.a, nd.b, M - 1);
               break;
            }
            case FZ_IR_ADD: o << sval(nd.a) << " + " << sval(nd.b); break;
            case FZ_IR_SUB: o << sval(nd.a) << " - " << sval(nd.b); break;
            case FZ_IR_MUL: o << sval(nd.a) << " * " << sval(nd.b); break;
            case FZ_IR_DIV: o << sval(nd.a) << " / " << sval(nd.b); break;
            case FZ_IR_NEG: o << "-" << sval(nd.a); break;
            default: fail(FZ_E_GRAPH, "internal: unexpected node in the scalar suffix");
         }
         o << ";\n";
      }
      o << "      y[0] = " << sval(g.outputs[0]) << ";\n";
   }
   // atom `a` of segment `seg` is mask bit seg * M + a
   auto act = [&](uint32_t seg, uint32_t a) { return "(!MASKED || ((mask >> " + std::to_string(seg * M + a) + ") & 1u))"; };
   for (size_t l = 0; l < sp.lines.size(); ++l)
      for (uint32_t i = 0; i < NS; ++i) {
         const uint32_t fr = sp.lines[l].frame;                               // the line lives in the time frame of atom fr: pushed when that atom runs
         const std::string cur = w(sp.lines[l].srcs, i, fr);
         for (uint32_t a = sp.lines[l].depth; a >= 1; --a) {
            const std::string nv = a == 1 ? cur : q(l, i, a - 1);
            o << "      " << q(l, i, a) << " = (fz_f2){" << act(i, fr) << " ? " << nv << ".x : " << q(l, i, a) << ".x, "
              << act(i + NS, fr) << " ? " << nv << ".y : " << q(l, i, a) << ".y};\n";
         }
      }
   for (uint32_t a = 1; a < M; ++a)                                           // the internal cut wires travel on to the next atom
      for (uint32_t i = 0; i < NS; ++i) {
         const std::string cur = w(sp.icuts[a - 1], i);
         o << "      " << cy(a, i) << " = (fz_f2){" << act(i, a - 1) << " ? " << cur << ".x : " << cy(a, i) << ".x, " << act(i + NS, a - 1) << " ? "
           << cur << ".y : " << cy(a, i) << ".y};\n";
      }
   for (uint32_t l : sp.prefix_lines) {
      const Line& L = g.lines[l];
      const std::string cur = g.nodes[L.src].kind == FZ_IR_INPUT ? std::string("x[0]") : "u" + std::to_string(L.src);
      for (uint32_t a = L.depth; a >= 1; --a)
         o << "      " << ps(l, a) << " = " << act(0, 0) << " ? " << (a == 1 ? cur : ps(l, a - 1)) << " : " << ps(l, a) << ";\n";
   }
   for (uint32_t l : sp.suffix_lines) {
      const Line& L = g.lines[l];
      const std::string cur = L.src == sp.cuts[K] ? w(root, NS - 1) + ".y" : "z" + std::to_string(L.src);
      for (uint32_t a = L.depth; a >= 1; --a)
         o << "      " << ss(l, a) << " = " << act(K - 1, M - 1) << " ? " << (a == 1 ? cur : ss(l, a - 1)) << " : " << ss(l, a) << ";\n";
   }
   for (uint32_t i = 0; i < NS; ++i)
      o << "      cq" << i << " = (fz_f2){" << act(i, M - 1) << " ? " << w(root, i) << ".x : cq" << i << ".x, " << act(i + NS, M - 1) << " ? "
        << w(root, i) << ".y : cq" << i << ".y};\n";
   o << "   }\n";
   o << "};\n";
   return o.str();
}

std::string full_source(const Graph& g, const Variant& v)
{
   // single translation unit view (for inspection and for hashing the kernel cache key)
   std::string s;
   s += "// ==== fz_graph_config.h ====\n" + gen_config(g, v);
   s += "// ==== fz_graph_body.h ====\n" + gen_body(g, v);
   s += "// ==== fz_block_kernel.hip.inc ====\n";
   s += skeleton_source(v.flags);
   return s;
}

}  // namespace fz
