// Internal structures of libflowz_hip: expression trees, the lowered per-sample DAG, programs.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "flowz_hip.h"

namespace fz {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const std::string& msg);
struct Error {
   int code;
   std::string msg;
};
[[noreturn]] void fail(int code, const std::string& msg);

// ---- expression tree (the EDSL surface, flowz.hpp:68-93) ---------------------------------------
enum class EK : uint8_t { Placeholder, Delayed, Literal, Uniform, Param, Arith, Neg, Channel, Parallel, Sequence, Feedback, Modulator };

}  // namespace fz

struct fz_expr {
   std::atomic<int> refs{1};
   fz::EK kind;
   uint32_t i = 0;        // placeholder index / param index
   uint32_t n = 0;        // delay
   float value = 0.f;     // literal
   double value64 = 0.0;  // float64 literal
   bool f64 = false;      // literal is a C++ double
   bool cplx = false;     // literal is a std::complex<float> (value, value_im)
   float value_im = 0.f;
   double value64_im = 0.0; // imaginary part of a std::complex<double> literal (cplx && f64)
   fz_op op = FZ_OP_ADD;  // arith
   fz_expr* a = nullptr;
   fz_expr* b = nullptr;
   int in_arity = 0;
   int out_arity = 1;
};

namespace fz {

// an expression as text and back (fz_expr.cpp): what a kernel manifest records of a program
std::string serialize_expr(const fz_expr* root);
fz_expr* parse_expr(const std::string& text);

// stage packing (fz_split.cpp): the graph is K isomorphic segments in series; segment j runs at
// time t-j and segments 2i, 2i+1 share one packed float2 operation per node
struct PackedLine {
   std::vector<uint32_t> srcs;   // per segment: the source node of this delay line
   uint32_t depth;
   uint32_t frame = 0;           // the sub-atom in whose time frame this copy of the line lives (its readers' sub-atom)
};
struct StageSplit {
   bool ok = false;
   uint32_t K = 0;                                  // number of segments (even)
   std::vector<uint32_t> cuts;                      // c_0 = input node ... c_K = output node
   std::vector<std::vector<uint32_t>> tuples;       // per node of segment 0: its partner in every segment, evaluation order
   // Sub-atoms (round 3): every segment is itself cut at m - 1 internal wires into m ATOMS in series (a DF1 biquad: its
   // feed-forward sum and its recursion), and atom a of segment j runs at time t - (j * m + a).  The atoms of a step are then
   // independent of each other: twice the instruction-level parallelism, dependent chains half as long -- what a wave that
   // carries a single packed pair of segments (the parts of a wave split) needs to stop waiting for its own results.
   // The packed value of an internal cut travels to the next atom through a carry register, one step later.
   uint32_t m = 1;                                  // atoms per segment
   std::vector<uint32_t> sub;                       // per tuple: its atom (0 .. m-1); leaves (constants, parameters): the reader's
   std::vector<std::vector<uint32_t>> icuts;        // icuts[a-1]: the tuple whose value crosses from atom a-1 into atom a (a = 1 .. m-1)
   uint32_t atoms() const { return K * m; }         // skewed units in series = masked steps at either end of a block + 1
   std::vector<PackedLine> lines;
   std::vector<uint32_t> prefix;                    // scalar prefix (nodes cuts[0] depends on), evaluation order
   std::vector<uint32_t> prefix_lines;              // delay lines private to the prefix (indices into Graph::lines)
   std::vector<uint32_t> suffix;                    // scalar suffix (what the output makes of the chain's end wire), evaluation order
   std::vector<uint32_t> suffix_lines;              // delay lines private to the suffix
};

// ---- lowered DAG ---------------------------------------------------------------------------------
struct Node {
   uint32_t kind;   // fz_ir_kind
   uint32_t a = 0, b = 0, c = 0;
   float value = 0.f;
   bool f64 = false;   // the node's arithmetic type is double (a double literal is among its ancestors)
   double value64 = 0.0;
};

struct Line {
   uint32_t src;     // node whose value is pushed every sample
   uint32_t depth;   // deepest delayed read
   uint32_t row0;    // first state row
   bool f64 = false; // typed programs: the line stores doubles (two float rows per slot)
   uint8_t part = 0; // typed programs: 1 / 2 = the real / imaginary part of a std::complex<float> wire
   bool in_lds;      // ring buffer in LDS instead of registers
   uint32_t lds_slot0 = 0, lds_size = 0;   // ring placement (size is a power of two >= depth)
   // FAR lines (depth > kLdsMaxDepth): the line's state rows ARE a ring buffer in HBM, written
   // every sample and read `n` samples later through the prefetch path; one extra state row holds
   // the ring phase.  Reads with n <= kRegMaxDepth use `shadow` register copies of the newest values.
   bool far = false;
   uint32_t phase_row = 0;
   uint32_t shadow = 0;
};

struct FarRead {
   uint32_t line;   // index into Graph::lines
   uint32_t n;      // delay
};

struct Graph {
   uint32_t n_in = 0, n_out = 0, n_param = 0;   // n_in / n_out: frame SLOTS (floats per frame)
   uint32_t n_mod = 0;               // sample-rate modulators (fz_modulator)
   bool typed = false;               // fz_compile_typed: wire types carried through inputs, state and outputs
   uint32_t ref_divergent = 0;       // != 0: a feedback the SHIPPED reference evaluates against its own arity table (fz_info.differs_from_reference)
   uint32_t sym_tag = 0;             // low 32 bits of graph_structure_hash: the "_g<tag>" of the kernel symbols (kernel_symbol)
   std::vector<uint8_t> in_dtype;    // per input wire: fz_dtype
   std::vector<Node> nodes;          // topological order
   std::vector<uint32_t> outputs;    // node ids, one per output frame slot
   std::vector<uint8_t> out_part;    // per slot: 0 real wire, 1 / 2 = re / im of a complex wire, 3 / 4 = low / high word of a double (typed)
   uint32_t n_out_wires = 0;         // output_arity (complex wires take two slots)
   std::vector<Line> lines;          // ordered by src
   std::vector<float> consts;        // uniform coefficient slots
   std::vector<double> consts64;     // float64 literal terminals
   std::map<uint32_t, uint32_t> uniform_slot;   // fz_uniform id -> coefficient slot (never shared)
   uint32_t n_state = 0, max_delay = 0, n_ops = 0, n_lds_slots = 0;
   std::vector<int> line_of_node;    // node -> line index or -1
   StageSplit split;                 // stage packing, when the graph allows it
   std::vector<FarRead> far_reads;   // distinct (far line, delay) pairs, in first-use order
   std::vector<uint32_t> far_lines;  // indices of far lines
   uint32_t far_min_read = 0;        // smallest delay read from a far line's HBM ring (> kRegMaxDepth); 0: none
   // Wave split (FZ_VF_WAVES(W)): the W parts of a serial graph cut at wires of its stage split, each a graph of its own
   // (1 in, 1 out; constants and state rows are the parent's) that one wave of a W-tuple evaluates.  wave_splits[W] for
   // W = 2, 3, 4; empty when the graph does not allow it.
   std::vector<std::vector<Graph>> wave_splits;
   // (W = 1: the graph itself, when it is stage-packable -- the compute wave next to an I/O wave)
   const std::vector<Graph>* wave_roles(uint32_t W) const { return W && W < wave_splits.size() && wave_splits[W].size() == W ? &wave_splits[W] : nullptr; }
};

// max_atoms: upper bound for K * m (the wave-split hand-offs and the long-run stream-major body bound the total skew)
// force_atoms: cut every segment into atoms even when the chain has three or more packed pairs (a wave that carries ONE pair needs them)
StageSplit find_stage_split(const Graph& g, bool plain = false, uint32_t divisor = 0, uint32_t max_atoms = 13, bool force_atoms = false);
std::vector<Graph> find_wave_roles(const Graph& g, uint32_t W);   // fz_split.cpp; {} or W graphs
// number of waves per stream tuple a variant asks for (flags bits 10..11: 1024 -> 2, 2048 -> 3, 3072 -> 4), 0 = no wave split
inline uint32_t wave_split_of(uint32_t flags) { const uint32_t b = (flags >> 10) & 3u; return b ? b + 1 : 0; }
// FZ_VF_IO_WAVE: one more wave per tuple does the frame I/O.  Parts of the graph a tuple's compute waves evaluate (0: the ordinary
// kernels), and waves per tuple
inline uint32_t ws_io(uint32_t flags) { return (flags & FZ_VF_IO_WAVE) ? 1u : 0u; }
inline uint32_t ws_parts(uint32_t flags) { const uint32_t W = wave_split_of(flags); return W ? W : ws_io(flags); }
inline uint32_t ws_io_waves(uint32_t flags) { return ws_io(flags) ? ((flags & FZ_VF_IO_WAVE2) ? 2u : 1u) : 0u; }   // FZ_VF_IO_WAVE2: loader and storer are two waves
inline uint32_t ws_waves(uint32_t flags) { return ws_parts(flags) + (ws_parts(flags) ? ws_io_waves(flags) : 0u); }

// variant flag bits that mean nothing (any more): rounds 1-5 kept experiment knobs there (plain loads / block order, the SLP vectoriser, cache
// policies, the persistent launch) -- compile-time switches of the kernel source now (FLOWZ_HIP_EXTRA_OPTS=-DFZ_DBG_...).  Refused by check_request.
constexpr uint32_t FZ_VF_RESERVED = 1u | 2u | 4u | (7u << 12) | (7u << 16) | (1u << 27);
// internal variant flag: the stream count is not a multiple of the streams per lane -- the last lane's accesses run past the rows' ends,
// where the per-row buffer descriptors return zeros / drop the writes (frame kernel in lockstep, plain time-major rows)
constexpr uint32_t FZ_VF_RAGGED = 1u << 28;
// internal: output rows of plain time-major frames that do not start on the store grid (kStoreGridBytes): neighbouring waves share the
// sectors / lines at the ends of their footprints, and the frame stores must let L2 merge them (nt instead of nt | sc1; set by
// finalize_variant, profiles/r04/rows_off_the_grid_store_policy.txt)
constexpr uint32_t FZ_VF_ST_MERGE = 1u << 29;
// internal: four streams per lane as TWO PAIRS 128 streams apart (the wave still covers 256 adjacent streams): frames whose lane slice
// would leave in two 16-byte stores -- typed frames of 8 bytes per stream -- then store whole 32-byte sectors per instruction (lanes'
// 16-byte pieces side by side) and can be written through like every other frame (fz_block_kernel.hip.inc: FZ_PAIRS; set by finalize_variant)
constexpr uint32_t FZ_VF_LANE_PAIRS = 1u << 30;
// internal: ... as SINGLE streams 64 apart (two or four streams per lane of frames with four floats per stream: every 16-byte access of a
// lane is one stream's frame, contiguous across the lanes of the wave)
constexpr uint32_t FZ_VF_LANE_SINGLES = 1u << 31;
constexpr uint32_t kChipCUs = 256;       // MI355X (gfx950): 8 XCDs x 32 CUs -- what chip_cus() answers on a box without a GPU
unsigned chip_cus();                     // compute units of the current device (fz_launch.cpp)

// register-resident delay lines up to this depth; deeper ones become LDS rings
constexpr uint32_t kRegMaxDepth = 8;
constexpr uint32_t kLdsMaxDepth = 256;   // deeper lines live in HBM (ring in the state buffer)
constexpr uint32_t kFarMinDelay = 32;    // far reads at least this old allow the full 16-step prefetch chunk (a read must be two chunks old)

struct LowerOptions {
   bool typed = false;               // ResultType semantics for inputs, state and outputs
   std::vector<uint8_t> in_dtype;    // per input wire (typed only); missing entries: float
};
Graph lower(const fz_expr* e, const LowerOptions& opt = LowerOptions());   // throws Error
std::vector<uint32_t> max_input_delays(const fz_expr* e);
uint32_t feedback_promise_inputs(const fz_expr* fb);   // fz_expr.cpp; SURVEY App. C.1

// ---- code generation -------------------------------------------------------------------------------
struct Variant {
   uint32_t P = 2, U = 8, block = 256, flags = 0;
   bool operator<(const Variant& o) const {
      if (P != o.P) return P < o.P;
      if (U != o.U) return U < o.U;
      if (block != o.block) return block < o.block;
      return flags < o.flags;
   }
};

std::string kernel_name(const Graph& g, const Variant& v);     // the variant: fz_block_kernel_p<P>u<U>b<block>...f<flags>
std::string kernel_symbol(const Graph& g, const Variant& v);   // the symbol in the code object: kernel_name + "_g<graph tag>"
// how a frame kernel keeps its LDS rings (fz_codegen.cpp: ring_plan): vectorised in time where the graph allows
struct RingPlan {
   bool vec = false;          // lane-major rows, 16-byte accesses of 4 / P time steps, reads fetched / pushes flushed per sub-chunk
   uint32_t G = 0;            // steps per sub-chunk
   uint32_t TW = 1;           // time steps per 16-byte access
   uint32_t pad = 0;          // padding slots per lane and line
   uint32_t lane_floats = 0;  // floats per lane of all rings
   uint32_t slots = 0;        // V slots per lane the kernel declares (FZ_LDS_SLOTS)
};
RingPlan ring_plan(const Graph& g, const Variant& v);
std::string gen_config(const Graph& g, const Variant& v); // generated "fz_graph_config.h"
std::string gen_body(const Graph& g, const Variant& v);   // generated "fz_graph_body.h"
const std::string& skeleton_source(uint32_t flags);      // hand-written kernel text of a variant: the common head + the one body its flags select
std::string full_source(const Graph& g, const Variant& v);

// ---- runtime ---------------------------------------------------------------------------------------------
// what the code object's metadata says the kernel needs (AMDGPU msgpack notes)
struct KernelResources {
   uint32_t vgprs = 0, agprs = 0, sgprs = 0;
   uint32_t scratch_bytes = 0;      // .private_segment_fixed_size: bytes of scratch memory per lane (register spills)
   uint32_t lds_bytes = 0;
   uint32_t vgpr_spills = 0, sgpr_spills = 0;
};

struct Kernel {
   std::mutex mu;                 // cache lookup / build / module loading of THIS variant (the program mutex is not held meanwhile)
   std::atomic<bool> built{false};
   KernelResources res;
   struct Loaded {
      int device;
      void* module;     // hipModule_t
      void* function;   // hipFunction_t
   };
   std::vector<char> code;        // code object (device independent: gfx950)
   std::string cache_path;        // on-disk cache file it came from / went to ("" = none)
   std::string code_id;           // 16 hex digits: hash of (source, options, the compiler that BUILT this object) = its file name in the cache
   std::vector<Loaded> loaded;    // one module per device the kernel ran on
   void* function_on_current_device(const std::string& symbol);   // loads on first use (caller holds `mu`)
   ~Kernel();                     // unloads the modules (fz_kernel_cache.cpp)
};

}  // namespace fz

namespace fz {
struct SideStream {        // hipStream_t / hipEvent_t, opaque here
   void* stream = nullptr;
   void* fork = nullptr;
   void* join = nullptr;
   std::unique_ptr<std::mutex> mu;   // held from the fork's record to the join's wait: the event pair belongs to one launch at a time
};
}  // namespace fz

struct fz_program {
   fz::Graph g;
   std::mutex mu;
   std::map<fz::Variant, std::shared_ptr<fz::Kernel>> kernels;
   // measured plans (fz_program_tune): (n_streams, tile_streams, device) -> the variant to use when the
   // caller passes none
   std::map<std::tuple<uint64_t, uint32_t, int>, fz_variant> plans;
   std::set<std::tuple<uint64_t, uint32_t, int>> tuned_default;   // shapes measured already (FLOWZ_HIP_AUTOTUNE)
   std::set<std::tuple<uint64_t, uint32_t, int>> measuring;       // shapes whose first big launch is measuring its plan right now (other launches of the shape wait)
   std::condition_variable measured;
   std::set<std::tuple<uint64_t, uint32_t, int>> plan_looked_up;  // shapes whose persisted plan (plans.txt of the kernel cache) was consulted
   std::string recipe;                                             // "typed <0|1> <input dtypes...>\n" + serialize_expr: how to compile this program again (kernel manifests); "" = not recorded
   uint64_t graph_hash = 0;                                        // structure of the lowered graph (no coefficient values)
   // FZ_VF_GRID_SYNC: arrival counters per device: 16 slices of `second` bytes, handed to the launches in turn (launches on
   // different streams may overlap and must not share counters; a slice comes round again after 15 other launches)
   std::map<int, std::pair<void*, size_t>> sync_dev;              // device -> (buffer, bytes per slice)
   std::vector<void*> sync_retired;                                // counter buffers that were outgrown: a captured hipGraph may still use them
   uint32_t sync_next = 0;
   std::map<int, fz::SideStream> side;                             // device -> side stream for remainder launches next to the laps (fz_launch.cpp)
   const float* mod_dev = nullptr;                                 // fz_program_set_modulation
   uint32_t mod_stride = 0;
   ~fz_program();                                                  // frees the counters (fz_launch.cpp)
};

namespace fz {
// the library's choice for a block shape; tile_streams 0 = plain time-major rows (stream-major frames: FZ_VF_STREAM_MAJOR in v->flags)
// allow_lockstep: the most streams per lane the CU-wide lockstep workgroups of plain time-major frames may use (0: not chosen at all)
Variant resolve_variant(const Graph& g, const fz_variant* v, uint64_t n_streams, uint32_t n_samples = 1u << 20, uint32_t tile_streams = 0,
                        uint32_t allow_lockstep = 4);
// plain time-major frames of many streams: streams per lane, lanes per workgroup, laps (one launch each), rows per chunk buffer, and
// the streams those laps cover (the rest: a remainder launch) -- fz_plan.cpp
struct TmGeometry {
   uint32_t P = 0, lanes = 0, laps = 0, U = 0;
   uint64_t main_streams = 0;
   double score = 0.0;
};
TmGeometry time_major_geometry(uint64_t n_streams, uint32_t max_p, bool heavy_ops, bool ragged_ok, uint32_t only_p = 0);
uint64_t lockstep_streams(const Graph& g, const fz_variant* uv, const Variant& v, uint64_t n_streams, uint32_t tile_streams);
// the kernel a launch of that shape runs: resolved, fitted to the tile / the 4 GiB chunk limit, unroll lowered until nothing spills
Variant finalize_variant(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams, bool settle = true);
// the second kernel of a launch whose lockstep laps leave `rem` streams to a remainder launch (lockstep_streams(...) < n_streams)
Variant remainder_variant(fz_program* p, const fz_variant* uv, uint64_t n_streams, uint32_t n_samples, uint64_t rem);
// builds (or fetches from the caches) the kernel of variant v; fn_out != null: also load it on the
// current device and return its hipFunction_t
std::shared_ptr<Kernel> get_kernel(fz_program* p, const Variant& v, void** fn_out);
// the variant that runs: `v` with its unroll lowered until the kernel keeps its values in registers (no scratch memory)
Variant settle_variant(fz_program* p, Variant v);
int launch(fz_program* p, const float* in, float* out, float* state, const float* params,
           uint64_t n_streams, uint32_t n_samples, const fz_variant* v, void* stream, uint32_t tile_streams = 0,
           uint32_t rows_total = 0, uint32_t row0 = 0, uint32_t mod_row0 = 0);   // mod_row0: row of the modulator arrays that goes with row 0 of the frame buffers
int tune(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
         uint32_t n_samples, uint32_t tile_streams, void* stream, fz_variant* chosen, float* chosen_ms, bool implicit = false);
std::vector<fz_variant> tune_candidates(const Graph& g, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams);
int device_count();
uint64_t graph_structure_hash(const Graph& g);
// the plan a launch without a variant would use for this shape on the current device: in memory, else persisted, else {0,0,0,0}
fz_variant planned_variant(fz_program* p, uint64_t n_streams, uint32_t tile_streams);
void drop_plan(fz_program* p, uint64_t n_streams, uint32_t tile_streams);
// while one of these lives on a thread, get_kernel on that thread refuses to BUILD (FZ_E_UNSUPPORTED): cached objects only
extern thread_local bool tl_no_jit;
struct NoJitScope {
   bool before;
   explicit NoJitScope(bool on) : before(tl_no_jit) { tl_no_jit = on || before; }
   ~NoJitScope() { tl_no_jit = before; }
};
// is the kernel's code object at hand (in memory or in the on-disk cache), i.e. can it run without a hiprtc build?
bool kernel_at_hand(fz_program* p, const Variant& v);
std::string kernel_code_id(fz_program* p, const Variant& v);
int manifest_build(const std::string& path, unsigned n_workers, uint32_t counts[4]);
}  // namespace fz
