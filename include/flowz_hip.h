/* flowz_hip.h -- C ABI of libflowz_hip.so: MI355X (gfx950) evaluator for Flowz flow-graphs.
 *
 * This is the drop-in boundary of the hot path.  The reference has no FFI: its path sits
 * behind a C++ header API (/root/reference/flowz/flowz.hpp).  Each group below names the
 * reference interface it replaces; the C++ front end (include/flowz/flowz.hpp) and the
 * Python mirror (zignal_amd/flowz.py) are thin layers over exactly these entry points.
 *
 * Conventions: plain pointers and sizes, no exceptions cross the boundary.  Functions
 * returning int give FZ_OK (0) or a negative fz_status; fz_last_error() returns a
 * thread-local message for the last failure on the calling thread.
 * All arithmetic is IEEE float32, one rounding per expression node in the user's
 * association order, no FMA contraction, denormals kept (flowz.hpp:769-772,
 * CMakeLists.txt:18).
 */
#ifndef FLOWZ_HIP_H
#define FLOWZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fz_status {
   FZ_OK            =  0,
   FZ_E_INVALID     = -1,   /* bad argument (null, misaligned, size mismatch)              */
   FZ_E_GRAPH       = -2,   /* malformed flow-graph (arity, delay-free loop, ...)          */
   FZ_E_NO_DEVICE   = -3,   /* no HIP device: the product path has no CPU fallback         */
   FZ_E_HIP         = -4,   /* HIP runtime error                                           */
   FZ_E_COMPILE     = -5,   /* hiprtc failed to build the generated kernel                 */
   FZ_E_UNSUPPORTED = -6    /* valid graph, beyond this build (e.g. delay too long)        */
} fz_status;

const char* fz_last_error(void);
const char* fz_version(void);

/* ------------------------------------------------------------------------------------------
 * Expression construction  == the EDSL surface, flowz.hpp:68-93 and :1252-1257.
 * Handles are immutable, reference counted trees; every constructor returns a NEW handle
 * (refcount 1) and retains its operands, so the caller releases what it created
 * (value semantics of proto's copy_domain, flowz.hpp:46-61).  NULL on error.
 * ---------------------------------------------------------------------------------------- */
typedef struct fz_expr fz_expr;

typedef enum fz_op { FZ_OP_ADD = 1, FZ_OP_SUB = 2, FZ_OP_MUL = 3, FZ_OP_DIV = 4, FZ_OP_NEG = 5,
                     /* the comparison and logical operators of C++ (proto::_default applies whatever operator a node is, flowz.hpp:51-55,
                        :769-772): the result is the operator's bool as it behaves in arithmetic -- 1 or 0, taking the type of what it
                        meets next (bool * float is a float multiplication, bool * double a double one); an output frame or a delay
                        line receives it as 1.0f / 0.0f.  Operands are compared in their common type (double if one is), IEEE semantics
                        (every comparison with a NaN is false, != true).  Not for std::complex wires.  a && b, a || b, !a test their
                        operands against zero as C++ does for arithmetic types; both sides are always evaluated (no side effects to skip). */
                     FZ_OP_LT = 6, FZ_OP_LE = 7, FZ_OP_GT = 8, FZ_OP_GE = 9, FZ_OP_EQ = 10, FZ_OP_NE = 11,
                     FZ_OP_NOT = 12, FZ_OP_AND = 13, FZ_OP_OR = 14 } fz_op;

fz_expr* fz_placeholder(uint32_t i);                 /* _i          make_placeholder<i>() :78-82   */
fz_expr* fz_delayed(uint32_t i, uint32_t n);         /* _i[_n]      delayed_placeholder   :84-85   */
fz_expr* fz_literal(float value);                    /* terminal held by value  make_terminal :68-72 */
fz_expr* fz_literal_f64(double value);               /* a C++ `double` literal terminal: the operators above
                                                        it evaluate in double (usual arithmetic conversions,
                                                        proto::_default :769-772; test/tests.cpp:200-231),
                                                        delay lines and output frames stay float32 (:1245) */
fz_expr* fz_literal_c32(float re, float im);         /* a std::complex<float> terminal (test/tests.cpp:206-207):
                                                        the wire above it is complex -- two float32 slots
                                                        (re, im) of the output frame; operators follow
                                                        std::complex<float> (scalar mul/div and add touch the
                                                        parts as <complex> does, complex*complex is the
                                                        (ac-bd, ad+bc) of __mulsc3 for finite values; z/w and
                                                        s/w are libgcc's __divsc3 as g++ links it: the four
                                                        parts widened to double, x = (ac+bd)/(cc+dd),
                                                        y = (bc-ad)/(cc+dd), rounded to float once).  Under
                                                        fz_compile a complex wire cannot enter a delay line
                                                        (they are float, :1245; fz_compile_typed stores it)
                                                        nor meet a double operand (no such operator in C++):
                                                        FZ_E_GRAPH                                        */
fz_expr* fz_literal_c64(double re, double im);       /* a std::complex<double> terminal: as fz_literal_c32 with double parts.  Operators
                                                        follow std::complex<double>: z*w is the (ac-bd, ad+bc) of __muldc3, z/w and
                                                        s/w are libgcc's __divdc3 = Smith's method (|c| < |d| ? ratio c/d : ratio d/c;
                                                        both sides are evaluated and selected: FZ_IR_ABSLT / FZ_IR_SELECT); it mixes
                                                        with double scalars only (as in C++: no operator for complex<double> with
                                                        float or complex<float>)                                              */
fz_expr* fz_stream_param(uint32_t k);                /* per-stream, block-constant coefficient k:
                                                        the std::ref terminal of flowz/README.md:42-61,
                                                        one value per stream                          */
fz_expr* fz_uniform(uint32_t k, float initial);      /* uniform run-time coefficient k (same value for
                                                        all streams, constant during a block): what a
                                                        std::ref(x) terminal is when the closure is called
                                                        (flowz/README.md:42-61); set with
                                                        fz_program_set_uniform between blocks             */
fz_expr* fz_modulator(uint32_t k);                   /* the std::ref(x) terminal at SAMPLE rate (flowz/README.md:42-61: the
                                                        reference re-reads the referenced variable on every call, i.e. every
                                                        sample): modulator k has one value per sample of a block, the same
                                                        for all streams, read from the array fz_program_set_modulation names --
                                                        an input wire without the per-stream HBM traffic (scalar loads)       */
fz_expr* fz_arith(fz_op op, fz_expr* a, fz_expr* b); /* any C++ arithmetic, comparison or logical operator,
                                                        _default :769-772; b is ignored (may be NULL) for
                                                        FZ_OP_NEG and FZ_OP_NOT                       */
fz_expr* fz_channel (fz_expr* a, fz_expr* b);        /* a , b       channel_operator   :90           */
fz_expr* fz_parallel(fz_expr* a, fz_expr* b);        /* a | b       parallel_operator  :91           */
fz_expr* fz_sequence(fz_expr* a, fz_expr* b);        /* a |= b      sequence_operator  :92           */
fz_expr* fz_feedback(fz_expr* a);                    /* ~a          feedback_operator  :93           */
void     fz_expr_retain (fz_expr* e);
void     fz_expr_release(fz_expr* e);

/* Static analysis transforms (flowz.hpp:162-246, :443-506; asserted by test/tests.cpp:63-102). */
int fz_input_arity (const fz_expr* e);               /* >= 0, or negative fz_status                   */
int fz_output_arity(const fz_expr* e);
/* per external input wire, deepest delayed read; writes min(n, cap) entries, returns n        */
int fz_max_input_delays(const fz_expr* e, uint32_t* out, uint32_t cap);

/* ------------------------------------------------------------------------------------------
 * compile()  == flowz::compile, flowz.hpp:1233-1249: arity, front panel, feedback
 * resolution, state layout -- at run time instead of C++ template instantiation.
 * Pure host work: succeeds without a GPU.
 * ---------------------------------------------------------------------------------------- */
typedef struct fz_program fz_program;

typedef struct fz_info {
   uint32_t n_in;        /* external input wires  (frame width of `in`)                       */
   uint32_t n_out;       /* output wires          (frame width of `out`)                      */
   uint32_t n_nodes;     /* nodes of the lowered per-sample DAG                               */
   uint32_t n_ops;       /* arithmetic nodes = float32 operations per stream-sample           */
   uint32_t n_lines;     /* delay lines (one per delayed wire, shared by all its readers)     */
   uint32_t n_state;     /* floats of state per stream = sum of line depths                   */
   uint32_t n_const;     /* distinct uniform float32 coefficients (literal terminals)         */
   uint32_t n_param;     /* per-stream coefficients (highest fz_stream_param index + 1)       */
   uint32_t max_delay;   /* deepest delay line                                                */
   uint32_t n_lds_slots; /* ring-buffer slots kept in LDS (lines deeper than the register cap) */
   uint32_t stage_packable; /* 1 when the graph is a series of isomorphic segments (FZ_VF_STAGE_PACK) */
   uint32_t n_const64;   /* distinct float64 literal terminals                                */
   uint32_t n_out_wires; /* output wires (output_arity); < n_out when some wires are complex         */
   uint32_t n_in_wires;  /* input wires (input_arity); < n_in when a typed program has double / complex inputs */
   uint32_t typed;       /* 1 for fz_compile_typed programs                                           */
   uint32_t n_mod;       /* sample-rate modulators (highest fz_modulator index + 1)                            */
   uint32_t differs_from_reference; /* != 0: the graph holds a feedback ~(a |= b) whose first part keeps that many EXTERNAL inputs for itself
                            while the part behind it reads external inputs too (the graphs of test/tests.cpp:67-77).  The reference's
                            shipped binary_feedback hands that second part the wrong wires (flowz.hpp:1045-1050: tuple_drop<std::min(0, ..)>,
                            "TODO" there); this library routes per the reference's arity table (:162-246) -- e.g.
                            ~(_1 + _2[_1] |= _1[_1] + _2) on (10,1),(20,2),(30,3) gives 1, 3, 16 here and 10, 30, 70 from the shipped header.
                            fz_compile succeeds and leaves a note in fz_last_error()                                          */
} fz_info;

int  fz_compile(const fz_expr* e, fz_program** out);

/* compile() with the wire types of the reference's ResultType transform (flowz.hpp:585-644, asserted by
 * test/tests.cpp:184-232) carried through INPUTS, STATE and OUTPUTS instead of the float state that compile()
 * hard-codes today (flowz.hpp:1245, "TODO" there):
 *   - input wire i arrives as in_dtypes[i] (fz_dtype; NULL or n_in_wires == 0: all float) -- the reference's callable
 *     is a template over its argument types (flowz.hpp:1225-1229), so f(1.0) or f(std::complex<float>{..}) are legal;
 *   - a delay line stores the type that is pushed into it and a delayed read returns that type (tests.cpp:219);
 *     the type of a fed-back wire is what the "absorber" rule of ResultType gives: the least type that is consistent
 *     around the loop, e.g. ~(_1[_1] + 1.0*_2) is double (tests.cpp:224-226); a loop that never meets another type
 *     stays float;
 *   - frames carry every wire in its own type: float = 1 float slot, double = 2 slots (low word, high word: the
 *     frame is a double[] there), std::complex<float> = 2 slots (re, im).  n_in / n_out of fz_info count SLOTS,
 *     n_in_wires / n_out_wires count wires.  FZ_VF_OUT_F64 does not apply (rejected).
 * State rows: double lines come first and take two float rows per delay slot (one row of n_streams doubles);
 * a complex wire has one float line for each part.  Double lines: registers up to 8 samples, LDS rings up to 256.
 * std::complex<double> wires (fz_literal_c64, FZ_DT_CF64 inputs) take four slots: the double of the real part, then
 * the double of the imaginary part; their delay lines are two double lines.                                         */
typedef enum fz_dtype { FZ_DT_F32 = 0, FZ_DT_F64 = 1, FZ_DT_CF32 = 2, FZ_DT_CF64 = 3 } fz_dtype;   /* CF64: 4 slots (re, im doubles) */
int  fz_compile_typed(const fz_expr* e, const uint32_t* in_dtypes, uint32_t n_in_wires, fz_program** out);
/* type of every input wire (fz_dtype); writes min(n, cap), returns n = n_in_wires */
int  fz_program_input_dtypes(const fz_program* p, uint32_t* dtypes, uint32_t cap);
/* storage type of every delay line, in fz_program_lines order: 0 float, 1 double (two state rows per slot),
 * 2 / 3 the real / imaginary part of a std::complex<float> wire (float rows), 4 / 5 the real / imaginary part of a
 * std::complex<double> wire (double lines: two state rows per slot); writes min(n, cap), returns n */
int  fz_program_line_dtypes(const fz_program* p, uint32_t* dtypes, uint32_t cap);
void fz_program_destroy(fz_program* p);
int  fz_program_info(const fz_program* p, fz_info* info);

/* Lowered IR, for inspection and for tests (the product never interprets it on the CPU). */
typedef enum fz_ir_kind {
   FZ_IR_INPUT = 1,   /* a = input wire index                                                  */
   FZ_IR_CONST = 2,   /* a = coefficient slot, value = its float                               */
   FZ_IR_PARAM = 3,   /* a = per-stream coefficient index                                      */
   FZ_IR_DELAY = 4,   /* a = source node, b = n : value of node a, n samples ago               */
   FZ_IR_ADD = 5, FZ_IR_SUB = 6, FZ_IR_MUL = 7, FZ_IR_DIV = 8,   /* a (op) b                    */
   FZ_IR_NEG = 9,     /* -a                                                                    */
   FZ_IR_WIDEN = 10,  /* (double)a : float -> double, exact                                    */
   FZ_IR_NARROW = 11, /* (float)a  : double -> float, one IEEE rounding (both only appear where C++ itself converts
                         inside an operator: the float complex division of libgcc's __divsc3, see fz_arith)           */
   FZ_IR_MOD = 12,    /* a = modulator index: value of sample-rate modulator a at this sample (fz_modulator)          */
   FZ_IR_ABSLT = 13,  /* |a| < |b| ? 1 : 0  (in the operands' type)                                                    */
   FZ_IR_SELECT = 14, /* a != 0 ? b : c   (the data-dependent branch of __divdc3; both sides are evaluated)            */
   FZ_IR_LT = 15, FZ_IR_LE = 16, FZ_IR_GT = 17, FZ_IR_GE = 18, FZ_IR_EQ = 19, FZ_IR_NE = 20
                      /* a (cmp) b ? 1.0f : 0.0f -- a float32 node (dtype 0) whose operands are compared in double when one of them is     */
} fz_ir_kind;

typedef struct fz_ir_node {
   uint32_t kind, a, b;
   float value;        /* FZ_IR_CONST, dtype 0 */
   uint32_t dtype;     /* 0 = float32, 1 = float64 (the node's C++ arithmetic type) */
   double value64;     /* FZ_IR_CONST, dtype 1 */
   uint32_t c;         /* third operand (FZ_IR_SELECT) */
} fz_ir_node;

/* nodes are in evaluation (topological) order; writes min(n, cap), returns n */
int fz_program_ir(const fz_program* p, fz_ir_node* nodes, uint32_t cap);
/* node id of each output frame slot (a complex wire takes two: re, im); writes min(n_out, cap), returns n_out */
int fz_program_outputs(const fz_program* p, uint32_t* node_ids, uint32_t cap);
/* arithmetic type of each output frame slot before it is narrowed to the float32 frame: 0 = float, 1 = double,
 * 2 / 3 = real / imaginary part of a std::complex<float> wire; fz_compile_typed programs: 4 / 5 = low / high word
 * of a double wire (never 1: nothing is narrowed), 6 / 7 / 8 / 9 = low / high word of the real, low / high word of
 * the imaginary part of a std::complex<double> wire; fz_compile programs: 10 / 11 = real / imaginary part of a
 * std::complex<double> wire (narrowed to the float frame like code 1)
 * (the ResultType inference of flowz.hpp:585-644 / test/tests.cpp:200-231, with compile()'s float delay
 * lines: a delayed read is float whatever was pushed, flowz.hpp:1245); writes min(n_out, cap), returns n_out */
int fz_program_output_dtypes(const fz_program* p, uint32_t* dtypes, uint32_t cap);
/* delay lines: source node and depth of line l; state rows of line l start at the sum of the
 * depths before it, row (start + j) holds the wire's value at t-1-j (j = 0 newest).
 * Lines deeper than 256 samples are rings in HBM instead: their `depth` rows are ring slots, one
 * extra state row per such line (after all line rows) holds the ring phase p (as a float), and the
 * value at t-1-j sits in slot (p - 1 - j) mod depth.  n_state counts those phase rows.          */
int fz_program_lines(const fz_program* p, uint32_t* src_nodes, uint32_t* depths, uint32_t cap);
/* read / overwrite a uniform coefficient (literal terminal) between blocks */
int fz_program_get_const(const fz_program* p, uint32_t slot, float* value);
int fz_program_set_const(fz_program* p, uint32_t slot, float value);
/* overwrite uniform run-time coefficient k (fz_uniform) between blocks */
int fz_program_set_uniform(fz_program* p, uint32_t k, float value);
/* Sample-rate modulators (fz_modulator): mod_dev is a DEVICE array [n_mod][stride] of floats, stride >= the rows the frame
 * buffers of the following launches hold; sample t of a block (row row0 + t of a window) reads modulator k at
 * mod_dev[k * stride + row0 + t].  The pointer is remembered by the program until it is set again (like fz_program_set_uniform:
 * set it before the launch that needs it; a launch of a graph with modulators and no array fails with FZ_E_INVALID).
 * Graphs with modulators are not stage-packed (their segments run at different times).                                   */
int fz_program_set_modulation(fz_program* p, const float* mod_dev, uint32_t stride);

/* ------------------------------------------------------------------------------------------
 * Kernel variants.  One fused HIP kernel per (graph, variant) is generated and built with
 * hiprtc for gfx950 (building needs no GPU; code objects are cached on disk).
 * ---------------------------------------------------------------------------------------- */
typedef struct fz_variant {
   uint32_t streams_per_lane;  /* 1, 2 (v_pk_* float2) or 4; 0 = choose from n_streams         */
   uint32_t unroll;            /* time steps per prefetch chunk (1..32); 0 = default           */
   uint32_t block_threads;     /* 64..1024, multiple of 64; 0 = default (256)                  */
   uint32_t flags;             /* FZ_VF_* ; 0 = default                                        */
} fz_variant;

enum { FZ_VF_STAGE_PACK = 8u,   /* one stream per lane; the K isomorphic segments of a serial graph (e.g. the
                                   stages of a cascade, after an optional scalar prefix) run skewed in time,
                                   segment j at t-j, and segments i, i+K/2 share one v_pk_* per node; chosen
                                   automatically below 2^18 streams when fz_info.stage_packable          */
       FZ_VF_NO_STAGE_PACK = 16u,
       FZ_VF_PREFETCH3 = 32u,   /* three input chunk buffers: loads run two chunks (2 x unroll steps) ahead
                                   (not with delay lines beyond 256 samples)                              */
       FZ_VF_STREAM_MAJOR = 128u,   /* set by fz_run_block_stream_major (the frame layout is part of the kernel) */
       FZ_VF_SM_LONG = 256u,    /* stream-major frames, 1-in/1-out graphs: the long-run body -- 512-byte runs per stream (unroll 128;
                                   64 selectable), one in-place LDS patch per wave, one wave per SIMD; chosen automatically for
                                   blocks of >= 256 samples; FZ_VF_SM_SHORT keeps the 32-sample chunks.  With streams_per_lane = 2
                                   (unroll 64): the PAIR body -- two streams per lane, every node one packed instruction, halves of
                                   64 samples, 256-byte in-runs and 512-byte out-runs; the default for deep graphs with uniform
                                   coefficients from 2^19 (even) streams on                                              */
       FZ_VF_SM_SHORT = 512u,
       FZ_VF_WAVE_SPLIT = 1024u, /* fewer streams than lanes: a serial graph of K isomorphic segments is cut into W parts of K / W
                                   segments, W waves of a workgroup evaluate the parts for the same 64 streams (the cut wires
                                   travel through LDS, every wave one chunk behind the one before); one stream per lane,
                                   block_threads counts the streams of a workgroup (a multiple of 64).  This bit: W = 2;
                                   FZ_VF_WAVES(3), FZ_VF_WAVES(4): three / four parts.  Chosen automatically for few streams */
       FZ_VF_WAVE_SPLIT3 = 2048u,
       FZ_VF_IO_WAVE = 32768u,  /* one more wave per 64 streams does all the frame I/O: it loads the input rows two rounds ahead and
                                   hands them to the compute wave(s) through LDS, and stores the rows they hand back; the compute
                                   wave is left with arithmetic and LDS accesses.  Alone (a stage-packable graph: one compute wave
                                   + one I/O wave, 4 such pairs per workgroup) or together with FZ_VF_WAVES(n)                   */
       FZ_VF_LOCKSTEP = 524288u, /* time-major / tiled frames: the waves of a workgroup meet at a barrier after every chunk and so walk the
                                   same rows at the same time -- with plain time-major frames of many streams (rows megabytes apart)
                                   that keeps the pages a CU has in flight few; chosen automatically there                        */
       FZ_VF_GRID_SYNC = 8388608u, /* with FZ_VF_LOCKSTEP: the workgroups of one XCD (one contiguous 1/8 of every row) also walk the rows together:
                                   arrival counters in device memory (zeroed in stream order before the launch), bounded waits -- never a
                                   hang, never a different bit; chosen automatically when the chip holds all workgroups at once      */
       FZ_VF_IO_WAVE2 = 33554432u, /* with FZ_VF_IO_WAVE: TWO I/O waves per tuple -- one loads the input rows, one stores the output rows (chosen automatically for
                                   stage-packable graphs of <= 64 operations between 32 768 and 65 536 streams: one compute wave per SIMD next to them).  A wave
                                   issues its vector-memory instructions in order, one row of 64 streams x 4 bytes each: at one I/O wave per
                                   tuple that wave's 2 x n_samples instructions are what a round waits for (profiles/r04/few_streams_floor.txt) */
       FZ_VF_OUT_F64 = 64u };   /* `out` holds float64 frames [..][n_out] of doubles (pass the double* cast to
                                   float*): the results of graphs with double literals leave un-narrowed, float
                                   wires are widened exactly (tuple<double> results, test/tests.cpp:201-231)   */
/* bits 10..11 of flags: wave split into n = 2, 3 or 4 parts (see FZ_VF_WAVE_SPLIT) */
#define FZ_VF_WAVES(n) ((n) >= 2 && (n) <= 4 ? ((uint32_t)((n) - 1) << 10) : 0u)
/* bits 20..22 of flags: at most n workgroups per CU (the kernel pads its LDS); 0 = as many as fit.
 * Fewer, fatter waves keep fewer frame tiles in flight: which occupancy streams fastest from HBM depends
 * on the board -- let fz_program_tune measure it                                                   */
#define FZ_VF_MAX_WG(n) (((uint32_t)(n) & 7u) << 20)
/* bits 0..2, 12..14, 16..18 and 27 are reserved (FZ_E_INVALID): rounds 1-5 had experiment knobs there (cache policies, the SLP vectoriser, the
 * plain block order); those are compile-time switches of the kernel source now (FLOWZ_HIP_EXTRA_OPTS) */

int fz_program_build(fz_program* p, const fz_variant* v);           /* JIT (or cache hit) only   */
/* the same with the variant's automatic fields resolved as a launch of this block shape would: the shape of a launch is
 * (n_streams, n_samples, tile_streams) -- tile_streams as for fz_run_block_tiled, 0 = plain time-major rows (the library's choice
 * depends on the layout: see FZ_VF_LOCKSTEP); stream-major frames are named by FZ_VF_STREAM_MAJOR in v->flags            */
int fz_program_build_for(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams);
/* Part k of the wave split into n_parts (FZ_VF_WAVES(n_parts); n_parts = 1: the graph itself next to an I/O wave) as a
 * program of its own, for inspection (fz_program_ir, fz_program_lines, fz_program_info): input = the cut wire before the
 * part (the graph input for k = 0), output = the cut wire behind it; constant slots are the parent's.  FZ_E_UNSUPPORTED when
 * the graph does not split that way.  Destroy with fz_program_destroy.                                                */
int fz_program_wave_part(const fz_program* p, uint32_t n_parts, uint32_t k, fz_program** out);
/* Registers, LDS and scratch memory of a variant's kernel, from the code object's metadata (JITs it; no device needed).
 * The unroll of a variant is an UPPER bound for time-major / tiled frames: a kernel whose prefetch buffers and delay lines
 * do not fit the register file would keep some of them in scratch memory, so fz_run_block halves the unroll until nothing
 * spills (stream-major frames keep theirs: it is also the length of a stream's run in memory).  as_launched = 1: the kernel that
 * fz_run_block launches (`unroll` = what is left of the variant's); 0: the variant exactly as given.                   */
typedef struct fz_kernel_resources {
   uint32_t vgprs, agprs, sgprs;
   uint32_t scratch_bytes;      /* per lane; 0 = nothing spills */
   uint32_t lds_bytes;          /* static LDS of a workgroup */
   uint32_t vgpr_spills, sgpr_spills;
   uint32_t unroll;
} fz_kernel_resources;
int fz_program_kernel_resources(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                                int as_launched, fz_kernel_resources* out);
/* name of the variant's kernel, e.g. "fz_block_kernel_p2u16b256f2097152" (streams per lane, rows per chunk, lanes per workgroup,
 * flags): exactly the kernel a launch of the shape (n_streams, n_samples, tile_streams) runs -- one resolution shared with the
 * launch path; returns length.  (A plain time-major block whose laps leave a few streams over runs a SECOND kernel next to them, on those
 * streams: name, symbol, code id and resources describe the laps' kernel; fz_program_build_for builds both.) */
long fz_program_kernel_name(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                            char* buf, size_t cap);
/* ... and the SYMBOL of that kernel as profilers show it (rocprofv3 --kernel-trace --stats): the name + "_g<8 hex digits>", a tag of
 * the graph's structure -- two graphs that run the same variant are different rows of a profile (graphs that differ only in
 * coefficient values share the symbol and the code object) */
long fz_program_kernel_symbol(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                              char* buf, size_t cap);
/* ... and the identity of that kernel's CODE: 16 hex digits, a hash of (generated source, build options, compiler identity) -- the file name of
 * its code object in the kernel cache.  The symbol names variant and graph; this names the instructions: counters measured on one build
 * of a kernel are not this run's when the id differs (profiles/pmc_traffic.json is keyed by it). */
long fz_program_kernel_code_id(fz_program* p, const fz_variant* v, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                               char* buf, size_t cap);
/* An expression as text (one line per node of the DAG, values as bit patterns) and back: what a kernel manifest records of a program.
 * fz_expr_recipe returns the length and writes <= cap bytes; fz_expr_from_recipe returns a new reference, NULL (+ fz_last_error) for
 * text that is not a recipe. */
long fz_expr_recipe(const fz_expr* e, char* buf, size_t cap);
fz_expr* fz_expr_from_recipe(const char* text);
/* Kernel manifests.  With FLOWZ_HIP_MANIFEST=<file> in the environment every kernel a process resolves for the first time is appended to
 * <file> as (the program's expression, input types, variant).  fz_manifest_build replays such a file: compiles the programs again and
 * builds -- in n_workers parallel compiler processes, no GPU needed -- whatever the kernel cache does not hold yet.  The records name
 * expressions and variants, not generated code: a replay after the library changed builds the new kernels of the same launches.
 * counts[4] = {records, already in the cache, built now, failed (a graph or variant this build no longer accepts)}. */
int fz_manifest_build(const char* path, uint32_t n_workers, uint32_t* counts);
/* generated HIP source of a variant (skeleton + graph body); returns length, writes <= cap    */
long fz_program_source(fz_program* p, const fz_variant* v, char* buf, size_t cap);

/* ------------------------------------------------------------------------------------------
 * fz_run_block -- the hot path: stateful_lambda::operator() (flowz.hpp:1225-1229) applied to
 * n_samples consecutive samples of n_streams independent closures in ONE kernel launch
 * (the caller's per-sample loop, test/benchmark.cpp:137-147, moves into the kernel).
 *
 * All pointers are DEVICE pointers, 16-byte aligned, owned by the caller:
 *   in     [n_samples][n_streams][n_in]   time-major interleaved frames (NULL iff n_in == 0)
 *   out    [n_samples][n_streams][n_out]
 *   state  [n_state][n_streams]  in/out; zero it before the first block (flowz.hpp:1245);
 *          carries the closure state from block to block (may be NULL iff n_state == 0)
 *   params [n_param][n_streams]  (NULL iff n_param == 0)
 * n_samples == 0 or n_streams == 0 is an empty block: FZ_OK, nothing is touched.
 * Asynchronous on `hip_stream` (hipStream_t, NULL = default stream); the caller synchronises.
 * `v` may be NULL (all defaults).  A program may run concurrently on different state buffers.
 *
 * WHICH KERNEL RUNS (v == NULL): the plan fz_program_tune measured for this (n_streams, tile_streams, device) -- in this process or,
 * persisted, in an earlier one --, else the library's static choice.  A launch never measures anything by itself.
 * Opt-in, FLOWZ_HIP_AUTOTUNE=1 in the environment (rounds 3-5 did this by default): the first big launch of a shape (the block is the
 * whole buffer, n_streams * n_samples >= 2^26) makes fz_program_tune's measurement on the caller's buffers before it runs:
 *   - it synchronises `hip_stream` and takes the time of a few dozen blocks (>= 100 ms of warm-up, every candidate timed twice);
 *     a candidate replaces the library's static choice only when it wins by more than 3 % (fz_program_tune: 1.5 %);
 *   - it allocates a copy of `state` (n_state * n_streams floats), runs the candidates on the caller's in / out / state buffers and
 *     puts the state back; a state that cannot be put back is FZ_E_HIP (the message says so), never a silent advance; without room
 *     for the copy nothing is measured;
 *   - only candidates whose code objects are already built (in memory or in the kernel cache) take part, nothing is compiled for
 *     it; a candidate that fails -- a HIP error included -- is skipped, and if the measurement itself fails the library's static
 *     choice runs: the launch fails only where a plain launch would;
 *   - it is skipped while `hip_stream` is being captured into a hipGraph, when `in` and `out` overlap, and for windows;
 *   - other launches of the same shape on this program wait until the plan is known (they then use it).
 * ---------------------------------------------------------------------------------------- */
int fz_run_block(fz_program* p, const float* in, float* out, float* state, const float* params,
                 uint64_t n_streams, uint32_t n_samples, const fz_variant* v, void* hip_stream);

/* Same hot path with STREAM-TILED frames, the layout recommended for HBM3E on MI355X:
 *   in   [n_streams / tile_streams][n_samples][tile_streams][n_in]
 *   out  [n_streams / tile_streams][n_samples][tile_streams][n_out]
 * i.e. every tile of tile_streams adjacent streams is its own time-major block.  32 KiB row
 * segments (tile_streams * n_in * 4 bytes; 8192 streams for one wire) measured +15...20 % over
 * 4 MiB rows (profiles/r01/hbm_copy_patterns_microbench.txt).  n_streams must be a multiple of
 * tile_streams and tile_streams a multiple of streams_per_lane * block_threads;
 * tile_streams == 0 or == n_streams is fz_run_block.  state / params stay [rows][n_streams]. */
int fz_run_block_tiled(fz_program* p, const float* in, float* out, float* state, const float* params,
                       uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams,
                       const fz_variant* v, void* hip_stream);

/* tile_streams that makes the row segments ~32 KiB for this graph's frame widths (power of two) */
/* A block that is a WINDOW in time of larger frame buffers: samples [row0, row0 + n_samples) of buffers
 * holding rows_total samples (per tile when tile_streams != 0, else time-major).  Lets a long stream-tiled
 * recording be processed in pieces -- e.g. one block per control period with new coefficients -- without
 * re-laying it out.                                                                                     */
int  fz_run_block_window(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                         uint32_t rows_total, uint32_t row0, uint32_t n_samples, uint32_t tile_streams, const fz_variant* v,
                         void* hip_stream);
/* The block kernel on STREAM-MAJOR buffers, without a layout pass: in [n_streams][rows_total][n_in],
 * out [n_streams][rows_total][n_out] -- one contiguous buffer per stream, as every closure of the reference
 * consumes its samples (test/benchmark.cpp:137-147); the block is the window [row0, row0 + n_samples).
 * Waves fetch [64 streams][unroll samples] patches along the rows and transpose them through LDS.  One
 * stream per lane (no lane or stage packing: VALU-bound for deep graphs at large stream counts), no delay
 * lines beyond 256 samples, float32 frames; rows_total * n_in, row0 * n_in (and the same for n_out) must
 * be multiples of 4 floats.  For the fastest path convert once with fz_transpose_frames instead.      */
int  fz_run_block_stream_major(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                               uint32_t rows_total, uint32_t row0, uint32_t n_samples, const fz_variant* v, void* hip_stream);
uint32_t fz_recommended_tile_streams(const fz_program* p);

/* Plan selection (what FFTW_MEASURE is to FFTW): time the candidate kernel variants of this program
 * for this shape on the caller's own device buffers and remember the fastest; later fz_run_block /
 * fz_run_block_tiled / fz_bank_process* calls with the same (n_streams, tile_streams) on this device and
 * variant == NULL use it.  All variants compute bit-identical results.  The buffers are used as by
 * fz_run_block_tiled (tile_streams 0: time-major) for a few dozen blocks: `out` is overwritten and
 * `state` advances -- reset it afterwards.  chosen / chosen_ms (may be NULL): the winner and its time
 * per block.  Synchronises hip_stream.  (The boards are power-managed -- a kernel at the package power
 * cap runs its first ~100 ms faster than it sustains --, so the default is run for >= 100 ms first and
 * every candidate is then timed twice, in a forward and a backward pass over the list.)            */
int fz_program_tune(fz_program* p, const float* in, float* out, float* state, const float* params, uint64_t n_streams,
                    uint32_t n_samples, uint32_t tile_streams, void* hip_stream, fz_variant* chosen, float* chosen_ms);

/* The winner is also PERSISTED: a line in <kernel cache dir>/plans.txt keyed by (graph structure -- coefficient values do
 * not matter --, n_streams, tile_streams, the board's UUID); the first launch of that shape without a variant in a later
 * process picks it up, so callers that tuned once never sit on the library default again.  FLOWZ_HIP_NO_PLAN_CACHE=1
 * disables reading and writing.  fz_program_plan: the variant such a launch would use now on the current device
 * ({0,0,0,0} = library default).                                                                                  */
int fz_program_plan(fz_program* p, uint64_t n_streams, uint32_t tile_streams, fz_variant* out);

/* the variants fz_program_tune would measure for this shape (the first entry is the library default {0,0,0,0});
 * writes min(n, cap) entries, returns n.  Pure host work: lets a build step pre-compile them (fz_program_build). */
int fz_program_tune_candidates(fz_program* p, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams, fz_variant* out, uint32_t cap);

/* ------------------------------------------------------------------------------------------
 * fz_bank -- device-resident closure state for n_streams streams: the `state_` member of
 * stateful_lambda (flowz.hpp:1190-1191).  clone == copying the closure (snapshot, :1206).
 * The *_host entry points stage through device memory (H2D, kernel, D2H, synchronous); they
 * exist so that the reference's per-sample call protocol works unchanged on top.
 * ---------------------------------------------------------------------------------------- */
typedef struct fz_bank fz_bank;

int  fz_bank_create(fz_program* p, uint64_t n_streams, fz_bank** out);   /* zero state        */
int  fz_bank_clone(const fz_bank* b, fz_bank** out);
void fz_bank_destroy(fz_bank* b);
int  fz_bank_reset(fz_bank* b);                                          /* state := 0        */
int  fz_bank_set_params_host(fz_bank* b, const float* params /* [n_param][n_streams] */);
float* fz_bank_state_device(fz_bank* b);                                 /* [n_state][n_streams] */
int  fz_bank_process(fz_bank* b, const float* in_dev, float* out_dev, uint32_t n_samples,
                     const fz_variant* v, void* hip_stream);
/* fz_bank_process with stream-tiled frames (see fz_run_block_tiled) */
int  fz_bank_process_tiled(fz_bank* b, const float* in_dev, float* out_dev, uint32_t n_samples,
                           uint32_t tile_streams, const fz_variant* v, void* hip_stream);
/* fz_run_block_stream_major on the bank's state: in [n_streams][rows_total][n_in], out alike */
int  fz_bank_process_stream_major(fz_bank* b, const float* in_dev, float* out_dev, uint32_t rows_total, uint32_t row0,
                                  uint32_t n_samples, const fz_variant* v, void* hip_stream);
/* Control-rate modulation (the std::ref terminals of flowz/README.md:42-61 at block rate): the
 * rows_total samples of the frame buffers are processed in blocks of block_len samples, block k with the
 * per-stream coefficient set params_blocks[k] (device, [n_blocks][n_param][n_streams]; NULL: the bank's
 * own set).  ceil(rows_total / block_len) back-to-back launches, state carried.                          */
int  fz_bank_process_blocks(fz_bank* b, const float* in_dev, float* out_dev, uint32_t rows_total, uint32_t block_len,
                            const float* params_blocks, uint32_t tile_streams, const fz_variant* v, void* hip_stream);
/* fz_program_tune on the bank's own state and per-stream coefficients (the state advances: fz_bank_reset) */
int  fz_bank_tune(fz_bank* b, const float* in_dev, float* out_dev, uint32_t n_samples, uint32_t tile_streams,
                  void* hip_stream, fz_variant* chosen, float* chosen_ms);
int  fz_bank_process_host(fz_bank* b, const float* in_host, float* out_host, uint32_t n_samples);
/* host buffers in the reference's own calling convention: one contiguous sample buffer per stream,
 * in [n_streams][n_samples][n_in] -> out [n_streams][n_samples][n_out] (2-D copies of time chunks, the
 * stream-major kernel, the same three-stream pipeline)                                               */
int  fz_bank_process_host_stream_major(fz_bank* b, const float* in_host, float* out_host, uint32_t n_samples);
/* the same with float64 result frames (FZ_VF_OUT_F64): what a closure with double literals returns
 * in the reference (tuple<double>, flowz.hpp:1225-1229 with the ResultType of test/tests.cpp:201) */
int  fz_bank_process_host_f64(fz_bank* b, const float* in_host, double* out_host, uint32_t n_samples);

/* ------------------------------------------------------------------------------------------
 * Device utilities used by the measurement harness (bench.py) and tests.
 * ---------------------------------------------------------------------------------------- */
int fz_device_count(void);                /* 0 when no GPU is visible                          */
/* synthetic frames (t, s, w) = unit(hash32(seed, (stream0+s)*n_wires + w, t0+t)) in [-1,1), stored
 * time-major (tile_streams == 0) or stream-tiled as fz_run_block_tiled expects                */
int fz_synth_fill(float* dst_dev, uint64_t n_streams, uint32_t n_samples, uint32_t n_wires,
                  uint32_t seed, uint64_t stream0, uint64_t t0, uint32_t tile_streams, void* hip_stream);
/* Per-stream biquad coefficients on the device: the RBJ low-pass equations of the reference's
 * reactive_equations/reactive_filter_coeff.cpp:38-58 with its parameter types (every PARAMETER is
 * float, the literals `1.`, `2.` are double):
 *     w0 = two_pi*freq/sr (float)   cosw0 = cos(w0)   alpha = sin(w0)/(2.*Q)
 *     b0 = (1.-cosw0)/2.   b1 = 1.-cosw0   b2 = (1.-cosw0)/2.   a0 = 1.+alpha   a1 = -2.*cosw0   a2 = 1.-alpha
 * sin / cos of the float w0: argument reduction + polynomial in IEEE double, rounded to float once -- the correctly rounded
 * float (a libm's sinf / cosf, which is what the reference's std::sin(float) is, stays within 1 ULP of that; glibc's agrees
 * on 98.7 % of the arguments); |w0| >= 2^20 gives NaN coefficients.  raw6 [6][n_streams] receives a0 a1 a2 b0 b1 b2 (may be NULL);
 * df1 [5][n_streams] receives the rows a Flowz DF1 stage with fz_stream_param coefficients reads,
 * b0/a0 b1/a0 b2/a0 -a1/a0 -a2/a0 in float (may be NULL): pass a pointer into the `params` buffer. */
int fz_rbj_lowpass(const float* freq_dev, const float* q_dev, float sample_rate, uint64_t n_streams,
                   float* raw6_dev, float* df1_dev, void* hip_stream);
/* plain float4 copy kernel: the measured-copy-bandwidth yardstick of the roofline report      */
int fz_copy_probe(const float* src_dev, float* dst_dev, uint64_t n_floats, void* hip_stream);
/* Layout adapter for callers that hold one contiguous buffer per stream, as every closure of the
 * reference does (the sample loop of test/benchmark.cpp:137-147):
 *   stream-major  [n_streams][n_samples][n_wires]   <->   frames [n_samples][n_streams][n_wires]
 * (frames stream-tiled as fz_run_block_tiled takes them when tile_streams != 0: a multiple of 64 that
 * divides n_streams).  to_stream_major == 0: src is stream-major, dst are frames; != 0: the reverse.
 * One pass through LDS patches, reads and writes in 4 KiB runs; n_wires <= 64.                       */
int fz_transpose_frames(const float* src_dev, float* dst_dev, uint64_t n_streams, uint32_t n_samples, uint32_t n_wires,
                        uint32_t tile_streams, int to_stream_major, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWZ_HIP_H */
