// flowz.hpp -- the Flowz EDSL front end for the MI355X evaluator (host side, plain C++14).
//
// Same surface as the reference header /root/reference/flowz/flowz.hpp, re-built without
// Boost.Proto on top of the C ABI of libflowz_hip.so (include/flowz_hip.h):
//
//    using namespace flowz;
//    _1 .. _6, make_placeholder<N>()                 flowz.hpp:1252-1257, :78-82
//    _1[_2]            wire 1 delayed by 2           :84-85   (also _1[-2]: the spelling of
//                                                    experimental_steps/delay_expression.cpp:99-100)
//    a , b   a | b   a |= b   ~a                     :90-93   channel / parallel / sequence / feedback
//    + - * / unary -   with literals                 :68-72, :769-772 (float terminals; a `double` literal
//                                                    makes the operators above it evaluate in float64)
//    std::ref(x)       external modulation           flowz/README.md:42-61 (read at every call / block)
//    compile(expr)     -> callable closure           :1233-1249
//    f(x1..xN) -> std::tuple<float x M>              :1225-1229, fewer args -> curried copy :1203-1212
//
// What differs, on purpose: the arities are still compile-time (so results are tuples), but
// everything else happens at run time -- the expression is a tree of C-ABI handles, compile()
// lowers it and builds ONE fused gfx950 kernel, and the closure's state lives in HBM.  The
// per-sample call works (it launches the kernel for 1 stream x 1 sample), but the intended use
// is the block API: f.bank(n_streams).process(in, out, n_samples) evaluates n_samples samples
// of n_streams independent closures per launch.  The bank also takes stream-tiled frames
// (process_tiled), one contiguous buffer per stream (process_stream_major, the reference's own calling
// convention), host memory (process_host, process_host_stream_major: pipelined over PCIe), blocks with
// per-block coefficient sets (process_blocks) and can measure its kernel plan once (tune).
// std::complex<float> terminals make complex wires (results through call_flat), double literals double
// sub-expressions (call_f64).  There is no CPU evaluation path.
// Malformed graphs throw flowz::error from compile() instead of failing template instantiation.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <complex>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "flowz_hip.h"

namespace flowz {

struct error : std::runtime_error {
   int code;
   error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

namespace detail {

inline void check(int rc)
{
   if (rc < 0) throw error(rc, fz_last_error());
}

// value-semantic owner of an fz_expr handle (copy = retain: trees are immutable)
class handle {
   fz_expr* p_ = nullptr;

public:
   handle() = default;
   explicit handle(fz_expr* p) : p_(p)
   {
      if (!p_) throw error(FZ_E_INVALID, fz_last_error());
   }
   handle(const handle& o) : p_(o.p_) { fz_expr_retain(p_); }
   handle(handle&& o) noexcept : p_(o.p_) { o.p_ = nullptr; }
   handle& operator=(handle o) noexcept
   {
      std::swap(p_, o.p_);
      return *this;
   }
   ~handle() { fz_expr_release(p_); }
   fz_expr* get() const { return p_; }
};

// std::ref terminals: (uniform coefficient id, address of the referenced variable)
using ref_list = std::vector<std::pair<uint32_t, const float*>>;

inline uint32_t next_uniform_id()
{
   static std::atomic<uint32_t> n{0};      // expressions may be built on several threads
   return n.fetch_add(1, std::memory_order_relaxed);
}

constexpr int imax(int a, int b) { return a > b ? a : b; }

// one argument of a typed call -> the float slots of its input wire
template <class T>
inline void put_typed(float*& w, uint32_t dtype, const T& v)
{
   if (dtype == FZ_DT_F32) {
      *w++ = static_cast<float>(v);
   } else if (dtype == FZ_DT_F64) {
      const double d = static_cast<double>(v);
      std::memcpy(w, &d, sizeof d);
      w += 2;
   } else if (dtype == FZ_DT_CF64) {           // a real argument on a complex<double> wire: (v, 0.0)
      const double d[2] = {static_cast<double>(v), 0.0};
      std::memcpy(w, d, sizeof d);
      w += 4;
   } else {
      *w++ = static_cast<float>(v);            // a real argument on a complex wire: (v, 0)
      *w++ = 0.f;
   }
}
inline void put_typed(float*& w, uint32_t dtype, const std::complex<double>& v)
{
   if (dtype != FZ_DT_CF64) throw std::invalid_argument("flowz: a std::complex<double> argument needs an input wire declared FZ_DT_CF64");
   const double d[2] = {v.real(), v.imag()};
   std::memcpy(w, d, sizeof d);
   w += 4;
}
inline void put_typed(float*& w, uint32_t dtype, const std::complex<float>& v)
{
   if (dtype != FZ_DT_CF32) throw std::invalid_argument("flowz: a std::complex<float> argument needs an input wire declared FZ_DT_CF32");
   *w++ = v.real();
   *w++ = v.imag();
}

struct expr_tag {};

}  // namespace detail

// An expression with In input wires and Out output wires (input_arity / output_arity,
// flowz.hpp:162-246, evaluated by the type system exactly like the reference does).
template <int In, int Out>
struct expr : detail::expr_tag {
   static constexpr int ins = In;
   static constexpr int outs = Out;
   detail::handle h;
   detail::ref_list refs;
   expr(detail::handle hh, detail::ref_list r = {}) : h(std::move(hh)), refs(std::move(r)) {}
};

template <class T>
using is_expr = std::is_base_of<detail::expr_tag, typename std::decay<T>::type>;

namespace detail {

inline ref_list merge(const ref_list& a, const ref_list& b)
{
   ref_list r = a;
   r.insert(r.end(), b.begin(), b.end());
   return r;
}

// literal terminals are held by value (make_terminal, :68-72).  A C++ `double` literal stays a
// double: the operators above it evaluate in float64 exactly as the built-in operators of the
// reference do (proto::_default :769-772); everything else (float, integers) is a float32 terminal.
template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
expr<0, 1> as_expr(T v)
{
   if (std::is_same<T, double>::value || std::is_same<T, long double>::value)
      return expr<0, 1>(handle(fz_literal_f64(static_cast<double>(v))));
   return expr<0, 1>(handle(fz_literal(static_cast<float>(v))));
}
// a std::complex<float> terminal (test/tests.cpp:206-207): the wire above it is complex
inline expr<0, 1> as_expr(const std::complex<float>& z) { return expr<0, 1>(handle(fz_literal_c32(z.real(), z.imag()))); }
// a std::complex<double> terminal: double parts; it meets double operands only, as in C++
inline expr<0, 1> as_expr(const std::complex<double>& z) { return expr<0, 1>(handle(fz_literal_c64(z.real(), z.imag()))); }
inline expr<0, 1> as_expr(std::reference_wrapper<float> r)
{
   const uint32_t id = next_uniform_id();
   return expr<0, 1>(handle(fz_uniform(id, r.get())), ref_list{{id, &r.get()}});
}
inline expr<0, 1> as_expr(std::reference_wrapper<const float> r)
{
   const uint32_t id = next_uniform_id();
   return expr<0, 1>(handle(fz_uniform(id, r.get())), ref_list{{id, &r.get()}});
}
template <int I, int O>
const expr<I, O>& as_expr(const expr<I, O>& e)
{
   return e;
}

template <class T>
struct is_operand : std::integral_constant<bool, std::is_arithmetic<typename std::decay<T>::type>::value ||
                                                    std::is_same<typename std::decay<T>::type, std::complex<float>>::value ||
                                                    std::is_same<typename std::decay<T>::type, std::complex<double>>::value ||
                                                    std::is_same<typename std::decay<T>::type, std::reference_wrapper<float>>::value ||
                                                    std::is_same<typename std::decay<T>::type, std::reference_wrapper<const float>>::value> {};

template <class A, class B>
using enable_binary = typename std::enable_if<(is_expr<A>::value && (is_expr<B>::value || is_operand<B>::value)) ||
                                              (is_operand<A>::value && is_expr<B>::value)>::type;

template <int Ia, int Oa, int Ib, int Ob>
expr<imax(Ia, Ib), 1> arith(fz_op op, const expr<Ia, Oa>& a, const expr<Ib, Ob>& b)
{
   static_assert(Oa == 1 && Ob == 1, "flowz: an arithmetic operand must have exactly one output wire");
   return expr<imax(Ia, Ib), 1>(handle(fz_arith(op, a.h.get(), b.h.get())), merge(a.refs, b.refs));
}

}  // namespace detail

// ---- placeholders and delays -----------------------------------------------------------------------
template <int I>
struct placeholder : expr<I, 1> {
   placeholder() : expr<I, 1>(detail::handle(fz_placeholder(I))) {}
   template <int N>
   expr<I, 1> operator[](const placeholder<N>&) const          // _i[_n]
   {
      return expr<I, 1>(detail::handle(fz_delayed(I, N)));
   }
   expr<I, 1> operator[](int n) const                            // _i[-n]
   {
      return expr<I, 1>(detail::handle(fz_delayed(I, static_cast<uint32_t>(n < 0 ? -n : n))));
   }
};

template <int N>
placeholder<N> make_placeholder()
{
   return placeholder<N>();
}

template <class X>
auto make_terminal(X x) -> decltype(detail::as_expr(x))
{
   return detail::as_expr(x);
}

// per-stream, block-constant coefficient k: one value per stream (array given to the bank)
inline expr<0, 1> stream_param(uint32_t k) { return expr<0, 1>(detail::handle(fz_stream_param(k))); }
// sample-rate modulator k: what a std::ref(x) terminal is when the caller changes x between CALLS (flowz/README.md:42-61),
// for the block API -- one value per sample, the same for all streams (bank.set_modulation)
inline expr<0, 1> modulator(uint32_t k) { return expr<0, 1>(detail::handle(fz_modulator(k))); }

// ---- arithmetic (any C++ operator on evaluated children, proto::_default :769-772) --------------------------
#define FLOWZ_BINARY_OP(SYM, OP)                                                                         \
   template <class A, class B, class = detail::enable_binary<A, B>>                                      \
   auto operator SYM(const A& a, const B& b)->decltype(detail::arith(OP, detail::as_expr(a), detail::as_expr(b))) \
   {                                                                                                     \
      return detail::arith(OP, detail::as_expr(a), detail::as_expr(b));                                  \
   }
FLOWZ_BINARY_OP(+, FZ_OP_ADD)
FLOWZ_BINARY_OP(-, FZ_OP_SUB)
FLOWZ_BINARY_OP(*, FZ_OP_MUL)
FLOWZ_BINARY_OP(/, FZ_OP_DIV)
// the comparison and logical operators: the bool C++ yields, as it behaves in arithmetic -- 1 or 0, taking the type of what it meets
// (a result tuple or a delay line receives 1.0f / 0.0f).  && and || evaluate both sides (there is nothing to skip).
FLOWZ_BINARY_OP(<, FZ_OP_LT)
FLOWZ_BINARY_OP(<=, FZ_OP_LE)
FLOWZ_BINARY_OP(>, FZ_OP_GT)
FLOWZ_BINARY_OP(>=, FZ_OP_GE)
FLOWZ_BINARY_OP(==, FZ_OP_EQ)
FLOWZ_BINARY_OP(!=, FZ_OP_NE)
FLOWZ_BINARY_OP(&&, FZ_OP_AND)
FLOWZ_BINARY_OP(||, FZ_OP_OR)
#undef FLOWZ_BINARY_OP

template <int I, int O>
expr<I, 1> operator!(const expr<I, O>& a)
{
   static_assert(O == 1, "flowz: an arithmetic operand must have exactly one output wire");
   return expr<I, 1>(detail::handle(fz_arith(FZ_OP_NOT, a.h.get(), nullptr)), a.refs);
}

template <int I, int O>
expr<I, 1> operator-(const expr<I, O>& a)
{
   static_assert(O == 1, "flowz: an arithmetic operand must have exactly one output wire");
   return expr<I, 1>(detail::handle(fz_arith(FZ_OP_NEG, a.h.get(), nullptr)), a.refs);
}

// ---- block composition operators (flowz.hpp:90-93) --------------------------------------------------------------
template <int Ia, int Oa, int Ib, int Ob>
expr<detail::imax(Ia, Ib), Oa + Ob> operator,(const expr<Ia, Oa>& a, const expr<Ib, Ob>& b)        // channel
{
   return {detail::handle(fz_channel(a.h.get(), b.h.get())), detail::merge(a.refs, b.refs)};
}

template <int Ia, int Oa, int Ib, int Ob>
expr<Ia + Ib, Oa + Ob> operator|(const expr<Ia, Oa>& a, const expr<Ib, Ob>& b)                      // parallel
{
   return {detail::handle(fz_parallel(a.h.get(), b.h.get())), detail::merge(a.refs, b.refs)};
}

template <int Ia, int Oa, int Ib, int Ob>
expr<Ia + detail::imax(0, Ib - Oa), Ob + detail::imax(0, Oa - Ib)> operator|=(const expr<Ia, Oa>& a, const expr<Ib, Ob>& b)   // sequence
{
   return {detail::handle(fz_sequence(a.h.get(), b.h.get())), detail::merge(a.refs, b.refs)};
}

template <int I, int O>
expr<detail::imax(0, I - O), O> operator~(const expr<I, O>& a)                                       // feedback
{
   return {detail::handle(fz_feedback(a.h.get())), a.refs};
}

// static analysis (flowz.hpp:162-246, :443-506)
template <int I, int O>
constexpr int input_arity(const expr<I, O>&) { return I; }
template <int I, int O>
constexpr int output_arity(const expr<I, O>&) { return O; }
template <int I, int O>
std::vector<uint32_t> max_input_delays(const expr<I, O>& e)
{
   std::vector<uint32_t> v(static_cast<size_t>(fz_max_input_delays(e.h.get(), nullptr, 0)));
   if (!v.empty()) fz_max_input_delays(e.h.get(), v.data(), static_cast<uint32_t>(v.size()));
   return v;
}

// ---- compiled graphs --------------------------------------------------------------------------------------------
namespace detail {

struct program_deleter {
   void operator()(fz_program* p) const { fz_program_destroy(p); }
};
using program_ptr = std::shared_ptr<fz_program>;

template <class Tuple, class T, size_t... K>
Tuple to_tuple(const T* v, std::index_sequence<K...>)
{
   return Tuple(v[K]...);
}

template <class T, int N, class = std::make_index_sequence<N>>
struct value_tuple;
template <class T, int N, size_t... K>
struct value_tuple<T, N, std::index_sequence<K...>> {
   template <size_t>
   using f = T;
   using type = std::tuple<f<K>...>;
};
template <int N>
using float_tuple = value_tuple<float, N>;

}  // namespace detail

// Device-resident state of n_streams independent closures of one compiled graph
// (the `state_` of stateful_lambda, flowz.hpp:1190-1191, times n_streams, in HBM).
class stream_bank {
   detail::program_ptr prog_;
   detail::ref_list refs_;
   fz_bank* bank_ = nullptr;
   uint64_t n_streams_ = 0;

   void refresh_refs() const
   {
      for (const auto& r : refs_) detail::check(fz_program_set_uniform(prog_.get(), r.first, *r.second));
   }

public:
   stream_bank(detail::program_ptr p, detail::ref_list refs, uint64_t n_streams)
       : prog_(std::move(p)), refs_(std::move(refs)), n_streams_(n_streams)
   {
      detail::check(fz_bank_create(prog_.get(), n_streams, &bank_));
   }
   stream_bank(const stream_bank& o) : prog_(o.prog_), refs_(o.refs_), n_streams_(o.n_streams_)   // copy = snapshot (:1206)
   {
      detail::check(fz_bank_clone(o.bank_, &bank_));
   }
   stream_bank(stream_bank&& o) noexcept : prog_(std::move(o.prog_)), refs_(std::move(o.refs_)), bank_(o.bank_), n_streams_(o.n_streams_)
   {
      o.bank_ = nullptr;
   }
   stream_bank& operator=(stream_bank o) noexcept
   {
      std::swap(prog_, o.prog_);
      std::swap(refs_, o.refs_);
      std::swap(bank_, o.bank_);
      std::swap(n_streams_, o.n_streams_);
      return *this;
   }
   ~stream_bank() { fz_bank_destroy(bank_); }

   uint64_t n_streams() const { return n_streams_; }
   void reset() { detail::check(fz_bank_reset(bank_)); }
   void set_stream_params(const float* host /* [n_param][n_streams] */) { detail::check(fz_bank_set_params_host(bank_, host)); }
   // values of the sample-rate modulators for the following blocks: DEVICE array [n_mod][stride], stride >= the rows of a block
   void set_modulation(const float* mod_dev, uint32_t stride) { detail::check(fz_program_set_modulation(prog_.get(), mod_dev, stride)); }
   float* state_device() { return fz_bank_state_device(bank_); }

   // frames are time-major: in [n_samples][n_streams][n_in], out [n_samples][n_streams][n_out]
   void process(const float* in_dev, float* out_dev, uint32_t n_samples, void* hip_stream = nullptr, const fz_variant* v = nullptr)
   {
      refresh_refs();
      detail::check(fz_bank_process(bank_, in_dev, out_dev, n_samples, v, hip_stream));
   }
   // stream-tiled frames (the HBM-friendly layout): in [n_streams/tile][n_samples][tile][n_in], out alike;
   // recommended_tile_streams() gives ~32 KiB row segments for this graph
   void process_tiled(const float* in_dev, float* out_dev, uint32_t n_samples, uint32_t tile_streams,
                      void* hip_stream = nullptr, const fz_variant* v = nullptr)
   {
      refresh_refs();
      detail::check(fz_bank_process_tiled(bank_, in_dev, out_dev, n_samples, tile_streams, v, hip_stream));
   }
   // one contiguous buffer per stream ([n_streams][rows_total][wires], the reference's calling convention):
   // samples [row0, row0 + n_samples) straight through the block kernel, no layout pass
   void process_stream_major(const float* in_dev, float* out_dev, uint32_t rows_total, uint32_t row0, uint32_t n_samples,
                             void* hip_stream = nullptr, const fz_variant* v = nullptr)
   {
      refresh_refs();
      detail::check(fz_bank_process_stream_major(bank_, in_dev, out_dev, rows_total, row0, n_samples, v, hip_stream));
   }
   // control-rate modulation: the buffers hold rows_total samples; block k (block_len samples) runs with the
   // per-stream coefficient set params_blocks_dev[k] ([n_blocks][n_param][n_streams]; nullptr: the bank's own)
   void process_blocks(const float* in_dev, float* out_dev, uint32_t rows_total, uint32_t block_len, const float* params_blocks_dev = nullptr,
                       uint32_t tile_streams = 0, void* hip_stream = nullptr, const fz_variant* v = nullptr)
   {
      refresh_refs();
      detail::check(fz_bank_process_blocks(bank_, in_dev, out_dev, rows_total, block_len, params_blocks_dev, tile_streams, v, hip_stream));
   }
   uint32_t recommended_tile_streams() const { return fz_recommended_tile_streams(prog_.get()); }
   // measure the kernel variants for this shape on these buffers once and keep the fastest for later
   // process()/process_tiled() calls (fz_program_tune); the bank's state advances: reset() afterwards
   fz_variant tune(const float* in_dev, float* out_dev, uint32_t n_samples, uint32_t tile_streams = 0, void* hip_stream = nullptr)
   {
      refresh_refs();
      fz_variant v{0, 0, 0, 0};
      detail::check(fz_bank_tune(bank_, in_dev, out_dev, n_samples, tile_streams, hip_stream, &v, nullptr));
      return v;
   }
   void process_host(const float* in_host, float* out_host, uint32_t n_samples)
   {
      refresh_refs();
      detail::check(fz_bank_process_host(bank_, in_host, out_host, n_samples));
   }
   // host buffers, one contiguous row per stream ([n_streams][n_samples][wires]): the reference's own convention
   void process_host_stream_major(const float* in_host, float* out_host, uint32_t n_samples)
   {
      refresh_refs();
      detail::check(fz_bank_process_host_stream_major(bank_, in_host, out_host, n_samples));
   }
   // float64 result frames: double sub-expression results leave un-narrowed, float wires widen exactly
   void process_host(const float* in_host, double* out_host, uint32_t n_samples)
   {
      refresh_refs();
      detail::check(fz_bank_process_host_f64(bank_, in_host, out_host, n_samples));
   }
};

// compile() result: the reference's stateful_lambda (flowz.hpp:1181-1230).
template <int In, int Out>
class stateful_lambda {
   detail::program_ptr prog_;
   detail::ref_list refs_;
   std::unique_ptr<stream_bank> own_;     // lazily created 1-stream bank behind operator()
   std::vector<uint32_t> in_dtypes_;      // compile_typed(): fz_dtype of every input wire

   stream_bank& own()
   {
      if (!own_) own_.reset(new stream_bank(prog_, refs_, 1));
      return *own_;
   }

   using result_t = typename detail::float_tuple<Out>::type;
   using result_f64_t = typename detail::value_tuple<double, Out>::type;

   template <class T, class... Args>
   typename detail::value_tuple<T, Out>::type call_as(const Args&... args)
   {
      if (info().n_out != static_cast<uint32_t>(Out))
         throw error(FZ_E_INVALID, "this graph has std::complex output wires: call it with call_flat(x...)");
      const float in[In > 0 ? In : 1] = {static_cast<float>(args)...};
      T out[Out];
      own().process_host(In > 0 ? in : nullptr, out, 1);
      return detail::to_tuple<typename detail::value_tuple<T, Out>::type>(out, std::make_index_sequence<Out>{});
   }

   template <class... Args>
   result_t call(std::integral_constant<int, 0>, const Args&... args)
   {
      return call_as<float>(args...);
   }

   template <int Missing, class... Args>
   auto call(std::integral_constant<int, Missing>, const Args&... args)      // currying, flowz.hpp:1203-1212
   {
      return [args..., self = *this](const auto&... rest) mutable { return self(args..., rest...); };
   }

public:
   static constexpr int ins = In;
   static constexpr int outs = Out;

   explicit stateful_lambda(const expr<In, Out>& e) : refs_(e.refs)
   {
      fz_program* p = nullptr;
      detail::check(fz_compile(e.h.get(), &p));
      prog_ = detail::program_ptr(p, detail::program_deleter());
   }
   // compile_typed(): the wire types of the reference's ResultType transform (flowz.hpp:585-644) carried through inputs,
   // state and outputs (fz_compile_typed); in_dtypes: one fz_dtype per input wire, empty = all float
   stateful_lambda(const expr<In, Out>& e, const std::vector<uint32_t>& in_dtypes) : refs_(e.refs), in_dtypes_(in_dtypes)
   {
      if (in_dtypes_.empty()) in_dtypes_.assign(In > 0 ? In : 0, static_cast<uint32_t>(FZ_DT_F32));
      fz_program* p = nullptr;
      detail::check(fz_compile_typed(e.h.get(), in_dtypes_.empty() ? nullptr : in_dtypes_.data(), static_cast<uint32_t>(in_dtypes_.size()), &p));
      prog_ = detail::program_ptr(p, detail::program_deleter());
   }
   stateful_lambda(const stateful_lambda& o)
      : prog_(o.prog_), refs_(o.refs_), own_(o.own_ ? new stream_bank(*o.own_) : nullptr), in_dtypes_(o.in_dtypes_)
   {
   }
   stateful_lambda(stateful_lambda&&) = default;
   stateful_lambda& operator=(stateful_lambda o)
   {
      std::swap(prog_, o.prog_);
      std::swap(refs_, o.refs_);
      std::swap(own_, o.own_);
      std::swap(in_dtypes_, o.in_dtypes_);
      return *this;
   }

   // one sample of a compile_typed() closure: the arguments are converted to the declared type of their input wire
   // (float / double / std::complex<float> / std::complex<double>), the result is the raw output frame -- 1 float slot per
   // float wire, 2 per double wire (the 8 bytes of the double), 2 per complex<float> wire (re, im), 4 per complex<double>
   // wire (the double of re, then of im); read it with typed_f32/f64/c32/c64 below and
   // the wire types of output_dtypes()
   template <class... Args, class = typename std::enable_if<(sizeof...(Args) == In)>::type>
   std::vector<float> call_typed(const Args&... args)
   {
      if (!info().typed) throw error(FZ_E_INVALID, "call_typed needs a closure made by compile_typed()");
      std::vector<float> in(info().n_in > 0 ? info().n_in : 1), out(info().n_out);
      float* w = in.data();
      size_t k = 0;
      (void)std::initializer_list<int>{(detail::put_typed(w, in_dtypes_.at(k++), args), 0)...};
      own().process_host(info().n_in ? in.data() : nullptr, out.data(), 1);
      return out;
   }
   // wire types of the outputs (fz_dtype per wire), ResultType of the expression for compile_typed() closures
   std::vector<uint32_t> output_dtypes() const
   {
      std::vector<uint32_t> slots(info().n_out), wires;
      detail::check(fz_program_output_dtypes(prog_.get(), slots.data(), static_cast<uint32_t>(slots.size())));
      for (uint32_t c : slots)
         if (c == 0) wires.push_back(FZ_DT_F32);
         else if (c == 1 || c == 4) wires.push_back(FZ_DT_F64);
         else if (c == 2) wires.push_back(FZ_DT_CF32);
         else if (c == 6 || c == 10) wires.push_back(FZ_DT_CF64);
      return wires;
   }

   // one sample of one stream; fewer arguments return a curried copy of the closure
   template <class... Args, class = typename std::enable_if<(sizeof...(Args) <= In)>::type>
   auto operator()(const Args&... args)
   {
      return call(std::integral_constant<int, In - static_cast<int>(sizeof...(Args))>{}, args...);
   }

   // one sample with the results as doubles: what the reference's closure returns for graphs with
   // double literals (tuple<double>, test/tests.cpp:201-231); float wires widen exactly.  The wire
   // types themselves are run-time here: fz_program_output_dtypes()
   template <class... Args, class = typename std::enable_if<(sizeof...(Args) == In)>::type>
   result_f64_t call_f64(const Args&... args)
   {
      return call_as<double>(args...);
   }

   // one sample, results as the raw output frame: one float per real wire, (re, im) per std::complex
   // wire (the wire types are run-time here: fz_program_output_dtypes)
   template <class... Args, class = typename std::enable_if<(sizeof...(Args) == In)>::type>
   std::vector<float> call_flat(const Args&... args)
   {
      const float in[In > 0 ? In : 1] = {static_cast<float>(args)...};
      std::vector<float> out(info().n_out);
      own().process_host(In > 0 ? in : nullptr, out.data(), 1);
      return out;
   }

   // the block API: n_streams independent closures with zeroed state in HBM
   stream_bank bank(uint64_t n_streams) const { return stream_bank(prog_, refs_, n_streams); }

   fz_program* program() const { return prog_.get(); }
   fz_info info() const
   {
      fz_info i;
      detail::check(fz_program_info(prog_.get(), &i));
      return i;
   }
};

struct compile_fn {
   template <int I, int O>
   stateful_lambda<I, O> operator()(const expr<I, O>& e) const
   {
      return stateful_lambda<I, O>(e);
   }
};
static const compile_fn compile{};

// compile() with ResultType semantics (flowz.hpp:585-644, test/tests.cpp:184-232): every wire keeps its C++ type through
// inputs, delay lines and outputs.  in_dtypes: fz_dtype of each input wire (the reference's closure is a template over its
// argument types); call the closure with call_typed(x...).
struct compile_typed_fn {
   template <int I, int O>
   stateful_lambda<I, O> operator()(const expr<I, O>& e, const std::vector<uint32_t>& in_dtypes = {}) const
   {
      return stateful_lambda<I, O>(e, in_dtypes);
   }
};
static const compile_typed_fn compile_typed{};

// readers of a call_typed() frame: the value that starts at slot k
inline float typed_f32(const std::vector<float>& frame, size_t k) { return frame.at(k); }
inline double typed_f64(const std::vector<float>& frame, size_t k)
{
   double d;
   std::memcpy(&d, &frame.at(k + 1) - 1, sizeof d);
   return d;
}
inline std::complex<float> typed_c32(const std::vector<float>& frame, size_t k) { return {frame.at(k), frame.at(k + 1)}; }
inline std::complex<double> typed_c64(const std::vector<float>& frame, size_t k) { return {typed_f64(frame, k), typed_f64(frame, k + 2)}; }

// layout adapter for callers that keep one contiguous buffer per stream ([stream][t][wire], what each
// closure of the reference loops over): to / from the frames the block API takes (fz_transpose_frames)
inline void frames_from_stream_major(const float* src_dev, float* frames_dev, uint64_t n_streams, uint32_t n_samples,
                                     uint32_t n_wires = 1, uint32_t tile_streams = 0, void* hip_stream = nullptr)
{
   detail::check(fz_transpose_frames(src_dev, frames_dev, n_streams, n_samples, n_wires, tile_streams, 0, hip_stream));
}
inline void frames_to_stream_major(const float* frames_dev, float* dst_dev, uint64_t n_streams, uint32_t n_samples,
                                   uint32_t n_wires = 1, uint32_t tile_streams = 0, void* hip_stream = nullptr)
{
   detail::check(fz_transpose_frames(frames_dev, dst_dev, n_streams, n_samples, n_wires, tile_streams, 1, hip_stream));
}

static const placeholder<1> _1{};
static const placeholder<2> _2{};
static const placeholder<3> _3{};
static const placeholder<4> _4{};
static const placeholder<5> _5{};
static const placeholder<6> _6{};

}  // namespace flowz
