// Host-only checks of the C++ EDSL front end (no GPU): arities, delays, lowering, error paths.
// Mirrors the analysis asserts of the reference's test/tests.cpp:63-102.
#include <complex>
#include <cstdio>
#include <cstdlib>

#include <flowz/flowz.hpp>

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main()
{
   using namespace flowz;

   // test/tests.cpp:88-90, 96-98
   auto wire_around_prev_box = (_1 |= _2);
   static_assert(decltype(wire_around_prev_box)::ins == 2 && decltype(wire_around_prev_box)::outs == 1, "");
   auto wire_around_succ_box = ((_1, _1) |= _1);
   static_assert(decltype(wire_around_succ_box)::ins == 1 && decltype(wire_around_succ_box)::outs == 2, "");

   // test/tests.cpp:67-77
   auto x = ~(_1 + _2[_1] |= _1[_1] + _2);
   static_assert(decltype(x)::ins == 2 && decltype(x)::outs == 1, "");
   CHECK((max_input_delays(x) == std::vector<uint32_t>{1, 0}));
   auto y = ~(_1 + _3[_1] |= _1[_1] + _2);
   static_assert(decltype(y)::ins == 3 && decltype(y)::outs == 1, "");
   CHECK((max_input_delays(y) == std::vector<uint32_t>{0, 1, 0}));
   CHECK(input_arity(x) == 2 && output_arity(x) == 1);

   // the benchmark graphs (test/benchmark.cpp:18-33, 79-88)
   const float b0 = 0.2, b1 = -0.3, b2 = 1.1, a1 = -0.2, a2 = 0.8;
   auto fwd = (b0 * _1 + b1 * _1[_1] + b2 * _1[_2]);
   auto bwd = ~(_2 + a1 * _1[_1] + a2 * _1[_2]);
   auto df1 = compile(fwd |= bwd);
   static_assert(decltype(df1)::ins == 1 && decltype(df1)::outs == 1, "");
   CHECK(df1.info().n_ops == 9 && df1.info().n_state == 4 && df1.info().n_const == 5);
   auto df2 = compile(bwd |= fwd);
   CHECK(df2.info().n_ops == 9 && df2.info().n_state == 2);
   auto delay_add_2 = (_1[_1] + _2 |= _1[_1] + _2);
   auto fwdt = ((b2 * _1, b1 * _1, b0 * _1) |= delay_add_2);
   auto bwdt = ((-a2 * _1, -a1 * _1) |= delay_add_2);
   static_assert(decltype(fwdt)::ins == 1 && decltype(bwdt)::ins == 2, "");
   auto df1t = compile(~bwdt |= fwdt);
   // 9 tree nodes, but -a1 == b0 == 0.2f here, so b0*u and (-a1)*u are ONE value-identical node
   CHECK(df1t.info().n_ops == 8 && df1t.info().n_state == 4);
   auto chain = compile(fwd |= bwd |= fwd |= bwd |= fwd |= bwd |= fwd |= bwd |= fwd |= bwd |= fwd |= bwd);
   CHECK(chain.info().n_ops == 54 && chain.info().n_state == 14);

   // _1[-2] sugar == _1[_2]
   CHECK(compile(_1[-2]).info().max_delay == 2);

   // std::ref terminals get their own run-time coefficient slot (flowz/README.md:42-61)
   float a = 1.f;
   auto one_pole = compile(~(std::ref(a) * _1[_1] + 0.1 * _2));
   CHECK(one_pole.info().n_const == 1 && one_pole.info().n_const64 == 1);   // 0.1 is a double literal

   {  // comparison and logical operators: proto::_default applies whatever C++ operator a node is (flowz.hpp:51-55, :769-772)
      auto clip = compile(_1 * ((_1 > -0.5f) && (_1 < 0.5f)) + 0.5f * (_1 >= 0.5f) + -0.5f * (_1 <= -0.5f));
      static_assert(decltype(clip)::ins == 1 && decltype(clip)::outs == 1, "");
      CHECK(clip.info().n_ops == 12 && clip.info().stage_packable == 0);
      CHECK(compile(!_1).info().n_ops == 1);
      auto pick = compile((_1 == _2) || (_1 != 1.0));
      static_assert(decltype(pick)::ins == 2 && decltype(pick)::outs == 1, "");
      CHECK(pick.info().n_const64 == 1);                                                       // compared in double, the result a float
      CHECK(compile_typed((_1 < 1.0) * _1).output_dtypes() == std::vector<uint32_t>{FZ_DT_F32});   // bool * float: a float multiplication
      CHECK(compile_typed((_1 < 1.0) * 2.0).output_dtypes() == std::vector<uint32_t>{FZ_DT_F64});  // bool * double: a double one
   }

   // malformed graphs throw at compile() instead of failing template instantiation
   bool threw = false;
   try { compile(~(_1 + _2)); } catch (const flowz::error& e) { threw = e.code == FZ_E_GRAPH; }
   CHECK(threw);

   {  // test_result_type_transform (tests.cpp:184-232) through compile_typed(): ResultType itself, no GPU needed
      using cplx = std::complex<float>;
      using T = std::vector<uint32_t>;
      const uint32_t F = FZ_DT_F32, D = FZ_DT_F64, C = FZ_DT_CF32;
      CHECK(compile_typed(_1).output_dtypes() == T{F});                                    // :200
      CHECK(compile_typed(_1 * 1.0).output_dtypes() == T{D});                              // :201
      CHECK(compile_typed(_1 * 1.0 |= _1).output_dtypes() == T{D});                        // :204
      CHECK(compile_typed(_1 |= cplx{1, 0} * _1).output_dtypes() == T{C});                 // :206
      CHECK(compile_typed((_1 * 1.0, _1) |= (_2, _1)).output_dtypes() == (T{F, D}));       // :214
      CHECK(compile_typed(_1 |= _1[_1]).output_dtypes() == T{F});                          // :218
      CHECK(compile_typed((_1[_1], 1.0 * _1) |= _2[_1]).output_dtypes() == T{D});          // :219  a double THROUGH a delay line
      CHECK(compile((_1[_1], 1.0 * _1) |= _2[_1]).output_dtypes() == T{F});                //       compile(): float state (flowz.hpp:1245)
      CHECK(compile_typed(~(_1[_1] + _2)).output_dtypes() == T{F});                        // :221
      CHECK(compile_typed(~(1.0 * _1[_1] + _2)).output_dtypes() == T{D});                  // :222
      CHECK(compile_typed(~(_1[_1] + 1.0 * _2)).output_dtypes() == T{D});                  // :223
      CHECK(compile_typed(~(_1[_1] + 1.0)).output_dtypes() == T{D});                       // :224
      CHECK(compile_typed(~(cplx{0.5f, 0.5f} * _1[_1] + _2)).output_dtypes() == T{C});     // complex state
      CHECK(compile_typed(_1 * 2.f, {FZ_DT_F64}).output_dtypes() == T{D});                 // a double ARGUMENT: f(1.0)
      using cd = std::complex<double>;
      const uint32_t Z = FZ_DT_CF64;
      CHECK(compile_typed(_1 |= cd{1, 0} * _1, {FZ_DT_F64}).output_dtypes() == T{Z});      // complex<double> * double
      CHECK(compile_typed(~(cd{0.5, 0.5} * _1[_1] + 1.0 * _2)).output_dtypes() == T{Z});   // complex<double> state; float*1.0 is double
      CHECK(compile_typed(_1 / cd{2, 1}, {Z}).output_dtypes() == T{Z});                    // a complex<double> argument
      bool threw3 = false;
      try { (void)compile_typed(cd{1, 0} * _1); } catch (const flowz::error& e) { threw3 = e.code == FZ_E_GRAPH; }
      CHECK(threw3);                                                                       // complex<double> * float: no such operator
      threw3 = false;
      try { (void)compile_typed((cd{1, 0} * _1) * cplx{1, 0}, {FZ_DT_F64}); } catch (const flowz::error& e) { threw3 = e.code == FZ_E_GRAPH; }
      CHECK(threw3);                                                                       // complex<double> * complex<float>: neither
      bool threw2 = false;
      try { (void)compile(~(cplx{0.5f, 0.5f} * _1[_1] + _2)); } catch (const flowz::error&) { threw2 = true; }
      CHECK(threw2);                                                                       // compile() cannot store a complex
   }

   // without a GPU the per-sample call must fail loudly (no CPU fallback)
   if (fz_device_count() == 0) {
      bool nodev = false;
      try { auto f = compile(_1); (void)f(1337); } catch (const flowz::error& e) { nodev = e.code == FZ_E_NO_DEVICE; }
      CHECK(nodev);
   }
   std::printf(failures ? "%d FAILURES\n" : "all host EDSL checks passed\n", failures);
   return failures ? 1 : 0;
}
