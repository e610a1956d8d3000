// The reference's evaluation tests (test/tests.cpp:88-178) and README examples, written against
// include/flowz/flowz.hpp exactly like the reference writes them -- every call launches the
// fused kernel on the GPU (1 stream x 1 sample).  Plus currying, closure copies and the block API.
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include <flowz/flowz.hpp>

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main()
{
   using namespace flowz;

   {  // test_wires_around_boxes, tests.cpp:88-102
      auto wp = compile(_1 |= _2);
      auto wpr = wp(2, 1337);
      CHECK(std::make_tuple(1337) == wpr);
      CHECK(std::tuple_size<decltype(wpr)>::value == 1);
      auto ws = compile((_1, _1) |= _1);
      auto wsr = ws(1337);
      CHECK(std::tuple_size<decltype(wsr)>::value == 2);
      CHECK(std::make_tuple(1337, 1337) == wsr);
   }
   {  // test_simple_expressions, tests.cpp:106-136
      auto identity = compile(_1);
      CHECK(std::make_tuple(1337) == identity(1337));
      CHECK(std::make_tuple(42) == identity(42));
      auto unit_delay = compile(_1[_1]);
      CHECK(std::make_tuple(0) == unit_delay(1337));
      CHECK(std::make_tuple(1337) == unit_delay(42));
      CHECK(std::make_tuple(42) == unit_delay(17));
      auto differentiator = compile(_1 - _1[_1]);
      CHECK(std::make_tuple(1337) == differentiator(1337));
      CHECK(std::make_tuple(42 - 1337) == differentiator(42));
      CHECK(std::make_tuple(17 - 42) == differentiator(17));
      auto integrator = compile(~(_1[_1] + _2));
      CHECK(std::make_tuple(1337) == integrator(1337));
      CHECK(std::make_tuple(1337 + 42) == integrator(42));
      CHECK(std::make_tuple(1337 + 42 + 17) == integrator(17));
   }
   {  // test_delayed_sequences, tests.cpp:139-154
      auto f = compile(_1 |= _1[_1]);
      CHECK(std::make_tuple(0) == f(1337));
      CHECK(std::make_tuple(1337) == f(0));
      CHECK(std::make_tuple(0) == f(0));
      auto g = compile(_1 |= (_1[_1], _2[_2]));
      CHECK(std::make_tuple(0, 0) == g(1337, 42));
      CHECK(std::make_tuple(1337, 0) == g(0, 0));
      CHECK(std::make_tuple(0, 42) == g(0, 0));
   }
   {  // test_feedback_expressions, tests.cpp:158-179
      auto i1 = compile(~(_1[_1] + _2));
      auto i2 = compile(~(_1[_1] + _2 |= _1));
      auto i3 = compile(~(_1 |= _1[_1] + _2));
      const int xs[3] = {1337, 42, 17}, want[3] = {1337, 1337 + 42, 1337 + 42 + 17};
      for (int k = 0; k < 3; ++k) {
         CHECK(std::make_tuple(want[k]) == i1(xs[k]));
         CHECK(std::make_tuple(want[k]) == i2(xs[k]));
         CHECK(std::make_tuple(want[k]) == i3(xs[k]));
      }
   }
   {  // comparison and logical operators through the same call protocol: a hard clipper spelled with them
      auto clip = compile(_1 * ((_1 > -0.5f) && (_1 < 0.5f)) + 0.5f * (_1 >= 0.5f) + -0.5f * (_1 <= -0.5f));
      CHECK(std::make_tuple(0.25f) == clip(0.25f));
      CHECK(std::make_tuple(0.5f) == clip(3.f));
      CHECK(std::make_tuple(-0.5f) == clip(-0.5f));
      CHECK(std::make_tuple(-0.5f) == clip(-7.f));
      auto less = compile(_1 < _2);
      CHECK(std::make_tuple(1.f) == less(1.f, 2.f));
      CHECK(std::make_tuple(0.f) == less(2.f, 2.f));
      auto either = compile(!(_1 == _2) || (_1 > 10.f));
      CHECK(std::make_tuple(0.f) == either(3.f, 3.f));
      CHECK(std::make_tuple(1.f) == either(3.f, 4.f));
      CHECK(std::make_tuple(1.f) == either(11.f, 11.f));
   }
   {  // README integrator, flowz/README.md:33-37
      auto integrator = compile(~(_1[_1] + _2));
      const int want[4] = {1, 3, 6, 10};
      int k = 0;
      for (auto x : {1, 2, 3, 4}) CHECK(std::get<0>(integrator(x)) == want[k++]);
   }
   {  // currying (flowz.hpp:1203-1212) and copy = snapshot (:1206)
      auto seq = compile(_1 + _2 * _3);
      auto c = seq(1.f, 3.f);
      CHECK(std::make_tuple(1.f + 3.f * 4.f) == c(4.f));
      auto integ = compile(~(_1[_1] + _2));
      integ(5);
      auto snap = integ;                       // independent clone with the same state
      CHECK(std::make_tuple(12) == integ(7));
      CHECK(std::make_tuple(6) == snap(1));
      CHECK(std::make_tuple(14) == integ(2));
   }
   {  // DF1 biquad of test/benchmark.cpp:18-33 driven like sum_dirac: first 8 samples equal the
      // reference's hand-written lambda (tests/golden/ref_biquad_vectors.json)
      const float b0 = 0.2, b1 = -0.3, b2 = 1.1, a1 = -0.2, a2 = 0.8;
      auto fwd = (b0 * _1 + b1 * _1[_1] + b2 * _1[_2]);
      auto bwd = ~(_2 + a1 * _1[_1] + a2 * _1[_2]);
      auto f = compile(fwd |= bwd);
      const float h[8] = {0x1.99999ap-3f, -0x1.5c28f6p-2f, 0x1.53f7cep+0f, -0x1.13405p-1f,
                          0x1.2b7fep+0f, -0x1.540034p-1f, 0x1.119986p+0f, -0x1.7d70c6p-1f};
      for (int n = 0; n < 8; ++n) CHECK(std::get<0>(f(n == 0 ? 1.f : 0.f)) == h[n]);
   }
   {  // external modulation with std::ref, flowz/README.md:42-61.  `0.1` is a double literal there:
      // the product 0.1*x and the sum are evaluated in double (C++ usual arithmetic conversions), the
      // fed-back value is truncated to float in the delay line (flowz.hpp:1245); a *= 0.9f per call
      float a = 1.f;
      auto one_pole = compile(~(std::ref(a) * _1[_1] + 0.1 * _2));
      float ar = 1.f, y1 = 0.f;
      for (int n = 0; n < 10; ++n) {
         const float x = n == 0 ? 1.f : 0.25f;
         const double want = ar * y1 + 0.1 * x;            // float*float + double*float -> double
         CHECK(std::get<0>(one_pole(x)) == static_cast<float>(want));
         y1 = static_cast<float>(want);
         a *= 0.9f;
         ar *= 0.9f;
      }
      // the all-float spelling rounds differently
      float a2 = 1.f;
      auto one_pole_f = compile(~(std::ref(a2) * _1[_1] + 0.1f * _2));
      CHECK(one_pole_f.info().n_const64 == 0 && one_pole.info().n_const64 == 1);
      // call_f64: the double result itself, as the reference's tuple<double> (tests.cpp:222,229)
      float a3 = 0.75f;
      auto one_pole_d = compile(~(std::ref(a3) * _1[_1] + 0.1 * _2));
      float yd1 = 0.f;
      bool wide = false;
      for (int n = 0; n < 10; ++n) {
         const float x = n == 0 ? 1.f : 0.25f;
         const double want = a3 * yd1 + 0.1 * x;
         const std::tuple<double> got = one_pole_d.call_f64(x);
         CHECK(std::get<0>(got) == want);
         wide = wide || want != static_cast<double>(static_cast<float>(want));
         yd1 = static_cast<float>(want);
      }
      CHECK(wide);
      auto two = compile((_1 , 0.5 * _1));                  // (float wire, double wire)
      const std::tuple<double, double> r = two.call_f64(0.3f);
      CHECK(std::get<0>(r) == static_cast<double>(0.3f) && std::get<1>(r) == 0.5 * 0.3f);
   }
   {  // std::complex<float> terminals (test/tests.cpp:206-207): the wire is complex, the frame holds (re, im)
      using cplx = std::complex<float>;
      auto rot = compile(_1 |= cplx{0.6f, 0.8f} * _1);
      CHECK(rot.info().n_out == 2 && rot.info().n_out_wires == 1);
      const std::vector<float> r = rot.call_flat(0.7f);
      const cplx want = cplx{0.6f, 0.8f} * 0.7f;
      CHECK(r.size() == 2 && r[0] == want.real() && r[1] == want.imag());
      auto chain = compile(2 * _1 |= _1 * cplx{0.3f, -0.4f} |= (_1 * cplx{0.6f, 0.8f} + 0.25f));
      volatile float xin = -0.3f, two = 2.f;           // run-time values: no compile-time (MPC, exactly rounded) folding
      const std::vector<float> q = chain.call_flat(static_cast<float>(xin));
      const float x2 = two * xin;
      volatile float b_re = 0.3f, b_im = -0.4f, c_re = 0.6f, c_im = 0.8f;
      const cplx w2 = (x2 * cplx{b_re, b_im}) * cplx{c_re, c_im} + 0.25f;
      CHECK(q.size() == 2 && q[0] == w2.real() && q[1] == w2.imag());
      bool threw = false;
      try { rot(0.7f); } catch (const flowz::error&) { threw = true; }
      CHECK(threw);
   }
   {  // block API: 96 independent integrators, 33 samples in one launch, then 7 more (state carried)
      auto f = compile(~(_1[_1] + _2));
      const int ns = 96;
      auto bank = f.bank(ns);
      std::vector<float> in(40 * ns), out(40 * ns);
      for (int t = 0; t < 40; ++t)
         for (int s = 0; s < ns; ++s) in[t * ns + s] = float(s + 1);
      bank.process_host(in.data(), out.data(), 33);
      bank.process_host(in.data() + 33 * ns, out.data() + 33 * ns, 7);
      bool ok = true;
      for (int t = 0; t < 40; ++t)
         for (int s = 0; s < ns; ++s) ok = ok && out[t * ns + s] == float((t + 1) * (s + 1));
      CHECK(ok);
   }
   {  // compile_typed(): ResultType through inputs, state and outputs -- against std::complex<float> / double computed right here
      using cplx = std::complex<float>;
      const cplx c{0.6f, 0.7f}, A{0.6f, 0.8f}, B{1.5f, -0.75f};
      auto pole = compile_typed(~(c * _1[_1] + _2));                       // complex delay line
      auto divs = compile_typed(A * _1 / (B + _1) + _1 / (B + _1));        // z / w and s / w (__divsc3)
      auto acc = compile_typed(~(_1[_1] + 1.0 * _2));                      // tests.cpp:223: the accumulator is a double
      auto dbl_in = compile_typed(_1[_1] * 0.5f + _1, {FZ_DT_F64});        // a double argument and a double delay line
      cplx z{0.f, 0.f};
      double a = 0.0, d1 = 0.0;
      bool ok = true;
      for (int t = 0; t < 50; ++t) {
         const float x = 0.25f * float((t * 7) % 11) - 1.f;
         z = c * z + x;
         ok = ok && typed_c32(pole.call_typed(x), 0) == z;
         const cplx w = B + x, r = A * x / w + x / w;
         ok = ok && typed_c32(divs.call_typed(x), 0) == r;
         a = a + 1.0 * x;
         ok = ok && typed_f64(acc.call_typed(x), 0) == a;
         const double xd = 1.0 / 3.0 + t, yd = d1 * 0.5f + xd;
         ok = ok && typed_f64(dbl_in.call_typed(xd), 0) == yd;
         d1 = xd;
      }
      CHECK(ok);
      CHECK(pole.info().typed == 1 && pole.info().n_out == 2 && pole.info().n_out_wires == 1);
   }
   {  // std::complex<double>: complex<double> state (two double delay lines), z / w and s / w (__divdc3: a data-dependent
      // branch), a complex<double> ARGUMENT -- against std::complex<double> computed right here
      using cd = std::complex<double>;
      const cd c{0.6, 0.7}, B{1.5, -0.75};
      auto pole = compile_typed(~(c * _1[_1] + _2), {FZ_DT_F64});
      auto divs = compile_typed(_1 / (B + _2) + _2 / (B + _2), {FZ_DT_CF64, FZ_DT_F64});
      cd z{0.0, 0.0};
      bool ok = true;
      for (int t = 0; t < 60; ++t) {
         const double x = 0.37 * double((t * 7) % 11) - 2.3;          // B.real() + x changes sides of |c| < |d|
         z = c * z + x;
         ok = ok && typed_c64(pole.call_typed(x), 0) == z;
         const cd w = B + x, r = z / w + x / w;
         ok = ok && typed_c64(divs.call_typed(z, x), 0) == r;
      }
      CHECK(ok);
      CHECK(pole.info().typed == 1 && pole.info().n_out == 4 && pole.info().n_out_wires == 1 && pole.info().n_in == 2);
      CHECK(pole.output_dtypes() == std::vector<uint32_t>{FZ_DT_CF64});
   }
   std::printf(failures ? "%d FAILURES\n" : "all GPU EDSL checks passed\n", failures);
   return failures ? 1 : 0;
}
