// TEST INFRASTRUCTURE: the complex_mix graph (tests/graphs.py) written with std::complex<float> and
// the natural C++ operators -- what proto::_default<eval_it> (flowz.hpp:769-772) applies to the
// evaluated children when a terminal is a std::complex<float> (test/tests.cpp:206-207).
// It pins the restatements of <complex> in flowz_oracle.c (float _Complex) and flowz_oracle.py
// (_Cplx) against the <complex> of this toolchain.  g++ -O3 -ffp-contract=off.
#include <complex>
#include <cstddef>

extern "C" void fzo_complex_mix_std(float are, float aim, float bre, float bim, const float* c /* [5] */,
                                    const float* x, ptrdiff_t xss, ptrdiff_t xts,
                                    float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   using cplx = std::complex<float>;
   const cplx A{are, aim}, B{bre, bim};
   for (long s = 0; s < n_streams; ++s) {
      float acc = 0.f;
      for (long t = 0; t < T; ++t) {
         const float x0 = x[s * xss + t * xts];
         const cplx z1 = A * x0;
         const cplx z2 = (x0 * x0) * B;
         const cplx z3 = z1 * z2;
         const cplx z4 = z3 + c[0];
         const cplx z5 = c[1] - z4;
         const cplx z6 = z5 / c[2];
         const cplx z7 = (-z6) - z1;
         const cplx z8 = z7 + z2;
         const cplx z9 = c[3] + z8;
         const cplx z10 = z9 - c[4];
         acc = acc + x0;
         float* o = y + s * yss + t * yts;
         o[0] = z10.real();
         o[1] = z10.imag();
         o[2] = acc;
      }
   }
}

// typed programs: complex state and both spellings of the complex division, with std::complex<float>
extern "C" void fzo_complex_one_pole_std(float cre, float cim, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                                         float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   using cplx = std::complex<float>;
   const cplx c{cre, cim};
   for (long s = 0; s < n_streams; ++s) {
      cplx z1{0.f, 0.f};
      for (long t = 0; t < T; ++t) {
         const cplx z = c * z1 + x[s * xss + t * xts];
         float* o = y + s * yss + t * yts;
         o[0] = z.real();
         o[1] = z.imag();
         z1 = z;
      }
   }
}

extern "C" void fzo_complex_div_mix_std(float are, float aim, float bre, float bim, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                                        float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   using cplx = std::complex<float>;
   const cplx A{are, aim}, B{bre, bim};
   for (long s = 0; s < n_streams; ++s)
      for (long t = 0; t < T; ++t) {
         const float x0 = x[s * xss + t * xts];
         const cplx z1 = A * x0;
         const cplx w = B + x0;
         const cplx r = z1 / w + x0 / w;
         float* o = y + s * yss + t * yts;
         o[0] = r.real();
         o[1] = r.imag();
      }
}

// std::complex<double>: complex state, complex / complex and scalar / complex (tests/graphs.py cdouble_resonator)
extern "C" void fzo_cdouble_resonator_std(double cre, double cim, double bre, double bim, const double* x, double* y, long n_streams,
                                          long T)
{
   using cplx = std::complex<double>;
   const cplx C{cre, cim}, B{bre, bim};
   for (long s = 0; s < n_streams; ++s) {
      cplx z1{0.0, 0.0};
      for (long t = 0; t < T; ++t) {
         const double x0 = x[t * n_streams + s];
         const cplx z = C * z1 + x0;
         const cplx w = B + x0;
         const cplx r = z / w + x0 / w;
         y[(t * n_streams + s) * 2 + 0] = r.real();
         y[(t * n_streams + s) * 2 + 1] = r.imag();
         z1 = z;
      }
   }
}
