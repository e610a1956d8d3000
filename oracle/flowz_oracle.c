/* CPU ORACLE (compiled scalar closures) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  Nothing under zignal_amd/ or include/ links or calls it.
 *
 * What it restates: what the reference's compile()-callable computes per call for the
 * BASELINE workload graphs, i.e. stateful_lambda::operator() (flowz/flowz.hpp:1225-1229)
 * after template inlining: a scalar closure with zero-initialised float state
 * (flowz.hpp:1245), one IEEE float32 rounding per operator in the expression tree's
 * association order (proto::_default, flowz.hpp:769-772), shift-register delay lines
 * (rotate_push_back, flowz.hpp:130-148).  The per-sample recurrences are the normative
 * traces of SURVEY.md 3.5, each derived from the cited reference expression.
 * "Mode A, reference-faithful" (BASELINE.md 3): one closure per stream, one call per sample.
 *
 * Build: gcc -O3 -ffp-contract=off (no -march, no -ffast-math), see oracle/Makefile --
 * the reference's flags are `-O3 --std=c++1y` without -march (CMakeLists.txt:18).
 *
 * Parity status: PINNED through tests/test_oracle_golden.py + tests/test_oracle_c.py:
 * each closure here is bit-compared with the generic Python oracle (which is itself pinned
 * to test/tests.cpp known answers) and with the reference's own hand-written lambdas
 * built from the reference sources (oracle/build_ref.sh, tests/golden/ref_biquad_vectors.json).
 *
 * Addressing: element (stream s, time t, wire w) of a signal lives at
 *     p[s*ss + t*ts + w]           (ss/ts in floats)
 * so the same code walks time-major device-style frames (ss = n_wires, ts = n_streams*n_wires)
 * and per-stream contiguous arrays (ss = T*n_wires, ts = n_wires).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct { float b0, b1, b2, a1, a2; } fzo_coef;

/* ---- DF1  `fwd |= bwd`  (test/benchmark.cpp:25-26,32):
 *   f = (b0*x + b1*x1) + b2*x2 ;  y = (f + a1*y1) + a2*y2                               */
typedef struct { float x1, x2, y1, y2; } fzo_df1_state;

static inline float df1_call(const fzo_coef* c, fzo_df1_state* s, float x0)
{
   float f = (c->b0 * x0 + c->b1 * s->x1) + c->b2 * s->x2;
   float y = (f + c->a1 * s->y1) + c->a2 * s->y2;
   s->x2 = s->x1; s->x1 = x0;          /* rotate_push_back, front-panel / sequence node */
   s->y2 = s->y1; s->y1 = y;           /* rotate_push_back, binary_feedback node :1067  */
   return y;
}

#define FZO_MAX_STAGES 64

/* n_stage x DF1 in series.  coef: [n_stage] uniform, or per-stream when coef_ss != 0
 * (coefficient j of stage k for stream s at coef_ps[(k*5+j)*coef_ss + s]).              */
/* A closure of the reference is a TYPE: its stages are inlined and their delay lines live in registers (the compile()-callable like the
 * hand-written lambdas of test/benchmark.cpp:35-47).  A cascade of a depth known at compile time gets the same treatment here -- the
 * stage loop unrolled, state and coefficients in locals --, so that what bench.py times as the CPU baseline is not held back by a
 * run-time stage loop over state in memory (round 6: 1.9 x per thread; the arithmetic is df1_call either way).              */
#define FZO_CASCADE_FIXED(N)                                                                                                  \
   static void fzo_cascade_##N(const fzo_coef* coef, const float* xp, ptrdiff_t xts, float* yp, ptrdiff_t yts, long T)          \
   {                                                                                                                          \
      fzo_df1_state st[N] = {{0}};                                                                                            \
      fzo_coef c[N];                                                                                                          \
      for (int k = 0; k < N; ++k) c[k] = coef[k];                                                                             \
      for (long t = 0; t < T; ++t) {                                                                                          \
         float v = xp[t * xts];                                                                                               \
         _Pragma("GCC unroll 8") for (int k = 0; k < N; ++k) v = df1_call(&c[k], &st[k], v);                                 \
         yp[t * yts] = v;                                                                                                     \
      }                                                                                                                       \
   }
FZO_CASCADE_FIXED(1)
FZO_CASCADE_FIXED(2)
FZO_CASCADE_FIXED(4)
FZO_CASCADE_FIXED(6)
FZO_CASCADE_FIXED(8)

void fzo_df1_cascade(const fzo_coef* coef, int n_stage,
                     const float* x, ptrdiff_t xss, ptrdiff_t xts,
                     float* y, ptrdiff_t yss, ptrdiff_t yts,
                     long n_streams, long T)
{
   void (*fixed)(const fzo_coef*, const float*, ptrdiff_t, float*, ptrdiff_t, long) =
      n_stage == 6 ? fzo_cascade_6 : n_stage == 1 ? fzo_cascade_1 : n_stage == 2 ? fzo_cascade_2 : n_stage == 4 ? fzo_cascade_4 : n_stage == 8 ? fzo_cascade_8 : 0;
   if (fixed) {
      for (long s = 0; s < n_streams; ++s) fixed(coef, x + s * xss, xts, y + s * yss, yts, T);
      return;
   }
   for (long s = 0; s < n_streams; ++s) {
      fzo_df1_state st[FZO_MAX_STAGES] = {{0}};
      const float* xp = x + s * xss;
      float* yp = y + s * yss;
      for (long t = 0; t < T; ++t) {
         float v = xp[t * xts];
         for (int k = 0; k < n_stage; ++k) v = df1_call(&coef[k], &st[k], v);
         yp[t * yts] = v;
      }
   }
}

/* "Mode B" (SURVEY 8d): the same closures with the streams in SoA form so that the compiler can
 * vectorise ACROSS streams (the arithmetic per stream is untouched: same operations, same order, no
 * FMA) -- a CPU stronger than the reference's scalar closure, reported separately by bench.py.
 * Time-major frames x[t][s], y[t][s]; `work` holds 4 * n_stage * W floats of state (W streams per pass).
 * target_clones: the library is built in one container and runs on another host.                    */
#define FZO_SOA_W 256
__attribute__((target_clones("avx512f", "avx2", "default")))
void fzo_df1_cascade_soa(const fzo_coef* coef, int n_stage, const float* x, float* y, long n_streams, long T)
{
   float st[FZO_MAX_STAGES][4][FZO_SOA_W];
   float v[FZO_SOA_W];
   for (long s0 = 0; s0 < n_streams; s0 += FZO_SOA_W) {
      const long w = n_streams - s0 < FZO_SOA_W ? n_streams - s0 : FZO_SOA_W;
      for (int k = 0; k < n_stage; ++k)
         for (int j = 0; j < 4; ++j)
            for (long i = 0; i < w; ++i) st[k][j][i] = 0.f;
      for (long t = 0; t < T; ++t) {
         const float* xp = x + t * n_streams + s0;
         float* yp = y + t * n_streams + s0;
         for (long i = 0; i < w; ++i) v[i] = xp[i];
         for (int k = 0; k < n_stage; ++k) {
            const fzo_coef c = coef[k];
            float* x1 = st[k][0]; float* x2 = st[k][1]; float* y1 = st[k][2]; float* y2 = st[k][3];
            for (long i = 0; i < w; ++i) {
               const float x0 = v[i];
               const float f = (c.b0 * x0 + c.b1 * x1[i]) + c.b2 * x2[i];
               const float o = (f + c.a1 * y1[i]) + c.a2 * y2[i];
               x2[i] = x1[i]; x1[i] = x0;
               y2[i] = y1[i]; y1[i] = o;
               v[i] = o;
            }
         }
         for (long i = 0; i < w; ++i) yp[i] = v[i];
      }
   }
}

/* ---- DF2  `bwd |= fwd` (test/benchmark.cpp:62):
 *   u = (x + a1*u1) + a2*u2 ;  y = (b0*u + b1*u1) + b2*u2                               */
void fzo_df2(const fzo_coef* c, const float* x, ptrdiff_t xss, ptrdiff_t xts,
             float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float u1 = 0.f, u2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float x0 = x[s * xss + t * xts];
         float u = (x0 + c->a1 * u1) + c->a2 * u2;
         float o = (c->b0 * u + c->b1 * u1) + c->b2 * u2;
         u2 = u1; u1 = u;
         y[s * yss + t * yts] = o;
      }
   }
}

/* ---- DF1T `~bwdt |= fwdt` (test/benchmark.cpp:79-81,87); na1 = -a1, na2 = -a2 are
 *   negated in C++ before they become terminals:
 *   u = w1 + x ; y = v1 + b0*u ; v1' = v2 + b1*u ; v2' = b2*u ; w1' = w2 + na1*u ; w2' = na2*u */
void fzo_df1t(const fzo_coef* c, const float* x, ptrdiff_t xss, ptrdiff_t xts,
              float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   const float na1 = -c->a1, na2 = -c->a2;
   for (long s = 0; s < n_streams; ++s) {
      float v1 = 0.f, v2 = 0.f, w1 = 0.f, w2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float x0 = x[s * xss + t * xts];
         float u = w1 + x0;
         float o = v1 + c->b0 * u;
         float nv1 = v2 + c->b1 * u, nv2 = c->b2 * u;
         float nw1 = w2 + na1 * u, nw2 = na2 * u;
         v1 = nv1; v2 = nv2; w1 = nw1; w2 = nw2;
         y[s * yss + t * yts] = o;
      }
   }
}

/* ---- DF2T as the Flowz GRAPH `fwdt |= ~bwdt` (test/benchmark.cpp:113) -- NOT the merged
 * two-state lambda of :116-126, which rounds differently:
 *   f = v1 + b0*x ; v1' = v2 + b1*x ; v2' = b2*x ; y = w1 + f ; w1' = w2 + na1*y ; w2' = na2*y */
void fzo_df2t_flowz(const fzo_coef* c, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                    float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   const float na1 = -c->a1, na2 = -c->a2;
   for (long s = 0; s < n_streams; ++s) {
      float v1 = 0.f, v2 = 0.f, w1 = 0.f, w2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float x0 = x[s * xss + t * xts];
         float f = v1 + c->b0 * x0;
         float nv1 = v2 + c->b1 * x0, nv2 = c->b2 * x0;
         float o = w1 + f;
         float nw1 = w2 + na1 * o, nw2 = na2 * o;
         v1 = nv1; v2 = nv2; w1 = nw1; w2 = nw2;
         y[s * yss + t * yts] = o;
      }
   }
}

/* ---- integrator `~(_1[_1] + _2)` (test/tests.cpp:130): y = y1 + x */
void fzo_integrator(const float* x, ptrdiff_t xss, ptrdiff_t xts,
                    float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float y1 = 0.f;
      for (long t = 0; t < T; ++t) { y1 = y1 + x[s * xss + t * xts]; y[s * yss + t * yts] = y1; }
   }
}

/* ---- one_quad `~(0.9f*_1[_1] - 0.8f*_1[_2] + _2)`
 * (experimental_steps/multi_wires_feedback.cpp:705): y = (c1*y1 - c2*y2) + x */
void fzo_one_quad(float c1, float c2, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                  float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float y1 = 0.f, y2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float o = (c1 * y1 - c2 * y2) + x[s * xss + t * xts];
         y2 = y1; y1 = o;
         y[s * yss + t * yts] = o;
      }
   }
}

/* ---- cross_wire `~( (_2[_1],_3,_1[_1]) |= (.9f*_1 + _2) | (.2f*_1) )`
 * (...feedback.cpp:721): u1' = c1*u2 + x ; u2' = c2*u1 ; two output wires (new u1, u2) */
void fzo_cross_wire(float c1, float c2, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                    float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float u1 = 0.f, u2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float n1 = c1 * u2 + x[s * xss + t * xts];
         float n2 = c2 * u1;
         u1 = n1; u2 = n2;
         y[s * yss + t * yts + 0] = n1;
         y[s * yss + t * yts + 1] = n2;
      }
   }
}

/* ---- config 3: (bq|bq|bq|bq) |= (_1+_2+_3+_4), 4 DF1 boxes, frames of 4 input wires
 * (wiring pattern experimental_steps/multi_wires_with_parallel_and_delay.cpp:573-577):
 *   y = ((q0 + q1) + q2) + q3.   fanout != 0: all four boxes read wire 0 (1-wire frames). */
void fzo_par4_sum(const fzo_coef* coef4, int fanout,
                  const float* x, ptrdiff_t xss, ptrdiff_t xts,
                  float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      fzo_df1_state st[4] = {{0}};
      for (long t = 0; t < T; ++t) {
         const float* f = x + s * xss + t * xts;
         float q0 = df1_call(&coef4[0], &st[0], f[0]);
         float q1 = df1_call(&coef4[1], &st[1], fanout ? f[0] : f[1]);
         float q2 = df1_call(&coef4[2], &st[2], fanout ? f[0] : f[2]);
         float q3 = df1_call(&coef4[3], &st[3], fanout ? f[0] : f[3]);
         y[s * yss + t * yts] = ((q0 + q1) + q2) + q3;
      }
   }
}

/* ---- config 4: resonator `~(k*_1[_1] - _1[_2] + _2)` |= n x DF1, per-stream coefficients.
 * params: [1 + 5*n_stage][n_streams] planar (param j of stream s at params[j*pss + s]):
 * param 0 = k, then b0,b1,b2,a1,a2 per stage.   r = (k*r1 - r2) + x                      */
void fzo_osc_chain(const float* params, ptrdiff_t pss, int n_stage,
                   const float* x, ptrdiff_t xss, ptrdiff_t xts,
                   float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      fzo_coef c[FZO_MAX_STAGES];
      fzo_df1_state st[FZO_MAX_STAGES] = {{0}};
      const float k = params[s];
      for (int j = 0; j < n_stage; ++j) {
         const float* p = params + (ptrdiff_t)(1 + 5 * j) * pss + s;
         c[j].b0 = p[0]; c[j].b1 = p[pss]; c[j].b2 = p[2 * pss]; c[j].a1 = p[3 * pss]; c[j].a2 = p[4 * pss];
      }
      float r1 = 0.f, r2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float v = (k * r1 - r2) + x[s * xss + t * xts];
         r2 = r1; r1 = v;
         for (int j = 0; j < n_stage; ++j) v = df1_call(&c[j], &st[j], v);
         y[s * yss + t * yts] = v;
      }
   }
}

/* ---- one-pole of flowz/README.md:52  `~( a*_1[_1] + 0.1*_2 )`: `0.1` is a DOUBLE literal, so by the
 * usual arithmetic conversions of the built-in operators (flowz.hpp:769-772) 0.1*x and the sum are
 * double; the fed-back value is truncated to float when pushed into the delay line (:130-137, :1245) */
void fzo_one_pole_readme(float a, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                         float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float y1 = 0.f;
      for (long t = 0; t < T; ++t) {
         float x0 = x[s * xss + t * xts];
         double o = a * y1 + 0.1 * x0;          /* float*float + double*float -> double */
         y1 = (float)o;
         y[s * yss + t * yts] = (float)o;        /* output frames are float32 */
      }
   }
}

/* the same closure, results handed out as the C++ type of the output wire (double: tuple<double> of
 * tests.cpp:222,229): what the FZ_VF_OUT_F64 frames hold                                           */
void fzo_one_pole_readme_f64out(float a, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                                double* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float y1 = 0.f;
      for (long t = 0; t < T; ++t) {
         double o = a * y1 + 0.1 * x[s * xss + t * xts];
         y1 = (float)o;
         y[s * yss + t * yts] = o;
      }
   }
}

/* ---- DF1 whose feed-forward coefficients are double literals: f in double, feedback products in
 * float, y = (f + (double)(a1*y1)) + (double)(a2*y2) in double, truncated to float for the delay
 * line and the frame (tests/graphs.py: mixed_precision_biquad)                                   */
void fzo_mixed_precision_biquad(double b0, double b1, double b2, float a1, float a2,
                                const float* x, ptrdiff_t xss, ptrdiff_t xts,
                                float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float x0 = x[s * xss + t * xts];
         double f = (b0 * x0 + b1 * x1) + b2 * x2;
         double o = (f + a1 * y1) + a2 * y2;     /* a1*y1 is a float product, then promoted */
         x2 = x1; x1 = x0;
         y2 = y1; y1 = (float)o;
         y[s * yss + t * yts] = (float)o;
      }
   }
}

void fzo_mixed_precision_biquad_f64out(double b0, double b1, double b2, float a1, float a2,
                                       const float* x, ptrdiff_t xss, ptrdiff_t xts,
                                       double* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
      for (long t = 0; t < T; ++t) {
         float x0 = x[s * xss + t * xts];
         double f = (b0 * x0 + b1 * x1) + b2 * x2;
         double o = (f + a1 * y1) + a2 * y2;
         x2 = x1; x1 = x0;
         y2 = y1; y1 = (float)o;
         y[s * yss + t * yts] = o;
      }
   }
}

/* ---- std::complex<float> wires (ResultType cases test/tests.cpp:206-207), with C99 `float _Complex`
 * -- the very representation std::complex<float> wraps (_M_value) and whose operators it forwards to:
 * complex (op) real touches the parts without cross terms, complex * complex is __mulsc3.
 * tests/graphs.py: complex_mix -- every supported operator once, next to a real integrator wire:
 *   z1 = A*x   z2 = (x*x)*B   z3 = z1*z2   z4 = z3 + c0   z5 = c1 - z4   z6 = z5 / c2
 *   z7 = (-z6) - z1   z8 = z7 + z2   z9 = c3 + z8   z10 = z9 - c4 ;   frame = (re z10, im z10, integ) */
void fzo_complex_mix(float are, float aim, float bre, float bim, const float* c /* [5] */,
                     const float* x, ptrdiff_t xss, ptrdiff_t xts,
                     float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   const float _Complex A = __builtin_complex(are, aim), B = __builtin_complex(bre, bim);
   for (long s = 0; s < n_streams; ++s) {
      float acc = 0.f;
      for (long t = 0; t < T; ++t) {
         const float x0 = x[s * xss + t * xts];
         float _Complex z1 = A;  z1 *= x0;                   /* operator*(complex, T): r = z; r *= s   */
         float _Complex z2 = B;  z2 *= (x0 * x0);            /* operator*(T, complex): r = z; r *= s   */
         float _Complex z3 = z1; z3 *= z2;                   /* complex *= complex: __mulsc3            */
         float _Complex z4 = z3; z4 += c[0];
         float _Complex z5 = -z4; z5 += c[1];                /* operator-(T, complex): r = -z; r += s   */
         float _Complex z6 = z5; z6 /= c[2];
         float _Complex z7 = -z6; z7 -= z1;
         float _Complex z8 = z7; z8 += z2;
         float _Complex z9 = z8; z9 += c[3];                 /* operator+(T, complex): r = z; r += s    */
         float _Complex z10 = z9; z10 -= c[4];
         acc = acc + x0;                                     /* ~(_1[_1] + _2)                          */
         float* o = y + s * yss + t * yts;
         o[0] = __real__ z10;
         o[1] = __imag__ z10;
         o[2] = acc;
      }
   }
}

/* ---- typed programs (fz_compile_typed): ResultType carried through state (flowz.hpp:585-644) -------------------
 * tests/graphs.py: complex_one_pole  ~( c*_1[_1] + _2 ): z = c*z1 + x, complex state.  frame = (re, im)           */
void fzo_complex_one_pole(float cre, float cim, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                          float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   const float _Complex c = __builtin_complex(cre, cim);
   for (long s = 0; s < n_streams; ++s) {
      float _Complex z1 = 0.f;
      for (long t = 0; t < T; ++t) {
         float _Complex z = c; z *= z1;                       /* complex * complex: __mulsc3 */
         z += x[s * xss + t * xts];                           /* complex + float: real part only */
         float* o = y + s * yss + t * yts;
         o[0] = __real__ z;
         o[1] = __imag__ z;
         z1 = z;
      }
   }
}

/* std::complex<float> division as g++ links it: operator/= forwards to the _Complex float divide, i.e. libgcc's
 * __divsc3, which "handles float with double precision" (libgcc2.c): the simple formula on the widened parts, one
 * rounding to float at the end.  Spelled out here because gcc's C front end lowers a `float _Complex` divide with a
 * real-only numerator differently from the library call the C++ operator makes; oracle/complex_std.cpp is the pin.
 * (NaN-recovery branch of __divsc3 not restated: finite values, nonzero divisor.)                                   */
static float _Complex cdiv_wide(float _Complex z, float _Complex w)
{
   const double aa = __real__ z, bb = __imag__ z, cc = __real__ w, dd = __imag__ w;
   const double denom = (cc * cc) + (dd * dd);
   const float x = (float)(((aa * cc) + (bb * dd)) / denom);
   const float y = (float)(((bb * cc) - (aa * dd)) / denom);
   return __builtin_complex(x, y);
}

/* tests/graphs.py: complex_div_mix  z1 = A*x ; w = B + x ; out = z1/w + x/w   (both are __divsc3)               */
void fzo_complex_div_mix(float are, float aim, float bre, float bim, const float* x, ptrdiff_t xss, ptrdiff_t xts,
                         float* y, ptrdiff_t yss, ptrdiff_t yts, long n_streams, long T)
{
   const float _Complex A = __builtin_complex(are, aim), B = __builtin_complex(bre, bim);
   for (long s = 0; s < n_streams; ++s)
      for (long t = 0; t < T; ++t) {
         const float x0 = x[s * xss + t * xts];
         float _Complex z1 = A;  z1 *= x0;
         float _Complex w = B;   w += x0;
         float _Complex z2 = cdiv_wide(z1, w);
         float _Complex z3 = cdiv_wide(__builtin_complex(x0, 0.f), w);   /* operator/(T, complex): r = s; r /= w */
         float _Complex r = z2;  r += z3;
         float* o = y + s * yss + t * yts;
         o[0] = __real__ r;
         o[1] = __imag__ r;
      }
}

/* tests/graphs.py: double_accumulator  ~( _1[_1] + 1.0*_2 ) with a DOUBLE accumulator (typed state); y: doubles   */
void fzo_double_accumulator(const float* x, ptrdiff_t xss, ptrdiff_t xts, double* y, ptrdiff_t yss, ptrdiff_t yts,
                            long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      double acc = 0.0;
      for (long t = 0; t < T; ++t) {
         acc = acc + 1.0 * x[s * xss + t * xts];
         y[s * yss + t * yts] = acc;
      }
   }
}

/* std::complex<double> division = __divdc3 (libgcc2.c): Smith's method.  The scaling branches newer libgcc adds for
 * extreme magnitudes (|d| >= DBL_MAX/2, tiny operands, subnormal ratio) and the NaN recovery are not restated;
 * oracle/complex_std.cpp (std::complex<double> compiled by g++) is the pin.                                       */
static void cdiv_smith(double a, double b, double c, double d, double* x, double* y)
{
   if (fabs(c) < fabs(d)) {
      const double ratio = c / d, denom = (c * ratio) + d;
      *x = ((a * ratio) + b) / denom;
      *y = ((b * ratio) - a) / denom;
   } else {
      const double ratio = d / c, denom = (d * ratio) + c;
      *x = ((b * ratio) + a) / denom;
      *y = (b - (a * ratio)) / denom;
   }
}

/* tests/graphs.py: cdouble_resonator (typed, double input x)
 *    z = C*z[-1] + x  (std::complex<double> state) ;  w = B + x ;  out = z/w + x/w   (both are __divdc3)
 * x: [T][n_streams] doubles, y: [T][n_streams][2] doubles (re, im)                                                */
void fzo_cdouble_resonator(double cre, double cim, double bre, double bim, const double* x, double* y, long n_streams, long T)
{
   for (long s = 0; s < n_streams; ++s) {
      double zr = 0.0, zi = 0.0;
      for (long t = 0; t < T; ++t) {
         const double x0 = x[t * n_streams + s];
         const double ac = cre * zr, bd = cim * zi, ad = cre * zi, bc = cim * zr;   /* __muldc3, finite values */
         zr = (ac - bd) + x0;                                                          /* z += s: real part only */
         zi = ad + bc;
         const double wr = bre + x0, wi = bim;
         double q1r, q1i, q2r, q2i;
         cdiv_smith(zr, zi, wr, wi, &q1r, &q1i);
         cdiv_smith(x0, 0.0, wr, wi, &q2r, &q2i);                                      /* r = s; r /= w */
         y[(t * n_streams + s) * 2 + 0] = q1r + q2r;
         y[(t * n_streams + s) * 2 + 1] = q1i + q2i;
      }
   }
}

/* ---- RBJ low-pass coefficients, reactive_equations/reactive_filter_coeff.cpp:38-58, with the
 * reference's types: every PARAMETER is float, `1.` `2.` are double literals; std::cos/std::sin of a
 * float.  PARITY UNPINNED against the reference (reactive_expressions needs Boost).  sin and cos are NOT
 * taken from a libm -- two libms need not agree in the last bit of a double, and the device has its own --
 * but from fzo_sincos_f32: reduction by multiples of pi/2 and Taylor polynomials, IEEE double operations in a
 * fixed order (built with -ffp-contract=off), the result rounded to float once.  The device generator spells
 * the same operations; tests compare it with this BIT FOR BIT, and this with glibc's sinf / cosf -- what the
 * reference's std::sin(float) is on this box -- (fzo_rbj_lowpass_libmf below).
 * raw6: [6][n] a0 a1 a2 b0 b1 b2;  df1: [5][n] b0/a0 b1/a0 b2/a0 -a1/a0 -a2/a0 (either may be NULL) */
void fzo_sincos_f32(float xf, float* sn, float* cs)
{
   static const double S[8] = {-0x1.5555555555555p-3, 0x1.1111111111111p-7, -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
                               -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33, -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49};   /* (-1)^k / (2k+1)! */
   static const double C[9] = {-0x1.0000000000000p-1, 0x1.5555555555555p-5, -0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-16, -0x1.27e4fb7789f5cp-22,
                               0x1.1eed8eff8d898p-29, -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45, -0x1.6827863b97d97p-53};   /* (-1)^k / (2k)! */
   const double x = (double)xf;
   if (!(fabs(x) < 0x1p20)) {
      *sn = *cs = NAN;
      return;
   }
   const double t = x * 0x1.45f306dc9c883p-1;                /* 2/pi */
   const int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
   const double kd = (double)k;
   double r = x - kd * 0x1.921fb54400000p+0;                 /* pi/2 in three parts; the first product is exact */
   r = r - kd * 0x1.0b4611a600000p-34;
   r = r - kd * 0x1.3198a2e037073p-69;
   const double z = r * r;
   double ps = S[7], pc = C[8];
   for (int i = 6; i >= 0; --i) ps = S[i] + z * ps;
   for (int i = 7; i >= 0; --i) pc = C[i] + z * pc;
   const double s = r + r * (z * ps), c = 1.0 + z * pc;
   switch (k & 3) {
      case 0: *sn = (float)s; *cs = (float)c; break;
      case 1: *sn = (float)c; *cs = (float)-s; break;
      case 2: *sn = (float)-s; *cs = (float)-c; break;
      default: *sn = (float)-c; *cs = (float)s; break;
   }
}

void fzo_rbj_lowpass(const float* freq, const float* q, float sr, long n, float* raw6, float* df1)
{
   const float two_pi = 8. * atan(1.);
   for (long s = 0; s < n; ++s) {
      const float w0 = two_pi * freq[s] / sr;
      float sinw0, cosw0;
      fzo_sincos_f32(w0, &sinw0, &cosw0);
      const float alpha = sinw0 / (2. * q[s]);
      const float b0 = (1. - cosw0) / 2.;
      const float b1 = 1. - cosw0;
      const float b2 = (1. - cosw0) / 2.;
      const float a0 = 1. + alpha;
      const float a1 = -2. * cosw0;
      const float a2 = 1. - alpha;
      if (raw6) {
         raw6[0 * n + s] = a0; raw6[1 * n + s] = a1; raw6[2 * n + s] = a2;
         raw6[3 * n + s] = b0; raw6[4 * n + s] = b1; raw6[5 * n + s] = b2;
      }
      if (df1) {
         df1[0 * n + s] = b0 / a0; df1[1 * n + s] = b1 / a0; df1[2 * n + s] = b2 / a0;
         df1[3 * n + s] = -a1 / a0; df1[4 * n + s] = -a2 / a0;
      }
   }
}

/* sin / cos of n floats: the checker's polynomial pair (libm = 0) or glibc's sinf / cosf (libm = 1) */
void fzo_sincos_array(const float* x, long n, int libm, float* sn, float* cs)
{
   for (long i = 0; i < n; ++i) {
      if (libm) {
         sn[i] = sinf(x[i]);
         cs[i] = cosf(x[i]);
      } else {
         fzo_sincos_f32(x[i], &sn[i], &cs[i]);
      }
   }
}

/* the reference's own spelling, std::cos / std::sin on float (cosf / sinf), for the 1-ULP claim */
void fzo_rbj_lowpass_libmf(const float* freq, const float* q, float sr, long n, float* raw6)
{
   const float two_pi = 8. * atan(1.);
   for (long s = 0; s < n; ++s) {
      const float w0 = two_pi * freq[s] / sr;
      const float cosw0 = cosf(w0);
      const float alpha = sinf(w0) / (2. * q[s]);
      raw6[0 * n + s] = 1. + alpha; raw6[1 * n + s] = -2. * cosw0; raw6[2 * n + s] = 1. - alpha;
      raw6[3 * n + s] = (1. - cosw0) / 2.; raw6[4 * n + s] = 1. - cosw0; raw6[5 * n + s] = (1. - cosw0) / 2.;
   }
}

/* ---- synthetic input, identical to oracle/flowz_oracle.py: synth_input -------------- */
static inline uint32_t fmix32(uint32_t h)
{
   h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
   return h;
}

void fzo_synth_fill(float* dst, ptrdiff_t ss, ptrdiff_t ts, uint32_t seed,
                    uint64_t stream0, long n_streams, long T, int n_wires, uint64_t t0)
{
   for (long s = 0; s < n_streams; ++s)
      for (long t = 0; t < T; ++t)
         for (int w = 0; w < n_wires; ++w) {
            uint64_t sid = (stream0 + (uint64_t)s) * (uint64_t)n_wires + (uint64_t)w;
            uint32_t h = seed ^ (uint32_t)(sid * 0x9E3779B9ull) ^ (uint32_t)((t0 + (uint64_t)t) * 0x85EBCA6Bull);
            h = fmix32(fmix32(h));
            dst[s * ss + t * ts + w] = (float)(int32_t)(h >> 8) * 0x1p-23f - 1.0f;
         }
}
