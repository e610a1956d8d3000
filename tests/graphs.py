"""Workload graphs in the neutral s-expression notation (see oracle/flowz_oracle.py).

Every graph cites the reference expression it restates.  Pure data + tiny builders; used to
drive BOTH the oracle and the product (zignal_amd.flowz.from_sexpr) with the same graph.
"""
import numpy as np

F32 = np.float32


def lit(v):
    return ("lit", float(F32(v)))


def lit64(v):
    return ("lit64", float(v))


def IN(i):
    return ("in", i)


def DEL(i, n):
    return ("del", i, n)


def add(a, b):
    return ("add", a, b)


def sub(a, b):
    return ("sub", a, b)


def mul(a, b):
    return ("mul", a, b)


def seq(*xs):
    """a |= b |= c ... ; C++ `|=` is right-associative: a |= (b |= c)."""
    r = xs[-1]
    for x in reversed(xs[:-1]):
        r = ("seq", x, r)
    return r


def seq_left(*xs):
    r = xs[0]
    for x in xs[1:]:
        r = ("seq", r, x)
    return r


def par(*xs):
    r = xs[0]
    for x in xs[1:]:
        r = ("par", r, x)
    return r


def chan(*xs):
    """(a, b, c) ; C++ comma is left-associative: ((a, b), c)."""
    r = xs[0]
    for x in xs[1:]:
        r = ("chan", r, x)
    return r


def fb(a):
    return ("fb", a)


# test/benchmark.cpp:18-23  (const float initialised from double literals)
B0, B1, B2, A1, A2 = (F32(0.2), F32(-0.3), F32(1.1), F32(-0.2), F32(0.8))


def fwd(b0=B0, b1=B1, b2=B2):
    # test/benchmark.cpp:25   b0*_1 + b1*_1[_1] + b2*_1[_2]
    return add(add(mul(lit(b0), IN(1)), mul(lit(b1), DEL(1, 1))), mul(lit(b2), DEL(1, 2)))


def bwd(a1=A1, a2=A2):
    # test/benchmark.cpp:26   ~( _2 + a1*_1[_1] + a2*_1[_2] )
    return fb(add(add(IN(2), mul(lit(a1), DEL(1, 1))), mul(lit(a2), DEL(1, 2))))


def df1(b0=B0, b1=B1, b2=B2, a1=A1, a2=A2):
    return seq(fwd(b0, b1, b2), bwd(a1, a2))          # test/benchmark.cpp:32


def df2(b0=B0, b1=B1, b2=B2, a1=A1, a2=A2):
    return seq(bwd(a1, a2), fwd(b0, b1, b2))          # test/benchmark.cpp:62


def delay_add_2():
    # test/benchmark.cpp:79   _1[_1] + _2  |=  _1[_1] + _2
    return seq(add(DEL(1, 1), IN(2)), add(DEL(1, 1), IN(2)))


def fwdt(b0=B0, b1=B1, b2=B2):
    # test/benchmark.cpp:80   ( b2*_1 , b1*_1 , b0*_1 ) |= delay_add_2
    return seq(chan(mul(lit(b2), IN(1)), mul(lit(b1), IN(1)), mul(lit(b0), IN(1))), delay_add_2())


def bwdt(a1=A1, a2=A2):
    # test/benchmark.cpp:81   ( -a2*_1 , -a1*_1 ) |= delay_add_2    (-a2 is negated in C++, before Proto)
    return seq(chan(mul(lit(-F32(a2)), IN(1)), mul(lit(-F32(a1)), IN(1))), delay_add_2())


def df1t(**kw):
    return seq(fb(bwdt()), fwdt())                    # test/benchmark.cpp:87


def df2t(**kw):
    return seq(fwdt(), fb(bwdt()))                    # test/benchmark.cpp:113


def df1_cascade(n, coeffs=None):
    """n x DF1 in series.  coeffs: list of (b0,b1,b2,a1,a2) or None."""
    if coeffs is None:
        coeffs = [STABLE] * n
    return seq(*[df1(*c) for c in coeffs])


# SURVEY 8d config 2 "stable set": b = (0.2,-0.3,1.1)*0.25, recursion a1=+0.2, a2=-0.8
STABLE = (F32(0.2 * 0.25), F32(-0.3 * 0.25), F32(1.1 * 0.25), F32(0.2), F32(-0.8))


def integrator():
    return fb(add(DEL(1, 1), IN(2)))                  # test/tests.cpp:130


def one_quad():
    # experimental_steps/multi_wires_feedback.cpp:705  ~(0.9f*_1[_1] - 0.8f*_1[_2] + _2)
    return fb(add(sub(mul(lit(0.9), DEL(1, 1)), mul(lit(0.8), DEL(1, 2))), IN(2)))


def one_quad_chain():
    return seq(one_quad(), one_quad())                # ...feedback.cpp:710-712


def cross_wire():
    # ...feedback.cpp:721  ~( (_2[_1],_3,_1[_1]) |= (.9f*_1 + _2) | (.2f*_1) )
    return fb(seq(chan(DEL(2, 1), IN(3), DEL(1, 1)),
                  par(add(mul(lit(0.9), IN(1)), IN(2)), mul(lit(0.2), IN(1)))))


def stable_biquad(r, phi, b=(0.25, -0.1, 0.15)):
    """DF1 (+a convention): poles r*exp(+-i*phi)  ->  a1 = 2 r cos(phi), a2 = -r^2."""
    a1 = F32(2.0 * r * np.cos(phi))
    a2 = F32(-(r * r))
    return (F32(b[0]), F32(b[1]), F32(b[2]), a1, a2)


PAR4_SETS = [stable_biquad(0.80, 0.4), stable_biquad(0.85, 0.9),
             stable_biquad(0.90, 1.7), stable_biquad(0.95, 2.5)]


def par4_sum():
    """(bq|bq|bq|bq) |= (_1+_2+_3+_4): config 3 primary, 4 input wires (wiring pattern of
    experimental_steps/multi_wires_with_parallel_and_delay.cpp:573-577)."""
    boxes = par(*[df1(*c) for c in PAR4_SETS])
    return seq(boxes, add(add(add(IN(1), IN(2)), IN(3)), IN(4)))


def par4_sum_fanout():
    """(_1,_1,_1,_1) |= (bq|bq|bq|bq) |= (_1+_2+_3+_4): config 3 fan-out variant (1 in / 1 out)."""
    return seq(chan(IN(1), IN(1), IN(1), IN(1)), par4_sum())


def param(k):
    return ("param", k)


def one_pole_readme(a=0.9):
    """flowz/README.md:52  ~( a*_1[_1] + 0.1*_2 ) with the README's DOUBLE literal 0.1: the product
    0.1*_2 and the sum are float64, the fed-back value is truncated to float in the delay line."""
    return fb(add(mul(lit(a), DEL(1, 1)), mul(lit64(0.1), IN(2))))


def mixed_precision_biquad():
    """DF1 whose feed-forward coefficients are double literals (b0*_1 + ... in double), feedback in float"""
    f = add(add(mul(lit64(0.05), IN(1)), mul(lit64(-0.075), DEL(1, 1))), mul(lit64(0.275), DEL(1, 2)))
    return seq(f, bwd(F32(0.2), F32(-0.8)))


def resonator_param(k):
    # oscillator = 2-pole resonator  ~( k*_1[_1] - _1[_2] + _2 )  
    return fb(add(sub(mul(param(k), DEL(1, 1)), DEL(1, 2)), IN(2)))


def df1_param(base):
    """DF1 with per-stream coefficients param(base .. base+4) = b0,b1,b2,a1,a2."""
    f = add(add(mul(param(base), IN(1)), mul(param(base + 1), DEL(1, 1))), mul(param(base + 2), DEL(1, 2)))
    b = fb(add(add(IN(2), mul(param(base + 3), DEL(1, 1))), mul(param(base + 4), DEL(1, 2))))
    return seq(f, b)


def osc_chain(n=6):
    """resonator(param 0) |= n x DF1(params 1+5j ..): config 4, C_ps = 1 + 5n."""
    return seq(resonator_param(0), *[df1_param(1 + 5 * j) for j in range(n)])


def litc(re, im):
    return ("litc", float(F32(re)), float(F32(im)))


def complex_mix(A=(0.6, 0.8), B=(0.3, -0.4), c=(0.5, 0.25, 1.5, -0.125, 0.75)):
    """std::complex<float> wires (ResultType, test/tests.cpp:206-207): every supported operator once,
    next to a real integrator wire.  Output frame: (re, im, integrator).  oracle/flowz_oracle.c: fzo_complex_mix"""
    x = IN(1)
    z1 = mul(litc(*A), x)
    z2 = mul(mul(x, x), litc(*B))
    z3 = mul(z1, z2)
    z4 = add(z3, lit(c[0]))
    z5 = sub(lit(c[1]), z4)
    z6 = ("div", z5, lit(c[2]))
    z7 = sub(("neg", z6), z1)
    z8 = add(z7, z2)
    z9 = add(lit(c[3]), z8)
    z10 = sub(z9, lit(c[4]))
    return chan(z10, fb(add(DEL(1, 1), IN(2))))
