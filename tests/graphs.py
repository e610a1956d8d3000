"""Workload graphs in the neutral s-expression notation (see oracle/flowz_oracle.py).

The BASELINE workloads (DF1/DF2/transposed biquads, cascades, the 4-parallel sum, the oscillator chain, their
coefficient sets) live in the package, zignal_amd/workloads.py; this module re-exports them and adds the
graphs only the tests use.  The same s-expression drives BOTH the oracle and the product
(zignal_amd.flowz.from_sexpr).
"""
import numpy as np

from zignal_amd.workloads import *          # noqa: F401,F403
from zignal_amd.workloads import (A1, A2, B0, B1, B2, DEL, F32, IN, PAR4_SETS, STABLE, add, bwd, bwdt, chan, delay_add_2,   # noqa: F401
                                  df1, df1_cascade, df1_param, df1t, df2, df2t, fb, fwd, fwdt, lit, lit64, mul, osc_chain, par,
                                  par4_sum, par4_sum_fanout, param, resonator_param, seq, seq_left, stable_biquad, sub)


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


def one_pole_readme(a=0.9):
    """flowz/README.md:52  ~( a*_1[_1] + 0.1*_2 ) with the README's DOUBLE literal 0.1: the product
    0.1*_2 and the sum are float64, the fed-back value is truncated to float in the delay line."""
    return fb(add(mul(lit(a), DEL(1, 1)), mul(lit64(0.1), IN(2))))


def mixed_precision_biquad():
    """DF1 whose feed-forward coefficients are double literals (b0*_1 + ... in double), feedback in float"""
    f = add(add(mul(lit64(0.05), IN(1)), mul(lit64(-0.075), DEL(1, 1))), mul(lit64(0.275), DEL(1, 2)))
    return seq(f, bwd(F32(0.2), F32(-0.8)))


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


# ---- typed programs (fz_compile_typed: ResultType through inputs, state and outputs) -------------------------
def complex_one_pole(c=(0.6, 0.7)):
    """~( c*_1[_1] + _2 ) with a std::complex<float> coefficient: the fed-back wire and its delay line are complex
    (ResultType, flowz.hpp:585-644: the absorber takes the type it meets).  oracle: fzo_complex_one_pole[_std]"""
    return fb(add(mul(litc(*c), DEL(1, 1)), IN(2)))


def complex_div_mix(A=(0.6, 0.8), B=(1.5, -0.75)):
    """z1 = A*x ; w = B + x ; z2 = z1 / w (complex / complex) ; z3 = x / w (scalar / complex) ; out = z2 + z3
    -- both spellings of __divsc3.  oracle: fzo_complex_div_mix[_std]"""
    x = IN(1)
    z1 = mul(litc(*A), x)
    w = add(litc(*B), x)
    return add(("div", z1, w), ("div", x, w))


def litc64(re, im):
    return ("litc64", float(re), float(im))


def cdouble_resonator(C=(0.6, 0.7), B=(1.5, -0.75)):
    """typed, one DOUBLE input x:  z = ~( C*_1[_1] + _2 ) with a std::complex<double> coefficient (two double delay lines),
    w = B + x,  out = z / w + x / w  -- complex state and both spellings of __divdc3 (Smith's method: a data-dependent
    branch).  oracle: fzo_cdouble_resonator[_std]"""
    z = fb(add(mul(litc64(*C), DEL(1, 1)), IN(2)))
    w = add(litc64(*B), IN(2))
    return seq(chan(z, IN(1)), add(("div", IN(1), w), ("div", IN(2), w)))


def double_accumulator():
    """test/tests.cpp:223  ~( _1[_1] + 1.0*_2 ): ResultType says the loop is double (tuple<double>); with typed state the
    accumulator itself is a double.  oracle: fzo_double_accumulator"""
    return fb(add(DEL(1, 1), mul(lit64(1.0), IN(2))))


def typed_delay_of_double():
    """test/tests.cpp:219  (_1[_1], 1.0*_1) |= _2[_1]: the delayed read of a double wire is double (ResultType)"""
    return seq(chan(DEL(1, 1), mul(lit64(1.0), IN(1))), DEL(2, 1))


def mod(k):
    return ("mod", k)


def one_pole_modulated():
    """flowz/README.md:42-61  ~( std::ref(a)*_1[_1] + _2 ) with `a` changed by the caller between calls: at block rate the
    coefficient is a sample-rate modulator (one value per sample, the same for all streams)"""
    return fb(add(mul(mod(0), DEL(1, 1)), IN(2)))


def modulated_mix():
    """two modulators in a non-recursive / recursive mix: (m0*_1 + m1*_1[_2]) |= ~(0.5*_1[_1] + m1*_2)"""
    return seq(add(mul(mod(0), IN(1)), mul(mod(1), DEL(1, 2))), fb(add(mul(lit(0.5), DEL(1, 1)), mul(mod(1), IN(2)))))


def canonical_shape_bodies():
    """The bodies X of the unary feedbacks ~X whose canonical (binary-feedback) TYPE the reference checks in test/tests.cpp:26-60.
    This library has no such type-level transform -- fz_feedback orders the loop by a delay-breaking topological sort -- so what is
    checked for these expressions is behaviour: ~X lowers (or is refused) exactly as the oracle has it, and computes the same bits."""
    _1, _2 = IN(1), IN(2)
    d11, d21 = DEL(1, 1), DEL(2, 1)
    return {
        "t26_delay": d11,
        "t27_wire_then_delay": seq(_1, d11),
        "t29_left_nested_wires": seq(seq_left(_1, _1), d11),
        "t30_right_nested_wires": seq(_1, seq(_1, d11)),
        "t32_two_delayed_pairs": seq_left(seq(_1, d11), seq(_1, d11)),
        "t33_delay_between_wires": seq_left(seq(_1, d11, _1), d11),
        "t34_wire_then_chain": seq_left(_1, seq(d11, _1, d11)),
        "t36_delay_plus_input": seq(_1, add(d11, _2)),
        "t37_offset_then_delay_plus_input": seq(add(_1, lit(2.0)), add(d11, _2)),
        "t38_offset_delay_minus_const": seq(add(_1, lit(2.0)), add(sub(d11, lit(13.0)), _2)),
        "t40_sum_delay_fanout": seq(add(_1, _2), d11, chan(_1, _1)),
        "t42_second_wire": seq(_2, _1),
        "t43_second_wire_two_delays": seq(_2, add(d11, d21)),
        "t44_second_wire_wire_two_delays": seq(_2, _1, add(d11, d21)),
        "t46_sum_then_delay": seq(add(_1, _2), d11),
        "t51_fanout_two_delays": seq(chan(_1, _1), add(d11, d21)),
        "t52_fanout_sum_delay": seq(chan(_1, _1), add(_1, _2), d11),
        "t57_nested_two_delays": fb(add(d11, d21)),
        "t59_delay_into_nested": seq(d11, fb(add(d11, _2))),
    }
