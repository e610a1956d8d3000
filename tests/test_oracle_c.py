"""The compiled scalar closures (oracle/flowz_oracle.c) vs the generic Python oracle and the
reference-built golden vectors."""
import json
import os

import numpy as np
import pytest

import graphs as G
import workloads as W
from oracle import coracle as C
from oracle import flowz_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "ref_biquad_vectors.json")))
REFC = (G.B0, G.B1, G.B2, G.A1, G.A2)


def bits(hexlist):
    return np.array([int(h, 16) for h in hexlist], np.uint32).view(np.float32)


def same(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("drive", ["dirac", "noise"])
def test_c_closures_vs_reference_lambdas(drive):
    x = bits(REF["inputs"][drive])[:, None, None]
    out = REF["outputs"][drive]
    assert same(C.df1_cascade([REFC], x)[:, 0, 0], bits(out["df1"]))
    assert same(C.df1_cascade([REFC, REFC], x)[:, 0, 0], bits(out["df1x2"]))
    assert same(C.df1_cascade([REFC] * 6, x)[:, 0, 0], bits(out["df1x6"]))       # six reference closures in series (a composition: oracle/ref_harness.inc)
    assert same(C.df2(REFC, x)[:, 0, 0], bits(out["df2"]))
    assert same(C.df1t(REFC, x)[:, 0, 0], bits(out["df1t"]))
    y = C.cross_wire(x)
    assert same(y[:, 0, 0], bits(out["xwire0"])) and same(y[:, 0, 1], bits(out["xwire1"]))


def test_c_synth_matches_python():
    a = C.synth_fill(20160512, 5, 7, 33, n_wires=4, t0=3)
    b = O.synth_input(20160512, np.arange(5, 12), 33, n_wires=4, t0=3)
    assert same(a, b)
    a = C.synth_fill(1, 0, 3, 16, stream_major=True)
    assert same(a.transpose(1, 0, 2), O.synth_input(1, np.arange(3), 16))


NS, T = 5, 300


def test_c_vs_python_generic_oracle():
    x = O.synth_input(99, np.arange(NS), T)
    assert same(C.df1_cascade([G.STABLE] * 6, x), O.compile(G.df1_cascade(6), NS).run(x))
    assert same(C.df2(G.STABLE, x), O.compile(G.df2(*G.STABLE), NS).run(x))
    assert same(C.df2t_flowz(REFC, x), O.compile(G.df2t(), NS).run(x))
    assert same(C.df1t(REFC, x), O.compile(G.df1t(), NS).run(x))
    assert same(C.integrator(x), O.compile(G.integrator(), NS).run(x))
    assert same(C.one_quad(x), O.compile(G.one_quad(), NS).run(x))
    assert same(C.cross_wire(x), O.compile(G.cross_wire(), NS).run(x))


def test_c_par4_vs_python():
    x4 = O.synth_input(5, np.arange(NS), T, n_wires=4)
    assert same(C.par4_sum(G.PAR4_SETS, x4), O.compile(G.par4_sum(), NS).run(x4))
    x1 = O.synth_input(5, np.arange(NS), T)
    assert same(C.par4_sum(G.PAR4_SETS, x1, fanout=True), O.compile(G.par4_sum_fanout(), NS).run(x1))


def test_c_osc_chain_vs_python():
    P = W.osc_chain_params(20160513, np.arange(NS))
    x = np.zeros((T, NS, 1), np.float32)
    x[0] = 1.0
    got = C.osc_chain(P, x)
    want = O.compile(G.osc_chain(6), NS, params=P).run(x)
    assert same(got, want)
    assert np.isfinite(got).all() and np.abs(got).max() > 0


def test_vectorised_across_streams_variant_is_bit_identical():
    """"Mode B" CPU baseline: SoA state, compiler-vectorised over streams -- the same bits as the scalar closures."""
    x = O.synth_input(5, np.arange(700), 96)                       # ragged: 2 full passes of 256 + 188
    coefs = [G.STABLE, G.PAR4_SETS[0], G.PAR4_SETS[1], G.PAR4_SETS[2], G.STABLE, G.PAR4_SETS[3]]
    assert same(C.df1_cascade_soa(coefs, x), C.df1_cascade(coefs, x))


def test_stream_major_layout_equivalent():
    x = O.synth_input(3, np.arange(4), 50)
    a = C.df1_cascade([G.STABLE] * 6, x)
    b = C.df1_cascade([G.STABLE] * 6, np.ascontiguousarray(x.transpose(1, 0, 2)), stream_major=True)
    assert same(a, b.transpose(1, 0, 2))


def test_stable_set_stays_bounded_over_block():
    x = O.synth_input(20160512, np.arange(8), 4096)
    y = C.df1_cascade([G.STABLE] * 6, x)
    assert np.isfinite(y).all() and np.abs(y).max() < 4.0


def test_double_literals_follow_cpp_usual_arithmetic_conversions():
    """`0.1*_2` with a double literal (flowz/README.md:52): the generic oracle's float64 promotion
    equals compiled C with the same types, and differs from the all-float spelling."""
    x = O.synth_input(21, np.arange(NS), T)
    assert same(C.one_pole_readme(0.9, x), O.compile(G.one_pole_readme(0.9), NS).run(x))
    assert same(C.mixed_precision_biquad(x), O.compile(G.mixed_precision_biquad(), NS).run(x))
    allf = ("fb", ("add", ("mul", G.lit(0.9), G.DEL(1, 1)), ("mul", G.lit(0.1), G.IN(2))))
    assert not same(O.compile(allf, NS).run(x), O.compile(G.one_pole_readme(0.9), NS).run(x))


def test_complex_wires_three_spellings_agree():
    """std::complex<float> wires (tests.cpp:206-207): the generic Python oracle (_Cplx pairs), the C
    restatement (float _Complex) and the std::complex<float> spelling compiled by g++ are bit-identical."""
    x = O.synth_input(77, np.arange(257), 96)
    g = G.complex_mix()
    assert O.output_dtypes(g) == ["cf32", "f32"]
    want = C.complex_mix(x, std=True)
    assert same(C.complex_mix(x), want)
    assert same(O.compile(g, 257).run(x), want)
    assert np.isfinite(want).all() and (want[..., 1] != 0).any()
    # per-sample protocol: the complex wire comes back as one complex value
    f = O.compile(("mul", ("litc", 0.0, 1.0), ("in", 1)))
    (z,) = f.step(2.0)
    assert z.dtype == np.complex64 and z[0] == 2j
    # what C++ rejects is rejected
    with pytest.raises(O.GraphError):
        O.compile(("mul", ("litc", 1.0, 0.0), ("lit64", 2.0)))                    # complex<float> * double
    with pytest.raises(O.GraphError):
        O.compile(("seq", ("mul", ("litc", 1.0, 0.0), ("in", 1)), ("del", 1, 1)))  # float delay line


def test_typed_state_and_complex_division_three_spellings_agree():
    """complex STATE (~(c*_1[_1] + _2)), both spellings of the complex division (z/w, s/w = libgcc's
    __divsc3 as g++ links it: float handled with double precision) and a double accumulator -- the typed Python oracle,
    the C restatement and std::complex<float> compiled by g++ are bit-identical."""
    x = O.synth_input(78, np.arange(129), 200)
    for g, cf in ((G.complex_one_pole(), C.complex_one_pole), (G.complex_div_mix(), C.complex_div_mix)):
        want = cf(x, std=True)
        assert same(cf(x), want) and np.isfinite(want).all() and (want[..., 1] != 0).any()
        assert O.output_dtypes_typed(g) == ["cf32"]
        y = O.run_typed(O.compile(g, 129, typed=True), [x[:, :, 0]])[0]
        assert y.dtype == np.complex64 and same(np.stack([y.real, y.imag], -1), want)
    y = O.run_typed(O.compile(G.double_accumulator(), 129, typed=True), [x[:, :, 0]])[0]
    assert y.dtype == np.float64 and np.array_equal(y, C.double_accumulator(x)[:, :, 0])
    with pytest.raises(O.GraphError):
        O.compile(G.complex_one_pole())                                           # compile(): float state (flowz.hpp:1245)


def _cdouble_input(T, ns, seed=5):
    """double input whose sum with B.real() = 1.5 crosses |c| < |d| (|1.5 + x| < 0.75) about one time in seven"""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, (T, ns))
    x[::3] *= 3.0
    return x


def test_complex_double_state_and_smith_division_three_spellings_agree():
    """std::complex<double> wires: complex<double> state and both spellings of the division (__divdc3 = Smith's method with
    its data-dependent branch) -- the typed Python oracle, the C restatement and std::complex<double> compiled by g++ are
    bit-identical, and both sides of the branch are taken."""
    x = _cdouble_input(240, 33)
    want = C.cdouble_resonator(x, std=True)
    took = np.abs(1.5 + x) < 0.75
    assert 0.05 < took.mean() < 0.5 and np.isfinite(want.view(np.float64)).all()
    assert np.array_equal(C.cdouble_resonator(x).view(np.int64), want.view(np.int64))
    g = G.cdouble_resonator()
    assert O.output_dtypes_typed(g, ["f64"]) == ["cf64"]
    y = O.run_typed(O.compile(g, 33, in_dtypes=["f64"]), [x])[0]
    assert y.dtype == np.complex128 and np.array_equal(y.view(np.int64), want.view(np.int64))
    # what C++ has no operator for: complex<double> with float / complex<float>; double with complex<float>
    for bad, dts in ((G.mul(G.litc64(1, 0), G.IN(1)), ["f32"]), (G.mul(G.litc64(1, 0), G.litc(1, 0)), None),
                     (G.add(G.IN(1), G.IN(2)), ["cf64", "cf32"]), (G.add(G.IN(1), G.IN(2)), ["cf32", "f64"])):
        with pytest.raises(O.GraphError):
            O.compile(bad, 1, typed=True, in_dtypes=dts)
    # a float recursion variable is absorbed by the complex<double> it meets (ResultType's absorber, flowz.hpp:602-620)
    assert O.output_dtypes_typed(G.fb(G.add(G.mul(G.litc64(0.5, 0.5), G.DEL(1, 1)), G.mul(G.lit64(1.0), G.IN(2))))) == ["cf64"]
    # ... but a float wire that STAYS float and meets a complex<double> is the compile error it is in C++
    with pytest.raises(O.GraphError):
        O.compile(G.fb(G.chan(G.add(G.DEL(1, 1), G.IN(3)), G.mul(G.litc64(1, 0), G.DEL(1, 1)))), 1, typed=True)


def test_rbj_lowpass_oracle_matches_reference_spelling_within_1ulp():
    """reactive_filter_coeff.cpp:38-58: the reference calls std::sin / std::cos on a float -- glibc's sinf / cosf on this box, accurate to
    0.56 ULP, i.e. NOT always the correctly rounded float.  The checker (and, operation for operation, the device generator) evaluates its
    own polynomial pair in double and rounds once: correctly rounded, hence within 1 ULP of the reference's and equal in ~98.7 % of the
    arguments.  The coefficients inherit that bound."""
    rng = np.random.default_rng(3)
    freq = rng.uniform(20.0, 20000.0, 4096).astype(np.float32)
    q = rng.uniform(0.3, 12.0, 4096).astype(np.float32)
    raw6, df1 = C.rbj_lowpass(freq, q, 44100.0)
    ref = C.rbj_lowpass(freq, q, 44100.0, libmf=True)
    ulp = np.abs(raw6.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    # cancellation in 1.-cosw0 / 1.-alpha can amplify a 1-ULP sin/cos difference: bound it relative to 1.0
    assert (np.abs(raw6 - ref) <= 2.0 ** -23).all()
    assert (ulp == 0).mean() > 0.9
    # the reference's own demo point: sr 44100, freq 440, Q 1/sqrt(2)  (:38-40)
    r, d = C.rbj_lowpass([440.0], [1.0 / np.sqrt(2.0)], 44100.0)
    a0, a1, a2, b0, b1, b2 = (float(v) for v in r[:, 0])
    assert abs(a0 - 1.0443) < 1e-3 and abs(a1 + 1.99607) < 1e-4 and abs(b1 - 2 * b0) < 1e-9 and b0 == b2
    assert np.allclose(d[:, 0], [b0 / a0, b1 / a0, b2 / a0, -a1 / a0, -a2 / a0], rtol=1e-7)


def test_polynomial_sincos_is_correctly_rounded_and_within_1ulp_of_libm():
    """fzo_sincos_f32 (oracle/flowz_oracle.c): sin / cos of a float by reduction + Taylor polynomials in IEEE double, rounded to float once --
    the pair the RBJ generator uses on BOTH sides so that device and checker agree bit for bit.  Against a 200-bit evaluation it is the
    correctly rounded float on every sample; against glibc's sinf / cosf (the reference's std::sin(float) here) within 1 ULP."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0.0, np.pi, 400000), rng.uniform(-40.0, 40.0, 100000), rng.uniform(-1.0e6, 1.0e6, 100000),
                        [0.0, -0.0, 1e-30, -1e-30, np.pi / 2, np.pi / 4, 3 * np.pi / 4, 1.0, 2.0 ** 20 - 1.0]]).astype(np.float32)
    s, c = C.sincos_f32(x)
    sl, cl = C.sincos_f32(x, libm=True)
    ulps = lambda a, b: np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))   # noqa: E731
    tiny = (np.abs(sl) < 1e-6) | (np.abs(cl) < 1e-6)          # (results next to zero: an absolute error of 2^-24 ULPs of 1 is many ULPs of 1e-7)
    assert ulps(s, sl)[~tiny].max() <= 1 and ulps(c, cl)[~tiny].max() <= 1
    assert (ulps(s, sl) == 0).mean() > 0.98 and (ulps(c, cl) == 0).mean() > 0.98
    assert np.abs(s - sl).max() <= 2.0 ** -24 and np.abs(c - cl).max() <= 2.0 ** -24
    assert s[-9] == 0.0 and c[-9] == 1.0 and s[-8] == 0.0 and s[-7] == np.float32(1e-30)
    # out of the domain (|x| >= 2^20: k * pi/2-head would no longer be exact), inf, nan: NaN -- on the device too
    sn, cn = C.sincos_f32(np.array([2.0 ** 20, -3e9, np.inf, np.nan], np.float32))
    assert np.isnan(sn).all() and np.isnan(cn).all()
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.prec = 200
    for i in rng.integers(0, len(x), 3000):
        xi = mpmath.mpf(float(x[i]))
        assert np.float32(float(mpmath.sin(xi))) == s[i] and np.float32(float(mpmath.cos(xi))) == c[i], float(x[i]).hex()
