"""The BASELINE workload graphs and their coefficient sets.

Graphs are written in the neutral s-expression notation that zignal_amd.flowz.from_sexpr turns into
EDSL expressions; every builder cites the reference expression it restates.  Pure data + tiny
builders (numpy only): bench.py, __graft_entry__.py, examples/ and the test-suite all take the
workloads from here, so the package can run its own headline without the test tree.
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


def cmp(op, a, b):
    """('lt'|'le'|'gt'|'ge'|'eq'|'ne'|'and'|'or', a, b): the C++ comparison / logical operators, 1.0 or 0.0"""
    return (op, a, b)


def hard_clipper(lo=-0.5, hi=0.5):
    """x limited to [lo, hi] with comparison operators only, as a C++ user of the reference would spell it without <algorithm>:
    x * ((x > lo) && (x < hi)) + hi * (x >= hi) + lo * (x <= lo)"""
    x = IN(1)
    return add(add(mul(x, cmp("and", cmp("gt", x, lit(lo)), cmp("lt", x, lit(hi)))), mul(lit(hi), cmp("ge", x, lit(hi)))), mul(lit(lo), cmp("le", x, lit(lo))))


def clipped_biquad(lo=-0.5, hi=0.5):
    """a DF1 biquad whose recursion runs through the hard clipper: fwd |= ~( clip( a1 _1[_1] + a2 _1[_2] + _2 ) )"""
    clip_of = lambda e: add(add(mul(e, cmp("and", cmp("gt", e, lit(lo)), cmp("lt", e, lit(hi)))), mul(lit(hi), cmp("ge", e, lit(hi)))), mul(lit(lo), cmp("le", e, lit(lo))))   # noqa: E731
    rec = add(add(mul(DEL(1, 1), lit(STABLE[3])), mul(DEL(1, 2), lit(STABLE[4]))), IN(2))
    return seq(fwd(*STABLE[:3]), fb(clip_of(rec)))


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




# ---- the graphs bench.py's `next_rows` object times ------------------------------------
def mod(k):
    return ("mod", k)


def litc(re, im):
    return ("litc", float(F32(re)), float(F32(im)))


def lds_ring_comb():
    """f1: `(_1 + 0.5*_1[_40]) |= ~(0.7*_1[_23] + _2)` -- a feed-forward and a feedback comb whose lines (40 and 23 samples) are
    LDS ring buffers, one column per lane (the `long_delay_lds` graph of the parity tests)."""
    return seq(add(IN(1), mul(lit(0.5), DEL(1, 40))), fb(add(mul(lit(0.7), DEL(1, 23)), IN(2))))


def far_comb(depth=300):
    """f1: `~(0.5*_1[_300] + _2)` -- a feedback comb whose line is a ring in HBM (deeper than the 256 samples LDS rings hold):
    one appended row and one far read per sample, i.e. 16 algorithmic bytes per stream-sample."""
    return fb(add(mul(lit(0.5), DEL(1, depth)), IN(2)))


def df1_cascade_params(n=6):
    """f2: n x DF1 whose 5n coefficients are per-stream parameters (std::ref terminals at block rate, flowz/README.md:42-61):
    the graph fz_bank_process_blocks runs with one coefficient set per 64-sample window."""
    return seq(*[df1_param(5 * j) for j in range(n)])


def df1_mod(k, b=(F32(0.2 * 0.25), F32(-0.3 * 0.25), F32(1.1 * 0.25)), a2=F32(-0.8)):
    """DF1 whose recursion coefficient a1 is the sample-rate modulator k (the std::ref(a) of flowz/README.md:52, re-read every sample)"""
    f = add(add(mul(lit(b[0]), IN(1)), mul(lit(b[1]), DEL(1, 1))), mul(lit(b[2]), DEL(1, 2)))
    r = fb(add(add(IN(2), mul(mod(k), DEL(1, 1))), mul(lit(a2), DEL(1, 2))))
    return seq(f, r)


def df1_cascade_modulated(n=6):
    """f2: n x DF1, every stage's a1 read from sample-rate modulator 0 (one value per sample, the same for all streams)"""
    return seq(*[df1_mod(0) for _ in range(n)])


def df1_double():
    """f3: one DF1 biquad whose coefficients are C++ `double` literals: under fz_compile_typed every wire, both delay lines and the
    output frame are double (ResultType, flowz.hpp:585-644; test/tests.cpp:222-231)."""
    f = add(add(mul(lit64(0.05), IN(1)), mul(lit64(-0.075), DEL(1, 1))), mul(lit64(0.275), DEL(1, 2)))
    r = fb(add(add(IN(2), mul(lit64(0.2), DEL(1, 1))), mul(lit64(-0.8), DEL(1, 2))))
    return seq(f, r)


def complex_one_pole(c=(0.6, 0.7)):
    """f3: `~( c*_1[_1] + _2 )` with a std::complex<float> coefficient (test/tests.cpp:206-207): complex wire, complex delay line"""
    return fb(add(mul(litc(*c), DEL(1, 1)), IN(2)))


# ---- synthetic inputs and per-stream coefficients -------------------------------------------
SEED = 20160512


def hash32(seed, s, t):
    """The integer mixer of SURVEY 8d: murmur3's fmix32, twice, over seed ^ s*0x9E3779B9 ^ t*0x85EBCA6B
    (uint32 arithmetic; the device generator fz_synth_fill computes the same bits)."""
    M = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        h = (np.uint64(seed) ^ (np.asarray(s, np.uint64) * np.uint64(0x9E3779B9))
             ^ (np.asarray(t, np.uint64) * np.uint64(0x85EBCA6B))) & M
        for _ in range(2):
            h = h ^ (h >> np.uint64(16))
            h = (h * np.uint64(0x85EBCA6B)) & M
            h = h ^ (h >> np.uint64(13))
            h = (h * np.uint64(0xC2B2AE35)) & M
            h = h ^ (h >> np.uint64(16))
    return h.astype(np.uint32)


def _unit01(seed, streams, j):
    """hash -> [0,1) float64, deterministic in (seed, stream, j)."""
    return hash32(seed, streams, j).astype(np.float64) / 4294967296.0


def osc_chain_params(seed, streams, n_stage=6):
    """[1+5n, n_streams] float32: k = 2cos(theta), theta in (0.05,3.0); per stage r in [0.5,0.95],
    a1 = 2 r cos(phi), a2 = -r^2, b in [-0.5,0.5]  (SURVEY 8d config 4)."""
    streams = np.asarray(streams, np.uint64)
    P = np.empty((1 + 5 * n_stage, len(streams)), F32)
    theta = 0.05 + 2.95 * _unit01(seed, streams, 0)
    P[0] = (2.0 * np.cos(theta)).astype(F32)
    for j in range(n_stage):
        base = 1 + 5 * j
        for q in range(3):
            P[base + q] = (_unit01(seed, streams, 10 * (j + 1) + q) - 0.5).astype(F32)
        r = 0.5 + 0.45 * _unit01(seed, streams, 10 * (j + 1) + 3)
        phi = np.pi * _unit01(seed, streams, 10 * (j + 1) + 4)
        P[base + 3] = (2.0 * r * np.cos(phi)).astype(F32)
        P[base + 4] = (-(r * r)).astype(F32)
    return P


# name -> builder of the graphs the BASELINE configs run (bench.py, build() pre-builds their kernels)
BASELINE_GRAPHS = {
    "df1_cascade6": lambda: df1_cascade(6),
    "par4_sum": par4_sum,
    "par4_sum_fanout": par4_sum_fanout,
    "osc_chain6": lambda: osc_chain(6),
}


# Stream-major frames ([stream][t][wire], the reference's calling convention): kernel variants (P, U, block, flags without
# FZ_VF_STREAM_MAJOR) a caller may want to measure -- the library default, the pair long-run body (two streams per lane), the one-stream long-run body with and
# without stage packing, short chunks.  fz_program_tune measures frame layouts only; bench.py's stream-major leg times these (build() pre-builds them).
SM_CANDIDATES = [(0, 0, 0, 0), (2, 64, 0, 256), (1, 128, 0, 256), (1, 128, 0, 256 | 16), (1, 32, 0, 512 | 16), (0, 0, 0, 512)]
