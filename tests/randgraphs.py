"""Seeded random Flowz graphs (s-expressions) for fuzzing the lowering and the kernels.

Graphs are built from the reference's combinators with their arity rules; all
feedback paths go through small coefficients so that responses stay bounded."""
import os

import numpy as np

from graphs import DEL, IN, add, chan, fb, lit, mul, par, seq, sub


def _coef(rng, lo=0.05, hi=0.45):
    c = rng.uniform(lo, hi)
    return lit(c if rng.random() < 0.5 else -c)


def _leaf(rng, n_in, delayed_only=()):
    i = int(rng.integers(1, n_in + 1))
    if i in delayed_only or rng.random() < 0.45:
        return DEL(i, int(rng.integers(1, 5)))
    return IN(i)


def box(rng, n_in, delayed_only=(), must_use=None):
    """arithmetic box over _1.._n_in with one output"""
    terms = []
    uses = list(range(1, n_in + 1)) if must_use is None else list(must_use)
    rng.shuffle(uses)
    n_terms = max(len(uses), int(rng.integers(1, 4)))
    for k in range(n_terms):
        if k < len(uses):
            i = uses[k]
            leaf = DEL(i, int(rng.integers(1, 5))) if (i in delayed_only or rng.random() < 0.4) else IN(i)
        else:
            leaf = _leaf(rng, n_in, delayed_only)
        t = mul(_coef(rng), leaf) if rng.random() < 0.8 else leaf
        if rng.random() < 0.15:
            t = ("neg", t)
        terms.append(t)
    e = terms[0]
    for t in terms[1:]:
        e = add(e, t) if rng.random() < 0.7 else sub(e, t)
    if rng.random() < 0.1:
        e = ("div", e, lit(rng.uniform(1.5, 3.0)))
    return e


def graph(rng, n_in, depth):
    """random graph with exactly n_in input wires; returns (sexpr, n_out)"""
    r = rng.random()
    if depth <= 0 or r < 0.25:
        return box(rng, n_in), 1
    if r < 0.45:                                     # a |= b
        a, oa = graph(rng, n_in, depth - 1)
        b, ob = graph(rng, oa, depth - 1)
        return seq(a, b), ob
    if r < 0.6 and n_in >= 2:                        # a | b
        k = int(rng.integers(1, n_in))
        a, oa = graph(rng, k, depth - 1)
        b, ob = graph(rng, n_in - k, depth - 1)
        return par(a, b), oa + ob
    if r < 0.75:                                     # a , b  (same inputs)
        a, oa = graph(rng, n_in, depth - 1)
        b, ob = graph(rng, n_in, depth - 1)
        return chan(a, b), oa + ob
    if r < 0.9:                                      # ~( feedback of one wire through a delayed read )
        body = box(rng, n_in + 1, delayed_only=(1,), must_use=range(1, n_in + 2))
        g = fb(body)
        if rng.random() < 0.5:
            post, op = graph(rng, 1, depth - 1)
            return seq(g, post), op
        return g, 1
    # two cross-coupled fed-back wires (the cross_wire pattern, multi_wires_feedback.cpp:721)
    inner = chan(DEL(2, int(rng.integers(1, 3))), *[IN(2 + k) for k in range(1, n_in + 1)], DEL(1, int(rng.integers(1, 3))))
    left = box(rng, 1 + n_in, must_use=range(1, n_in + 2))
    right = mul(_coef(rng), IN(1))
    return fb(seq(inner, par(left, right))), 2


def make(seed, max_in=3, depth=3):
    rng = np.random.default_rng(seed)
    n_in = int(rng.integers(1, max_in + 1))
    g, n_out = graph(rng, n_in, depth)
    return g, n_in, n_out


def _retype(rng, e, p64):
    """flip float literals into C++ double literals with probability p64"""
    if not isinstance(e, tuple):
        return e
    if e[0] == "lit":
        return ("lit64", float(e[1])) if rng.random() < p64 else e
    return tuple(_retype(rng, c, p64) if isinstance(c, tuple) else c for c in e)


def make_typed(seed, max_in=3, depth=3):
    """as make(), with mixed wire types: some coefficients are double literals (float64 sub-expressions,
    narrowed when they enter a delay line), or -- float-only graphs -- a feed-forward
    std::complex<float> stage behind the first output wire, or a std::complex<double> stage there.
    Returns (sexpr, n_in, n_out_wires, kind)."""
    rng = np.random.default_rng(seed + 77000)
    g, n_in, n_out = make(seed, max_in, depth)
    r2 = np.random.default_rng(seed + 99000)          # (its own stream: the graphs of the other kinds stay what they were)
    if r2.random() < 0.2 and not os.environ.get("RANDGRAPHS_NO_CDOUBLE"):   # (the switch: to re-create graphs of runs before this kind existed)
        # a feed-forward std::complex<double> stage behind the first output wire: the wire is widened by a double
        # literal first (complex<double> meets double operands only); z*w, z+s, s-z, -z, z/w and s/w (__divdc3: Smith's
        # method, both sides of its branch are reached: the divisor's real part runs through |c| = |d|)
        c64 = lambda lo, hi: ("litc64", float(r2.uniform(lo, hi)), float(r2.uniform(-1, 1)))   # noqa: E731
        xd = mul(("lit64", float(r2.uniform(0.5, 1.5))), IN(1))
        post = add(mul(mul(c64(-1, 1), xd), c64(-1, 1)), mul(("lit64", float(r2.uniform(-1, 1))), xd))
        if r2.random() < 0.5:
            post = sub(("lit64", float(r2.uniform(-1, 1))), ("neg", post))
        div = add(("litc64", float(r2.uniform(-0.5, 0.5)), float(r2.choice([-1, 1]) * r2.uniform(0.4, 1.0))), xd)
        r = r2.random()
        if r < 0.4:
            post = ("div", post, div)
        elif r < 0.7:
            post = ("div", xd, add(div, mul(("lit64", 0.01), post)))
        return seq(g, post), n_in, n_out, "cdouble"
    if rng.random() < 0.45:                           # the stage reads wire 1, the other outputs pass around it
        z = ("litc", float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)))
        w = ("litc", float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)))
        post = add(mul(mul(z, IN(1)), w), mul(_coef(rng), IN(1)))
        if rng.random() < 0.5:
            post = sub(lit(rng.uniform(-1, 1)), ("neg", post))
        r = rng.random()
        if r < 0.2:                                   # complex / complex (__divsc3), the divisor kept away from zero
            post = ("div", post, add(("litc", float(rng.uniform(2, 3)), float(rng.uniform(-1, 1))), mul(lit(0.01), IN(1))))
        elif r < 0.35:                                # scalar / complex
            post = ("div", mul(lit(0.5), IN(1)), add(("litc", float(rng.uniform(2, 3)), float(rng.uniform(-1, 1))), mul(lit(0.01), post)))
        return seq(g, post), n_in, n_out, "complex"
    return _retype(rng, g, 0.35), n_in, n_out, "double"


def _gate(rng, e, p):
    """sprinkle comparison / logical operators over arithmetic nodes: (a op b) becomes (a op b) * (a cmp b), (a op b) + c * ((a cmp x) && / || (b cmp y))
    or (a op b) * !(a cmp b) -- the operands of the node itself are compared, so arities and delays stay what they were"""
    if not isinstance(e, tuple):
        return e
    e = tuple(_gate(rng, c, p) if isinstance(c, tuple) else c for c in e)
    if e[0] in ("add", "sub", "mul") and rng.random() < p:
        a, b = e[1], e[2]
        cmp_ = lambda x, y: (str(rng.choice(["lt", "le", "gt", "ge", "eq", "ne"])), x, y)   # noqa: E731
        th = lambda: lit(float(rng.uniform(-0.3, 0.3)))                                   # noqa: E731
        r = rng.random()
        if r < 0.4:
            return ("mul", e, cmp_(a, b))
        if r < 0.7:
            return ("add", e, ("mul", lit(float(rng.uniform(-0.1, 0.1))), (str(rng.choice(["and", "or"])), cmp_(a, th()), cmp_(b, th()))))
        if r < 0.85:
            return ("mul", e, ("not", cmp_(a, b)))
        return ("sub", e, ("mul", lit(0.05), cmp_(("mul", ("lit64", 1.0), a), b)))             # compared in double
    return e


def make_cmp(seed, max_in=3, depth=3, p=0.35):
    """as make(), with C++ comparison and logical operators among the arithmetic (round 6: proto::_default applies whatever operator a node is)"""
    g, n_in, n_out = make(seed, max_in, depth)
    return _gate(np.random.default_rng(seed + 55000), g, p), n_in, n_out
