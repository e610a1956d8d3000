"""CPU ORACLE for Flowz flow-graph evaluation -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (zignal_amd/, include/) never does: it fails loudly when the
HIP library is missing.

What this restates
------------------
The per-sample evaluator of the reference, /root/reference/flowz/flowz.hpp:

* arity rules            input_arity  :162-214, output_arity :217-246
* wire routing           sequence :960-1001, parallel :1076-1101, channel :765-768,
                         binary_feedback :1031-1074 (with the arity-table routing, see
                         SURVEY.md App. C.1 -- the shipped `std::min(0, ...)` drop is a
                         reference quirk and such graphs are excluded from parity)
* delay lines            rotate_push_back :130-148 (shift register, newest at the back),
                         place_delay :950-958 (read `xs[N-n]` = value n samples ago),
                         zero-initialised float state :1245, depth = max delay read on the
                         wire (max_input_delays :502, build_state :685-725)
* arithmetic             proto::_default<eval_it> :769-772 -- the built-in C++ operator on
                         the evaluated children: one IEEE float32 rounding per operator, the
                         user's association order, no FMA contraction (CMakeLists.txt:18)
* call protocol          stateful_lambda::operator() :1225-1229 / call_impl :1193-1201:
                         N inputs -> M outputs per call, state persists across calls

How it is written (deliberately NOT like the product): the expression tree is elaborated
once into persistent lazy `Wire` objects; every sample forces the output wires by demand
(memoised recursion), and only afterwards pushes the current value of every delayed wire
into its FIFO -- "all reads see the values pushed in previous samples; pushes happen after
the consumer was evaluated" (flowz.hpp:994, :1067).  The product instead lowers to a
CSE'd, topologically sorted flat DAG with register/LDS delay lines.  Values are numpy
float32 arrays with one element per stream, so many independent streams are evaluated per
Python-level operation.

Parity status: PINNED.  tests/test_oracle_golden.py checks this oracle against every
evaluation known-answer of the reference's own test/tests.cpp (:88-178, transcribed into
tests/golden/tests_cpp_known_answers.json) and against impulse/noise responses of the
reference's Boost-free hand-written biquads (test/benchmark.cpp:35-126,
experimental_steps/multi_wires_feedback.cpp:768-775) compiled from the reference sources
where they lie (oracle/build_ref.sh -> oracle/_ref/, vectors committed as
tests/golden/ref_biquad_vectors.json).

Graph notation ("s-expressions", plain nested tuples; a shared data format, no code):
    ('in', i)            placeholder _i                     flowz.hpp:1252-1257
    ('del', i, n)        delayed placeholder _i[_n]         flowz.hpp:84-85
    ('lit', v)           literal terminal (float32)         flowz.hpp:68-72
    ('lit64', v)         C++ `double` literal terminal: operators above it evaluate in float64
                         (usual arithmetic conversions applied by the built-in operators,
                         flowz.hpp:769-772; test/tests.cpp:200-231); the value is truncated to
                         float when it enters a delay line (rotate_push_back :130-137, state is
                         float :1245) and when it leaves the graph (float32 output frames)
    ('litc', re, im)     std::complex<float> terminal (test/tests.cpp:206-207): the wire above it is
                         complex; operators as <complex> defines them for complex<float> (class _Cplx
                         below); it takes two slots (re, im) of the output frame; it cannot enter a
                         delay line (float state, :1245) nor meet a double operand (no such operator)
    ('litc64', re, im)   a std::complex<double> terminal: as 'litc' with double parts (division: __divdc3, Smith's method);
                         typed programs carry it through delay lines and frames un-narrowed
    ('param', k)         per-stream coefficient k (block-constant std::ref analogue,
                         flowz/README.md:42-61)
    ('mod', k)           sample-rate modulator k: a std::ref(x) terminal whose variable the caller changes between calls
                         (flowz/README.md:42-61: the reference re-reads it on every call); one value per sample, the same
                         for all streams: step(..., mod=[...]) / run(x, mod=[n_mod, T])
    ('add'|'sub'|'mul'|'div', a, b), ('neg', a)             flowz.hpp:769-772
    ('lt'|'le'|'gt'|'ge'|'eq'|'ne'|'and'|'or', a, b), ('not', a)   the comparison / logical operators through the same _default: 1.0 / 0.0
    ('chan', a, b)       a , b                              flowz.hpp:90
    ('par', a, b)        a | b                              flowz.hpp:91
    ('seq', a, b)        a |= b                             flowz.hpp:92
    ('fb', a)            ~a                                 flowz.hpp:93
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

_ARITH = ("add", "sub", "mul", "div")
# comparison and logical operators of C++ (proto::_default applies whatever operator the node is, flowz.hpp:51-55, :769-772): the bool they
# yield, as it behaves in arithmetic -- 1 or 0 taking the type of what it meets (here: a float32 1.0 / 0.0, which numpy promotes the same way)
_CMP = ("lt", "le", "gt", "ge", "eq", "ne", "and", "or")


class GraphError(ValueError):
    pass


# ----------------------------------------------------------------------------------------
# arity rules (flowz.hpp:162-246; cross-checked with experimental_steps/flowz2.cpp:65-78)
# ----------------------------------------------------------------------------------------

def input_arity(e) -> int:
    k = e[0]
    if k in ("in", "del"):
        return int(e[1])                      # :163-170  arity of _i is i
    if k in ("lit", "lit64", "litc", "litc64", "param", "mod"):
        return 0                              # :171-174
    if k == "fb":                             # :175-181
        return max(0, input_arity(e[1]) - output_arity(e[1]))
    if k == "par":                            # :195-198
        return input_arity(e[1]) + input_arity(e[2])
    if k == "seq":                            # :199-208
        return input_arity(e[1]) + max(0, input_arity(e[2]) - output_arity(e[1]))
    if k in _ARITH or k in _CMP or k == "chan":   # :209-212  nary_expr -> max over children
        return max(input_arity(e[1]), input_arity(e[2]))
    if k in ("neg", "not"):
        return input_arity(e[1])
    raise GraphError(f"unknown node {k!r}")


def output_arity(e) -> int:
    k = e[0]
    if k == "chan":                           # :218-221
        return output_arity(e[1]) + output_arity(e[2])
    if k == "fb":                             # :222-225
        return output_arity(e[1])
    if k == "par":                            # :230-233
        return output_arity(e[1]) + output_arity(e[2])
    if k == "seq":                            # :234-243
        return output_arity(e[2]) + max(0, output_arity(e[1]) - input_arity(e[2]))
    return 1                                  # :244


def max_input_delays(e) -> tuple:
    """Per external input wire: deepest delayed read (flowz.hpp:443-502).

    Tuple length follows the reference's generator (`make_arity`, :286-301): a leaf `_i`
    contributes i entries, so the tuple may be shorter than input_arity for wires that are
    only consumed by a later box."""
    k = e[0]

    def zipmax(a, b):                         # max_delay_of_wires :364-379
        n = min(len(a), len(b))
        return tuple(max(x, y) for x, y in zip(a[:n], b[:n])) + tuple(a[n:]) + tuple(b[n:])

    if k == "del":
        return (0,) * (e[1] - 1) + (int(e[2]),)
    if k == "in":
        return (0,) * e[1]
    if k in ("lit", "lit64", "litc", "litc64", "param", "mod"):
        return ()
    if k == "fb":                             # :459-465
        return max_input_delays(e[1])[output_arity(e[1]):]
    if k == "par":                            # :479-482
        return max_input_delays(e[1]) + max_input_delays(e[2])
    if k == "seq":                            # :483-492
        return max_input_delays(e[1]) + max_input_delays(e[2])[output_arity(e[1]):]
    if k in ("neg", "not"):
        return max_input_delays(e[1])
    if k in _ARITH or k in _CMP or k == "chan":   # :493-496 fold with max-zip
        return zipmax(max_input_delays(e[2]), max_input_delays(e[1]))
    raise GraphError(f"unknown node {k!r}")


# ----------------------------------------------------------------------------------------
# std::complex<float> values: the operators of <complex> on (re, im) float32 pairs
# ----------------------------------------------------------------------------------------

class _Cplx:
    """std::complex<float> / std::complex<double>, one per stream (dt = the part type).  libstdc++/libc++ <complex>:
         z *= s, z /= s        scale both parts          z += s, z -= s     real part only
         s - z                 complex r = -z; r += s    -z                 both parts
         z * w                 _Complex multiply = __mulsc3 / __muldc3: ac = a*c, bd = b*d, ad = a*d,
                               bc = b*c, (ac - bd, ad + bc), every operation rounded to the part type (its
                               inf/nan recovery branch is not restated: finite values only)
       complex<float>:
       z / w, s / w          _Complex float divide = libgcc's __divsc3 as g++ links it here ("float is handled with
                             double precision", libgcc2.c): aa..dd = the four parts widened to double,
                             denom = cc*cc + dd*dd, x = (float)((aa*cc + bb*dd)/denom), y = (float)((bb*cc - aa*dd)/denom);
                             s / w is complex<float>(s) /= w (<complex>), i.e. b = +0.f.  Its NaN-recovery branch is not
                             restated (finite values, nonzero divisor).
       complex<double>:
       z / w, s / w          __divdc3 = Smith's method: |c| < |d| ? (ratio = c/d, denom = c*ratio + d, x = (a*ratio + b)/denom,
                             y = (b*ratio - a)/denom) : (ratio = d/c, denom = d*ratio + c, x = (b*ratio + a)/denom,
                             y = (b - a*ratio)/denom).  The scaling branches of newer libgcc for extreme magnitudes
                             (|d| >= DBL_MAX/2, tiny operands, subnormal ratio) and the NaN recovery are not restated.
       tests/test_oracle_c.py pins both against std::complex compiled by g++ (oracle/complex_std.cpp).
       The operand types must agree as C++ demands (operator(complex<T>, T), operator(complex<T>, complex<T>)): nothing
       converts between complex<float> / double or complex<double> / float / complex<float>."""

    __array_ufunc__ = None            # numpy arrays defer to the reflected operators below
    __slots__ = ("re", "im", "dt")
    lenient = False                   # while the typed-state fixpoint still raises line types: see FlowzOracle.__init__

    def __init__(self, re, im):
        re, im = np.asarray(re), np.asarray(im)
        self.dt = np.float64 if (re.dtype == np.float64 or im.dtype == np.float64) else F32
        self.re = re.astype(self.dt, copy=False)
        self.im = im.astype(self.dt, copy=False)

    def _scalar(self, o):
        o = np.asarray(o)
        if o.dtype != self.dt:
            if _Cplx.lenient:
                return o.astype(np.float64)
            raise GraphError("std::complex<float> and double operands (std::complex<double> and float ones) do not mix: no such operator in C++")
        return o

    def _peer(self, o):
        if o.dt != self.dt and not _Cplx.lenient:
            raise GraphError("std::complex<float> and std::complex<double> operands do not mix (no such operator in C++)")
        return o

    def __add__(self, o):
        if isinstance(o, _Cplx):
            self._peer(o)
            return _Cplx(self.re + o.re, self.im + o.im)
        return _Cplx(self.re + self._scalar(o), self.im)

    def __radd__(self, o):
        return _Cplx(self.re + self._scalar(o), self.im)

    def __sub__(self, o):
        if isinstance(o, _Cplx):
            self._peer(o)
            return _Cplx(self.re - o.re, self.im - o.im)
        return _Cplx(self.re - self._scalar(o), self.im)

    def __rsub__(self, o):
        return _Cplx((-self.re) + self._scalar(o), -self.im)

    def __mul__(self, o):
        if isinstance(o, _Cplx):
            self._peer(o)
            ac, bd = self.re * o.re, self.im * o.im
            ad, bc = self.re * o.im, self.im * o.re
            return _Cplx(ac - bd, ad + bc)
        o = self._scalar(o)
        return _Cplx(self.re * o, self.im * o)

    def __rmul__(self, o):
        o = self._scalar(o)
        return _Cplx(self.re * o, self.im * o)

    @staticmethod
    def _wide_div(a, b, c, d):
        aa, bb, cc, dd = (np.asarray(v, F32).astype(np.float64) for v in (a, b, c, d))
        denom = (cc * cc) + (dd * dd)
        x = ((aa * cc) + (bb * dd)) / denom
        y = ((bb * cc) - (aa * dd)) / denom
        return _Cplx(x.astype(F32), y.astype(F32))

    @staticmethod
    def _smith_div(a, b, c, d):
        a, b, c, d = (np.asarray(v, np.float64) for v in (a, b, c, d))
        r1 = c / d
        den1 = (c * r1) + d
        x1, y1 = ((a * r1) + b) / den1, ((b * r1) - a) / den1
        r2 = d / c
        den2 = (d * r2) + c
        x2, y2 = ((b * r2) + a) / den2, (b - (a * r2)) / den2
        m = np.abs(c) < np.abs(d)
        return _Cplx(np.where(m, x1, x2), np.where(m, y1, y2))

    def _div(self, a, b, c, d):
        return self._smith_div(a, b, c, d) if self.dt == np.float64 else self._wide_div(a, b, c, d)

    def __truediv__(self, o):
        if isinstance(o, _Cplx):
            self._peer(o)
            return self._div(self.re, self.im, o.re, o.im)
        o = self._scalar(o)
        return _Cplx(self.re / o, self.im / o)

    def __rtruediv__(self, o):
        o = self._scalar(o)
        return self._div(o, np.zeros_like(self.re), self.re, self.im)

    def __neg__(self):
        return _Cplx(-self.re, -self.im)


# ----------------------------------------------------------------------------------------
# lazy wires
# ----------------------------------------------------------------------------------------

def _kind_of(v):
    if isinstance(v, _Cplx):
        return "cf64" if v.dt == np.float64 else "cf32"
    return "f64" if np.asarray(v).dtype == np.float64 else "f32"


class _Wire:
    """One signal wire.  `fn()` computes the value of the current sample on demand."""

    __slots__ = ("fn", "stamp", "val", "busy", "depth", "fifo")

    def __init__(self, fn=None):
        self.fn = fn
        self.stamp = -1
        self.val = None
        self.busy = False
        self.depth = 0          # deepest delayed read of this wire
        self.fifo = None        # list of `depth` arrays, fifo[-1] = newest (rotate_push_back)


class FlowzOracle:
    """compile()-callable of the reference for `n_streams` independent streams at once.

    step(*inputs) == one call of stateful_lambda::operator() per stream (flowz.hpp:1225)."""

    def __init__(self, expr, n_streams: int = 1, params=None, out_f64: bool = False, typed: bool = False, in_dtypes=None):
        """typed: the wire types of the reference's ResultType transform (flowz.hpp:585-644, test/tests.cpp:184-232) carried
        through inputs (in_dtypes: 'f32' / 'f64' / 'cf32' per input wire -- the reference's callable is a template over
        its argument types, :1225-1229), delay lines (a line stores what is pushed: tests.cpp:219; a fed-back wire gets
        the least type consistent around its loop = what the absorber of :602-620 leaves) and outputs (nothing narrowed)."""
        self.expr = expr
        self.typed = bool(typed or in_dtypes is not None)
        self.in_dtypes = list(in_dtypes) if in_dtypes is not None else None
        self.out_dtype = np.float64 if out_f64 else F32      # float64: outputs leave un-narrowed
        self.n_streams = int(n_streams)
        self.n_in = input_arity(expr)
        self.n_out = output_arity(expr)
        self._t = 0
        self._wires = []
        self._params = None
        if params is not None:
            p = np.asarray(params, dtype=F32)
            if p.ndim == 1:
                p = p[:, None]
            self._params = np.ascontiguousarray(np.broadcast_to(p, (p.shape[0], self.n_streams)))
        self._cur_in = [None] * self.n_in
        self._mod_now = [F32(0)] * 256
        # add_front_panel (:261-277): one loose wire per external input
        self._inputs = [self._new(self._mk_input(i)) for i in range(self.n_in)]
        outs = self._elab(expr, self._inputs)
        if len(outs) != self.n_out:
            raise GraphError(f"arity table says {self.n_out} outputs, routing produced {len(outs)}")
        self._outs = outs
        self._delayed = [w for w in self._wires if w.depth > 0]
        for w in self._delayed:                      # zero-initialised float state (:1245)
            w.fifo = [np.zeros(self.n_streams, F32) for _ in range(w.depth)]
        if self.in_dtypes is None:
            self.in_dtypes = ["f32"] * self.n_in
        if len(self.in_dtypes) != self.n_in:
            raise GraphError("one input dtype per input wire")

        def zero_of(kind):
            if kind in ("cf32", "cf64"):
                dt = np.float64 if kind == "cf64" else F32
                return _Cplx(np.zeros(self.n_streams, dt), np.zeros(self.n_streams, dt))
            return np.zeros(self.n_streams, np.float64 if kind == "f64" else F32)

        kind_of = _kind_of

        # output frame slots: a complex wire takes two (re, im).  Types are static: probe them once.
        for i in range(self.n_in):
            self._cur_in[i] = zero_of(self.in_dtypes[i])
        rank = {"f32": 0, "f64": 1, "cf32": 2, "cf64": 3}
        with np.errstate(all="ignore"):
            # typed state: raise the lines' types until they settle (least fixpoint from float).  While a line's type is
            # still an assumption an operator may meet a pair C++ has no operator for (a float recursion variable times a
            # complex<double>): the verdict waits for the settled types (the strict pass below).
            _Cplx.lenient = self.typed
            try:
                for _ in range(8):
                    self._t += 1
                    changed = False
                    for w in self._delayed:
                        k = kind_of(self._value(w))
                        if k in ("cf32", "cf64") and not self.typed:
                            raise GraphError("a std::complex wire cannot enter a delay line: compile() stores float state")
                        had = kind_of(w.fifo[0])
                        if self.typed and k != had:
                            lo, hi = sorted((k, had), key=rank.get)
                            if lo != "f32" and (lo, hi) != ("f64", "cf64"):
                                raise GraphError("a delayed wire has two types C++ does not convert into each other (double / std::complex<float> / std::complex<double>)")
                            w.fifo = [zero_of(hi) for _ in range(w.depth)]
                            changed = True
                    if not changed:
                        break
            finally:
                _Cplx.lenient = False
            self._t += 1
            for w in self._delayed:
                self._value(w)
            probe = [self._value(w) for w in self._outs]
        self.out_types = [kind_of(v) for v in probe]
        self.n_slots = sum(2 if k in ("cf32", "cf64") else 1 for k in self.out_types)

    # -- construction ------------------------------------------------------------------
    def _new(self, fn=None):
        w = _Wire(fn)
        self._wires.append(w)
        return w

    def _mk_input(self, i):
        return lambda: self._cur_in[i]

    def _value(self, w: _Wire):
        if w.stamp == self._t:
            return w.val
        if w.busy:
            raise GraphError("delay-free feedback loop (every cycle needs >= 1 delayed read)")
        w.busy = True
        try:
            v = w.fn()
        finally:
            w.busy = False
        w.val = v
        w.stamp = self._t
        return v

    def _elab(self, e, ins):
        """Elaborate node `e` fed by wires `ins`; returns its output wires."""
        k = e[0]
        if k == "in":                                           # place_the_holder :941-948
            i = e[1]
            if i > len(ins):
                raise GraphError(f"placeholder _{i} has no wire to bind to")
            return [ins[i - 1]]
        if k == "del":                                          # place_delay :950-958
            i, n = e[1], int(e[2])
            if n < 1:
                raise GraphError("delay must be >= 1")
            if i > len(ins):
                raise GraphError(f"placeholder _{i} has no wire to bind to")
            src = ins[i - 1]
            src.depth = max(src.depth, n)
            return [self._new(lambda src=src, n=n: src.fifo[len(src.fifo) - n])]
        if k == "lit":
            c = F32(e[1])
            return [self._new(lambda c=c: np.full(self.n_streams, c, F32))]
        if k == "lit64":
            c64 = np.float64(e[1])
            return [self._new(lambda c64=c64: np.full(self.n_streams, c64, np.float64))]
        if k == "litc":
            cr, ci = F32(e[1]), F32(e[2])
            return [self._new(lambda cr=cr, ci=ci: _Cplx(np.full(self.n_streams, cr, F32), np.full(self.n_streams, ci, F32)))]
        if k == "litc64":
            zr, zi = np.float64(e[1]), np.float64(e[2])
            return [self._new(lambda zr=zr, zi=zi: _Cplx(np.full(self.n_streams, zr, np.float64), np.full(self.n_streams, zi, np.float64)))]
        if k == "param":
            idx = int(e[1])
            return [self._new(lambda idx=idx: self._params[idx])]
        if k == "mod":                                          # the referenced variable as it is at this call
            idx = int(e[1])
            return [self._new(lambda idx=idx: np.full(self.n_streams, F32(self._mod_now[idx]), F32))]
        if k in _ARITH:                                         # _default<eval_it> :769-772
            a = self._one(e[1], ins)
            b = self._one(e[2], ins)
            if k == "add":
                f = lambda a=a, b=b: self._value(a) + self._value(b)
            elif k == "sub":
                f = lambda a=a, b=b: self._value(a) - self._value(b)
            elif k == "mul":
                f = lambda a=a, b=b: self._value(a) * self._value(b)
            else:
                f = lambda a=a, b=b: self._value(a) / self._value(b)
            return [self._new(f)]
        if k == "neg":
            a = self._one(e[1], ins)
            return [self._new(lambda a=a: -self._value(a))]
        if k in _CMP or k == "not":                             # the same _default<eval_it>: C++'s operator on the operands' common type, a bool
            a = self._one(e[1], ins)
            b = self._one(e[2], ins) if k != "not" else None

            def truth(k=k, a=a, b=b):
                x = self._value(a)
                y = self._value(b) if b is not None else None
                if isinstance(x, _Cplx) or isinstance(y, _Cplx):
                    raise GraphError("comparison operators do not apply to std::complex wires")
                with np.errstate(invalid="ignore"):
                    m = {"lt": lambda: x < y, "le": lambda: x <= y, "gt": lambda: x > y, "ge": lambda: x >= y, "eq": lambda: x == y, "ne": lambda: x != y,
                         "and": lambda: (x != 0) & (y != 0), "or": lambda: (x != 0) | (y != 0), "not": lambda: x == 0}[k]()
                return np.where(m, F32(1), F32(0)).astype(F32)
            return [self._new(truth)]
        if k == "chan":                                         # :765-768 same inputs to both
            return self._elab(e[1], ins) + self._elab(e[2], ins)
        if k == "par":                                          # :1087-1099 split at in(a)
            na = input_arity(e[1])
            nb = input_arity(e[2])
            if na + nb > len(ins):
                raise GraphError("parallel box needs more wires than available")
            return self._elab(e[1], ins[:na]) + self._elab(e[2], ins[na:na + nb])
        if k == "seq":                                          # :974-999
            na = input_arity(e[1])
            nb = input_arity(e[2])
            ao = self._elab(e[1], ins[:na])
            bi = ao + ins[na:]
            bo = self._elab(e[2], bi)
            return bo + ao[nb:]
        if k == "fb":                                           # binary_feedback :1031-1074
            n_fb = output_arity(e[1])
            fwd = [self._new(None) for _ in range(n_fb)]
            ao = self._elab(e[1], fwd + ins)
            if len(ao) != n_fb:
                raise GraphError("feedback body arity mismatch")
            for f, o in zip(fwd, ao):
                f.fn = (lambda o=o: self._value(o))
            return ao
        raise GraphError(f"unknown node {k!r}")

    def _one(self, e, ins):
        ws = self._elab(e, ins)
        if len(ws) != 1:
            raise GraphError("arithmetic operand must have exactly one output wire")
        return ws[0]

    # -- evaluation ----------------------------------------------------------------------
    def step(self, *inputs, mod=None):
        """One sample for every stream.  inputs: n_in scalars or (n_streams,) arrays; mod: the values of the sample-rate
        modulators at this call."""
        if mod is not None:
            self._mod_now = [F32(v) for v in mod] + [F32(0)] * 8
        if len(inputs) != self.n_in:
            raise GraphError(f"expected {self.n_in} inputs, got {len(inputs)}")
        self._t += 1
        for i, x in enumerate(inputs):
            dt = self.in_dtypes[i]
            if dt in ("cf32", "cf64"):
                z = np.broadcast_to(np.asarray(x, dtype=np.complex128 if dt == "cf64" else np.complex64), (self.n_streams,))
                self._cur_in[i] = _Cplx(np.ascontiguousarray(z.real), np.ascontiguousarray(z.imag))
            else:
                self._cur_in[i] = np.ascontiguousarray(
                    np.broadcast_to(np.asarray(x, dtype=np.float64 if dt == "f64" else F32), (self.n_streams,)))
        outs = []
        for w in self._outs:                       # complex wires come out as numpy complex (exact pair)
            v = self._value(w)
            if isinstance(v, _Cplx):
                wide = self.out_dtype == np.float64 or (self.typed and v.dt == np.float64)
                c = np.empty(self.n_streams, np.complex128 if wide else np.complex64)
                c.real, c.imag = v.re, v.im
                outs.append(c)
            elif self.typed:
                outs.append(np.array(v, copy=True))                     # its own type: nothing is narrowed
            else:
                outs.append(np.array(v, dtype=self.out_dtype, copy=True))
        # consumers first, pushes last (:994, :1067)
        if self.typed:                               # a line stores the type that is pushed (ResultType)
            new = []
            for w in self._delayed:
                v = self._value(w)
                new.append(_Cplx(v.re.copy(), v.im.copy()) if isinstance(v, _Cplx) else np.array(v, copy=True))
        else:
            new = [np.array(self._value(w), dtype=F32, copy=True) for w in self._delayed]
        for w, v in zip(self._delayed, new):
            w.fifo.pop(0)               # rotate_push_back :130-137
            w.fifo.append(v)
        return tuple(outs)

    def run(self, x, mod=None):
        """x: float32 [T, n_streams, n_in] (time-major frames) -> [T, n_streams, n_slots]; mod: [n_mod, T] modulator values
        (n_slots == n_out unless some output wires are complex: those take two slots, re then im)."""
        x = np.asarray(x, dtype=F32)
        if x.ndim == 2 and self.n_in == 1:
            x = x[:, :, None]
        T = x.shape[0]
        y = np.empty((T, self.n_streams, self.n_slots), self.out_dtype)
        with np.errstate(all="ignore"):
            for t in range(T):
                o = self.step(*[x[t, :, i] for i in range(self.n_in)], mod=None if mod is None else [m[t] for m in mod])
                k = 0
                for j in range(self.n_out):
                    if self.out_types[j] in ("cf32", "cf64"):
                        y[t, :, k] = o[j].real
                        y[t, :, k + 1] = o[j].imag
                        k += 2
                    else:
                        y[t, :, k] = o[j]
                        k += 1
        return y


def run_typed(orc, wires, T=None):
    """orc: a typed FlowzOracle; wires: per input wire an array [T, n_streams] in its own dtype (float32 / float64 /
    complex64) -> per output wire an array [T, n_streams] in ITS dtype.  T: number of samples (needed without inputs)."""
    if T is None:
        T = np.shape(wires[0])[0] if wires else 0
    outs = None
    with np.errstate(all="ignore"):
        for t in range(T):
            o = orc.step(*[w[t] for w in wires])
            if outs is None:
                outs = [np.empty((T, orc.n_streams), v.dtype) for v in o]
            for j, v in enumerate(o):
                outs[j][t] = v
    return outs


def output_dtypes_typed(expr, in_dtypes=None):
    """ResultType of the expression (flowz.hpp:585-644): the type of every output wire when state keeps the pushed type."""
    return list(FlowzOracle(expr, 1, typed=True, in_dtypes=in_dtypes).out_types)


def output_dtypes(expr):
    """Arithmetic type of every output wire as the evaluator produces it: 'f32', 'f64' or 'cf32'
    (std::complex<float>, tests.cpp:206-207).
    Built-in operators promote by the usual arithmetic conversions (flowz.hpp:769-772; the cases of
    test/tests.cpp:200-231 without delays); a delayed read is float because compile() builds float
    delay lines (flowz.hpp:1245) -- the unused ResultType transform (:585-644) would keep the pushed type."""
    return list(FlowzOracle(expr, 1).out_types)


def compile(expr, n_streams: int = 1, params=None, out_f64: bool = False, typed: bool = False, in_dtypes=None) -> FlowzOracle:  # noqa: A001 (mirrors flowz::compile)
    return FlowzOracle(expr, n_streams, params, out_f64, typed, in_dtypes)


# ----------------------------------------------------------------------------------------
# deterministic synthetic input (SURVEY.md 8d / BASELINE.md 3): integer hash -> [-1, 1)
# ----------------------------------------------------------------------------------------

def hash32(seed, s, t):
    """murmur3 fmix32 of seed ^ s*0x9E3779B9 ^ t*0x85EBCA6B, applied twice (uint32 exact)."""
    with np.errstate(over="ignore"):
        s = np.asarray(s, dtype=np.uint64)
        t = np.asarray(t, dtype=np.uint64)
        h = (np.uint64(seed) ^ (s * np.uint64(0x9E3779B9)) ^ (t * np.uint64(0x85EBCA6B))) & np.uint64(0xFFFFFFFF)
        for _ in range(2):
            h ^= h >> np.uint64(16)
            h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
            h ^= h >> np.uint64(13)
            h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
            h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def u32_to_unit(h):
    """(float)(int32)(h >> 8) * 2^-23 - 1  in [-1, 1), exact in float32."""
    return ((h >> np.uint32(8)).astype(np.int32).astype(F32) * F32(2.0 ** -23) - F32(1.0)).astype(F32)


def synth_input(seed, streams, T, n_wires=1, t0=0):
    """Frames [T, len(streams), n_wires]; element (t, s, w) hashes (seed, s*n_wires + w, t0+t)."""
    streams = np.asarray(streams, dtype=np.uint64)
    t = (np.arange(T, dtype=np.uint64) + np.uint64(t0))[:, None, None]
    w = np.arange(n_wires, dtype=np.uint64)[None, None, :]
    sid = streams[None, :, None] * np.uint64(n_wires) + w
    return u32_to_unit(hash32(seed, sid, t))
