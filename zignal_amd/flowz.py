"""Python mirror of the Flowz EDSL (reference: flowz/flowz.hpp) over the C ABI of libflowz_hip.

Same surface as the C++ front end in include/flowz/flowz.hpp, spelled with what Python's
grammar allows:

    reference (C++)            here
    _1 .. _6                   _1 .. _6, placeholder(i)            flowz.hpp:1252-1257, :78-82
    _1[_2]                     _1[_2]  (or _1[-2], _1[2])          :84-85
    a , b                      chan(a, b, ...)  or a tuple (a, b)  :90
    a | b                      a | b                               :91
    a |= b                     a >> b  (`|=` is a statement in Python; `>>` is the sequence
                               operator of the reference's first prototype,
                               experimental_steps/wires_mono_only.cpp:37), seq(a, b, ...) for
                               C++'s right-associative chaining
    ~a                         ~a                                  :93
    + - * / unary -            same; Python numbers become float32 literal terminals   :68-72, :769-772
    std::ref(x)                param(k): per-stream, block-constant coefficient k (flowz/README.md:42-61)
    compile(expr)              compile(expr) -> Program            :1233-1249

A Program evaluates blocks on the GPU only (Program.run_block / Bank); there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, Optional, Sequence

from . import _capi as C
from ._capi import FlowzError, NoDeviceError, Variant  # noqa: F401  (re-exported)


class Expr:
    """Immutable handle of a Flowz expression tree (value semantics, flowz.hpp:46-61)."""

    __slots__ = ("_h",)

    def __init__(self, handle):
        self._h = C.check_ptr(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and C is not None:
            try:
                C.lib.fz_expr_release(h)
            except Exception:
                pass

    # -- analysis (flowz.hpp:162-246, :443-506) ------------------------------------------
    @property
    def ins(self) -> int:
        return C.check(C.lib.fz_input_arity(self._h))

    @property
    def outs(self) -> int:
        return C.check(C.lib.fz_output_arity(self._h))

    def max_input_delays(self):
        n = C.check(C.lib.fz_max_input_delays(self._h, None, 0))
        buf = (ctypes.c_uint32 * max(n, 1))()
        C.check(C.lib.fz_max_input_delays(self._h, buf, n))
        return tuple(buf[i] for i in range(n))

    # -- arithmetic ------------------------------------------------------------------------
    def _ar(self, op, other, swap=False):
        o = as_expr(other)
        a, b = (o, self) if swap else (self, o)
        return Expr(C.lib.fz_arith(op, a._h, b._h))

    def __add__(self, o): return self._ar(C.FZ_OP_ADD, o)
    def __radd__(self, o): return self._ar(C.FZ_OP_ADD, o, True)
    def __sub__(self, o): return self._ar(C.FZ_OP_SUB, o)
    def __rsub__(self, o): return self._ar(C.FZ_OP_SUB, o, True)
    def __mul__(self, o): return self._ar(C.FZ_OP_MUL, o)
    def __rmul__(self, o): return self._ar(C.FZ_OP_MUL, o, True)
    def __truediv__(self, o): return self._ar(C.FZ_OP_DIV, o)
    def __rtruediv__(self, o): return self._ar(C.FZ_OP_DIV, o, True)
    def __neg__(self): return Expr(C.lib.fz_arith(C.FZ_OP_NEG, self._h, None))

    # -- comparison and logical operators (proto::_default applies whatever C++ operator a node is, flowz.hpp:769-772): 1.0 / 0.0 -----------
    # (== and != are the methods eq / ne: Python needs __eq__ for its own purposes; `and` / `or` / `not` cannot be overloaded at all)
    def __lt__(self, o): return self._ar(C.FZ_OP_LT, o)
    def __le__(self, o): return self._ar(C.FZ_OP_LE, o)
    def __gt__(self, o): return self._ar(C.FZ_OP_GT, o)
    def __ge__(self, o): return self._ar(C.FZ_OP_GE, o)
    def eq(self, o): return self._ar(C.FZ_OP_EQ, o)
    def ne(self, o): return self._ar(C.FZ_OP_NE, o)
    def logical_and(self, o): return self._ar(C.FZ_OP_AND, o)
    def logical_or(self, o): return self._ar(C.FZ_OP_OR, o)
    def logical_not(self): return Expr(C.lib.fz_arith(C.FZ_OP_NOT, self._h, None))

    def __bool__(self):
        # (`_1 < _2` is an expression, not a Python truth value: `if a < b`, `max(a, b)`, `a and b` would silently take the wrong branch)
        raise TypeError("a Flowz expression has no truth value: comparisons build graph nodes (use .logical_and / .logical_or / .logical_not to combine them)")

    # -- combinators -----------------------------------------------------------------------
    def __or__(self, o): return Expr(C.lib.fz_parallel(self._h, as_expr(o)._h))
    def __ror__(self, o): return Expr(C.lib.fz_parallel(as_expr(o)._h, self._h))
    def __rshift__(self, o): return Expr(C.lib.fz_sequence(self._h, as_expr(o)._h))
    def __rrshift__(self, o): return Expr(C.lib.fz_sequence(as_expr(o)._h, self._h))
    def __invert__(self): return Expr(C.lib.fz_feedback(self._h))


class Placeholder(Expr):
    __slots__ = ("index",)

    def __init__(self, i: int):
        super().__init__(C.lib.fz_placeholder(int(i)))
        self.index = int(i)

    def __getitem__(self, d):
        """_i[_n] (reference syntax), _i[-n] (delay_expression.cpp:99-100 spelling) or _i[n]."""
        n = d.index if isinstance(d, Placeholder) else abs(int(d))
        return Expr(C.lib.fz_delayed(self.index, n))


def placeholder(i: int) -> Placeholder:
    return Placeholder(i)


_1, _2, _3, _4, _5, _6 = (Placeholder(i) for i in range(1, 7))


def lit(v: float) -> Expr:
    return Expr(C.lib.fz_literal(float(v)))


def lit64(v: float) -> Expr:
    """A C++ `double` literal terminal: the operators above it evaluate in float64 (usual arithmetic
    conversions); delay lines and frames stay float32.  Plain Python numbers are float32 literals."""
    return Expr(C.lib.fz_literal_f64(float(v)))


def litc(re: float, im: float = 0.0) -> Expr:
    """A std::complex<float> terminal (test/tests.cpp:206-207): the wire above it is complex and takes
    two float32 slots (re, im) of the output frame.  Python complex numbers become these."""
    return Expr(C.lib.fz_literal_c32(float(re), float(im)))


def litc64(re: float, im: float = 0.0) -> Expr:
    """A std::complex<double> terminal: as litc with double parts (z / w is libgcc's __divdc3, Smith's method).  In a
    typed program the wire takes four float32 slots: the double of the real part, then that of the imaginary part."""
    return Expr(C.lib.fz_literal_c64(float(re), float(im)))


def uniform(k: int, initial: float = 0.0) -> Expr:
    """Uniform run-time coefficient k (the std::ref(x) terminal): Program.set_uniform(k, v)."""
    return Expr(C.lib.fz_uniform(int(k), float(initial)))


def param(k: int) -> Expr:
    return Expr(C.lib.fz_stream_param(int(k)))


def modulator(k: int) -> Expr:
    """Sample-rate modulator k: the std::ref(x) terminal whose variable changes between calls (flowz/README.md:42-61), for
    block evaluation -- one value per sample, the same for all streams: Program.set_modulation(tensor [n_mod, rows])."""
    return Expr(C.lib.fz_modulator(int(k)))


def as_expr(x) -> Expr:
    if isinstance(x, Expr):
        return x
    if isinstance(x, tuple):
        return chan(*x)
    if isinstance(x, complex):
        return litc(x.real, x.imag)
    if isinstance(x, (int, float)) or hasattr(x, "__float__"):
        return lit(float(x))
    raise TypeError(f"cannot use {type(x).__name__} in a Flowz expression")


def chan(*xs) -> Expr:
    """(a, b, c): C++ comma is left-associative."""
    r = as_expr(xs[0])
    for x in xs[1:]:
        r = Expr(C.lib.fz_channel(r._h, as_expr(x)._h))
    return r


def par(*xs) -> Expr:
    r = as_expr(xs[0])
    for x in xs[1:]:
        r = r | as_expr(x)
    return r


def seq(*xs) -> Expr:
    """a |= b |= c: C++ `|=` is right-associative, a |= (b |= c)."""
    r = as_expr(xs[-1])
    for x in reversed(xs[:-1]):
        r = as_expr(x) >> r
    return r


_CMP_OPS = {"lt": C.FZ_OP_LT, "le": C.FZ_OP_LE, "gt": C.FZ_OP_GT, "ge": C.FZ_OP_GE, "eq": C.FZ_OP_EQ, "ne": C.FZ_OP_NE, "and": C.FZ_OP_AND, "or": C.FZ_OP_OR}


def from_sexpr(e) -> Expr:
    """Build from the neutral s-expression notation shared with the test-suite."""
    k = e[0]
    if k == "in": return Placeholder(e[1])
    if k == "del": return Placeholder(e[1])[int(e[2])]
    if k == "lit": return lit(e[1])
    if k == "lit64": return lit64(e[1])
    if k == "litc": return litc(e[1], e[2])
    if k == "litc64": return litc64(e[1], e[2])
    if k == "param": return param(e[1])
    if k == "mod": return modulator(e[1])
    if k == "uniform": return uniform(e[1], e[2])
    if k == "neg": return -from_sexpr(e[1])
    if k == "not": return from_sexpr(e[1]).logical_not()
    if k == "fb": return ~from_sexpr(e[1])
    a, b = from_sexpr(e[1]), from_sexpr(e[2])
    if k == "add": return a + b
    if k == "sub": return a - b
    if k == "mul": return a * b
    if k == "div": return a / b
    if k in _CMP_OPS: return a._ar(_CMP_OPS[k], b)
    if k == "chan": return chan(a, b)
    if k == "par": return a | b
    if k == "seq": return a >> b
    raise ValueError(f"unknown s-expression node {k!r}")


def make_variant(streams_per_lane=0, unroll=0, block_threads=0, flags=0) -> Variant:
    return Variant(int(streams_per_lane), int(unroll), int(block_threads), int(flags))


def _require(ok, what):
    """Argument check that survives `python -O` (the C ABI takes raw pointers: a wrong shape is an out-of-bounds access)."""
    if not ok:
        raise FlowzError(C.FZ_E_INVALID, str(what))


def _check_dev(t, shape, what, dtype=None):
    """A caller-supplied device buffer handed to the C ABI as a raw pointer: float32 (or `dtype`), CUDA, contiguous and of
    exactly the shape the kernel will address -- anything else would be a silent out-of-bounds device access."""
    import torch

    dtype = dtype or torch.float32
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
        raise FlowzError(C.FZ_E_INVALID, f"{what}: expected a contiguous CUDA {dtype} tensor of shape {tuple(shape)}, got "
                                         f"{t.dtype} {tuple(t.shape)} on {t.device}" + ("" if t.is_contiguous() else " (not contiguous)"))
    return t


def _check_frames(x, last, what="x"):
    """Input frames handed over as a raw pointer: contiguous float32 CUDA tensor whose last axis is the wire count."""
    import torch

    if not x.is_cuda:
        raise NoDeviceError(C.FZ_E_NO_DEVICE, f"{what}: needs CUDA (ROCm) tensors: zignal_amd has no CPU path")
    if not (x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == last):
        raise FlowzError(C.FZ_E_INVALID, f"{what}: expected contiguous float32 frames with {last} wire(s) on the last axis, got "
                                         f"{x.dtype} {tuple(x.shape)}" + ("" if x.is_contiguous() else " (not contiguous)"))
    return x


class Program:
    """compile() result: lowered graph + its fused gfx950 kernels (flowz.hpp:1233-1249)."""

    def __init__(self, expr, typed: bool = False, in_dtypes: Optional[Sequence[str]] = None):
        self.expr = as_expr(expr)
        h = ctypes.c_void_p()
        if typed or in_dtypes is not None:
            # ResultType semantics (flowz.hpp:585-644): wire types carried through inputs, state and outputs
            n = self.expr.ins
            dts = list(in_dtypes) if in_dtypes is not None else ["f32"] * n
            arr = (ctypes.c_uint32 * max(len(dts), 1))(*[C.DTYPES[d] for d in dts])
            C.check(C.lib.fz_compile_typed(self.expr._h, arr, len(dts), ctypes.byref(h)))
        else:
            C.check(C.lib.fz_compile(self.expr._h, ctypes.byref(h)))
        self._adopt(h)
        # a graph the shipped reference evaluates against its own arity table (fz_info.differs_from_reference): the library's note about it
        self.note = C.lib.fz_last_error().decode() if self.differs_from_reference else ""

    def _adopt(self, h):
        self._h = h
        info = C.Info()
        C.check(C.lib.fz_program_info(self._h, ctypes.byref(info)))
        self.info = info
        for f, _ in C.Info._fields_:
            setattr(self, f, getattr(info, f))

    def wave_part(self, n_parts: int, k: int) -> "Program":
        """Part k of the wave split into n_parts (FZ_VF_WAVES) as a program of its own, for inspection: its input is the cut
        wire before the part, its output the cut wire behind it."""
        h = ctypes.c_void_p()
        C.check(C.lib.fz_program_wave_part(self._h, int(n_parts), int(k), ctypes.byref(h)))
        q = Program.__new__(Program)
        q.expr = None
        q._adopt(h)
        return q

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and C is not None:
            try:
                C.lib.fz_program_destroy(h)
            except Exception:
                pass

    # -- inspection ------------------------------------------------------------------------
    def ir(self):
        n = C.check(C.lib.fz_program_ir(self._h, None, 0))
        buf = (C.IrNode * max(n, 1))()
        C.check(C.lib.fz_program_ir(self._h, buf, n))
        # 4th field: the literal of a 'const' node; the third operand of a 'select' node (a != 0 ? b : c)
        return [(C.IR_KINDS[buf[i].kind], buf[i].a, buf[i].b,
                 buf[i].c if buf[i].kind == 14 else (buf[i].value64 if buf[i].dtype else buf[i].value))
                for i in range(n)]

    def ir_dtypes(self):
        """per IR node: 'f32' or 'f64' (the node's C++ arithmetic type)"""
        n = C.check(C.lib.fz_program_ir(self._h, None, 0))
        buf = (C.IrNode * max(n, 1))()
        C.check(C.lib.fz_program_ir(self._h, buf, n))
        return ["f64" if buf[i].dtype else "f32" for i in range(n)]

    def outputs(self):
        buf = (ctypes.c_uint32 * max(self.n_out, 1))()
        C.check(C.lib.fz_program_outputs(self._h, buf, self.n_out))
        return [buf[i] for i in range(self.n_out)]

    def output_dtypes(self):
        """'f32' / 'f64' / 'cf32' per output WIRE: its C++ type before narrowing to the float32 frame
        (a 'cf32' wire, std::complex<float>, takes two frame slots: re, im)."""
        buf = (ctypes.c_uint32 * max(self.n_out, 1))()
        C.check(C.lib.fz_program_output_dtypes(self._h, buf, self.n_out))
        names = {0: "f32", 1: "f64", 2: "cf32", 4: "f64", 6: "cf64", 10: "cf64"}   # the first slot of a wire names it
        return [names[buf[i]] for i in range(self.n_out) if buf[i] in names]

    def output_slot_codes(self):
        """raw per-slot codes of fz_program_output_dtypes: 0 float, 1 double (narrowed to the float frame), 2 / 3 re / im
        of a complex wire, 4 / 5 low / high word of a double wire (typed programs), 6 / 7 / 8 / 9 low / high word of the
        real, low / high word of the imaginary part of a std::complex<double> wire (typed programs), 10 / 11 re / im of a
        std::complex<double> wire narrowed to the float frame (compile())"""
        buf = (ctypes.c_uint32 * max(self.n_out, 1))()
        C.check(C.lib.fz_program_output_dtypes(self._h, buf, self.n_out))
        return [buf[i] for i in range(self.n_out)]

    def input_dtypes(self):
        """'f32' / 'f64' / 'cf32' per input WIRE (typed programs; all 'f32' otherwise)"""
        n = max(self.n_in_wires, 1)
        buf = (ctypes.c_uint32 * n)()
        k = C.check(C.lib.fz_program_input_dtypes(self._h, buf, n))
        return [("f32", "f64", "cf32", "cf64")[buf[i]] for i in range(k)]

    def line_dtypes(self):
        """storage type per delay line (in lines() order): 'f32', 'f64' (two state rows per slot), 're' / 'im' (the
        float lines of a std::complex<float> wire)"""
        n = max(self.n_lines, 1)
        buf = (ctypes.c_uint32 * n)()
        k = C.check(C.lib.fz_program_line_dtypes(self._h, buf, n))
        return [("f32", "f64", "re", "im", "re64", "im64")[buf[i]] for i in range(k)]

    def lines(self):
        n = self.n_lines
        s, d = (ctypes.c_uint32 * max(n, 1))(), (ctypes.c_uint32 * max(n, 1))()
        C.check(C.lib.fz_program_lines(self._h, s, d, n))
        return [(s[i], d[i]) for i in range(n)]

    def consts(self):
        out = []
        for k in range(self.n_const):
            v = ctypes.c_float()
            C.check(C.lib.fz_program_get_const(self._h, k, ctypes.byref(v)))
            out.append(v.value)
        return out

    def set_const(self, slot: int, value: float):
        C.check(C.lib.fz_program_set_const(self._h, int(slot), float(value)))

    def set_uniform(self, k: int, value: float):
        C.check(C.lib.fz_program_set_uniform(self._h, int(k), float(value)))

    def set_modulation(self, mod):
        """mod: CUDA float32 [n_mod, rows] (rows >= the samples the frame buffers of the next launches hold): sample t of a block
        reads modulator k at mod[k, row0 + t].  The tensor must stay alive until those launches have run."""
        import torch

        if not (hasattr(mod, "is_cuda") and mod.is_cuda and mod.dtype == torch.float32 and mod.is_contiguous() and mod.dim() == 2
                and mod.shape[0] >= self.n_mod):
            raise FlowzError(C.FZ_E_INVALID, f"set_modulation: need a contiguous CUDA float32 tensor [>= {self.n_mod}, rows]")
        self._mod_keepalive = mod
        C.check(C.lib.fz_program_set_modulation(self._h, mod.data_ptr(), int(mod.shape[1])))

    def recommended_tile_streams(self) -> int:
        """Streams per frame tile that gives ~32 KiB row segments (see fz_run_block_tiled)."""
        return int(C.lib.fz_recommended_tile_streams(self._h))

    def kernel_name(self, variant: Optional[Variant] = None, n_streams: int = 0, n_samples: int = 0, tile_streams: int = 0) -> str:
        """Name of the kernel variant that a launch of this shape runs (kernel_symbol: the symbol profilers show).  tile_streams: the frame layout (0 = plain
        time-major rows, as for run_block on a 3-D tensor); stream-major frames: FZ_VF_STREAM_MAJOR in the variant's flags."""
        vp = ctypes.byref(variant) if variant is not None else None
        buf = ctypes.create_string_buffer(128)
        C.check(C.lib.fz_program_kernel_name(self._h, vp, int(n_streams), int(n_samples), int(tile_streams), buf, 128))
        return buf.value.decode()

    def kernel_symbol(self, variant: Optional[Variant] = None, n_streams: int = 0, n_samples: int = 0, tile_streams: int = 0) -> str:
        """kernel_name + "_g<graph tag>": the symbol in the code object, what rocprofv3 --kernel-trace --stats lists."""
        vp = ctypes.byref(variant) if variant is not None else None
        buf = ctypes.create_string_buffer(160)
        C.check(C.lib.fz_program_kernel_symbol(self._h, vp, int(n_streams), int(n_samples), int(tile_streams), buf, 160))
        return buf.value.decode()

    def kernel_code_id(self, variant: Optional[Variant] = None, n_streams: int = 0, n_samples: int = 0, tile_streams: int = 0) -> str:
        """16 hex digits naming the CODE of that kernel (hash of generated source + build options + compiler): the code object's file name
        in the kernel cache.  Needs no GPU and builds nothing."""
        vp = ctypes.byref(variant) if variant is not None else None
        buf = ctypes.create_string_buffer(32)
        C.check(C.lib.fz_program_kernel_code_id(self._h, vp, int(n_streams), int(n_samples), int(tile_streams), buf, 32))
        return buf.value.decode()

    def source(self, variant: Optional[Variant] = None) -> str:
        vp = ctypes.byref(variant) if variant is not None else None
        n = C.check(C.lib.fz_program_source(self._h, vp, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        C.check(C.lib.fz_program_source(self._h, vp, buf, n + 1))
        return buf.value.decode()

    def plan(self, n_streams: int, tile_streams: int = 0) -> Variant:
        """The variant a launch of this shape WITHOUT a variant would use on the current device: tuned in this process,
        else persisted by an earlier one (plans.txt in the kernel cache), else Variant(0,0,0,0) = the library default."""
        v = Variant(0, 0, 0, 0)
        C.check(C.lib.fz_program_plan(self._h, int(n_streams), int(tile_streams), ctypes.byref(v)))
        return v

    def tune_candidates(self, n_streams: int, n_samples: int, tile_streams: int = 0):
        """The variants Program.tune would measure for this shape (the first one is the library default)."""
        n = C.check(C.lib.fz_program_tune_candidates(self._h, int(n_streams), int(n_samples), int(tile_streams), None, 0))
        buf = (Variant * max(n, 1))()
        C.check(C.lib.fz_program_tune_candidates(self._h, int(n_streams), int(n_samples), int(tile_streams), buf, n))
        return [Variant(buf[i].streams_per_lane, buf[i].unroll, buf[i].block_threads, buf[i].flags) for i in range(n)]

    def build(self, variant: Optional[Variant] = None, n_streams: int = 0, n_samples: int = 0, tile_streams: int = 0):
        """JIT-compile (or fetch from the on-disk cache) the kernel of `variant`; needs no GPU.  With a block shape the
        variant's automatic fields resolve as a launch of that shape would (else as for a large stream count)."""
        vp = ctypes.byref(variant) if variant is not None else None
        if n_streams:
            C.check(C.lib.fz_program_build_for(self._h, vp, int(n_streams), int(n_samples or 4096), int(tile_streams)))
        else:
            C.check(C.lib.fz_program_build(self._h, vp))
        return self

    def kernel_resources(self, variant: Optional[Variant] = None, n_streams: int = 1 << 20, n_samples: int = 4096,
                         as_launched: bool = True, tile_streams: int = 0) -> dict:
        """registers / LDS / scratch bytes per lane of a variant's kernel (JITs it; needs no GPU).  as_launched: with the
        unroll lowered until nothing spills, as run_block does ('unroll' = what runs); False: the variant exactly as given."""
        vp = ctypes.byref(variant) if variant is not None else None
        r = C.KernelResources()
        C.check(C.lib.fz_program_kernel_resources(self._h, vp, int(n_streams), int(n_samples), int(tile_streams), int(as_launched), ctypes.byref(r)))
        return {n: getattr(r, n) for n, _ in C.KernelResources._fields_}

    # -- the hot path ----------------------------------------------------------------------
    def run_block_ptr(self, in_ptr, out_ptr, state_ptr, params_ptr, n_streams, n_samples,
                      variant: Optional[Variant] = None, stream=None, tile_streams: int = 0):
        vp = ctypes.byref(variant) if variant is not None else None
        if tile_streams:
            C.check(C.lib.fz_run_block_tiled(self._h, in_ptr, out_ptr, state_ptr, params_ptr, int(n_streams),
                                             int(n_samples), int(tile_streams), vp, stream))
        else:
            C.check(C.lib.fz_run_block(self._h, in_ptr, out_ptr, state_ptr, params_ptr, int(n_streams),
                                       int(n_samples), vp, stream))

    def run_window(self, x, out, state, row0: int, n_samples: int, params=None, variant: Optional[Variant] = None):
        """Samples [row0, row0 + n_samples) of the frame buffers x / out (laid out as for run_block, holding
        more samples than the block): fz_run_block_window.  state advances; params as for run_block."""
        import torch

        _check_frames(x, max(self.n_in, 1))
        if x.dim() == 4:
            n_tiles, rows, tile, _ = x.shape
            ns = n_tiles * tile
            oshape = (n_tiles, rows, tile, self.n_out)
        else:
            rows, ns, _ = x.shape
            tile = 0
            oshape = (rows, ns, self.n_out)
        _check_dev(out, oshape, "out")
        if self.n_state:
            _check_dev(state, (self.n_state, ns), "state")
        pp = _check_dev(params, (self.n_param, ns), "params").data_ptr() if self.n_param else None
        vp = ctypes.byref(variant) if variant is not None else None
        C.check(C.lib.fz_run_block_window(self._h, x.data_ptr() if self.n_in else None, out.data_ptr(),
                                          state.data_ptr() if self.n_state else None, pp, ns, rows, int(row0), int(n_samples),
                                          tile, vp, torch.cuda.current_stream().cuda_stream))
        return out, state

    def run_block_stream_major(self, x, state=None, params=None, out=None, variant: Optional[Variant] = None, row0: int = 0,
                               n_samples: Optional[int] = None):
        """x: CUDA float32 [n_streams, rows, n_in] -- one contiguous buffer per stream (the reference's calling
        convention); out [n_streams, rows, n_out].  No layout pass (fz_run_block_stream_major); the block is
        rows [row0, row0 + n_samples).  Returns (out, state)."""
        import torch

        if x.dim() == 2:
            x = x.unsqueeze(-1)
        _check_frames(x, max(self.n_in, 1))
        ns, rows, _ = x.shape
        n = rows - row0 if n_samples is None else int(n_samples)
        if out is None:
            out = torch.empty((ns, rows, self.n_out), dtype=torch.float32, device=x.device)
        if state is None:
            state = torch.zeros((max(self.n_state, 1), ns), dtype=torch.float32, device=x.device)
        _check_dev(out, (ns, rows, self.n_out), "out")
        _check_dev(state, (max(self.n_state, 1), ns), "state")
        pp = _check_dev(params, (self.n_param, ns), "params").data_ptr() if self.n_param else None
        vp = ctypes.byref(variant) if variant is not None else None
        C.check(C.lib.fz_run_block_stream_major(self._h, x.data_ptr() if self.n_in else None, out.data_ptr(),
                                                state.data_ptr() if self.n_state else None, pp, ns, rows, int(row0), n, vp,
                                                torch.cuda.current_stream().cuda_stream))
        return out, state

    def run_block(self, x, state=None, params=None, out=None, variant: Optional[Variant] = None, out_f64: bool = False):
        """x: CUDA float32 frames, either time-major [T, n_streams, n_in] or stream-tiled
        [n_tiles, T, tile_streams, n_in] (the HBM-friendly layout, see fz_run_block_tiled).
        state: [n_state, n_streams] in/out (allocated zeroed when None), params: [n_param, n_streams].
        out_f64: float64 output frames (results of double sub-expressions leave un-narrowed).
        Launches on torch's current stream; returns (out, state), out laid out like x."""
        import torch

        if not x.is_cuda:
            raise NoDeviceError(C.FZ_E_NO_DEVICE, "run_block needs CUDA (ROCm) tensors: zignal_amd has no CPU path")
        if x.dim() == 2:
            x = x.unsqueeze(-1)
        _check_frames(x, self.n_in)
        if x.dim() == 4:
            n_tiles, T, tile, _ = x.shape
            ns = n_tiles * tile
            oshape = (n_tiles, T, tile, self.n_out)
        else:
            T, ns, _ = x.shape
            tile = 0
            oshape = (T, ns, self.n_out)
        odt = torch.float64 if out_f64 else torch.float32
        if out_f64:
            v0 = variant if variant is not None else Variant(0, 0, 0, 0)
            variant = Variant(v0.streams_per_lane, v0.unroll, v0.block_threads, v0.flags | C.FZ_VF_OUT_F64)
        if out is None:
            out = torch.empty(oshape, dtype=odt, device=x.device)
        _check_dev(out, oshape, "out", odt)
        if state is None:
            state = torch.zeros((max(self.n_state, 1), ns), dtype=torch.float32, device=x.device)
        if self.n_state:
            _check_dev(state, (self.n_state, ns), "state")
        pp = None
        if self.n_param:
            if params is None:
                raise FlowzError(C.FZ_E_INVALID, f"params: the graph has {self.n_param} per-stream coefficient(s), none given")
            pp = _check_dev(params, (self.n_param, ns), "params").data_ptr()
        self.run_block_ptr(x.data_ptr() if self.n_in else None, out.data_ptr(),
                           state.data_ptr() if self.n_state else None, pp, ns, T, variant,
                           torch.cuda.current_stream().cuda_stream, tile)
        return out, state


def _tune(self, x, state=None, params=None, out=None):
    """Measure the candidate kernel variants for this shape on these buffers (fz_program_tune) and
    remember the fastest: later run_block calls of the same shape without a variant use it.
    x as for run_block; `state` advances (a scratch one is used when None), `out` is overwritten.
    Returns (Variant, milliseconds per block)."""
    import torch

    _check_frames(x, x.shape[-1])
    if x.dim() == 4:
        n_tiles, T, tile, _ = x.shape
        ns = n_tiles * tile
        oshape = (n_tiles, T, tile, self.n_out)
    else:
        T, ns, _ = x.shape
        tile = 0
        oshape = (T, ns, self.n_out)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    if state is None:
        state = torch.zeros((max(self.n_state, 1), ns), dtype=torch.float32, device=x.device)
    _check_dev(out, oshape, "out")
    _check_dev(state, (max(self.n_state, 1), ns), "state")
    pp = _check_dev(params, (self.n_param, ns), "params").data_ptr() if self.n_param else None
    chosen, ms = Variant(0, 0, 0, 0), ctypes.c_float(0)
    C.check(C.lib.fz_program_tune(self._h, x.data_ptr() if self.n_in else None, out.data_ptr(),
                                  state.data_ptr() if self.n_state else None, pp, ns, T, tile,
                                  torch.cuda.current_stream().cuda_stream, ctypes.byref(chosen), ctypes.byref(ms)))
    return chosen, float(ms.value)


Program.tune = _tune


class Bank:
    """Device-resident closure state of n_streams streams of one program (fz_bank: the `state_` member of
    the reference's stateful_lambda, flowz.hpp:1190-1191, times n_streams)."""

    def __init__(self, prog: Program, n_streams: int):
        self.prog, self.n_streams = prog, int(n_streams)
        h = ctypes.c_void_p()
        C.check(C.lib.fz_bank_create(prog._h, self.n_streams, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and C is not None:
            C.lib.fz_bank_destroy(h)

    def reset(self):
        C.check(C.lib.fz_bank_reset(self._h))

    def set_params(self, params):
        """params: numpy float32 [n_param, n_streams]"""
        import numpy as np
        p = np.ascontiguousarray(params, dtype=np.float32)
        _require(p.shape == (self.prog.n_param, self.n_streams), "p: wrong shape, dtype, layout or device for this call")
        C.check(C.lib.fz_bank_set_params_host(self._h, p.ctypes.data))

    def process_blocks(self, x, out, block_len: int, params_blocks=None, variant: Optional[Variant] = None):
        """Control-rate modulation: the frame buffers x / out (CUDA, time-major or stream-tiled) are processed in
        blocks of block_len samples, block k with the per-stream coefficient set params_blocks[k]
        (CUDA float32 [n_blocks, n_param, n_streams]); fz_bank_process_blocks."""
        import torch

        _require(x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == max(self.prog.n_in, 1), "x: wrong shape, dtype, layout or device for this call")
        rows, tile = (x.shape[1], x.shape[2]) if x.dim() == 4 else (x.shape[0], 0)
        ns = x.shape[0] * x.shape[2] if x.dim() == 4 else x.shape[1]
        if ns != self.n_streams:
            raise FlowzError(C.FZ_E_INVALID, f"frames hold {ns} streams, the bank {self.n_streams}")
        _check_dev(out, tuple(x.shape[:-1]) + (self.prog.n_out,), "out")
        pp = None
        if params_blocks is not None:
            nb = (rows + block_len - 1) // block_len
            pp = _check_dev(params_blocks, (nb, self.prog.n_param, self.n_streams), "params_blocks").data_ptr()
        vp = ctypes.byref(variant) if variant is not None else None
        C.check(C.lib.fz_bank_process_blocks(self._h, x.data_ptr() if self.prog.n_in else None, out.data_ptr(), rows,
                                             int(block_len), pp, tile, vp, torch.cuda.current_stream().cuda_stream))
        return out

    def process_host_stream_major(self, x, out=None):
        """Host buffers, one contiguous row per stream: x [n_streams, T, n_in] -> [n_streams, T, n_out] (numpy
        arrays or CPU torch tensors; pinned memory overlaps the PCIe directions)."""
        import numpy as np

        is_torch = hasattr(x, "data_ptr")
        T = int(x.shape[1])
        if is_torch:
            import torch
            _require(x.dtype == torch.float32 and x.is_contiguous() and not x.is_cuda, "x: wrong shape, dtype, layout or device for this call")
            if out is None:
                out = torch.empty((self.n_streams, T, self.prog.n_out), dtype=torch.float32, pin_memory=x.is_pinned())
            xp, op = x.data_ptr(), out.data_ptr()
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            if out is None:
                out = np.empty((self.n_streams, T, self.prog.n_out), np.float32)
            xp, op = x.ctypes.data, out.ctypes.data
        _require(tuple(x.shape) in ((self.n_streams, T, max(self.prog.n_in, 1)), (self.n_streams, T)), x.shape)
        _require(tuple(out.shape) == (self.n_streams, T, self.prog.n_out) and (out.is_contiguous() if is_torch else out.flags.c_contiguous), "out: wrong shape, dtype, layout or device for this call")
        _require((out.dtype == torch.float32) if is_torch else (out.dtype == np.float32), "out: wrong shape, dtype, layout or device for this call")
        C.check(C.lib.fz_bank_process_host_stream_major(self._h, xp if self.prog.n_in else None, op, T))
        return out

    def process_host(self, x, out=None, out_f64: bool = False):
        """Host frames in, host frames out (time-major [T, n_streams, n_in] float32 -> [T, n_streams, n_out]).
        x / out: numpy arrays or CPU torch tensors; pinned tensors let both PCIe directions overlap with
        the kernels (long blocks are pipelined in time chunks)."""
        import numpy as np

        is_torch = hasattr(x, "data_ptr")
        T = int(x.shape[0])
        if is_torch:
            import torch
            _require(x.dtype == torch.float32 and x.is_contiguous() and not x.is_cuda, "x: wrong shape, dtype, layout or device for this call")
            if out is None:
                out = torch.empty((T, self.n_streams, self.prog.n_out), dtype=torch.float64 if out_f64 else torch.float32,
                                  pin_memory=x.is_pinned())
            xp, op = x.data_ptr(), out.data_ptr()
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            if out is None:
                out = np.empty((T, self.n_streams, self.prog.n_out), np.float64 if out_f64 else np.float32)
            xp, op = x.ctypes.data, out.ctypes.data
        _require(tuple(x.shape[1:]) in ((self.n_streams, self.prog.n_in), (self.n_streams,)) or self.prog.n_in == 0, "x: wrong shape, dtype, layout or device for this call")
        _require(tuple(out.shape) == (T, self.n_streams, self.prog.n_out) and (out.is_contiguous() if is_torch else out.flags.c_contiguous), "out: wrong shape, dtype, layout or device for this call")
        want_dt = (torch.float64 if out_f64 else torch.float32) if is_torch else (np.float64 if out_f64 else np.float32)
        _require(out.dtype == want_dt, (out.dtype, want_dt))
        fn = C.lib.fz_bank_process_host_f64 if out_f64 else C.lib.fz_bank_process_host
        C.check(fn(self._h, xp if self.prog.n_in else None, op, T))
        return out


Program.bank = lambda self, n_streams: Bank(self, n_streams)


def to_tiled(x, tile_streams: int):
    """time-major [T, n_streams, w] -> stream-tiled [n_tiles, T, tile_streams, w] (torch or numpy)."""
    T, ns, w = x.shape
    _require(ns % tile_streams == 0, "the stream count must be a multiple of tile_streams")
    y = x.reshape(T, ns // tile_streams, tile_streams, w)
    y = y.permute(1, 0, 2, 3).contiguous() if hasattr(y, "permute") else y.transpose(1, 0, 2, 3).copy()
    return y


def from_tiled(y):
    """stream-tiled [n_tiles, T, tile_streams, w] -> time-major [T, n_streams, w]."""
    n_tiles, T, tile, w = y.shape
    z = y.permute(1, 0, 2, 3) if hasattr(y, "permute") else y.transpose(1, 0, 2, 3)
    return z.reshape(T, n_tiles * tile, w)


def compile(expr, typed: bool = False, in_dtypes: Optional[Sequence[str]] = None) -> Program:  # noqa: A001  (mirrors flowz::compile)
    """typed / in_dtypes: fz_compile_typed -- every wire keeps its C++ type through inputs, state and outputs (ResultType,
    flowz.hpp:585-644); frames then hold 'f64' and 'cf32' wires in two float slots (see pack_typed / unpack_typed)."""
    return Program(expr, typed, in_dtypes)


def pack_typed(wires, dtypes):
    """numpy helper: per-wire arrays [T, n_streams] (float32 / float64 / complex64 / complex128) -> float32 frames [T, n_streams, slots]
    as a typed program reads them."""
    import numpy as np

    cols = []
    for w, dt in zip(wires, dtypes):
        if dt == "f32":
            cols.append(np.asarray(w, np.float32)[..., None])
        elif dt == "f64":
            cols.append(np.ascontiguousarray(np.asarray(w, np.float64)).view(np.float32).reshape(np.shape(w) + (2,)))
        elif dt == "cf64":
            cols.append(np.ascontiguousarray(np.asarray(w, np.complex128)).view(np.float32).reshape(np.shape(w) + (4,)))
        else:
            cols.append(np.ascontiguousarray(np.asarray(w, np.complex64)).view(np.float32).reshape(np.shape(w) + (2,)))
    return np.ascontiguousarray(np.concatenate(cols, axis=-1))


def unpack_typed(frames, dtypes):
    """inverse of pack_typed: float32 frames [T, n_streams, slots] -> list of per-wire arrays in their own dtype"""
    import numpy as np

    frames = np.ascontiguousarray(frames, np.float32)
    out, k = [], 0
    for dt in dtypes:
        if dt == "f32":
            out.append(frames[..., k].copy())
            k += 1
        elif dt == "cf64":
            quad = np.ascontiguousarray(frames[..., k:k + 4])
            out.append(quad.view(np.complex128)[..., 0])
            k += 4
        else:
            pair = np.ascontiguousarray(frames[..., k:k + 2])
            out.append(pair.view(np.float64 if dt == "f64" else np.complex64)[..., 0])
            k += 2
    return out


def manifest_build(path: str, workers: int = 0) -> dict:
    """Replay a kernel manifest (FLOWZ_HIP_MANIFEST=<file> records one while a process runs; *.gz is unpacked first): build, in `workers`
    parallel compiler processes and without a GPU, every kernel of it the kernel cache lacks.  Returns the counts."""
    import gzip
    import os
    import tempfile

    workers = workers or max(1, len(os.sched_getaffinity(0)))
    counts = (ctypes.c_uint32 * 4)()
    if path.endswith(".gz"):
        with tempfile.NamedTemporaryFile(suffix=".fzm") as tmp:
            tmp.write(gzip.open(path, "rb").read())
            tmp.flush()
            C.check(C.lib.fz_manifest_build(tmp.name.encode(), workers, counts))
    else:
        C.check(C.lib.fz_manifest_build(path.encode(), workers, counts))
    return dict(zip(("records", "at_hand", "built", "failed"), (int(c) for c in counts)))


def device_count() -> int:
    return C.lib.fz_device_count()


def synth_fill(dst, seed: int, stream0: int = 0, t0: int = 0):
    """Fill CUDA tensor dst with the deterministic hash noise: time-major [T, n_streams, n_wires]
    or stream-tiled [n_tiles, T, tile_streams, n_wires] (same values, tiled placement)."""
    import torch

    if dst.dim() == 4:
        n_tiles, T, tile, nw = dst.shape
        ns = n_tiles * tile
    else:
        T, ns, nw = dst.shape
        tile = 0
    _require(dst.is_cuda and dst.dtype == torch.float32 and dst.is_contiguous(), "dst: wrong shape, dtype, layout or device for this call")
    C.check(C.lib.fz_synth_fill(dst.data_ptr(), ns, T, nw, int(seed), int(stream0), int(t0), int(tile),
                                torch.cuda.current_stream().cuda_stream))
    return dst


def rbj_lowpass(freq, q, sample_rate: float, raw6=None, df1=None):
    """Per-stream RBJ low-pass coefficients on the device (reactive_filter_coeff.cpp:38-58).
    freq, q: CUDA float32 [n]; raw6: [6, n] (a0 a1 a2 b0 b1 b2) and/or df1: [5, n] rows
    (b0/a0 b1/a0 b2/a0 -a1/a0 -a2/a0), e.g. a slice of a `params` buffer."""
    import torch

    n = freq.numel()
    _require(freq.is_cuda and q.is_cuda and q.numel() == n and freq.dtype == q.dtype == torch.float32, "freq: wrong shape, dtype, layout or device for this call")
    for t, rows in ((raw6, 6), (df1, 5)):
        _require(t is None or (t.is_cuda and t.is_contiguous() and tuple(t.shape) == (rows, n)), "t: wrong shape, dtype, layout or device for this call")
    C.check(C.lib.fz_rbj_lowpass(freq.data_ptr(), q.data_ptr(), float(sample_rate), n,
                                 raw6.data_ptr() if raw6 is not None else None,
                                 df1.data_ptr() if df1 is not None else None,
                                 torch.cuda.current_stream().cuda_stream))


def frames_from_stream_major(x, tile_streams: int = 0, out=None):
    """x: CUDA float32 [n_streams, n_samples, n_wires] (one contiguous buffer per stream, as the reference's
    closures consume them) -> frames for run_block: [n_samples, n_streams, n_wires], or stream-tiled
    [n_tiles, n_samples, tile_streams, n_wires].  One device pass (fz_transpose_frames)."""
    import torch

    if x.dim() == 2:
        x = x.unsqueeze(-1)
    ns, T, w = x.shape
    _check_frames(x, x.shape[-1])
    tile = tile_streams if tile_streams and tile_streams < ns else 0
    shape = (ns // tile, T, tile, w) if tile else (T, ns, w)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _require(tuple(out.shape) == shape and out.is_contiguous(), "out: wrong shape, dtype, layout or device for this call")
    C.check(C.lib.fz_transpose_frames(x.data_ptr(), out.data_ptr(), ns, T, w, tile, 0, torch.cuda.current_stream().cuda_stream))
    return out


def frames_to_stream_major(y, out=None):
    """frames ([n_samples, n_streams, w] or stream-tiled [n_tiles, n_samples, tile, w]) -> [n_streams, n_samples, w]."""
    import torch

    _require(y.is_cuda and y.dtype == torch.float32 and y.is_contiguous(), "y: wrong shape, dtype, layout or device for this call")
    if y.dim() == 4:
        n_tiles, T, tile, w = y.shape
        ns = n_tiles * tile
    else:
        T, ns, w = y.shape
        tile = 0
    if out is None:
        out = torch.empty((ns, T, w), dtype=torch.float32, device=y.device)
    _require(tuple(out.shape) == (ns, T, w) and out.is_contiguous(), "out: wrong shape, dtype, layout or device for this call")
    C.check(C.lib.fz_transpose_frames(y.data_ptr(), out.data_ptr(), ns, T, w, tile, 1, torch.cuda.current_stream().cuda_stream))
    return out


def copy_probe(src, dst):
    import torch

    _require(src.is_cuda and dst.is_cuda and src.numel() == dst.numel() and src.numel() % 4 == 0, "src: wrong shape, dtype, layout or device for this call")
    C.check(C.lib.fz_copy_probe(src.data_ptr(), dst.data_ptr(), src.numel(),
                                torch.cuda.current_stream().cuda_stream))
