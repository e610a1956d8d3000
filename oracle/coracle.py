"""ctypes wrapper of oracle/libflowz_oracle.so (compiled scalar closures) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All entry points take/return numpy float32 arrays of time-major frames [T, n_streams, n_wires]
(the device layout) unless `stream_major=True` ([n_streams, T, n_wires], the CPU-friendly
layout used for the timed CPU baseline).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
F32 = np.float32
_fp = ctypes.POINTER(ctypes.c_float)
_pd = ctypes.c_ssize_t


class Coef(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("b0", "b1", "b2", "a1", "a2")]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libflowz_oracle.so", "libflowz_oracle_std.so"])


_LIB_STD = None


def lib_std():
    """std::complex<float> spelling of the complex graphs (oracle/complex_std.cpp)."""
    global _LIB_STD
    if _LIB_STD is None:
        path = os.path.join(_HERE, "libflowz_oracle_std.so")
        if not os.path.exists(path):
            build()
        _LIB_STD = ctypes.CDLL(path)
    return _LIB_STD


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libflowz_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(_fp)


def _strides(T, ns, nw, stream_major):
    # (stream stride, time stride) in floats
    return (T * nw, nw) if stream_major else (nw, ns * nw)


def _prep(x, n_in, stream_major):
    x = np.ascontiguousarray(x, dtype=F32)
    if x.ndim == 2:
        x = x[:, :, None]
    if stream_major:
        ns, T, nw = x.shape
    else:
        T, ns, nw = x.shape
    assert nw == n_in, (nw, n_in)
    return x, T, ns


def _coefs(cs):
    arr = (Coef * len(cs))()
    for i, c in enumerate(cs):
        arr[i] = Coef(*[float(F32(v)) for v in c])
    return arr


def _run(fname, pre_args, x, n_in, n_out, stream_major, out_dtype=F32, library=None, out=None):
    x, T, ns = _prep(x, n_in, stream_major)
    shape = (ns, T, n_out) if stream_major else (T, ns, n_out)
    if out is None:
        y = np.empty(shape, out_dtype)
    else:                                   # caller-owned (pre-touched) result buffer: nothing is allocated in the call
        y = out
        assert y.shape == shape and y.dtype == out_dtype and y.flags.c_contiguous
    xss, xts = _strides(T, ns, n_in, stream_major)
    yss, yts = _strides(T, ns, n_out, stream_major)
    getattr(library or lib(), fname)(*pre_args, _p(x), _pd(xss), _pd(xts), _p(y), _pd(yss), _pd(yts),
                          ctypes.c_long(ns), ctypes.c_long(T))
    return y


def df1_cascade(coefs, x, stream_major=False, out=None):
    return _run("fzo_df1_cascade", (_coefs(coefs), ctypes.c_int(len(coefs))), x, 1, 1, stream_major, out=out)


def df1_cascade_soa(coefs, x, out=None):
    """Vectorised-across-streams variant ("Mode B"): x time-major [T, n_streams] or [T, n_streams, 1] -> same shape."""
    x = np.ascontiguousarray(x, dtype=F32)
    T, ns = x.shape[0], x.shape[1]
    y = np.empty_like(x) if out is None else out
    assert y.shape == x.shape and y.dtype == F32 and y.flags.c_contiguous
    lib().fzo_df1_cascade_soa(_coefs(coefs), ctypes.c_int(len(coefs)), _p(x), _p(y), ctypes.c_long(ns), ctypes.c_long(T))
    return y


def df2(c, x, stream_major=False):
    return _run("fzo_df2", (_coefs([c]),), x, 1, 1, stream_major)


def df1t(c, x, stream_major=False):
    return _run("fzo_df1t", (_coefs([c]),), x, 1, 1, stream_major)


def df2t_flowz(c, x, stream_major=False):
    return _run("fzo_df2t_flowz", (_coefs([c]),), x, 1, 1, stream_major)


def integrator(x, stream_major=False):
    return _run("fzo_integrator", (), x, 1, 1, stream_major)


def one_quad(x, c1=0.9, c2=0.8, stream_major=False):
    return _run("fzo_one_quad", (ctypes.c_float(float(F32(c1))), ctypes.c_float(float(F32(c2)))), x, 1, 1, stream_major)


def cross_wire(x, c1=0.9, c2=0.2, stream_major=False):
    return _run("fzo_cross_wire", (ctypes.c_float(float(F32(c1))), ctypes.c_float(float(F32(c2)))), x, 1, 2, stream_major)


def par4_sum(coefs4, x, fanout=False, stream_major=False):
    return _run("fzo_par4_sum", (_coefs(coefs4), ctypes.c_int(int(fanout))), x, 1 if fanout else 4, 1, stream_major)


def osc_chain(params, x, n_stage=6, stream_major=False):
    """params: [1+5*n_stage, n_streams] planar per-stream coefficients."""
    params = np.ascontiguousarray(params, dtype=F32)
    assert params.shape[0] == 1 + 5 * n_stage
    return _run("fzo_osc_chain", (_p(params), _pd(params.shape[1]), ctypes.c_int(n_stage)), x, 1, 1, stream_major)


def one_pole_readme(a, x, stream_major=False, out_f64=False):
    if out_f64:
        return _run("fzo_one_pole_readme_f64out", (ctypes.c_float(float(F32(a))),), x, 1, 1, stream_major, np.float64)
    return _run("fzo_one_pole_readme", (ctypes.c_float(float(F32(a))),), x, 1, 1, stream_major)


def mixed_precision_biquad(x, b=(0.05, -0.075, 0.275), a=(0.2, -0.8), stream_major=False, out_f64=False):
    pre = tuple(ctypes.c_double(float(v)) for v in b) + tuple(ctypes.c_float(float(F32(v))) for v in a)
    if out_f64:
        return _run("fzo_mixed_precision_biquad_f64out", pre, x, 1, 1, stream_major, np.float64)
    return _run("fzo_mixed_precision_biquad", pre, x, 1, 1, stream_major)


def complex_mix(x, A=(0.6, 0.8), B=(0.3, -0.4), c=(0.5, 0.25, 1.5, -0.125, 0.75), stream_major=False, std=False):
    """-> frames of 3 slots (re, im, integrator): tests/graphs.py complex_mix.
    std=True: the std::complex<float> spelling (oracle/complex_std.cpp) instead of float _Complex."""
    cc = np.ascontiguousarray(c, F32)
    pre = tuple(ctypes.c_float(float(F32(v))) for v in (*A, *B)) + (_p(cc),)
    if std:
        return _run("fzo_complex_mix_std", pre, x, 1, 3, stream_major, library=lib_std())
    return _run("fzo_complex_mix", pre, x, 1, 3, stream_major)


def complex_one_pole(x, c=(0.6, 0.7), stream_major=False, std=False):
    """-> frames (re, im): tests/graphs.py complex_one_pole (typed programs: complex state)"""
    pre = (ctypes.c_float(float(F32(c[0]))), ctypes.c_float(float(F32(c[1]))))
    if std:
        return _run("fzo_complex_one_pole_std", pre, x, 1, 2, stream_major, library=lib_std())
    return _run("fzo_complex_one_pole", pre, x, 1, 2, stream_major)


def complex_div_mix(x, A=(0.6, 0.8), B=(1.5, -0.75), stream_major=False, std=False):
    """-> frames (re, im): tests/graphs.py complex_div_mix (complex / complex and scalar / complex)"""
    pre = tuple(ctypes.c_float(float(F32(v))) for v in (*A, *B))
    if std:
        return _run("fzo_complex_div_mix_std", pre, x, 1, 2, stream_major, library=lib_std())
    return _run("fzo_complex_div_mix", pre, x, 1, 2, stream_major)


def cdouble_resonator(x, C=(0.6, 0.7), B=(1.5, -0.75), std=False):
    """x: float64 [T, n_streams] -> complex128 [T, n_streams]: tests/graphs.py cdouble_resonator (std::complex<double> state
    and both spellings of __divdc3).  std=True: the std::complex<double> spelling (oracle/complex_std.cpp)."""
    x = np.ascontiguousarray(x, np.float64)
    T, ns = x.shape
    y = np.empty((T, ns, 2), np.float64)
    L = lib_std() if std else lib()
    fn = getattr(L, "fzo_cdouble_resonator_std" if std else "fzo_cdouble_resonator")
    fn.restype = None
    fn(*(ctypes.c_double(float(v)) for v in (*C, *B)), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
       ctypes.c_long(ns), ctypes.c_long(T))
    return y.view(np.complex128)[..., 0]


def double_accumulator(x, stream_major=False):
    """-> float64 frames: tests/graphs.py double_accumulator with double state"""
    return _run("fzo_double_accumulator", (), x, 1, 1, stream_major, np.float64)


def rbj_lowpass(freq, q, sr, libmf=False):
    """-> (raw6 [6,n], df1 [5,n]); libmf=True: the reference's own float sinf/cosf spelling (raw6 only)."""
    freq = np.ascontiguousarray(freq, F32)
    q = np.ascontiguousarray(q, F32)
    n = len(freq)
    raw6 = np.empty((6, n), F32)
    if libmf:
        lib().fzo_rbj_lowpass_libmf(_p(freq), _p(q), ctypes.c_float(sr), ctypes.c_long(n), _p(raw6))
        return raw6
    df1 = np.empty((5, n), F32)
    lib().fzo_rbj_lowpass(_p(freq), _p(q), ctypes.c_float(sr), ctypes.c_long(n), _p(raw6), _p(df1))
    return raw6, df1


def sincos_f32(x, libm=False):
    """-> (sin, cos) float32 of float32 arguments: the checker's polynomial pair (what the RBJ generator uses on both sides), or with
    libm=True glibc's sinf / cosf -- what the reference's std::sin / std::cos of a float are on this box."""
    x = np.ascontiguousarray(x, F32)
    sn, cs = np.empty_like(x), np.empty_like(x)
    lib().fzo_sincos_array(_p(x), ctypes.c_long(x.size), ctypes.c_int(1 if libm else 0), _p(sn), _p(cs))
    return sn, cs


def synth_fill(seed, stream0, n_streams, T, n_wires=1, t0=0, stream_major=False):
    out = np.empty((n_streams, T, n_wires) if stream_major else (T, n_streams, n_wires), F32)
    ss, ts = _strides(T, n_streams, n_wires, stream_major)
    lib().fzo_synth_fill(_p(out), _pd(ss), _pd(ts), ctypes.c_uint32(seed), ctypes.c_uint64(stream0),
                         ctypes.c_long(n_streams), ctypes.c_long(T), ctypes.c_int(n_wires), ctypes.c_uint64(t0))
    return out
