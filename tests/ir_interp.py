"""TESTS ONLY: evaluate the product's lowered IR (fz_program_ir) with numpy.

Lets the GPU-less container check the lowering (wiring, feedback resolution, sharing, state
layout) against the oracle.  It is not part of the product: zignal_amd never interprets IR on
the CPU."""
import numpy as np

F32 = np.float32


def run_ir(prog, x, params=None, state=None):
    """x: [T, ns, n_in] float32 -> (y [T, ns, n_out], state [n_state, ns]) following the
    documented state layout: line l rows start at sum of previous depths; row+j = value at t-1-j."""
    x = np.asarray(x, F32)
    if x.ndim == 2:
        x = x[:, :, None]
    T, ns, _ = x.shape
    ir = prog.ir()
    dts = prog.ir_dtypes()
    outs = prog.outputs()
    lines = prog.lines()
    row0, r = {}, 0
    for src, depth in lines:
        row0[src] = (r, depth)
        r += depth
    if state is None:
        state = np.zeros((max(r, 1), ns), F32)
    else:
        state = np.array(state, F32, copy=True)
    y = np.empty((T, ns, len(outs)), F32)
    with np.errstate(all="ignore"):
        for t in range(T):
            v = [None] * len(ir)
            for i, (kind, a, b, val) in enumerate(ir):
                if kind == "input":
                    v[i] = x[t, :, a]
                elif kind == "const":
                    v[i] = np.full(ns, val, np.float64) if dts[i] == "f64" else np.full(ns, F32(val), F32)
                elif kind == "param":
                    v[i] = np.asarray(params[a], F32)
                elif kind == "delay":
                    r0, depth = row0[a]
                    assert 1 <= b <= depth
                    v[i] = state[r0 + b - 1].copy()
                elif kind == "add":
                    v[i] = v[a] + v[b]
                elif kind == "sub":
                    v[i] = v[a] - v[b]
                elif kind == "mul":
                    v[i] = v[a] * v[b]
                elif kind == "div":
                    v[i] = v[a] / v[b]
                elif kind == "neg":
                    v[i] = -v[a]
                else:
                    raise AssertionError(kind)
            for j, o in enumerate(outs):
                y[t, :, j] = v[o]
            for src, (r0, depth) in row0.items():
                state[r0 + 1:r0 + depth] = state[r0:r0 + depth - 1].copy()
                state[r0] = v[src]
    return y, state
